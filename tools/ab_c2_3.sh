# three alternations of the tree against variants/libpfhe_hip_$1.so: config 2 (N = 2^14) and the headline shape
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; V=$R/primus-fhe_amd/variants/libpfhe_hip_$1.so
for rep in 1 2 3; do
  for which in tree variant; do
    if [ $which = variant ]; then export PFHE_LIB_PATH=$V; else unset PFHE_LIB_PATH; fi
    echo -n "$which c2: "; python3 $R/tools/perf_config2.py 2>&1 | grep -o "forward: [0-9.]* ms\|inverse: [0-9.]* ms" | paste - -
    echo -n "$which n16: "; REPS=30 python3 $R/tools/perf_passes.py 2>&1 | tail -1 | grep -o "fwd_total=[^ ]*\|inv_total=[^ ]*" | paste - -
  done
done
