# same-box A/B of a variant library against the tree: tools/ab_variants2.sh NAME [tool...]   (variants/libpfhe_hip_NAME.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; V=$R/primus-fhe_amd/variants/libpfhe_hip_$1.so; shift
TOOLS=${@:-perf_passes.py}
for rep in 1 2; do
  for which in tree variant; do
    if [ $which = variant ]; then export PFHE_LIB_PATH=$V; else unset PFHE_LIB_PATH; fi
    for t in $TOOLS; do echo -n "$which: "; REPS=20 python3 $R/tools/$t 2>&1 | tail -1; done
  done
done
