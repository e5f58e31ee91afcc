cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for which in tree chain14 dittime; do
    if [ $which = tree ]; then unset PFHE_LIB_PATH; else export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_$which.so; fi
    echo "== $which"; python3 $R/tools/perf_config2.py 2>&1 | tail -2
    if [ $which != chain14 ]; then python3 $R/tools/perf_polymul.py 2>&1 | head -3; fi
  done
done
