for v in "" nochain "" nochain; do
  if [ -n "$v" ]; then export PFHE_LIB_PATH=$GRAFT_REPO_ROOT/primus-fhe_amd/variants/libpfhe_hip_$v.so; else unset PFHE_LIB_PATH; fi
  echo "== variant: ${v:-default}"
  PFHE_DISABLE_PERSIST=1 python tools/perf_config2.py 2>&1 | grep config2
  python tools/perf_passes.py 2>&1 | tail -12
done
