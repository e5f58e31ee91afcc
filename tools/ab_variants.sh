# A/B of builds of the library on one box: bash tools/ab_variants.sh NAME [NAME ...]   ("" = the in-tree build)
# (variants: tools/build_variant.sh NAME [flags] -> primus-fhe_amd/variants/libpfhe_hip_NAME.so)
for v in "$@"; do
  if [ -n "$v" ] && [ "$v" != "default" ]; then export PFHE_LIB_PATH=$GRAFT_REPO_ROOT/primus-fhe_amd/variants/libpfhe_hip_$v.so; else unset PFHE_LIB_PATH; fi
  echo "== variant: ${v:-default}"
  REPS=20 python tools/perf_passes.py 2>&1 | tail -1
done
