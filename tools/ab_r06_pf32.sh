# same-box alternation: tree vs a variant library on the u32 external product (fused kernels), three rounds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-pf32}
for rep in 1 2 3; do
  for which in tree $V; do
    if [ $which = tree ]; then unset PFHE_LIB_PATH; else export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_$which.so; fi
    echo "== $which"; COEFF_ONLY=1 python3 $R/tools/perf_extprod32.py 2>&1 | grep "fused"
  done
done
