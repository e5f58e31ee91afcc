cd $GRAFT_REPO_ROOT
for t in 0 24 48 64 96 128 192 256 384; do
  if [ $t = 0 ]; then unset PFHE_PIPE_TILES; else export PFHE_PIPE_TILES=$t; fi
  echo "tiles=$t $(REPS=30 python tools/perf_passes.py 2>&1 | tail -1 | grep -o 'fwd_total.*')"
done
