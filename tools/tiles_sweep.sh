# forward / inverse ms per 12 288 transforms against the tile count of the pipelined form (PFHE_PIPE_TILES); gpurun -- 'bash tools/tiles_sweep.sh'
cd $GRAFT_REPO_ROOT
for t in 0 12 24 32 48 64; do
  if [ $t = 0 ]; then unset PFHE_PIPE_TILES; else export PFHE_PIPE_TILES=$t; fi
  echo "tiles=$t $(REPS=30 python tools/perf_passes.py 2>&1 | tail -1 | grep -o 'fwd_total.*')"
done
