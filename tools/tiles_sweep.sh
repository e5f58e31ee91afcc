cd $GRAFT_REPO_ROOT
python tools/stream_check.py
PFHE_PIPE_LAG2=1 PFHE_PIPE_TILES=192 python tools/stream_check.py
BATCH=701 PFHE_PIPE_LAG2=1 PFHE_PIPE_TILES=37 python tools/stream_check.py
BATCH=701 python tools/stream_check.py
for lag2 in 0 1; do for t in 24 48 96 192 384; do
  export PFHE_PIPE_TILES=$t; if [ $lag2 = 1 ]; then export PFHE_PIPE_LAG2=1; else unset PFHE_PIPE_LAG2; fi
  echo "lag2=$lag2 tiles=$t $(REPS=30 python tools/perf_passes.py 2>&1 | tail -1 | grep -o 'fwd_total.*')"
done; done
