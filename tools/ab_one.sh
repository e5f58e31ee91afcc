# A/B of the forward transform at the bench shape: tree (tiles + 1 launches) against the ONE-launch experiment
# (PFHE_PIPE_ONE = lag in workgroups, PFHE_PIPE_ONE_MODE: bit 0 coherent intermediate, bit 1 flags)
# usage (GPU box): bash tools/ab_one.sh "lag:mode ..." [parity]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
VARS=${1:-0:0 8192:0 8192:3 2048:3}
for rep in 1 2; do
for v in $VARS; do
  lag=${v%%:*}; mode=${v##*:}
  if [ $lag != 0 ]; then export PFHE_PIPE_ONE=$lag PFHE_PIPE_ONE_MODE=$mode; else unset PFHE_PIPE_ONE PFHE_PIPE_ONE_MODE; fi
  echo -n "lag=$lag mode=$mode "; REPS=20 python3 $R/tools/perf_passes.py 2>&1 | tail -1 | sed 's/.*fwd_total/fwd_total/'
done
done
if [ "$2" = parity ]; then
cd $R
for v in $VARS; do
  lag=${v%%:*}; mode=${v##*:}
  if [ $lag != 0 ]; then export PFHE_PIPE_ONE=$lag PFHE_PIPE_ONE_MODE=$mode; else unset PFHE_PIPE_ONE PFHE_PIPE_ONE_MODE; fi
  echo "lag=$lag mode=$mode parity:"; timeout 900 python3 -m pytest tests/test_gpu_ntt.py -m gpu -x -q -k "pipelined_form_tile or config3_full_batch_every" 2>&1 | tail -2
done
fi
