# A/B of the forward pipelined transform: plain form (tree) against the LDS-DMA form (PFHE_PIPE_DMA = pairs per workgroup)
# usage (on the GPU box): bash tools/ab_dma.sh "0 4 8 16" [pmc]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_dma; mkdir -p $O
VARS=${1:-0 8 16}
cd $R
for v in $VARS; do
  if [ $v != 0 ]; then export PFHE_PIPE_DMA=$v; else unset PFHE_PIPE_DMA; fi
  echo "PFHE_PIPE_DMA=$v parity:"; [ -n "$SKIP_PARITY" ] || timeout 900 python3 -m pytest tests/test_gpu_ntt.py -m gpu -x -q -k "pipelined_form or config3_full_batch_every or generic_primes_large" 2>&1 | tail -2
done
cd /tmp
for v in $VARS $VARS; do
  if [ $v != 0 ]; then export PFHE_PIPE_DMA=$v; else unset PFHE_PIPE_DMA; fi
  echo -n "PFHE_PIPE_DMA=$v "; REPS=20 python3 $R/tools/perf_passes.py 2>&1 | tail -1 | sed 's/.*fwd_total/fwd_total/'
done
if [ "$2" = pmc ]; then
for v in $VARS; do
  if [ $v != 0 ]; then export PFHE_PIPE_DMA=$v; else unset PFHE_PIPE_DMA; fi
  for c in VALUBusy LdsBankConflict MemUnitStalled OccupancyPercent LdsUtil; do
    BATCH=2048 REPS=2 rocprofv3 --pmc $c --output-format csv -d $O/${v}_$c -- python3 $R/tools/perf_passes.py > $O/${v}_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ab_dma'
for d in sorted(glob.glob(O+'/*_*/')):
    tag=os.path.basename(d.rstrip('/'))
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not f: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].replace('void ','').replace('pfhe::(anonymous namespace)::','').replace('pfhe::','').split('(')[0]
        if 'ntt_pipe_fwd' in k: acc[(k[:60], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()): print(f"{tag:22s} {k[0]:42s} grid {k[1]:9d} n={len(v):4d} avg={sum(v)/len(v):12.2f}")
PY
find $O -name "*.csv" -size +1M -delete
fi
