# A/B of the pipelined forward transform: four waves per SIMD (tree) against the five-wave experiment builds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_p5; rm -rf $O; mkdir -p $O
VARS=${@:-new p5 p5l}
cd $R
for v in $VARS; do
  if [ $v != new ]; then export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_$v.so; else unset PFHE_LIB_PATH; fi
  python3 -m pytest tests/test_gpu_ntt.py -m gpu -x -q -k "full_batch or 16" 2>&1 | tail -1
done
cd /tmp
for v in $VARS $VARS; do
  if [ $v != new ]; then export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_$v.so; else unset PFHE_LIB_PATH; fi
  REPS=20 python3 $R/tools/perf_passes.py 2>&1 | tail -1
done
for v in $VARS; do
  if [ $v != new ]; then export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_$v.so; else unset PFHE_LIB_PATH; fi
  for c in VALUBusy LdsBankConflict OccupancyPercent WRITE_SIZE FETCH_SIZE; do
    BATCH=2048 REPS=2 rocprofv3 --pmc $c --output-format csv -d $O/${v}_$c -- python3 $R/tools/perf_passes.py > $O/${v}_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ab_p5'
for d in sorted(glob.glob(O+'/*_*/')):
    tag=os.path.basename(d.rstrip('/'))
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not f: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].replace('void ','').replace('pfhe::(anonymous namespace)::','').replace('pfhe::','').split('(')[0]
        if 'ntt_pipe_fwd' in k and int(r['Grid_Size'])>2090000: acc[k[:60]].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()): print(f"{tag:22s} {k:42s} n={len(v):4d} avg={sum(v)/len(v):12.2f}")
PY
find $O -name "*.csv" -size +1M -delete
