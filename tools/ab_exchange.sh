# A/B of the LDS exchange against the in-register DPP transposition (variant build -DPFHE_EXCHANGE_DPP) on one box:
# parity of the variant, per-pass times, then VALUBusy / LdsUtil / OccupancyPercent of the block pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_xchg; rm -rf $O; mkdir -p $O
cd $R && PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_dpp.so python3 -m pytest tests/test_gpu_ntt.py -m gpu -x -q -k "forward_inverse_match_oracle or full_batch" 2>&1 | tail -2
cd /tmp
for v in new dpp new dpp; do
  if [ $v != new ]; then export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_$v.so; else unset PFHE_LIB_PATH; fi
  REPS=20 python3 $R/tools/perf_passes.py 2>&1 | tail -1
done
for v in new dpp; do
  if [ $v != new ]; then export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_$v.so; else unset PFHE_LIB_PATH; fi
  for c in VALUBusy LdsUtil OccupancyPercent; do
    BATCH=2048 REPS=2 rocprofv3 --pmc $c --output-format csv -d $O/${v}_$c -- python3 $R/tools/perf_passes.py > $O/${v}_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ab_xchg'
for d in sorted(glob.glob(O+'/*_*/')):
    tag=os.path.basename(d.rstrip('/'))
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not f: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].replace('void ','').replace('pfhe::(anonymous namespace)::','').replace('pfhe::','').split('(')[0]
        if 'ntt_block' in k or 'ntt_pipe' in k: acc[k[:60]].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()): print(f"{tag:22s} {k:62s} n={len(v):4d} avg={sum(v)/len(v):10.2f}")
PY
find $O -name "*.csv" -size +1M -delete
