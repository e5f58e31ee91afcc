cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; mkdir -p $O
for v in base new base new; do
  if [ $v = base ]; then export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_base.so; else unset PFHE_LIB_PATH; fi
  echo "== $v"; REPS=20 python3 $R/tools/perf_passes32.py 2>&1 | tail -1
done
for v in base new; do
  if [ $v = base ]; then export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_base.so; else unset PFHE_LIB_PATH; fi
  for c in VALUBusy LdsUtil MemUnitStalled OccupancyPercent LdsBankConflict; do
    REPS=2 rocprofv3 --pmc $c --output-format csv -d $O/${v}_$c -- python3 $R/tools/perf_passes32.py > $O/${v}_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4d'
for v in ('base','new'):
    for c in ('VALUBusy','LdsUtil','MemUnitStalled','OccupancyPercent','LdsBankConflict'):
        f=glob.glob(f'{O}/{v}_{c}/**/*counter_collection.csv',recursive=True)
        if not f: print(v,c,'no file'); continue
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            k=r['Kernel_Name']
            if 'ntt_block' in k or 'ntt_strided' in k:
                acc[k.split('(')[0][-70:]].append(float(r['Counter_Value']))
        for k,vv in acc.items(): print(v,c,k,round(sum(vv)/len(vv),2),len(vv))
PY
find $O -name "*.csv" -size +1M -delete
