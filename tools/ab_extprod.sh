# A/B of two builds of the library on the external product (one box): throughput, then FETCH_SIZE / WRITE_SIZE / VALUBusy
# of its kernels.  usage: bash tools/ab_extprod.sh [variant ...]   ("new" = the in-tree build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_ep; rm -rf $O; mkdir -p $O
VARS=${@:-base new}
for v in $VARS $VARS; do
  if [ $v != new ]; then export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_$v.so; else unset PFHE_LIB_PATH; fi
  echo "== $v"; python3 $R/tools/perf_extprod.py 2>&1 | grep ext-products
done
export COEFF_ONLY=1
for v in $VARS; do
  if [ $v != new ]; then export PFHE_LIB_PATH=$R/primus-fhe_amd/variants/libpfhe_hip_$v.so; else unset PFHE_LIB_PATH; fi
  for c in FETCH_SIZE WRITE_SIZE VALUBusy; do
    rocprofv3 --pmc $c --output-format csv -d $O/${v}_$c -- python3 $R/tools/perf_extprod.py > $O/${v}_$c.log 2>&1
  done
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ab_ep'
for d in sorted(glob.glob(O+'/*_*/')):
    tag=os.path.basename(d.rstrip('/'))
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not f: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('pfhe::(anonymous namespace)::','').replace('pfhe::','')
        if 'fill' in k or k.startswith('__amd'): continue
        acc[k[:60]].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()): print(f"{tag:18s} {k:62s} n={len(v):4d} avg={sum(v)/len(v):14.2f} sum={sum(v):16.1f}")
PY
find $O -name "*.csv" -size +1M -delete
