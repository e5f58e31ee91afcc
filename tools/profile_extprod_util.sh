R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eputil; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in VALUBusy OccupancyPercent MemUnitStalled; do
  BATCH=512 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/tools/perf_extprod.py > $O/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/eputil'
def short(n): return n.replace("void ","").replace("pfhe::(anonymous namespace)::","").replace("pfhe::","").split("(")[0]
for d in sorted(glob.glob(O+'/*/')):
    c=os.path.basename(d.rstrip('/')); fs=glob.glob(d+'/**/*counter_collection.csv', recursive=True)
    if not fs: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])): acc[(short(r['Kernel_Name']), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        if k[0].startswith('__amd') or 'fill' in k[0]: continue
        print(f"{c:18s} {k[0]:50s} {k[1]:10d} n={len(v):4d} {sum(v)/len(v):8.2f}")
PY
rm -rf $O
