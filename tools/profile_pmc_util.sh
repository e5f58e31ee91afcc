# Utilisation counters for the NTT kernels, one rocprofv3 --pmc pass per counter (never combined with traces).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/util; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep "Counter_Name" | grep -i -E "valu|memunit|ldsutil|ldsbank|occupancypercent|fetchsize|writesize" | sort -u > $O/avail.txt
for c in VALUBusy VALUUtilization MemUnitBusy MemUnitStalled LdsUtil LdsBankConflict OccupancyPercent; do
  if grep -q -w "$c" $O/avail.txt; then
    PFHE_PROFILE_BATCH=2048 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/tools/profile_ntt.py > $O/$c.log 2>&1
  fi
done
python3 - <<'PY'
import csv, glob, os, collections, json
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/util'
def short(n):
    return n.replace("void ","").replace("pfhe::(anonymous namespace)::","").replace("pfhe::","").split("(")[0]
table=collections.defaultdict(dict)
for d in sorted(glob.glob(O+'/*/')):
    c=os.path.basename(d.rstrip('/'))
    fs=glob.glob(d+'/**/*counter_collection.csv', recursive=True)
    if not fs: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        acc[(short(r['Kernel_Name']), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k,v in acc.items(): table[k][c]=sum(v)/len(v)
cols=sorted({c for v in table.values() for c in v})
lines=[f"{'kernel':58s} {'grid':>10s} "+" ".join(f"{c:>16s}" for c in cols)]
for k in sorted(table, key=lambda k:(k[0],-k[1])):
    if k[0].startswith('__amd'): continue
    lines.append(f"{k[0]:58s} {k[1]:10d} "+" ".join(f"{table[k].get(c,float('nan')):16.2f}" for c in cols))
open(O+'/summary.txt','w').write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
find $O -name "*.csv" -size +1M -delete
