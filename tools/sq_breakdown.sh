# Where a wave's cycles go (MI355X_MICROARCH.md: SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES): one
# counter per rocprofv3 pass over tools/perf_passes.py (forward / inverse passes and the pipelined transforms).
# usage (GPU box): bash tools/sq_breakdown.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-sq}; rm -rf $O; mkdir -p $O
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE; do
  BATCH=2048 REPS=2 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/tools/perf_passes.py > $O/$c.log 2>&1
done
python3 - ${1:-sq} <<'PY'
import csv,glob,collections,os,sys
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'+(sys.argv[1] if len(sys.argv)>1 else 'sq')
tab=collections.defaultdict(dict)
for d in sorted(glob.glob(O+'/*/')):
    c=os.path.basename(d.rstrip('/'))
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not f: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].replace('void ','').replace('pfhe::(anonymous namespace)::','').replace('pfhe::','').split('(')[0]
        if 'fill' in k or k.startswith('__amd'): continue
        acc[(k[:44], int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k,v in acc.items(): tab[k][c]=sum(v)/len(v)
cols=sorted({c for v in tab.values() for c in v})
out=[f"{'kernel':44s} {'grid':>9s} "+" ".join(f"{c.replace('SQ_',''):>15s}" for c in cols)]
for k in sorted(tab):
    out.append(f"{k[0]:44s} {k[1]:9d} "+" ".join(f"{tab[k].get(c,float('nan')):15.4g}" for c in cols))
    t=tab[k]
    if 'SQ_WAVE_CYCLES' in t and t['SQ_WAVE_CYCLES']:
        w=t['SQ_WAVE_CYCLES']
        out.append(f"{'':44s} {'':9s}   share of wave cycles: parked (WAIT_ANY) {t.get('SQ_WAIT_ANY',0)/w:.3f}  issue-stalled (WAIT_INST_ANY) {t.get('SQ_WAIT_INST_ANY',0)/w:.3f}  issuing (ACTIVE_INST_ANY) {t.get('SQ_ACTIVE_INST_ANY',0)/w:.3f}  of which VALU {t.get('SQ_ACTIVE_INST_VALU',0)/w:.3f} LDS {t.get('SQ_ACTIVE_INST_LDS',0)/w:.3f} VMEM {t.get('SQ_ACTIVE_INST_VMEM',0)/w:.3f} scalar {t.get('SQ_ACTIVE_INST_SCA',0)/w:.3f}")
open(O+'/summary.txt','w').write("\n".join(out)+"\n")
print("\n".join(out))
PY
find $O -name "*.csv" -size +1M -delete
