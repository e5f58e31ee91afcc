# BASELINE config 2 (N = 2^14, batch 4096, ntt_persist_kernel) under the utilisation counters and the SQ wave-cycle counters,
# one counter per rocprofv3 pass (never combined with a trace).  usage (GPU box): bash tools/pmc_config2.sh TAG
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-c2}; rm -rf $O; mkdir -p $O
for c in VALUBusy LdsUtil LdsBankConflict MemUnitStalled OccupancyPercent SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS; do
  WARM=3 REPS=6 rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $R/tools/perf_config2.py > $O/$c.log 2>&1
done
python3 - ${1:-c2} <<'PY'
import csv,glob,collections,os,sys
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'+sys.argv[1]
tab=collections.defaultdict(dict)
for d in sorted(glob.glob(O+'/*/')):
    c=os.path.basename(d.rstrip('/'))
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not f: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].replace('void ','').replace('pfhe::(anonymous namespace)::','').replace('pfhe::','').split('(')[0]
        if 'fill' in k or k.startswith('__amd'): continue
        acc[k[:48]].append(float(r['Counter_Value']))
    for k,v in acc.items(): tab[k][c]=sum(v)/len(v)
out=["rocprofv3 --pmc <one counter per pass> -- python3 tools/perf_config2.py (N = 2^14, batch 4096); averages over launches"]
for k in sorted(tab):
    t=tab[k]
    out.append(k)
    out.append("   "+"  ".join(f"{c}={t[c]:.4g}" for c in ("VALUBusy","LdsUtil","LdsBankConflict","MemUnitStalled","OccupancyPercent") if c in t))
    w=t.get('SQ_WAVE_CYCLES')
    if w:
        out.append(f"   share of wave cycles: parked (WAIT_ANY) {t.get('SQ_WAIT_ANY',0)/w:.3f}  ready-not-issued (WAIT_INST_ANY) {t.get('SQ_WAIT_INST_ANY',0)/w:.3f}  (waiting on LDS issue {t.get('SQ_WAIT_INST_LDS',0)/w:.3f})  issuing {t.get('SQ_ACTIVE_INST_ANY',0)/w:.3f}: VALU {t.get('SQ_ACTIVE_INST_VALU',0)/w:.3f} LDS {t.get('SQ_ACTIVE_INST_LDS',0)/w:.3f} VMEM {t.get('SQ_ACTIVE_INST_VMEM',0)/w:.3f}")
open(O+'/summary.txt','w').write("\n".join(out)+"\n")
print("\n".join(out))
PY
find $O -name "*.csv" -size +1M -delete
