# SQ-level counters of the forward block / strided passes (one rocprofv3 --pmc pass per counter group, never combined
# with traces).  usage (on the GPU box): bash tools/profile_sq.sh TAG [PFHE_LIB_PATH]
R=$GRAFT_REPO_ROOT; TAG=${1:-cur}; O=$R/gpurun_out/sq_$TAG; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
[ -n "$2" ] && export PFHE_LIB_PATH=$2
rocprofv3 -L > $O/avail.txt 2>&1
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
G2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"
G4="GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_WAVE32_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_MFMA SQ_CYCLES"
i=0
for G in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  BATCH=2048 rocprofv3 --pmc $G --output-format csv -d $O/g$i -- python3 $R/tools/profile_block.py > $O/g$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/sq_'+os.environ.get('TAG_','cur')
PY
TAG_=$TAG python3 - <<'PY'
import csv, glob, os, collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/sq_'+os.environ['TAG_']
def short(n):
    return n.replace("void ","").replace("pfhe::(anonymous namespace)::","").replace("pfhe::","").split("(")[0]
table=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=(short(r['Kernel_Name']), int(r['Grid_Size']))
        table[k][r['Counter_Name']].append(float(r['Counter_Value']))
out=[]
for k in sorted(table):
    if k[0].startswith('__amd') or 'fill' in k[0]: continue
    out.append(f"{k[0]} grid={k[1]}")
    for c,v in sorted(table[k].items()):
        out.append(f"    {c:28s} {sum(v)/len(v):18.1f}  (n={len(v)})")
open(O+'/summary.txt','w').write("\n".join(out)+"\n")
print("\n".join(out))
PY
find $O -name "*.csv" -size +1M -delete
