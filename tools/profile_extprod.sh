R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ep; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/perf_extprod.py > $O/log.txt 2>&1
cat $O/log.txt | grep ext-products
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ep/t/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'].split('(')[0][-64:], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us', r['Percentage'])
PY
rm -rf $O/t
