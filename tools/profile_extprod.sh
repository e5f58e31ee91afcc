R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ep; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/perf_extprod.py > $O/log.txt 2>&1
cat $O/log.txt | grep ext-products
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ep/t/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    name = r['Name'].replace('void ', '').replace('pfhe::(anonymous namespace)::', '').replace('pfhe::', '').split('(')[0]
    print(f"{name:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):6.2f} %")
PY
rm -rf $O/t
