# rocprofv3 kernel trace + HBM traffic counters (separate --pmc passes) of tools/perf_elementwise.py;
# run on the GPU box:  gpurun -- bash tools/profile_elementwise.sh ; the summary lands in gpurun_out/elementwise/
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/elementwise; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/perf_elementwise.py > $O/perf.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/perf_elementwise.py > $O/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/perf_elementwise.py > $O/p3.log 2>&1
cd $R && python tools/pmc_summary.py $O/trace $O/fetch $O/write $O/elementwise_rocprof
cat $O/perf.log | tail -14
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +2M -delete; du -sh $O
