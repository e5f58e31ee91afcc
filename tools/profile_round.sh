set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/profile_ntt.py > $O/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/profile_ntt.py > $O/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/profile_ntt.py > $O/p3.log 2>&1
cd $R && python tools/pmc_summary.py $O/trace $O/fetch $O/write $O/r01_e_rocprof
find $O -name "*kernel_stats.csv" | head; ls $O
# keep the merge small: drop the raw per-dispatch traces
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +2M -delete; du -sh $O
