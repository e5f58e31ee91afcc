# Round evidence in one GPU call: bench.py, rocprofv3 kernel trace + stats of the same command, HBM counters
# (FETCH_SIZE / WRITE_SIZE, one --pmc pass each, never combined with traces), utilisation counters of the NTT kernels and
# of the external product AT THE BENCH SHAPE (batch 1024, default chunk).  usage: bash tools/profile_round2.sh TAG
set -x
R=$GRAFT_REPO_ROOT; TAG=${1:-r03_a}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R && python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --no-cpu-baseline --ext-total 0 > $O/bench_prof.json 2> $O/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/profile_ntt.py > $O/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/profile_ntt.py > $O/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/profile_ntt.py > $O/p3.log 2>&1
for c in VALUBusy MemUnitStalled LdsUtil LdsBankConflict OccupancyPercent; do
  PFHE_PROFILE_BATCH=2048 rocprofv3 --pmc $c --output-format csv -d $O/util_$c -- python3 $R/tools/profile_ntt.py > $O/u_$c.log 2>&1
done
# external product at the bench shape: kernel trace, then one counter per pass
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ep_trace -- python3 $R/tools/perf_extprod.py > $O/ep.log 2>&1
export COEFF_ONLY=1   # counter passes: coefficient-form products only (4 x 1024 products), so sums are per product
for c in VALUBusy OccupancyPercent LdsUtil MemUnitStalled FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/ep_$c -- python3 $R/tools/perf_extprod.py > $O/epu_$c.log 2>&1
done
unset COEFF_ONLY
# the <u32> external product at its bench shape: kernel trace, then the HBM counters (coefficient form, fused kernels only)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ep32_trace -- python3 $R/tools/perf_extprod32.py > $O/ep32.log 2>&1
export COEFF_ONLY=1
for c in FETCH_SIZE WRITE_SIZE VALUBusy OccupancyPercent; do
  rocprofv3 --pmc $c --output-format csv -d $O/ep32_$c -- python3 $R/tools/perf_extprod32.py > $O/ep32u_$c.log 2>&1
done
unset COEFF_ONLY
cd $R && python tools/pmc_summary.py $O/trace $O/fetch $O/write $O/rocprof
python tools/collect_profiles2.py $TAG
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +2M -delete; du -sh $O
