#!/bin/bash
# tools/microbench12_register_hazard.hip on the GPU box under MALLOC_CHECK_=3 (every block on the brk heap: addresses reused),
# with the host's page-migration settings; MODE=full adds the single-mode controls.
ulimit -c 0
mkdir -p gpurun_out
O=gpurun_out/microbench12.txt
: > $O
( echo "THP: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>&1)  defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag 2>&1)"; \
  echo "numa_balancing: $(cat /proc/sys/kernel/numa_balancing 2>&1)  nodes: $(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l)  kernel: $(uname -r)"; \
  echo "compaction_proactiveness: $(cat /proc/sys/vm/compaction_proactiveness 2>&1)  host: $(hostname)"; \
  grep -m1 "model name" /proc/cpuinfo ) >> $O
run() { echo "== $*" >> $O; ( "$@" 2>&1 | tail -14 ) >> $O; }
B=tools/microbench12
run env MALLOC_CHECK_=3 timeout 300 $B 12000 PZD 4 24 2
run env MALLOC_CHECK_=3 timeout 300 $B 12000 PZD 14 24 2
run env MALLOC_CHECK_=3 timeout 300 $B 12000 Z 7 24 2
if [ "${MODE:-}" = full ]; then
  run env MALLOC_CHECK_=3 timeout 300 $B 12000 D 8 24 2
  run env MALLOC_CHECK_=3 timeout 300 $B 12000 P 9 24 2
  run env MALLOC_CHECK_=3 timeout 300 $B 4000 H 10 24 2
  run timeout 300 $B 12000 Z 12 24 2
fi
tools/probe_host_kinds >> $O 2>&1
cat $O
