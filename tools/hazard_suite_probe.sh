#!/bin/bash
# Hazard probe: the FULL -m gpu suite, RUNS times, with every host slice registered per call either by the CALLER
# (MODE=caller: tests/conftest.py's PFHE_TEST_CALLER_REGISTER wrapper) or by the LIBRARY (MODE=library:
# PFHE_STAGE_REGISTER_PAGEABLE=1), optionally under glibc's MALLOC_CHECK_.  Writes gpurun_out/hazard_<mode>.log.
# usage: tools/hazard_suite_probe.sh caller|library|off RUNS [MALLOC_CHECK]
set -u
MODE=${1:-caller}; RUNS=${2:-4}; MC=${3:-3}
ulimit -c 0
mkdir -p gpurun_out
LOG=gpurun_out/hazard_${MODE}.log
: > "$LOG"
# tests that assert WHICH staging path ran are meaningless while the probe changes the path
SKIP="not test_every_staging_path_is_the_one_meant_and_exact and not test_slice_spanning_two_registrations"
for i in $(seq 1 "$RUNS"); do
  case "$MODE" in
    caller)  export PFHE_TEST_CALLER_REGISTER=1; unset PFHE_STAGE_REGISTER_PAGEABLE ;;
    library) export PFHE_STAGE_REGISTER_PAGEABLE=1; unset PFHE_TEST_CALLER_REGISTER ;;
    off)     unset PFHE_STAGE_REGISTER_PAGEABLE PFHE_TEST_CALLER_REGISTER ;;
  esac
  if [ "$MC" != 0 ]; then export MALLOC_CHECK_=$MC; fi
  echo "== run $i mode=$MODE MALLOC_CHECK_=$MC" >> "$LOG"
  timeout 900 python -m pytest tests -q -m gpu -k "$SKIP" -p no:cacheprovider -s 2>&1 \
    | grep -E "FAILED|passed|failed|error|Aborted|caller-register probe|Error" | head -40 >> "$LOG"
  echo "exit ${PIPESTATUS[0]}" >> "$LOG"
done
cat "$LOG"
