#!/bin/bash
# Hazard probe: the FULL -m gpu suite, RUNS times, with every host slice registered per call by the CALLER (MODE=caller:
# tests/conftest.py's PFHE_TEST_CALLER_REGISTER wrapper) or as it is (MODE=off), optionally under glibc's MALLOC_CHECK_.
# Writes gpurun_out/hazard_<mode>.log.        usage: tools/hazard_suite_probe.sh caller|off RUNS [MALLOC_CHECK]
set -u
MODE=${1:-caller}; RUNS=${2:-4}; MC=${3:-3}
ulimit -c 0
mkdir -p gpurun_out
LOG=gpurun_out/hazard_${MODE}.log
: > "$LOG"
# (tests that assert WHICH staging path ran skip themselves under the probe: _default_paths_only in test_gpu_staging.py)
for i in $(seq 1 "$RUNS"); do
  case "$MODE" in
    caller)  export PFHE_TEST_CALLER_REGISTER=1 ;;
    off)     unset PFHE_TEST_CALLER_REGISTER ;;
  esac
  if [ "$MC" != 0 ]; then export MALLOC_CHECK_=$MC; fi
  echo "== run $i mode=$MODE MALLOC_CHECK_=$MC" >> "$LOG"
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -s 2>&1 \
    | grep -E "FAILED|passed|failed|error|Aborted|caller-register probe|Error" | head -40 >> "$LOG"
  echo "exit ${PIPESTATUS[0]}" >> "$LOG"
done
cat "$LOG"
