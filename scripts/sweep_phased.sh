#!/bin/bash
# Sweep of the phased-gather kernels' run-time knobs (LOOPS_PHASED_PARTS x LOOPS_PHASED_TICKS) over matrix sizes:
# usage scripts/sweep_phased.sh "<log_rows log_nnz [log_cols]>;..." "<parts list>" "<ticks list>"   (one process per point)
IFS=';' read -ra SIZES <<< "$1"
for s in "${SIZES[@]}"; do
  set -- $s
  export LOG_ROWS=$1 LOG_NNZ=$2 LOG_COLS=${3:-$1}
  unset LOOPS_PHASED_PARTS LOOPS_PHASED_TICKS PHASED_ONLY
  timeout 300 python tests/perf/ab_phased.py 2>&1 | grep -v amdgpu.ids | sed "s/ eq=True//g"
  export PHASED_ONLY=1
  for p in $PARTS; do for t in $TICKS; do
    LOOPS_PHASED_PARTS=$p LOOPS_PHASED_TICKS=$t timeout 300 python tests/perf/ab_phased.py 2>&1 | grep -v amdgpu.ids | sed "s/ eq=True//g"
  done; done
done
