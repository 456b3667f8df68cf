#!/bin/bash
# round-4 experiment: wide kernel B (fp64 LDS accumulators shared by the workgroup) against the windowed one
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
out=gpurun_out/exp_wide.txt; : > $out
export PYTHONPATH=tests/perf
run() { LOOPS_PANEL_REDUCE=$1 PANEL_HW=$2 timeout 600 python tests/perf/exp_panel_reduce.py $3 2>&1 | grep -v "^$" | tee -a $out; }
run 0 0 "c2 c5_shard"
for v in 84 44 82 164; do run $v 512,1024,2048 c2; run $v 2048,4096,8192 c5_shard; done
