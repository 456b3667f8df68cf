#!/bin/bash
# Round-6 evidence for the plan-less entries: usage scripts/prof_r06_oneshot.sh <outdir>
#  rocprofv3 --kernel-trace --stats of tests/perf/oneshot_calls.py (which kernels a plan-less call launches and what each costs).
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r --output-format csv -- python $R/tests/perf/oneshot_calls.py > $OUT/out.log 2> $OUT/stats.err
echo "oneshot stats rc=$?"
cd $R
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
