#!/bin/bash
# Round-5 profile set: usage scripts/prof_r05.sh <outdir>  (everything under `timeout`; counters in their own passes)
#  1. rocprofv3 --kernel-trace --stats of the default bench.py command           -> <outdir>/bench_stats/
#  2. the C2 counter passes of the headline kernel (scripts/pmc_c2.sh)            -> <outdir>/pmc_c2/summary.json
#  3. kernel stats + counters of the row-band kernels, one directory per case    -> <outdir>/rowband_<case>/
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -o r --output-format csv -- python $R/bench.py --steps 200 --warmup 20 > $OUT/bench_under_rocprof.json 2> $OUT/bench_stats.err
echo "bench stats rc=$?"
cd $R; bash scripts/pmc_c2.sh $1/pmc_c2 | tail -3
for c in c2 c3_band65536; do bash scripts/pmc_rowband.sh $1/rowband_$c $c > $OUT/rowband_$c.log 2>&1; echo "rowband $c rc=$?"; done
