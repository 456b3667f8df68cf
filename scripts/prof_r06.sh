#!/bin/bash
# Round-6 profile set: usage scripts/prof_r06.sh <outdir>  (everything under `timeout`; counters in their own passes)
#  1. rocprofv3 --kernel-trace --stats of the default bench.py command (--steps 200)   -> <outdir>/bench_stats/
#  2. the block-band BCSR plan on C4: kernel stats + three counter passes (scripts/prof_r06_bcsr_band.sh) -> <outdir>/bcsr_band/
#  3. group_mapped on the R-MAT stand-in: kernel stats                                  -> <outdir>/group_mapped_rmat/
# The C2 counters of the headline kernel are the round-5 ones (profiles/r05_c2_pmc_summary_512x8*.json): its sources did not
# change (bench.py checks the digest: roofline.counters.digest_matches_head).
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -o r --output-format csv -- python $R/bench.py --steps 200 --warmup 20 > $OUT/bench_under_rocprof.json 2> $OUT/bench_stats.err
echo "bench stats rc=$?"
cd $R; bash scripts/prof_r06_bcsr_band.sh $1/bcsr_band | tail -4
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/group_mapped_rmat -o r --output-format csv -- python $R/tests/perf/bench_group_mapped.py rmat > $OUT/group_mapped_rmat.json 2> $OUT/group_mapped_rmat.err
echo "group_mapped stats rc=$?"
