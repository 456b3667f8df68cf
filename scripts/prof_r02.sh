#!/bin/bash
# Round-2 profile set: usage scripts/prof_r02.sh <outdir>  (everything under `timeout`)
#  1. rocprofv3 --kernel-trace --stats of the default bench.py command           -> <outdir>/bench_stats/
#  2. the C2 counter passes (scripts/pmc_c2.sh)                                    -> <outdir>/pmc_c2/
#  3. rocprofv3 --kernel-trace --stats + one PMC pass of the C4 BCSR MFMA kernel   -> <outdir>/bcsr_stats/, <outdir>/bcsr_pmc/
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -o r --output-format csv -- python $R/bench.py --steps 200 --warmup 20 > $OUT/bench_under_rocprof.json 2> $OUT/bench_stats.err
echo "bench stats rc=$?"
cd $R; bash scripts/pmc_c2.sh $1/pmc_c2; cd /tmp
BCSR_SHAPES=1142 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/bcsr_stats -o r --output-format csv -- python $R/tests/perf/bench_bcsr.py > $OUT/bcsr_under_rocprof.json 2> $OUT/bcsr_stats.err
echo "bcsr stats rc=$?"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  BCSR_SHAPES=1142 timeout 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/bcsr_pmc/p$i -o r --output-format csv -- python $R/tests/perf/bench_bcsr.py > /dev/null 2> $OUT/bcsr_pmc_p$i.err
  echo "bcsr pmc pass $i rc=$?"
done
cd $R
python scripts/pmc_summarize.py $OUT/bcsr_pmc bcsr
