#!/bin/bash
# Round-6 evidence for the block-band BCSR plan on BASELINE C4: usage scripts/prof_r06_bcsr_band.sh <outdir>
#  kernel stats (rocprofv3 --kernel-trace --stats) and three counter passes of tests/perf/bench_bcsr_band.py (automatic plan).
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT
cd /tmp
BB_HB=0 BB_CHUNKS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r --output-format csv -- python $R/tests/perf/bench_bcsr_band.py > $OUT/under_rocprof.json 2> $OUT/stats.err
echo "bcsr band stats rc=$?"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  BB_HB=0 BB_CHUNKS=0 BB_NO_TUNE=1 timeout 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc/p$i -o r --output-format csv -- python $R/tests/perf/bench_bcsr_band.py > /dev/null 2> $OUT/pmc_p$i.err
  echo "bcsr band pmc pass $i rc=$?"
done
cd $R
python scripts/pmc_summarize.py $OUT/pmc bcsr_band | tail -2
