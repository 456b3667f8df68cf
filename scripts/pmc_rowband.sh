#!/bin/bash
# Counter passes + kernel stats of the row-band kernels: usage scripts/pmc_rowband.sh <outdir> <case> [H] [target_chunks]
# (tests/perf/run_rowband.py; kernel shape through LOOPS_ROWBAND_CFG).  Separate --pmc runs, kernel-trace only, every pass
# under `timeout`.
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; shift
mkdir -p $OUT; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r --output-format csv -- python $R/tests/perf/run_rowband.py "$@" > $OUT/run.txt 2> $OUT/stats.err
echo "stats rc=$?"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_EA0_WRREQ_sum WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o r --output-format csv -- python $R/tests/perf/run_rowband.py "$@" > /dev/null 2> $OUT/p$i.err
  echo "pmc pass $i ($set) rc=$?"
done
cd $R; python scripts/pmc_summarize.py $OUT rowband
