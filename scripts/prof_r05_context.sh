#!/bin/bash
# Round-5 counters for the kernels behind the bench line's context figures: usage scripts/prof_r05_context.sh <outdir>
#  1. the one-shot tuned schedules on the three C3 stand-ins (scripts/pmc_c3.sh: merge_path_flat / work_oriented / group_mapped)
#  2. the C4 BCSR 4x4 MFMA kernel: kernel stats + counters
#  3. the panel-binned kernels on the host-blocked C3 stand-in and the C5 shard (scripts/pmc_panel.sh)
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT
cd $R
for w in 0 65536 -4; do bash scripts/pmc_c3.sh $1/c3_w$w $w > $OUT/c3_w$w.log 2>&1; echo "c3 window $w rc=$?"; done
cd /tmp
BCSR_SHAPES=142 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/bcsr_stats -o r --output-format csv -- python $R/tests/perf/bench_bcsr.py > $OUT/bcsr_under_rocprof.json 2> $OUT/bcsr_stats.err
echo "bcsr stats rc=$?"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  BCSR_SHAPES=142 timeout 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/bcsr_pmc/p$i -o r --output-format csv -- python $R/tests/perf/bench_bcsr.py > /dev/null 2> $OUT/bcsr_pmc_p$i.err
  echo "bcsr pmc pass $i rc=$?"
done
cd $R
python scripts/pmc_summarize.py $OUT/bcsr_pmc bcsr | tail -2
for c in c3_host_blocked c5_shard; do bash scripts/pmc_panel.sh $1/panel_$c $c > $OUT/panel_$c.log 2>&1; echo "panel $c rc=$?"; done
