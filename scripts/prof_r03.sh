#!/bin/bash
# Round-3 profile set: usage scripts/prof_r03.sh <outdir>  (everything under `timeout`; counters in their own passes)
#  1. rocprofv3 --kernel-trace --stats of the default bench.py command                      -> <outdir>/bench_stats/
#  2. the C2 counter passes of the headline kernel (scripts/pmc_c2.sh)                       -> <outdir>/pmc_c2/summary.json
#  3. the same fabric-side counters for the column-blocked copy of C2                        -> <outdir>/pmc_c2_blocked/summary.json
#  4. kernel stats + counters of the panel-binned kernels on a C5 shard and the C3 stand-in  -> <outdir>/panel_stats/, panel_pmc/
#  5. kernel stats of the BCSR kernels of every compiled shape                               -> <outdir>/bcsr_stats/
#  6. kernel stats of the schedules on C2 (flat_partitioned_stitched_spmv among them)       -> <outdir>/sched_stats/
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/bench_stats -o r --output-format csv -- python $R/bench.py --steps 200 --warmup 20 > $OUT/bench_under_rocprof.json 2> $OUT/bench_stats.err
echo "bench stats rc=$?"
cd $R; bash scripts/pmc_c2.sh $1/pmc_c2; cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/pmc_c2_blocked/p$i -o r --output-format csv -- python $R/bench.py --layout blocked --steps 30 --warmup 5 --no-cpu-baseline --no-context --no-check > /dev/null 2> $OUT/pmc_c2_blocked_p$i.err
  echo "blocked pmc pass $i rc=$?"
done
cd $R; python scripts/pmc_summarize.py $OUT/pmc_c2_blocked merge_path; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/panel_stats -o r --output-format csv -- python $R/tests/perf/bench_panel.py c2 c5_shard c3_uniform > $OUT/panel_under_rocprof.json 2> $OUT/panel_stats.err
echo "panel stats rc=$?"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_sum WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/panel_pmc/p$i -o r --output-format csv -- python $R/tests/perf/bench_panel.py c5_shard > /dev/null 2> $OUT/panel_pmc_p$i.err
  echo "panel pmc pass $i rc=$?"
done
cd $R; python scripts/pmc_summarize.py $OUT/panel_pmc panel; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/bcsr_stats -o r --output-format csv -- python $R/tests/perf/bench_bcsr_shapes.py > $OUT/bcsr_shapes_under_rocprof.json 2> $OUT/bcsr_stats.err
echo "bcsr stats rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/sched_stats -o r --output-format csv -- python $R/tests/perf/bench_schedules.py > $OUT/schedules_under_rocprof.json 2> $OUT/sched_stats.err
echo "sched stats rc=$?"
cd $R
