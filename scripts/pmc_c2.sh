#!/bin/bash
# Counter passes over bench.py's own command (C2): usage scripts/pmc_c2.sh <outdir> [tile, default 512x8+phased = what
# bench.py's autotuner picks on C2; 512x8 = the plain kernel]
# Separate --pmc runs, kernel-trace only, every pass under `timeout` (a counter set this build cannot schedule aborts
# and then hangs: NOTES.md).  Writes <outdir>/summary.json in the format bench.py's pmc_traffic() reads
# (profiles/rNN_c2_pmc_summary_<tile>[_phased].json): per kernel, the mean of every counter over its dispatches.
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; TILE=${2:-512x8+phased}
mkdir -p $OUT; cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_REQ_sum WRITE_SIZE GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o r --output-format csv -- python $R/bench.py --tile $TILE --steps 30 --warmup 5 --no-cpu-baseline --no-context --no-check > /dev/null 2> $OUT/p$i.err
  echo "pass $i ($set) rc=$?"
done
cd $R
python scripts/pmc_summarize.py $OUT merge_path
