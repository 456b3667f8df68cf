#!/bin/bash
# Counter passes over the C3 stand-in (indochina-2004's shape, uniform columns): usage scripts/pmc_c3.sh <outdir> [window]
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; W=${2:-0}
mkdir -p $OUT; cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o r --output-format csv -- python $R/tests/perf/bench_schedules.py --rows 7414866 --nnz 194109311 --window $W --tag c3 --tuned-only > /dev/null 2> $OUT/p$i.err
  echo "pass $i rc=$?"
done
cd $R
python scripts/pmc_summarize.py $OUT _fused
