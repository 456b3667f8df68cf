#!/bin/bash
# L2 / fabric counters per cache policy of the fused kernel (tests/perf/ab_policy.py): usage scripts/pmc_policy.sh <outdir> [ab_policy args]
# Separate --pmc passes (4 TCC slots each), kernel-trace only, every pass under `timeout`.
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; shift
mkdir -p $OUT; cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_STREAMING_REQ_sum TCC_BUBBLE_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o r --output-format csv -- python $R/tests/perf/ab_policy.py --iters 10 "$@" > $OUT/p$i.out 2> $OUT/p$i.err
  echo "pass $i rc=$?"
done
cd $R
python scripts/pmc_summarize.py $OUT merge_path_spmv_fused
