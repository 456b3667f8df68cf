#!/bin/bash
# Counter passes (L2 hit rate, L1->L2 round trip) over bench.py's C2 command for a variant library: usage
#   scripts/pmc_exp_phased.sh <outdir> <tile> [lib.so]
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; TILE=$2
[ -n "$3" ] && export LOOPS_AMD_LIB=$R/$3
mkdir -p $OUT; cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o r --output-format csv -- python $R/bench.py --tile $TILE --steps 30 --warmup 5 --no-cpu-baseline --no-context --no-check > /dev/null 2> $OUT/p$i.err
  echo "pass $i ($set) rc=$?"
done
cd $R
python scripts/pmc_summarize.py $OUT merge_path_spmv_fused
