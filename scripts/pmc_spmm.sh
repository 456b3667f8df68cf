#!/bin/bash
# PMC passes over the SpMM kernel (separate --pmc runs, kernel-trace only): usage pmc_spmm.sh <outdir> <bench_spmm args...>
# Every pass runs under `timeout`: a counter set this rocprofv3 build cannot schedule (e.g. TCC_TAG_STALL / TCC_BUSY
# together with TCC_HIT/MISS) aborts and then hangs until killed -- it once cost 25 GPU-minutes.
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; shift
mkdir -p $OUT; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o r --output-format csv -- python $R/tests/perf/bench_spmm.py --slow-width 0 "$@" > /dev/null 2> $OUT/p$i.err
done
cd $R
python - "$OUT" <<'PY'
import csv, collections, sys, glob, json
out = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/p*/r_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "spmm" in k and "fixup" not in k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = sum(v) / len(v)
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
for k, v in out.items():
    print(k[-60:])
    for c, x in sorted(v.items()):
        print(f"   {c:42s} {x:16.0f}")
PY
