#!/bin/bash
# PMC passes over the self-completing merge_path_flat kernel on the FEM-like band matrix (scripts/pmc_local.py):
# usage scripts/pmc_local.sh <outdir>.  Separate --pmc runs, kernel-trace only, every pass under `timeout`.
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr" \
           "FETCH_SIZE WRITE_SIZE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o r --output-format csv -- python $R/scripts/pmc_local.py 30 > /dev/null 2> $OUT/p$i.err
done
cd $R
python - "$OUT" <<'PY'
import csv, collections, sys, glob, json
out = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/p*/r_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "merge_path_spmv_fused" in k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
for k, v in out.items():
    print(k[-70:])
    for c, x in sorted(v.items()):
        print(f"   {c:42s} {x['mean']:16.0f}  ({x['dispatches']})")
PY
