#!/bin/bash
# HBM-traffic PMC passes over bench.py's own command (C2, tile 512x8): usage scripts/pmc_c2.sh <outdir>.
# Separate --pmc runs, kernel-trace only, every pass under `timeout`.  Writes <outdir>/summary.json in the format
# bench.py's pmc_traffic() reads (profiles/r01_c2_pmc_summary_512x8.json).
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT; cd /tmp
i=0
for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o r --output-format csv -- python $R/bench.py --tile 512x8 --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/p$i.err
done
cd $R
python - "$OUT" <<'PY'
import csv, collections, sys, glob, json
out = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/p*/r_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "merge_path" in k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
for k, v in out.items():
    print(k[-70:], {c: round(x["mean"]) for c, x in v.items()})
PY
