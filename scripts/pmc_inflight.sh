#!/bin/bash
# usage: scripts/pmc_inflight.sh <outdir>
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1
mkdir -p $OUT; cd /tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "GRBM_GUI_ACTIVE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o r --output-format csv -- python $R/scripts/probe_inflight.py > /dev/null 2> $OUT/p$i.err
  echo "pass $i rc=$?"
done
cd $R
python - "$OUT" <<'PY'
import csv, collections, glob, json, sys
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "") + f" grid={r.get('Grid_Size', '?')}"
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in out.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if "GRBM_GUI_ACTIVE" in m and "TCP_TCC_READ_REQ_LATENCY_sum" in m and m.get("TCP_TCC_READ_REQ_sum", 0) > 0:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        m["reads_in_flight_per_CU"] = m["TCP_TCC_READ_REQ_LATENCY_sum"] / cyc / 256
        m["avg_latency_clks"] = m["TCP_TCC_READ_REQ_LATENCY_sum"] / m["TCP_TCC_READ_REQ_sum"]
        m["pending_stall_frac"] = m["TCP_PENDING_STALL_CYCLES_sum"] / m["TCP_GATE_EN1_sum"]
        m["L2_requests_per_clk_per_XCD"] = m.get("TCC_REQ_sum", 0) / 8 / cyc
    res[k] = m
    print(k[-80:], {c: round(x, 2) for c, x in m.items() if c in ("reads_in_flight_per_CU", "avg_latency_clks", "pending_stall_frac", "L2_requests_per_clk_per_XCD")})
json.dump(res, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
