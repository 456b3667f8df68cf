#!/bin/bash
# the reference's unchanged work_oriented / merge_path example binaries on a matrix with scattered columns over an x of 8 MB (16 parts)
export TMPDIR=/tmp
R=$PWD
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from loops_amd import generate as G
rows, cols = 1 << 19, 1 << 21
deg = G.powerlaw_degrees(rows, 1 << 22)
off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, None)
r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(off)) + 1
with open("/tmp/scattered8.mtx", "w") as f:
    f.write("%%MatrixMarket matrix coordinate real general\n")
    f.write(f"{rows} {cols} {idx.size}\n")
    np.savetxt(f, np.column_stack([r, idx.astype(np.int64) + 1, val.astype(np.float64)]), fmt="%d %d %.6f")
print("wrote", idx.size, flush=True)
PY
cd /tmp
for ex in work_oriented merge_path group_mapped; do
  echo "== $ex"
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dropin_$ex -o r --output-format csv -- $R/build/examples/loops.spmv.$ex.f32 -m /tmp/scattered8.mtx --validate 2>&1 | grep -i "error\|elapsed\|matrix" | tail -4
  python - <<PY
import csv
for r in csv.DictReader(open("/tmp/dropin_$ex/r_kernel_stats.csv")):
    if "spmv" in r["Name"] and "fixup" not in r["Name"]: print("   kernel:", r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
