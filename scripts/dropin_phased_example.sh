#!/bin/bash
# The reference's UNCHANGED example binary (build/examples/loops.spmv.merge_path.f32 = /root/reference/examples/spmv/merge_path.cu compiled
# in place against include/loops) on a generated Matrix-Market file with scattered columns (2^20 x 2^20, 2^22 entries) and on a banded one:
# which kernel the plan-less drop-in wrapper launches (rocprofv3 kernel trace) and the elapsed time it prints.
export TMPDIR=/tmp
R=$PWD; OUT=$R/${1:-gpurun_out/dropin}; mkdir -p $OUT
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from loops_amd import generate as G
rows = cols = 1 << 20
deg = G.powerlaw_degrees(rows, 1 << 22)
for tag, window in (("scattered", None), ("band8192", 8192)):
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window)
    r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(off)) + 1
    with open(f"/tmp/{tag}.mtx", "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{rows} {cols} {idx.size}\n")
        np.savetxt(f, np.column_stack([r, idx.astype(np.int64) + 1, val.astype(np.float64)]), fmt="%d %d %.6f")
    print("wrote", tag, idx.size, flush=True)
PY
cd /tmp
for tag in scattered band8192; do
  echo "== $tag"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$tag -o r --output-format csv -- $R/build/examples/loops.spmv.merge_path.f32 -m /tmp/$tag.mtx --validate 2>&1 | grep -v "^W20\|rocprofv3" | tail -6
  python - <<PY
import csv
for r in csv.DictReader(open("$OUT/$tag/r_kernel_stats.csv")):
    if "merge_path_spmv" in r["Name"]:
        print("   kernel:", r["Name"].split("(")[0][:110], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
done
