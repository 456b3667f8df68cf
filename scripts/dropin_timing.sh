#!/bin/bash
# Drop-in wrappers on a matrix large enough to time (round 4): the reference's OWN example binaries, compiled unchanged
# against include/loops (build/examples, scripts/build_reference_examples.sh), on a generated 2^18-row / 2^22-nonzero
# power-law matrix in Matrix-Market form: the elapsed time each prints, --validate's verdict, and the kernels rocprofv3
# sees them launch.  usage: scripts/dropin_timing.sh <outdir>
export TMPDIR=/tmp
R=$PWD; OUT=$R/$1; mkdir -p $OUT
MTX=/tmp/dropin_powerlaw_2e18.mtx
python - <<PY
import sys, numpy as np, pandas as pd
sys.path.insert(0, "$R")
from loops_amd import generate as G
rows = cols = 1 << 18
deg = G.powerlaw_degrees(rows, 1 << 22, cap=1 << 12)
off, idx, val = G.csr_from_degrees(deg, cols, 1)
r = np.repeat(np.arange(rows), np.diff(off)) + 1
with open("$MTX", "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (rows, cols, idx.size))
pd.DataFrame({"r": r, "c": idx.astype(np.int64) + 1, "v": val}).to_csv("$MTX", sep=" ", header=False, index=False, mode="a")
print("wrote", "$MTX", idx.size, "entries")
# a banded matrix for DIA: 2^20 rows, 11 diagonals
n = 1 << 20
offs = np.arange(-5, 6)
rr = np.repeat(np.arange(n), offs.size)
cc = rr + np.tile(offs, n)
keep = (cc >= 0) & (cc < n)
rr, cc = rr[keep], cc[keep]
vv = ((rr * 7 + cc) % 8 + 1) / 8.0
with open("/tmp/dropin_band_2e20.mtx", "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (n, n, rr.size))
pd.DataFrame({"r": rr + 1, "c": cc + 1, "v": vv}).to_csv("/tmp/dropin_band_2e20.mtx", sep=" ", header=False, index=False, mode="a")
print("wrote band", rr.size, "entries")
PY
cd /tmp
: > $OUT/elapsed.txt
for exe in loops.spmm.thread_mapped loops.spmv.coo_thread_mapped.f32 loops.spmv.ell_thread_mapped.f32 loops.spmv.dia_thread_mapped.f32 \
           loops.spmv.csc_thread_mapped.f32 loops.spmv.merge_path.f32 loops.spmv.thread_mapped.f32; do
  M=$MTX
  case $exe in *dia*|*ell*) M=/tmp/dropin_band_2e20.mtx;; esac   # (a power-law matrix has ~2^18 distinct diagonals and a 4096-wide ELL pitch: DIA and ELL get a banded one)
  args="-m $M --validate"
  echo "== $exe" >> $OUT/elapsed.txt
  timeout 300 $R/build/examples/$exe $args 2>&1 | grep -E "^[a-z_]+,|Errors|Elapsed|elapsed" | head -5 >> $OUT/elapsed.txt
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$exe -o r --output-format csv -- $R/build/examples/$exe -m $M > /dev/null 2> $OUT/trace_$exe.err
  python - >> $OUT/elapsed.txt <<PY
import csv, glob
for f in glob.glob("$OUT/trace_$exe/**/r_kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print("   kernel", row["Name"][:150], "calls", row["Calls"], "avg_ns", row["AverageNs"])
PY
done
cd $R; cat $OUT/elapsed.txt
