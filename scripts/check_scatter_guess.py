import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/loops_amd") else os.getcwd())
import numpy as np, torch
sys.argv=[sys.argv[0]]
from loops_amd import generate as G, spmv as S
src=open("tests/perf/sweep_structures.py").read()
# reuse the case definitions of the sweep (up to the loop over cases)
pre=src[:src.index("so = os.path.join")]
exec(pre)
for name,(deg,cols,window) in cases.items():
    if isinstance(window, str) and window.startswith("rmat:"):
        off, idx, val = G.rmat_csr(20, 16, relabel=window[5:])
        deg = np.diff(off.astype(np.int64))
    else:
        off, idx, val = scale_free(deg, cols) if isinstance(window, str) else chunked(deg, cols, window)
    csr = S.CSR.from_numpy(deg.size, cols, off, idx, val)
    print(f"{name:50s} guess_scattered={S.columns_look_scattered(csr)}", flush=True)
    del csr
rows = cols = 7_414_866; nnz = 194_109_311
deg = G.powerlaw_degrees(rows, nnz)
for tag, window in (("uniform", None), ("band_65536", 65536), ("host_blocked", G.HOST_BLOCKED)):
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=G.host_blocks(cols) if window == G.HOST_BLOCKED else None)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    print(f"C3 stand-in {tag:38s} guess_scattered={S.columns_look_scattered(csr)}", flush=True)
    del csr
# two R-MAT graphs of C3's size (2^23 vertices x 23 edges): measured (tests/perf/bench_schedules.py --rmat 23,23,none|random): generator order
# plain 1.21 ms / phased 1.43 -> NOT scattered; labels scattered: plain 2.30 / phased 1.73 -> scattered
for rl in ("none", "random"):
    off, idx, val = G.rmat_csr(23, 23, relabel=rl)
    csr = S.CSR.from_numpy(1 << 23, 1 << 23, off, idx, val)
    print(f"R-MAT scale 23 x 23, labels {rl:27s} guess_scattered={S.columns_look_scattered(csr)}", flush=True)
    del csr
