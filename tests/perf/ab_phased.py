"""A/B aid for the phased-gather kernel (LOOPS_VARIANT_PHASED): C2 (or 2^LOG_ROWS rows / 2^LOG_NNZ nnz of the same generator),
tile kernel alone and whole step, plain against phased, for the two shapes that have a phased twin.  Run once per library
(LOOPS_AMD_LIB) and compare.  Prints us per launch (back-to-back batch between one event pair) + bit-equality."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S, _lib

def batch(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

lr, ln = int(os.environ.get("LOG_ROWS", "20")), int(os.environ.get("LOG_NNZ", "24"))
rows = 1 << lr
cols = 1 << int(os.environ.get("LOG_COLS", str(lr)))
nnz_target = int(os.environ.get("NNZ", str(1 << ln)))
deg = G.powerlaw_degrees(rows, nnz_target)
off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, None)
f64 = os.environ.get("DTYPE", "f32") == "f64"
dt = np.float64 if f64 else np.float32
csr = S.CSR.from_numpy(rows, cols, off, idx, val.astype(dt))
x = torch.from_numpy(G.uniform_distribution_int(cols).astype(dt)).cuda()
y = torch.empty(rows, device="cuda", dtype=x.dtype); y0 = torch.empty(rows, device="cuda", dtype=x.dtype)
out = []
tiles = os.environ.get("TILES", "512x8,256x16").split(",")
for tile in tiles:
    plan = S.MergePathPlan(csr, tile)
    S.merge_path_flat(csr, x, y0, plan=plan, variant=0)
    for v in ((_lib.VARIANT_PHASED,) if os.environ.get("PHASED_ONLY") else (0, _lib.VARIANT_PHASED)):
        k = min(batch(lambda: S.merge_path_flat_stage(csr, x, y, plan, 0, v)) for _ in range(3)) if not f64 else float('nan')  # (the stage entry is f32 only)
        s = min(batch(lambda: S.merge_path_flat(csr, x, y, plan=plan, variant=v)) for _ in range(3))
        out.append(f"{tile}{'+phased' if v else ''}: kernel {k:6.1f} step {s:6.1f} eq={bool(torch.equal(y, y0))}")
print(("f64 " if f64 else "") + f"2^{lr} rows x {cols} cols {nnz_target} nnz ({(rows + nnz_target + 4095) // 4096} tiles of 4096) parts={os.environ.get('LOOPS_PHASED_PARTS', 'auto')} ticks={os.environ.get('LOOPS_PHASED_TICKS', 'auto')} | " + " | ".join(out), flush=True)
