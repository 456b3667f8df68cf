"""CSR SpMM on the C2 matrix (2^20 rows / 2^24 nnz power-law, fp32), B with n columns.
Tuned merge-path SpMM (with a held plan) vs the reference-shaped thread_mapped SpMM (ours and, with
--ref-gpu, the reference's own HIP build on the same GPU).  Correctness: column j of C must equal
the tuned SpMV with x = B[:, j] bit-exactly (exactly-summable inputs).
Algorithmic bytes: nnz * 8 + (rows + 1) * 4 + cols * n * 4 (B once) + rows * n * 4 (C once)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S


def ev(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.mean(ts)), float(np.median(ts))


ap = argparse.ArgumentParser()
ap.add_argument("--log2-rows", type=int, default=20)
ap.add_argument("--log2-nnz", type=int, default=24)
ap.add_argument("--widths", default="8,10,16,32,64,128")
ap.add_argument("--ref-gpu", action="store_true")
ap.add_argument("--window", type=int, default=0, help="columns drawn from a band of this width around the diagonal (0: uniform)")
ap.add_argument("--slow-width", type=int, default=32, help="largest n the thread_mapped kernels are timed at")
a = ap.parse_args()
rows = cols = 1 << a.log2_rows
nnz = 1 << a.log2_nnz
off, idx, val = G.powerlaw_csr(rows, cols, nnz, window=a.window or None)
csr = S.CSR.from_numpy(rows, cols, off, idx, val)
plan = S.MergePathPlan(csr)
rng = np.random.default_rng(5)
out = {"workload": f"CSR SpMM, power-law {rows} rows / {nnz} nnz (C2 matrix" + (f", columns in a band of {a.window}" if a.window else "") + "), fp32, B = cols x n row-major", "rows": {}}
for n in [int(w) for w in a.widths.split(",")]:
    Bh = rng.integers(1, 11, size=(cols, n)).astype(np.float32)
    B = torch.from_numpy(Bh).cuda()
    Cd = torch.empty((rows, n), device="cuda")
    abytes = nnz * 8 + (rows + 1) * 4 + cols * n * 4 + rows * n * 4
    flops = 2 * nnz * n
    avg, med = ev(lambda: S.spmm(csr, B, Cd, plan=plan))
    ok = all(torch.equal(Cd[:, j], S.spmv("merge_path_flat", csr, B[:, j].contiguous())) for j in sorted({0, n // 2, n - 1}))
    row = {"merge_path_flat": {"avg_ms": round(avg, 4), "median_ms": round(med, 4), "GFLOPs": round(flops / avg / 1e6, 1),
                               "GBps_algorithmic": round(abytes / avg / 1e6, 1), "bit_exact_vs_spmv": bool(ok)}}
    print(f"n={n:4d} merge_path_flat {avg*1e3:9.1f} us  {flops/avg/1e6:9.1f} GFLOP/s  {abytes/avg/1e6:8.1f} GB/s alg  exact={ok}", file=sys.stderr)
    if n <= a.slow_width:
        C2 = torch.empty_like(Cd)
        avg2, med2 = ev(lambda: S.spmm(csr, B, C2, schedule="thread_mapped"), iters=3, warm=1)
        row["thread_mapped (reference-shaped)"] = {"avg_ms": round(avg2, 3), "GFLOPs": round(flops / avg2 / 1e6, 1),
                                                   "equal": bool(torch.equal(C2, Cd))}
        print(f"        thread_mapped   {avg2*1e3:9.1f} us  {flops/avg2/1e6:9.1f} GFLOP/s  equal={torch.equal(C2, Cd)}", file=sys.stderr)
        so = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref", "libloops_ref_gpu.so")
        if a.ref_gpu and os.path.exists(so):
            import ctypes as C
            from loops_amd import _lib
            R = _lib.load_shared(so)
            Ch = np.zeros((rows, n), np.float32)
            ms = C.c_float()
            p = lambda t: t.ctypes.data_as(C.c_void_p)  # noqa: E731
            rc = R.refgpu_spmm_f32(C.c_long(rows), C.c_long(cols), C.c_long(nnz), p(off.astype(np.int32)), p(idx.astype(np.int32)),
                                   p(val), p(Bh), C.c_long(n), p(Ch), 2, C.byref(ms))
            row["reference HIP build, spmm::thread_mapped"] = {"rc": rc, "best_ms": round(ms.value, 3),
                                                                 "equal": bool(np.array_equal(Ch, Cd.cpu().numpy()))}
            print(f"        reference build {ms.value*1e3:9.1f} us  equal={np.array_equal(Ch, Cd.cpu().numpy())}", file=sys.stderr)
    out["rows"][f"n={n}"] = row
print(json.dumps(out))
