"""BCSR SpMV for the block shapes the reference ships (2x2, 3x3) plus 4x4 / 8x8, fp32 and fp64, on C4-sized inputs
(2^18 block-rows x 16 blocks, BASELINE configs[3]) and on inputs of C4's BYTE size (~295 MB: beyond the Infinity Cache):
thread-per-block-row (the reference's kernel shape) vs the coalesced lane-group kernels vs the reference's own HIP kernel
on this GPU, against the B_bcsr roofline of SURVEY 8(d).  usage: bench_bcsr_shapes.py [--explicit] [--ref]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from loops_amd import _lib, generate as G, spmv as S

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
explicit = "--explicit" in sys.argv
with_ref = "--ref" in sys.argv
ref = None
if with_ref and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libloops_ref_gpu.so")):
    ref = _lib.load_shared(os.path.join(ROOT, "oracle", "_ref", "libloops_ref_gpu.so"))


def batch_ms(fn, iters=30, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


out = {"rows": []}
for R in (2, 3, 4, 8):
    for dtype in (np.float32, np.float64):
        vb = np.dtype(dtype).itemsize
        block_bytes = R * R * vb + 4
        for tag, nbr in (("C4 block counts", 1 << 18), ("C4 bytes", max(1 << 14, int(295e6 / (16 * block_bytes))))):
            per = 16
            boff, bcols, _ = G.uniform_bcsr(nbr, nbr, per, R, R)
            nb = int(bcols.size)
            rng = np.random.default_rng(R)
            bvals = (rng.integers(1, 9, size=nb * R * R) / 8.0).astype(dtype)
            xh = G.uniform_distribution_int(nbr * R).astype(dtype)
            b = S.BCSR(R, R, nbr * R, nbr * R, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
            xd = torch.from_numpy(xh).cuda()
            y = torch.empty(nbr * R, dtype=xd.dtype, device="cuda")
            abytes = nb * block_bytes + (nbr + 1) * 4 + 2 * nbr * R * vb   # SURVEY 8d B_bcsr
            S.bcsr_thread_mapped(b, xd, y, mfma="thread")
            want = y.clone()
            row = {"shape": f"{R}x{R}", "dtype": np.dtype(dtype).name, "input": tag, "block_rows": nbr, "blocks": nb, "algorithmic_bytes": abytes}
            modes = [("thread", "thread"), ("coalesced", "coalesced"), ("tuned", "tuned")]
            if explicit:
                modes += [(f"h{h}u{u}", 100000 + 100 * h + u) for h in (1, 4, 16) for u in (1, 2, 4)]
            for name, mode in modes:
                ms = batch_ms(lambda: S.bcsr_thread_mapped(b, xd, y, mfma=mode))
                row[name] = {"ms": round(ms, 5), "GBps": round(abytes / ms / 1e6, 1), "frac": round(abytes / ms / 1e6 / 8000, 4),
                             "equal_to_thread_mapped": bool(torch.equal(y, want))}
            if ref is not None:
                p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
                yr = np.zeros(nbr * R, dtype)
                ms = C.c_float()
                rc = ref.refgpu_bcsr_spmv(R, int(dtype == np.float64), C.c_long(nbr * R), C.c_long(nbr * R), C.c_long(nbr), C.c_long(nbr),
                                          C.c_long(nb), p(boff), p(bcols), p(bvals), p(xh), p(yr), 5, C.byref(ms))
                row["reference_kernel_on_this_gpu"] = {"rc": rc, "best_ms": round(ms.value, 5), "frac": round(abytes / ms.value / 1e6 / 8000, 4),
                                                       "equal": bool(np.array_equal(yr, want.cpu().numpy()))}
            out["rows"].append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
            del b, xd, y, want
print(json.dumps(out))
