"""Experiment: gathers of a tile issued in M passes by column range (tests/perf/exp_phased_gather.hip).
Prints G gathers / s for the C2-sized problem (2^24 items, 4 MB table, two 64 MB streams)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

lib = C.CDLL(os.path.join(ROOT, "build", "variants", "libexp_phased.so"))
vp, ci = C.c_void_p, C.c_int
lib.exp_phased_gather.argtypes = [vp, vp, vp, vp, C.c_longlong, ci, ci, ci, ci, ci, ci, vp]


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


n = 1 << 24
g = torch.Generator(device="cuda")
g.manual_seed(1)
log_table = int(os.environ.get("LOG_TABLE", "20"))
tbl = 1 << log_table
table = torch.rand(tbl, device="cuda")
idx = torch.randint(0, tbl, (n,), device="cuda", dtype=torch.int32, generator=g)
vals = torch.rand(n, device="cuda")
out = torch.empty(8192 * 512, device="cuda")
ref = None
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print(f"table 2^{log_table} floats, {n} items", flush=True)
for ipt in (8, 16, 32):
    tiles = n // (512 * ipt)
    for blocks in sorted({tiles, 1024, 2048}):
        if blocks > tiles:
            continue
        for m, sync, lock in ((1, 0, 0), (2, 0, 0), (2, 1, 0), (4, 0, 0), (4, 1, 0), (8, 0, 0), (8, 1, 0), (2, 0, 1000), (4, 0, 500), (2, 0, 2000), (4, 0, 1000)):
            def run():
                rc = lib.exp_phased_gather(table.data_ptr(), idx.data_ptr(), vals.data_ptr(), out.data_ptr(), n, tbl, m, ipt, sync, blocks, lock, stream)
                assert rc == 0, rc
            out.zero_()
            run()
            torch.cuda.synchronize()
            s = float(out.double().sum())
            if ref is None:
                ref = float((vals.double() * table[idx.long()].double()).sum())
            ms = timed(run)
            print(f"ipt {ipt:2d} blocks {blocks:5d} M {m} sync {sync} lock {lock:5d}: {ms * 1e3:7.1f} us  {n / ms / 1e6:6.1f} G/s  rel.err {abs(s - ref) / abs(ref):.1e}", flush=True)
