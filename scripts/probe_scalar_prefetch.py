"""Does a scalar (SMEM) prefetch add memory-level parallelism on top of the vector L1's ~94 outstanding reads per CU?
Read-only stream of 1 GiB (beyond the Infinity Cache) and of 128 MiB (inside it), 16 B per lane, with wave-uniform
dword touches D steps (KB) ahead; GB/s per (distance, touch stride, resident waves per CU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loops_amd import probes as PR

def ev(fn, iters=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

sink = torch.zeros(16, device="cuda")
for mib in (1024, 128):
    src = torch.empty(mib << 18, dtype=torch.float32, device="cuda").normal_()
    for waves in (32, 16, 8):
        row = []
        for d, lw in ((0, 32), (1, 32), (2, 32), (4, 32), (8, 32), (16, 32), (2, 16), (4, 16), (8, 16)):
            ms = ev(lambda: PR.stream_read_prefetch(src, sink, d, lw, waves))
            row.append(f"D={d}/{lw*4}B {src.numel()*4/ms/1e6:7.0f}")
        print(f"{mib:5d} MiB, {waves:2d} waves/CU: " + "  ".join(row), flush=True)
    del src
