"""Copy rate of one MI355X as a function of how the copy is written (loops_stream_copy_tuned_f32): vectors in flight per lane,
non-temporal loads / stores, grid-stride vs chunk per workgroup, workgroups per CU; 1 GiB (beyond the Infinity Cache) and
128 MiB.  Bytes counted: read + written.  The guide quotes 6.29 TB/s for a float4 copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loops_amd import probes as P


def ev(fn, iters=6):
    for _ in range(2):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


for mib in (1024, 128):
    n = mib * (1 << 20) // 4
    src = torch.rand(n, device="cuda"); dst = torch.empty_like(src)
    print(f"--- {mib} MiB: plain probe {2 * n * 4 / ev(lambda: P.stream_copy(src, dst)) / 1e9:.2f} TB/s; "
          f"torch copy_ {2 * n * 4 / ev(lambda: dst.copy_(src)) / 1e9:.2f} TB/s", flush=True)
    for flags in range(8):
        row = []
        for unroll in (1, 2, 4, 8):
            best = max((2 * n * 4 / ev(lambda: P.stream_copy_tuned(src, dst, unroll, flags, 256 * wpc)) / 1e9, wpc) for wpc in (2, 4, 8, 16, 32))
            row.append(f"U{unroll} {best[0]:.2f} (x{best[1]})")
        print(f"nt-load {flags & 1} nt-store {flags >> 1 & 1} chunked {flags >> 2 & 1}: " + " | ".join(row), flush=True)
    del src, dst
