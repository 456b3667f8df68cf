"""Cases and timing helpers shared by bench_panel.py and exp_panel_reduce.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from loops_amd import generate as G


def build_ms(make):
    """Wall time of one plan creation (device work included), second of two creations (the first pays allocator warm-up)."""
    import time
    make().close()
    torch.cuda.synchronize()
    t = time.perf_counter()
    plan = make()
    torch.cuda.synchronize()
    return plan, (time.perf_counter() - t) * 1e3


def batch_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


CASES = {
    "c2": (1 << 20, 1 << 20, 1 << 24, None),
    "c5_shard": (1 << 21, 1 << 24, 1 << 26, None),
    "c3_uniform": (7_414_866, 7_414_866, 194_109_311, None),
    "c3_host_blocked": (7_414_866, 7_414_866, 194_109_311, G.HOST_BLOCKED),
    "c3_band65536": (7_414_866, 7_414_866, 194_109_311, 65536),
    "short_rows_8M": (1 << 23, 1 << 23, 1 << 24, None),
}
