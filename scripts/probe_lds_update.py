"""LDS update rates on one MI355X: ds_add_f32 / ds_add_u32 / plain read-add-write / ds_add_rtn_f32 under four address
patterns (loops_probes.h: loops_lds_update_rate_f32).  What panel_reduce's choice between LDS atomics and run-combining
rests on.  Prints lanes per clock and CU at the 2.4 GHz the chip reports."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loops_amd import probes as P


def ev(fn, iters=10):
    for _ in range(2):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


reps = 4096
for blocks in (256 * 2, 256 * 8):   # 8 and 32 wavefronts per CU
    out = torch.zeros(blocks * 256, device="cuda")
    for mode, mname in ((0, "ds_add_f32"), (1, "ds_add_u32"), (2, "read-add-write"), (3, "ds_add_rtn_f32"), (4, "cas loop"), (10, "ds_add_f64"), (11, "read-add-write 64"), (12, "cas loop 64")):
        row = []
        for pat, pname in ((0, "conflict-free"), (1, "hashed"), (2, "one word"), (3, "lane pairs")):
            ms = ev(lambda: P.lds_update_rate(mode, pat, reps, blocks, out))
            row.append(f"{pname} {blocks * 256 * reps / (ms * 1e-3) / 2.4e9 / 256:6.2f}")
        print(f"{blocks // 256 * 4:3d} waves/CU  {mname:15s} lanes/clk/CU: " + " | ".join(row), flush=True)
