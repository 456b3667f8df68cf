import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/loops_amd") else os.getcwd())
import numpy as np, torch
from loops_amd import probes as S
def ev(fn, iters=10):
    for _ in range(2): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))
table = torch.rand(1 << 20, device="cuda")
for blocks in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    out = torch.zeros(blocks * 256, device="cuda")
    reps = 4096 * 2048 // blocks if blocks < 2048 else 2048
    reps = min(reps, 65536)
    ms = ev(lambda: S.address_rate(table, reps, 1, blocks, out))
    lanes = blocks * 256 * reps
    print(f"blocks {blocks:5d} (x256 thr) hashed 4 MB: {lanes/ms/1e6:8.1f} G loads/s", flush=True)
