import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loops_amd import probes as S
def ev(fn, iters=10):
    for _ in range(2): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))
blocks, reps = 256 * 8, 2048
out = torch.zeros(blocks * 256, device="cuda")
for words in (1 << 10, 1 << 13, 1 << 16, 1 << 20):
    table = torch.rand(words, device="cuda")
    row = []
    for pat, name in ((0, "consecutive"), (1, "hashed"), (2, "broadcast")):
        ms = ev(lambda: S.address_rate(table, reps, pat, blocks, out))
        lanes = blocks * 256 * reps
        row.append(f"{name} {lanes/ms/1e6:8.1f} G loads/s ({lanes/ms/1e6/ (256*2.1):.2f} per clk per CU @2.1GHz)")
    print(f"table {words*4/1024:8.1f} KiB: " + " | ".join(row), flush=True)
