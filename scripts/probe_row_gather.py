"""Row-gather bandwidth of the memory system (the SpMM's B access pattern in isolation):
random rows of 32..1024 bytes from tables of 1 MB (L2-resident) .. 1 GB (HBM), 2^24 rows gathered."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loops_amd import probes as S

def ev(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))

count = 1 << 24
out = {}
g = torch.Generator(device="cuda"); g.manual_seed(1)
for row_floats in (8, 16, 32, 64, 128, 256):
    for table_mb in (1, 32, 256, 1024):
        nrows = table_mb * (1 << 20) // (row_floats * 4)
        table = torch.ones(nrows * row_floats, device="cuda")
        idx = torch.randint(0, nrows, (count,), device="cuda", dtype=torch.int32, generator=g)
        for blocks in (256 * 8,):
            o = torch.zeros(blocks * 256, device="cuda")
            ms = ev(lambda: S.row_gather(table, idx, row_floats, blocks, o))
            gb = count * row_floats * 4 / ms / 1e6
            out[f"row {row_floats*4} B, table {table_mb} MB"] = {"ms": round(ms, 4), "GBps": round(gb, 1), "Grows_per_s": round(count / ms / 1e6, 2)}
            print(f"row {row_floats*4:5d} B  table {table_mb:5d} MB  {ms*1e3:8.1f} us  {gb:8.1f} GB/s  {count/ms/1e6:7.2f} Grows/s", file=sys.stderr)
        del table
print(json.dumps(out))
