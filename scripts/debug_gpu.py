"""Ad-hoc GPU debugging aid (not part of the product or the test suite)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S, _lib
from oracle import oracle as O

print("torch", torch.__version__, torch.cuda.get_device_name(0), "lib", _lib.lib().loops_version())
for rows, nnz, cap in ((64, 600, 64), (1 << 12, 1 << 16, 1 << 11), (1 << 14, 1 << 18, 1 << 12)):
    deg = G.powerlaw_degrees(rows, nnz, cap=cap)
    off, idx, val = G.powerlaw_csr(rows, rows, nnz, degrees=deg)
    x = G.uniform_distribution_int(rows)
    ref = O.spmv_f32(off, idx, val, x)
    csr = S.CSR.from_numpy(rows, rows, off, idx, val)
    xd = torch.from_numpy(x).cuda()
    for tile in ("256x8", "128x7"):
        plan = S.MergePathPlan(csr, tile)
        co = plan.coords(); want = O.merge_path_coords(off, *_lib.TILES[tile][1:])
        print(rows, tile, "coords ok", np.array_equal(co, want), co[:3].tolist(), want[:3].tolist())
        for variant in (0, 2):
            y = torch.full((rows,), -7.0, device="cuda")
            S.merge_path_flat(csr, xd, y, plan=plan, variant=variant)
            torch.cuda.synchronize()
            y = y.cpu().numpy()
            bad = np.flatnonzero(y != ref)
            print(f"  rows={rows} tile={tile} variant={variant} mismatches={bad.size}", bad[:8], y[bad[:8]], ref[bad[:8]])
    for sched in ("thread_mapped", "original", "work_oriented", "group_mapped", "flat_partitioned", "merge_path_flat"):
        y = S.spmv(sched, csr, xd).cpu().numpy()
        print(f"  {sched}: mismatches={(y != ref).sum()}")
