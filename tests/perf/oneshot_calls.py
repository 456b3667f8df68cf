"""The plan-less entries as a profiler sees them (scripts/prof_r06_oneshot.sh: rocprofv3 --kernel-trace --stats of this file):
50 whole calls each of merge_path_flat / work_oriented / group_mapped / thread_mapped on C2, work_oriented over a held plan, the one-shot CSC
product of the same matrix, BCSR mode "tuned" on C4 and on the hub case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S

rows = cols = 1 << 20
off, idx, val = G.powerlaw_csr(rows, cols, 1 << 24)
csr = S.CSR.from_numpy(rows, cols, off, idx, val)
x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda(); y = torch.empty(rows, device="cuda")
for sched in ("merge_path_flat", "work_oriented", "group_mapped", "thread_mapped"):
    for _ in range(50):
        S.spmv(sched, csr, x, y)
    torch.cuda.synchronize()
plan = S.MergePathPlan(csr, "256x8")
for _ in range(50):
    S.work_oriented(csr, x, y, plan=plan)     # held plan: merge_path_spmv_fused_auto<256, 8, 8> over the plan's tiles
torch.cuda.synchronize()
import scipy.sparse as sp
c = sp.csr_matrix((val, idx, off), shape=(rows, cols)).tocsc(); c.sort_indices()
dc = [torch.from_numpy(a).cuda() for a in (c.indptr.astype(np.int32), c.indices.astype(np.int32), c.data.astype(np.float32))]
for _ in range(50):
    S.csc_spmv(rows, cols, dc[0], dc[1], dc[2], x, y, tuned=True)   # one-shot CSC: count / scan / scatter / reduce
torch.cuda.synchronize()

rng = np.random.default_rng(3)
for name in ("c4", "hubs"):
    if name == "c4":
        nbr = 1 << 18
        boff, bcols, bvals = G.uniform_bcsr(nbr, nbr, 16)
    else:
        nbr = 1 << 17
        lens = np.full(nbr, 8, np.int64); lens[rng.choice(nbr, size=64, replace=False)] = 16384
        boff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        bcols = np.concatenate([np.sort(rng.choice(nbr, size=int(n), replace=False)) for n in lens]).astype(np.int32)
        bvals = (rng.integers(1, 9, size=bcols.size * 16) / 8.0).astype(np.float32)
    b = S.BCSR(4, 4, nbr * 4, nbr * 4, torch.from_numpy(boff).cuda(), torch.from_numpy(bcols).cuda(), torch.from_numpy(bvals).cuda())
    xb = torch.from_numpy(G.uniform_distribution_int(nbr * 4)).cuda(); yb = torch.empty(nbr * 4, device="cuda")
    S.bcsr_thread_mapped(b, xb, yb, mfma="tuned"); torch.cuda.synchronize()   # first call: tiles + probe
    for _ in range(50):
        S.bcsr_thread_mapped(b, xb, yb, mfma="tuned")
    torch.cuda.synchronize()
    print(name, S.bcsr_row_length_class(b))
