"""A/B of a library variant over SpMV workloads with and without column locality.
usage: ab_xcd.py gen | run   (gen writes matrices under /tmp/ab; run times the lib selected by LOOPS_AMD_LIB)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
CASES = [("c2 uniform", 20, 24, 0), ("c2 band 65536", 20, 24, 65536), ("c2 band 4096", 20, 24, 4096),
         ("4M/64M uniform", 22, 26, 0), ("4M/64M band 2^18", 22, 26, 1 << 18), ("4M/64M band 2^15", 22, 26, 1 << 15)]
os.makedirs("/tmp/ab", exist_ok=True)
if sys.argv[1] == "gen":
    from loops_amd import generate as G
    for i, (name, lr, ln, w) in enumerate(CASES):
        rows = 1 << lr
        deg = G.powerlaw_degrees(rows, 1 << ln)
        parts, bounds = [], np.linspace(0, rows, max(1, (1 << ln) >> 25) + 1).astype(np.int64)
        for a, b in zip(bounds[:-1], bounds[1:]):
            parts.append(G.csr_from_degrees(deg[a:b], rows, 1, int(a), True, w or None))
        off = np.concatenate([[0]] + [p[0][1:].astype(np.int64) + sum(int(q[0][-1]) for q in parts[:k]) for k, p in enumerate(parts)]).astype(np.int32)
        np.save(f"/tmp/ab/{i}_off.npy", off); np.save(f"/tmp/ab/{i}_idx.npy", np.concatenate([p[1] for p in parts])); np.save(f"/tmp/ab/{i}_val.npy", np.concatenate([p[2] for p in parts]))
else:
    import torch
    from loops_amd import generate as G, spmv as S
    def ev(fn, iters=30, warm=3):
        for _ in range(warm): fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs]))
    for i, (name, lr, ln, w) in enumerate(CASES):
        rows = 1 << lr
        csr = S.CSR.from_numpy(rows, rows, np.load(f"/tmp/ab/{i}_off.npy"), np.load(f"/tmp/ab/{i}_idx.npy"), np.load(f"/tmp/ab/{i}_val.npy"))
        x = torch.from_numpy(G.uniform_distribution_int(rows)).cuda()
        plan = S.MergePathPlan(csr)
        y = torch.empty(rows, device="cuda")
        t = {"merge_path_flat": ev(lambda: S.merge_path_flat(csr, x, y, plan=plan))}
        ref = y.clone()
        for sch in ("work_oriented", "group_mapped"):
            t[sch] = ev(lambda: S.spmv(sch, csr, x, y))
            assert torch.equal(y, ref), sch
        print(f"{name:20s} " + "  ".join(f"{k} {v*1e3:8.1f} us" for k, v in t.items()), flush=True)
