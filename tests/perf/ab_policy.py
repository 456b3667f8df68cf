"""Cache-policy A/B of the fused merge_path_flat kernel on C2 (and optionally a banded C2): every compiled policy
of libloops_probes.so (loops_probe_merge_path_f32: explicit sc0 / sc1 / nt bits on the col_idx stream, the values
stream and the x gather), tile kernel only, back-to-back launches between one pair of events; bit-exactness of
kernel + fix-up against policy 0.

    python tests/perf/ab_policy.py [--iters 50] [--window W] [--only 0,4,13] [--json out.json]
    python tests/perf/ab_policy.py --rows 7414866 --nnz 194109311 --window -4 --only 0,18,19,20 --no-shapes   (LDS window of x)

Run it under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ... --kernel-trace` to get the L2 hit rate per policy: the
kernels differ in one template argument (the engine's NT parameter), printed here as `nt_param`."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, probes as PR, spmv as S

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--window", type=int, default=0)
ap.add_argument("--only", default="")
ap.add_argument("--json", default="")
ap.add_argument("--log2-rows", type=int, default=20)
ap.add_argument("--log2-nnz", type=int, default=24)
ap.add_argument("--rows", type=int, default=0, help="exact row count (C3 stand-ins: 7414866)")
ap.add_argument("--nnz", type=int, default=0, help="exact nonzero count (C3 stand-ins: 194109311)")
ap.add_argument("--cap", type=int, default=1 << 14)
ap.add_argument("--no-shapes", action="store_true", help="skip the tile-shape instantiations")
ap.add_argument("--persistent", default="", help="comma-separated workgroup counts: the tiles walked by persistent workgroups, "
                "plain and with each tile's gathers issued before the next tile's stream loads (loops_probe_persistent_f32)")
args = ap.parse_args()

# --window: 0 = uniform columns, W > 0 = a band of W columns, -4 (generate.HOST_BLOCKED) = the host-blocked stand-in
rows = cols = args.rows or (1 << args.log2_rows)
deg = G.powerlaw_degrees(rows, args.nnz or (1 << args.log2_nnz), cap=args.cap)
off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, args.window or None,
                                   hosts=G.host_blocks(cols) if args.window == G.HOST_BLOCKED else None)
csr = S.CSR.from_numpy(rows, cols, off, idx, val)
x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
y = torch.empty(rows, device="cuda")
run = PR.PolicyRunner(csr)
names = PR.policies()
only = [int(t) for t in args.only.split(",") if t] or list(range(len(names)))
run.run(0, x, y)
torch.cuda.synchronize()
ref = y.clone()
out = {}
for p in only:
    y.zero_()
    run.run(p, x, y)
    ok = bool(torch.equal(y, ref))
    for _ in range(3):
        run.run(p, x, y, stages=1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        run.run(p, x, y, stages=1)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / args.iters * 1e3
    out[p] = {"name": names[p], "us": round(us, 2), "exact": ok}
    print(f"policy {p:2d} {names[p]:28s} {us:8.2f} us  exact={ok}", flush=True)
# tile shapes the product does not ship (same kernel, plain loads)
shapes = None if args.no_shapes else PR.ShapeRunner(csr)
for i, name in enumerate([] if args.no_shapes else PR.SHAPES):
    y.zero_()
    shapes.run(i, x, y)
    ok = bool(torch.equal(y, ref))
    for _ in range(3):
        shapes.run(i, x, y, stages=1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        shapes.run(i, x, y, stages=1)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / args.iters * 1e3
    out[f"shape {name}"] = {"name": "tile " + name, "us": round(us, 2), "exact": ok}
    print(f"tile shape {name:8s} {us:8.2f} us  exact={ok}", flush=True)
pers = PR.PersistentRunner(csr) if args.persistent else None
for groups in [int(t) for t in args.persistent.split(",") if t]:
    for pipelined in ((0, 1, 2, 3) if csr.nnzs % 4 == 0 else (0, 3)):
        y.zero_()
        pers.run(pipelined, groups, x, y)
        ok = bool(torch.equal(y, ref))
        for _ in range(3):
            pers.run(pipelined, groups, x, y, stages=1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            pers.run(pipelined, groups, x, y, stages=1)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / args.iters * 1e3
        name = f"persistent {groups} workgroups, " + ["plain", "next tile's streams behind the gathers", "next tile's streams once the gathers returned", "phased gathers"][pipelined]
        out[name] = {"name": name, "us": round(us, 2), "exact": ok}
        print(f"{name:70s} {us:8.2f} us  exact={ok}", flush=True)
if args.json:
    json.dump(out, open(args.json, "w"), indent=1)
