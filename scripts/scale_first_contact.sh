#!/bin/bash
# First contact with a multi-GPU node (nothing in this repository has run over RCCL / xGMI with more than one rank yet).
# usage: scripts/scale_first_contact.sh [outdir]           (from the repository root, on the node; ~10 minutes at N = 8)
#
#  1. parity of the native exchange on 2 ranks: loops_allgatherv_f32 (one ncclGroupStart / Send / Recv / GroupEnd on the
#     library's communicator) against torch.distributed.all_gather on the same ragged slices -- before any timing;
#  2. bench.py --gpus {2,4,8} with --exchange forced to EVERY candidate in turn, so a candidate that fails or hangs shows up as a
#     row with its error instead of being dropped silently by the start-up probe; then once with --exchange auto (the probe);
#  3. rocprofv3 --kernel-trace --stats of the adopted step on every rank at the largest N, summarised into ONE JSON
#     (per-rank dominant kernels and their average durations).
# Every step runs under `timeout`; a failure is recorded and the script goes on.  Output: <outdir>/first_contact.json + logs.
# FIRST_CONTACT_DRY_RUN=1: the plumbing only, on ONE GPU -- two ranks share cuda:0 over gloo on a reduced matrix (no RCCL, no
# parity leg): what `gpurun` can check of this script.
set -u
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
R=$PWD
OUT=$R/${1:-gpurun_out/first_contact}
mkdir -p "$OUT"
NGPU=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "GPUs visible: $NGPU" | tee "$OUT/summary.txt"
PORT=29617
EXTRA=""
CANDIDATES="p2p padded native-p2p p2p-chunked p2p-chunked-4 fused-stores auto"
if [ "${FIRST_CONTACT_DRY_RUN:-0}" != "0" ]; then
  EXTRA="--backend gloo --single-device --c5-log2-rows 17 --c5-log2-nnz 21 --no-one-gpu-reference"
  CANDIDATES="p2p padded no-such-exchange auto"
  NGPU=2
  echo "DRY RUN: two ranks on cuda:0 over gloo, reduced matrix" | tee -a "$OUT/summary.txt"
fi

# ---- 1. two-rank parity of the native exchange -------------------------------------------------------------------------------
cat > "$OUT/parity_native_allgatherv.py" <<'PY'
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from loops_amd import partition as P
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl")
rows = 1_000_003
bounds = np.linspace(0, rows, world + 1).astype(np.int64); bounds[1:-1] += np.arange(1, world) * 37      # ragged slices
shard = P.Shard(rank, world, int(bounds[rank]), int(bounds[rank + 1]), bounds)
want = torch.arange(rows, dtype=torch.float32, device="cuda") * 0.5
out = {"world": world}
for name, make in (("native-p2p", lambda y: P.NativeAllgatherv(y, shard)), ("p2p", lambda y: P.Allgatherv(y, shard, "p2p")),
                   ("padded", lambda y: P.Allgatherv(y, shard, "padded"))):
    y = torch.full((rows,), -1.0, device="cuda")
    y[shard.row_begin:shard.row_end] = want[shard.row_begin:shard.row_end]
    try:
        make(y).run(); torch.cuda.synchronize()
        out[name] = bool(torch.equal(y, want))
    except Exception as e:  # noqa: BLE001
        out[name] = f"{type(e).__name__}: {e}"
ok = torch.tensor([1.0 if all(v is True for k, v in out.items() if k != "world") else 0.0], device="cuda")
dist.all_reduce(ok, op=dist.ReduceOp.MIN)
if rank == 0:
    out["all_ranks_ok"] = bool(ok.item() == 1.0)
    print(json.dumps(out))
dist.destroy_process_group()
PY
if [ "$NGPU" -ge 2 ] && [ -z "$EXTRA" ]; then
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT \
      "$OUT/parity_native_allgatherv.py" > "$OUT/parity.json" 2> "$OUT/parity.err"
  echo "parity rc=$? $(tail -1 "$OUT/parity.json")" | tee -a "$OUT/summary.txt"
else
  echo "parity: skipped (one GPU / dry run)" | tee -a "$OUT/summary.txt"
fi

# ---- 2. every exchange candidate forced, per N --------------------------------------------------------------------------------
for N in 2 4 8; do
  [ "$NGPU" -ge $N ] || continue
  for X in $CANDIDATES; do
    PORT=$((PORT + 1))
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
        bench.py --gpus $N --steps 20 --warmup 5 --exchange $X $EXTRA > "$OUT/bench_n${N}_${X}.json" 2> "$OUT/bench_n${N}_${X}.err"
    echo "N=$N exchange=$X rc=$? $(grep -c '^{' "$OUT/bench_n${N}_${X}.json") record(s)" | tee -a "$OUT/summary.txt"
  done
done

# ---- 3. per-rank kernel statistics of the adopted step at the largest N -------------------------------------------------------
NMAX=1; for N in 2 4 8; do [ "$NGPU" -ge $N ] && NMAX=$N; done
if [ "$NMAX" -gt 1 ]; then
  PORT=$((PORT + 1))
  ( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_n$NMAX" -o r --output-format csv -- \
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $NMAX --master-addr 127.0.0.1 --master-port $PORT \
      "$R/bench.py" --gpus $NMAX --steps 20 --warmup 5 $EXTRA > "$OUT/bench_n${NMAX}_under_rocprof.json" 2> "$OUT/prof_n$NMAX.err" )
  echo "rocprofv3 N=$NMAX rc=$?" | tee -a "$OUT/summary.txt"
fi

python - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
rec = {"parity": None, "runs": {}, "kernel_stats_per_process": {}}
try:
    rec["parity"] = json.loads([l for l in open(os.path.join(out, "parity.json")) if l.startswith("{")][-1])
except Exception as e:  # noqa: BLE001
    rec["parity"] = f"no record ({type(e).__name__})"
for path in sorted(glob.glob(os.path.join(out, "bench_n*_*.json"))):
    name = os.path.basename(path)[6:-5]
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        lines_e = open(path[:-5] + ".err").read().strip().splitlines() if os.path.exists(path[:-5] + ".err") else []
        err = [l for l in lines_e if "Error" in l or "error:" in l or "unavailable" in l][-4:] or lines_e[-3:]
        rec["runs"][name] = {"error": err}
        continue
    d = json.loads(lines[-1])
    c = d["config"]
    rec["runs"][name] = {"metric": d["metric"], "value_GFLOPs": d["value"], "ms_per_step": d["ms_per_step"], "scaling_detail": d.get("scaling_detail"),
                         "exchange_probe_ms": c.get("allgatherv_probe_ms_per_step"), "dropped": c.get("allgatherv_candidates_dropped"),
                         "merge_path_flat_csr_same_shards": c.get("merge_path_flat_csr_same_shards"), "watchdog": c.get("watchdog"),
                         "parity_vs_oracle_bit_exact": c.get("parity_vs_oracle_bit_exact")}
for path in sorted(glob.glob(os.path.join(out, "prof_n*", "**", "*kernel_stats.csv"), recursive=True)):
    rows = list(csv.DictReader(open(path)))[:6]
    rec["kernel_stats_per_process"][os.path.relpath(path, out)] = [{"kernel": r["Name"][:120], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)} for r in rows]
json.dump(rec, open(os.path.join(out, "first_contact.json"), "w"), indent=1)
print("wrote", os.path.join(out, "first_contact.json"))
PY
