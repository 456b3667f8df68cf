"""Per-kernel means of rocprofv3 counter-collection CSVs: usage pmc_summarize.py <dir> [substring]
Reads every <dir>/p*/*_counter_collection.csv, keeps kernels whose name contains `substring` (default: any), writes
<dir>/summary.json {kernel: {counter: {dispatches, mean}}} and prints one line per kernel."""
import collections, csv, glob, json, sys

d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
out = collections.defaultdict(dict)
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "")
        k = k.split("(")[0] if "<" not in k else k[: k.rfind(">") + 1]
        if want in k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = {"dispatches": len(v), "mean": sum(v) / len(v)}
# tie the counters to the code they were collected from: bench.py prints `traffic: null` + the reason when the digest of the
# headline kernel's sources differs from this one (bench.kernel_sources_digest)
import os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    out["_kernel_sources_sha256"] = bench.kernel_sources_digest()
    out["_kernel_sources"] = list(bench.KERNEL_SOURCES)
except Exception as e:  # noqa: BLE001
    print("no source digest:", e)
out["_kernel_build"] = os.environ.get("LOOPS_GIT_HEAD", "")
json.dump(out, open(d + "/summary.json", "w"), indent=1)
for k, v in out.items():
    if not isinstance(v, dict):
        continue
    print(k[-110:], {c: round(x["mean"]) for c, x in v.items()})
