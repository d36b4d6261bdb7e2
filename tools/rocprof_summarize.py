"""profiles/rNN_rocprof_summary.json from a rocprofv3 --kernel-trace --stats kernel-stats CSV of `python3 bench.py`: per kernel calls and
average duration, with the sha of every kernel source, so that bench.py quotes the figure only for the source it was measured on.
usage: rocprof_summarize.py <kernel_stats.csv> <out.json> [n nb]"""
import csv, hashlib, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "mixed-precision_lu_factorization_amd", "csrc")
shas = {f: hashlib.sha256(open(os.path.join(src, f), "rb").read()).hexdigest()[:16] for f in sorted(os.listdir(src))}
kern = {}
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Name"]
    short = name.split("(")[0].replace("void ", "").strip()
    kern[short] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "total_ns": int(row["TotalDurationNs"]), "pct": float(row["Percentage"])}
out = {"n": int(sys.argv[3]) if len(sys.argv) > 3 else 32768, "nb": int(sys.argv[4]) if len(sys.argv) > 4 else 256,
       "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 (see tools/profile_round.sh)",
       "sources_sha16": shas, "kernels": kern}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2], len(kern), "kernels")
