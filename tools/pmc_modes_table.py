"""Joins the PMCORDER line of tools/pmc_hgemm_modes.py with a rocprofv3 counter_collection.csv: one row per (tile, mode) with
every counter of the pass (second launch of each pair).  usage: pmc_modes_table.py <counter csv> <probe log> [out.json]"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
order = None
for line in open(sys.argv[2]):
    if line.startswith("PMCORDER "): order = json.loads(line[9:])
disp = {}
for r in rows:
    if "hgemm_big_kernel" not in r["Kernel_Name"]: continue
    disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)[order["skip_first_big"]:]
assert len(ids) == len(order["launches"]), (len(ids), len(order["launches"]))
out = []
for i, (d, l) in enumerate(zip(ids, order["launches"])):
    if i % 2 == 1: out.append({**l, **disp[d]})
for o in out: print(json.dumps(o))
if len(sys.argv) > 3: json.dump({"probe": {k: order[k] for k in ("m", "k", "split")}, "rows": out}, open(sys.argv[3], "w"), indent=1)
