"""Turns the two rocprofv3 --pmc passes of tools/pmc_probe.py (FETCH_SIZE, WRITE_SIZE; separate runs) into
profiles/<round>_pmc_summary.json.  FETCH_SIZE is doubled (gfx950 reports half of the bytes of a coalesced stream: the
guide's correction, re-checked on the stream copy whose byte count is known); WRITE_SIZE is exact.  Units of the CSV: KiB.

    python tools/pmc_summarize.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            out.setdefault(name, []).append(float(row["Counter_Value"]) * 1024.0)
    return out


def sha16(name):
    with open(os.path.join(ROOT, "mixed-precision_lu_factorization_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
m = n = 16384; k = 256
res = {"how": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace -- python3 tools/pmc_probe.py; CSV units KiB; "
              "FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE exact",
       "launch": "m = n = 16384, k = 256, ldc = 32768", "calibration": {}}
sc_f, sc_w = fetch.get("stream_copy_kernel", []), write.get("stream_copy_kernel", [])
if sc_f and sc_w:
    res["calibration"]["stream_copy_kernel"] = {"known_bytes_each_way": 2 << 30, "fetch_x2_over_known": round(2 * sc_f[-1] / (2 << 30), 4),
                                                "write_over_known": round(sc_w[-1] / (2 << 30), 4)}
for kern, src in (("dgemm_minus_kernel", "trailing_f64.hip"), ("hgemm_ring_kernel<false>", "trailing_f16.hip")):
    f = [v for key, vals in fetch.items() if key.startswith(kern.split("<")[0]) for v in vals]
    w = [v for key, vals in write.items() if key.startswith(kern.split("<")[0]) for v in vals]
    if not f or not w:
        continue
    opb = 8 if kern.startswith("dgemm") else 2
    res[kern.split("<")[0]] = {"source": src, "source_sha16": sha16(src), "algorithmic_read_bytes": m * n * 8 + (m + n) * k * opb,
                               "algorithmic_write_bytes": m * n * 8, "fetch_bytes": int(2 * max(f)), "write_bytes": int(max(w))}
with open(sys.argv[3], "w") as fo:
    json.dump(res, fo, indent=1)
print(json.dumps(res, indent=1))
