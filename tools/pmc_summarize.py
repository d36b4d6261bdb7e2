"""Turns the rocprofv3 --pmc passes of tools/pmc_probe.py (FETCH_SIZE, WRITE_SIZE, MFMA-busy; separate runs) into
profiles/<round>_pmc_summary.json.  FETCH_SIZE is doubled (gfx950 reports half of the bytes of a coalesced stream: the guide's
correction, re-checked on the stream copy whose byte count is known); WRITE_SIZE is exact.  Units of the CSV: KiB.
MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128 SIMDs per XCD), checked on the register-only loops.

    python tools/pmc_summarize.py <fetch.csv> <write.csv> <mfma.csv or -> <probe stdout log> <out.json>
"""
import csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def by_dispatch(path):
    """dispatch id -> (kernel name, {counter: value}) in launch order"""
    out = {}
    for r in rows(path):
        d = int(r["Dispatch_Id"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out.setdefault(d, (name, {}))[1][r["Counter_Name"]] = float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


def pick(seq, prefix):
    return [c for name, c in seq if name.startswith(prefix)]


def sha16(name):
    with open(os.path.join(ROOT, "mixed-precision_lu_factorization_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


fetch, write = by_dispatch(sys.argv[1]), by_dispatch(sys.argv[2])
mfma = by_dispatch(sys.argv[3]) if sys.argv[3] != "-" else None
info = None
for line in open(sys.argv[4]):
    if line.startswith("PMCINFO "):
        info = json.loads(line[8:])
res = {"how": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
              "(pass 3) --kernel-trace -- python3 tools/pmc_probe.py; CSV units KiB; FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request), "
              "WRITE_SIZE exact; launches: see the entries", "probe": info, "calibration": {}}
KB = 1024.0
sc_f, sc_w = pick(fetch, "stream_copy_kernel"), pick(write, "stream_copy_kernel")
if sc_f and sc_w:
    res["calibration"]["stream_copy_kernel"] = {"known_bytes_each_way": 2 << 30, "fetch_x2_over_known": round(2 * sc_f[-1]["FETCH_SIZE"] * KB / (2 << 30), 4),
                                                "write_over_known": round(sc_w[-1]["WRITE_SIZE"] * KB / (2 << 30), 4)}


def busy(c):
    return round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0), 4) if c.get("GRBM_GUI_ACTIVE") else None


if mfma:
    for key, prefix in (("f64_mfma_loop", "mfma_f64_rate_kernel"), ("f16_mfma_loop", "mfma_f16_rate_kernel")):
        cs = pick(mfma, prefix)
        if cs:
            res["calibration"][key + "_mfma_busy"] = busy(cs[-1])


def entry(key, prefix, idx, src):
    f, w = pick(fetch, prefix), pick(write, prefix)
    if len(f) <= idx or len(w) <= idx or not info or key not in info:
        return
    e = dict(info[key])
    e.update({"kernel": prefix, "source": src, "source_sha16": sha16(src), "fetch_bytes": int(2 * f[idx]["FETCH_SIZE"] * KB),
              "write_bytes_pmc": int(w[idx]["WRITE_SIZE"] * KB)})
    e["traffic_over_algorithmic"] = round((e["fetch_bytes"] + e["write_bytes_pmc"]) / (e["read_bytes"] + e["write_bytes"]), 4)
    e["fetch_over_algorithmic_reads"] = round(e["fetch_bytes"] / e["read_bytes"], 4)
    if mfma:
        mm = pick(mfma, prefix)
        if len(mm) > idx:
            e["mfma_busy"] = busy(mm[idx])
    res[key] = e


# launch order inside the probe: dgemm k256, k1024; hgemm_big plain/split at k512, then k1024
entry("dgemm_k256", "dgemm_minus_kernel8d", 0, "trailing_f64.hip")
entry("dgemm_k1024", "dgemm_minus_kernel8d", 1, "trailing_f64.hip")
entry("hgemm_big_k512_plain", "hgemm16_big_kernel<true", 0, "hgemm16.hip")     # (round 5: the plain-operand kernel on v_mfma_f32_16x16x32_f16)
entry("hgemm_big_k1024_plain", "hgemm16_big_kernel<true", 1, "hgemm16.hip")
entry("hgemm_big_k512_split", "hgemm_big_kernel<true", 0, "trailing_f16.hip")
entry("hgemm_big_k1024_split", "hgemm_big_kernel<true", 1, "trailing_f16.hip")
if "dgemm_k256" in res:   # the name bench.py looks up for roofline.traffic
    d = res["dgemm_k256"]
    res["dgemm_minus_kernel"] = {"source": d["source"], "source_sha16": d["source_sha16"], "algorithmic_read_bytes": d["read_bytes"],
                                 "algorithmic_write_bytes": d["write_bytes"], "fetch_bytes": d["fetch_bytes"], "write_bytes": d["write_bytes_pmc"]}
with open(sys.argv[5], "w") as fo:
    json.dump(res, fo, indent=1)
print(json.dumps(res, indent=1))
