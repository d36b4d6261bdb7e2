"""Per-kernel HBM-side traffic of whole factorizations from two rocprofv3 --pmc passes over tools/pmc_factor_probe.py (FETCH_SIZE in
one, WRITE_SIZE in the other; CSV unit KiB; FETCH_SIZE x 2: gfx950 counts 64 B per 128-byte request, MI355X_MICROARCH.md HBM) and
the kernel trace of the same runs (durations with one dispatch at a time).  Algorithmic bytes per kernel family are computed from
N, nb and the schedule (formulas in the output); traffic / algorithmic and GB/s = measured bytes / summed duration.
    python tools/pmc_factor_summarize.py <fetch counter csv> <fetch kernel trace csv> <write counter csv> <probe log> <out.json>"""
import csv, json, re, sys
KB = 1024.0


def short(name):
    name = name.replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)", name)
    base = m.group(1) if m else name
    tpl = re.search(r"<([^>]*)>", name)
    return base + ("<" + tpl.group(1) + ">" if tpl and len(tpl.group(1)) < 40 else "")


def counters(path, cname):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname: continue
        k = short(r["Kernel_Name"])
        e = out.setdefault(k, [0, 0.0])
        e[0] += 1; e[1] += float(r["Counter_Value"])
    return out


def durations(path):
    out = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        e = out.setdefault(k, [0, 0.0])
        e[0] += 1; e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return out


fetch, dur, write = counters(sys.argv[1], "FETCH_SIZE"), durations(sys.argv[2]), counters(sys.argv[3], "WRITE_SIZE")
info = None
for line in open(sys.argv[4]):
    if line.startswith("PMCFACTOR "): info = json.loads(line[10:])
n, nb = info["n"], info["nb"]
npan = (n + nb - 1) // nb
nfac = len(info["modes"])
# algorithmic bytes of ONE factorization, per kernel family (panel k: rows = n - k, cols = nb)
rows = [n - k * nb for k in range(npan)]
alg = {
    "hgetf2_lds_kernel": (sum(r * nb * 8 for r in rows if r > 1), "hgetf2_lds_kernel + hgetf2_win_kernel: sum over panels of rows x cols x 8 B, the fp64 panel read once (the fp16 panel stays in LDS / registers); the rest is hand-off granules (write-through) and polls"),
    "dpanel": (sum(r * nb * 16 for r in rows if r > 1), "dpanel_sub + dpanel_fused + tiles: rows x cols x 16 B per panel (read and written once)"),
    "transpose64_kernel": (None, "every byte read once and written once: fetch ~ write expected"),
    "wt_rows": (None, "2 passes (gather, scatter) over <= 2 nb moved rows x the columns right of the panel: fetch ~ write expected"),
}
res = {"how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python3 tools/pmc_factor_probe.py; FETCH_SIZE x 2 (gfx950), "
              "WRITE_SIZE exact, KiB; durations: the FETCH pass's kernel trace (one dispatch at a time, chain_pipeline = 0)",
       "probe": info, "kernels": {}}
fam = {"hgetf2_lds_kernel": ("hgetf2_lds_kernel", "hgetf2_win_kernel"), "dpanel": ("dpanel_sub_kernel", "dpanel_fused_kernel", "dpanel_update_kernel", "dpanel_tiles_store_kernel"), "wt_rows": ("wt_rows_gather_kernel", "wt_rows_scatter_kernel")}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, [0, 0.0]); w = write.get(k, [0, 0.0]); d = dur.get(k, [0, 0.0])
    fb, wb = 2.0 * f[1] * KB, w[1] * KB
    e = {"launches": f[0], "fetch_bytes": round(fb), "write_bytes": round(wb), "seconds": round(d[1], 6),
         "GBps_fetch_plus_write": round((fb + wb) / d[1] / 1e9, 1) if d[1] > 0 else None,
         "avg_launch_us": round(d[1] / d[0] * 1e6, 2) if d[0] else None}
    res["kernels"][k] = e
# families against their algorithmic bytes (fp64 + fp16 factorizations both run the panel kernels: x number of factorizations)
res["families"] = {}
for name, (ab, what) in alg.items():
    members = [k for k in res["kernels"] if any(k.startswith(p) for p in fam.get(name, (name,)))]
    fb = sum(res["kernels"][k]["fetch_bytes"] for k in members); wb = sum(res["kernels"][k]["write_bytes"] for k in members)
    sec = sum(res["kernels"][k]["seconds"] for k in members)
    e = {"members": members, "fetch_bytes": fb, "write_bytes": wb, "seconds": round(sec, 6), "GBps_fetch_plus_write": round((fb + wb) / sec / 1e9, 1) if sec > 0 else None,
         "algorithmic": what}
    if ab:
        e["algorithmic_bytes"] = ab * nfac
        e["traffic_over_algorithmic"] = round((fb + wb) / (ab * nfac), 3)
    res["families"][name] = e
import hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res["sources_sha16"] = {f: hashlib.sha256(open(os.path.join(ROOT, "mixed-precision_lu_factorization_amd", "csrc", f), "rb").read()).hexdigest()[:16]
                        for f in ("fp16_panel.hip", "dpanel.hip", "laswp.hip", "trailing_f64.hip", "trailing_f16.hip", "hgemm_pp.hip", "hgemm16.hip", "ir.hip")}
json.dump(res, open(sys.argv[5], "w"), indent=1)
for name, e in res["families"].items():
    print(name, {k: v for k, v in e.items() if k not in ("members", "algorithmic")})
top = sorted(res["kernels"].items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"]))[:14]
for k, e in top: print(f"{k:60s} {e['launches']:6d} launches  fetch {e['fetch_bytes']/1e9:8.2f} GB  write {e['write_bytes']/1e9:8.2f} GB  {e['GBps_fetch_plus_write']} GB/s")
