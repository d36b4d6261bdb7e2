"""Digest of a bench.py line: usage  bench_digest.py [file]  (default gpurun_out/r05_bench_final.json; '-' = stdin)"""
import json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05_bench_final.json"
txt = sys.stdin.read() if src == "-" else open(src).read()
d = json.loads([l for l in txt.strip().splitlines() if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "check", (d.get("check") or {}).get("passed"))
print("roofline", {k: (d.get("roofline") or {}).get(k) for k in ("achieved", "frac", "rocprof", "traffic", "traffic_over_algorithmic")})
print("ir", d.get("ir"))
for k in ("mxp", "mxp_x3"):
    m = d.get(k) or {}
    print(k, {kk: m.get(kk) for kk in ("factor_ms", "ir_ms", "ir_iterations", "superpanel")},
          {kk: (m.get("roofline") or {}).get(kk) for kk in ("achieved", "frac_of_fp16_mfma_peak_spec", "traffic", "traffic_over_algorithmic")})
rs = d.get("reference_style") or {}
print("ref_style", {k: rs.get(k) for k in ("ms", "h2d_ms", "d2h_ms", "factor_ms", "rows_streamed", "late_segments", "first_call_ms")})
