import json
d=json.loads(open("gpurun_out/r05_bench_final.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "check", d["check"]["passed"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved","frac","rocprof","traffic","traffic_over_algorithmic")})
print("ir", d["ir"])
for k in ("mxp","mxp_x3"):
    m=d.get(k) or {}
    print(k, {kk: m.get(kk) for kk in ("factor_ms","ir_ms","ir_iterations","superpanel")}, {kk: (m.get("roofline") or {}).get(kk) for kk in ("achieved","frac_of_fp16_mfma_peak_spec","traffic","traffic_over_algorithmic")})
print("ref_style", {k: d["reference_style"].get(k) for k in ("ms","h2d_ms","d2h_ms","factor_ms")})
