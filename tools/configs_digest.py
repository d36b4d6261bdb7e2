"""One line per bench.py output file: the mxp leg and the `configs` object's times.  usage: configs_digest.py file [...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    c = d.get("configs") or {}
    print(f, "mxp", (d.get("mxp") or {}).get("factor_ms"), {k: ({kk: v.get(kk) for kk in ("ms", "factor_ms", "total_ms", "tflops") if kk in v} if isinstance(v, dict) else v) for k, v in c.items()})
