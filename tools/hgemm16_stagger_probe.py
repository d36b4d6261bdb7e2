"""The staggered start of the big fp16 update (option hgemm_stagger_pct) against none, both forms (104 = one tile per workgroup,
105 = persistent), interleaved rounds, m = n = 28672 on the fp32 copy.  usage: hgemm16_stagger_probe.py [K ...]  env PCTS=0,50,100,150"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
ks = [int(a) for a in sys.argv[1:]] or [1024, 2048]
pcts = [int(t) for t in os.environ.get("PCTS", "0,50,100,150").split(",")]
forms = [int(t) for t in os.environ.get("FORMS", "4,5").split(",")]
m = int(os.environ.get("M", "28672"))
ctx.L.mpf_debug_hgemm_again.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32]
ctx.L.mpf_debug_hgemm_again.restype = C.c_int
Cm = torch.randn(m, m, dtype=torch.float32, device=ctx.device).t()
for k in ks:
    A = torch.randn(k, m, dtype=torch.float64, device=ctx.device).t(); B = torch.randn(m, k, dtype=torch.float64, device=ctx.device).t()
    ctx.set_option("hgemm_mfma16", 1); ctx.set_option("hgemm_dbg", 0)
    ctx.hgemm_minus_f32(Cm, A, B); ctx.synchronize()
    res = {}
    for rd in range(5):
        for f in forms:
            for p in pcts:
                ctx.set_option("hgemm_big_tile", f); ctx.set_option("hgemm_stagger_pct", p)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3): assert ctx.L.mpf_debug_hgemm_again(ctx.h, m, m, k, Cm.data_ptr(), m, 0) == 0
                e1.record(); torch.cuda.synchronize()
                if rd: res.setdefault((f, p), []).append(e0.elapsed_time(e1) / 3)
    for f in forms:
        for p in pcts:
            v = sorted(res[(f, p)]); med = v[len(v) // 2]
            print(f"m=n={m} K={k} form={f} stagger {p:3d} %: median {med:.3f} ms  min {v[0]:.3f} ms  {2.0*m*m*k/med/1e9:.0f} TFLOP/s", flush=True)
    del A, B
