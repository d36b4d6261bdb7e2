"""The big fp16 update ALONE in its three probe modes (whole / K loop only / C stream only), tile forms from TILES, K = 1024,
m = n = 28672 on the fp32 copy -- for wave-level PMC passes (rocprofv3 --pmc ... --kernel-trace -- python3 tools/pmc_hgemm_modes.py).
Prints the launch order as PMCORDER json; tools/pmc_modes_table.py joins it with the counter CSV."""
import ctypes as C, importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
m = int(os.environ.get("M", "28672")); k = int(os.environ.get("K", "1024")); split = int(os.environ.get("SPLIT", "0"))
tiles = [int(t) for t in os.environ.get("TILES", "0,2").split(",")]
ctx.L.mpf_debug_hgemm_again.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32]
ctx.L.mpf_debug_hgemm_again.restype = C.c_int
Cm = torch.randn(m, m, dtype=torch.float32, device=ctx.device).t()
A = torch.randn(k, m, dtype=torch.float64, device=ctx.device).t()
B = torch.randn(m, k, dtype=torch.float64, device=ctx.device).t()
ctx.hgemm_minus_f32(Cm, A, B, split=bool(split)); ctx.synchronize()
order = []
for t in tiles:
    for d in (0, 1, 2):
        ctx.set_option("hgemm_big_tile", t); ctx.set_option("hgemm_dbg", d)
        for _ in range(2):
            assert ctx.L.mpf_debug_hgemm_again(ctx.h, m, m, k, Cm.data_ptr(), m, split) == 0
            order.append({"tile": t, "mode": ["whole", "kloop", "cstream"][d]})
        ctx.synchronize()
print("PMCORDER " + json.dumps({"m": m, "k": k, "split": split, "skip_first_big": 1, "launches": order}))
