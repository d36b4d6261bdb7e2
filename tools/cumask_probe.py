"""Does hipExtStreamCreateWithCUMask partition the chip here?  Times dgemm / hgemm on masked streams."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(8), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
dev = torch.device("cuda", 0)
m = 16384
Ag = torch.randn(256, m, dtype=torch.float64, device=dev).t()
Bg = torch.randn(m, 256, dtype=torch.float64, device=dev).t()
Cg = torch.randn(m, m, dtype=torch.float64, device=dev).t()
full = (1 << 256) - 1
cases = {"all 256": full, "low 128 bits": (1 << 128) - 1, "low 192 bits": (1 << 192) - 1, "even bits": int("55" * 32, 16),
         "bits 0-223": (1 << 224) - 1, "every 8th bit off": full & ~int("01" * 32, 16)}
for name, bits in cases.items():
    st = masked_stream(bits)
    ctx = mpf.MPFContext(0, probe=True, stream=st)
    with torch.cuda.stream(st):
        for kind, fn in (("dgemm", lambda: ctx.dgemm_minus(Cg, Ag, Bg)), ("hgemm", lambda: ctx.hgemm_minus(Cg, Ag, Bg))):
            fn(); st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(3): fn()
            e1.record(st); e1.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print(f"{name:20s} {kind}: {ms:.3f} ms ({2*m*m*256/ms/1e9:.1f} TF)", flush=True)
    ctx.close()
