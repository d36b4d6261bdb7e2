"""A/B of the big fp16 update's tile forms in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): the kernel
ALONE (images converted once, probe library's mpf_debug_hgemm_again), whole / K loop only / C stream only (option
hgemm_dbg), m = n = 28672 on the fp32 copy.  usage: hgemm_tile_probe.py [K ...]   env TILES=0,3,4,100 (0 = hgemm_pp_kernel here: a stand-alone call; 100 + t = the 16x16x32 family of round 5,
hgemm16.hip: 100 = hgemm16_big_kernel; below 100: round 4's 32x32x16 kernels) SPLIT=0 M=28672 ROUNDS=5"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0, probe=True)
dev = ctx.device
ks = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]
tiles = [int(t) for t in os.environ.get("TILES", "0,1,2").split(",")]
split = int(os.environ.get("SPLIT", "0"))
m = int(os.environ.get("M", "28672"))
rounds = int(os.environ.get("ROUNDS", "5"))
dbgs = [int(t) for t in os.environ.get("DBG", "0,1,2").split(",")]
ctx.L.mpf_debug_hgemm_again.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32]
ctx.L.mpf_debug_hgemm_again.restype = C.c_int
Cm = torch.randn(m, m, dtype=torch.float32, device=dev).t()   # column-major, leading dimension m (16-byte aligned blocks)
for k in ks:
    A = torch.randn(k, m, dtype=torch.float64, device=dev).t()      # m x k column-major
    B = torch.randn(m, k, dtype=torch.float64, device=dev).t()      # k x m
    ctx.set_option("hgemm_mfma16", 0); ctx.set_option("hgemm_big_tile", 0); ctx.set_option("hgemm_dbg", 0)
    ctx.hgemm_minus_f32(Cm, A, B, split=bool(split)); ctx.synchronize()      # leaves the images in the context
    res = {}
    for rd in range(rounds + 1):
        for t in tiles:
            for d in dbgs:
                ctx.set_option("hgemm_mfma16", int(t >= 100)); ctx.set_option("hgemm_big_tile", 0 if t >= 100 else t); ctx.set_option("hgemm_dbg", d)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    rc = ctx.L.mpf_debug_hgemm_again(ctx.h, m, m, k, Cm.data_ptr(), m, split)
                    assert rc == 0, ctx.last_error() if hasattr(ctx, "last_error") else rc
                e1.record(); torch.cuda.synchronize()
                if rd: res.setdefault((t, d), []).append(e0.elapsed_time(e1) / 3)
    stamps = {}
    if not split:
        for t in tiles:
            for d in dbgs:
                if d == 0 or d == 2 or (t >= 100 and d in (3, 5)): continue
                if d == 6 and t != 0: continue
                ctx.set_option("hgemm_mfma16", int(t >= 100)); ctx.set_option("hgemm_big_tile", 0 if t >= 100 else t); ctx.set_option("hgemm_dbg", d)
                ctx.microbench(78)
                for _ in range(3):
                    ctx.L.mpf_debug_hgemm_again(ctx.h, m, m, k, Cm.data_ptr(), m, split)
                ctx.synchronize()
                cyc, ticks, cnt, epi = ctx.microbench(70), ctx.microbench(71), ctx.microbench(72), ctx.microbench(73)
                if cnt: stamps[(t, d)] = (cyc / cnt, ticks / cnt, epi / cnt)
    for t in tiles:
        for d in dbgs:
            v = sorted(res[(t, d)]); med, mn = v[len(v) // 2], v[0]
            what = {0: "whole", 1: "K loop only", 2: "C stream only", 3: "K loop, no DMA", 4: "K loop, no frag reads", 5: "K loop, MFMA only", 6: "K loop, blocked images"}.get(d, str(d))
            if t >= 100 and d == 3: what = "all but the C stores"
            if t >= 100 and d == 5: what = "all but the C loads"
            if t >= 100 and (d & 7) == 4: what = "whole, stamped" + (", C stores default policy" if d & 1024 else "") + (", C loads default policy" if d & 2048 else "")
            fl = 2.0 * m * m * k * (3 if split else 1)
            print(f"m=n={m} K={k} split={split} tile={t} {what:21s}: median {med:.3f} ms  min {mn:.3f} ms"
                  + (f"  {2.0*m*m*k/med/1e9:.0f} TFLOP/s (MFMA work {fl/med/1e9:.0f})" if d != 2 else f"  C traffic {8.0*m*m/med/1e6:.0f} GB/s")
                  + (f"  K loop per workgroup: {stamps[(t, d)][0]:.0f} cycles = {stamps[(t, d)][0] / (k / 32):.0f} per 32-k stage, "
                     f"{stamps[(t, d)][1] / 100:.2f} us, clock {stamps[(t, d)][0] / stamps[(t, d)][1] / 10:.3f} GHz"
                     + (f"; tile switch + C stream {stamps[(t, d)][2] / 100:.2f} us per tile" if stamps[(t, d)][2] else "") if (t, d) in stamps else ""), flush=True)
    del A, B
ctx.set_option("hgemm_dbg", 0)
