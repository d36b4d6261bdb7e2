"""fp64 panel alone: new (10 launches) against the multi-launch form (MPF_DPANEL_MULTI=1 in the environment)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mpf = importlib.import_module("mixed-precision_lu_factorization_amd")
ctx = mpf.MPFContext(0)
dev = ctx.device
big = (torch.randint(0, 100, (512, 32768), device=dev, dtype=torch.int32).to(torch.float64) / 10.0).t()
for rows in (256, 2048, 8192, 32768):
    Wc = ctx.colmajor(rows, 256)
    src = big[:rows, 256:512].clone()
    src[:256, :256] += 50 * torch.eye(256, device=dev, dtype=torch.float64)
    times = []
    for rep in range(4):
        Wc.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.dgetf2_npv(Wc); e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    print(f"dgetf2_npv rows={rows} cols=256 ({'multi-launch' if os.environ.get('MPF_DPANEL_MULTI') == '1' else 'step+below'}): {min(times)*1e3:.0f} us")
