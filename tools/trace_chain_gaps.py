"""Kernel trace of tools/trace_fp16_factor.py -> for every panel of the LAST factorization: what runs between the end of one pivot
kernel and the start of the next (name, start relative to the pivot kernel's end, duration), summed per kernel name."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows))
piv = [e for e in ev if e[2].startswith("hgetf2_lds_kernel") or e[2].startswith("hgetf2_win_kernel")]
npan = 127
piv = piv[-npan:]                      # the last factorization's pivot kernels
tot = collections.defaultdict(lambda: [0, 0.0]); gap_sum = 0.0
detail = []
for a, b in zip(piv[:-1], piv[1:]):
    lo, hi = a[1], b[0]
    gap_sum += (hi - lo) / 1e3
    ins = [e for e in ev if e[1] > lo and e[0] < hi and not e[2].startswith("hgetf2_")]
    for e in ins:
        d = (min(e[1], hi) - max(e[0], lo)) / 1e3
        tot[e[2]][0] += 1; tot[e[2]][1] += d
    detail.append((lo, hi, ins))
print(f"{len(piv)} pivot kernels, sum of gaps {gap_sum/1e3:.2f} ms, pivot kernels {sum(p[1]-p[0] for p in piv)/1e6:.2f} ms")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {k[:70]:70s} {v[0]:6d} x  {v[1]/1e3:8.2f} ms inside gaps")
for i in (2, 40, 100, 120):
    lo, hi, ins = detail[i]
    print(f"panel {i}: gap {(hi-lo)/1e3:.1f} us")
    for e in sorted(ins):
        print(f"    {e[2][:60]:60s} start {(e[0]-lo)/1e3:8.1f} us  dur {(e[1]-e[0])/1e3:7.1f} us")
