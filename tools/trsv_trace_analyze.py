"""Per-step durations and gaps of the triangular solves from a rocprofv3 kernel-trace CSV of tools/ir_time_probe.py.
usage: trsv_trace_analyze.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "trsv_step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST complete lower solve: the last run of 129 consecutive <false> launches
low = [r for r in rows if "<false>" in r["Kernel_Name"]]
n = 129
seq = low[-n:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seq]
g = [(int(seq[i + 1]["Start_Timestamp"]) - int(seq[i]["End_Timestamp"])) / 1e3 for i in range(n - 1)]
for i in (0, 1, 2, 4, 8, 16, 32, 48, 64, 80, 96, 112, 120, 126, 127, 128):
    print(f"step {i:3d}: {d[i]:6.2f} us   grid {seq[i]['Grid_Size_X']}   gap to next {g[i] if i < n - 1 else 0:5.2f} us")
print(f"sum of durations {sum(d) / 1e3:.3f} ms, sum of gaps {sum(g) / 1e3:.3f} ms, span {(int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e6:.3f} ms")
