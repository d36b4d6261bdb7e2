"""Compact event trace of one kernel's gfx950 ISA: M = MFMA, D = LDS-DMA piece, L / S = buffer load / store, r / w = ds_read /
ds_write, | = s_barrier, [vN] / [lN] = s_waitcnt vmcnt / lgkmcnt, labels and branches on their own lines.
usage: isa_trace.py <file.s> <substring of the mangled kernel name>   (make asm writes lib/asm/*.s)"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
lines = s.split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and pat in l.split(":")[0]]
assert start, "kernel not found"
i0 = start[0]
ev = []
for l in lines[i0 + 1:]:
    l = l.strip()
    if l.startswith(".Lfunc_end") or l.startswith("s_endpgm"):
        ev.append("END\n")
        if l.startswith(".Lfunc_end"): break
        continue
    t = None
    if l.startswith("v_mfma"): t = "M"
    elif l.startswith("buffer_load") and "lds" not in l: t = "L"
    elif l.startswith("global_load_lds") or (l.startswith("buffer_load") and "lds" in l): t = "D"
    elif l.startswith("global_load"): t = "G"
    elif l.startswith("buffer_store") or l.startswith("global_store"): t = "S"
    elif l.startswith("ds_read"): t = "r"
    elif l.startswith("ds_write"): t = "w"
    elif l.startswith("s_barrier"): t = "|"
    elif l.startswith("s_waitcnt"):
        t = ""
        m = re.search(r"vmcnt\((\d+)\)", l)
        if m: t += "[v%s]" % m.group(1)
        m = re.search(r"lgkmcnt\((\d+)\)", l)
        if m: t += "[l%s]" % m.group(1)
    elif l.startswith("s_cbranch") or l.startswith("s_branch"): t = "<%s>\n" % l.split()[-1]
    elif re.match(r"\.LBB\d+_\d+:", l): t = "\n" + l + " "
    elif l.startswith("scratch_"): t = "!SCRATCH!"
    if t: ev.append(t)
print("".join(ev))
