#!/usr/bin/env python
"""Instruction mix of the loops of one kernel in hipcc -S output: tools/dev/loop_isa.py <file.s> <kernel-name-prefix>"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur, name, hdr = [], [], "entry", ""
for line in lines[start + 1:end]:
    l = line.strip()
    m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if m:
        blocks.append((name, hdr, cur)); name, hdr, cur = m.group(1), m.group(2), []
    elif l and not l.startswith(";") and not l.startswith("."):
        cur.append(l)
blocks.append((name, hdr, cur))
for b in blocks:
    nm = sum(1 for x in b[2] if x.startswith("v_mfma"))
    if len(b[2]) > 60 or nm:
        cnt = collections.Counter(x.split()[0] for x in b[2])
        valu = sum(c for k, c in cnt.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        print("%-12s %4d instr: VALU %3d (mov %d, trans %d) MFMA %d SALU %d ds %d vmem %d nop %d | %s" % (
            b[0], len(b[2]), valu, sum(c for k, c in cnt.items() if k.startswith("v_mov")),
            sum(c for k, c in cnt.items() if k.split("_")[1] in ("sqrt", "exp", "rcp", "log", "rsq")), nm,
            sum(c for k, c in cnt.items() if k.startswith("s_") and k != "s_nop"), sum(c for k, c in cnt.items() if k.startswith("ds_")),
            sum(c for k, c in cnt.items() if k.startswith("global_")), cnt["s_nop"], b[1].strip()[:40]))
