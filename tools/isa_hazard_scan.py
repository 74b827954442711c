#!/usr/bin/env python3
"""Static scan of gfx950 ISA listings for the instruction pattern behind the round-2 S0 loss.

The pattern (bnp_vae_amd/csrc/mstep_common.h, "S0 accumulation"): a packed fp32 VALU instruction with
a non-default op_sel / op_sel_hi that is the FIRST VALU instruction of a branch-target block and reads
a register pair some packed instruction wrote a few instructions in front of the branch.

    python tools/isa_hazard_scan.py [--build] [--window 8] listing.s ...

--build produces the listings itself (hipcc --cuda-device-only -S of every kernel source, the four
padded-D instantiations of the E-step and the unpinned M-step build the pattern was found in) into
/tmp/isa.  Three classes are reported per kernel:
  HIT   first VALU of a branch target, packed fp32 with op_sel, source written by a packed
        instruction within the window on some way into the block   (the round-2 pattern)
  NEAR  the same without the packed producer in the window
  (count) packed-with-op_sel instructions in all, for scale
No GPU is needed.  The summary of the shipped build is kept in profiles/r03_isa_hazard_scan.txt.
"""
import argparse
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bnp_vae_amd", "csrc")
PLAIN = ["api", "noalloc", "split_stats", "mstep", "mstep_dma", "mstep_wide", "estep", "tree_kernels"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-pass-failed",
         "-I" + os.path.join(CSRC, "..", "..", "include"), "--cuda-device-only", "-S"]

LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^(_Z\w+|\w+):\s*(;.*)?$")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
BRANCH = re.compile(r"^\s+s_c?branch\w*\s+(\.LBB\d+_\d+)")


def build(outdir):
    os.makedirs(outdir, exist_ok=True)
    jobs = [(n, [n + ".hip"], []) for n in PLAIN]
    jobs += [("estep_dp%d" % d, ["estep_inst.hip"], ["-DBNPVI_DP=%d" % d]) for d in (32, 64, 128, 256)]
    jobs += [("mstep_dma_nopin", ["mstep_dma.hip"], ["-DBNPVI_NO_S0_PIN"])]
    procs = []
    for name, src, defs in jobs:
        out = os.path.join(outdir, name + ".s")
        procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + defs + src + ["-o", out], cwd=CSRC)))
    outs = []
    for out, p in procs:
        if p.wait() != 0:
            sys.exit("hipcc failed for " + out)
        outs.append(out)
    return outs


def regs_of(text):
    """set of VGPR numbers named in an operand string"""
    got = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            got.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            got.add(int(m.group(3)))
    return got


def split_insn(line):
    body = line.split(";")[0].strip()
    if not body or body.startswith("."):
        return None
    parts = body.split(None, 1)
    mnem = parts[0]
    ops = parts[1] if len(parts) > 1 else ""
    first = ops.split(",")[0] if ops else ""
    rest = ops[len(first):]
    return mnem, first, rest


def is_valu(mnem):
    return mnem.startswith("v_") and not mnem.startswith("v_mfma") and not mnem.startswith("v_accvgpr") \
        and not mnem.startswith("v_smfmac")


def scan(path, window):
    lines = open(path).read().split("\n")
    # per function: list of (kind, payload): ("label", name) / ("insn", (mnem, dst, srcs, raw))
    funcs, cur, name = {}, None, None
    for ln in lines:
        m = LABEL.match(ln)
        if m and cur is not None:
            cur.append(("label", m.group(1)))
            continue
        if not ln.startswith(("\t", " ")):
            f = FUNC.match(ln)
            if f and not ln.startswith("."):
                name = f.group(1)
                cur = funcs.setdefault(name, [])
            continue
        if cur is None:
            continue
        if ln.strip().startswith(".end_amdhsa_kernel") or ln.strip().startswith(".section"):
            cur = None
            continue
        ins = split_insn(ln)
        if ins:
            cur.append(("insn", ins + (ln.strip(),)))
    report = []
    for fn, items in funcs.items():
        n_pk_sel = 0
        label_at = {p: i for i, (k, p) in enumerate(items) if k == "label"}
        # ways into a label: fall-through (index before it) and every branch naming it
        into = {}
        for i, (k, p) in enumerate(items):
            if k == "insn" and p[0].startswith(("s_cbranch", "s_branch")):
                m = re.search(r"(\.LBB\d+_\d+)", p[3])
                if m:
                    into.setdefault(m.group(1), []).append(i)
        hits, near = [], []
        for i, (k, p) in enumerate(items):
            if k == "insn" and p[0].startswith("v_pk_") and "op_sel" in p[3]:
                n_pk_sel += 1
        for lab, at in label_at.items():
            if lab not in into:
                continue  # not a branch target
            # first VALU instruction of the block
            j = at + 1
            first = None
            while j < len(items) and items[j][0] == "insn":
                mn = items[j][1][0]
                if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                    break
                if is_valu(mn):
                    first = items[j][1]
                    break
                j += 1
            if first is None or not first[0].startswith("v_pk_") or "f32" not in first[0] or "op_sel" not in first[3]:
                continue
            srcs = regs_of(first[2].split("op_sel")[0])
            producer = None
            ways = list(into[lab])
            if at > 0 and items[at - 1][0] == "insn" and not items[at - 1][1][0].startswith("s_branch"):
                ways.append(at)  # fall-through
            for w in ways:
                seen, b = 0, w - 1
                while b >= 0 and seen < window:
                    if items[b][0] == "insn":
                        pm, pd, _, praw = items[b][1]
                        seen += 1
                        if pm.startswith("v_pk_") and regs_of(pd) & srcs:
                            producer = praw
                            break
                    b -= 1
                if producer:
                    break
            (hits if producer else near).append((lab, first[3], producer))
        if n_pk_sel or hits or near:
            report.append((fn, n_pk_sel, hits, near))
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--outdir", default="/tmp/isa")
    ap.add_argument("--window", type=int, default=8)
    ap.add_argument("listings", nargs="*")
    a = ap.parse_args()
    files = list(a.listings)
    if a.build:
        files += build(a.outdir)
    if not files:
        ap.error("no listings")
    total_hits = 0
    for f in sorted(files):
        rep = scan(f, a.window)
        print("== %s" % os.path.basename(f))
        if not rep:
            print("   no packed fp32 instruction with op_sel")
        for fn, n, hits, near in rep:
            short = fn if len(fn) < 90 else fn[:87] + "..."
            print("   %-90s packed+op_sel %4d   HIT %d   NEAR %d" % (short, n, len(hits), len(near)))
            for lab, ins, prod in hits:
                print("      HIT  %s: %s\n           producer: %s" % (lab, ins, prod))
                total_hits += 1
            for lab, ins, _ in near:
                print("      NEAR %s: %s" % (lab, ins))
    print("hits in all: %d" % total_hits)


if __name__ == "__main__":
    main()
