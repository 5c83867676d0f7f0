#!/usr/bin/env python
"""Scan generated gfx950 assembly for VALU instructions that WRITE a register which one of the last few MFMAs READS as its A/B
operand (write-after-read against the matrix pipe: the VALU write has to wait until the MFMA has fetched its sources).
usage: python tools/war_scan.py build/asm/conv_bf16x3.s [kernel-name-substring] [window]"""
import re
import sys
from collections import Counter

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
window = int(sys.argv[3]) if len(sys.argv) > 3 else 1     # how many preceding MFMAs to check against
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


cur, stats = None, {}
recent = []      # (source regs, distance in instructions)
for raw in open(path):
    line = raw.strip()
    if line.startswith("_Z") and ":" in line.split(";")[0]:
        cur = line.split(":")[0]
        stats[cur] = dict(mfma=0, hits=Counter(), valu=0)
        recent = []
        continue
    if cur is None or not line or line.startswith(";") or line.startswith("."):
        continue
    if want and want not in cur:
        continue
    op = line.split()[0]
    if op.startswith("v_mfma"):
        ops = line.split(None, 1)[1].split(",")
        src = regs(ops[1]) | regs(ops[2])
        recent = [(src, 0)] + [(s, d) for s, d in recent][: window - 1]
        stats[cur]["mfma"] += 1
        continue
    recent = [(s, d + 1) for s, d in recent]
    if op.startswith("v_") and not op.startswith("v_accvgpr") and not op.startswith("v_cmp"):
        stats[cur]["valu"] += 1
        dst = regs(line.split(None, 1)[1].split(",")[0]) if len(line.split(None, 1)) > 1 else set()
        for s, d in recent:
            if dst & s and d <= 12:
                stats[cur]["hits"][op] += 1
                break
for k, v in stats.items():
    if v["mfma"] and (not want or want in k):
        print(f"{k[:110]}\n   mfma {v['mfma']}  valu {v['valu']}  VALU writes into a source of the preceding {window} MFMA(s): {sum(v['hits'].values())}  {dict(v['hits'].most_common(6))}")
