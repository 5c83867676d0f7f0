#!/usr/bin/env python
"""Instruction mix, wait counts and scratch traffic of the MFMA region of one kernel in a hipcc -save-temps .s file:
   python tools/asm_loop_report.py file.s kernel_substring [context]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r"^(\S*%s\S*):" % re.escape(name), s, re.M)
k = m.start()
body = s[k:s.index(".Lfunc_end", k)].split("\n")
mf = [i for i, l in enumerate(body) if "v_mfma" in l and not l.rstrip().endswith(", 0")]   # (not the accumulator-clearing ones)
print(m.group(1), "lines", len(body), "mfma", len(mf), "first", mf[0], "last", mf[-1])
reg = body[mf[0] - 60:mf[-1] + 5]
c = collections.Counter()
for l in reg:
    t = l.strip().split()
    if t and not t[0].startswith((";", ".")):
        c[t[0]] += 1
print("  ".join(f"{v} {k_}" for k_, v in c.most_common(40)))
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for i, l in enumerate(reg):
    if "s_waitcnt" in l or "scratch_" in l or "s_barrier" in l:
        if ctx:
            print("\n".join("      " + x.strip() for x in reg[max(0, i - ctx):i]))
        print(i, l.strip())
