#!/usr/bin/env python
"""Device-timeline gaps between consecutive kernel dispatches of a rocprofv3 --kernel-trace rocpd database:
how much of the step is kernels, how much is launch-to-launch dependency bubbles.
usage: python tools/prof_gaps.py <db> [first_dispatch_to_skip]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = db.execute("select name, start, end from kernels order by start").fetchall()[skip:]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 50_000]
print(f"dispatches {len(rows)}  span {span / 1e6:.3f} ms  kernels {busy / 1e6:.3f} ms ({100 * busy / span:.1f}%)")
print(f"gaps < 50 us: n={len(small)} total {sum(small) / 1e6:.3f} ms  mean {sum(small) / max(1, len(small)) / 1e3:.2f} us  "
      f"median {sorted(small)[len(small) // 2] / 1e3:.2f} us;  gaps >= 50 us: n={len(gaps) - len(small)} total {sum(g for g in gaps if g >= 50_000) / 1e6:.3f} ms")
