#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace) into a per-kernel table (markdown).

usage: python tools/prof_summary.py gpurun_out/prof_x/bench_results.db [steps] > profiles/xyz.md
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"| kernel | calls | total ms | avg us | min us | max us | % | vgpr | lds |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx, vg, lds in rows:
        short = name.replace("pf::", "").replace("(pf::ConvP)", "").replace("void ", "")
        if len(short) > 90:
            short = short[:87] + "..."
        print(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.1f} | {vg} | {lds} |")
    print(f"\nall kernels: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches" + (f"; {total / 1e6 / steps:.3f} ms per step ({steps} steps incl. warm-up)" if steps else ""))


if __name__ == "__main__":
    main()
