#!/usr/bin/env python
"""Per-kernel PMC summary from rocprofv3 rocpd databases (one --pmc pass per db).

usage: python tools/pmc_summary.py db1 [db2 ...]   -> markdown table: per kernel name, per counter: sum over dispatches / dispatches
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    per = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        q = ("select name, counter_name, sum(counter_value), count(distinct dispatch_id), sum(duration) from pmc_events "
             "group by name, counter_name")
        try:
            rows = db.execute(q).fetchall()
        except sqlite3.OperationalError:
            cols = [c[1] for c in db.execute("pragma table_info('pmc_events')")]
            raise SystemExit(f"unexpected pmc_events schema: {cols}")
        for name, ctr, val, n, _dur in rows:
            per[name][ctr] += val
            calls[name] = max(calls[name], n)
    ctrs = sorted({c for v in per.values() for c in v})
    print("| kernel | dispatches | " + " | ".join(ctrs) + " |")
    print("|---|---|" + "---|" * len(ctrs))
    order = sorted(per, key=lambda k: -per[k].get("SQ_WAVE_CYCLES", per[k].get("FETCH_SIZE", 0)))
    for name in order:
        short = name.replace("pf::", "").replace("(pf::ConvP)", "").replace("void ", "")[:70]
        print(f"| `{short}` | {calls[name]} | " + " | ".join(f"{per[name].get(c, 0) / max(calls[name], 1):.4g}" for c in ctrs) + " |")


if __name__ == "__main__":
    main()
