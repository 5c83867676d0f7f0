#!/usr/bin/env python
"""Derived per-kernel metrics from ONE rocprofv3 --pmc pass that holds
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name").fetchall()
per = defaultdict(dict)
n = {}
for name, c, v, k in rows:
    per[name][c] = v
    n[name] = k
print("| kernel | disp | kcyc/disp | MFMA busy % | waves/SIMD | wait_any % | wait_inst % | active % | valu % | VALU inst/wave |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name in sorted(per, key=lambda k: -per[k].get("GRBM_GUI_ACTIVE", 0)):
    p = per[name]
    if "GRBM_GUI_ACTIVE" not in p or p["GRBM_GUI_ACTIVE"] == 0:
        continue
    dur = p["GRBM_GUI_ACTIVE"] / 8.0           # cycles (summed over 8 XCDs)
    wc = 4.0 * p.get("SQ_WAVE_CYCLES", 0)      # quad-cycles -> cycles
    f = lambda c: 100.0 * 4.0 * p.get(c, 0) / wc if wc else 0
    waves = p.get("SQ_WAVES", 0)
    short = name.replace("pf::", "").replace("(pf::ConvP)", "").replace("void ", "")[:60]
    print(f"| `{short}` | {n[name]} | {dur / n[name] / 1e3:.1f} | {100 * p.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * dur):.1f} | "
          f"{wc / (1024 * dur):.2f} | {f('SQ_WAIT_ANY'):.0f} | {f('SQ_WAIT_INST_ANY'):.0f} | {f('SQ_ACTIVE_INST_ANY'):.0f} | "
          f"{f('SQ_ACTIVE_INST_VALU'):.0f} | {p.get('SQ_INSTS_VALU', 0) / waves if waves else 0:.0f} |")
