#!/usr/bin/env python3
"""Per-kernel SQ counter table from tools/pmc_sq.sh output.  usage: sq_summary.py gpurun_out <prefix> [name filter]"""
import csv, sys, subprocess
from collections import defaultdict
d, pfx = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ''
def short(n):
    if n.startswith('_Z'):
        try: n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
        except Exception: pass
    return n.replace('ira::', '').replace('void ', '').split('(')[0]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in (f'{d}/{pfx}_sq/sq_counter_collection.csv', f'{d}/{pfx}_sq2/sq2_counter_collection.csv'):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = (short(r['Kernel_Name']), r['Grid_Size'])
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (k, r['Dispatch_Id']) not in seen and f.endswith('sq_counter_collection.csv'):
            seen.add((k, r['Dispatch_Id'])); cnt[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CYCLES', 0))
print('kernel | grid | n | GUI_ACTIVE/launch | wave_cyc(quad) | WAIT_ANY% | WAIT_INST% | ACTIVE% | MFMA_BUSY/(4*SIMD*GUI)% | LDS insts | VMEM insts')
for k, c in rows[:25]:
    if flt and flt not in k[0]: continue
    n = max(cnt[k], 1)
    wc = c.get('SQ_WAVE_CYCLES', 1)
    gui = c.get('GRBM_GUI_ACTIVE', 0) / n
    mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / n
    print(f"{k[0][:60]} | {k[1]} | {n} | {gui:.0f} | {wc/n:.3g} | {100*c.get('SQ_WAIT_ANY',0)/wc:.0f} | {100*c.get('SQ_WAIT_INST_ANY',0)/wc:.0f} | {100*c.get('SQ_ACTIVE_INST_ANY',0)/wc:.0f} | "
          f"{100*mf/max(gui*1024,1):.1f} | {c.get('SQ_INSTS_LDS',0)/n:.3g} | {c.get('SQ_INSTS_VMEM',0)/n:.3g}")
