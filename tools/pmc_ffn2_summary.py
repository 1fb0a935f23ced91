#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_ffn2.sh (one row per kernel, counters per launch)."""
import csv, glob, subprocess, sys
from collections import defaultdict
d = sys.argv[1]
def short(n):
    if n.startswith('_Z'):
        try: n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
        except Exception: pass
    return n.replace('ira::', '').replace('void ', '').split('(')[0][:48]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(f'{d}/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    if 'ffn' not in k and 'pw_' not in k: continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0) or 1
    print(f"== {k}  ({len(next(iter(c.values())))} launches)")
    for n in sorted(m):
        print(f"   {n:28s} {m[n]:14.4g}   {100 * m[n] / wc:7.2f} % of SQ_WAVE_CYCLES")
