#!/usr/bin/env python3
"""Idle time between consecutive kernels of a bench run, from a rocprofv3 --kernel-trace CSV (tools/prof_gpu.sh): over the last 1 / parts of
the dispatches (the timed step and the PSNR check that follows it; the head of the trace is set-up and warm-up).
usage: trace_gaps.py <dir with *_kernel_trace.csv> [parts = 2]"""
import csv, glob, os, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = [p for p in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)][0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the timed step = the last 1/steps of the dispatches that belong to the network (drop the leading setup kernels by taking the tail)
n = len(rows) // steps
last = rows[-n:]
busy = sum(e - s for s, e, _ in last); span = last[-1][1] - last[0][0]
gaps = [(last[i + 1][0] - last[i][1], last[i][2][:60], last[i + 1][2][:60]) for i in range(len(last) - 1)]
pos = [g for g in gaps if g[0] > 0]
print(f'{len(last)} dispatches, span {span / 1e6:.3f} ms, kernels busy {busy / 1e6:.3f} ms, idle between kernels {sum(g[0] for g in pos) / 1e6:.3f} ms '
      f'(median gap {sorted(g[0] for g in gaps)[len(gaps) // 2] / 1e3:.2f} us, {sum(1 for g in gaps if g[0] > 5000)} gaps > 5 us)')
for g in sorted(pos, key=lambda g: -g[0])[:8]: print(f'   {g[0] / 1e3:8.1f} us  after {g[1]}  before {g[2]}')
