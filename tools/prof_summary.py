#!/usr/bin/env python3
"""Join rocprofv3 --kernel-trace and --pmc FETCH_SIZE / WRITE_SIZE passes of the same command
(tools/prof_gpu.sh) into a per-kernel table: calls, average duration, HBM bytes per launch.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced read, so reads = 2 * FETCH_SIZE * 1024;
WRITE_SIZE * 1024 is exact for 16-byte-per-lane stores.  Dispatches are matched across passes by
(kernel name, grid size, ordinal), which is valid because the command is deterministic.

usage: prof_summary.py <dir with <prefix>_trace/ <prefix>_fetch/ <prefix>_write/> [out.md] [traffic.json] [--prefix prof]
(traffic.json is merged into, not overwritten)
"""
import csv
import json
import re
import subprocess
import sys
from collections import defaultdict


def short(name):
    n = name
    if n.startswith('_Z'):
        try:
            n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
        except Exception:
            pass
    n = n.replace('ira::', '').replace('void ', '')
    return n.split('(')[0]


def load_trace(path):
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
            rows[(short(r['Kernel_Name']), g)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return rows


def load_pmc(path, counter):
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] == counter:
                rows[(short(r['Kernel_Name']), int(r['Grid_Size']))].append(float(r['Counter_Value']))
    return rows


def main():
    d = sys.argv[1]
    pfx = 'prof'
    if '--prefix' in sys.argv:
        i = sys.argv.index('--prefix')
        pfx = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    tr = load_trace(f'{d}/{pfx}_trace/trace_kernel_trace.csv')
    fe = load_pmc(f'{d}/{pfx}_fetch/fetch_counter_collection.csv', 'FETCH_SIZE')
    wr = load_pmc(f'{d}/{pfx}_write/write_counter_collection.csv', 'WRITE_SIZE')
    total = sum(sum(v) for v in tr.values())
    out = []
    out.append('| kernel | grid (threads) | calls | avg us | % time | HBM read MB/launch (2x FETCH_SIZE) | HBM write MB/launch | HBM GB/s |')
    out.append('|---|---|---|---|---|---|---|---|')
    traffic = {}
    for key, durs in sorted(tr.items(), key=lambda kv: -sum(kv[1])):
        n = len(durs)
        avg = sum(durs) / n
        f = fe.get(key, [])
        w = wr.get(key, [])
        rd = 2.0 * 1024 * sum(f) / len(f) if f else float('nan')
        wb = 1024.0 * sum(w) / len(w) if w else float('nan')
        gbs = (rd + wb) / avg if f and w else float('nan')
        out.append(f'| {key[0]} | {key[1]} | {n} | {avg / 1e3:.1f} | {100 * sum(durs) / total:.1f} | {rd / 1e6:.1f} | {wb / 1e6:.1f} | {gbs:.0f} |')
        traffic[f'{key[0]}|grid{key[1]}'] = {'calls': n, 'avg_us': avg / 1e3, 'hbm_read_bytes': rd, 'hbm_write_bytes': wb}
    # traffic of the two fused kernels under the keys bench.py uses (profile run = bench defaults: T = 40, tile 512)
    import re as _re
    for key in list(tr.keys()):
        m_ = _re.match(r'k_(ffn|qkv)_fused<(\d+)', key[0]) or _re.match(r'_ZN3ira11k_(ffn|qkv)_fusedILi(\d+)', key[0])
        m2 = _re.match(r'k_ffn2<(\d+), \d+, \w+, \w+, (true|false)', key[0]) or _re.match(r'_ZN3ira6k_ffn2ILi(\d+)ELi\dELb\dELb\dELb(\d)', key[0])
        if m2 and key in fe and key in wr:
            patches = key[1] // 512
            fam = 'attn_ffn_fused' if m2.group(2) in ('true', '1') else 'ffn_fused'
            for T_, side in ((40, 512), (40, 256), (8, 512), (8, 256)):
                if patches == T_ * ((side + 7) // 8) * ((side + 31) // 32):
                    traffic[f'{fam}|C{m2.group(1)}|{side}x{side}|T{T_}'] = int(2.0 * 1024 * sum(fe[key]) / len(fe[key]) + 1024.0 * sum(wr[key]) / len(wr[key]))
                    break
        if m_ and key in fe and key in wr:
            patches = key[1] // 512
            for T_, side in ((40, 512), (40, 256), (8, 512), (8, 256)):
                if patches == T_ * ((side + 7) // 8) * ((side + 31) // 32):
                    traffic[f'{m_.group(1)}_fused|C{m_.group(2)}|{side}x{side}|T{T_}'] = int(2.0 * 1024 * sum(fe[key]) / len(fe[key]) + 1024.0 * sum(wr[key]) / len(wr[key]))
                    break
    # NAFNet bench (16 tiles of 512x512): the two level-1 kernels bench.py can name as dominant
    for key in list(tr.keys()):
        for pat, fam, grids in ((r'k_naf_dw_sg', 'naf_dw_gate', {4194304: ('C32', 512), 2097152: ('C64', 256)}),
                                (r'k_naf_tail<32>|k_naf_tailILi32', 'naf_tail_fused', {1048576: ('C32', 512)}),
                                (r'k_naf_tail<64>|k_naf_tailILi64', 'naf_tail_fused', {262144: ('C64', 256)})):
            if _re.search(pat, key[0]) and key[1] in grids and key in fe and key in wr:
                cc, side = grids[key[1]]
                traffic[f'{fam}|{cc}|{side}x{side}|T16'] = int(2.0 * 1024 * sum(fe[key]) / len(fe[key]) + 1024.0 * sum(wr[key]) / len(wr[key]))
    text = f'total kernel time {total / 1e6:.2f} ms over {sum(len(v) for v in tr.values())} dispatches\n\n' + '\n'.join(out) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(text)
    if len(sys.argv) > 3:
        import os
        import datetime, glob, hashlib
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
        hsh = hashlib.sha1()
        for f in sorted(glob.glob(os.path.join(root, 'image-restoration-agent_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(root, 'image-restoration-agent_amd', 'csrc', '*.h'))):
            hsh.update(os.path.basename(f).encode())
            hsh.update(open(f, 'rb').read())
        sha = hsh.hexdigest()[:16]                 # the same digest bench.py::csrc_sha() computes: traffic is quoted only for the sources it was measured on
        old = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
        if old.get('_stamp', {}).get('csrc_sha') != sha:
            old = {}                               # measurements of other sources are dropped, not mixed
        old.update(traffic)
        old['_stamp'] = {'csrc_sha': sha, 'when': datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%MZ'),
                         'how': 'tools/prof_gpu.sh: rocprofv3 --kernel-trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE as separate passes of bench.py --steps 1'}
        json.dump(old, open(sys.argv[3], 'w'), indent=1)
    print(text)


if __name__ == '__main__':
    main()
