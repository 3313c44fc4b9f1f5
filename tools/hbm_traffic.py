#!/usr/bin/env python
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over the same command.
usage: python tools/hbm_traffic.py <fetch.db> <write.db> <steps_per_pass>
Units: the counters are KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts the
128-B requests of wide coalesced reads as 64 B -> doubled ("fetch_x2"); WRITE_SIZE is exact."""
import re
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    con = sqlite3.connect(path)
    rows = con.execute('select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name = ? '
                       'group by dispatch_id', (counter,)).fetchall()
    tot, cnt = defaultdict(float), defaultdict(int)
    for _, name, val in rows:
        short = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')[:28]
        tot[short] += val
        cnt[short] += 1
    return tot, cnt


def main(fetch_db, write_db, steps):
    f, fc = per_kernel(fetch_db, 'FETCH_SIZE')
    w, _ = per_kernel(write_db, 'WRITE_SIZE')
    names = sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0)))
    print(f'{"kernel":30s} {"calls":>6s} {"FETCH_SIZE":>12s} {"fetch_x2_MB":>12s} {"WRITE_MB":>10s}   (averages per dispatch)')
    tf = tw = 0.0
    for k in names:
        n = max(fc.get(k, 1), 1)
        tf += 2 * f.get(k, 0) * 1024
        tw += w.get(k, 0) * 1024
        print(f'{k:30s} {n:6d} {f.get(k, 0)/n:12.1f} {2*f.get(k, 0)*1024/n/1e6:12.1f} {w.get(k, 0)*1024/n/1e6:10.1f}')
    print(f'# per denoising step ({steps} steps, setup kernels included): fetch_x2 {tf/steps/1e9:.2f} GB + write '
          f'{tw/steps/1e9:.2f} GB = {(tf+tw)/steps/1e9:.2f} GB')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
