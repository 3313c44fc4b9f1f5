#!/usr/bin/env python
"""Launch-by-launch timeline of the last `n` kernel dispatches in a rocprofv3 rocpd database: start offset, duration, gap to
the previous kernel's end, queue.  Shows what sits on the critical path of a latency-bound step.
usage: python tools/rocpd_timeline.py <results.db> [n=80]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    name = re.sub(r'\(.*', '', name)
    return name[:48]


def main(path, n=80):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    sel = f'name, start, end, {qcol}' if qcol else 'name, start, end, 0'
    rows = cur.execute(f'select {sel} from kernels order by start desc limit {n}').fetchall()[::-1]
    t0 = rows[0][1]
    busy_end = rows[0][1]
    print(f'{"kernel":48s} {"queue":>6s} {"start_us":>10s} {"dur_us":>8s} {"gap_us":>8s}')
    tot_gap = 0.0
    for name, st, en, q in rows:
        gap = (st - busy_end) / 1e3
        if gap > 0: tot_gap += gap
        print(f'{short(name):48s} {str(q):>6s} {(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.1f} {gap:8.1f}')
        busy_end = max(busy_end, en)
    print(f'# span {(busy_end - t0) / 1e3:.1f} us, idle (no kernel running) {tot_gap:.1f} us')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80)
