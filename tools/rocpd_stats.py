#!/usr/bin/env python
"""Kernel-time summary (like `rocprofv3 --stats`) from a rocprofv3 rocpd sqlite database.
usage: python tools/rocpd_stats.py <results.db> [top_n]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    return name if len(name) <= 90 else name[:87] + '...'


def main(path, top=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    rows = cur.execute('select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name '
                       'order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    print(f'# {path}: {sum(r[1] for r in rows)} kernel dispatches, total kernel time {total/1e6:.3f} ms')
    print(f'{"kernel":92s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>10s} {"min_us":>9s} {"max_us":>9s} {"pct":>6s}')
    for name, n, tot, mn, mx in rows[:top]:
        print(f'{short(name):92s} {n:7d} {tot/1e6:10.3f} {tot/n/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*tot/total:6.2f}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
