#!/usr/bin/env python
"""Per-kernel PMC summary from a rocprofv3 rocpd database: for each dispatch sum every counter over
its instances, then average over dispatches of the same (kernel, grid).
usage: python tools/rocpd_pmc.py <results.db> [name_filter]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, filt=''):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute('select dispatch_id, kernel_name, grid_size_x, grid_size_y, counter_name, sum(value), '
                       'max(end-start) from counters_collection group by dispatch_id, counter_name').fetchall()
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for did, name, gx, gy, cname, val, d in rows:
        short = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')
        if filt and filt not in short:
            continue
        key = (short[:40], gx, gy)
        per[key][cname].append(val)
        dur[key].append(d)
    for key in sorted(per, key=lambda k: -sum(dur[k])):
        c = {k: sum(v) / len(v) for k, v in per[key].items()}
        n = len(next(iter(per[key].values())))
        d = sum(dur[key]) / len(dur[key]) / 1e3
        print(f'{key[0]} grid=({key[1]},{key[2]}) x{n}  avg {d:9.1f} us')
        for k in sorted(c):
            print(f'     {k:34s} {c[k]:18.1f}')
        # GRBM_GUI_ACTIVE is reported once per XCD (8 on MI355X) and summed over them by the query above
        if 'GRBM_GUI_ACTIVE' in c and d > 0:
            print(f'     -> effective clock {c["GRBM_GUI_ACTIVE"]/8/d/1e3:.3f} GHz  (gui_active / 8 XCDs / duration)')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
            print(f'     -> MfmaUtil {c["SQ_VALU_MFMA_BUSY_CYCLES"]/(c["GRBM_GUI_ACTIVE"]/8*1024)*100:.1f} %  (busy cycles summed over SIMDs / (cycles * 1024 SIMDs))')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
