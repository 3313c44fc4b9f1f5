#!/usr/bin/env python
"""How much of one steady-state step the chip spends idle, on ONE kernel, and on two or more (two-stream schedule), from a launch-by-launch
timeline written by tools/rocpd_timeline.py (profiles/rNN_b64_timeline.txt): the step between two consecutive sampler_update_k launches.

    python tools/timeline_overlap.py profiles/r06_b64_timeline.txt > profiles/r06_b64_overlap.txt
"""
import sys
from collections import defaultdict

path = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r06_b64_timeline.txt'
rows = []
for line in open(path):
    p = line.split()
    if len(p) >= 5 and p[-4].isdigit():
        try:
            rows.append((p[0], int(p[-4]), float(p[-3]), float(p[-2])))
        except ValueError:
            pass
idx = [i for i, r in enumerate(rows) if r[0].startswith('sampler_update')]
if len(idx) < 2:
    raise SystemExit('need two sampler_update_k launches (one whole step) in the timeline')
seg = rows[idx[0] + 1:idx[1] + 1]
t0, t1 = min(r[2] for r in seg), max(r[2] + r[3] for r in seg)
pts = sorted({x for r in seg for x in (r[2], r[2] + r[3])})
idle = single = multi = 0.0
alone = defaultdict(float)
gaps = 0
for a, b in zip(pts[:-1], pts[1:]):
    m = 0.5 * (a + b)
    act = [r for r in seg if r[2] <= m < r[2] + r[3]]
    if not act:
        idle += b - a
        gaps += 1
    elif len(act) == 1:
        single += b - a
        alone[act[0][0][:28]] += b - a
    else:
        multi += b - a
print(f'# tools/timeline_overlap.py {path}: one steady-state step (between two sampler_update_k launches), {len(seg)} launches on '
      f'{len({r[1] for r in seg})} queues')
print(f'step wall {(t1 - t0) / 1e3:.3f} ms: idle {idle / 1e3:.3f} ms ({gaps} gaps), ONE kernel on the chip {single / 1e3:.3f} ms, two or more {multi / 1e3:.3f} ms')
print('kernels running alone (us):')
for k, v in sorted(alone.items(), key=lambda kv: -kv[1]):
    print(f'  {k:30s} {v:8.1f}')
