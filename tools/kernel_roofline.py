#!/usr/bin/env python
"""Per-kernel roofline table of the B=64 step from the tracked rocprofv3 summaries (serial single-stream schedule, so that a launch's
duration is the kernel's own): algorithmic FLOPs per launch (SURVEY.md section 8d terms) / PMC-pass duration vs the fp32 MFMA peak,
MfmaUtil and effective clock from the counters, HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE passes) vs the 8 TB/s peak.
usage: python tools/kernel_roofline.py [round tag, default r03]  > profiles/<tag>_kernel_roofline.txt"""
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
B, T, H, L, F, E, Nt, D = 64, 196, 12, 128, 512, 16, 77, 1536
N = 2 * B * T * H                    # tokens of the CFG-doubled batch
rows = 2 * B * T
# algorithmic GFLOP of the AVERAGE launch of the serial schedule (multiply-add = 2); None: not an MFMA kernel (HBM-bound).  Kernels whose
# launches differ in work but not in grid are averaged over one step (an average duration against the full-batch FLOPs
# overstated them): the persistent gemm_wp_k runs 7 FiLM GEMMs + the pose-encoder GEMM (K = 352) per step; base layer 0 launches gate-less
# half of the expert / proj work (CFG twin dedupe), so those kernels average 3.5 / 4 of a full launch.
TWIN = 3.5 / 4
FLOPS = {
    ('gemm_wp_k', 131072): (7 * 2.0 * rows * D * D + 2.0 * (B * T) * 352 * D) / 8,
    ('mlp2_k<128, 0>', None): TWIN * 2.0 * N * 2 * (L * 4 * L) * 2,            # top-2: two experts per token, FC1 + FC2
    ('mlp2_k<128, 1>', None): 2.0 * N * (L * F) * 2,
    ('mlp2d_k<128, 0>', None): TWIN * 2.0 * N * 2 * (L * 4 * L) * 2,           # round 4: the same MLPs with LDS-DMA staged weight chunks
    ('mlp2d_k<128, 1>', None): 2.0 * N * (L * F) * 2,
    ('projqkv_k<128>', None): 2.0 * N * (L * 4 * L + L * 3 * L),
    ('pqbody_k<128, 12>', None): TWIN * (2.0 * N * (L * 4 * L + L * 3 * L) + 2.0 * rows * (2 * H * H * L + 8 * 2 * H * (L // 8) ** 2 * 2)),      # + static and dynamic body topology
    ('gemm_small16_k<3, false>', None): 2.0 * (B * T) * 322 * D * 2,
    ('gemm_tail_k<3>', None): 2.0 * (B * T) * 322 * D * 2,                      # round 4: the folded decoder tail in one pass
    ('gemm_tail2_k<16>', None): 2.0 * (B * T) * 322 * D * 2,                    # round 5: the same as block ranges, A read once (both K groups in one launch)
    ('temporal_k<128, false>', None): 2.0 * B * H * ((Nt + T) * L * L + T * L * L) * 2,
    ('gate_k<128>', 602112): 2.0 * N * (L * 256 + 256 * E),
    ('gemm_small_k<false>', None): 2.0 * (B * T) * 322 * D * 2,
    ('gemm_k<4>', None): 2.0 * (B * T) * 324 * D,
}


def parse_pmc(path):
    out, cur = [], None
    for line in open(path):
        m = re.match(r'(\S.*?) grid=\((\d+),(\d+)\) x(\d+)\s+avg\s+([\d.]+) us', line)
        if m:
            cur = dict(name=m.group(1), grid=int(m.group(2)), gy=int(m.group(3)), n=int(m.group(4)), us=float(m.group(5)))
            out.append(cur)
            continue
        m = re.search(r'effective clock ([\d.]+) GHz', line)
        if m and cur:
            cur['clk'] = float(m.group(1))
        m = re.search(r'MfmaUtil ([\d.]+) %', line)
        if m and cur:
            cur['util'] = float(m.group(1))
    return out


def parse_hbm(path):
    out = {}
    for line in open(path):
        p = line.split()
        if len(p) >= 5 and re.match(r'^[\d.]+$', p[-1]) and re.match(r'^[\d.]+$', p[-2]) and not line.startswith('#'):
            try:
                out[' '.join(p[:-4])[:24]] = (float(p[-2]), float(p[-1]))
            except ValueError:
                pass
    return out


def parse_ledger(path):
    """profiles/<tag>_flop_ledger_serial.txt (tools/flop_ledger.py under the serial chain mask): (kernel, grid) -> (launches, GFLOP) per step"""
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        if line.startswith('#') or not line.strip():
            continue
        key, calls, gf = line.rstrip('\n').split('\t')
        name, grid = key.rsplit('@', 1)
        if name.startswith('setup:'):        # once-per-batch kernels (mc_ctx_set_condition): priced per launch, not part of the step's sums
            setup[(name[6:], int(grid))] = (int(calls), float(gf))
        else:
            out[(name, int(grid))] = (int(calls), float(gf))
    return out


setup = {}


def csrc_digest():
    """sha256 over the kernel sources the profiled library was built from (bench.py recomputes it: a table from other sources is `stale`)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'motioncraft_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


STEPS = 4                            # steps of the PMC pass (bench.py --steps 3 --warmup 1)
for line in open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_mfma_busy.txt')):
    if line.startswith('# commit'):
        print(line.rstrip())
        break
print(f'# csrc sha256 {csrc_digest()}')
pmc = parse_pmc(os.path.join(ROOT, 'profiles', f'{tag}_pmc_mfma_busy.txt'))
hbm = parse_hbm(os.path.join(ROOT, 'profiles', f'{tag}_pmc_hbm_traffic.txt'))
ledger = parse_ledger(os.path.join(ROOT, 'profiles', f'{tag}_flop_ledger_serial.txt'))
print(f'# tools/kernel_roofline.py {tag}: B=64 step, serial single-stream schedule (single-stream chain mask, PMC pass: profiles/{tag}_pmc_mfma_busy.txt), HBM bytes per launch of the')
print(f'# default two-stream schedule (half-batch launches: profiles/{tag}_pmc_hbm_traffic.txt).  GFLOP = useful multiply-add work (x 2) of the AVERAGE launch of a step as the')
if ledger:
    print(f'# launchers booked it (profiles/{tag}_flop_ledger_serial.txt: mc_debug_flop_ledger over one step of the same schedule, keyed kernel@grid; expert MLPs at the slot count')
    print('# of their routing, i.e. before capacity drops); peak 157.3 TFLOP/s fp32 MFMA at 2.4 GHz, HBM 8 TB/s.  `-` = not an MFMA kernel (row / routing / sampler passes).')
else:
    print('# reference performs it (layer-0 launches of the expert / proj kernels do half: twin dedupe; gemm_wp_k: 7 FiLM GEMMs + the encoder); peak 157.3 TFLOP/s fp32 MFMA at 2.4 GHz, HBM 8 TB/s.')
print(f'{"kernel":28s} {"grid":>9s} {"calls":>5s} {"us":>8s} {"GFLOP":>8s} {"TFLOP/s":>8s} {"%peak":>6s} {"MfmaUtil":>8s} {"GHz":>6s} {"MB/launch (2-stream)":>22s}')
sum_us = sum_gf = 0.0
used = set()
for r in pmc:
    if r['us'] < 10 or 'rocclr' in r['name'] or r['name'].startswith('at::'):
        continue
    fl = None
    per_step = r['n'] / STEPS
    if ledger:
        hit = [(k, v) for k, v in ledger.items() if r['name'].startswith(k[0]) and k[1] == r['grid'] * max(r['gy'], 1)]
        if not hit:
            hit = [(k, v) for k, v in ledger.items() if r['name'].startswith(k[0]) and k[1] == r['grid']]
        is_setup = False
        if hit:
            used.update(k for k, _ in hit)
            calls = sum(v[0] for _, v in hit)
            fl = sum(v[1] for _, v in hit) * 1e9 / calls
            if per_step >= 1:
                sum_gf += fl / 1e9 * per_step
        else:
            hs = [(k, v) for k, v in setup.items() if r['name'].startswith(k[0]) and k[1] in (r['grid'] * max(r['gy'], 1), r['grid'])]
            if hs:
                is_setup = True
                fl = sum(v[1] for _, v in hs) * 1e9 / sum(v[0] for _, v in hs)
                r['name'] = '(setup) ' + r['name']
    else:
        is_setup = False
        fl = FLOPS.get((r['name'], r['grid']), FLOPS.get((r['name'], None)))
    if per_step >= 1 and not is_setup and 'spin_k' not in r['name']:
        sum_us += r['us'] * per_step
    tf = fl / r['us'] / 1e6 if fl else None
    hb = next((v for k, v in hbm.items() if r['name'].startswith(k[:20]) or k.startswith(r['name'][:20])), None)
    print(f'{r["name"][:28]:28s} {r["grid"]:9d} {r["n"]:5d} {r["us"]:8.1f} {(f"{fl / 1e9:8.1f}" if fl else "       -")} '
          f'{(f"{tf:8.1f}" if tf else "       -")} {(f"{tf / 1.573:6.1f}" if tf else "     -")} {r.get("util", 0):8.1f} {r.get("clk", 0):6.2f} '
          f'{(f"{hb[0] + hb[1]:10.0f} ({hb[0]:.0f} rd + {hb[1]:.0f} wr)" if hb else ""):>22s}')
if ledger:
    missed = {k: v for k, v in ledger.items() if k not in used}
    total_gf = sum(v[1] for v in ledger.values())
    print(f'# closing: sum(calls x us) of the listed kernels = {sum_us / 1e3:.3f} ms per step of the serial schedule (kernels under 10 us not listed); sum of their GFLOP = '
          f'{sum_gf:.1f} of the {total_gf:.1f} GFLOP the ledger books per step = {total_gf / B:.3f} GFLOP per sample and step')
    print('#   (bench.py `executed_gflop_per_sample_step` counts the expert MLPs at tokens x top-2 as well; the reference-counted figure is `algorithmic_gflop_per_sample_step`)')
    if missed:
        print('# ledger rows without a listed kernel (launches under 10 us, or setup-only): ' + ', '.join(f'{k[0]}@{k[1]} {v[1]:.2f} GFLOP' for k, v in sorted(missed.items())))
