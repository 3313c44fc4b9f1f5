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


for line in open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_mfma_busy.txt')):
    if line.startswith('# commit'):
        print(line.rstrip())
        break
pmc = parse_pmc(os.path.join(ROOT, 'profiles', f'{tag}_pmc_mfma_busy.txt'))
hbm = parse_hbm(os.path.join(ROOT, 'profiles', f'{tag}_pmc_hbm_traffic.txt'))
print(f'# tools/kernel_roofline.py {tag}: B=64 step, serial single-stream schedule (single-stream chain mask, PMC pass: profiles/{tag}_pmc_mfma_busy.txt), HBM bytes per launch of the')
print(f'# default two-stream schedule (half-batch launches: profiles/{tag}_pmc_hbm_traffic.txt).  GFLOP = algorithmic work of the AVERAGE launch of a step as the reference')
print('# performs it (layer-0 launches of the expert / proj kernels do half: twin dedupe; gemm_wp_k: 7 FiLM GEMMs + the encoder); peak 157.3 TFLOP/s fp32 MFMA at 2.4 GHz, HBM 8 TB/s.')
print(f'{"kernel":28s} {"grid":>9s} {"calls":>5s} {"us":>8s} {"GFLOP":>8s} {"TFLOP/s":>8s} {"%peak":>6s} {"MfmaUtil":>8s} {"GHz":>6s} {"MB/launch (2-stream)":>22s}')
for r in pmc:
    if r['us'] < 10 or 'rocclr' in r['name'] or r['name'].startswith('at::'):
        continue
    fl = FLOPS.get((r['name'], r['grid']), FLOPS.get((r['name'], None)))
    tf = fl / r['us'] / 1e6 if fl else None
    hb = next((v for k, v in hbm.items() if r['name'].startswith(k[:20]) or k.startswith(r['name'][:20])), None)
    print(f'{r["name"][:28]:28s} {r["grid"]:9d} {r["n"]:5d} {r["us"]:8.1f} {(fl / 1e9 if fl else 0):8.1f} '
          f'{(f"{tf:8.1f}" if tf else "       -")} {(f"{tf / 1.573:6.1f}" if tf else "     -")} {r.get("util", 0):8.1f} {r.get("clk", 0):6.2f} '
          f'{(f"{hb[0] + hb[1]:10.0f} ({hb[0]:.0f} rd + {hb[1]:.0f} wr)" if hb else ""):>22s}')
