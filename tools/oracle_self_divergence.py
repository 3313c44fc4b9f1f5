#!/usr/bin/env python
"""Does the REFERENCE disagree with itself at the capacity cut?  (CPU only.)

tutel's batch-prioritised routing (st_attention.py:28-45: capacity_factor 1.5, batch_prioritized_routing) keeps, per expert,
the `capacity` tokens with the largest max gate score.  At B=64 x 196 frames x 12 parts a routing ranks 301 056 tokens, and
neighbouring scores at a capacity threshold are ~1e-6 apart -- the size of fp32 rounding differences between two correct
evaluations of the same gate.  This tool measures that on the oracle (the bit-equal restatement of the reference modules,
oracle/stmogen_oracle.py) alone, no GPU involved: the SAME weights, inputs and noise are run

  base      torch-CPU fp32, T0 threads
  threads   the same code at another thread count (a different GEMM partition, if the BLAS reduces differently)
  permuted  the cosine projector's matmul with its K (=latent) dimension visited in a permuted order
            (x[:, p] @ W[:, p].T: the same real-number product, another fp32 summation order)
  splitk    the projector matmul as the sum of two half-K products

and every variant is compared with `base`: (a) on the FIRST denoiser call (identical input x_T) the number of (token, choice)
pairs whose expert id / keep decision differs, per layer; (b) free-running over `--steps` DDPM steps with identical per-step
noise: max |x - x_base| after every step, and the first step at which it exceeds 1e-3 (the north-star tolerance).

    python tools/oracle_self_divergence.py [--batch 64] [--frames 196] [--steps 10] [--threads 16,32] > profiles/r04_oracle_self_divergence.txt
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from oracle import stmogen_oracle as O, tutel_restated as TR, weights as W    # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--frames', type=int, default=196)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--threads', default='16,32')
a = ap.parse_args()
T0, T1 = (int(v) for v in a.threads.split(','))

dims = W.default_dims()
sd = W.make_state_dict(dims, 0)
B, T = a.batch, a.frames
g = torch.Generator().manual_seed(0)
x_T = torch.randn(B, T, 322, generator=g)
xf = F.layer_norm(torch.randn(B, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],))
mask = torch.ones(B, T)
noise = [torch.randn(B, T, 322, generator=g) for _ in range(a.steps)]
sched = O.Schedule(1000, None)

_orig_gate = TR.gate_scores
_perm = {}


def gate_permuted(x, proj_w, proj_b, sim_matrix, temperature):
    K = x.shape[1]
    if K not in _perm:
        _perm[K] = torch.randperm(K, generator=torch.Generator().manual_seed(123))
    p = _perm[K]
    return _orig_gate(x[:, p].contiguous(), proj_w[:, p].contiguous(), proj_b, sim_matrix, temperature)


def gate_splitk(x, proj_w, proj_b, sim_matrix, temperature):
    import math
    dt = torch.float32
    K = x.shape[1] // 2
    proj = (x[:, :K] @ proj_w[:, :K].t() + x[:, K:] @ proj_w[:, K:].t()) + proj_b
    logits = torch.matmul(F.normalize(proj, dim=1), F.normalize(sim_matrix.to(dt), dim=0))
    logit_scale = torch.clamp(temperature.to(dt), max=math.log(1.0 / 0.01)).exp()
    return F.softmax(logits * logit_scale, dim=1)


VARIANTS = [('base', T0, _orig_gate), ('threads', T1, _orig_gate), ('permuted', T0, gate_permuted), ('splitk', T0, gate_splitk)]


def run(name, threads, gate):
    torch.set_num_threads(threads)
    TR.gate_scores = gate
    try:
        tf = O.precompute_text(sd, xf, dims)
        x = x_T
        traj, routing0 = [], None
        for n in range(a.steps):
            i = 999 - n
            cap = {} if n == 0 else None
            x0 = O.denoise(sd, dims, x, sched.timestep_map[i], xf, mask, text_feats=tf, cap=cap)
            if n == 0:
                routing0 = [(torch.stack(cap[f'layer{l}']['routing']['indices'], 1), torch.stack(cap[f'layer{l}']['routing']['keeps'], 1),
                             cap[f'layer{l}']['routing']['scores'].max(dim=1)[0]) for l in range(dims['NL'])]
            x = O.ddpm_step(sched, i, x, x0, noise[n])
            traj.append(x)
        return traj, routing0
    finally:
        TR.gate_scores = _orig_gate


print(f'# tools/oracle_self_divergence.py --batch {B} --frames {T} --steps {a.steps} --threads {a.threads}   (torch {torch.__version__}, '
      f'{os.cpu_count()} CPUs; oracle = bit-equal restatement of the reference modules, tests/golden/make_golden.py)')
print(f'# tokens per routing: {2 * B * T * dims["H"]} (CFG-doubled), (token, choice) pairs: {4 * B * T * dims["H"]}, '
      f'capacity per expert: {TR.capacity_of(2 * B * T * dims["H"], dims["E"], 2, 1.5)}')
res = {}
for name, th, gate in VARIANTS:
    t0 = time.time()
    res[name] = run(name, th, gate)
    print(f'# variant {name:9s} ({th} threads): {time.time() - t0:.0f} s', flush=True)
base_traj, base_r = res['base']
# how close are neighbouring importance scores at the cut?  (layer 0 of the first call)
half = base_r[0][2].shape[0] // 2
for l in (0, dims['NL'] - 1):
    imp = torch.sort(base_r[l][2][:half] if l == 0 else base_r[l][2]).values      # (layer 0: the second CFG half repeats the first exactly)
    gaps = (imp[1:] - imp[:-1])
    print(f'# layer-{l} importance scores (max gate score per token{", first CFG half" if l == 0 else ""}): median gap between neighbours in rank order '
          f'{float(gaps.median()):.2e}, {int((gaps == 0).sum())} exact ties, fp32 ulp at the median score {float(torch.finfo(torch.float32).eps * imp.median()):.2e}')
print('variant    | first call, identical input: differing (token, choice) pairs per layer [expert id / keep]      | free-running max|x - x_base| per step')
for name, _, _ in VARIANTS[1:]:
    traj, r = res[name]
    flips = ['%d/%d' % (int((r[l][0] != base_r[l][0]).sum()), int((r[l][1] != base_r[l][1]).sum())) for l in range(dims['NL'])]
    dev = [float((traj[n] - base_traj[n]).abs().max()) for n in range(a.steps)]
    first = next((n + 1 for n, d in enumerate(dev) if d > 1e-3), None)
    print(f'{name:10s} | {"  ".join(flips):40s} | ' + ' '.join(f'{d:.1e}' for d in dev) + f' | first step > 1e-3: {first}')
