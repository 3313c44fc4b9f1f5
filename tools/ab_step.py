#!/usr/bin/env python
"""Same-box A/B of per-context kernel-selection options (mc_ctx_set_option) on the headline step (B=64, 196 frames, one
mc_sample_loop step = denoiser + CFG + DDPM update + device noise).  Configurations are given as comma-separated key=value lists,
one context each, timed interleaved over several rounds (box drift hits every configuration alike):

    python tools/ab_step.py "chain=65527" "chain=32759" [--batch 64] [--rounds 6] [--steps 10] [--prec f32]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd.diffusion import build_diffusion
from motioncraft_amd.engine import NativeModel
from motioncraft_amd.synthetic import default_dims, make_state_dict

ap = argparse.ArgumentParser()
ap.add_argument('configs', nargs='+')
ap.add_argument('--batch', type=int, default=64)
ap.add_argument('--frames', type=int, default=196)
ap.add_argument('--rounds', type=int, default=6)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--prec', default='f32')
a = ap.parse_args()

dims = default_dims()
nm = NativeModel(dims, make_state_dict(dims, 0), cfg_scale=6.5)
B, T = a.batch, a.frames
g = torch.Generator().manual_seed(0)
xf = torch.nn.functional.layer_norm(torch.randn(B, 77, 256, generator=g), (256,)).cuda()
mask = torch.ones(B, T).cuda()
diff = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large'))
ctxs, xs = [], []
for cfg in a.configs:
    c = nm.context(B, T, max_steps=8)
    for kv in filter(None, cfg.split(',')):
        k, v = kv.split('=')
        c.set_option(k.strip(), int(v))
    c.set_precision(a.prec)
    c.set_timesteps(diff.timestep_map[-8:])
    c.set_condition(xf, mask)
    ctxs.append(c)
    xs.append(torch.randn(B, T, 322, generator=g).cuda())
coefs = [diff.step_coefs(992 + j, 'ddpm', 6.5) for j in range(8)]


def run(j, n):
    # ONE mc_sample_loop call over n steps (what a real loop is: the sampler update of a step can prepare the next step's operands)
    order = [s % 8 for s in range(n)]
    ctxs[j].sample_loop(xs[j], order, [coefs[s] for s in order], noise=None, seed=7, draw0=0)


for j in range(len(ctxs)):
    run(j, 4)
torch.cuda.synchronize()
ts = [[] for _ in ctxs]
for r in range(a.rounds):
    for j in range(len(ctxs)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(j, a.steps)
        e1.record()
        torch.cuda.synchronize()
        ts[j].append(e0.elapsed_time(e1) / a.steps)
        xs[j].normal_()
for cfg, t in zip(a.configs, ts):
    t = sorted(t)
    print(f'{cfg or "(defaults)":40s} median {t[len(t) // 2]:.3f} ms/step  min {t[0]:.3f}  max {t[-1]:.3f}   (B={B}, {a.prec})')
