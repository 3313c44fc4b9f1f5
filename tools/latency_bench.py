#!/usr/bin/env python
"""Per-step latency of the full sampler step (mc_sample_step) at small batches: BASELINE configs[0] is B=1, 50-step DDIM."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd.engine import NativeModel
from motioncraft_amd.diffusion import build_diffusion
from motioncraft_amd.synthetic import make_state_dict, default_dims

dims = default_dims()
nm = NativeModel(dims, make_state_dict(dims, 0), cfg_scale=6.5)
d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                         model_var_type='fixed_large', respace='15,15,8,6,6'))
# usage: latency_bench.py [B ...] [--reps N]   (N > 1: that many back-to-back loops, median and last reported)
args = sys.argv[1:]
REPS = 1
if '--reps' in args:
    k = args.index('--reps'); REPS = int(args[k + 1]); del args[k:k + 2]
for B in [int(v) for v in (args or ['1', '2', '4', '8'])]:
    ctx = nm.context(B, 196, max_steps=50)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 196, 322, generator=g).cuda()
    xf = torch.nn.functional.layer_norm(torch.randn(B, 77, 256, generator=g), (256,)).cuda()
    ctx.set_timesteps(d.timestep_map); ctx.set_condition(xf, torch.ones(B, 196).cuda())
    eps = torch.zeros_like(x); nxt = torch.empty_like(x)
    coefs = [d.step_coefs(i, 'ddim', 6.5) for i in range(50)]

    def loop():
        global x, nxt
        for i in range(49, -1, -1):
            ctx.sample_step(x, i, coefs[i], eps, x_prev=nxt)
            x, nxt = nxt, x
    loop(); torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter(); loop(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    print(f'B={B}: 50-step DDIM {dt*1e3:.1f} ms  ({dt*1e3/50:.3f} ms/step, {B*196/dt:.0f} frames/s)' +
          (f'  [{REPS} loops: min {min(ts)*1e3:.1f} max {max(ts)*1e3:.1f} last {ts[-1]*1e3:.1f}]' if REPS > 1 else ''))
    ctx.close()
