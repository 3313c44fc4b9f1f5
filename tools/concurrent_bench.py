#!/usr/bin/env python
"""Two (or more) independent batches of B samples in flight on ONE GPU, each on its own HIP stream and context
(the reference's test loop would run them back to back): fills the tile-count tails of one batch's kernels with the
other's.  Reports aggregate frames/s next to the single-stream number."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd.engine import NativeModel
from motioncraft_amd.synthetic import make_state_dict, default_dims

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dims = default_dims()
nm = NativeModel(dims, make_state_dict(dims, 0), cfg_scale=6.5)
g = torch.Generator().manual_seed(0)
streams = [torch.cuda.Stream() for _ in range(NS)]
ctxs, xs, outs = [], [], []
for s in streams:
    with torch.cuda.stream(s):
        c = nm.context(B, 196, max_steps=4)
        x = torch.randn(B, 196, 322, generator=g).cuda()
        xf = torch.nn.functional.layer_norm(torch.randn(B, 77, 256, generator=g), (256,)).cuda()
        c.set_timesteps([999, 500, 57, 0]); c.set_condition(xf, torch.ones(B, 196).cuda())
        ctxs.append(c); xs.append(x); outs.append(torch.empty(2 * B, 196, 322, device='cuda'))
torch.cuda.synchronize()
MODE = sys.argv[3] if len(sys.argv) > 3 else 'denoise'
from motioncraft_amd.diffusion import build_diffusion
COEF = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                            model_var_type='fixed_large')).step_coefs(500, 'ddpm', 6.5)
EPS = torch.randn(B, 196, 322, device='cuda')
nxt = [torch.empty_like(x) for x in xs]


def run(n_streams, reps):
    for _ in range(reps):
        for i in range(n_streams):
            with torch.cuda.stream(streams[i]):
                if MODE == 'denoise':
                    ctxs[i].denoise(xs[i], 1, out2=outs[i])
                else:
                    ctxs[i].sample_step(xs[i], 1, COEF, EPS, x_prev=nxt[i])


for ns in (1, NS):
    run(ns, 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    run(ns, 6)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    print(f'{ns} concurrent batch(es) of {B}: {dt*1e3:.2f} ms per round of steps -> {ns*B*196/dt/1000:.1f} frames/s @1000 steps')
