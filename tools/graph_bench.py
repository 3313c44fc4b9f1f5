#!/usr/bin/env python
"""hipGraph replay vs eager launches of the sampler step (BASELINE.json configs[4]: "hipGraph-captured 50-step DDIM").

    python tools/graph_bench.py [batch ...]        (default: 1 4 16 64; MC_PREC=f32|f16|f16x3)

For every batch size: 50-step DDIM (respacing '15,15,8,6,6') at 196 frames on the 0.125b architecture, three full loops
each way after a warm-up loop, wall-clock per step (host submit + device, synchronised at the end of a loop).  The
graph is ONE captured mc_sample_step with the step index in device memory (mc_ctx_graph_capture); noise is refilled in
place before every step in both arms.  Prints one line per (batch, arm) and the ratio."""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd.diffusion import build_diffusion                     # noqa: E402
from motioncraft_amd.engine import NativeModel                            # noqa: E402
from motioncraft_amd.synthetic import default_dims, make_state_dict      # noqa: E402

PREC = os.environ.get('MC_PREC', 'f32')
dims = default_dims()
nm = NativeModel(dims, make_state_dict(dims, 0), cfg_scale=6.5)
d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                         model_var_type='fixed_large', respace='15,15,8,6,6'))
S = d.num_timesteps
coefs = [d.step_coefs(i, 'ddim', 6.5) for i in range(S)]
stream = torch.cuda.Stream()
for B in [int(v) for v in sys.argv[1:]] or [1, 4, 16, 64]:
    with torch.cuda.stream(stream):
        g = torch.Generator(device='cuda').manual_seed(0)
        ctx = nm.context(B, 196, max_steps=S)
        ctx.set_precision(PREC)
        xf = torch.nn.functional.layer_norm(torch.randn(B, 77, 256, device='cuda', generator=g), (256,))
        ctx.set_timesteps(d.timestep_map)
        ctx.set_condition(xf, torch.ones(B, 196, device='cuda'))
        x = torch.randn(B, 196, 322, device='cuda', generator=g)
        noise = torch.empty_like(x)

        def eager():
            for i in range(S - 1, -1, -1):
                noise.normal_(generator=g)
                ctx.sample_step(x, i, coefs[i], noise, x_prev=x)

        def replay():
            for i in range(S - 1, -1, -1):
                noise.normal_(generator=g)
                ctx.graph_step(i)

        res = {}
        for name, fn in (('eager', eager), ('graph', replay)):
            if name == 'graph':
                ctx.graph_capture(x, noise, coefs)
            fn()
            stream.synchronize()
            ts = []
            for _ in range(3):
                x.normal_(generator=g)
                stream.synchronize()
                t0 = time.perf_counter()
                fn()
                stream.synchronize()
                ts.append((time.perf_counter() - t0) / S * 1e3)
            res[name] = min(ts)
            print(f'B={B:3d} precision={PREC:5s} {name}: {res[name]:.3f} ms/step  ({S * res[name]:.1f} ms per 50-step DDIM, '
                  f'{B * 196 / (S * res[name] * 1e-3):.0f} frames/s)', flush=True)
        print(f'B={B:3d} precision={PREC:5s} graph/eager time ratio {res["graph"] / res["eager"]:.3f}', flush=True)
        ctx.close()
