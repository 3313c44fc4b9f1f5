#!/usr/bin/env python
"""Quick per-step timing of the denoiser at B=64 (no CPU baseline), for A/B of tuning env vars."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd.engine import NativeModel
from motioncraft_amd.synthetic import make_state_dict, default_dims

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dims = default_dims()
nm = NativeModel(dims, make_state_dict(dims, 0), cfg_scale=6.5)
ctx = nm.context(B, 196, max_steps=4)
PREC = os.environ.get('MC_PREC', 'f32')      # f32 | f16 | f16x3 (mc_ctx_set_precision)
ctx.set_precision(PREC)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 196, 322, generator=g).cuda()
xf = torch.nn.functional.layer_norm(torch.randn(B, 77, 256, generator=g), (256,)).cuda()
mask = torch.ones(B, 196).cuda()
ctx.set_timesteps([999, 500, 57, 0]); ctx.set_condition(xf, mask)
out2 = torch.empty(2 * B, 196, 322, device='cuda')
for _ in range(3): ctx.denoise(x, 0, out2=out2)
torch.cuda.synchronize()
ts = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): ctx.denoise(x, 1, out2=out2)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 4)
ts.sort()
print(f'B={B} precision={PREC} MC_GEMM_TUNE={os.environ.get("MC_GEMM_TUNE","default")}: median {ts[2]:.3f} ms/step  min {ts[0]:.3f}  '
      f'-> {37.08e9*B/ts[2]/1e9:.1f} TFLOP/s, {B*196/ts[2]:.1f} frames/s @1000 steps')
