#!/usr/bin/env python
"""Per-GPU share of BASELINE configs[2] / [3]: the plug-and-play control branch at its real widths.
  s2g: S2G_Beats2_no_face_loss_025b (L=128, 8 layers, copy_blocks_num=2, raw-audio condition through the WavEncoder),
       batch 256 over 8 GPUs = 32 per GPU, 196 frames, 50-step DDIM
  m2d: M2D_finedance_no_face_loss (L=64, 4 layers, copy_blocks_num=3, 35-d music features), batch 128 over 4 GPUs = 32 per
       GPU, 120-frame windows, 50-step DDIM
Prints the one-off condition encoding time and the sampler-step time."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd.engine import NativeModel
from motioncraft_amd.diffusion import build_diffusion
from motioncraft_amd.synthetic import control_param_shapes, default_dims, make_state_dict, make_wav_encoder_state
from motioncraft_amd.wav_encoder import NativeWavEncoder

case = sys.argv[1] if len(sys.argv) > 1 else 's2g'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if case == 's2g':
    dims, copy, feats, T = default_dims(NL=8), 2, 1536, 196
else:
    dims, copy, feats, T = default_dims(L=64, F=256), 3, 35, 120
nm = NativeModel(dims, make_state_dict(dims, 0, shapes=control_param_shapes(dims, copy, feats)), cfg_scale=dims['scale'])
d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large',
                         respace='15,15,8,6,6'))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, 322, generator=g).cuda()
xf = torch.nn.functional.layer_norm(torch.randn(B, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],)).cuda()
ctx = nm.context(B, T, max_steps=50)
PREC = os.environ.get('MC_PREC', 'f32')          # f32 | f16 | f16x3 (mc_ctx_set_precision; BASELINE configs[4] is the mixed control config in fp16)
for kv in filter(None, os.environ.get('MC_OPTS', '').split(',')):      # per-context options (mc_ctx_set_option), e.g. MC_OPTS=chain=1048567 (no temporal_h_k)
    k_, v_ = kv.split('=')
    ctx.set_option(k_.strip(), int(v_))
ctx.set_precision(PREC)
ctx.set_timesteps(d.timestep_map)
ctx.set_condition(xf, torch.ones(B, T).cuda())
if case == 's2g':
    enc = NativeWavEncoder(feats, 2, make_wav_encoder_state(feats, 2, 0))
    wav = torch.randn(B, 105300, 2, generator=g).cuda()          # 196 condition frames
    c = enc(wav); torch.cuda.synchronize()
    t0 = time.perf_counter(); c = enc(wav); torch.cuda.synchronize()
    print(f'{case}: WavEncoder {B} x 105300 samples -> {tuple(c.shape)} in {(time.perf_counter() - t0) * 1e3:.1f} ms (once per batch)')
else:
    c = torch.randn(B, T, feats, generator=g).cuda()
ctx.set_control(c)
eps, nxt = torch.zeros_like(x), torch.empty_like(x)
coefs = [d.step_coefs(i, 'ddim', dims['scale']) for i in range(50)]


def loop():
    global x, nxt
    for i in range(49, -1, -1):
        ctx.sample_step(x, i, coefs[i], eps, x_prev=nxt)
        x, nxt = nxt, x


loop(); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'{case}: precision {PREC} B={B} T={T} NL={dims["NL"]}+{copy} control copies: 50-step DDIM {dt * 1e3:.1f} ms ({dt * 20:.2f} ms/step) -> '
      f'{B * T / dt:.0f} sampled frames/s per GPU')
