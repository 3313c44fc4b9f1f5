#!/usr/bin/env python
"""BASELINE configs[0] through the reference API: build_architecture -> MotionDiffusion.forward, B=1, 196 frames, 50-step DDIM,
eager launches vs hipGraph replay (inference_kwargs graph=True; the graph is cached on the context across calls)."""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
import motioncraft_amd as mc
from motioncraft_amd.synthetic import default_dims, make_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = mc.Config.fromfile(os.path.join(ROOT, 'tests', 'configs', 'stmogen_small.py'))
d = default_dims()
m = cfg.model.model
m.max_seq_len, m.latent_dim, m.num_layers, m.time_embed_dim = 196, d['L'] * d['H'], d['NL'], d['Te']
blk = m.ca_block_cfg
blk.latent_dim, blk.text_latent_dim, blk.time_embed_dim, blk.max_seq_len, blk.max_text_seq_len = d['L'], d['Dt'], d['Te'], 196, d['Nt']
blk.ffn_dim = d['F']
m.ffn_cfg.latent_dim, m.ffn_cfg.ffn_dim, m.ffn_cfg.time_embed_dim = d['L'], d['F'], d['Te']
m.text_encoder.latent_dim = d['Dt']
m.pose_encoder_cfg.latent_dim = m.pose_decoder_cfg.latent_dim = d['L']
arch = mc.build_architecture(cfg.model)
arch.load_state_dict({'model.' + k: v for k, v in make_state_dict(arch.model.dims, 0).items()})
T = 196
g = torch.Generator().manual_seed(0)
xf = torch.nn.functional.layer_norm(torch.randn(B, d['Nt'], d['Dt'], generator=g), (d['Dt'],)).cuda()
mask = torch.ones(B, T).cuda()
kw = dict(motion=torch.zeros(B, T, 322).cuda(), motion_mask=mask, motion_length=mask.sum(1, keepdim=True).long(),
          motion_metas=[{'text': ''}] * B, xf_out=xf)
gen = torch.Generator(device='cuda').manual_seed(1)
for name, inf in (('eager', {}), ('graph', {'graph': True})):
    for _ in range(2):
        arch(**kw, inference_kwargs=dict(generator=gen, **inf))
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        arch(**kw, inference_kwargs=dict(generator=gen, **inf))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f'B={B} MotionDiffusion.forward (50-step DDIM, 196 frames, incl. condition hoist + result split) {name}: min {min(ts) * 1e3:.1f} ms  median {sorted(ts)[2] * 1e3:.1f} ms')
