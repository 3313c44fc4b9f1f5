#!/usr/bin/env python
"""Thin counterpart of the reference's tools/visualize.py for the MI355X path: config + checkpoint -> sampled motion
-> de-normalised SMPL-X .npz (motionx) or raw .npy, everything between the condition features and the finished
arrays on the device.

    python tools/sample.py CONFIG CHECKPOINT --text "a person walks" --motion_length 120 --out ./samples \\
        [--clip_feat feats.npy | --xf_out xf.npy | --random-condition SEED]  [--mean mean.npy --std std.npy]

The CLIP tokenizer is not available offline: prompts only name the output file unless the `clip` package is importable
(then they are tokenized and encoded by the device CLIP tower when the checkpoint carries clip.* weights).
CHECKPOINT may be "synthetic[:SEED]" for deterministic random-init weights of the configured architecture.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)

import motioncraft_amd as mc                                    # noqa: E402
from motioncraft_amd import postprocess, synthetic              # noqa: E402
from motioncraft_amd.checkpoint import load_checkpoint          # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='motioncraft_amd sampling')
    p.add_argument('config')
    p.add_argument('checkpoint')
    p.add_argument('--text', nargs='+', required=True)
    p.add_argument('--motion_length', type=int, nargs='+', required=True)
    p.add_argument('--out', default='./samples')
    p.add_argument('--clip_feat', help='.npy [n,77,512] CLIP text features (ln_final output)')
    p.add_argument('--xf_out', help='.npy [n,77,text_latent_dim] frozen condition embedding')
    p.add_argument('--random-condition', type=int, default=None, metavar='SEED')
    p.add_argument('--mean'), p.add_argument('--std')
    p.add_argument('--seed', type=int, default=0)
    # long-sequence / RePaint options read by the sampler through cfg.model['opt'] (tools/visualize.py:96-110)
    p.add_argument('--repaint', action='store_true'), p.add_argument('--overlap_len', type=int, default=0)
    p.add_argument('--same_overlap_noisy', action='store_true'), p.add_argument('--no_resample', action='store_true')
    p.add_argument('--timestep_respacing', default='ddim50'), p.add_argument('--jump_n_sample', type=int, default=5)
    p.add_argument('--jump_length', type=int, default=3), p.add_argument('--addBlend', type=bool, default=True)
    p.add_argument('--no_repaint', action='store_true')
    # MI355X options: fp16 MFMA (also switched on by a top-level `fp16 = dict(...)` in the config, tools/test.py:95-97), replay
    p.add_argument('--fp16', choices=['split', 'plain'], default=None, help='fp16-MFMA mode: split = fp32-class hi/lo form')
    p.add_argument('--graph', action='store_true', help='hipGraph replay of the sampler step')
    return p.parse_args()


def main():
    a = parse_args()
    assert len(a.text) == len(a.motion_length)
    cfg = mc.Config.fromfile(a.config)
    cfg.model['opt'] = a
    model = mc.build_architecture(cfg.model)
    if a.checkpoint.startswith('synthetic'):
        seed = int(a.checkpoint.split(':')[1]) if ':' in a.checkpoint else 0
        model.load_state_dict({'model.' + k: v for k, v in synthetic.make_state_dict(model.model.dims, seed).items()})
    else:
        load_checkpoint(model, a.checkpoint, map_location='cpu')
    model.eval()
    if a.fp16 or cfg.get('fp16', None) is not None:
        mc.wrap_fp16_model(model, split=(a.fp16 != 'plain'))
    dims = model.model.dims
    n, T, C = len(a.text), max(a.motion_length), dims['input_feats']
    if not 1 <= T <= dims['max_seq_len']:
        raise ValueError(f'motion_length must be in [1, {dims["max_seq_len"]}]')
    dev = torch.device('cuda', torch.cuda.current_device())
    mask = torch.zeros(n, T, device=dev)
    for i, m in enumerate(a.motion_length):
        mask[i, :m] = 1
    kw = dict(motion=torch.zeros(n, T, C, device=dev), motion_mask=mask,
              motion_length=torch.tensor(a.motion_length, device=dev).long(), num_intervals=n,
              motion_metas=[{'text': t} for t in a.text],
              inference_kwargs=dict(generator=torch.Generator(device=dev).manual_seed(a.seed), **({'graph': True} if a.graph else {})))
    if a.xf_out:
        kw['xf_out'] = torch.from_numpy(np.load(a.xf_out)).float().to(dev)
    elif a.clip_feat:
        kw['clip_feat'] = torch.from_numpy(np.load(a.clip_feat)).float().to(dev)
    elif a.random_condition is not None:
        g = torch.Generator().manual_seed(a.random_condition)
        kw['xf_out'] = torch.nn.functional.layer_norm(torch.randn(n, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],)).to(dev)
    out = model(**kw)
    os.makedirs(a.out, exist_ok=True)
    mean = np.load(a.mean) if a.mean else None
    std = np.load(a.std) if a.std else None
    if dims.get('dataset', 'motionx') == 'motionx':
        pred = torch.stack([o['pred_motion'] for o in out]).to(dev).contiguous()
        path = postprocess.save_smplx_npz(a.out, a.text[0], pred, a.motion_length, mean, std)
    else:
        arrs = [o['pred_motion'][:m].numpy() * (std if std is not None else 1.0) + (mean if mean is not None else 0.0)
                for o, m in zip(out, a.motion_length)]
        path = os.path.join(a.out, postprocess.result_name(a.text[0], a.motion_length[0]) + '.npy')
        np.save(path, np.concatenate(arrs, axis=0))
    print(f'pred_motion: {n} x {T} x {C} -> {path}')


if __name__ == '__main__':
    main()
