"""State-dict -> packed HBM layouts of libmotioncraft_amd.so.

Input: a flat ``{name: tensor}`` dict with the reference's key names for ``STMoGenTransformer``
(optional ``model.`` prefix of a ``MotionDiffusion`` checkpoint is stripped,
condition-encoder keys ``clip.* / textTransEncoder.* / text_ln.* / text_pre_proj.*`` are ignored:
the text encoder is off the per-step path).  Output: name -> contiguous fp32 numpy array in the
layouts the kernels read (documented per entry below).  Pure host-side, one-off work.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from .synthetic import PART_NAMES, smplx_part_slices  # 322-d motionx part layout (reference stmogen.py:53-68)


def strip_prefix(state_dict):
    out = {}
    for k, v in state_dict.items():
        for pre in ('model.base_model.', 'model.'):
            if k.startswith(pre):
                k = k[len(pre):]
                break
        out[k] = v
    return out


def _f(t):
    return np.ascontiguousarray(t.detach().to(torch.float32).cpu().numpy())


def pack_moe(sd, pre, out, key):
    """tutel moe_layer + MOE wrapper params (st_attention.py:17-47)."""
    m = pre + 'model.'
    emb = sd[pre + 'embedding']                                   # [1, S, G, Din]
    out[key + 'emb'] = _f(emb.reshape(-1, emb.shape[-1]))         # [(s,g), Din]
    out[key + 'gate_w'] = _f(sd[m + 'gates.0.cosine_projector.weight'])   # [256, Din]
    out[key + 'gate_b'] = _f(sd[m + 'gates.0.cosine_projector.bias'])
    sim = sd[m + 'gates.0.sim_matrix'].float()
    out[key + 'sim_n'] = _f(torch.nn.functional.normalize(sim, dim=0))     # [256, E], unit columns
    temp = sd[m + 'gates.0.temperature'].float()
    out[key + 'scale'] = _f(torch.clamp(temp, max=math.log(1.0 / 0.01)).exp().reshape(1))
    out[key + 'fc1_w'] = _f(sd[m + 'experts.batched_fc1_w'])               # [E, 4Din, Din]  (N, K)
    out[key + 'fc1_b'] = _f(sd[m + 'experts.batched_fc1_bias'])
    out[key + 'fc2_wt'] = _f(sd[m + 'experts.batched_fc2_w'].permute(0, 2, 1))   # [E, Din, 4Din] (N, K)
    out[key + 'fc2_b'] = _f(sd[m + 'experts.batched_fc2_bias'])
    out[key + 'proj_w'] = _f(sd[pre + 'proj.weight'])
    out[key + 'proj_b'] = _f(sd[pre + 'proj.bias'])


def pack_state_dict(state_dict, dims):
    """dims: dict(input_feats, max_seq_len, L, H, NL, F, Te, Dt, Nt, E)."""
    sd = strip_prefix(state_dict)
    L, H, NL, C = dims['L'], dims['H'], dims['NL'], dims['input_feats']
    D = L * H
    Cp = (C + 3) // 4 * 4
    sl = smplx_part_slices()
    assert H == len(PART_NAMES) + 1, 'only the 12-part motionx layout is on this path'
    out = OrderedDict()

    # PoseEncoder -> one dense [D, Cp] weight (rows = output channel (part, j); columns = pose channel)
    enc_w = torch.zeros(D, Cp)
    enc_b = torch.zeros(D)
    for p, n in enumerate(PART_NAMES):
        enc_w[p * L:(p + 1) * L, sl[n]] = sd[f'joint_embed.{n}_embed.weight'].float()
        enc_b[p * L:(p + 1) * L] = sd[f'joint_embed.{n}_embed.bias'].float()
    body = [c for n in PART_NAMES for c in sl[n]]
    enc_w[(H - 1) * L:, body] = sd['joint_embed.body_embed.weight'].float()
    enc_b[(H - 1) * L:] = sd['joint_embed.body_embed.bias'].float()
    out['enc.w'], out['enc.b'] = _f(enc_w), _f(enc_b)
    out['seq_emb'] = _f(sd['sequence_embedding'])
    out['time.w0'], out['time.b0'] = _f(sd['time_embed.0.weight']), _f(sd['time_embed.0.bias'])
    out['time.w2'], out['time.b2'] = _f(sd['time_embed.2.weight']), _f(sd['time_embed.2.bias'])

    # PoseDecoder -> one dense [C, D] weight, (scatter + body_out) / 2 folded in; body_out is NOT un-permuted
    dec_w = torch.zeros(C, D)
    dec_b = torch.zeros(C)
    for p, n in enumerate(PART_NAMES):
        dec_w[sl[n], p * L:(p + 1) * L] = 0.5 * sd[f'out.{n}_out.weight'].float()
        dec_b[sl[n]] = 0.5 * sd[f'out.{n}_out.bias'].float()
    dec_w[:, (H - 1) * L:] = 0.5 * sd['out.body_out.weight'].float()
    dec_b += 0.5 * sd['out.body_out.bias'].float()
    out['dec.w'], out['dec.b'] = _f(dec_w), _f(dec_b)

    for i in range(NL):
        ca = f'temporal_decoder_blocks.{i}.ca_block.'
        ff = f'temporal_decoder_blocks.{i}.ffn.'
        k = f'l{i}.'
        out[k + 'norm.g'], out[k + 'norm.b'] = _f(sd[ca + 'norm.weight']), _f(sd[ca + 'norm.bias'])
        out[k + 'text_norm.g'], out[k + 'text_norm.b'] = _f(sd[ca + 'text_norm.weight']), _f(sd[ca + 'text_norm.bias'])
        out[k + 'body_wsm'] = _f(torch.softmax(sd[ca + 'body_weight'].float(), dim=1))
        pack_moe(sd, ca + 'motion_moe.', out, k + 'mm.')
        pack_moe(sd, ca + 'text_moe.', out, k + 'tm.')
        out[k + 'dyn.norm.g'] = _f(sd[ca + 'body_d_attn.norm.weight'])
        out[k + 'dyn.norm.b'] = _f(sd[ca + 'body_d_attn.norm.bias'])
        out[k + 'dyn.qkv_w'] = _f(torch.cat([sd[ca + f'body_d_attn.{n}.weight'] for n in ('query', 'key', 'value')], 0))
        out[k + 'dyn.qkv_b'] = _f(torch.cat([sd[ca + f'body_d_attn.{n}.bias'] for n in ('query', 'key', 'value')], 0))
        for blk, src in (('ca.', ca + 'proj_out.'), ('ffn.', ff + 'proj_out.')):
            out[k + blk + 'film_w'] = _f(sd[src + 'emb_layers.1.weight'])
            out[k + blk + 'film_b'] = _f(sd[src + 'emb_layers.1.bias'])
            out[k + blk + 'ln_g'], out[k + blk + 'ln_b'] = _f(sd[src + 'norm.weight']), _f(sd[src + 'norm.bias'])
            out[k + blk + 'out_w'] = _f(sd[src + 'out_layers.2.weight'])
            out[k + blk + 'out_b'] = _f(sd[src + 'out_layers.2.bias'])
        out[k + 'ffn.w1'] = _f(torch.stack([sd[ff + f'linear1_list.{p}.weight'] for p in range(H)]))   # [H, F, L]
        out[k + 'ffn.b1'] = _f(torch.stack([sd[ff + f'linear1_list.{p}.bias'] for p in range(H)]))
        out[k + 'ffn.w2'] = _f(torch.stack([sd[ff + f'linear2_list.{p}.weight'] for p in range(H)]))   # [H, L, F]
        out[k + 'ffn.b2'] = _f(torch.stack([sd[ff + f'linear2_list.{p}.bias'] for p in range(H)]))
    return out
