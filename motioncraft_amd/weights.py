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

from .synthetic import part_layout  # pose-vector part layouts (reference stmogen.py:12-112,188-230)


def strip_prefix(state_dict):
    """Normalise checkpoint key names: drop the ``model.`` attribute prefix of ``MotionDiffusion``
    (diffusion_architecture.py:83) and the ``base_model.`` prefix a ``ControlT2MHalf`` wrapper adds
    (controlnet.py:427-439); ``controlnet.*`` / ``control_cond_input.*`` keys are kept as they are."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith('model.'):
            k = k[len('model.'):]
        if k.startswith('base_model.'):
            k = k[len('base_model.'):]
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
    sim_n = torch.nn.functional.normalize(sim, dim=0)
    out[key + 'sim_n'] = _f(sim_n)                                         # [256, E], unit columns
    simT = torch.zeros(32, 256)                                            # transposed + zero rows: MFMA "A" operand of gate_k
    simT[:sim_n.shape[1]] = sim_n.T
    out[key + 'sim_nT'] = _f(simT)
    temp = sd[m + 'gates.0.temperature'].float()
    out[key + 'scale'] = _f(torch.clamp(temp, max=math.log(1.0 / 0.01)).exp().reshape(1))
    out[key + 'fc1_w'] = _f(sd[m + 'experts.batched_fc1_w'])               # [E, 4Din, Din]  (N, K)
    out[key + 'fc1_b'] = _f(sd[m + 'experts.batched_fc1_bias'])
    out[key + 'fc2_wt'] = _f(sd[m + 'experts.batched_fc2_w'].permute(0, 2, 1))   # [E, Din, 4Din] (N, K)
    out[key + 'fc2_b'] = _f(sd[m + 'experts.batched_fc2_bias'])
    out[key + 'proj_w'] = _f(sd[pre + 'proj.weight'])
    out[key + 'proj_b'] = _f(sd[pre + 'proj.bias'])


def pack_layer(sd, src, out, k, H):
    """One DecoderLayer (STMA ca_block + SFFN ffn) -> packed entries with prefix `k`."""
    ca, ff = src + 'ca_block.', src + 'ffn.'
    out[k + 'norm.g'], out[k + 'norm.b'] = _f(sd[ca + 'norm.weight']), _f(sd[ca + 'norm.bias'])
    out[k + 'text_norm.g'], out[k + 'text_norm.b'] = _f(sd[ca + 'text_norm.weight']), _f(sd[ca + 'text_norm.bias'])
    out[k + 'body_wsm'] = _f(torch.softmax(sd[ca + 'body_weight'].float(), dim=1))
    pack_moe(sd, ca + 'motion_moe.', out, k + 'mm.')
    pack_moe(sd, ca + 'text_moe.', out, k + 'tm.')
    out[k + 'dyn.norm.g'] = _f(sd[ca + 'body_d_attn.norm.weight'])
    out[k + 'dyn.norm.b'] = _f(sd[ca + 'body_d_attn.norm.bias'])
    out[k + 'dyn.qkv_w'] = _f(torch.cat([sd[ca + f'body_d_attn.{n}.weight'] for n in ('query', 'key', 'value')], 0))
    out[k + 'dyn.qkv_b'] = _f(torch.cat([sd[ca + f'body_d_attn.{n}.bias'] for n in ('query', 'key', 'value')], 0))
    for blk, s_ in (('ca.', ca + 'proj_out.'), ('ffn.', ff + 'proj_out.')):
        out[k + blk + 'film_w'] = _f(sd[s_ + 'emb_layers.1.weight'])
        out[k + blk + 'film_b'] = _f(sd[s_ + 'emb_layers.1.bias'])
        out[k + blk + 'ln_g'], out[k + blk + 'ln_b'] = _f(sd[s_ + 'norm.weight']), _f(sd[s_ + 'norm.bias'])
        out[k + blk + 'out_w'] = _f(sd[s_ + 'out_layers.2.weight'])
        out[k + blk + 'out_b'] = _f(sd[s_ + 'out_layers.2.bias'])
    out[k + 'ffn.w1'] = _f(torch.stack([sd[ff + f'linear1_list.{p}.weight'] for p in range(H)]))   # [H, F, L]
    out[k + 'ffn.b1'] = _f(torch.stack([sd[ff + f'linear1_list.{p}.bias'] for p in range(H)]))
    out[k + 'ffn.w2'] = _f(torch.stack([sd[ff + f'linear2_list.{p}.weight'] for p in range(H)]))   # [H, L, F]
    out[k + 'ffn.b2'] = _f(torch.stack([sd[ff + f'linear2_list.{p}.bias'] for p in range(H)]))


def control_info(state_dict):
    """(copy_blocks_num, control_cond_feats) found in a (possibly prefixed) state dict; (0, 0) if none."""
    sd = strip_prefix(state_dict)
    n = 0
    while f'controlnet.{n}.after_proj.weight' in sd:
        n += 1
    return (n, int(sd['control_cond_input.weight'].shape[1])) if n else (0, 0)


def pack_state_dict(state_dict, dims):
    """dims: dict(input_feats, max_seq_len, L, H, NL, F, Te, Dt, Nt, E)."""
    sd = strip_prefix(state_dict)
    L, H, NL, C = dims['L'], dims['H'], dims['NL'], dims['input_feats']
    D = L * H
    Cp = (C + 31) // 32 * 32        # pose channels padded to the GEMM's k-step: the library pads x_t rows to match
    names, sl, body = part_layout(dims.get('dataset', 'motionx'))
    assert H == len(names) + 1 and len(body) == C, 'num_heads / input_feats do not match the dataset part layout'
    out = OrderedDict()

    # PoseEncoder -> one dense [D, Cp] weight (rows = output channel (part, j); columns = pose channel)
    enc_w = torch.zeros(D, Cp)
    enc_b = torch.zeros(D)
    for p, n in enumerate(names):
        enc_w[p * L:(p + 1) * L, sl[n]] = sd[f'joint_embed.{n}_embed.weight'].float()
        enc_b[p * L:(p + 1) * L] = sd[f'joint_embed.{n}_embed.bias'].float()
    enc_w[(H - 1) * L:, body] = sd['joint_embed.body_embed.weight'].float()
    enc_b[(H - 1) * L:] = sd['joint_embed.body_embed.bias'].float()
    out['enc.w'], out['enc.b'] = _f(enc_w), _f(enc_b)
    out['seq_emb'] = _f(sd['sequence_embedding'])
    out['time.w0'], out['time.b0'] = _f(sd['time_embed.0.weight']), _f(sd['time_embed.0.bias'])
    out['time.w2'], out['time.b2'] = _f(sd['time_embed.2.weight']), _f(sd['time_embed.2.bias'])

    # PoseDecoder -> one dense [C, D] weight, (scatter + body_out) / 2 folded in; body_out is NOT un-permuted
    dec_w = torch.zeros(C, D)
    dec_b = torch.zeros(C)
    for p, n in enumerate(names):
        dec_w[sl[n], p * L:(p + 1) * L] = 0.5 * sd[f'out.{n}_out.weight'].float()
        dec_b[sl[n]] = 0.5 * sd[f'out.{n}_out.bias'].float()
    dec_w[:, (H - 1) * L:] = 0.5 * sd['out.body_out.weight'].float()
    dec_b += 0.5 * sd['out.body_out.bias'].float()
    out['dec.w'], out['dec.b'] = _f(dec_w), _f(dec_b)
    # the Linear of the LAST StylizationBlock (h += a W^T + b) followed by the affine decoder, folded (in fp64) into one
    # [C, D] matrix for the sampler entry points:  dec(h + a W^T + b) = dec(h) + a (Wd W)^T + Wd b
    lastp = f'temporal_decoder_blocks.{NL - 1}.ffn.proj_out.out_layers.2.'
    Wl, bl = sd[lastp + 'weight'].double(), sd[lastp + 'bias'].double()
    out['dec.wf'] = _f((dec_w.double() @ Wl).float())
    out['dec.bf'] = _f((dec_b.double() + dec_w.double() @ bl).float())

    for i in range(NL):
        pack_layer(sd, f'temporal_decoder_blocks.{i}.', out, f'l{i}.', H)
    # plug-and-play control branch (ControlT2MHalf, controlnet.py:107-183)
    ncopy = 0
    while f'controlnet.{ncopy}.after_proj.weight' in sd:
        ncopy += 1
    for j in range(ncopy):
        pack_layer(sd, f'controlnet.{j}.copied_block.', out, f'c{j}.', H)
        if j == 0:
            out['c0.before_w'] = _f(sd['controlnet.0.before_proj.weight'])
            out['c0.before_b'] = _f(sd['controlnet.0.before_proj.bias'])
        out[f'c{j}.after_w'] = _f(sd[f'controlnet.{j}.after_proj.weight'])
        out[f'c{j}.after_b'] = _f(sd[f'controlnet.{j}.after_proj.bias'])
    if ncopy:
        w = sd['control_cond_input.weight'].float()                      # [D, Fc] -> rows padded to a multiple of 4
        fc = w.shape[1]
        wp = torch.zeros(w.shape[0], (fc + 3) // 4 * 4)
        wp[:, :fc] = w
        out['ctrl_in.w'], out['ctrl_in.b'] = _f(wp), _f(sd['control_cond_input.bias'])
    return out
