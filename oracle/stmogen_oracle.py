"""CPU restatement of the STMoGen per-step sampling path.  TEST INFRASTRUCTURE ONLY.

Pure torch-CPU, functional, weights passed as a flat ``{state-dict key: tensor}``
dict with the reference's key names.  No reference
imports, so this file travels to the GPU box where it is the checker for the HIP
path and the ``cpu_baseline`` leg of ``bench.py``.  It is pinned in the build
container against the reference's own modules by ``tests/golden/make_golden.py``
(max-abs <= 1e-5) and against the committed fixtures by ``tests/test_oracle.py``.

PARITY UNPINNED for the tutel MoE boundary only (see ``oracle/tutel_restated.py``):
the MoE arithmetic is a restatement of a third-party dependency the reference
neither vendors nor tests.

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import tutel_restated
from .weights import part_layout


# ------------------------------------------------------------------------------------
# a8: timestep embedding + time MLP
# ------------------------------------------------------------------------------------
def timestep_embedding(timesteps, dim, max_period=10000):
    """mogen/models/utils/position_encoding.py:42-60 (cos || sin)."""
    half = dim // 2
    idx = torch.arange(start=0, end=half, dtype=torch.float32)
    freqs = torch.exp(-math.log(max_period) * idx / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def time_embed(p, timesteps, D):
    """mogen/models/transformers/diffusion_transformer.py:89-93,206-208."""
    e = timestep_embedding(timesteps, D).to(p['time_embed.0.weight'].dtype)
    e = F.linear(e, p['time_embed.0.weight'], p['time_embed.0.bias'])
    e = F.silu(e)
    return F.linear(e, p['time_embed.2.weight'], p['time_embed.2.bias'])


# ------------------------------------------------------------------------------------
# a9 / a14: pose encoder / decoder (motionx, joints=False, patch_size 1)
# ------------------------------------------------------------------------------------
def pose_encoder(p, motion, dataset='motionx'):
    """mogen/models/transformers/stmogen.py:336-353 (motionx), :354-365 (human_ml3d / kit_ml), :376-378."""
    names, sl, body = part_layout(dataset)
    feats = [F.linear(motion[:, :, sl[n]], p[f'joint_embed.{n}_embed.weight'], p[f'joint_embed.{n}_embed.bias'])
             for n in names]
    feats.append(F.linear(motion[:, :, body], p['joint_embed.body_embed.weight'],
                          p['joint_embed.body_embed.bias']))
    return torch.cat(feats, dim=-1)


def pose_decoder(p, h, L, out_dim=322, dataset='motionx'):
    """mogen/models/transformers/stmogen.py:505-544 (motionx), :545-562 (human_ml3d / kit_ml).
    NB body_out is added WITHOUT un-permuting."""
    names, sl, _ = part_layout(dataset)
    B, T, _ = h.shape
    out = torch.zeros(B, T, out_dim, dtype=h.dtype)
    for i, n in enumerate(names):
        out[:, :, sl[n]] = F.linear(h[:, :, i * L:(i + 1) * L], p[f'out.{n}_out.weight'], p[f'out.{n}_out.bias'])
    body = F.linear(h[:, :, len(names) * L:], p['out.body_out.weight'], p['out.body_out.bias'])
    return (out + body) / 2.0


# ------------------------------------------------------------------------------------
# a16 + MOE wrapper
# ------------------------------------------------------------------------------------
CAPACITY_FACTOR = 1.5      # st_attention.py:33 (hard-coded in the reference; tests shrink it to reach the overflow branches)


def moe_wrapper(p, pre, z, return_routing=False, forced=None):
    """mogen/models/attentions/st_attention.py:49-56 (class MOE.forward); tutel boundary a16."""
    B, S, G, Din = z.shape
    x = (z + p[pre + 'embedding'][:, :S]).reshape(-1, Din)
    m = pre + 'model.'
    r = tutel_restated.moe_forward(
        x, p[m + 'gates.0.cosine_projector.weight'], p[m + 'gates.0.cosine_projector.bias'],
        p[m + 'gates.0.sim_matrix'], p[m + 'gates.0.temperature'],
        p[m + 'experts.batched_fc1_w'], p[m + 'experts.batched_fc1_bias'],
        p[m + 'experts.batched_fc2_w'], p[m + 'experts.batched_fc2_bias'],
        top_k=2, capacity_factor=CAPACITY_FACTOR, batch_prioritized_routing=True, return_routing=return_routing, forced=forced)
    y, routing = r if return_routing else (r, None)
    y = F.linear(F.gelu(y), p[pre + 'proj.weight'], p[pre + 'proj.bias']).reshape(B, S, G, -1)
    return (y, routing) if return_routing else y


# ------------------------------------------------------------------------------------
# a12.6: StylizationBlock
# ------------------------------------------------------------------------------------
def stylization(p, pre, h, emb):
    """mogen/models/utils/stylization_block.py:29-40."""
    D = h.shape[-1]
    emb_out = F.linear(F.silu(emb), p[pre + 'emb_layers.1.weight'], p[pre + 'emb_layers.1.bias']).unsqueeze(1)
    scale, shift = torch.chunk(emb_out, 2, dim=2)
    h = F.layer_norm(h, (D,), p[pre + 'norm.weight'], p[pre + 'norm.bias']) * (1 + scale) + shift
    return F.linear(F.silu(h), p[pre + 'out_layers.2.weight'], p[pre + 'out_layers.2.bias'])


# ------------------------------------------------------------------------------------
# a12.4: dynamic body topology = EfficientSelfAttention(heads=8, time_embed_dim=None), mask == 1
# ------------------------------------------------------------------------------------
def efficient_self_attention(p, pre, x, heads=8):
    """mogen/models/attentions/efficient_attention.py:25-46 with src_mask == ones."""
    B, T, D = x.shape
    n = F.layer_norm(x, (D,), p[pre + 'norm.weight'], p[pre + 'norm.bias'])
    q = F.linear(n, p[pre + 'query.weight'], p[pre + 'query.bias'])
    k = F.linear(n, p[pre + 'key.weight'], p[pre + 'key.bias'])
    v = F.linear(n, p[pre + 'value.weight'], p[pre + 'value.bias']).view(B, T, heads, -1)
    q = F.softmax(q.view(B, T, heads, -1), dim=-1)
    k = F.softmax(k.view(B, T, heads, -1), dim=1)
    att = torch.einsum('bnhd,bnhl->bhdl', k, v)
    y = torch.einsum('bnhd,bhdl->bnhl', q, att).reshape(B, T, D)
    return x + y


# ------------------------------------------------------------------------------------
# a12: STMA (MC-Attn)
# ------------------------------------------------------------------------------------
def text_kv(p, pre, xf, dims):
    """a12.2, st_attention.py:116-118: step-invariant text K/V of one layer: [B2, Nt, 1, 2L]."""
    B, Nt, Dt = xf.shape
    tf = F.layer_norm(xf.reshape(B, Nt, 1, Dt), (Dt,), p[pre + 'text_norm.weight'], p[pre + 'text_norm.bias'])
    return moe_wrapper(p, pre + 'text_moe.', tf)


def stma(p, pre, x, xf, emb, src_mask, cond_type, dims, text_feat=None, cap=None, forced=None):
    """mogen/models/attentions/st_attention.py:105-179.
    x [B,T,D], xf [B,Nt,Dt], emb [B,Te], src_mask [B,T] or [B,T,1], cond_type [B,1,1]."""
    B, T, D = x.shape
    H, L = dims['H'], dims['L']
    x4 = x.reshape(B, T, H, L)
    if text_feat is None:
        text_feat = text_kv(p, pre, xf, dims)
    mn = F.layer_norm(x4, (L,), p[pre + 'norm.weight'], p[pre + 'norm.bias'])
    if cap is not None:
        motion_feat, routing = moe_wrapper(p, pre + 'motion_moe.', mn, return_routing=True, forced=forced)
        cap['routing'] = routing
    else:
        motion_feat = moe_wrapper(p, pre + 'motion_moe.', mn, forced=forced)
    if cap is not None:
        cap['motion_feat'] = motion_feat
        cap['text_feat'] = text_feat
    body_weight = F.softmax(p[pre + 'body_weight'], dim=1)
    body_value = motion_feat[..., :L]
    body_feat = torch.einsum('hl,bnld->bnhd', body_weight, body_value).reshape(B, T, D)
    d_body = efficient_self_attention(p, pre + 'body_d_attn.', body_value.reshape(B * T, H, L),
                                      heads=dims.get('dyn_heads', 8)).reshape(B, T, D)
    body_feat = body_feat + d_body
    if cap is not None:
        cap['y_s'] = body_feat

    tct = (cond_type % 10 > 0).to(x.dtype).reshape(B, 1, 1).unsqueeze(-1)   # [B,1,1,1]
    mask = src_mask.reshape(B, T, 1, 1).to(x.dtype)
    key_text = (text_feat[..., :L] + (1 - tct) * -1000000).repeat(1, 1, H, 1)
    key_motion = motion_feat[..., L:2 * L] + (1 - mask) * -1000000
    key = F.softmax(torch.cat((key_text, key_motion), dim=1), dim=1)
    value_text = (text_feat[..., L:] * tct).repeat(1, 1, H, 1)
    value_motion = motion_feat[..., 2 * L:3 * L] * mask
    value = torch.cat((value_text, value_motion), dim=1)
    query = F.softmax(motion_feat[..., 3 * L:], dim=-1)
    attention = torch.einsum('bnhd,bnhl->bhdl', key, value)
    y_t = torch.einsum('bnhd,bhdl->bnhl', query, attention).reshape(B, T, D)
    if cap is not None:
        cap['y_t'] = y_t
    return x + stylization(p, pre + 'proj_out.', body_feat + y_t, emb)


# ------------------------------------------------------------------------------------
# a13: SFFN
# ------------------------------------------------------------------------------------
def sffn(p, pre, x, emb, dims, cap=None):
    """mogen/models/transformers/stmogen.py:596-607."""
    B, T, D = x.shape
    H, L = dims['H'], dims['L']
    x4 = x.reshape(B, T, H, L)
    outs = []
    for i in range(H):
        f = F.gelu(F.linear(x4[:, :, i], p[pre + f'linear1_list.{i}.weight'], p[pre + f'linear1_list.{i}.bias']))
        outs.append(F.linear(f, p[pre + f'linear2_list.{i}.weight'], p[pre + f'linear2_list.{i}.bias']))
    y = torch.cat(outs, dim=-1)
    if cap is not None:
        cap['ffn_z'] = y
    return x + stylization(p, pre + 'proj_out.', y, emb)


# ------------------------------------------------------------------------------------
# a10/a11: the denoiser (DiffusionTransformer.forward + STMoGenTransformer.forward_test)
# ------------------------------------------------------------------------------------
def precompute_text(p, xf_out, dims):
    """Step-invariant hoist: per-layer text K/V for the
    CFG-doubled batch -- identical arithmetic to calling text_moe inside every step."""
    xf2 = xf_out.repeat(2, 1, 1)
    return [text_kv(p, f'temporal_decoder_blocks.{i}.ca_block.', xf2, dims) for i in range(dims['NL'])]


def control_forward_c(p, c, T):
    """ControlT2MHalf.forward_c (controlnet.py:186-199), condition_pre_encoder == identity:
    control_cond_input -> zero-pad to T frames -> + sequence_embedding on the first Tc frames."""
    c = F.linear(c, p['control_cond_input.weight'], p['control_cond_input.bias'])
    B, Tc, D = c.shape
    c_new = torch.cat([c, torch.zeros(B, T - Tc, D, dtype=c.dtype)], dim=-2)
    c_new[:, :Tc] = c_new[:, :Tc] + p['base_model.sequence_embedding'].unsqueeze(0)[:, :Tc]
    return c_new


def precompute_text_control(p, xf_out, dims, copy_blocks_num):
    """The step-invariant text K/V of every layer slot of the control wrapper (base layers 0..NL-1, control copy j at
    NL + j), routed over the CFG-doubled condition batch like `precompute_text` -- lets a lockstep test evaluate a
    SUB-batch of samples with the text features the full batch produced (the text MoE's capacity couples the batch)."""
    xf2 = xf_out.repeat(2, 1, 1)
    out = {i: text_kv(p, f'base_model.temporal_decoder_blocks.{i}.ca_block.', xf2, dims) for i in range(dims['NL'])}
    for j in range(copy_blocks_num):
        out[dims['NL'] + j] = text_kv(p, f'controlnet.{j}.copied_block.ca_block.', xf2, dims)
    return out


def denoise_control(p, dims, x_t, t_orig, xf_out, motion_mask, c, copy_blocks_num, condition_cfg=True, cap=None,
                    forced_routing=None, text_feats=None):
    """ControlT2MHalf.forward + forward_test (controlnet.py:201-266, 340-424) with a condition `c`
    [B, Tc, cond_feats]; `p` uses the wrapper's key names (base_model.* / controlnet.* / control_cond_input.*).
    ``forced_routing`` / ``cap['routing'][slot]``: per layer slot, base layers 0..NL-1 first, control copy j at NL + j."""
    B, T, C = x_t.shape
    L, H, NL = dims['L'], dims['H'], dims['NL']
    D = L * H
    pb = {k[len('base_model.'):]: v for k, v in p.items() if k.startswith('base_model.')}
    ts = torch.full((B,), int(t_orig), dtype=torch.long)
    emb = time_embed(pb, ts, D)
    h = pose_encoder(pb, x_t, dims.get('dataset', 'motionx'))
    cc = control_forward_c(p, c, T)
    h = h + pb['sequence_embedding'].unsqueeze(0)[:, :T, :]
    cond = torch.cat((torch.ones(B, 1, 1), torch.zeros(B, 1, 1)), dim=0)
    h = h.repeat(2, 1, 1)
    xf2, emb2 = xf_out.repeat(2, 1, 1), emb.repeat(2, 1)
    mask2 = motion_mask.reshape(B, T).repeat(2, 1)

    if cap is not None:
        cap['routing'] = {}

    def layer(pp, pre, x, slot):
        lcap = {} if cap is not None else None
        x = stma(pp, pre + 'ca_block.', x, xf2, emb2, mask2, cond, dims, cap=lcap,
                 text_feat=None if text_feats is None else text_feats[slot],
                 forced=None if forced_routing is None else forced_routing[slot])
        if cap is not None:
            cap['routing'][slot] = lcap['routing']
        return sffn(pp, pre + 'ffn.', x, emb2, dims)

    h = layer(pb, 'temporal_decoder_blocks.0.', h, 0)
    cc = cc.repeat(2, 1, 1)
    if condition_cfg:
        cc = cc * cond
    for index in range(1, copy_blocks_num + 1):
        j = index - 1
        pre = f'controlnet.{j}.'
        if j == 0:                                                      # ControlT2MBlock.forward (controlnet.py:53-88)
            cc = F.linear(cc, p[pre + 'before_proj.weight'], p[pre + 'before_proj.bias'])
            cc = layer(p, pre + 'copied_block.', h + cc, NL + j)
        else:
            cc = layer(p, pre + 'copied_block.', cc, NL + j)
        c_skip = F.linear(cc, p[pre + 'after_proj.weight'], p[pre + 'after_proj.bias'])
        if cap is not None:
            cap[f'c_skip{j}'] = c_skip
        h = layer(pb, f'temporal_decoder_blocks.{index}.', h + c_skip, index)
    for index in range(copy_blocks_num + 1, NL):
        h = layer(pb, f'temporal_decoder_blocks.{index}.', h, index)
    out = pose_decoder(pb, h, L, C, dims.get('dataset', 'motionx')).view(2 * B, T, -1)
    if cap is not None:
        cap['out2'] = out
    w = (1 - (1000 - int(t_orig)) / 1000) * dims['scale'] + 1
    return out[:B] * w + out[B:] * (1 - w)


def denoise(p, dims, x_t, t_orig, xf_out, motion_mask, text_feats=None, cap=None, forced_routing=None,
            capacity_factor=None):
    """``_denoise`` with the module-level tutel capacity factor temporarily replaced (test knob)."""
    global CAPACITY_FACTOR
    old = CAPACITY_FACTOR
    if capacity_factor is not None:
        CAPACITY_FACTOR = float(capacity_factor)
    try:
        return _denoise(p, dims, x_t, t_orig, xf_out, motion_mask, text_feats, cap, forced_routing)
    finally:
        CAPACITY_FACTOR = old


def _denoise(p, dims, x_t, t_orig, xf_out, motion_mask, text_feats=None, cap=None, forced_routing=None):
    """mogen/models/transformers/diffusion_transformer.py:186-238 + stmogen.py:725-761.
    x_t [B,T,C]; t_orig: int original (un-spaced) timestep, identical for the batch;
    returns the CFG-combined x0 prediction [B,T,C]."""
    B, T, C = x_t.shape
    L, H, NL = dims['L'], dims['H'], dims['NL']
    D = L * H
    ts = torch.full((B,), int(t_orig), dtype=torch.long)
    emb = time_embed(p, ts, D)
    h = pose_encoder(p, x_t, dims.get('dataset', 'motionx')) + p['sequence_embedding'].unsqueeze(0)[:, :T, :]
    if cap is not None:
        cap['h0'] = h
        cap['emb'] = emb
    cond = torch.cat((torch.ones(B, 1, 1), torch.zeros(B, 1, 1)), dim=0)
    h = h.repeat(2, 1, 1)
    xf2 = xf_out.repeat(2, 1, 1)
    emb2 = emb.repeat(2, 1)
    mask2 = motion_mask.reshape(B, T).repeat(2, 1)
    for i in range(NL):
        lcap = {} if cap is not None else None
        pre = f'temporal_decoder_blocks.{i}.'
        h = stma(p, pre + 'ca_block.', h, xf2, emb2, mask2, cond, dims,
                 text_feat=None if text_feats is None else text_feats[i], cap=lcap,
                 forced=None if forced_routing is None else forced_routing[i])
        if lcap is not None:
            lcap['after_stma'] = h
        h = sffn(p, pre + 'ffn.', h, emb2, dims, cap=lcap)
        if lcap is not None:
            lcap['after_ffn'] = h
            cap[f'layer{i}'] = lcap
    out = pose_decoder(p, h, L, C, dims.get('dataset', 'motionx')).view(2 * B, T, -1)
    if cap is not None:
        cap['out2'] = out
    w = (1 - (1000 - int(t_orig)) / 1000) * dims['scale'] + 1       # stmogen.py:655-659
    return out[:B] * w + out[B:] * (1 - w)


# ------------------------------------------------------------------------------------
# a1-a6: schedules and samplers
# ------------------------------------------------------------------------------------
def linear_betas(num_steps=1000):
    """gaussian_diffusion.py:235-252."""
    scale = 1000 / num_steps
    return np.linspace(scale * 0.0001, scale * 0.02, num_steps, dtype=np.float64)


def space_timesteps(num_timesteps, section_counts):
    """gaussian_diffusion.py:1346-1404 (list/'a,b,c' form and 'ddimN')."""
    if isinstance(section_counts, str):
        if section_counts.startswith('ddim'):
            desired = int(section_counts[len('ddim'):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError('cannot create exactly %d steps with an integer stride' % num_timesteps)
        section_counts = [int(x) for x in section_counts.split(',')]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f'cannot divide section of {size} steps into {cnt}')
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class Schedule:
    """Float64 tables of GaussianDiffusion.__init__ (gaussian_diffusion.py:336-387) and the
    SpacedDiffusion re-derivation (:1416-1431).  ``timestep_map[i]`` is the original
    timestep handed to the network at spaced step i (_WrappedModel, :1458-1463)."""

    def __init__(self, num_steps=1000, respace=None):
        base = linear_betas(num_steps)
        if respace is not None:
            use = space_timesteps(num_steps, respace)
            ac = np.cumprod(1.0 - base, axis=0)
            last, new_betas, tmap = 1.0, [], []
            for i, a in enumerate(ac):
                if i in use:
                    new_betas.append(1 - a / last)
                    last = a
                    tmap.append(i)
            betas = np.array(new_betas, dtype=np.float64)
            self.timestep_map = tmap
        else:
            betas = base
            self.timestep_map = list(range(num_steps))
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        # FIXED_LARGE (p_mean_variance :524-530)
        self.model_variance = np.append(self.posterior_variance[1], betas[1:])
        self.model_log_variance = np.log(self.model_variance)


def _f32(v):
    """_extract_into_tensor (:1330-1343): fp64 table entry -> .float() BEFORE arithmetic."""
    return torch.tensor(float(v), dtype=torch.float64).float()


def ddpm_step(sched, i, x, x0, noise):
    """p_sample (:634-696) given the network's x0 prediction (START_X, clip_denoised=False)."""
    mean = _f32(sched.posterior_mean_coef1[i]) * x0 + _f32(sched.posterior_mean_coef2[i]) * x
    nonzero = 0.0 if i == 0 else 1.0
    return mean + nonzero * torch.exp(0.5 * _f32(sched.model_log_variance[i])) * noise


def ddim_step(sched, i, x, x0, noise, eta=0.0):
    """ddim_sample (:799-852)."""
    eps = (_f32(sched.sqrt_recip_alphas_cumprod[i]) * x - x0) / _f32(sched.sqrt_recipm1_alphas_cumprod[i])
    ab, abp = _f32(sched.alphas_cumprod[i]), _f32(sched.alphas_cumprod_prev[i])
    sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
    mean_pred = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
    nonzero = 0.0 if i == 0 else 1.0
    return mean_pred + nonzero * sigma * noise


def q_sample(sched, i, x_start, noise):
    """gaussian_diffusion.py:405-421: sqrt(alpha_bar_i) x_0 + sqrt(1 - alpha_bar_i) noise, tables cast to fp32 first."""
    return _f32(np.sqrt(sched.alphas_cumprod[i])) * x_start + _f32(np.sqrt(1.0 - sched.alphas_cumprod[i])) * noise


def sample_loop(p, dims, sched, mode, x_T, xf_out, motion_mask, step_noise=None, generator=None,
                num_steps=None, trajectory=None, hoist_text=True, pre_seq=None, transl_req=None, draws=None):
    """p_sample_loop_progressive (:746-797) / ddim_sample_loop_progressive (:998-1049) with eta=0.

    ``step_noise`` (callable i -> tensor) or ``generator`` supplies the per-step
    ``randn_like(x)`` the reference draws EVERY step, DDIM included (:685, :847).
    ``num_steps`` truncates the loop (first num_steps iterations from i = S-1 down).
    ``pre_seq`` [B,Tp,C] / ``transl_req`` [[channel, v0, v1], ...] (DDPM only): p_sample :664-674 and ddim_sample
    :816-820 overwrite ``x[:, :Tp]`` with q_sample(pre_seq, t, randn_like(pre_seq)) and ``x[:, :2, channel]`` with
    q_sample(transl, t, randn(2)) before every network call; ``draws`` then is an iterator yielding the random tensors in
    the order the reference draws them (pre_seq noise, one randn(2) per item, randn_like(x)); None = torch's global RNG.
    """
    x = x_T.to(torch.float32) if x_T.dtype != torch.float64 else x_T
    text_feats = precompute_text(p, xf_out, dims) if hoist_text else None
    S = sched.num_timesteps
    indices = list(range(S))[::-1]
    if num_steps is not None:
        indices = indices[:num_steps]
    seeded = pre_seq is not None or bool(transl_req)
    if seeded:
        x = x.clone()
        draws = iter(draws) if draws is not None else None
        rnd = lambda shape: next(draws) if draws is not None else torch.randn(shape, generator=generator)
    for n, i in enumerate(indices):
        if seeded:
            if pre_seq is not None:
                x[:, :pre_seq.shape[1], :] = q_sample(sched, i, pre_seq, rnd(pre_seq.shape))
            for item in (transl_req or []) if mode == 'ddpm' else []:
                x[:, :2, int(item[0])] = q_sample(sched, i, torch.tensor(item[1:], dtype=torch.float32), rnd((2,)))
        x0 = denoise(p, dims, x, sched.timestep_map[i], xf_out, motion_mask, text_feats=text_feats)
        if seeded:
            noise = rnd(x.shape)
        elif step_noise is not None:
            noise = step_noise(i)
        else:
            noise = torch.randn(x.shape, generator=generator, dtype=x.dtype)
        x = (ddpm_step if mode == 'ddpm' else ddim_step)(sched, i, x, x0, noise)
        if trajectory is not None:
            trajectory.append((i, x.clone(), x0.clone()))
    return x


# ------------------------------------------------------------------------------------
# RePaint / outpainting mode of the DDIM loop (long-sequence windows)
# ------------------------------------------------------------------------------------
def jump_schedule(time_respacing, jump_length=1, jump_n_sample=1):
    """scheduler.py:178-208 ``get_schedule_jump_cjm_ddim``: descend from t_T-1 (t_T = 60 % of the DDIM steps, 15 for
    25 steps) to 0; at every multiple of jump_length below t_T-jump_length climb back jump_length steps
    (jump_n_sample-1 times).  Returns the visited indices followed by -1."""
    t_T = 15 if time_respacing == 25 else int(time_respacing * 0.6)
    budget = {j: jump_n_sample - 1 for j in range(0, t_T - jump_length, jump_length)}
    t, ts = t_T, []
    while t >= 1:
        t -= 1
        ts.append(t)
        if budget.get(t, 0) > 0:
            budget[t] -= 1
            for _ in range(jump_length):
                t += 1
                ts.append(t)
    ts.append(-1)
    return ts


def ddim_step_repaint(sched, i, x, x0_model, noise, keep, gt, gt_noise, overlap_len, add_blend=True, eta=0.0):
    """p_mean_variance :492-501 (x0 overwritten by gt on the kept region) + ddim_sample :799-883 (sample on the kept
    region replaced by gt re-noised to alpha_bar_prev; linear cross-fade over the first overlap_len frames once
    sqrt(1 - alpha_bar_prev) < 0.2).  ``keep`` is the bool outpainting_mask, ``gt_noise`` the second randn_like."""
    x0 = x0_model * ~keep + gt * keep
    sample = ddim_step(sched, i, x, x0, noise, eta)
    abp = _f32(sched.alphas_cumprod_prev[i])
    noise_weight = torch.sqrt(1 - abp)
    weighed_gt = torch.sqrt(abp) * gt + noise_weight * gt_noise
    if float(noise_weight) < 0.2 and add_blend:
        lw = torch.linspace(0, 1, overlap_len).view(1, -1, 1).expand(x.shape[0], -1, -1)
        weighed_gt = weighed_gt.clone()
        weighed_gt[:, :overlap_len, :] = weighed_gt[:, :overlap_len, :] * (1 - lw) + sample[:, :overlap_len, :] * lw
    return weighed_gt * keep + sample * ~keep, x0


def undo_step(sched, i, x, noise):
    """_undo :429-435: one forward-diffusion step q(x_{i+1} | x_i) with beta[i]."""
    beta = _f32(sched.betas[i])
    return torch.sqrt(1 - beta) * x + torch.sqrt(beta) * noise


def sample_loop_repaint(p, dims, sched, x_T, xf_out, motion_mask, keep, gt, draws, overlap_len, time_respacing,
                        jump_length=3, jump_n_sample=5, no_resample=False, add_blend=True, trajectory=None):
    """ddim_sample_loop :962-976 -> ddim_sample_loop_progressive_harmonize :1050-1118 (eta=0).
    ``draws`` yields the randn_like tensors in the order the reference consumes them: two per denoise step
    (DDIM noise, then the gt re-noising), one per undo step."""
    times = jump_schedule(time_respacing) if no_resample else jump_schedule(time_respacing, jump_length, jump_n_sample)
    text_feats = precompute_text(p, xf_out, dims)
    x = x_T
    draws = iter(draws)
    for t_last, t_cur in zip(times[:-1], times[1:]):
        if t_cur < t_last:
            x0m = denoise(p, dims, x, sched.timestep_map[t_last], xf_out, motion_mask, text_feats=text_feats)
            n1, n2 = next(draws), next(draws)
            x, x0 = ddim_step_repaint(sched, t_last, x, x0m, n1, keep, gt, n2, overlap_len, add_blend)
        else:
            x = undo_step(sched, t_last, x, next(draws))
        if trajectory is not None:
            trajectory.append((t_last, t_cur, x.clone()))
    return x
