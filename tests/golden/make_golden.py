#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ FROM THE REFERENCE'S OWN MODULES.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [--skip-full]

What it does
  1. imports the reference hot-path modules by path (oracle/ref_shim.py; stubs for
     mmcv / clip, and oracle/tutel_restated.py standing in for the un-vendored tutel),
  2. builds ``STMoGenTransformer`` / ``SpacedDiffusion`` / ``GaussianDiffusion`` from the
     reference, loads the deterministic (seed, key) weights of oracle/weights.py and
     checks the key set + shapes against ``param_shapes``,
  3. pins oracle/stmogen_oracle.py against the reference (asserts max-abs <= 1e-5),
  4. writes small .npz fixtures: inputs and REFERENCE outputs only (no source text).

Fixtures
  schedules.npz         a1/a2 tables for linear-1000 and respace '15,15,8,6,6'
  small_modules.npz     reduced config (L=32,NL=2,T=24,B=2, one padded sample): denoiser
                        call at t=777 with per-layer intermediates captured by hooks
  small_ddim.npz        reduced config: 50-step DDIM trajectory (every 10th x_t) + final
  small_ddpm.npz        reduced config: 1000-step schedule, first 20 DDPM steps
  preseq_small.npz      reduced config: pre_seq (5 frames) + transl_req seeding, first 12 DDPM steps and the 50-step DDIM loop
  control_small.npz     ControlT2MHalf (copy_blocks_num=2, 35-d condition of 20 frames, NL=3): x0 at t=640, 3
  repaint_small.npz     RePaint / outpainting DDIM mode (reduced config, first 6 frames kept): harmonize loop with
                        resampling (jump 3 x 5), without resampling, and no_repaint (plain 50 steps + blending)
  text_encoder.npz      encode_text(text, clip_feat): text_pre_proj + 2-layer nn.TransformerEncoder + text_ln (Dt=256)
  clip_tower_hf.npz     CLIP text transformer (reduced: width 128, 2 layers) -> features of transformers.CLIPTextModel, the
                        installed independent implementation of the un-vendored clip package's text tower
  wav_encoder.npz       WavEncoder(out_dim=64, audio_in=2), eval mode: 2 x 4000 samples -> reference output
  skeleton_parts.npz    8-part layouts: human_ml3d (263-d) and kit_ml (251-d) reduced configs (x0 at two t + 50-step
                        DDIM final), and the shipped T2M_humanml3d.py architecture (L=64, H=8) x0 at t=500
  control_wav_small.npz ControlT2MHalf S2G form: condition_pre_encode='wav' (WavEncoder on 9000 x 2 raw audio -> 17 frames)
  evaluator.npz         evaluation embedding model (ActorAgnosticEncoder + DistilbertActorAgnosticEncoder over a reduced
                        DistilBERT), the transformers WordPiece tokenizer on tricky sentences, mogen/core/evaluation
                        metric functions and the five evaluators driven by a stub embedding model
  t2m_evaluator.npz     HumanML3D/KIT evaluator (MovementConvEncoder + BiGRU motion and text heads), reduced widths
  full_denoise.npz      0.125b config, B=1, T=196: x0 prediction at t=999 and t=57
  full_ddim.npz         0.125b config, B=1: final sample of the 50-step DDIM loop
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)

from oracle import ref_shim, stmogen_oracle as O, weights as W  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

SMALL = W.default_dims(max_seq_len=24, L=32, NL=2, F=64, Te=64, Dt=32, Nt=8)
FULL = W.default_dims()
CTRL = W.default_dims(max_seq_len=24, L=32, NL=3, F=64, Te=64, Dt=32, Nt=8)   # control-branch fixture: copy_blocks_num=2
CTRL_COPY, CTRL_FEATS, CTRL_TC = 2, 35, 20
SMALL_SEED = 2   # weight seed whose routing overflows the expert capacity in both layers (drops are exercised)
DIFF_DDIM = dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                 model_var_type='fixed_large', respace='15,15,8,6,6')
DIFF_DDPM = dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                 model_var_type='fixed_large')


def synth_inputs(dims, B, T, seed, lengths=None):
    """Same generator recipe as tests/ and bench.py."""
    g = torch.Generator().manual_seed(seed)
    x_T = torch.randn(B, T, dims['input_feats'], generator=g)
    xf = torch.nn.functional.layer_norm(torch.randn(B, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],))
    mask = torch.ones(B, T)
    if lengths is not None:
        for b, n in enumerate(lengths):
            mask[b, n:] = 0
    return x_T, xf, mask


def build_ref(dims, seed=0):
    m = ref_shim.build_reference_denoiser(W.reference_model_cfg(dims))
    sd_ref = m.state_dict()
    shapes = W.param_shapes(dims)
    assert set(sd_ref.keys()) == set(shapes.keys()), (
        sorted(set(sd_ref) ^ set(shapes))[:10])
    for k, v in sd_ref.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    sd = W.make_state_dict(dims, seed)
    m.load_state_dict(sd)
    return m, sd


def model_kwargs(xf, mask):
    B, T = mask.shape
    return dict(xf_out=xf, motion_mask=mask, motion_length=mask.sum(1, keepdim=True).long(),
                num_intervals=1, c=None, y={}, patch_size=1, sample_idx=None)


def maxabs(a, b):
    return float((a.double() - b.double()).abs().max())


def schedules():
    gd = ref_shim.load().gaussian_diffusion
    out = {}
    for tag, cfg in (('ddim50', DIFF_DDIM), ('ddpm1000', DIFF_DDPM)):
        d = ref_shim.build_reference_diffusion(cfg)
        s = O.Schedule(1000, cfg.get('respace'))
        tmap = getattr(d, 'timestep_map', list(range(d.num_timesteps)))
        assert list(tmap) == list(s.timestep_map)
        fixed_large = np.append(d.posterior_variance[1], d.betas[1:])
        for name, ref_arr in (('betas', d.betas), ('alphas_cumprod', d.alphas_cumprod),
                              ('alphas_cumprod_prev', d.alphas_cumprod_prev),
                              ('sqrt_recip_alphas_cumprod', d.sqrt_recip_alphas_cumprod),
                              ('sqrt_recipm1_alphas_cumprod', d.sqrt_recipm1_alphas_cumprod),
                              ('posterior_mean_coef1', d.posterior_mean_coef1),
                              ('posterior_mean_coef2', d.posterior_mean_coef2),
                              ('model_log_variance', np.log(fixed_large))):
            assert np.array_equal(ref_arr, getattr(s, name)), (tag, name)
            out[f'{tag}.{name}'] = ref_arr
        out[f'{tag}.timestep_map'] = np.array(tmap, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, 'schedules.npz'), **out)
    print('schedules.npz: oracle tables bit-equal to reference')


def small_modules():
    dims, B, T = SMALL, 2, 24
    m, sd = build_ref(dims, SMALL_SEED)
    x_T, xf, mask = synth_inputs(dims, B, T, seed=11, lengths=[24, 17])
    t = 777
    cap = {}
    hooks = []
    for i, blk in enumerate(m.temporal_decoder_blocks):
        hooks.append(blk.ca_block.motion_moe.register_forward_hook(
            lambda mod, a, o, i=i: cap.__setitem__(f'layer{i}.motion_feat', o.detach().clone())))
        hooks.append(blk.ca_block.text_moe.register_forward_hook(
            lambda mod, a, o, i=i: cap.__setitem__(f'layer{i}.text_feat', o.detach().clone())))
        hooks.append(blk.ca_block.body_d_attn.register_forward_hook(
            lambda mod, a, o, i=i: cap.__setitem__(f'layer{i}.dyn', o.detach().clone())))
        hooks.append(blk.ca_block.register_forward_hook(
            lambda mod, a, o, i=i: cap.__setitem__(f'layer{i}.after_stma', o.detach().clone())))
        hooks.append(blk.ffn.register_forward_hook(
            lambda mod, a, o, i=i: cap.__setitem__(f'layer{i}.after_ffn', o.detach().clone())))
    hooks.append(m.joint_embed.register_forward_hook(
        lambda mod, a, o: cap.__setitem__('pose_enc', o.detach().clone())))
    hooks.append(m.time_embed.register_forward_hook(
        lambda mod, a, o: cap.__setitem__('emb', o.detach().clone())))
    hooks.append(m.out.register_forward_hook(
        lambda mod, a, o: cap.__setitem__('out2', o.detach().clone())))
    with torch.no_grad():
        x0_ref = m(x_T, torch.full((B,), t, dtype=torch.long), **model_kwargs(xf, mask))
    for h in hooks:
        h.remove()
    ocap = {}
    x0_or = O.denoise(sd, dims, x_T, t, xf, mask, cap=ocap)
    errs = {'x0': maxabs(x0_ref, x0_or), 'out2': maxabs(cap['out2'], ocap['out2']),
            'emb': maxabs(cap['emb'], ocap['emb'])}
    for i in range(dims['NL']):
        for k in ('motion_feat', 'text_feat', 'after_stma', 'after_ffn'):
            errs[f'layer{i}.{k}'] = maxabs(cap[f'layer{i}.{k}'], ocap[f'layer{i}'][k])
    print('small_modules: oracle vs reference max-abs:', {k: f'{v:.2e}' for k, v in errs.items()})
    assert max(errs.values()) <= 1e-5, errs
    # routing statistics (capacity overflow must be exercised)
    from oracle import tutel_restated as TR
    drops = {}
    for i in range(dims['NL']):
        pre = f'temporal_decoder_blocks.{i}.ca_block.'
        _, r = O.moe_wrapper(sd, pre + 'motion_moe.', torch.nn.functional.layer_norm(
            (ocap['h0'].repeat(2, 1, 1) if i == 0 else ocap[f'layer{i-1}']['after_ffn']).reshape(2 * B, T, 12, -1),
            (dims['L'],), sd[pre + 'norm.weight'], sd[pre + 'norm.bias']), return_routing=True)
        drops[i] = [int((~k).sum()) for k in r['keeps']]
    print('small_modules: dropped (choice0, choice1) per layer:', drops)
    save = dict(x_t=x_T, xf_out=xf, motion_mask=mask, t=np.int64(t), x0=x0_ref, out2=cap['out2'],
                emb=cap['emb'], pose_enc=cap['pose_enc'])
    for i in range(dims['NL']):
        for k in ('motion_feat', 'text_feat', 'dyn', 'after_stma', 'after_ffn'):
            save[f'layer{i}.{k}'] = cap[f'layer{i}.{k}']
        save[f'layer{i}.dropped'] = np.array(drops[i])
    np.savez_compressed(os.path.join(OUT, 'small_modules.npz'),
                        **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in save.items()})


def run_ref_loop(m, diff, mode, x_T, xf, mask, seed, num_steps=None):
    traj = []
    torch.manual_seed(seed)
    B, T, C = x_T.shape
    gen = (diff.p_sample_loop_progressive if mode == 'ddpm' else diff.ddim_sample_loop_progressive)
    kw = dict(noise=x_T.clone(), clip_denoised=False, model_kwargs=model_kwargs(xf, mask))
    if mode == 'ddim':
        kw['eta'] = 0
    with torch.no_grad():
        for n, s in enumerate(gen(m, (B, T, C), **kw)):
            traj.append(s['sample'].clone())
            if num_steps is not None and n + 1 >= num_steps:
                break
    return traj


def run_oracle_loop(sd, dims, sched, mode, x_T, xf, mask, seed, num_steps=None):
    traj = []
    torch.manual_seed(seed)
    O.sample_loop(sd, dims, sched, mode, x_T, xf, mask, num_steps=num_steps, trajectory=traj)
    return [t[1] for t in traj]


def small_loops():
    dims, B, T = SMALL, 2, 24
    m, sd = build_ref(dims, SMALL_SEED)
    x_T, xf, mask = synth_inputs(dims, B, T, seed=12, lengths=[20, 24])
    # DDIM 50
    diff = ref_shim.build_reference_diffusion(DIFF_DDIM)
    tr = run_ref_loop(m, diff, 'ddim', x_T, xf, mask, seed=5)
    to = run_oracle_loop(sd, dims, O.Schedule(1000, DIFF_DDIM['respace']), 'ddim', x_T, xf, mask, seed=5)
    e = [maxabs(a, b) for a, b in zip(tr, to)]
    print(f'small_ddim: oracle vs reference over 50 steps: max {max(e):.2e} final {e[-1]:.2e}')
    assert max(e) <= 1e-5
    np.savez_compressed(os.path.join(OUT, 'small_ddim.npz'), x_T=x_T.numpy(), xf_out=xf.numpy(),
                        motion_mask=mask.numpy(), noise_seed=np.int64(5),
                        traj=np.stack([t.numpy() for t in tr[9::10]]), final=tr[-1].numpy())
    # DDPM, first 20 of 1000
    diff = ref_shim.build_reference_diffusion(DIFF_DDPM)
    tr = run_ref_loop(m, diff, 'ddpm', x_T, xf, mask, seed=6, num_steps=20)
    to = run_oracle_loop(sd, dims, O.Schedule(1000, None), 'ddpm', x_T, xf, mask, seed=6, num_steps=20)
    e = [maxabs(a, b) for a, b in zip(tr, to)]
    print(f'small_ddpm: oracle vs reference over 20 steps: max {max(e):.2e}')
    assert max(e) <= 1e-5
    np.savez_compressed(os.path.join(OUT, 'small_ddpm.npz'), x_T=x_T.numpy(), xf_out=xf.numpy(),
                        motion_mask=mask.numpy(), noise_seed=np.int64(6),
                        traj=np.stack([t.numpy() for t in tr[4::5]]))


def preseq():
    """pre_seq / transl_req seeding of p_sample (:664-674) and ddim_sample (:816-820): the reference's own loops with
    ``pre_seq`` [2, 5, 322] (and two transl_req items for DDPM; B = 2 is the largest batch the reference's
    ``_extract_into_tensor(arr, t, (2,))`` accepts there), torch's global RNG seeded like the other loop fixtures."""
    dims, B, T = SMALL, 2, 24
    m, sd = build_ref(dims, SMALL_SEED)
    x_T, xf, mask = synth_inputs(dims, B, T, seed=14, lengths=[24, 21])
    g = torch.Generator().manual_seed(15)
    pre = torch.randn(B, 5, dims['input_feats'], generator=g)
    transl = [[309, 0.25, -0.5], [311, 1.0, 0.75]]

    def ref_loop(diff, mode, seed, num_steps=None):
        torch.manual_seed(seed)
        gen = diff.p_sample_loop_progressive if mode == 'ddpm' else diff.ddim_sample_loop_progressive
        kw = dict(noise=x_T.clone(), clip_denoised=False, model_kwargs=model_kwargs(xf, mask), pre_seq=pre.clone())
        if mode == 'ddim':
            kw['eta'] = 0
        else:
            kw['transl_req'] = transl
        traj = []
        with torch.no_grad():
            for n, sres in enumerate(gen(m, (B, T, dims['input_feats']), **kw)):
                traj.append(sres['sample'].clone())
                if num_steps is not None and n + 1 >= num_steps:
                    break
        return traj

    def oracle_loop(sched, mode, seed, num_steps=None):
        torch.manual_seed(seed)
        traj = []
        O.sample_loop(sd, dims, sched, mode, x_T, xf, mask, num_steps=num_steps, trajectory=traj, pre_seq=pre,
                      transl_req=transl if mode == 'ddpm' else None)
        return [t[1] for t in traj]

    diff = ref_shim.build_reference_diffusion(DIFF_DDPM)
    tr = ref_loop(diff, 'ddpm', 16, 12)
    to = oracle_loop(O.Schedule(1000, None), 'ddpm', 16, 12)
    e = [maxabs(a, b) for a, b in zip(tr, to)]
    print(f'preseq ddpm: oracle vs reference over 12 steps: max {max(e):.2e}')
    assert max(e) <= 1e-5
    diff = ref_shim.build_reference_diffusion(DIFF_DDIM)
    tr2 = ref_loop(diff, 'ddim', 17)
    to2 = oracle_loop(O.Schedule(1000, DIFF_DDIM['respace']), 'ddim', 17)
    e = [maxabs(a, b) for a, b in zip(tr2, to2)]
    print(f'preseq ddim: oracle vs reference over 50 steps: max {max(e):.2e} final {e[-1]:.2e}')
    assert max(e) <= 1e-5
    np.savez_compressed(os.path.join(OUT, 'preseq_small.npz'), x_T=x_T.numpy(), xf_out=xf.numpy(), motion_mask=mask.numpy(),
                        pre_seq=pre.numpy(), transl_req=np.array(transl, dtype=np.float64), ddpm_seed=np.int64(16),
                        ddim_seed=np.int64(17), ddpm_traj=np.stack([t.numpy() for t in tr[3::4]]),
                        ddim_traj=np.stack([t.numpy() for t in tr2[9::10]]), ddim_final=tr2[-1].numpy())


def control():
    """Plug-and-play control branch (a15): reference ControlT2MHalf, condition_pre_encode=False (M2D form:
    raw 35-d music features), condition shorter than the motion (zero padding), condition_cfg=True."""
    dims, B, T = CTRL, 2, 24
    m = ref_shim.build_reference_control(W.reference_model_cfg(dims), CTRL_COPY, CTRL_FEATS)
    shapes = W.control_param_shapes(dims, CTRL_COPY, CTRL_FEATS)
    assert set(m.state_dict().keys()) == set(shapes.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    sd = W.make_state_dict(dims, SMALL_SEED, shapes=shapes)
    torch.nn.Module.load_state_dict(m, sd)        # (the wrapper's own load_state_dict override re-keys base-only checkpoints)
    x_T, xf, mask = synth_inputs(dims, B, T, seed=13, lengths=[24, 18])
    g = torch.Generator().manual_seed(14)
    c = torch.randn(B, CTRL_TC, CTRL_FEATS, generator=g)
    save = dict(x_t=x_T.numpy(), xf_out=xf.numpy(), motion_mask=mask.numpy(), c=c.numpy())
    for t in (640, 3):
        with torch.no_grad():
            ref = m(x_T, torch.full((B,), t), motion_mask=mask, motion_length=mask.sum(1, keepdim=True).long(),
                    num_intervals=1, c=c, xf_out=xf, y={}, patch_size=1, sample_idx=None)
            ref_noc = m(x_T, torch.full((B,), t), motion_mask=mask, motion_length=mask.sum(1, keepdim=True).long(),
                        num_intervals=1, c=None, xf_out=xf, y={}, patch_size=1, sample_idx=None)
        out = O.denoise_control(sd, dims, x_T, t, xf, mask, c, CTRL_COPY)
        pb = {k[len('base_model.'):]: v for k, v in sd.items() if k.startswith('base_model.')}
        out_noc = O.denoise(pb, dims, x_T, t, xf, mask)
        print(f'control t={t}: oracle vs reference {maxabs(ref, out):.2e} (c=None: {maxabs(ref_noc, out_noc):.2e}); '
              f'|with c - without c| max {maxabs(ref, ref_noc):.2f}')
        assert maxabs(ref, out) <= 1e-5 and maxabs(ref_noc, out_noc) <= 1e-5
        save[f'x0_t{t}'] = ref.numpy()
        save[f'x0_noc_t{t}'] = ref_noc.numpy()
    np.savez_compressed(os.path.join(OUT, 'control_small.npz'), **save)


TEXT_CFG = dict(pretrained_model='clip', latent_dim=256, num_layers=2, ff_size=2048, dropout=0, use_text_proj=False)


def text_encoder():
    """the reference's own encode_text (diffusion_transformer.py:142-172) with clip_feat given."""
    from oracle import text_encoder_oracle as TO
    dims = W.default_dims(max_seq_len=24, L=32, NL=1, F=64, Te=64, Dt=256, Nt=77)
    m = ref_shim.build_reference_text_encoder(W.reference_model_cfg(dims), TEXT_CFG)
    shapes = W.text_encoder_param_shapes(256, 2, 2048)
    ref_sd = {k: v for k, v in m.state_dict().items() if k.split('.')[0] in ('text_pre_proj', 'textTransEncoder', 'text_ln')}
    assert set(ref_sd) == set(shapes), sorted(set(ref_sd) ^ set(shapes))[:8]
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    sd = W.make_text_encoder_state(shapes, seed=4)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(not k.startswith(('text_', 'textTrans')) for k in missing)
    g = torch.Generator().manual_seed(61)
    feat = torch.randn(3, 77, 512, generator=g)
    with torch.no_grad():
        ref = m.encode_text(['a', 'b', 'c'], feat, 'cpu')
    orc = TO.finetune_encoder(sd, feat, 2)
    print(f'text_encoder: xf_out {tuple(ref.shape)}; oracle vs reference {maxabs(ref, orc):.2e}')
    assert maxabs(ref, orc) <= 1e-5        # torch's fused encoder-layer fast path orders the fp32 sums differently
    np.savez_compressed(os.path.join(OUT, 'text_encoder.npz'), clip_feat=feat.numpy(), xf_out=ref.numpy(), seed=np.int64(4))


WAV_DIM, WAV_IN, WAV_SAMPLES = 64, 2, 4000


def clip_tower():
    """The CLIP text transformer (stage B of encode_text, diffusion_transformer.py:144-151) lives in the un-vendored `clip`
    package.  Its published architecture is pinned here against a SECOND independent implementation that is installed:
    transformers.CLIPTextModel (hidden_act='quick_gelu', causal mask, pre-LN blocks, final_layer_norm) -- the class the
    OpenAI checkpoints are served through on the HF hub.  The deterministic (seed, key) weights of oracle/weights.py are
    loaded into it under the OpenAI -> HF key mapping; its last_hidden_state = ln_final(transformer(tok + pos))."""
    from transformers import CLIPTextConfig, CLIPTextModel
    from oracle import text_encoder_oracle as TO
    width, layers, heads, ff, vocab, seed = 128, 2, 2, 256, 300, 4
    shapes = W.text_encoder_param_shapes(256, 2, 2048, clip_width=width, clip_layers=layers, clip_ff=ff, vocab=vocab)
    sd = W.make_text_encoder_state(shapes, seed=seed)
    cfg = CLIPTextConfig(vocab_size=vocab, hidden_size=width, intermediate_size=ff, num_hidden_layers=layers,
                         num_attention_heads=heads, max_position_embeddings=77, hidden_act='quick_gelu',
                         eos_token_id=vocab - 1, bos_token_id=vocab - 2, pad_token_id=0)
    m = CLIPTextModel(cfg).eval()
    hf = {'embeddings.token_embedding.weight': sd['clip.token_embedding.weight'],
          'embeddings.position_embedding.weight': sd['clip.positional_embedding'],
          'final_layer_norm.weight': sd['clip.ln_final.weight'], 'final_layer_norm.bias': sd['clip.ln_final.bias']}
    for i in range(layers):
        a, b = f'clip.transformer.resblocks.{i}.', f'encoder.layers.{i}.'
        wq, bq = sd[a + 'attn.in_proj_weight'], sd[a + 'attn.in_proj_bias']
        for j, n in enumerate(('q_proj', 'k_proj', 'v_proj')):
            hf[b + f'self_attn.{n}.weight'], hf[b + f'self_attn.{n}.bias'] = wq[j * width:(j + 1) * width], bq[j * width:(j + 1) * width]
        hf[b + 'self_attn.out_proj.weight'], hf[b + 'self_attn.out_proj.bias'] = sd[a + 'attn.out_proj.weight'], sd[a + 'attn.out_proj.bias']
        hf[b + 'layer_norm1.weight'], hf[b + 'layer_norm1.bias'] = sd[a + 'ln_1.weight'], sd[a + 'ln_1.bias']
        hf[b + 'layer_norm2.weight'], hf[b + 'layer_norm2.bias'] = sd[a + 'ln_2.weight'], sd[a + 'ln_2.bias']
        hf[b + 'mlp.fc1.weight'], hf[b + 'mlp.fc1.bias'] = sd[a + 'mlp.c_fc.weight'], sd[a + 'mlp.c_fc.bias']
        hf[b + 'mlp.fc2.weight'], hf[b + 'mlp.fc2.bias'] = sd[a + 'mlp.c_proj.weight'], sd[a + 'mlp.c_proj.bias']
    inner = getattr(m, 'text_model', m)             # (transformers >= 5 keeps the layers on the model itself)
    missing, unexpected = inner.load_state_dict(hf, strict=False)
    assert not unexpected and all('position_ids' in k for k in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(5)
    tokens = torch.randint(1, vocab - 2, (3, 77), generator=g)
    tokens[:, 0] = vocab - 2
    for b, n in enumerate((9, 30, 76)):                 # end token, then zero padding like clip.tokenize
        tokens[b, n] = vocab - 1
        tokens[b, n + 1:] = 0
    with torch.no_grad():
        ref = m(input_ids=tokens).last_hidden_state
    ours = TO.clip_text_features(sd, tokens, layers, heads=heads)
    e = maxabs(ours, ref)
    print(f'clip_tower: oracle restatement vs transformers.CLIPTextModel: max {e:.2e}')
    assert e <= 1e-5
    np.savez_compressed(os.path.join(OUT, 'clip_tower_hf.npz'), tokens=tokens.numpy(), feat=ref.numpy(), seed=np.int64(seed),
                        width=np.int64(width), layers=np.int64(layers), heads=np.int64(heads), ff=np.int64(ff), vocab=np.int64(vocab))


def wav_encoder():
    """the reference class itself (mogen/models/utils/blocks.py:56-71), eval mode."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_blocks', os.path.join(ref_shim.REF, 'mogen/models/utils/blocks.py'))
    blocks = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(blocks)
    from oracle import wav_encoder_oracle as WO
    from motioncraft_amd.wav_encoder import wav_encoder_param_shapes
    m = blocks.WavEncoder(out_dim=WAV_DIM, audio_in=WAV_IN)
    shapes = wav_encoder_param_shapes(WAV_DIM, WAV_IN)
    ref_sd = m.feat_extractor.state_dict()
    assert set(ref_sd) == set(shapes), sorted(set(ref_sd) ^ set(shapes))[:8]
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    sd = W.make_wav_encoder_state(WAV_DIM, WAV_IN, seed=3)
    m.feat_extractor.load_state_dict(sd)
    m.eval()
    g = torch.Generator().manual_seed(51)
    wav = torch.randn(2, WAV_SAMPLES, WAV_IN, generator=g)
    with torch.no_grad():
        ref = m(wav)
    orc = WO.wav_encoder(sd, wav)
    print(f'wav_encoder: out {tuple(ref.shape)}, |out| mean {float(ref.abs().mean()):.3f}; oracle vs reference {maxabs(ref, orc):.2e}')
    assert maxabs(ref, orc) <= 1e-6
    np.savez_compressed(os.path.join(OUT, 'wav_encoder.npz'), wav=wav.numpy(), out=ref.numpy(), seed=np.int64(3))


REPAINT_OVERLAP = 6


def repaint():
    """gaussian_diffusion.py:492-501, 855-883, 962-976, 1050-1118; scheduler.py:178-208."""
    import types
    dims, B, T = SMALL, 2, 24
    m, sd = build_ref(dims, SMALL_SEED)
    x_T, xf, mask = synth_inputs(dims, B, T, seed=41, lengths=[24, 21])
    g = torch.Generator().manual_seed(42)
    gt = torch.zeros(B, T, dims['input_feats'])
    gt[:, :REPAINT_OVERLAP] = torch.randn(B, REPAINT_OVERLAP, dims['input_feats'], generator=g)
    keep = torch.zeros(B, T, dims['input_feats'], dtype=torch.bool)
    keep[:, :REPAINT_OVERLAP] = True
    sched = O.Schedule(1000, DIFF_DDIM['respace'])
    save = dict(x_T=x_T.numpy(), xf_out=xf.numpy(), motion_mask=mask.numpy(), gt=gt.numpy(), keep=keep.numpy(),
                overlap_len=np.int64(REPAINT_OVERLAP), noise_seed=np.int64(9))
    for tag, over in (('resample', {}), ('noresample', dict(no_resample=True)), ('norepaint', dict(no_repaint=True))):
        opt = types.SimpleNamespace(same_overlap_noisy=False, no_repaint=False, addBlend=True, overlap_len=REPAINT_OVERLAP,
                                    no_resample=False, jump_length=3, jump_n_sample=5, timestep_respacing='ddim50')
        opt.__dict__.update(over)
        diff = ref_shim.build_reference_diffusion(DIFF_DDIM, opt)
        kw = model_kwargs(xf, mask)
        kw['y'] = dict(gt=gt.clone(), outpainting_mask=keep.clone())
        torch.manual_seed(9)
        with torch.no_grad():
            ref = diff.ddim_sample_loop(m, (B, T, dims['input_feats']), noise=x_T.clone(), clip_denoised=False,
                                        model_kwargs=kw, eta=0)
        gen = torch.Generator().manual_seed(9)
        draws = (torch.randn(B, T, dims['input_feats'], generator=gen) for _ in range(10 ** 6))
        if opt.no_repaint:
            x = x_T
            tf = O.precompute_text(sd, xf, dims)
            for i in range(49, -1, -1):
                x0m = O.denoise(sd, dims, x, sched.timestep_map[i], xf, mask, text_feats=tf)
                x, _ = O.ddim_step_repaint(sched, i, x, x0m, next(draws), keep, gt, next(draws), REPAINT_OVERLAP)
            orc = x
        else:
            orc = O.sample_loop_repaint(sd, dims, sched, x_T, xf, mask, keep, gt, draws, REPAINT_OVERLAP, 50,
                                        no_resample=opt.no_resample)
        print(f'repaint {tag}: oracle vs reference {maxabs(ref, orc):.2e}; |kept region - gt| {maxabs(ref[:, :1], gt[:, :1]):.2e}')
        assert maxabs(ref, orc) <= 1e-5
        save[f'final_{tag}'] = ref.numpy()
    np.savez_compressed(os.path.join(OUT, 'repaint_small.npz'), **save)


HML_SMALL = W.humanml3d_dims(max_seq_len=24, L=32, NL=2, F=64, Te=64, Dt=32, Nt=8)
KIT_SMALL = W.humanml3d_dims(max_seq_len=24, L=32, NL=2, F=64, Te=64, Dt=32, Nt=8, input_feats=251, dataset='kit_ml')
HML_FULL = W.humanml3d_dims()


def skeleton_parts():
    """the 8-part PoseEncoder/PoseDecoder layouts (reference stmogen.py:188-209,354-365,545-562)."""
    save = {}
    for tag, dims in (('hml', HML_SMALL), ('kit', KIT_SMALL)):
        B, T = 2, 24
        m, sd = build_ref(dims, SMALL_SEED)
        x_T, xf, mask = synth_inputs(dims, B, T, seed=31, lengths=[24, 17])
        save[f'{tag}_x_t'], save[f'{tag}_xf_out'], save[f'{tag}_motion_mask'] = x_T.numpy(), xf.numpy(), mask.numpy()
        for t in (901, 12):
            with torch.no_grad():
                r = m(x_T, torch.full((B,), t, dtype=torch.long), **model_kwargs(xf, mask))
            o = O.denoise(sd, dims, x_T, t, xf, mask)
            print(f'skeleton_parts {tag} t={t}: oracle vs reference {maxabs(r, o):.2e}')
            assert maxabs(r, o) <= 1e-5
            save[f'{tag}_x0_t{t}'] = r.numpy()
        diff = ref_shim.build_reference_diffusion(DIFF_DDIM)
        tr = run_ref_loop(m, diff, 'ddim', x_T, xf, mask, seed=8)
        to = run_oracle_loop(sd, dims, O.Schedule(1000, DIFF_DDIM['respace']), 'ddim', x_T, xf, mask, seed=8)
        e = [maxabs(a, b) for a, b in zip(tr, to)]
        print(f'skeleton_parts {tag} ddim: oracle vs reference max {max(e):.2e}')
        assert max(e) <= 1e-5
        save[f'{tag}_ddim_final'] = tr[-1].numpy()
    dims, B, T = HML_FULL, 1, 196
    m, sd = build_ref(dims)
    x_T, xf, mask = synth_inputs(dims, B, T, seed=32, lengths=[163])
    with torch.no_grad():
        r = m(x_T, torch.full((B,), 500, dtype=torch.long), **model_kwargs(xf, mask))
    o = O.denoise(sd, dims, x_T, 500, xf, mask)
    print(f'skeleton_parts T2M_humanml3d arch t=500 len=163: oracle vs reference {maxabs(r, o):.2e}')
    assert maxabs(r, o) <= 1e-5
    save['hmlfull_x0_t500_len163'] = r.numpy()
    np.savez_compressed(os.path.join(OUT, 'skeleton_parts.npz'), **save)


CTRL_WAV_IN, CTRL_WAV_SAMPLES = 2, 9000


def control_wav():
    """S2G form of the control branch (configs/stmogen/S2G_*: condition_pre_encode=True, type 'wav', beats2)."""
    from oracle import wav_encoder_oracle as WO
    dims, B, T = CTRL, 2, 24
    m = ref_shim.build_reference_control(W.reference_model_cfg(dims), CTRL_COPY, CTRL_WAV_IN, wav_pre_encode=True)
    shapes = W.control_wav_param_shapes(dims, CTRL_COPY, CTRL_WAV_IN)
    assert set(m.state_dict().keys()) == set(shapes.keys()), sorted(set(m.state_dict()) ^ set(shapes))[:8]
    sd = W.make_control_wav_state(dims, CTRL_COPY, CTRL_WAV_IN, SMALL_SEED)
    torch.nn.Module.load_state_dict(m, sd)
    m.eval()
    x_T, xf, mask = synth_inputs(dims, B, T, seed=15, lengths=[24, 19])
    g = torch.Generator().manual_seed(16)
    audio = torch.randn(B, CTRL_WAV_SAMPLES, CTRL_WAV_IN, generator=g)
    with torch.no_grad():
        ref = m(x_T, torch.full((B,), 420), motion_mask=mask, motion_length=mask.sum(1, keepdim=True).long(),
                num_intervals=1, c=audio, xf_out=xf, y={}, patch_size=1, sample_idx=None)
    pre = 'condition_pre_encoder.pre_encoder.feat_extractor.'
    c_enc = WO.wav_encoder({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, audio)
    out = O.denoise_control({k: v for k, v in sd.items() if not k.startswith(pre)}, dims, x_T, 420, xf, mask, c_enc, CTRL_COPY)
    print(f'control_wav: WavEncoder frames {c_enc.shape[1]}; oracle vs reference {maxabs(ref, out):.2e}')
    assert maxabs(ref, out) <= 1e-5
    np.savez_compressed(os.path.join(OUT, 'control_wav_small.npz'), x_t=x_T.numpy(), xf_out=xf.numpy(),
                        motion_mask=mask.numpy(), audio=audio.numpy(), x0_t420=ref.numpy())


sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import EVAL_BERT, EVAL_DIMS  # noqa: E402
EVAL_WORDS = ['a', 'person', 'walk', '##s', '##ing', 'forward', 'and', 'then', 'jump', 'run', '##ning', 'the', 'left', 'right',
              'hand', 'wave', '.', ',', '!', '-', "'", 'cafe', 'un', '##believ', '##able', '##ly', 'slow', '2', '##0', 'step',
              'turn', '##ed', 'man', 'wo', '##man', 'is', 'arm', 'both', 'up', 'down']
EVAL_TEXTS = ['A person walks forward.', 'jumping and then waves the left hand, running!',
              "The man's Caf\u00e9 unbelievably slow-walking 20 steps", 'xyzzy turned  \t right\n and then up', '']


def eval_vocab():
    base = ['[PAD]', '[UNK]', '[CLS]', '[SEP]', '[MASK]'] + list('bcdefghijklmnopqrstuvwxyz') + ['##' + c for c in 'abcdefghijklmnopqrstuvwxyz']
    return list(dict.fromkeys(base + EVAL_WORDS))


def evaluator():
    """T2MContrastiveModel_SMPLX's two encoders, the tokenizer, the metrics, the evaluators."""
    import tempfile
    from transformers import DistilBertConfig, DistilBertModel, DistilBertTokenizer
    from oracle import eval_encoder_oracle as EO
    from motioncraft_amd.wordpiece import WordPieceTokenizer
    ev = ref_shim.load_evaluation()
    vocab = eval_vocab()
    d = tempfile.mkdtemp()
    with open(os.path.join(d, 'vocab.txt'), 'w') as f:
        f.write('\n'.join(vocab) + '\n')
    DistilBertModel(DistilBertConfig(vocab_size=len(vocab), **EVAL_BERT)).save_pretrained(d)
    DistilBertTokenizer(os.path.join(d, 'vocab.txt')).save_pretrained(d)
    kw = {k: v for k, v in EVAL_DIMS.items() if k != 'nfeats'}
    menc = ev.rnns.ActorAgnosticEncoder(nfeats=EVAL_DIMS['nfeats'], vae=True, **kw).eval()
    tenc = ev.rnns.DistilbertActorAgnosticEncoder(modelpath=d, vae=True, **kw).eval()
    shapes = W.eval_encoder_param_shapes(bert=dict(EVAL_BERT, vocab_size=len(vocab)), **EVAL_DIMS)
    ref_keys = {'motionencoder.' + k: v for k, v in menc.state_dict().items()}
    ref_keys.update({'textencoder.' + k: v for k, v in tenc.state_dict().items()})
    assert set(ref_keys) == set(shapes), sorted(set(ref_keys) ^ set(shapes))[:8]
    for k, v in ref_keys.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    sd = W.make_eval_encoder_state(shapes, seed=6)
    assert maxabs(sd['motionencoder.sequence_pos_encoding.pe'], menc.state_dict()['sequence_pos_encoding.pe']) == 0
    menc.load_state_dict({k[len('motionencoder.'):]: v for k, v in sd.items() if k.startswith('motionencoder.')}, strict=True)
    tenc.load_state_dict({k[len('textencoder.'):]: v for k, v in sd.items() if k.startswith('textencoder.')}, strict=True)

    g = torch.Generator().manual_seed(71)
    motion = torch.randn(3, 24, EVAL_DIMS['nfeats'], generator=g)
    lengths = [24, 17, 9]
    with torch.no_grad():
        ref_m = menc(motion, torch.tensor(lengths), None).loc
        ref_t = tenc(EVAL_TEXTS, None, 'cpu').loc
    tok = tenc.tokenizer(EVAL_TEXTS, return_tensors='pt', padding=True)
    ids, mask = tok['input_ids'], tok['attention_mask']
    my_ids, my_mask = WordPieceTokenizer(d)(EVAL_TEXTS)
    assert np.array_equal(my_ids, ids.numpy()) and np.array_equal(my_mask, mask.numpy()), (my_ids, ids)
    om = EO.encode_motion(sd, motion, lengths, EVAL_DIMS['num_layers'], EVAL_DIMS['num_heads'])
    ot = EO.encode_text_tokens(sd, ids, mask, EVAL_BERT['n_layers'], EVAL_BERT['n_heads'], EVAL_DIMS['num_layers'], EVAL_DIMS['num_heads'])
    print(f'evaluator: motion mu {tuple(ref_m.shape)} oracle vs reference {maxabs(ref_m, om):.2e}; text mu oracle vs reference '
          f'{maxabs(ref_t, ot):.2e}; tokenizer ids identical ({tuple(ids.shape)})')
    assert maxabs(ref_m, om) <= 1e-5 and maxabs(ref_t, ot) <= 1e-5
    save = dict(motion=motion.numpy(), lengths=np.array(lengths), motion_mu=ref_m.numpy(), input_ids=ids.numpy().astype(np.int32),
                attention_mask=mask.numpy().astype(np.uint8), text_mu=ref_t.numpy(), seed=np.int64(6), vocab=np.array(vocab),
                texts=np.array(EVAL_TEXTS))

    # metric functions on seeded embeddings (float32 like .cpu().numpy() of the model output)
    rs = np.random.RandomState(3)
    a, b = rs.randn(40, 16).astype(np.float32), rs.randn(40, 16).astype(np.float32)
    dist = ev.utils.euclidean_distance_matrix(a, b)
    topk = ev.utils.calculate_top_k(np.argsort(dist, axis=1), 3)
    mu1, c1 = ev.utils.calculate_activation_statistics(a, 1.0)
    mu2, c2 = ev.utils.calculate_activation_statistics(b + 0.3, 1.0)
    fd = ev.utils.calculate_frechet_distance(mu1, c1, mu2, c2)
    np.random.seed(11)
    div = ev.utils.calculate_diversity(a, 20, 1.0, 1.0)
    mm = ev.utils.calculate_multimodality(a.reshape(5, 8, 16), 4)
    assert np.allclose(EO.pairwise_l2(a, b), dist) and abs(EO.frechet(mu1, c1, mu2, c2) - fd) < 1e-9
    np.random.seed(11)
    assert abs(EO.diversity(a, 20) - div) < 1e-12 and abs(EO.multimodality(a.reshape(5, 8, 16), 4) - mm) < 1e-12
    save.update(met_a=a, met_b=b, met_dist=dist, met_topk=topk, met_fid=np.float64(fd), met_div=np.float64(div), met_mm=np.float64(mm))

    # the five evaluators, replication_times=2, driven by the stub model
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import StubEvalModel, stub_eval_results
    stub, N, REP = StubEvalModel(), 48, 2
    common = dict(replication_times=REP, replication_reduction='statistics', evaluator_model=stub)
    np.random.seed(23)
    mme = ev.multimodality.MultiModalityEvaluator(data_len=N, num_samples=4, num_repeats=5, num_picks=3, **common)
    results = stub_eval_results(N, REP, mme.append_indexes)
    evs = [ev.precision.PrecisionEvaluator(data_len=N, top_k=3, batch_size=16, **common),
           ev.matching_score.MatchingScoreEvaluator(data_len=N, batch_size=16, **common),
           ev.fid.FIDEvaluator(data_len=N, emb_scale=1.0, **common),
           ev.diversity.DiversityEvaluator(data_len=N, num_samples=20, **common), mme]
    np.random.seed(29)
    metrics = {}
    for e in evs:
        metrics.update(e.evaluate(results))
    print('evaluator: reference metrics on the stub model', {k: round(float(v), 4) for k, v in metrics.items()})
    save['ev_names'] = np.array(list(metrics.keys()))
    save['ev_values'] = np.array([float(v) for v in metrics.values()])
    save['ev_append'] = np.stack(mme.append_indexes)
    np.savez_compressed(os.path.join(OUT, 'evaluator.npz'), **save)


from helpers import T2M_DIMS, T2M_TEXT  # noqa: E402


def t2m_evaluator():
    """HumanML3D / KIT evaluator (T2MContrastiveModel, t2m_bigru.py): movement conv encoder + BiGRU motion / text heads."""
    import importlib
    from oracle import t2m_eval_oracle as TO
    ref_shim.install()
    ref_shim._shell('mogen.models.rnns', ref_shim.REF + '/mogen/models/rnns')
    m = importlib.import_module('mogen.models.rnns.t2m_bigru')
    mov = m.MovementConvEncoder(T2M_DIMS['input_size'] - 4, T2M_DIMS['movement_hidden_size'], T2M_DIMS['movement_latent_size']).eval()
    mot = m.MotionEncoderBiGRUCo(T2M_DIMS['movement_latent_size'], T2M_DIMS['motion_hidden_size'], T2M_DIMS['motion_latent_size']).eval()
    txt = m.TextEncoderBiGRUCo(**T2M_TEXT).eval()
    shapes = W.t2m_eval_param_shapes(**T2M_DIMS, **T2M_TEXT)
    ref_keys = {}
    for pre, mod in (('movement_encoder.', mov), ('motion_encoder.', mot), ('text_encoder.', txt)):
        ref_keys.update({pre + k: v for k, v in mod.state_dict().items()})
    assert set(ref_keys) == set(shapes), sorted(set(ref_keys) ^ set(shapes))[:8]
    for k, v in ref_keys.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    sd = W.make_t2m_eval_state(shapes, seed=12)
    for pre, mod in (('movement_encoder.', mov), ('motion_encoder.', mot), ('text_encoder.', txt)):
        mod.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)

    class Enc(torch.nn.Module):          # T2MMotionEncoder.forward without its constructor (which only wires the two modules)
        movement_encoder, motion_encoder = mov, mot
        forward = m.T2MMotionEncoder.forward
    g = torch.Generator().manual_seed(81)
    B, T, S = 4, 40, 12
    motion = torch.randn(B, T, T2M_DIMS['input_size'], generator=g)
    lengths = torch.tensor([33, 40, 8, 21])
    word = torch.randn(B, S, T2M_TEXT['word_size'], generator=g)
    pos = F_one_hot(torch.randint(0, 15, (B, S), generator=g), 15)
    sent = torch.tensor([12, 5, 9, 3])
    import contextlib, io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        ref_m = Enc()(motion, lengths, None)
        ref_t = txt(word, pos, sent)
    om = TO.encode_motion(sd, motion, lengths)
    ot = TO.encode_text(sd, word, pos, sent)
    print(f't2m_evaluator: motion {tuple(ref_m.shape)} oracle vs reference {maxabs(ref_m, om):.2e}; text oracle vs reference {maxabs(ref_t, ot):.2e}')
    assert maxabs(ref_m, om) <= 1e-5 and maxabs(ref_t, ot) <= 1e-5
    np.savez_compressed(os.path.join(OUT, 't2m_evaluator.npz'), motion=motion.numpy(), lengths=lengths.numpy(), motion_emb=ref_m.numpy(),
                        word_emb=word.numpy(), pos_onehot=pos.numpy(), sent_len=sent.numpy(), text_emb=ref_t.numpy(), seed=np.int64(12))


def F_one_hot(idx, n):
    return torch.nn.functional.one_hot(idx, n).float()


def full():
    dims, B, T = FULL, 1, 196
    t0 = time.time()
    m, sd = build_ref(dims)
    print(f'full: built reference 0.125b in {time.time()-t0:.1f}s, '
          f'{sum(p.numel() for p in m.parameters())/1e6:.1f} M params')
    x_T, xf, mask = synth_inputs(dims, B, T, seed=21)
    save = dict(input_seed=np.int64(21))
    for t in (999, 57):
        with torch.no_grad():
            r = m(x_T, torch.full((B,), t, dtype=torch.long), **model_kwargs(xf, mask))
        o = O.denoise(sd, dims, x_T, t, xf, mask)
        print(f'full_denoise t={t}: oracle vs reference {maxabs(r, o):.2e}  |x0| mean {float(r.abs().mean()):.3f}')
        assert maxabs(r, o) <= 1e-5
        save[f'x0_t{t}'] = r.numpy()
    # padded sample
    _, _, mask2 = synth_inputs(dims, B, T, seed=21, lengths=[150])
    with torch.no_grad():
        r = m(x_T, torch.full((B,), 500, dtype=torch.long), **model_kwargs(xf, mask2))
    o = O.denoise(sd, dims, x_T, 500, xf, mask2)
    print(f'full_denoise t=500 len=150: oracle vs reference {maxabs(r, o):.2e}')
    assert maxabs(r, o) <= 1e-5
    save['x0_t500_len150'] = r.numpy()
    np.savez_compressed(os.path.join(OUT, 'full_denoise.npz'), **save)

    diff = ref_shim.build_reference_diffusion(DIFF_DDIM)
    t0 = time.time()
    tr = run_ref_loop(m, diff, 'ddim', x_T, xf, mask, seed=7)
    t_ref = time.time() - t0
    t0 = time.time()
    to = run_oracle_loop(sd, dims, O.Schedule(1000, DIFF_DDIM['respace']), 'ddim', x_T, xf, mask, seed=7)
    t_or = time.time() - t0
    e = [maxabs(a, b) for a, b in zip(tr, to)]
    print(f'full_ddim: reference {t_ref:.1f}s, oracle {t_or:.1f}s; oracle vs reference max {max(e):.2e} '
          f'final {e[-1]:.2e}')
    assert e[-1] <= 1e-4
    np.savez_compressed(os.path.join(OUT, 'full_ddim.npz'), input_seed=np.int64(21), noise_seed=np.int64(7),
                        final=tr[-1].numpy())


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--skip-full', action='store_true')
    ap.add_argument('--only', default=None, help='comma list of fixture groups to regenerate')
    a = ap.parse_args()
    torch.set_num_threads(min(32, os.cpu_count()))   # torch-CPU degrades badly on >64 threads
    groups = dict(schedules=schedules, small_modules=small_modules, small_loops=small_loops, preseq=preseq, control=control,
                  repaint=repaint, text_encoder=text_encoder, clip_tower=clip_tower, wav_encoder=wav_encoder, control_wav=control_wav, skeleton_parts=skeleton_parts, evaluator=evaluator, t2m_evaluator=t2m_evaluator, full=full)
    for name, fn in groups.items():
        if a.only is not None and name not in a.only.split(','):
            continue
        if name == 'full' and a.skip_full and a.only is None:
            continue
        fn()
    print('golden fixtures written to', OUT)
