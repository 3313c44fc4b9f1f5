"""GPU (MI355X): HIP path vs the CPU oracle and the committed golden vectors, through the C-ABI.

Tolerances: the north star asks <= 1e-3 abs fp32 on the final pose tensor; single denoiser calls
and short trajectories are held to 2e-4 (observed ~1e-5)."""
import ctypes
import os
import time

import numpy as np
import pytest
import torch

DEFAULT_CHAIN = 8388599 | (1 << 24) | (1 << 26) | (1 << 27) | (1 << 29)        # McOptions::chain (mc_model.hip): every schedule / fusion bit but 3, 23, 25 and 28

from helpers import (CTRL, CTRL_COPY, CTRL_FEATS, FULL, HML_FULL, HML_SMALL, KIT_SMALL, SMALL, SMALL_SEED, load,
                     step_noise_from_seed, synth_inputs)

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
TOL_STEP = 2e-4
TOL_FINAL = 1e-3


def T_(a):
    return torch.from_numpy(np.asarray(a))


def maxabs(a, b):
    return float((a.detach().cpu().double().reshape(-1) - b.detach().cpu().double().reshape(-1)).abs().max())


@pytest.fixture(scope='module')
def small_model():
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    sd = W.make_state_dict(SMALL, SMALL_SEED)
    nm = NativeModel(SMALL, sd, cfg_scale=SMALL['scale'])
    yield sd, nm
    nm.close()


@pytest.fixture(scope='module')
def full_model():
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    sd = W.make_state_dict(FULL, 0)
    nm = NativeModel(FULL, sd, cfg_scale=FULL['scale'])
    yield sd, nm
    nm.close()


def test_native_library_is_what_runs():
    from motioncraft_amd import lib
    l = lib.load(require_gpu=True)
    maps = open('/proc/self/maps').read()
    assert 'libmotioncraft_amd.so' in maps


def test_gemm_family_vs_fp64():
    from motioncraft_amd import lib as L_
    from motioncraft_amd.engine import _ptr, _stream
    lib = L_.load(require_gpu=True)
    g = torch.Generator().manual_seed(0)
    for (M, N, K, act, res) in [(128, 128, 32, 0, False), (300, 200, 64, 1, True), (77, 322, 1536, 0, False),
                                (1000, 64, 192, 2, False), (5, 3072, 2048, 0, False), (257, 129, 20, 0, True),
                                (1, 1, 4, 0, False),
                                # the small-M kernel picks its tile width per launch (mc_launch_gemm_small): 64 x 48 tiles
                                # (B=1 FiLM Linear), 64 x 96 (B=2), 48-wide with ragged N and M (scalar epilogue), 64 x 64
                                (392, 1536, 1536, 0, True), (784, 1536, 1536, 0, True), (37, 144, 96, 0, True),
                                (100, 96, 64, 0, False), (1176, 1536, 256, 0, True)]:
        a = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        ref = a.double() @ w.double().t() + b.double()
        ref = torch.nn.functional.gelu(ref) if act == 1 else torch.nn.functional.silu(ref) if act == 2 else ref
        if res:
            ref = ref + r.double()
        ad, wd, bd, rd = a.cuda(), w.cuda(), b.cuda(), r.cuda()
        c = torch.empty(M, N, device='cuda')
        L_.check(lib.mc_op_gemm(_ptr(ad), _ptr(wd), _ptr(bd), _ptr(rd if res else None), _ptr(c), M, N, K, K, act,
                                _stream()))
        torch.cuda.synchronize()
        assert maxabs(c, ref) <= 2e-5, (M, N, K)
    with pytest.raises(RuntimeError):
        L_.check(lib.mc_op_gemm(_ptr(ad), _ptr(wd), None, None, _ptr(c), 4, 4, 3, 3, 0, _stream()))  # K % 4 != 0


def test_small_gemm_random_shapes_vs_fp64():
    """The small-M GEMM picks its kernel and tile width per launch (64 x 64 on the 32x32 MFMA, 64 x 48 / 64 x 96 on the 16x16
    one, float4 or guarded scalar epilogue): random (M, N, K, bias, residual) against fp64, with guard rows behind C
    (a tile wider than the ragged edge of N must not store past it)."""
    import random
    from motioncraft_amd import lib as L_
    from motioncraft_amd.engine import _ptr, _stream
    lib = L_.load(require_gpu=True)
    rnd = random.Random(1)
    for it in range(90):
        M = rnd.choice([1, 5, 31, 33, 63, 64, 65, 100, 127, 200, 392, 500, 784, 1000, 1176, 2000])
        N = rnd.choice([4, 16, 44, 48, 52, 64, 96, 100, 144, 192, 240, 288, 322, 384, 480, 768, 1536])
        K = rnd.choice([32, 64, 96, 128, 256, 1536])
        bias, res = rnd.random() < 0.7, rnd.random() < 0.5
        g = torch.Generator().manual_seed(it)
        a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
        b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
        ad, wd, bd, rd = a.cuda(), w.cuda(), b.cuda(), r.cuda()
        c = torch.full((M + 2, N), 777.0, device='cuda')
        L_.check(lib.mc_op_gemm(_ptr(ad), _ptr(wd), _ptr(bd) if bias else None, _ptr(rd) if res else None, _ptr(c), M, N, K, K, 0,
                                _stream()))
        torch.cuda.synchronize()
        ref = a.double() @ w.double().t() + (b.double() if bias else 0) + (r.double() if res else 0)
        assert bool((c[M:] == 777.0).all()), ('stored past the last row', M, N, K)
        assert maxabs(c[:M], ref) <= 3e-5, (M, N, K, bias, res)


def test_folded_decoder_tail_op_both_kernels_vs_fp64():
    """mc_op_gemm_tail: C[r] = (wc h[r] + wu h[r + M]) W0^T + (wc a[r] + wu a[r + M]) W1^T + b0 + b1 (the CFG combination + last FiLM Linear
    + pose decoder of stmogen.py:505-544, 757-760 as one GEMM pass) -- the column-tile kernel (gemm_tail_k) and the block-range kernel
    (gemm_tail2_k: one contiguous range of 16 x 16 output blocks per workgroup, A read once) against fp64 on ragged shapes: M not a
    multiple of 16, N = 322 / 263 / 251 (the three pose widths) / 16, ranges that start and end inside a row tile, guard rows behind C."""
    from motioncraft_amd import lib as L_
    from motioncraft_amd.engine import _ptr, _stream
    lib = L_.load(require_gpu=True)
    ineligible = []
    for it, (M, N, K) in enumerate([(75, 263, 512), (72, 322, 1536), (1000, 322, 1536), (12544, 322, 1536), (3136, 251, 768), (200, 16, 64),
                                    (6272, 322, 1536), (4097, 322, 96), (5, 322, 64), (17, 100, 128), (300, 336, 96), (33, 17, 160), (19200, 322, 768)]):
        g = torch.Generator(device='cuda').manual_seed(it)
        h = torch.randn(2 * M, K, device='cuda', generator=g)
        a = torch.randn(2 * M, K, device='cuda', generator=g)
        w = torch.randn(2, N, K, device='cuda', generator=g) / K ** 0.5
        b = torch.randn(2, N, device='cuda', generator=g)
        wc, wu = 3.25, -2.25
        sub = torch.arange(0, M, max(1, M // 300), device='cuda')
        sub = torch.unique(torch.cat([sub, torch.tensor([0, M - 1], device='cuda')]))
        ref = ((wc * h[sub].double() + wu * h[sub + M].double()) @ w[0].double().t() + (wc * a[sub].double() + wu * a[sub + M].double()) @ w[1].double().t()
               + b[0].double() + b[1].double())
        for variant in (1, 2):
            c = torch.full((M + 2, N), 777.0, device='cuda')
            c2 = torch.full((M + 2, N), 777.0, device='cuda')
            rc = lib.mc_op_gemm_tail(_ptr(h), _ptr(a), _ptr(w), _ptr(b), _ptr(c), _ptr(c2), M, N, K, wc, wu, variant, _stream())
            if variant == 2 and rc != 0:
                # a request for the block-range form where it is not eligible is an ERROR, not a silent fall-back to the column tiles
                # (ADVICE r05): only shapes outside the step's (few rows x one column block: too many row tiles per range)
                assert b'not eligible' in lib.mc_last_error(), lib.mc_last_error()
                assert (N, K) != (322, 1536), (M, N, K)
                ineligible.append((M, N, K))
                continue
            L_.check(rc)
            torch.cuda.synchronize()
            assert bool((c[M:] == 777.0).all()) and bool((c2[M:] == 777.0).all()), ('stored past the last row', M, N, K, variant)
            assert bool(torch.isfinite(c[:M]).all())
            e = maxabs(c[sub], ref)
            assert e <= 1e-4, (M, N, K, variant, e)
    print('gemm_tail variant 2 refused (not eligible):', ineligible)
    assert len(ineligible) <= 3


def test_ln_rows_and_sampler_update_ops():
    from motioncraft_amd import lib as L_
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import _ptr, _stream
    from oracle import stmogen_oracle as O
    lib = L_.load(require_gpu=True)
    g = torch.Generator().manual_seed(1)
    for Lw in (32, 64, 128, 256):
        x = torch.randn(1000, Lw, generator=g) * 2 + 0.5
        ga, be, add = torch.randn(Lw, generator=g), torch.randn(Lw, generator=g), torch.randn(24, Lw, generator=g)
        ref = torch.nn.functional.layer_norm(x, (Lw,), ga, be) + add.repeat(42, 1)[:1000]
        xd, gd, bd, addd = x.cuda(), ga.cuda(), be.cuda(), add.cuda()
        y = torch.empty(1000, Lw, device='cuda')
        L_.check(lib.mc_op_ln_rows(_ptr(xd), Lw, _ptr(gd), _ptr(bd), _ptr(addd), 24, _ptr(y), 1000, Lw, _stream()))
        torch.cuda.synchronize()
        assert maxabs(y, ref) <= 1e-5
    base = dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large')
    n = 2 * 24 * 322
    xt, ot, on, nz = (torch.randn(n, generator=g) for _ in range(4))
    for mode, respace in (('ddpm', None), ('ddim', '15,15,8,6,6')):
        d = build_diffusion(dict(base, respace=respace))
        s = O.Schedule(1000, respace)
        for i in (d.num_timesteps - 1, 7, 0):
            c = d.step_coefs(i, mode, 6.5)
            x0_ref = ot * c.text_coef + on * c.none_coef
            ref = (O.ddpm_step if mode == 'ddpm' else O.ddim_step)(s, i, xt, x0_ref, nz)
            xtd, otd, ond, nzd = xt.cuda(), ot.cuda(), on.cuda(), nz.cuda()
            xp, x0 = torch.empty(n, device='cuda'), torch.empty(n, device='cuda')
            L_.check(lib.mc_op_sampler_update(_ptr(xtd), _ptr(otd), _ptr(ond), _ptr(nzd), _ptr(xp), _ptr(x0), n,
                                              ctypes.byref(c), _stream()))
            torch.cuda.synchronize()
            assert maxabs(x0, x0_ref) <= 2e-5 and maxabs(xp, ref) <= 5e-5, (mode, i)


def test_small_denoiser_stage_by_stage_vs_reference_golden(small_model):
    from oracle import stmogen_oracle as O
    sd, nm = small_model
    g = load('small_modules.npz')
    x_t, xf, mask = T_(g['x_t']), T_(g['xf_out']), T_(g['motion_mask'])
    B, T = 2, 24
    ctx = nm.context(B, T, max_steps=1)
    ctx.set_timesteps([int(g['t'])])
    ctx.set_condition(xf.cuda(), mask.cuda())
    assert maxabs(ctx.buffer('emb')[:SMALL['Te']], T_(g['emb'])[0]) <= 1e-5
    for i in range(SMALL['NL']):
        assert maxabs(ctx.buffer('tf', i), T_(g[f'layer{i}.text_feat'])) <= 1e-5
    xd = x_t.cuda()
    for i in range(SMALL['NL']):
        ctx.denoise(xd, 0, stop_after_layers=i + 1)
        assert maxabs(ctx.buffer('mf'), T_(g[f'layer{i}.motion_feat'])) <= TOL_STEP
        assert maxabs(ctx.buffer('h'), T_(g[f'layer{i}.after_ffn'])) <= TOL_STEP
        dropped = int((ctx.buffer('comb_w') == 0).sum())
        assert dropped == int(g[f'layer{i}.dropped'].sum())        # capacity overflow handled like the oracle
    out2 = ctx.denoise(xd, 0)
    assert maxabs(out2, T_(g['out2'])) <= TOL_STEP
    w = (1 - (1000 - int(g['t'])) / 1000) * SMALL['scale'] + 1
    assert maxabs(out2[:B] * w + out2[B:] * (1 - w), T_(g['x0'])) <= TOL_STEP
    # determinism: slot order inside an expert is atomics-dependent, the values must not be
    again = ctx.denoise(xd, 0)
    assert torch.equal(out2, again)
    ctx.close()


def _arch_small(sd):
    import motioncraft_amd as mc
    cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
    arch = mc.build_architecture(cfg.model)
    arch.load_state_dict({'model.' + k: v for k, v in sd.items()})
    return cfg, arch


def test_small_ddim_50_steps_through_the_reference_api(small_model):
    """configs -> build_architecture -> MotionDiffusion.forward(**kwargs) -> list of per-sample dicts."""
    sd, _ = small_model
    g = load('small_ddim.npz')
    cfg, arch = _arch_small(sd)
    B, T = 2, 24
    noises = step_noise_from_seed(int(g['noise_seed']), (B, T, 322), 50)
    mask = T_(g['motion_mask'])
    res = arch(motion=torch.zeros(B, T, 322), motion_mask=mask, motion_length=mask.sum(1, keepdim=True).long(),
               motion_metas=[{'text': 'a'}, {'text': 'b'}], xf_out=T_(g['xf_out']),
               inference_kwargs=dict(noise=T_(g['x_T']), step_noise=lambda i: noises[49 - i]))
    assert isinstance(res, list) and len(res) == B
    assert set(res[0]) == {'motion', 'pred_motion', 'motion_length', 'motion_mask', 'pred_motion_length',
                           'pred_motion_mask', 'text'}
    assert res[1]['text'] == 'b' and not res[0]['pred_motion'].is_cuda
    assert torch.equal(res[0]['pred_motion_mask'], mask[0]) and int(res[0]['pred_motion_length']) == int(mask[0].sum())
    final = torch.stack([r['pred_motion'] for r in res])
    assert maxabs(final, T_(g['final'])) <= TOL_FINAL
    arch.model.release()


def test_small_ddim_and_ddpm_trajectories(small_model):
    from motioncraft_amd.diffusion import build_diffusion
    sd, nm = small_model
    base = dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large')

    class M:        # minimal `model` object of the sampler API
        cfg_scale = SMALL['scale']

        def sampling_context(self, B, T, tmap, kw, device=None):
            ctx = nm.context(B, T, max_steps=len(tmap))
            ctx.set_timesteps(tmap)
            ctx.set_condition(kw['xf_out'].cuda(), kw['motion_mask'].cuda())
            self.ctx = ctx
            return ctx
    for name, mode, respace, nsteps, every in (('small_ddim.npz', 'ddim', '15,15,8,6,6', 50, 10),
                                               ('small_ddpm.npz', 'ddpm', None, 20, 5)):
        g = load(name)
        d = build_diffusion(dict(base, respace=respace))
        S = d.num_timesteps
        noises = step_noise_from_seed(int(g['noise_seed']), (2, 24, 322), nsteps)
        traj = []
        m = M()
        loop = d.ddim_sample_loop if mode == 'ddim' else d.p_sample_loop
        loop(m, (2, 24, 322), noise=T_(g['x_T']), clip_denoised=False,
             model_kwargs=dict(xf_out=T_(g['xf_out']), motion_mask=T_(g['motion_mask']), y={}),
             step_noise=lambda i: noises[S - 1 - i], num_steps=nsteps, trajectory=traj)
        for n, ref in zip(range(every - 1, nsteps, every), g['traj']):
            assert maxabs(traj[n][1], T_(ref)) <= TOL_FINAL, (name, n)
        m.ctx.close()


def test_pre_seq_and_transl_req_seeding_vs_reference_golden(small_model):
    """gaussian_diffusion.py:664-674, 816-820: every step overwrites x[:, :Tp] with q_sample(pre_seq, t) (and the
    requested translation channels of frames 0-1) before the network call -- fused into the pose-row padding pass
    (mc_sample_step_seeded).  Golden: the reference's p_sample_loop / ddim_sample_loop with pre_seq / transl_req,
    through MotionDiffusion.forward(inference_kwargs=...) like the reference passes them (diffusion_architecture.py:175)."""
    from motioncraft_amd.diffusion import build_diffusion
    sd, nm = small_model
    g = load('preseq_small.npz')
    x_T, xf, mask, pre = T_(g['x_T']), T_(g['xf_out']), T_(g['motion_mask']), T_(g['pre_seq'])
    transl = [[int(r[0]), float(r[1]), float(r[2])] for r in g['transl_req']]
    B, T, C = x_T.shape

    def draws(seed, nsteps, ntransl):
        # the reference's global-RNG stream: per step randn_like(pre_seq), randn(2) per item, randn_like(x)
        gen = torch.Generator().manual_seed(int(seed))
        for _ in range(nsteps):
            yield torch.randn(pre.shape, generator=gen)
            for _ in range(ntransl):
                yield torch.randn(2, generator=gen)
            yield torch.randn(B, T, C, generator=gen)

    class M:
        cfg_scale = SMALL['scale']

        def sampling_context(self, B, T, tmap, kw, device=None):
            ctx = nm.context(B, T, max_steps=len(tmap))
            ctx.set_timesteps(tmap)
            ctx.set_condition(kw['xf_out'].cuda(), kw['motion_mask'].cuda())
            self.ctx = ctx
            return ctx
    base = dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large')
    kw = dict(xf_out=xf, motion_mask=mask, y={})
    d, m, traj = build_diffusion(base), M(), []
    d.p_sample_loop(m, (B, T, C), noise=x_T, clip_denoised=False, model_kwargs=kw, pre_seq=pre, transl_req=transl,
                    step_noise=draws(g['ddpm_seed'], 12, len(transl)), num_steps=12, trajectory=traj)
    for n, ref in zip(range(3, 12, 4), g['ddpm_traj']):
        assert maxabs(traj[n][1], T_(ref)) <= TOL_FINAL, n
    m.ctx.close()
    with pytest.raises(ValueError):          # a per-step list cannot serve draws of three different shapes
        d.p_sample_loop(m, (B, T, C), noise=x_T, clip_denoised=False, model_kwargs=kw, pre_seq=pre, step_noise=[x_T] * 1000)
    m.ctx.close()
    with pytest.raises(RuntimeError):        # like the reference: transl_req only broadcasts for B <= 2
        d.p_sample_loop(m, (3, T, C), noise=torch.zeros(3, T, C), clip_denoised=False,
                        model_kwargs=dict(xf_out=xf[:1].repeat(3, 1, 1), motion_mask=mask[:1].repeat(3, 1), y={}), transl_req=transl)
    m.ctx.close()
    # DDIM through the architecture API
    cfg, arch = _arch_small(sd)
    res = arch(motion=torch.zeros(B, T, C), motion_mask=mask, motion_length=mask.sum(1, keepdim=True).long(),
               motion_metas=[{'text': 'a'}, {'text': 'b'}], xf_out=xf,
               inference_kwargs=dict(noise=x_T, pre_seq=pre, step_noise=draws(g['ddim_seed'], 50, 0)))
    final = torch.stack([r['pred_motion'] for r in res])
    err = maxabs(final, T_(g['ddim_final']))
    print(f'pre_seq DDIM 50 steps: |hip - reference| {err:.2e}')
    assert err <= TOL_FINAL
    arch.model.release()


def test_full_size_denoise_vs_reference_golden(full_model):
    sd, nm = full_model
    g = load('full_denoise.npz')
    x_T, xf, mask = synth_inputs(FULL, 1, 196, int(g['input_seed']))
    ctx = nm.context(1, 196, max_steps=3)
    ctx.set_timesteps([999, 57, 500])
    ctx.set_condition(xf.cuda(), mask.cuda())
    for s, t in ((0, 999), (1, 57)):
        out2 = ctx.denoise(x_T.cuda(), s)
        w = (1 - (1000 - t) / 1000) * FULL['scale'] + 1
        assert maxabs(out2[:1] * w + out2[1:] * (1 - w), T_(g[f'x0_t{t}'])) <= TOL_STEP, t
    _, _, mask2 = synth_inputs(FULL, 1, 196, int(g['input_seed']), lengths=[150])
    ctx.set_condition(xf.cuda(), mask2.cuda())
    out2 = ctx.denoise(x_T.cuda(), 2)
    w = (1 - (1000 - 500) / 1000) * FULL['scale'] + 1
    assert maxabs(out2[:1] * w + out2[1:] * (1 - w), T_(g['x0_t500_len150'])) <= TOL_STEP
    ctx.close()


def test_full_size_50_step_ddim_vs_reference_golden(full_model):
    """North-star bar: <= 1e-3 abs on the final 322-d pose tensor, identical noise seeds.

    tutel's capacity dropping makes the network discontinuous: if fp32 rounding lands one (token,
    choice) pair on the other side of an expert's capacity boundary, that token changes by O(1)
    and the trajectories separate for good (the same happens between two CPU BLAS builds).  So the
    test walks the HIP trajectory in LOCKSTEP: at every one of the 50 steps the oracle is evaluated
    on the HIP path's own x_t with the HIP path's discrete routing decisions and must reproduce
    x_{t-1} within 1e-3 (observed ~5e-6), and the discrete decisions are compared with what the
    oracle would have chosen freely on the same input (recorded runs: 0 or 1 differing pair over the
    200 routings of the loop -- a differing pair whose gate weight is tiny does not move the result).
    The distance of the final sample to the committed reference golden is asserted UNCONDITIONALLY
    (<= 1e-3, observed 6.4e-6 .. 6.7e-6): a flip that did separate the trajectories (O(0.5) on the
    final pose) fails the test, as the north-star bar says it should."""
    from motioncraft_amd.diffusion import build_diffusion
    from oracle import stmogen_oracle as O
    sd, nm = full_model
    g = load('full_ddim.npz')
    x_T, xf, mask = synth_inputs(FULL, 1, 196, int(g['input_seed']))
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                             model_var_type='fixed_large', respace='15,15,8,6,6'))
    sched = O.Schedule(1000, '15,15,8,6,6')
    ctx = nm.context(1, 196, max_steps=50)
    ctx.enable_capture()
    ctx.set_timesteps(d.timestep_map)
    ctx.set_condition(xf.cuda(), mask.cuda())
    noises = step_noise_from_seed(int(g['noise_seed']), (1, 196, 322), 50)
    torch.set_num_threads(min(32, os.cpu_count()))
    tf = O.precompute_text(sd, xf, FULL)
    x = x_T.cuda()
    flips, worst = 0, 0.0
    for n, i in enumerate(range(49, -1, -1)):
        x_in = x.cpu()
        x = ctx.sample_step(x, i, d.step_coefs(i, 'ddim', FULL['scale']), noises[n].cuda())
        forced = [ctx.routing(l) for l in range(FULL['NL'])]
        cap = {}
        x0 = O.denoise(sd, FULL, x_in, sched.timestep_map[i], xf, mask, text_feats=tf, forced_routing=forced, cap=cap)
        ref = O.ddim_step(sched, i, x_in, x0, noises[n])
        worst = max(worst, maxabs(x, ref))
        for l in range(FULL['NL']):
            free = cap[f'layer{l}']['routing']['free']
            flips += int((torch.stack(free['indices'], 1) != forced[l][0]).sum())
            flips += int((torch.stack(free['keeps'], 1) != forced[l][1]).sum())
    final_err = maxabs(x, T_(g['final']))
    print(f'50-step DDIM lockstep: worst per-step |hip - oracle| {worst:.2e}; routing flips {flips}; '
          f'final vs reference golden {final_err:.2e}')
    assert worst <= TOL_FINAL
    assert flips <= 2                      # observed: 0 or 1 (of 200 routings x 4704 pairs) in the recorded runs
    # the north-star assertion, unconditional: <= 1e-3 on the final 322-d pose tensor vs the reference's own result
    assert final_err <= TOL_FINAL, final_err
    ctx.close()


@pytest.mark.parametrize('B', [64, 16])
def test_baseline_batch_64_single_step_vs_oracle(full_model, B):
    """BASELINE.json configs[1] size (B=64, T=196, mixed lengths): one denoiser call vs the CPU oracle.  (B=16: the smallest
    batches of the two-stream schedule -- sample groups of 3136 rows, where the small-batch kernel choices end.)

    At N = 301 056 tokens neighbouring importance scores at an expert's capacity boundary are ~1e-6
    apart -- the size of fp32 rounding differences between ANY two implementations of the gate
    (including torch-CPU at two thread counts) -- so a handful of (token, choice) pairs can land on
    the other side of the drop threshold, each changing its token by O(1) (tutel's batch-prioritised
    dropping is discontinuous by construction).  Parity is therefore asserted in two parts:
      (a) the DISCRETE decisions differ from the free-running oracle in at most a tiny fraction of
          pairs, all of them capacity drops (never a different expert id for a confident token);
      (b) with the oracle teacher-forced to the HIP path's decisions, every CONTINUOUS quantity
          matches within the north-star tolerance of 1e-3."""
    from oracle import stmogen_oracle as O
    sd, nm = full_model
    T = 196
    g = torch.Generator().manual_seed(5)
    lengths = [int(v) for v in torch.randint(64, 197, (B,), generator=g)]
    x_T, xf, mask = synth_inputs(FULL, B, T, seed=33, lengths=lengths)
    ctx = nm.context(B, T, max_steps=1)
    ctx.enable_capture()
    ctx.set_timesteps([640])
    ctx.set_condition(xf.cuda(), mask.cuda())
    out2 = ctx.denoise(x_T.cuda(), 0)
    again = ctx.denoise(x_T.cuda(), 0)
    assert torch.equal(out2, again)                     # race-free / deterministic
    forced = [ctx.routing(i) for i in range(FULL['NL'])]
    torch.set_num_threads(min(32, os.cpu_count()))
    cap = {}
    ref = O.denoise(sd, FULL, x_T, 640, xf, mask, forced_routing=forced, cap=cap)
    w = (1 - (1000 - 640) / 1000) * FULL['scale'] + 1
    assert maxabs(out2[:B] * w + out2[B:] * (1 - w), ref) <= TOL_FINAL          # (b)
    npairs = 2 * 2 * B * T * FULL['H']
    for i in range(FULL['NL']):                                                 # (a)
        free = cap[f'layer{i}']['routing']['free']
        fidx, fkeep = torch.stack(free['indices'], 1), torch.stack(free['keeps'], 1)
        idx_diff = int((fidx != forced[i][0]).sum())
        keep_diff = int(((fkeep != forced[i][1]) & (fidx == forced[i][0])).sum())
        dropped = int((~fkeep).sum())
        print(f'layer {i}: dropped {dropped}, expert-id flips {idx_diff}, keep flips {keep_diff} of {npairs} pairs')
        # expert-id flips: the free-running ORACLE disagrees with itself by up to 7 per layer at this size (profiles/r04_oracle_self_divergence.txt);
        # keep flips: it never does (0 in every variant), and one moves its token by O(1): held to the observed level (0-1 per 7 routings)
        assert idx_diff <= 8 and keep_diff <= 2, (i, idx_diff, keep_diff)
    ctx.close()


def _routing_flips(free, forced):
    """(expert-id flips, keep flips among equal ids, dropped pairs) of the free-running oracle vs the HIP decisions."""
    fidx, fkeep = torch.stack(free['indices'], 1), torch.stack(free['keeps'], 1)
    idx_diff = int((fidx != forced[0]).sum())
    keep_diff = int(((fkeep != forced[1]) & (fidx == forced[0])).sum())
    return idx_diff, keep_diff, int((~fkeep).sum())


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_baseline_b64_ddpm_lockstep_and_complete_1000_step_loop(full_model, prec):
    """BASELINE.json configs[1] (B=64, T=196, 1000-step DDPM) as a LOOP: the complete 1000-step p_sample loop runs on the
    device (what bench.py extrapolates from its timed steps), and the CPU oracle walks beside it in lockstep -- from the
    HIP path's own x_t, with the same noise, teacher-forced to the HIP path's routing decisions:

      * the first 20 consecutive steps and every 200th step down to t = 0, on 4 of the 64 samples (with the discrete
        decisions forced every token is an independent row, so a sub-batch reproduces the full-batch arithmetic exactly
        at 1/16 of the CPU time): every x_{t-1} within 1e-3 (north-star tolerance);
      * at t = 999 and 0 on the FULL batch, where the free-running oracle's own decisions are compared too: at
        N = 301 056 tokens the scores next to a capacity boundary / the two best experts of a token are ~1e-6 apart, so a
        handful of the 602 112 (token, choice) pairs per routing legitimately differ;
      * every x_t finite, the statistics of x_t following the oracle's at each full-batch checkpoint.
    prec = 'f16x3': the same loop in the fp16-MFMA split mode (mc_ctx_set_precision) against the SAME fp32 oracle and the same
    1e-3 bound, with fewer checkpoints (first 8 steps + every 250th on the sub-batch, t = 500 on the full batch).
    (Checkpoint counts are sized so that the whole `-m gpu` suite stays within a few minutes of oracle CPU time.)"""
    from motioncraft_amd.diffusion import build_diffusion
    from oracle import stmogen_oracle as O
    sd, nm = full_model
    B, T, S, H = 64, 196, 1000, FULL['H']
    g = torch.Generator().manual_seed(5)
    lengths = [int(v) for v in torch.randint(64, 197, (B,), generator=g)]
    x_T, xf, mask = synth_inputs(FULL, B, T, seed=41, lengths=lengths)
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=S, model_mean_type='start_x', model_var_type='fixed_large'))
    sched = O.Schedule(S, None)
    ctx = nm.context(B, T, max_steps=S)
    ctx.set_precision(prec)
    ctx.enable_capture()
    ctx.set_timesteps(d.timestep_map)
    ctx.set_condition(xf.cuda(), mask.cuda())
    torch.set_num_threads(min(32, os.cpu_count()))
    sub = torch.arange(0, B, 16)                                  # 4 samples, mixed lengths
    tf_full = O.precompute_text(sd, xf, FULL)
    # the text K/V hoist is routed over the whole CFG-doubled condition batch: slice the full-batch result
    tf_sub = [t.view(2, B, *t.shape[1:])[:, sub].reshape(2 * len(sub), *t.shape[1:]) for t in tf_full]
    gen = torch.Generator(device='cuda').manual_seed(77)
    x = x_T.cuda()
    nxt = torch.empty_like(x)
    sub_steps = set(range(S - 1, S - (21 if prec == 'f32' else 9), -1)) | set(range(0, S, 200 if prec == 'f32' else 250))
    full_steps = {S - 1, 0} if prec == 'f32' else {500}
    worst_sub, worst_full, flips_idx, flips_keep, npairs = 0.0, 0.0, 0, 0, 2 * 2 * B * T * H
    t0 = time.time()
    for i in range(S - 1, -1, -1):
        eps = torch.randn(B, T, 322, device='cuda', generator=gen)
        chk = i in sub_steps or i in full_steps
        if chk:
            x_in = x.cpu()
        ctx.sample_step(x, i, d.step_coefs(i, 'ddpm', FULL['scale']), eps, x_prev=nxt)
        x, nxt = nxt, x
        if i % 50 == 0:
            assert bool(torch.isfinite(x).all()), i
        if not chk:
            continue
        forced = [ctx.routing(l) for l in range(FULL['NL'])]
        eps_c, x_c = eps.cpu(), x.cpu()
        if i in full_steps:
            cap = {}
            x0 = O.denoise(sd, FULL, x_in, sched.timestep_map[i], xf, mask, text_feats=tf_full, forced_routing=forced, cap=cap)
            ref = O.ddpm_step(sched, i, x_in, x0, eps_c)
            e = maxabs(x_c, ref)
            worst_full = max(worst_full, e)
            for l in range(FULL['NL']):
                a, b, dropped = _routing_flips(cap[f'layer{l}']['routing']['free'], forced[l])
                flips_idx, flips_keep = flips_idx + a, flips_keep + b
                print(f'  step {i} layer {l}: expert-id flips {a}, keep flips {b}, dropped {dropped} of {npairs} pairs')
                assert a <= 8 and b <= 2, (i, l, a, b)               # (see test_baseline_batch_64_single_step_vs_oracle)
            assert abs(float(x_c.std()) - float(ref.std())) <= 1e-3 and abs(float(x_c.mean()) - float(ref.mean())) <= 1e-3
            assert e <= TOL_FINAL, (i, e)
        else:
            fsub = [tuple(v.view(2, B, T * H, 2)[:, sub].reshape(-1, 2) for v in f) for f in forced]
            x0 = O.denoise(sd, FULL, x_in[sub], sched.timestep_map[i], xf[sub], mask[sub], text_feats=tf_sub, forced_routing=fsub)
            ref = O.ddpm_step(sched, i, x_in[sub], x0, eps_c[sub])
            e = maxabs(x_c[sub], ref)
            worst_sub = max(worst_sub, e)
            assert e <= TOL_FINAL, (i, e)
    print(f'B=64 complete 1000-step DDPM loop, precision {prec} ({time.time() - t0:.0f} s): lockstep |hip - oracle| worst {worst_sub:.2e} over '
          f'{len(sub_steps - full_steps)} steps x {len(sub)} samples, {worst_full:.2e} over {len(full_steps)} full-batch steps; free-running '
          f'oracle: expert-id flips {flips_idx}, keep flips {flips_keep} over {len(full_steps) * FULL["NL"]} routings of {npairs} pairs; '
          f'final x std {float(x.std()):.3f}')
    assert bool(torch.isfinite(x).all())
    ctx.close()


def test_baseline_b64_free_running_envelope_vs_free_running_oracle(full_model):
    """BASELINE.json configs[1], the other half of the parity argument (north star: "<= 1e-3 abs on the final pose on
    identical noise seeds"): NOTHING is teacher-forced here.  B=64 x 196 frames, the first 16 steps of the 1000-step DDPM loop
    (gaussian_diffusion.py:698-797), the same x_T / condition / per-step noise on both sides: the HIP path free-running (one
    mc_sample_loop call) against `oracle.sample_loop` free-running, each taking its OWN routing decisions.  The inputs are those of
    tools/oracle_self_divergence.py, so the figure stands beside the oracle-vs-ITSELF envelope of profiles/r04_oracle_self_divergence.txt
    (the same torch-CPU code at another thread count / another summation order of the gate matmul: 1.7e-4 .. 3.9e-4 over these 16
    steps): an implementation is inside the reference's own reproducibility when it stays inside that band."""
    from oracle import stmogen_oracle as O
    from motioncraft_amd.diffusion import build_diffusion
    sd, nm = full_model
    B, T, S, NSTEP = 64, 196, 1000, 16
    g = torch.Generator().manual_seed(0)
    x_T = torch.randn(B, T, 322, generator=g)
    xf = torch.nn.functional.layer_norm(torch.randn(B, FULL['Nt'], FULL['Dt'], generator=g), (FULL['Dt'],))
    mask = torch.ones(B, T)
    noise = [torch.randn(B, T, 322, generator=g) for _ in range(NSTEP)]
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=S, model_mean_type='start_x', model_var_type='fixed_large'))
    ctx = nm.context(B, T, max_steps=S)
    ctx.set_timesteps(d.timestep_map)
    ctx.set_condition(xf.cuda(), mask.cuda())
    order = list(range(S - 1, S - 1 - NSTEP, -1))
    coefs = [d.step_coefs(i, 'ddpm', FULL['scale']) for i in order]
    nz = torch.stack(noise).cuda()
    hip = []
    x = x_T.cuda()
    for n in range(NSTEP):            # (one loop call per step only to keep every intermediate x_t; same kernels as one call over all 16)
        ctx.sample_loop(x, order[n:n + 1], coefs[n:n + 1], noise=nz[n:n + 1])
        hip.append(x.cpu())
    ctx.close()
    torch.set_num_threads(min(32, os.cpu_count()))
    t0 = time.time()
    traj = []
    ref = O.sample_loop(sd, FULL, O.Schedule(S, None), 'ddpm', x_T, xf, mask, step_noise=lambda i: noise[S - 1 - i], num_steps=NSTEP,
                        trajectory=traj)
    per_step = [maxabs(hip[n], traj[n][1]) for n in range(NSTEP)]
    print(f'B=64 free-running HIP vs free-running oracle, {NSTEP} DDPM steps (oracle {time.time() - t0:.0f} s): max|x_hip - x_oracle| per step '
          + ' '.join(f'{e:.1e}' for e in per_step) + f'; final {per_step[-1]:.2e} (oracle vs itself: 1.7e-4 .. 3.9e-4, profiles/r04_oracle_self_divergence.txt)')
    assert bool(torch.isfinite(hip[-1]).all())
    assert max(per_step) <= TOL_FINAL, per_step
    assert maxabs(hip[-1], ref) <= TOL_FINAL


def test_flop_ledger_books_the_executed_flops_of_a_step(full_model):
    """mc_debug_flop_ledger (round 6): what the launchers book for ONE step of the headline workload (B = 64 x 196 frames, both the two-stream
    default and the single-stream schedule the per-kernel roofline is profiled on) equals the closed-form count `bench.py` prints as
    `executed_gflop_per_sample_step` -- the per-kernel roofline (tools/kernel_roofline.py) and the bench line are priced from the same FLOPs.
    (0.2 % allowance: the ledger books the pose-encoder GEMM at its padded K = 352, the closed form at 324.)"""
    import ctypes
    import sys
    from motioncraft_amd import lib as L_
    from motioncraft_amd.diffusion import build_diffusion
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    _, nm = full_model
    lib = L_.load(require_gpu=True)
    B, T = 64, 196
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large'))
    x, xf, mask = synth_inputs(FULL, B, T, seed=3)
    want = bench.executed_flops_per_sample_step(bench.DIMS, T) * B
    for tag, chain in (('two-stream', DEFAULT_CHAIN), ('single-stream', DEFAULT_CHAIN & ~((1 << 5) | (1 << 6) | (1 << 9) | (1 << 16)))):
        ctx = nm.context(B, T, max_steps=1000)
        ctx.set_option('chain', chain)
        ctx.set_timesteps(d.timestep_map)
        ctx.set_condition(xf.cuda(), mask.cuda())
        xd = x.cuda()
        ctx.sample_loop(xd, [999], [d.step_coefs(999, 'ddpm', FULL['scale'])], seed=1)
        torch.cuda.synchronize()
        lib.mc_debug_flop_ledger(1)
        ctx.sample_loop(xd, [998], [d.step_coefs(998, 'ddpm', FULL['scale'])], seed=1)
        torch.cuda.synchronize()
        lib.mc_debug_flop_ledger(0)
        n = lib.mc_debug_flop_ledger_dump(None, 0)
        buf = ctypes.create_string_buffer(int(n))
        lib.mc_debug_flop_ledger_dump(buf, n)
        rows = [l.split('\t') for l in buf.value.decode().splitlines() if l]
        got = sum(float(r[2]) for r in rows)
        print(f'{tag}: ledger {got / 1e9 / B:.3f} GFLOP per sample and step over {sum(int(r[1]) for r in rows)} launches of {len(rows)} (kernel, grid) kinds; '
              f'bench.py closed form {want / 1e9 / B:.3f}')
        assert abs(got - want) <= 2e-3 * want, (tag, got, want)
        assert all('@' in r[0] and int(r[1]) > 0 for r in rows)
        ctx.close()
    # off again: nothing is booked
    lib.mc_debug_flop_ledger(1)
    lib.mc_debug_flop_ledger(0)
    assert lib.mc_debug_flop_ledger_dump(None, 0) == 1


def test_complete_loop_free_running_b8_full_size_vs_the_oracles_own_band(full_model):
    """The north-star sentence at batch > 1, end to end and with NOTHING teacher-forced (VERDICT r05 "What's missing" 2; reference loop
    gaussian_diffusion.py:925-1049): full-size 0.125b, B = 8 x 196 frames, the COMPLETE 50-step DDIM loop, the same x_T / condition /
    per-step noise on every side.  Three free-running trajectories, each taking its own routing decisions:

      hip       one mc_sample_loop call per step (the product path)
      base      oracle.sample_loop (torch-CPU fp32)
      permuted  the same oracle with the cosine gate's projector matmul summed in another K order (tools/oracle_self_divergence.py's
                variant: the same real-number product, another fp32 rounding) -- the reference's OWN reproducibility band
      splitk    ... as the sum of two half-K products

    Printed: max|x_hip - x_base| beside max|x_variant - x_base| every 10 steps, and the first step at which each trajectory leaves the
    base oracle's by more than 1e-3.  Asserted: the final pose of the HIP path is within 1e-3 of the oracle's when the oracle's own
    variants are (the north-star bar as written); when a near-tied capacity decision separates the oracle from ITSELF by more than
    that (measured on MI355X + host, round 6: final 0.80 / 0.78 between oracle variants), the statement that holds without luck is
    relative to the oracle's own band: the HIP path stays within 1e-3 of the base oracle at least as long as the oracle's variants do.
    Which of the two applied is in the printed lines (and in DESIGN.md section 2)."""
    import math
    import torch.nn.functional as F
    from oracle import stmogen_oracle as O, tutel_restated as TR
    from motioncraft_amd.diffusion import build_diffusion
    sd, nm = full_model
    B, T, S = 8, 196, 50
    g = torch.Generator().manual_seed(0)
    x_T = torch.randn(B, T, 322, generator=g)
    xf = F.layer_norm(torch.randn(B, FULL['Nt'], FULL['Dt'], generator=g), (FULL['Dt'],))
    mask = torch.ones(B, T)
    mask[3, 150:] = 0
    noise = [torch.randn(B, T, 322, generator=g) for _ in range(S)]
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large',
                             respace='15,15,8,6,6'))
    ctx = nm.context(B, T, max_steps=S)
    ctx.set_timesteps(d.timestep_map)
    ctx.set_condition(xf.cuda(), mask.cuda())
    order = list(range(S - 1, -1, -1))
    coefs = [d.step_coefs(i, 'ddim', FULL['scale']) for i in order]
    nz = torch.stack(noise).cuda()
    hip = []
    x = x_T.cuda()
    for n in range(S):
        ctx.sample_loop(x, order[n:n + 1], coefs[n:n + 1], noise=nz[n:n + 1])
        hip.append(x.cpu())
    # the same loop as ONE library call: the same bits (what a sampling run issues)
    x1 = x_T.cuda()
    ctx.sample_loop(x1, order, coefs, noise=nz)
    assert torch.equal(x1.cpu(), hip[-1])
    ctx.close()

    orig = TR.gate_scores
    perm = {}

    def gate_permuted(xx, proj_w, proj_b, sim_matrix, temperature):
        K = xx.shape[1]
        if K not in perm:
            perm[K] = torch.randperm(K, generator=torch.Generator().manual_seed(123))
        q = perm[K]
        return orig(xx[:, q].contiguous(), proj_w[:, q].contiguous(), proj_b, sim_matrix, temperature)

    def gate_splitk(xx, proj_w, proj_b, sim_matrix, temperature):
        K = xx.shape[1] // 2
        proj = (xx[:, :K] @ proj_w[:, :K].t() + xx[:, K:] @ proj_w[:, K:].t()) + proj_b
        logits = torch.matmul(F.normalize(proj, dim=1), F.normalize(sim_matrix.to(torch.float32), dim=0))
        return F.softmax(logits * torch.clamp(temperature.to(torch.float32), max=math.log(1.0 / 0.01)).exp(), dim=1)

    torch.set_num_threads(min(32, os.cpu_count()))
    sched = O.Schedule(1000, '15,15,8,6,6')
    t0 = time.time()
    trajs = {}
    for name, gate in (('base', orig), ('permuted', gate_permuted), ('splitk', gate_splitk)):
        TR.gate_scores = gate
        try:
            tr = []
            O.sample_loop(sd, FULL, sched, 'ddim', x_T, xf, mask, step_noise=lambda i: noise[S - 1 - i], trajectory=tr)
            trajs[name] = [t[1] for t in tr]
        finally:
            TR.gate_scores = orig
    e_hip = [maxabs(hip[n], trajs['base'][n]) for n in range(S)]
    e_perm = [maxabs(trajs['permuted'][n], trajs['base'][n]) for n in range(S)]
    e_split = [maxabs(trajs['splitk'][n], trajs['base'][n]) for n in range(S)]
    marks = list(range(9, S, 10))
    print(f'B=8 full-size complete 50-step DDIM loop, free-running ({time.time() - t0:.0f} s of oracle): max-abs vs the base oracle after steps '
          f'{[m + 1 for m in marks]}: hip ' + ' '.join(f'{e_hip[m]:.1e}' for m in marks) + ' | oracle(permuted) '
          + ' '.join(f'{e_perm[m]:.1e}' for m in marks) + ' | oracle(splitk) ' + ' '.join(f'{e_split[m]:.1e}' for m in marks)
          + f'; final hip {e_hip[-1]:.2e}, permuted {e_perm[-1]:.2e}, splitk {e_split[-1]:.2e}')
    assert bool(torch.isfinite(hip[-1]).all())

    def first_over(e):            # 1-based step at which a trajectory has left the base oracle's by more than the north-star tolerance (S + 1: never)
        return next((n + 1 for n, v in enumerate(e) if v > TOL_FINAL), S + 1)
    n_hip, n_perm, n_split = first_over(e_hip), first_over(e_perm), first_over(e_split)
    band = max(e_perm[-1], e_split[-1])
    print(f'  first step beyond 1e-3 of the base oracle: hip {n_hip}, oracle(permuted) {n_perm}, oracle(splitk) {n_split} (of {S}; {S + 1} = never); '
          f'last figure before it: hip {max(e_hip[:n_hip - 1], default=0.0):.1e}')
    if band <= TOL_FINAL:
        print('  bar applied: the north-star 1e-3 on the final pose (the oracle stays inside it against itself)')
        assert e_hip[-1] <= TOL_FINAL, e_hip
    else:
        # tutel's capacity cut makes the network discontinuous: one near-tied (token, choice) pair landing on the other side of an expert's
        # capacity moves that token by O(1), and 40 more free-running steps spread it.  The reference restated in torch-CPU does this to
        # ITSELF under a different fp32 summation order of the gate matmul, so "<= 1e-3 on the final pose" cannot hold end to end at this
        # batch size for ANY two fp32 evaluations.  What can be asserted without luck: the HIP path stays within 1e-3 of the base oracle AT
        # LEAST AS LONG as the oracle's own variants do, and once separated it is separated by the same order of magnitude.
        print(f'  bar applied: the oracle separates from ITSELF (final {band:.2e}); asserted: hip follows the base oracle within 1e-3 at least as long '
              f'as the oracle\'s own variants, and ends within 3 x their final distance')
        assert n_hip >= min(n_perm, n_split), (n_hip, n_perm, n_split)
        assert e_hip[-1] <= 3.0 * band, (e_hip[-1], band)


@pytest.mark.parametrize('case', ['s2g_b32', 'm2d_160_windows'])
def test_baseline_control_configs_at_their_per_gpu_batches_vs_oracle(case):
    """BASELINE configs[2] / [3] at the batches one GPU sees, where the MoE capacity actually couples the samples: S2G 0.25b
    (L=128, 8 layers + 2 control copies) at 256 / 8 = 32 samples x 196 frames with a pre-encoded audio condition, and M2D
    (L=64, 4 layers + 3 copies, 35-d music features) at 128 sequences x 5 windows / 4 GPUs = 160 windows x 120 frames.
    One denoiser call through the control branch: teacher-forced oracle within 1e-3, routing flips of the free-running
    oracle reported and bounded (same two-part parity as the B=64 text-to-motion test)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    if case == 's2g_b32':
        dims, copy, feats, B, T, Tc = W.default_dims(NL=8), 2, 1536, 32, 196, 196
    else:
        dims, copy, feats, B, T, Tc = W.default_dims(L=64, F=256), 3, 35, 160, 120, 120
    sd = W.make_state_dict(dims, 0, shapes=W.control_param_shapes(dims, copy, feats))
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    g = torch.Generator().manual_seed(73)
    lengths = [int(v) for v in torch.randint(T // 2, T + 1, (B,), generator=g)]
    x, xf, mask = synth_inputs(dims, B, T, seed=74, lengths=lengths)
    c = torch.randn(B, Tc, feats, generator=g)
    ctx = nm.context(B, T, max_steps=1)
    ctx.enable_capture()
    ctx.set_timesteps([480])
    ctx.set_condition(xf.cuda(), mask.cuda())
    ctx.set_control(c.cuda())
    out2 = ctx.denoise(x.cuda(), 0)
    assert torch.equal(out2, ctx.denoise(x.cuda(), 0))                 # deterministic
    w = (1 - (1000 - 480) / 1000) * dims['scale'] + 1
    got = out2[:B] * w + out2[B:] * (1 - w)
    NL = dims['NL']
    forced = {slot: ctx.routing(slot) for slot in range(NL + copy)}
    torch.set_num_threads(min(32, os.cpu_count()))
    cap = {}
    ref = O.denoise_control(sd, dims, x, 480, xf, mask, c, copy, cap=cap, forced_routing=forced)
    err = maxabs(got, ref)
    tot_i = tot_k = 0
    for slot in range(NL + copy):
        a, b, dropped = _routing_flips(cap['routing'][slot]['free'], forced[slot])
        tot_i, tot_k = tot_i + a, tot_k + b
        print(f'  {case} slot {slot}: expert-id flips {a}, keep flips {b}, dropped {dropped}')
        assert a <= 8 and b <= 2, (slot, a, b)
    print(f'{case}: B={B} T={T} NL={NL}+{copy}: |hip - oracle (teacher-forced)| {err:.2e}; expert-id flips {tot_i}, keep flips '
          f'{tot_k} over {NL + copy} routings of {2 * 2 * B * T * dims["H"]} pairs')
    assert err <= TOL_FINAL
    ctx.close()
    nm.close()


@pytest.mark.parametrize('case', ['s2g_b32', 'm2d_160_windows'])
def test_baseline_control_configs_sampler_loop_lockstep(case):
    """BASELINE configs[2] / [3] as a LOOP at their per-GPU batches (tools/s2g_test.py:220, tools/m2d_test.py:139-232 drive
    `ddim_sample_loop` at these sizes): the first 8 steps of the 50-step DDIM schedule run on the device on the whole batch
    (S2G: 32 samples x 196 frames, 8 base layers + 2 control copies; M2D: 160 windows x 120 frames, 4 + 3 layers), and the CPU
    oracle walks beside it in lockstep on a 4-sample sub-batch -- from the HIP path's own x_t, with the same noise, the text
    K/V of the full condition batch, teacher-forced to the HIP path's routing decisions (with the discrete decisions forced
    every token is an independent row: the sub-batch reproduces the full-batch arithmetic).  Every x_{t-1} within 1e-3."""
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    if case == 's2g_b32':
        dims, copy, feats, B, T, Tc = W.default_dims(NL=8), 2, 1536, 32, 196, 196
    else:
        dims, copy, feats, B, T, Tc = W.default_dims(L=64, F=256), 3, 35, 160, 120, 120
    NL, H, NSTEP = dims['NL'], dims['H'], 8
    sd = W.make_state_dict(dims, 0, shapes=W.control_param_shapes(dims, copy, feats))
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    g = torch.Generator().manual_seed(83)
    lengths = [int(v) for v in torch.randint(T // 2, T + 1, (B,), generator=g)]
    x_T, xf, mask = synth_inputs(dims, B, T, seed=84, lengths=lengths)
    c = torch.randn(B, Tc, feats, generator=g)
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                             model_var_type='fixed_large', respace='15,15,8,6,6'))
    sched = O.Schedule(1000, '15,15,8,6,6')
    ctx = nm.context(B, T, max_steps=50)
    ctx.enable_capture()
    ctx.set_timesteps(d.timestep_map)
    ctx.set_condition(xf.cuda(), mask.cuda())
    ctx.set_control(c.cuda())
    torch.set_num_threads(min(32, os.cpu_count()))
    sub = torch.arange(0, B, B // 4)                              # 4 samples, mixed lengths
    tf_full = O.precompute_text_control(sd, xf, dims, copy)
    tf_sub = {k: t.view(2, B, *t.shape[1:])[:, sub].reshape(2 * len(sub), *t.shape[1:]) for k, t in tf_full.items()}
    gen = torch.Generator(device='cuda').manual_seed(85)
    x = x_T.cuda()
    worst = 0.0
    for i in range(49, 49 - NSTEP, -1):
        eps = torch.randn(B, T, dims['input_feats'], device='cuda', generator=gen)
        x_in = x.cpu()
        x = ctx.sample_step(x, i, d.step_coefs(i, 'ddim', dims['scale']), eps)
        forced = {slot: ctx.routing(slot) for slot in range(NL + copy)}
        fsub = {slot: tuple(v.view(2, B, T * H, 2)[:, sub].reshape(-1, 2) for v in f) for slot, f in forced.items()}
        x0 = O.denoise_control(sd, dims, x_in[sub], sched.timestep_map[i], xf[sub], mask[sub], c[sub], copy,
                               forced_routing=fsub, text_feats=tf_sub)
        ref = O.ddim_step(sched, i, x_in[sub], x0, eps.cpu()[sub])
        e = maxabs(x.cpu()[sub], ref)
        worst = max(worst, e)
        assert e <= TOL_FINAL, (case, i, e)
    assert bool(torch.isfinite(x).all())
    print(f'{case}: first {NSTEP} DDIM steps at B={B} T={T} NL={NL}+{copy}: lockstep |hip - oracle| worst {worst:.2e} '
          f'on {len(sub)} samples')
    ctx.close()
    nm.close()


@pytest.mark.parametrize('chain', [DEFAULT_CHAIN & ~7, DEFAULT_CHAIN])
def test_generic_fallback_path_vs_oracle(chain):
    """chain mask with bits 0-2 cleared: the generic path -- plain gemm_k launches + row kernels instead of the fused
    expert / SFFN MLP, the fused gate and the register-chained proj / q/k/v kernels; the library also takes it whenever a width
    is outside the fused kernels' range, so it keeps its own parity test: small-config denoiser call + 3 DDPM steps vs the CPU
    oracle, beside the default mask on the same inputs.  The mask is a per-context option (mc_ctx_set_option): both
    contexts live in this one process."""
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    dims, B, T = SMALL, 3, 24
    sd = W.make_state_dict(dims, SMALL_SEED)
    x_T, xf, mask = synth_inputs(dims, B, T, seed=31, lengths=[24, 19, 11])
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    ctx = nm.context(B, T, max_steps=1000)
    ctx.set_option('chain', chain)
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large'))
    ctx.set_timesteps(d.timestep_map)
    ctx.set_condition(xf.cuda(), mask.cuda())
    out2 = ctx.denoise(x_T.cuda(), 640)
    w = (1 - (1000 - 640) / 1000) * dims['scale'] + 1
    ref = O.denoise(sd, dims, x_T, 640, xf, mask)
    e1 = maxabs(out2[:B] * w + out2[B:] * (1 - w), ref)
    g = torch.Generator().manual_seed(1)
    noises = {i: torch.randn(B, T, dims['input_feats'], generator=g) for i in (999, 998, 997)}
    x = x_T.cuda()
    for i in (999, 998, 997):
        x = ctx.sample_step(x, i, d.step_coefs(i, 'ddpm', dims['scale']), noises[i].cuda())
    ref = O.sample_loop(sd, dims, O.Schedule(1000, None), 'ddpm', x_T, xf, mask, step_noise=lambda i: noises[i], num_steps=3)
    e2 = maxabs(x, ref)
    print(f'chain {chain}: errs {e1:.3e} {e2:.3e}')
    assert e1 <= TOL_STEP and e2 <= TOL_FINAL
    with pytest.raises(RuntimeError):
        ctx.set_option('no_such_switch', 1)
    ctx.close()
    nm.close()


def test_fused_proj_qkv_body_kernel_is_bit_identical_to_the_separate_kernels():
    """pqbody_k (round 4: combine + proj + body LayerNorm + q/k/v + static / dynamic body topology over frame-aligned tiles of 10
    frames, q/k/v exchanged through LDS, never in HBM) against projqkv_k + body_reg_k on the same context inputs: the mf rows it
    stores and the ys it writes must be the SAME BITS (same MFMA k order, same softmax / contraction arithmetic), the denoiser output
    too.  0.125b widths (L = 128, 12 parts) at a small batch pushed into the large-batch schedule by big_tokens = 0: sample groups
    of 72 frames = 7 full tiles + one of 2 frames (ragged), CFG twin aliasing in layer 0 on and (stop_after_layers) off, masked
    tails, and one whole-batch launch (split off: a tile that straddles nothing but ends ragged)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = FULL
    sd = W.make_state_dict(dims, 0)
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    B, T = 3, 24
    x, xf, mask = synth_inputs(dims, B, T, seed=5, lengths=[24, 20, 13])
    got = {}
    DFL = DEFAULT_CHAIN
    for tag, chain in (('fused', DFL), ('separate', DFL & ~(1 << 15)), ('fused_one_stream', DFL & ~(1 << 5)),
                       ('separate_one_stream', DFL & ~((1 << 15) | (1 << 5))), ('fused_no_twin_split', DFL & ~(1 << 16)),
                       ('mlp_reg_staged', DFL & ~(1 << 18))):
        ctx = nm.context(B, T, max_steps=2)
        ctx.set_option('big_tokens', 0)
        ctx.set_option('chain', chain)
        ctx.set_timesteps([800, 30])
        ctx.set_condition(xf.cuda(), mask.cuda())
        out = ctx.denoise(x.cuda(), 1).clone()
        ctx.denoise(x.cuda(), 0, stop_after_layers=1)            # layer 0 without twin aliasing: every row produced
        torch.cuda.synchronize()
        ys0, mf0 = ctx.buffer('ys').clone(), ctx.buffer('mf').clone()
        ctx.denoise(x.cuda(), 0, stop_after_layers=2)
        torch.cuda.synchronize()
        got[tag] = (out, ys0, mf0, ctx.buffer('ys').clone(), ctx.buffer('mf').clone(), ctx.buffer('h').clone())
        ctx.close()
    names = ('x0', 'ys layer 0', 'mf layer 0', 'ys layer 1', 'mf layer 1', 'h after 2 layers')
    for a, b, ks in (('fused', 'separate', range(6)), ('fused_one_stream', 'separate_one_stream', range(6)),
                     ('fused', 'fused_no_twin_split', range(6)),      # round 4: the twin layer's front as two sample sub-groups on two streams
                     ('fused', 'mlp_reg_staged', range(6)),           # mlp2d_k (LDS-DMA staged weight chunks, the default) vs mlp2_k
                     ('fused', 'fused_one_stream', (1, 2))):       # (later stages: the FiLM GEMM's tile width follows the launch's row count)
        for k in ks:
            assert bool(torch.isfinite(got[a][k]).all()), (a, names[k])
            assert torch.equal(got[a][k], got[b][k]), (a, b, names[k], float((got[a][k] - got[b][k]).abs().max()))
    nm.close()


def test_fp16_mfma_gemm_op_vs_fp64():
    """mc_half.hip gemm_h_k through the C-ABI: C = A W^T + bias + R with fp16 MFMA operands / fp32 accumulate.  The split form
    (x = hi + lo, three products) must be fp32-class, the single-rounding form fp16-class; ragged M (row guard), K = 32."""
    from motioncraft_amd import lib as L_
    lib = L_.load(require_gpu=True)
    _ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(9)
    for M, N, K in ((300, 128, 32), (1000, 384, 384), (12544, 1536, 1536)):
        a = torch.randn(M, K, generator=g)
        a[:, ::7] *= 1e-3                       # columns of small magnitude: the lo plane must not lose them
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b, r = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
        ref = a.double() @ w.double().T + b.double() + r.double()
        scale = float(ref.abs().max())
        ad, wd, bd, rd = a.cuda(), w.cuda(), b.cuda(), r.cuda()          # (kept alive: the ABI takes raw pointers)
        for split, tol in ((1, 4e-6), (0, 4e-3)):
            c = torch.full((M, N), float('nan'), device='cuda')
            L_.check(lib.mc_op_gemm_f16(_ptr(ad), _ptr(wd), _ptr(bd), _ptr(rd), _ptr(c), M, N, K, split, st))
            torch.cuda.synchronize()
            err = float((c.cpu().double() - ref).abs().max()) / scale
            print(f'fp16 MFMA gemm {M}x{N}x{K} split={split}: max rel err {err:.2e}')
            assert err <= tol, (M, N, K, split, err)
    c32 = torch.empty(M, N, device='cuda')          # the exact fp32 MFMA path on the same operands for comparison
    L_.check(lib.mc_op_gemm(_ptr(ad), _ptr(wd), _ptr(bd), _ptr(rd), _ptr(c32), M, N, K, K, 0, st))
    torch.cuda.synchronize()
    print(f'fp32 MFMA gemm {M}x{N}x{K}: max rel err {float((c32.cpu().double() - ref).abs().max()) / scale:.2e}')


@pytest.mark.parametrize('size', ['small', 'full_width'])
def test_mixed_text_audio_control_fp16_mfma_50_step_ddim(size):
    """BASELINE.json configs[4]: mixed text + audio plug-and-play control (text `xf_out` AND control condition `c` through
    ControlT2MHalf, controlnet.py:340-424), reduced-precision MFMA, 50-step DDIM.  Precision 'f16x3' (fp16 hi/lo split,
    fp32 accumulate; gate / routing / LayerNorm statistics stay fp32) walks the whole loop in lockstep with the fp32 CPU
    oracle -- teacher-forced to the HIP path's routing decisions, from the HIP path's own x_t: every x_{t-1} within
    1e-3, routing flips of the free-running oracle reported (all 50 steps at the small size, the first 30 at the 0.125b
    width; the replay test below runs all 50 there).  Plain 'f16' (one rounding to fp16 per operand, what
    mmcv's wrap_fp16_model does to the reference, tools/test.py:95-97) is walked over the first steps and held to the SAME 1e-3
    per sampler step (observed 1.1e-4: the update damps the x0 error; the x0 prediction itself is fp16-class, see
    test_fp16_mfma_large_batch_kernels_single_step_vs_oracle)."""
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    if size == 'small':
        dims, copy, feats, B, T, Tc = CTRL, CTRL_COPY, CTRL_FEATS, 2, 24, 20
        sd = W.make_state_dict(dims, SMALL_SEED, shapes=W.control_param_shapes(dims, copy, feats))
    else:       # the 0.125b architecture (L=128, 12 parts, 4 layers) + 2 control copies, pre-encoded audio of width D
        dims, copy, feats, B, T, Tc = FULL, 2, 1536, 2, 196, 196
        sd = W.make_state_dict(dims, 0, shapes=W.control_param_shapes(dims, copy, feats))
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    g = torch.Generator().manual_seed(91)
    x_T, xf, mask = synth_inputs(dims, B, T, seed=92, lengths=[T, T - 5] + [T - 40] * (B - 2))
    c = torch.randn(B, Tc, feats, generator=g)
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                             model_var_type='fixed_large', respace='15,15,8,6,6'))
    sched = O.Schedule(1000, '15,15,8,6,6')
    NL = dims['NL']
    torch.set_num_threads(min(32, os.cpu_count()))
    noises = step_noise_from_seed(93, (B, T, dims['input_feats']), 50)
    for prec, nsteps, tol in (('f16x3', 50 if size == 'small' else 30, TOL_FINAL), ('f16', 4, TOL_FINAL)):      # (f16: 1.1e-4 per step observed)
        ctx = nm.context(B, T, max_steps=50)
        ctx.set_precision(prec)
        ctx.enable_capture()
        ctx.set_timesteps(d.timestep_map)
        ctx.set_condition(xf.cuda(), mask.cuda())
        ctx.set_control(c.cuda())
        x = x_T.cuda()
        worst, fi, fk = 0.0, 0, 0
        for n, i in enumerate(range(49, 49 - nsteps, -1)):
            x_in = x.cpu()
            x = ctx.sample_step(x, i, d.step_coefs(i, 'ddim', dims['scale']), noises[n].cuda())
            forced = {slot: ctx.routing(slot) for slot in range(NL + copy)}
            cap = {}
            x0 = O.denoise_control(sd, dims, x_in, sched.timestep_map[i], xf, mask, c, copy, cap=cap, forced_routing=forced)
            ref = O.ddim_step(sched, i, x_in, x0, noises[n])
            worst = max(worst, maxabs(x, ref))
            for slot in range(NL + copy):
                a, b, _ = _routing_flips(cap['routing'][slot]['free'], forced[slot])
                fi, fk = fi + a, fk + b
        print(f'configs[4] {size} precision {prec}: {nsteps} DDIM steps in lockstep, worst |hip - fp32 oracle| {worst:.2e}; '
              f'routing flips vs the free-running oracle: expert-id {fi}, keep {fk} over {nsteps * (NL + copy)} routings')
        assert worst <= tol, (prec, worst)
        if prec == 'f16x3':
            assert fi + fk <= 4 * nsteps        # the gate is fp32 in every mode: flips stay at the fp32 path's level
        ctx.close()
    nm.close()


@pytest.mark.parametrize('prec', ['f16x3', 'f16'])
def test_fp16_fused_body_kernel_and_plane_gemm_equal_the_separate_kernels(prec):
    """Reduced-precision contexts, round 4: pqbody_h_k (projqkv_h_k + body-topology attention over frame-aligned tiles) must write the
    same mf / ys BITS as projqkv_h_k + body_reg_k (same MFMA order per row, same fp32 body arithmetic), and the FiLM GEMM fed from the
    fp16 planes film_rows_k writes (gemm_hd_k, LDS-DMA) must agree with the in-kernel split of gemm_h_k to fp32 round-off of the
    accumulation (same fp16 operand values, another k-tile grouping).  0.125b widths, B=3 x 24 frames pushed into the large-batch
    schedule (big_tokens = 0, half_min_rows = 0)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = FULL
    nm = NativeModel(dims, W.make_state_dict(dims, 0), cfg_scale=dims['scale'])
    B, T = 3, 24
    x, xf, mask = synth_inputs(dims, B, T, seed=5, lengths=[24, 20, 13])
    DFL = DEFAULT_CHAIN
    got = {}
    # (round 6: bit 27 = the plane GEMM prefetches its residual rows into registers, the same bits; bit 28 = its accumulators start as R + bias, plain f16 only)
    for tag, chain in (('new', DFL), ('no_fused_body', DFL & ~(1 << 15)), ('no_planes', DFL & ~(1 << 17)), ('mlp_reg_staged', DFL & ~(1 << 18)),
                       ('late_residual', DFL & ~(1 << 27)), ('acc_init', DFL | (1 << 28))):
        ctx = nm.context(B, T, max_steps=2)
        ctx.set_option('big_tokens', 0)
        ctx.set_option('half_min_rows', 0)
        ctx.set_option('chain', chain)
        ctx.set_precision(prec)
        ctx.set_timesteps([800, 30])
        ctx.set_condition(xf.cuda(), mask.cuda())
        out = ctx.denoise(x.cuda(), 1).clone()
        ctx.denoise(x.cuda(), 0, stop_after_layers=1)
        torch.cuda.synchronize()
        got[tag] = (out, ctx.buffer('ys').clone(), ctx.buffer('mf').clone(), ctx.buffer('h').clone())
        assert ctx.effective_precision == prec
        ctx.close()
    for k, name in enumerate(('x0', 'ys layer 0', 'mf layer 0', 'h after layer 0')):
        assert bool(torch.isfinite(got['new'][k]).all()), name
        assert torch.equal(got['new'][k], got['no_fused_body'][k]), (name, float((got['new'][k] - got['no_fused_body'][k]).abs().max()))
        # mlp2hd_k (LDS-DMA staged weight chunks) vs mlp2_h_k (register staged): the same MFMA order, the same bits
        assert torch.equal(got['new'][k], got['mlp_reg_staged'][k]), (name, float((got['new'][k] - got['mlp_reg_staged'][k]).abs().max()))
    assert torch.equal(got['new'][1], got['no_planes'][1]) and torch.equal(got['new'][2], got['no_planes'][2])
    e = maxabs(got['new'][3], got['no_planes'][3])
    e28 = maxabs(got['acc_init'][3], got['new'][3])
    print(f'{prec}: FiLM GEMM from fp16 planes vs in-kernel split: |dh| after layer 0 {e:.2e}, |dx0| {maxabs(got["new"][0], got["no_planes"][0]):.2e}; '
          f'accumulators started as R + bias (chain bit 28) vs (sum + bias) + R: |dh| {e28:.2e}, |dx0| {maxabs(got["acc_init"][0], got["new"][0]):.2e}')
    assert e <= 2e-5
    # residual rows prefetched at the top of the tile (bit 27) vs loaded in the epilogue: the same order, the same bits
    assert all(torch.equal(a, b) for a, b in zip(got['new'], got['late_residual']))
    if prec == 'f16x3':
        assert all(torch.equal(a, b) for a, b in zip(got['new'], got['acc_init']))      # the split mode ignores bit 28
    else:
        # another summation order of the same fp32 terms: the small products are added INTO the O(10..100) residual row one MFMA at a time (measured 2.6e-4 on h
        # after layer 0), against the 1e-2-level error one fp16 rounding per operand puts on the same rows
        assert 0 < e28 <= 1e-3
    nm.close()


@pytest.mark.parametrize('prec', ['f16x3', 'f16'])
def test_fragment_major_planes_and_register_fed_plane_gemm_are_bit_identical(prec):
    """chain bit 29 (round 6): film_rows_k writes the FiLM operand planes fragment-major -- per 32-row block and 16-wide k-step the 64 lanes' MFMA
    operands contiguous -- and gemm_hf_k reads its A fragments straight into registers (only W through LDS).  Same k order per output, the same
    three products per operand pair in the split mode: the SAME BITS as the row-major planes + gemm_hd_k.  0.125b widths, B = 16 x 196 frames in the
    two-stream schedule (sample groups of 3136 rows = 98 whole 32-row blocks) with ragged lengths; and B = 3 x 24 frames (72-row groups: no whole
    blocks, the launcher must fall back to the row-major path by itself)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = FULL
    nm = NativeModel(dims, W.make_state_dict(dims, 0), cfg_scale=dims['scale'])
    for B, T, lengths in ((16, 196, [196, 150, 64, 196, 100, 196, 77, 196, 196, 120, 196, 196, 90, 196, 196, 130]), (3, 24, [24, 20, 13])):
        x, xf, mask = synth_inputs(dims, B, T, seed=7, lengths=lengths)
        got = {}
        for tag, chain in (('frag_major', DEFAULT_CHAIN | (1 << 29)), ('row_major', DEFAULT_CHAIN & ~(1 << 29))):
            ctx = nm.context(B, T, max_steps=2)
            if B == 3:
                ctx.set_option('big_tokens', 0)
                ctx.set_option('half_min_rows', 0)
            ctx.set_option('chain', chain)
            ctx.set_precision(prec)
            ctx.set_timesteps([800, 30])
            ctx.set_condition(xf.cuda(), mask.cuda())
            out = ctx.denoise(x.cuda(), 1).clone()
            ctx.denoise(x.cuda(), 0, stop_after_layers=1)
            torch.cuda.synchronize()
            got[tag] = (out, ctx.buffer('h').clone())
            assert ctx.effective_precision == prec
            ctx.close()
        assert bool(torch.isfinite(got['frag_major'][0]).all())
        for k, name in enumerate(('x0', 'h after layer 0')):
            assert torch.equal(got['frag_major'][k], got['row_major'][k]), (prec, B, name, maxabs(got['frag_major'][k], got['row_major'][k]))
    nm.close()


@pytest.mark.parametrize('latent', [128, 64])
def test_fp16_temporal_attention_kernel_vs_the_fp32_kernel(latent):
    """temporal_h_k (round 4: the temporal linear attention's two contractions on the fp16 MFMA, K / V chunks transposed while staging,
    unnormalised exponentials with the 1 / sum applied in fp32) against temporal_k on the SAME mf / text rows of a reduced-precision
    context: y_t of base layer 0 with chain bit 20 on and off (every kernel before it is the same, so mf is bit-equal).  Ragged
    lengths (masked frames), the unconditional half (masked text rows), L = 128 and L = 64.  f16x3 must be fp32-class, plain f16
    within one fp16 rounding per operand."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = FULL if latent == 128 else W.default_dims(L=64, F=256, max_seq_len=24)
    nm = NativeModel(dims, W.make_state_dict(dims, 2), cfg_scale=dims['scale'])
    B, T = 3, 24
    x, xf, mask = synth_inputs(dims, B, T, seed=8, lengths=[24, 19, 9])
    for prec, tol in (('f16x3', 2e-5), ('f16', 4e-3)):
        got = {}
        for tag, chain in (('h', DEFAULT_CHAIN), ('f32', DEFAULT_CHAIN & ~(1 << 20))):
            ctx = nm.context(B, T, max_steps=1)
            ctx.set_option('big_tokens', 0)
            ctx.set_option('half_min_rows', 0)
            ctx.set_option('temporal_split', 0)
            ctx.set_option('chain', chain)
            ctx.set_precision(prec)
            ctx.set_timesteps([650])
            ctx.set_condition(xf.cuda(), mask.cuda())
            out = ctx.denoise(x.cuda(), 0).clone()
            ctx.denoise(x.cuda(), 0, stop_after_layers=1)
            torch.cuda.synchronize()
            got[tag] = (out, ctx.buffer('yt').clone(), ctx.buffer('mf').clone())
            ctx.close()
        assert torch.equal(got['h'][2], got['f32'][2])
        scale = float(got['f32'][1].abs().max())
        e = maxabs(got['h'][1], got['f32'][1])
        print(f'L = {latent} {prec}: |y_t fp16-MFMA - fp32-MFMA| {e:.2e} (|y_t| max {scale:.2f}), |x0| {maxabs(got["h"][0], got["f32"][0]):.2e}')
        assert bool(torch.isfinite(got['h'][1]).all()) and e <= tol * max(scale, 1.0)
    nm.close()


def test_fp16_mfma_large_batch_kernels_single_step_vs_oracle(full_model):
    """The large-batch kernel selection of the reduced-precision mode (N > 65 536 tokens: proj + body LayerNorm + q/k/v in
    one chained fp16-MFMA kernel, two sample groups on two streams, twin dedupe in layer 0): one denoiser call at B=16,
    196 frames, mixed lengths, against the fp32 oracle teacher-forced to the HIP path's routing.  'f16x3' meets the fp32 path's
    bound on the x0 PREDICTION; plain 'f16' does not (8e-3 observed: one fp16 rounding per operand, amplified by the CFG weights
    5.16 / -4.16 at t = 640) and is documented as outside the north-star tolerance on x0 (include/motioncraft_amd.h) -- what it
    does meet is 1e-3 on x_{t-1} of the sampler step, asserted here through the DDPM update of both predictions."""
    from oracle import stmogen_oracle as O
    sd, nm = full_model
    B, T = 16, 196
    g = torch.Generator().manual_seed(15)
    lengths = [int(v) for v in torch.randint(64, 197, (B,), generator=g)]
    x_T, xf, mask = synth_inputs(FULL, B, T, seed=34, lengths=lengths)
    torch.set_num_threads(min(32, os.cpu_count()))
    tf = O.precompute_text(sd, xf, FULL)
    w = (1 - (1000 - 640) / 1000) * FULL['scale'] + 1
    for prec, tol in (('f16x3', TOL_STEP), ('f16', 2e-2)):
        ctx = nm.context(B, T, max_steps=1)
        ctx.set_precision(prec)
        ctx.enable_capture()
        ctx.set_timesteps([640])
        ctx.set_condition(xf.cuda(), mask.cuda())
        out2 = ctx.denoise(x_T.cuda(), 0)
        assert torch.equal(out2, ctx.denoise(x_T.cuda(), 0))
        forced = [ctx.routing(i) for i in range(FULL['NL'])]
        ref = O.denoise(sd, FULL, x_T, 640, xf, mask, text_feats=tf, forced_routing=forced)
        err = maxabs(out2[:B] * w + out2[B:] * (1 - w), ref)
        sched = O.Schedule(1000, None)
        eps = torch.randn(x_T.shape, generator=torch.Generator().manual_seed(3))
        got0 = (out2[:B] * w + out2[B:] * (1 - w)).cpu()
        err_prev = maxabs(O.ddpm_step(sched, 640, x_T, got0, eps), O.ddpm_step(sched, 640, x_T, ref, eps))
        print(f'B=16 single step, precision {prec}: |hip - fp32 oracle (teacher-forced)| x0 {err:.2e}, x_(t-1) {err_prev:.2e}')
        assert err <= tol and err_prev <= TOL_FINAL, (prec, err, err_prev)
        ctx.close()


@pytest.mark.parametrize('prec', ['f16x3', 'f16'])
def test_configs4_as_benched_large_batch_graph_replay_and_lockstep(prec):
    """BASELINE.json configs[4] AT THE SHAPE bench.py REPORTS (`config.configs4`): 0.125b + 2 control copies, batch 32 x 196 frames
    (150 528 tokens > big_tokens: the large-batch TWO-STREAM schedule -- fork / join inside the capture, cooperative routing,
    per-group after_proj, fp16 planes into gemm_hd_k, gemm_tail_k with the CFG weights from the device table), width-D audio
    condition through ControlT2MHalf (controlnet.py:340-424), fp16 MFMA (tools/test.py:95-97), hipGraph replay.  Three arms over
    the first steps of the 50-step DDIM schedule, same inputs and noise:
      (a) eager mc_sample_step with the routing capture on: every step in lockstep with the fp32 CPU oracle on a 4-sample
          sub-batch, teacher-forced to the HIP path's routing, from the HIP path's own x_t: x_{t-1} within 1e-3;
      (b) mc_ctx_graph_capture + graph_step replays: the SAME BITS as (a);
      (c) eager again with the second sample group's stream held back 2 ms in front of every layer tail (the two groups far out of
          phase: ADVICE r04 -- the deferred last FiLM block's fp32 rows and the other group's fp16 planes must not share `a`):
          the SAME BITS as (a)."""
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    dims, copy, B, T, NSTEP = FULL, 2, 32, 196, 3
    feats = dims['L'] * dims['H']
    NL, H = dims['NL'], dims['H']
    sd = W.make_state_dict(dims, 0, shapes=W.control_param_shapes(dims, copy, feats))
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    g = torch.Generator().manual_seed(401)
    lengths = [int(v) for v in torch.randint(T // 2, T + 1, (B,), generator=g)]
    x_T, xf, mask = synth_inputs(dims, B, T, seed=402, lengths=lengths)
    c = torch.randn(B, T, feats, generator=g)
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                             model_var_type='fixed_large', respace='15,15,8,6,6'))
    sched = O.Schedule(1000, '15,15,8,6,6')
    S = d.num_timesteps
    coefs = [d.step_coefs(i, 'ddim', dims['scale']) for i in range(S)]
    noises = [n.cuda() for n in step_noise_from_seed(403, tuple(x_T.shape), NSTEP)]
    torch.set_num_threads(min(32, os.cpu_count()))
    sub = torch.arange(0, B, B // 4)
    tf_full = O.precompute_text_control(sd, xf, dims, copy)
    tf_sub = {k: t.view(2, B, *t.shape[1:])[:, sub].reshape(2 * len(sub), *t.shape[1:]) for k, t in tf_full.items()}
    outs, worst = {}, 0.0
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for arm in ('eager', 'graph', 'delayed'):
            ctx = nm.context(B, T, max_steps=S)
            ctx.set_precision(prec)
            if arm == 'eager':
                ctx.enable_capture()
            if arm == 'delayed':
                ctx.set_option('dbg_delay_us', 2000)
            ctx.set_timesteps(d.timestep_map)
            ctx.set_condition(xf.cuda(), mask.cuda())
            ctx.set_control(c.cuda())
            assert ctx.effective_precision == prec
            x, noise = x_T.cuda(), torch.empty(B, T, dims['input_feats'], device='cuda')
            if arm == 'graph':
                ctx.graph_capture(x, noise, coefs)
                x.copy_(x_T)
            traj = []
            for n, i in enumerate(range(S - 1, S - 1 - NSTEP, -1)):
                noise.copy_(noises[n])
                if arm == 'graph':
                    ctx.graph_step(i)
                else:
                    x_in = x.cpu() if arm == 'eager' else None
                    ctx.sample_step(x, i, coefs[i], noise, x_prev=x)
                stream.synchronize()
                traj.append(x.cpu())
                if arm == 'eager':
                    forced = {slot: ctx.routing(slot) for slot in range(NL + copy)}
                    fsub = {slot: tuple(v.view(2, B, T * H, 2)[:, sub].reshape(-1, 2) for v in f) for slot, f in forced.items()}
                    x0 = O.denoise_control(sd, dims, x_in[sub], sched.timestep_map[i], xf[sub], mask[sub], c[sub], copy,
                                           forced_routing=fsub, text_feats=tf_sub)
                    ref = O.ddim_step(sched, i, x_in[sub], x0, noises[n].cpu()[sub])
                    e = maxabs(traj[-1][sub], ref)
                    worst = max(worst, e)
                    assert e <= TOL_FINAL, (prec, i, e)
            outs[arm] = traj
            if arm == 'graph':
                ctx.graph_release()
            ctx.close()
    print(f'configs[4] as benched (B={B} x {T}, 4+{copy} layers, {prec}): {NSTEP} DDIM steps, lockstep |hip - fp32 oracle| worst {worst:.2e} '
          f'on {len(sub)} samples; graph replay and the delayed-stream run bit-identical to eager')
    for k in range(NSTEP):
        assert bool(torch.isfinite(outs['eager'][k]).all())
        assert torch.equal(outs['eager'][k], outs['graph'][k]), (prec, 'graph', k, maxabs(outs['eager'][k], outs['graph'][k]))
        assert torch.equal(outs['eager'][k], outs['delayed'][k]), (prec, 'delayed', k, maxabs(outs['eager'][k], outs['delayed'][k]))
    nm.close()


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_hipgraph_replay_of_the_sampler_step_is_bit_identical(prec):
    """BASELINE.json configs[4] "hipGraph-captured 50-step DDIM": ONE captured graph of mc_sample_step (step index in device
    memory, FiLM tables / sampler coefficients addressed inside the kernels) replayed for all 50 steps must reproduce the
    eager launch sequence bit for bit -- mixed text + audio control config, small size and the 0.125b width."""
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                             model_var_type='fixed_large', respace='15,15,8,6,6'))
    S = d.num_timesteps
    # (third case: B = 16 -- the large-batch two-stream schedule with its fork / join / ordering events inside the capture)
    for dims, copy, feats, B, T, seed in ((CTRL, CTRL_COPY, CTRL_FEATS, 2, 24, SMALL_SEED), (FULL, 2, 35, 3, 196, 0), (FULL, 2, 35, 16, 196, 0)):
        sd = W.make_state_dict(dims, seed, shapes=W.control_param_shapes(dims, copy, feats))
        nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
        x_T, xf, mask = synth_inputs(dims, B, T, seed=51, lengths=[T - 3] + [T] * (B - 1))
        c = torch.randn(B, T, feats, generator=torch.Generator().manual_seed(52))
        noises = [n.cuda() for n in step_noise_from_seed(53, tuple(x_T.shape), S)]
        coefs = [d.step_coefs(i, 'ddim', dims['scale']) for i in range(S)]
        outs = {}
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for arm in ('eager', 'graph'):
                ctx = nm.context(B, T, max_steps=S)
                ctx.set_precision(prec)
                ctx.set_timesteps(d.timestep_map)
                ctx.set_condition(xf.cuda(), mask.cuda())
                ctx.set_control(c.cuda())
                x, noise = x_T.cuda(), torch.empty(B, T, dims['input_feats'], device='cuda')
                if arm == 'graph':
                    ctx.graph_capture(x, noise, coefs)
                    x.copy_(x_T)                       # (the capture pass itself does not execute anything)
                for n, i in enumerate(range(S - 1, -1, -1)):
                    noise.copy_(noises[n])
                    if arm == 'graph':
                        ctx.graph_step(i)
                    else:
                        ctx.sample_step(x, i, coefs[i], noise, x_prev=x)
                stream.synchronize()
                outs[arm] = x.cpu()
                if arm == 'graph':
                    with pytest.raises(RuntimeError):
                        ctx.graph_step(S)
                    ctx.graph_release()
                    with pytest.raises(RuntimeError):
                        ctx.graph_step(0)
                ctx.close()
        assert bool(torch.isfinite(outs['eager']).all())
        assert torch.equal(outs['eager'], outs['graph']), (prec, dims['L'], maxabs(outs['eager'], outs['graph']))
        nm.close()


@pytest.mark.parametrize('arch', ['motionx_322', 'humanml3d_263'])
def test_one_pass_decoder_tail_equals_combine_plus_grouped_gemm(arch):
    """gemm_tail_k (round 4, chain bit 21: the folded decoder tail of the large-batch schedule with the CFG combination formed in its A
    staging and both K groups in one accumulator) against axpby_pair_k + the grouped gemm_small16_k + the sum in the sampler kernel:
    x0 and x_(t-1) of two DDIM steps within fp32 round-off of the other accumulation order (the operands are the same values).  0.125b
    widths at B=3 x 24 frames (72 rows: a ragged second row tile; N = 322: a ragged seventh column tile), pushed onto the kernel with
    small_gemm_rows = 0.  Under hipGraph replay the CFG weights come from the device table: replay must equal eager bit for bit.
    Second architecture: HumanML3D widths (263 features in 6 column tiles, 8 parts x 64, D = 512) at 25 frames -- 75 rows and
    B*T*C % 4 != 0, so the sampler update runs its element-wise form; there the two steps are also walked as ONE mc_sample_loop call
    (the sampler update of step 1 writes step 0's padded pose-encoder operand, 263 -> 288 columns) and must equal the per-step walk
    bit for bit."""
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = FULL if arch == 'motionx_322' else HML_FULL
    nm = NativeModel(dims, W.make_state_dict(dims, 4), cfg_scale=dims['scale'])
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large',
                             respace='2'))
    S = d.num_timesteps
    B, T = (3, 24) if arch == 'motionx_322' else (3, 25)
    x_T, xf, mask = synth_inputs(dims, B, T, seed=9, lengths=[T, 17, 12])
    coefs = [d.step_coefs(i, 'ddim', dims['scale']) for i in range(S)]
    noise = torch.randn(B, T, dims['input_feats'], generator=torch.Generator().manual_seed(3)).cuda()
    got = {}
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        # (round 5) the default one-pass form is gemm_tail2_k (block ranges, A read once, one partial product per K group added by the
        # sampler kernel; gemm_tune bit 10); 'column_tiles' = round 4's gemm_tail_k
        for arm, chain in (('one_pass', DEFAULT_CHAIN), ('grouped', DEFAULT_CHAIN & ~(1 << 21)), ('one_pass_graph', DEFAULT_CHAIN),
                           ('column_tiles', DEFAULT_CHAIN)):
            ctx = nm.context(B, T, max_steps=S)
            ctx.set_option('small_gemm_rows', 0)
            ctx.set_option('chain', chain)
            if arm == 'column_tiles':
                ctx.set_option('gemm_tune', 817)
            ctx.set_timesteps(d.timestep_map)
            ctx.set_condition(xf.cuda(), mask.cuda())
            x = x_T.cuda().clone()
            x0 = torch.empty_like(x)
            if arm.endswith('graph'):
                ctx.graph_capture(x, noise, coefs)
                x.copy_(x_T)
                for i in range(S - 1, -1, -1):
                    ctx.graph_step(i)
            else:
                for i in range(S - 1, -1, -1):
                    ctx.sample_step(x, i, coefs[i], noise, x_prev=x, x0=x0)
            stream.synchronize()
            got[arm] = (x.clone(), x0.clone())
            ctx.close()
    e_x, e_x0 = maxabs(got['one_pass'][0], got['grouped'][0]), maxabs(got['one_pass'][1], got['grouped'][1])
    print(f'one-pass tail vs combine + grouped GEMM: |dx_prev| {e_x:.2e}, |dx0| {e_x0:.2e} (|x0| max {float(got["grouped"][1].abs().max()):.2f})')
    assert bool(torch.isfinite(got["one_pass"][0]).all()) and e_x <= 6e-5 and e_x0 <= 6e-5
    e_c = maxabs(got['column_tiles'][1], got['grouped'][1])
    print(f'column-tile one-pass tail vs combine + grouped GEMM: |dx0| {e_c:.2e}')
    assert e_c <= 6e-5 and maxabs(got['column_tiles'][0], got['grouped'][0]) <= 6e-5
    assert torch.equal(got['one_pass'][0], got['one_pass_graph'][0])
    if arch != 'motionx_322':
        ctx = nm.context(B, T, max_steps=S)
        ctx.set_option('small_gemm_rows', 0)
        ctx.set_timesteps(d.timestep_map)
        ctx.set_condition(xf.cuda(), mask.cuda())
        x = x_T.cuda().clone()
        order = list(range(S - 1, -1, -1))
        ctx.sample_loop(x, order, [coefs[i] for i in order], noise=torch.stack([noise] * S))
        torch.cuda.synchronize()
        assert torch.equal(x, got['one_pass'][0]), maxabs(x, got['one_pass'][0])
        ctx.close()
    nm.close()


def test_wrap_fp16_model_and_graph_replay_through_the_reference_api(small_model):
    """The reference tools' fp16 hook (`if cfg.get('fp16') is not None: wrap_fp16_model(model)`, tools/test.py:95-97) and the
    replay option, through configs -> build_architecture -> MotionDiffusion.forward: fp16-split + hipGraph replay must
    reproduce the plain fp32 eager result of the same call within the north-star tolerance, replay itself bit for bit."""
    import motioncraft_amd as mc
    sd, _ = small_model
    g = load('small_ddim.npz')
    B, T = 2, 24
    noises = step_noise_from_seed(int(g['noise_seed']), (B, T, 322), 50)
    mask = T_(g['motion_mask'])
    call = lambda arch, **inf: torch.stack([r['pred_motion'] for r in arch(
        motion=torch.zeros(B, T, 322), motion_mask=mask, motion_length=mask.sum(1, keepdim=True).long(),
        motion_metas=[{'text': 'a'}, {'text': 'b'}], xf_out=T_(g['xf_out']),
        inference_kwargs=dict(noise=T_(g['x_T']), step_noise=lambda i: noises[49 - i], **inf))])
    cfg, arch = _arch_small(sd)
    ref32 = call(arch)
    assert maxabs(ref32, T_(g['final'])) <= TOL_FINAL
    assert torch.equal(call(arch, graph=True), ref32)                      # replay of the fp32 step: bit-identical
    assert mc.wrap_fp16_model(arch) is arch and arch.model.precision == 'f16x3'
    h3 = call(arch)
    assert torch.equal(call(arch, graph=True), h3)
    e3 = maxabs(h3, T_(g['final']))
    mc.wrap_fp16_model(arch, split=False)
    e1 = maxabs(call(arch), T_(g['final']))
    print(f'wrap_fp16_model through the API, 50-step DDIM final vs reference golden: f16x3 {e3:.2e}, f16 {e1:.2e}')
    assert e3 <= TOL_FINAL
    with pytest.raises(TypeError):
        mc.wrap_fp16_model(object())
    arch.model.release()


def test_control_branch_vs_reference_golden():
    """ControlT2MHalf (a15; BASELINE configs 3-5 form): copied DecoderLayers + zero-init projections + condition
    padding/CFG masking, through the reference-style wrapper API and through the raw context."""
    import motioncraft_amd as mc
    from oracle import weights as W
    g = load('control_small.npz')
    sd = W.make_state_dict(CTRL, SMALL_SEED, shapes=W.control_param_shapes(CTRL, CTRL_COPY, CTRL_FEATS))
    cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
    cfg.model.model.num_layers = 3
    cfg.merge_from_dict({'condition_encode_cfg': dict(dataset_name='nothing', condition_pre_encode=False,
                                                      condition_pre_encode_type='nothing', control_cond_feats=CTRL_FEATS,
                                                      condition_latent_dim=CTRL['L'] * CTRL['H'], condition_cfg=True)})
    arch = mc.build_architecture(cfg.model)
    arch.model = mc.ControlT2MHalf(arch.model, copy_blocks_num=CTRL_COPY, control_cond_feats=CTRL_FEATS, cfg=cfg)
    arch.load_state_dict({'model.' + k: v for k, v in sd.items()})
    x_t, xf, mask, c = (T_(g[k]) for k in ('x_t', 'xf_out', 'motion_mask', 'c'))
    for t in (640, 3):
        ts = torch.full((2,), t)
        x0 = arch.model(x_t, ts, motion_mask=mask, xf_out=xf, c=c)
        assert maxabs(x0, T_(g[f'x0_t{t}'])) <= TOL_STEP, t
        x0n = arch.model(x_t, ts, motion_mask=mask, xf_out=xf, c=None)        # forward_test with c=None
        assert maxabs(x0n, T_(g[f'x0_noc_t{t}'])) <= TOL_STEP, t
    with pytest.raises(ValueError):                                           # condition of the wrong width
        arch.model(x_t, ts, motion_mask=mask, xf_out=xf, c=torch.zeros(2, 100, 2))
    arch.model.release()


def test_skeleton_part_layouts_vs_reference_golden():
    """the 8-part human_ml3d (263-d) / kit_ml (251-d) layouts -- odd channel counts (padded
    to a multiple of 4 inside the library), H=8 body kernel -- single calls, the 50-step DDIM loop, and the shipped
    T2M_humanml3d.py architecture (L=64, H=8, F=256) at T=196 with a padded sample."""
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    g = load('skeleton_parts.npz')
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                             model_var_type='fixed_large', respace='15,15,8,6,6'))
    for tag, dims in (('hml', HML_SMALL), ('kit', KIT_SMALL)):
        sd = W.make_state_dict(dims, SMALL_SEED)
        nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
        x, xf, mask = T_(g[f'{tag}_x_t']), T_(g[f'{tag}_xf_out']), T_(g[f'{tag}_motion_mask'])
        B, T, C = x.shape
        ctx = nm.context(B, T, max_steps=50)
        ctx.set_timesteps([901, 12])
        ctx.set_condition(xf.cuda(), mask.cuda())
        for s_, t in ((0, 901), (1, 12)):
            out2 = ctx.denoise(x.cuda(), s_)
            w = (1 - (1000 - t) / 1000) * dims['scale'] + 1
            assert maxabs(out2[:B] * w + out2[B:] * (1 - w), T_(g[f'{tag}_x0_t{t}'])) <= TOL_STEP, (tag, t)
        ctx.set_timesteps(d.timestep_map)
        noises = step_noise_from_seed(8, (B, T, C), 50)
        xc = x.cuda()
        for n, i in enumerate(range(49, -1, -1)):
            xc = ctx.sample_step(xc, i, d.step_coefs(i, 'ddim', dims['scale']), noises[n].cuda())
        assert maxabs(xc, T_(g[f'{tag}_ddim_final'])) <= TOL_FINAL, tag
        ctx.close()
        nm.close()
    sd = W.make_state_dict(HML_FULL, 0)
    nm = NativeModel(HML_FULL, sd, cfg_scale=HML_FULL['scale'])
    x, xf, mask = synth_inputs(HML_FULL, 1, 196, seed=32, lengths=[163])
    ctx = nm.context(1, 196, max_steps=1)
    ctx.set_timesteps([500])
    ctx.set_condition(xf.cuda(), mask.cuda())
    out2 = ctx.denoise(x.cuda(), 0)
    w = (1 - (1000 - 500) / 1000) * HML_FULL['scale'] + 1
    assert maxabs(out2[:1] * w + out2[1:] * (1 - w), T_(g['hmlfull_x0_t500_len163'])) <= TOL_STEP
    ctx.close()
    nm.close()


def test_repaint_outpainting_mode_vs_reference_golden(small_model):
    """y = {gt, outpainting_mask} (first 6 frames kept) through ddim_sample_loop: the resampling
    harmonize loop (jump 3 x 5: 138 denoiser calls + 108 undo steps), no_resample, and no_repaint (plain 50 steps
    with the per-step gt blend).  Draw order of the reference: DDIM noise, gt re-noising, one per undo."""
    import types
    from motioncraft_amd.diffusion import build_diffusion
    sd, nm = small_model
    g = load('repaint_small.npz')
    x_T, xf, mask = T_(g['x_T']), T_(g['xf_out']), T_(g['motion_mask'])
    gt, keep, ov = T_(g['gt']), T_(g['keep']), int(g['overlap_len'])

    class M:
        cfg_scale = SMALL['scale']

        def sampling_context(self, B, T, tmap, kw, device=None):
            self.ctx = nm.context(B, T, max_steps=len(tmap))
            self.ctx.set_timesteps(tmap)
            self.ctx.set_condition(kw['xf_out'].cuda(), kw['motion_mask'].cuda())
            return self.ctx
    for tag, over in (('resample', {}), ('noresample', dict(no_resample=True)), ('norepaint', dict(no_repaint=True))):
        opt = types.SimpleNamespace(same_overlap_noisy=False, no_repaint=False, addBlend=True, overlap_len=ov,
                                    no_resample=False, jump_length=3, jump_n_sample=5, timestep_respacing='ddim50')
        opt.__dict__.update(over)
        d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                                 model_var_type='fixed_large', respace='15,15,8,6,6'), opt=opt)
        gen = torch.Generator().manual_seed(int(g['noise_seed']))
        draws = (torch.randn(x_T.shape, generator=gen) for _ in range(10 ** 6))
        m = M()
        out = d.ddim_sample_loop(m, tuple(x_T.shape), noise=x_T, clip_denoised=False, eta=0, step_noise=draws,
                                 model_kwargs=dict(xf_out=xf, motion_mask=mask, y=dict(gt=gt, outpainting_mask=keep)))
        err = maxabs(out, T_(g[f'final_{tag}']))
        print(f'repaint {tag}: |hip - reference| {err:.2e}')
        assert err <= TOL_FINAL, tag
        assert maxabs(out[:, 0], gt[:, 0]) <= 1e-6
        m.ctx.close()


def test_sample_tool_writes_the_amass_style_npz(tmp_path):
    """tools/sample.py (the counterpart of the reference's tools/visualize.py:170-260 flow: config + checkpoint -> sampled motion -> SMPL-X
    .npz) end to end in a child process on the small config with deterministic random-init weights: two prompts of different lengths in
    one call, then the fp16 split mode with hipGraph replay; the file holds the AMASS-style arrays of the concatenated valid frames."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    base = [sys.executable, os.path.join(root, 'tools', 'sample.py'), os.path.join(HERE, 'configs', 'stmogen_small.py'), 'synthetic:3',
            '--random-condition', '5', '--out', str(tmp_path)]
    r = subprocess.run(base + ['--text', 'a person walks', 'a dancer spins', '--motion_length', '24', '18'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    path = os.path.join(str(tmp_path), 'res_a_person_walks_24.npz')
    z = np.load(path)
    assert z['poses'].shape[0] == 24 + 18 and z['poses'].shape[1] == 165 and z['trans'].shape == (42, 3)
    assert all(np.isfinite(z[k]).all() for k in z.files if z[k].dtype.kind == 'f')
    r = subprocess.run(base + ['--text', 'a person walks', '--motion_length', '24', '--fp16', 'split', '--graph'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.load(path)['poses'].shape == (24, 165)


def test_smplx_postprocessing_vs_scipy_restatement(tmp_path):
    """de-normalise + 322 -> poses/expressions/trans + scipy gaussian_filter(mode='nearest') on the
    device, against oracle/postprocess_oracle.py (the reference tools' own numpy/scipy lines), for float32 stats files
    (motionx) and float64 ones (beats2 / finedance), mixed lengths (filter support clamps at each sample's end)."""
    from motioncraft_amd import postprocess as P
    from oracle import postprocess_oracle as PO
    g = torch.Generator().manual_seed(3)
    B, T = 3, 196
    pred = torch.randn(B, T, 322, generator=g)
    lens = [196, 64, 17]                       # 17 < filter radius (14): heavy edge replication
    for dt in (np.float32, np.float64):
        mean = torch.randn(322, generator=g).numpy().astype(dt)
        std = (0.5 + torch.rand(322, generator=g)).numpy().astype(dt)
        for sig, fn in ((P.SIGMAS_T2M, PO.t2m_result), (P.SIGMAS_S2G, PO.s2g_result)):
            post = P.postprocess_smplx(pred.cuda(), lens, mean, std, sig)
            for b, n in enumerate(lens):
                pose, expr, trans = fn(pred[b, :n].numpy() * std + mean)
                tol = 2e-6 if dt == np.float32 else 1e-9       # float32 stats: scipy rounds expr/trans to float32
                assert np.abs(post['poses'][b, :n].cpu().numpy() - pose).max() <= 1e-9
                assert np.abs(post['expressions'][b, :n].cpu().numpy() - expr).max() <= tol
                assert np.abs(post['trans'][b, :n].cpu().numpy() - trans).max() <= tol
                assert float(post['poses'][b, n:].abs().sum()) == 0.0
            # several intervals in one file (tools/visualize.py:216-246): concatenate the valid frames FIRST, then filter
            # the stitched sequence -- the smoothing crosses the seams, only the two outer ends replicate their edge
            st = P.postprocess_smplx_stitched(pred.cuda(), lens, mean, std, sig)
            pose, expr, trans = fn(PO.denormalise([pred[b, :n].numpy() for b, n in enumerate(lens)], mean, std))
            assert st['poses'].shape == (sum(lens), 165)
            assert np.abs(st['poses'].cpu().numpy() - pose).max() <= 1e-9
            assert np.abs(st['expressions'].cpu().numpy() - expr).max() <= tol
            assert np.abs(st['trans'].cpu().numpy() - trans).max() <= tol
            if sig is P.SIGMAS_T2M:          # ... which differs from filtering each interval on its own near a seam
                assert np.abs(st['poses'][lens[0] - 1].cpu().numpy() - post['poses'][0, lens[0] - 1].cpu().numpy()).max() > 1e-3
    path = P.save_smplx_npz(str(tmp_path), 'a person walks. fast/slow', pred.cuda(), lens, mean, std)
    z = np.load(path)
    assert z['poses'].shape == (sum(lens), 165)
    pose, expr, trans = PO.t2m_result(PO.denormalise([pred[b, :n].numpy() for b, n in enumerate(lens)], mean, std))
    assert np.abs(z['poses'] - pose).max() <= 1e-9 and np.abs(z['trans'] - trans).max() <= 1e-9
    path = P.save_smplx_npz(str(tmp_path), 'a person walks. fast/slow', pred[:1].cuda(), [120], mean, std)
    assert os.path.basename(path) == 'res_a_person_walks_fast_slow_120.npz'
    z = np.load(path)
    assert z['poses'].shape == (120, 165) and z['expressions'].shape == (120, 100) and z['trans'].shape == (120, 3)
    assert z['betas'].shape == (300,) and str(z['model']) == 'smplx2020' and int(z['mocap_frame_rate']) == 30


def test_wav_encoder_vs_reference_golden_and_oracle():
    """WavEncoder as implicit-GEMM convolutions (channels-last, padding in the buffer, BN folded,
    LeakyReLU / residual in the GEMM epilogue) vs the reference class (golden) and vs the CPU oracle on a longer clip
    (ragged tile counts, the unaligned first layer Cin=2 and Cin=1, both shortcut kinds)."""
    from motioncraft_amd.wav_encoder import NativeWavEncoder
    from oracle import wav_encoder_oracle as WO
    from oracle import weights as W
    g = load('wav_encoder.npz')
    sd = W.make_wav_encoder_state(64, 2, seed=int(g['seed']))
    enc = NativeWavEncoder(64, 2, {'pre_encoder.feat_extractor.' + k: v for k, v in sd.items()},
                           prefix='pre_encoder.feat_extractor.')
    out = enc(T_(g['wav']).cuda())
    assert tuple(out.shape) == tuple(g['out'].shape)
    assert maxabs(out, T_(g['out'])) <= 2e-5
    enc.close()
    torch.set_num_threads(min(32, os.cpu_count()))
    for dim, cin, B, S in ((256, 2, 3, 20000), (128, 1, 2, 9001)):
        sd = W.make_wav_encoder_state(dim, cin, seed=7)
        enc = NativeWavEncoder(dim, cin, sd)
        gen = torch.Generator().manual_seed(S)
        wav = torch.randn(B, S, cin, generator=gen)
        ref = WO.wav_encoder(sd, wav if cin > 1 else wav[..., 0])
        out = enc(wav.cuda() if cin > 1 else wav[..., 0].cuda())
        assert tuple(out.shape) == tuple(ref.shape) and out.shape[1] == enc.out_len(S)
        err = maxabs(out, ref)
        print(f'wav encoder dim={dim} cin={cin}: out {tuple(out.shape)}, |hip - oracle| {err:.2e} (|ref| max {float(ref.abs().max()):.2f})')
        assert err <= 5e-5
        enc.close()


def test_speech_to_gesture_control_form_vs_reference_golden():
    """S2G form of the control branch (configs/stmogen/S2G_*: condition_pre_encode=True, type 'wav', dataset beats2):
    raw audio [B, 9000, 2] -> device WavEncoder (17 frames x D) -> control_cond_input -> copied blocks, through the
    reference-style wrapper API, weights taken from one reference-keyed state dict."""
    import motioncraft_amd as mc
    from oracle import weights as W
    g = load('control_wav_small.npz')
    sd = W.make_control_wav_state(CTRL, CTRL_COPY, 2, SMALL_SEED)
    cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
    cfg.model.model.num_layers = 3
    cfg.merge_from_dict({'condition_encode_cfg': dict(dataset_name='beats2', condition_pre_encode=True,
                                                      condition_pre_encode_type='wav', control_cond_feats=2,
                                                      condition_latent_dim=CTRL['L'] * CTRL['H'], condition_cfg=True)})
    arch = mc.build_architecture(cfg.model)
    arch.model = mc.ControlT2MHalf(arch.model, copy_blocks_num=CTRL_COPY, control_cond_feats=2, cfg=cfg)
    arch.load_state_dict({'model.' + k: v for k, v in sd.items()})
    x_t, xf, mask, audio = (T_(g[k]) for k in ('x_t', 'xf_out', 'motion_mask', 'audio'))
    x0 = arch.model(x_t, torch.full((2,), 420), motion_mask=mask, xf_out=xf, c=audio)
    err = maxabs(x0, T_(g['x0_t420']))
    print(f'S2G control form: |hip - reference| {err:.2e}')
    assert err <= TOL_STEP
    assert arch.model.base_model.wav_encoder.out_len(9000) == 17
    arch.model.release()


def test_text_encoder_vs_reference_golden_and_clip_tower_vs_oracle():
    """SURVEY.md 8f.2.  Stage A (text_pre_proj -> 2-layer nn.TransformerEncoder -> text_ln) vs the reference's own
    encode_text(clip_feat=...) golden; stage B (CLIP text transformer from token ids: causal pre-LN blocks, QuickGELU)
    vs the restated architecture in oracle/text_encoder_oracle.py (the `clip` package is un-vendored: parity unpinned),
    with a reduced tower (3 layers, 1000-token vocabulary) of the true width / head count."""
    from motioncraft_amd.text_encoder import NativeTextEncoder
    from oracle import text_encoder_oracle as TO
    from oracle import weights as W
    g = load('text_encoder.npz')
    cfg = dict(pretrained_model='clip', latent_dim=256, num_layers=2, ff_size=2048, dropout=0, use_text_proj=False)
    sd = W.make_text_encoder_state(W.text_encoder_param_shapes(256, 2, 2048), seed=int(g['seed']))
    enc = NativeTextEncoder(cfg, sd)
    out = enc.encode_feat(T_(g['clip_feat']).cuda())
    err = maxabs(out, T_(g['xf_out']))
    print(f'text encoder stage A: |hip - reference| {err:.2e}')
    assert err <= 5e-5
    with pytest.raises(RuntimeError):           # no clip.* weights loaded -> loud failure
        enc.encode_tokens(torch.zeros(1, 77, dtype=torch.int32).cuda())
    enc.close()
    shapes = W.text_encoder_param_shapes(256, 2, 2048, clip_layers=3, vocab=1000)
    sd = W.make_text_encoder_state(shapes, seed=9)
    enc = NativeTextEncoder(cfg, sd, clip=dict(layers=3))
    gen = torch.Generator().manual_seed(2)
    tokens = torch.randint(0, 1000, (4, 77), generator=gen)
    xf, feat = enc.encode_tokens(tokens.cuda(), return_clip_feat=True)
    feat_ref = TO.clip_text_features(sd, tokens, 3)
    xf_ref = TO.finetune_encoder(sd, feat_ref, 2)
    e1, e2 = maxabs(feat, feat_ref), maxabs(xf, xf_ref)
    print(f'text encoder stage B (CLIP tower): |hip - oracle| features {e1:.2e}, xf_out {e2:.2e}')
    assert e1 <= 1e-4 and e2 <= 1e-4
    # the tower against an independent implementation: features of transformers.CLIPTextModel (clip_tower_hf.npz)
    gh = load('clip_tower_hf.npz')
    shapes_h = W.text_encoder_param_shapes(256, 2, 2048, clip_width=int(gh['width']), clip_layers=int(gh['layers']),
                                           clip_ff=int(gh['ff']), vocab=int(gh['vocab']))
    ench = NativeTextEncoder(cfg, W.make_text_encoder_state(shapes_h, seed=int(gh['seed'])),
                             clip=dict(width=int(gh['width']), layers=int(gh['layers']), heads=int(gh['heads']), ff=int(gh['ff'])))
    _, feat_h = ench.encode_tokens(torch.from_numpy(gh['tokens']).cuda(), return_clip_feat=True)
    e3 = maxabs(feat_h, T_(gh['feat']))
    print(f'text encoder stage B (CLIP tower): |hip - transformers.CLIPTextModel| {e3:.2e}')
    assert e3 <= 1e-4
    ench.close()
    # causality: changing a later token must not change earlier positions of the CLIP features
    tokens2 = tokens.clone()
    tokens2[:, 40:] = (tokens2[:, 40:] + 1) % 1000
    _, feat2 = enc.encode_tokens(tokens2.cuda(), return_clip_feat=True)
    assert torch.equal(feat[:, :40], feat2[:, :40]) and not torch.equal(feat[:, 40:], feat2[:, 40:])
    enc.close()


def test_text_to_motion_call_with_clip_features_through_the_reference_api():
    """get_precompute_condition(clip_feat=...) (stmogen.py:676-688 -> encode_text) feeding the sampler: the whole
    clip_feat -> xf_out -> 50-step DDIM call stays on the device; xf_out is checked against the oracle's stage A and
    the sample against the oracle driven with that xf_out."""
    import motioncraft_amd as mc
    from oracle import stmogen_oracle as O, text_encoder_oracle as TO, weights as W
    dims = W.default_dims(max_seq_len=24, L=32, NL=2, F=64, Te=64, Dt=256, Nt=77)
    cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
    cfg.model.model.ca_block_cfg.text_latent_dim = 256
    cfg.model.model.ca_block_cfg.max_text_seq_len = 77
    cfg.model.model.text_encoder.latent_dim = 256
    arch = mc.build_architecture(cfg.model)
    sd = W.make_state_dict(dims, SMALL_SEED)
    tsd = W.make_text_encoder_state(W.text_encoder_param_shapes(256, cfg.model.model.text_encoder.num_layers,
                                                                cfg.model.model.text_encoder.ff_size), seed=5)
    arch.load_state_dict({'model.' + k: v for k, v in {**sd, **tsd}.items()})
    g = torch.Generator().manual_seed(77)
    B, T = 2, 24
    feat = torch.randn(B, 77, 512, generator=g)
    xf = arch.model.get_precompute_condition(clip_feat=feat)['xf_out']
    xf_ref = TO.finetune_encoder(tsd, feat, cfg.model.model.text_encoder.num_layers)
    assert maxabs(xf, xf_ref) <= 5e-5
    x_T = torch.randn(B, T, 322, generator=g)
    mask = torch.ones(B, T)
    mask[1, 20:] = 0
    noises = step_noise_from_seed(3, (B, T, 322), 50)
    res = arch(motion=torch.zeros(B, T, 322), motion_mask=mask, motion_length=mask.sum(1, keepdim=True).long(),
               motion_metas=[{'text': 'a'}, {'text': 'b'}], clip_feat=feat,
               inference_kwargs=dict(noise=x_T, step_noise=lambda i: noises[49 - i]))
    final = torch.stack([r['pred_motion'] for r in res])
    ref = O.sample_loop(sd, dims, O.Schedule(1000, '15,15,8,6,6'), 'ddim', x_T, xf_ref, mask,
                        step_noise=lambda i: noises[49 - i])
    err = maxabs(final, ref)
    print(f'clip_feat -> motion, 50-step DDIM: |hip - oracle| {err:.2e}')
    assert err <= TOL_FINAL
    arch.model.release()


def test_small_batch_expert_mlp_split_choices_agree(full_model, monkeypatch):
    """Small batches split the hidden dimension of the fused MLPs over workgroups (partial FC2 sums; fixed-order reduce for the
    experts, folded into the FiLM row kernel for the SFFN); the number of ways comes from a load model (experts: 3 at B = 2 x 196
    frames -- uneven 5 / 5 / 6 chunk shares -- 4 elsewhere; SFFN: 4 / 2 / none up to 8192 residual rows).  The alternatives must
    agree to fp32 round-off on the residual stream after ONE full-size decoder layer (beyond it a 1e-6 difference may move a
    near-tie gate decision of the next layer, which is not what is tested here)."""
    sd, nm = full_model
    T = 196

    def one_layer(B, seed, lengths, var, values):
        x, xf, mask = synth_inputs(FULL, B, T, seed=seed, lengths=lengths)
        got = {}
        for v in values:
            monkeypatch.setenv(var, v)                         # read when the context is created; 0 = the model
            ctx = nm.context(B, T, max_steps=1)
            ctx.enable_capture()
            ctx.set_timesteps([400])
            ctx.set_condition(xf.cuda(), mask.cuda())
            ctx.denoise(x.cuda(), 0, stop_after_layers=1)
            torch.cuda.synchronize()
            got[v] = (ctx.buffer('h').clone(), ctx.routing(0))
            ctx.close()
        monkeypatch.delenv(var)
        for v in values[1:]:
            assert torch.equal(got[values[0]][1][0], got[v][1][0]) and torch.equal(got[values[0]][1][1], got[v][1][1])
            err = maxabs(got[values[0]][0], got[v][0])
            print(f'{var} at B={B}: model choice vs {v} ways: {err:.2e}')
            assert 0 < err <= 2e-5
    one_layer(2, 31, [196, 150], 'MC_SPLIT_EXPERT', ('0', '4', '2'))
    one_layer(8, 32, [196, 150, 196, 64, 196, 196, 100, 196], 'MC_SPLIT_SFFN', ('0', '1', '2'))


def test_fp16_modes_use_the_fp32_kernels_at_tiny_batches(monkeypatch):
    """Default policy (MC_HALF_MIN_ROWS unset = 512 residual rows): a B=1-sized context in the f16x3 mode runs the fp32
    small-batch kernels -- bit-identical to the f32 mode -- and the fp16 kernels once the limit is lifted."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = W.default_dims(max_seq_len=24, L=32, NL=2, F=64, Te=64, Dt=32, Nt=8)
    nm = NativeModel(dims, W.make_state_dict(dims, SMALL_SEED), cfg_scale=dims['scale'])
    B, T = 2, 24
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, T, 322, generator=g).cuda()
    xf = torch.nn.functional.layer_norm(torch.randn(B, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],)).cuda()

    def run(prec):
        ctx = nm.context(B, T, max_steps=50)
        ctx.set_precision(prec)
        ctx.set_timesteps(list(range(0, 1000, 20)))
        ctx.set_condition(xf, torch.ones(B, T).cuda())
        out = ctx.denoise(x, 30).clone()
        ctx.close()
        return out
    monkeypatch.delenv('MC_HALF_MIN_ROWS', raising=False)
    ref, same = run('f32'), run('f16x3')
    assert torch.equal(ref, same)
    monkeypatch.setenv('MC_HALF_MIN_ROWS', '0')
    other = run('f16x3')
    assert not torch.equal(ref, other) and maxabs(ref, other) < 1e-3
    nm.close()


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_two_stream_schedule_is_race_free_under_stream_skew(full_model, prec):
    """The large-batch schedule runs the two sample groups on two HIP streams that only meet at the routing step of a layer and in
    front of the decoder tail; in a normal run they stay within half a kernel of each other, which can hide a missing dependency.
    `dbg_delay_us` holds one group's stream for 1.5 ms in front of every layer tail (either group: > 0 the second, < 0 the first), so
    each group in turn runs a whole FiLM block + SFFN ahead of the other: two sampler steps at B=16 x 196 must give the SAME BITS as the
    undelayed run (fp32, and the fp16 split mode whose FiLM operand planes share one buffer between the groups)."""
    from motioncraft_amd.diffusion import build_diffusion
    sd, nm = full_model
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large'))
    # B = 16: the large-batch two-stream schedule; B = 2: the small-batch schedule (temporal branch / body branch forked onto the side stream
    # inside every layer: the same option holds the side or the main stream behind the fork)
    for B, T in ((16, 196), (2, 196)):
        g = torch.Generator().manual_seed(45)
        lengths = [int(v) for v in torch.randint(64, 197, (B,), generator=g)]
        x_T, xf, mask = synth_inputs(FULL, B, T, seed=46, lengths=lengths)
        eps = torch.randn(B, T, 322, generator=g).cuda()
        got = {}
        for arm, delay in (('plain', 0), ('second_late', 1500 if B > 2 else 300), ('first_late', -1500 if B > 2 else -300)):
            ctx = nm.context(B, T, max_steps=2)
            ctx.set_precision(prec)
            ctx.set_option('dbg_delay_us', delay)
            ctx.set_timesteps(d.timestep_map[-2:])
            ctx.set_condition(xf.cuda(), mask.cuda())
            x = x_T.cuda()
            for i in (1, 0):
                x = ctx.sample_step(x, i, d.step_coefs(998 + i, 'ddpm', FULL['scale']), eps)
            torch.cuda.synchronize()
            got[arm] = x.clone()
            ctx.close()
        assert bool(torch.isfinite(got['plain']).all())
        assert torch.equal(got['plain'], got['second_late']), (B, maxabs(got['plain'], got['second_late']))
        assert torch.equal(got['plain'], got['first_late']), (B, maxabs(got['plain'], got['first_late']))


def test_side_stream_is_picked_per_caller_stream_and_results_do_not_depend_on_it(full_model):
    """Round 6: a context times its four candidate side streams against the CALLER's stream (two HIP streams overlap only on different hardware
    queues; with an RCCL process group in the process the single side stream of round 5 shared the caller's queue and the two-stream
    schedule ran serially) and remembers the answer per caller stream.  The same two sampler steps issued on the default stream, on two
    other streams and on the default stream again (probe, probe, probe, memo) must give the SAME BITS -- which side stream carries the
    second sample group is a scheduling matter only; B = 16 (large-batch two-stream schedule) and B = 2 (temporal branch on the side stream)."""
    from motioncraft_amd.diffusion import build_diffusion
    sd, nm = full_model
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large'))
    for B, T in ((16, 196), (2, 196)):
        g = torch.Generator().manual_seed(51)
        x_T, xf, mask = synth_inputs(FULL, B, T, seed=52, lengths=[int(v) for v in torch.randint(64, 197, (B,), generator=g)])
        eps = torch.randn(B, T, 322, generator=g).cuda()
        ctx = nm.context(B, T, max_steps=2)
        ctx.set_timesteps(d.timestep_map[-2:])
        ctx.set_condition(xf.cuda(), mask.cuda())
        torch.cuda.synchronize()
        streams = [None, torch.cuda.Stream(), torch.cuda.Stream(), None, None]
        outs = []
        for st in streams:
            with torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.default_stream()):
                x = x_T.cuda()
                for i in (1, 0):
                    x = ctx.sample_step(x, i, d.step_coefs(998 + i, 'ddpm', FULL['scale']), eps)
                torch.cuda.current_stream().synchronize()
                outs.append(x.clone())
        torch.cuda.synchronize()
        ctx.close()
        assert bool(torch.isfinite(outs[0]).all())
        for k in range(1, len(outs)):
            assert torch.equal(outs[0], outs[k]), (B, k, maxabs(outs[0], outs[k]))


def test_unconditional_half_skips_its_text_rows_bit_identically(full_model):
    """chain bit 24 (round 5): in temporal_k the unconditional CFG half's text keys all carry the -1e6 of st_attention.py:153 and its text
    values are multiplied by c = 0 (:161) -- exact zeros in the column softmax and in K^T V as long as the sample has one valid frame --
    so whole leading blocks of those rows are skipped.  Must be the SAME BITS: B=16 x 196 frames in the large-batch schedule, ragged
    lengths, one sample of length 1 and one FULLY masked sample (no valid frame: its text rows are all it has, nothing may be skipped);
    y_t of base layer 0 and the decoded output, bit on / off.  Also at L = 64 (two parts per workgroup)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    for dims, tag in ((FULL, 'L128'), (W.default_dims(L=64, F=256), 'L64')):
        nm = full_model[1] if tag == 'L128' else NativeModel(dims, W.make_state_dict(dims, 3), cfg_scale=dims['scale'])
        B, T = 16, 196
        g = torch.Generator().manual_seed(25)
        lengths = [int(v) for v in torch.randint(64, 197, (B,), generator=g)]
        lengths[3], lengths[7] = 1, 0
        x_T, xf, mask = synth_inputs(dims, B, T, seed=36, lengths=lengths)
        for prec in (('f32', 'f16x3', 'f16') if tag == 'L128' else ('f32',)):      # (reduced-precision contexts: temporal_h_k skips the same chunks)
            got = {}
            for arm, chain in (('skip', DEFAULT_CHAIN), ('all_rows', DEFAULT_CHAIN & ~(1 << 24))):
                ctx = nm.context(B, T, max_steps=1)
                ctx.set_option('chain', chain)
                ctx.set_precision(prec)
                ctx.set_timesteps([640])
                ctx.set_condition(xf.cuda(), mask.cuda())
                out = ctx.denoise(x_T.cuda(), 0).clone()
                ctx.denoise(x_T.cuda(), 0, stop_after_layers=1)
                torch.cuda.synchronize()
                got[arm] = (out, ctx.buffer('yt').clone())
                ctx.close()
            assert bool(torch.isfinite(got['skip'][0]).all())
            assert torch.equal(got['skip'][1], got['all_rows'][1]), (tag, prec, maxabs(got['skip'][1], got['all_rows'][1]))
            assert torch.equal(got['skip'][0], got['all_rows'][0]), (tag, prec, maxabs(got['skip'][0], got['all_rows'][0]))
        if tag != 'L128':
            nm.close()


def test_gate_launch_cut_at_whole_rounds_is_bit_identical(full_model):
    """chain bit 23 (round 5; measured slower, off by default, kept as a switch): in the two-stream schedule the last sample group's gate launch is cut at a whole number of workgroup
    rounds and the partial last round runs as gate_small_k on the other group's stream (beside the big launch).  gate_small_k
    reproduces gate_k's accumulation order, so scores / expert choices / importance keys -- and with them everything downstream --
    must be the SAME BITS: B=32 x 196 frames (588 tiles per group: 512 + 76), one denoiser call + routing of every layer, bit on / off;
    and the same through two sampler steps."""
    from motioncraft_amd.diffusion import build_diffusion
    sd, nm = full_model
    B, T = 32, 196
    g = torch.Generator().manual_seed(15)
    lengths = [int(v) for v in torch.randint(64, 197, (B,), generator=g)]
    x_T, xf, mask = synth_inputs(FULL, B, T, seed=35, lengths=lengths)
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large'))
    eps = torch.randn(B, T, 322, generator=g).cuda()
    got = {}
    for tag, chain in (('cut', DEFAULT_CHAIN | (1 << 23)), ('whole', DEFAULT_CHAIN)):
        ctx = nm.context(B, T, max_steps=2)
        ctx.set_option('chain', chain)
        ctx.enable_capture()
        ctx.set_timesteps(d.timestep_map[-2:])
        ctx.set_condition(xf.cuda(), mask.cuda())
        out2 = ctx.denoise(x_T.cuda(), 1).clone()
        routes = [ctx.routing(i) for i in range(FULL['NL'])]
        x = x_T.cuda()
        for i in (1, 0):
            x = ctx.sample_step(x, i, d.step_coefs(998 + i, 'ddpm', FULL['scale']), eps)
        torch.cuda.synchronize()
        got[tag] = (out2, routes, x.clone())
        ctx.close()
    assert bool(torch.isfinite(got['cut'][0]).all())
    assert torch.equal(got['cut'][0], got['whole'][0]), maxabs(got['cut'][0], got['whole'][0])
    for a, b in zip(got['cut'][1], got['whole'][1]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(got['cut'][2], got['whole'][2])


@pytest.mark.parametrize('L', [32, 128])
def test_small_batch_gate_kernel_is_bit_identical_to_gate_k(L, monkeypatch):
    """gate_small_k (32-token workgroups, projector chunks split over the waves: batches of up to 12000 tokens) against
    gate_k on the same input: expert ids, renormalised gates, importance keys and the expert input z must match BIT FOR BIT --
    the kernel a batch size selects must never change a routing decision (a tree sum of the waves' partial logits differs
    by 1 ulp and flips near-tie choices of this flat-gate small model within a 50-step loop)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = W.default_dims(max_seq_len=24, L=L, NL=2, F=64, Te=64, Dt=32, Nt=8)
    nm = NativeModel(dims, W.make_state_dict(dims, SMALL_SEED), cfg_scale=dims['scale'])
    B, T = 2, 24
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, 322, generator=g).cuda()
    xf = torch.nn.functional.layer_norm(torch.randn(B, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],)).cuda()
    got = {}
    for mode in ('0', '1000000'):
        monkeypatch.setenv('MC_GATE_SMALL', mode)            # read when the context is created
        ctx = nm.context(B, T, max_steps=50)
        ctx.set_timesteps(list(range(0, 1000, 20)))
        ctx.set_condition(xf, torch.ones(B, T).cuda())
        ctx.denoise(x, 30, stop_after_layers=2)              # buffers hold the gate outputs of layer 1 (all N tokens, no CFG twins)
        torch.cuda.synchronize()
        got[mode] = {k: ctx.buffer(k, dtype=torch.int32).cpu().clone() for k in ('idx', 'gate', 'key', 'z')}
        ctx.close()
    for k in ('idx', 'gate', 'key', 'z'):
        assert torch.equal(got['0'][k], got['1000000'][k]), k
    nm.close()


@pytest.mark.parametrize('case', ['one_frame', 'fully_masked_sample', 'eight_experts', 'tiny_capacity', 'odd_tokens'])
def test_edge_cases_vs_oracle(case):
    """Shapes and routing regimes the reference's own paths can reach: a single frame (temporal softmax over the text
    tokens + 1 frame), a sample whose motion mask is all zero (keys at -1e6: the column softmax degenerates to the
    text rows / uniform), fewer experts than the 16 gate lanes, a capacity small enough that second choices are
    dropped wholesale (limit <= 0 branch) and most first choices too, and an odd token count (no CFG twin tiles of
    128, ragged everything)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    over, B, T, lengths, capf = {}, 2, 24, [24, 17], 1.5
    if case == 'one_frame':
        B, T, lengths = 3, 1, [1, 1, 1]
    elif case == 'fully_masked_sample':
        lengths = [24, 0]
    elif case == 'eight_experts':
        over = dict(E=8)
    elif case == 'tiny_capacity':
        capf = 0.5
    elif case == 'odd_tokens':
        B, T, lengths = 1, 7, [5]
    dims = W.default_dims(max_seq_len=24, L=32, NL=2, F=64, Te=64, Dt=32, Nt=8, **over)
    sd = W.make_state_dict(dims, SMALL_SEED)
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'], capacity_factor=capf)
    x, xf, mask = synth_inputs(dims, B, T, seed=91, lengths=lengths)
    ctx = nm.context(B, T, max_steps=2)
    ctx.enable_capture()
    ctx.set_timesteps([777, 0])
    ctx.set_condition(xf.cuda(), mask.cuda())
    for s_, t in ((0, 777), (1, 0)):
        out2 = ctx.denoise(x.cuda(), s_)
        assert torch.isfinite(out2).all()
        w = (1 - (1000 - t) / 1000) * dims['scale'] + 1
        got = out2[:B] * w + out2[B:] * (1 - w)
        forced = [ctx.routing(l) for l in range(dims['NL'])]
        cap = {}
        ref = O.denoise(sd, dims, x, t, xf, mask, forced_routing=forced, cap=cap, capacity_factor=capf)
        err = maxabs(got, ref)
        flips = 0
        for l in range(dims['NL']):
            free = cap[f'layer{l}']['routing']['free']
            flips += int((torch.stack(free['indices'], 1) != forced[l][0]).sum())
            flips += int((torch.stack(free['keeps'], 1) != forced[l][1]).sum())
        dropped = sum(int((~forced[l][1]).sum()) for l in range(dims['NL']))
        print(f'{case} t={t}: |hip - oracle| {err:.2e}, dropped pairs {dropped}, routing flips vs free-running oracle {flips}')
        assert err <= TOL_STEP and flips == 0
        if case == 'tiny_capacity':
            assert dropped > 0.3 * forced[0][1].numel()
    ctx.close()
    nm.close()


def test_long_sequence_windows_repaint_vs_oracle(small_model):
    """Caller side of the RePaint mode (tools/m2d_test.py:139-232 window loop): 3 windows of 24 frames advancing by
    18, each later window keeping its first 6 frames equal to the previous window's last 6; the HIP windows are
    checked against the CPU oracle run with the same draws and the same window plumbing."""
    import types
    import motioncraft_amd as mc
    from motioncraft_amd import longform
    from oracle import stmogen_oracle as O, weights as W
    sd, _ = small_model
    opt = types.SimpleNamespace(same_overlap_noisy=False, no_repaint=False, addBlend=True, overlap_len=6, no_resample=True,
                                jump_length=3, jump_n_sample=5, timestep_respacing='ddim50')
    cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
    cfg.model['opt'] = opt
    arch = mc.build_architecture(cfg.model)
    arch.load_state_dict({'model.' + k: v for k, v in sd.items()})
    g = torch.Generator().manual_seed(123)
    xf = torch.nn.functional.layer_norm(torch.randn(1, SMALL['Nt'], SMALL['Dt'], generator=g), (SMALL['Dt'],))
    first_gt = torch.randn(6, 322, generator=g)
    x_Ts = [torch.randn(1, 24, 322, generator=g) for _ in range(3)]
    total, L, pre = 60, 24, 6
    assert longform.window_starts(total, L, pre) == (3, 18)

    def draws(seed):
        gen = torch.Generator().manual_seed(seed)
        return (torch.randn(1, 24, 322, generator=gen) for _ in range(10 ** 6))
    rec, wins = longform.sample_long(arch, total, L, pre, repaint=True, overlap_len=6, first_gt=first_gt,
                                     condition_kwargs=dict(xf_out=xf.cuda()),
                                     inference_kwargs=lambda i: dict(noise=x_Ts[i], step_noise=draws(50 + i)))
    assert rec.shape == (18 + 18 + 24, 322) and len(wins) == 3
    # the same loop on the oracle
    sched, mask, prev, ref_parts = O.Schedule(1000, '15,15,8,6,6'), torch.ones(1, 24), None, []
    keep = torch.zeros(1, 24, 322, dtype=torch.bool)
    keep[:, :6] = True
    for i in range(3):
        gt = torch.zeros(1, 24, 322)
        gt[:, :6] = first_gt if i == 0 else prev[:, -6:]
        prev = O.sample_loop_repaint(sd, SMALL, sched, x_Ts[i], xf, mask, keep, gt, draws(50 + i), 6, 50, no_resample=True)
        ref_parts.append(prev[0] if i == 2 else prev[0, :18])
        err = maxabs(T_(wins[i]), prev[0])
        print(f'long-form window {i}: |hip - oracle| {err:.2e}')
        assert err <= TOL_FINAL
    assert maxabs(T_(rec), torch.cat(ref_parts)) <= TOL_FINAL
    assert np.abs(wins[1][0] - wins[0][-6]).max() <= 1e-6          # frame 0 of a later window IS the previous frame -6
    arch.model.release()


def test_batched_long_sequence_windows_vs_oracle_on_the_same_batches(small_model):
    """longform.sample_long_batched (BASELINE configs[3]; reference loop tools/m2d_test.py:139-232 at B = 1): S sequences x W windows
    folded into the batch.  The MoE capacity couples the windows of one model call, so the criterion is NOT "equals the window sampled
    alone" but "equals the ORACLE on the same batch" -- plain mode: ONE call over all S*W windows against O.sample_loop on that batch;
    RePaint mode: window i of all S sequences per call, chained through the previous outputs, against O.sample_loop_repaint with the
    same window plumbing.  Stitching is checked against stitch_windows of the oracle's windows."""
    import types
    import motioncraft_amd as mc
    from motioncraft_amd import longform
    from oracle import stmogen_oracle as O
    sd, _ = small_model
    opt = types.SimpleNamespace(same_overlap_noisy=False, no_repaint=False, addBlend=True, overlap_len=6, no_resample=True,
                                jump_length=3, jump_n_sample=5, timestep_respacing='ddim50')
    cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
    cfg.model['opt'] = opt
    arch = mc.build_architecture(cfg.model)
    arch.load_state_dict({'model.' + k: v for k, v in sd.items()})
    S, total, L, pre = 3, 42, 24, 6
    n_win, stride = longform.window_starts(total, L, pre)
    assert (n_win, stride) == (2, 18)
    # (seed: the SMALL config's gate distributions are flat, and a free-running 30-step loop over a capacity-coupled batch leaves the oracle's
    # trajectory through ONE near-tie routing flip for about 1 seed in 8 -- measured in round 5 with a scratch sweep over seeds)
    g = torch.Generator().manual_seed(int(os.environ.get('MC_TEST_SEED', 324)))
    xf = torch.nn.functional.layer_norm(torch.randn(S, SMALL['Nt'], SMALL['Dt'], generator=g), (SMALL['Dt'],))
    first_gt = torch.randn(S, 6, 322, generator=g)
    sched = O.Schedule(1000, '15,15,8,6,6')
    torch.set_num_threads(min(32, os.cpu_count()))
    # ---- plain mode: one batch of S * W = 6 windows, pairs in (sequence, window) order ----
    x_T = torch.randn(S * n_win, L, 322, generator=g)
    noises = step_noise_from_seed(77, (S * n_win, L, 322), 50)
    seen = []

    def inf_plain(pairs):
        seen.append(list(pairs))
        return dict(noise=x_T, step_noise=[noises[49 - i] for i in range(50)])       # step_noise[i] = the draw of schedule index i
    recs, wins = longform.sample_long_batched(arch, total, L, pre, text=['a'] * S, repaint=False, condition_kwargs=dict(xf_out=xf.cuda()),
                                              inference_kwargs=inf_plain, max_batch=160, shard=False)
    assert seen == [[(s, w) for s in range(S) for w in range(n_win)]]
    rows = torch.tensor([s for s, _ in seen[0]])
    ref = O.sample_loop(sd, SMALL, sched, 'ddim', x_T, xf[rows], torch.ones(S * n_win, L), step_noise=lambda i: noises[49 - i])
    e = max(maxabs(T_(wins[p]), ref[j]) for j, p in enumerate(seen[0]))
    print(f'batched windows, plain mode: {S} sequences x {n_win} windows in one call: |hip - oracle| {e:.2e}')
    assert e <= TOL_FINAL
    for s in range(S):
        want = longform.stitch_windows([ref[s * n_win + w].numpy() for w in range(n_win)], pre, False)
        assert recs[s].shape == (stride + L, 322) and float(np.abs(recs[s] - want).max()) <= TOL_FINAL
    # ---- RePaint mode: window i of all S sequences per call ----
    x_Ts = [torch.randn(S, L, 322, generator=g) for _ in range(n_win)]

    def draws(seed):
        gen = torch.Generator().manual_seed(seed)
        return (torch.randn(S, L, 322, generator=gen) for _ in range(10 ** 6))
    seen.clear()

    def inf_rep(pairs):
        seen.append(list(pairs))
        w = pairs[0][1]
        return dict(noise=x_Ts[w], step_noise=draws(60 + w))
    recs, wins = longform.sample_long_batched(arch, total, L, pre, text=['a'] * S, repaint=True, overlap_len=6, first_gt=first_gt,
                                              condition_kwargs=dict(xf_out=xf.cuda()), inference_kwargs=inf_rep, max_batch=160, shard=False)
    assert seen == [[(s, w) for s in range(S)] for w in range(n_win)]
    keep = torch.zeros(S, L, 322, dtype=torch.bool)
    keep[:, :6] = True
    prev, ref_w = None, []
    for w in range(n_win):
        gt = torch.zeros(S, L, 322)
        gt[:, :6] = first_gt if w == 0 else prev[:, -6:]
        prev = O.sample_loop_repaint(sd, SMALL, sched, x_Ts[w], xf, torch.ones(S, L), keep, gt, draws(60 + w), 6, 50, no_resample=True)
        ref_w.append(prev)
        e = max(maxabs(T_(wins[(s, w)]), prev[s]) for s in range(S))
        print(f'batched windows, RePaint mode, window {w} of {S} sequences: |hip - oracle| {e:.2e}', [f'{maxabs(T_(wins[(s, w)]), prev[s]):.1e}' for s in range(S)])
        assert e <= TOL_FINAL
    for s in range(S):
        want = longform.stitch_windows([ref_w[w][s].numpy() for w in range(n_win)], pre, True)
        assert float(np.abs(recs[s] - want).max()) <= TOL_FINAL
        assert np.abs(wins[(s, 1)][0] - wins[(s, 0)][-6]).max() <= 1e-6
    # the driver's plumbing, independent of the oracle (bit-exact): window 1 of all sequences = one direct ddim_sample_loop call on the batch
    # whose gt rows are the driver's own window-0 outputs
    gt = torch.zeros(S, L, 322)
    for s in range(S):
        gt[s, :6] = T_(wins[(s, 0)])[-6:]
    mk = dict(xf_out=xf.cuda(), motion_mask=torch.ones(S, L).cuda(), y=dict(gt=gt.cuda(), outpainting_mask=keep.cuda()))
    direct = arch.diffusion_test.ddim_sample_loop(arch.model, (S, L, 322), noise=x_Ts[1], clip_denoised=False, model_kwargs=mk, eta=0,
                                                  step_noise=draws(61)).cpu()
    for s in range(S):
        assert torch.equal(direct[s], T_(wins[(s, 1)])), s
    arch.model.release()


@pytest.mark.parametrize('case', ['s2g_025b', 'm2d_finedance'])
def test_baseline_control_configs_at_full_architecture_vs_oracle(case):
    """BASELINE configs[2] / [3] architectures at their real widths, one denoiser call at B=1 vs the CPU oracle:
    S2G (configs/stmogen/S2G_Beats2_no_face_loss_025b.py: L=128, 8 layers, copy_blocks_num=2, T=64... here 196 frames,
    pre-encoded audio condition of width D) and M2D (M2D_finedance_no_face_loss.py: L=64, 4 layers, copy_blocks_num=3,
    35-d music features, 120-frame windows)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    if case == 's2g_025b':
        dims, copy, feats, T, Tc = W.default_dims(NL=8), 2, 1536, 196, 150
    else:
        dims, copy, feats, T, Tc = W.default_dims(L=64, F=256), 3, 35, 120, 120
    sd = W.make_state_dict(dims, 0, shapes=W.control_param_shapes(dims, copy, feats))
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    assert nm.copy_blocks_num == copy and nm.control_cond_feats == feats
    x, xf, mask = synth_inputs(dims, 1, T, seed=71, lengths=[T - 9])
    g = torch.Generator().manual_seed(72)
    c = torch.randn(1, Tc, feats, generator=g)
    ctx = nm.context(1, T, max_steps=1)
    ctx.enable_capture()
    ctx.set_timesteps([480])
    ctx.set_condition(xf.cuda(), mask.cuda())
    ctx.set_control(c.cuda())
    out2 = ctx.denoise(x.cuda(), 0)
    w = (1 - (1000 - 480) / 1000) * dims['scale'] + 1
    got = out2[:1] * w + out2[1:] * (1 - w)
    torch.set_num_threads(min(32, os.cpu_count()))
    ref = O.denoise_control(sd, dims, x, 480, xf, mask, c, copy)
    err = maxabs(got, ref)
    print(f'{case}: NL={dims["NL"]} L={dims["L"]} copy={copy}: |hip - oracle| {err:.2e} (|ref| max {float(ref.abs().max()):.2f})')
    assert err <= TOL_STEP
    ctx.close()
    nm.close()


def test_control_branch_through_the_sampler_step():
    """The sampler entry point (CFG-combined, folded tail) with the control branch active: 12 DDIM steps of the small
    control config against the oracle's denoise_control + ddim_step, including the end of the schedule (i = 0)."""
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    g = load('control_small.npz')
    sd = W.make_state_dict(CTRL, SMALL_SEED, shapes=W.control_param_shapes(CTRL, CTRL_COPY, CTRL_FEATS))
    nm = NativeModel(CTRL, sd, cfg_scale=CTRL['scale'])
    x, xf, mask, c = (T_(g[k]) for k in ('x_t', 'xf_out', 'motion_mask', 'c'))
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                             model_var_type='fixed_large', respace='15,15,8,6,6'))
    sched = O.Schedule(1000, '15,15,8,6,6')
    ctx = nm.context(2, 24, max_steps=50)
    ctx.set_timesteps(d.timestep_map)
    ctx.set_condition(xf.cuda(), mask.cuda())
    ctx.set_control(c.cuda())
    noises = step_noise_from_seed(4, tuple(x.shape), 12)
    xg, xo = x.cuda(), x
    for n, i in enumerate([49, 48, 47, 30, 29, 11, 10, 4, 3, 2, 1, 0]):
        xg_in = xg.cpu()
        xg = ctx.sample_step(xg, i, d.step_coefs(i, 'ddim', CTRL['scale']), noises[n].cuda())
        x0 = O.denoise_control(sd, CTRL, xg_in, sched.timestep_map[i], xf, mask, c, CTRL_COPY)
        ref = O.ddim_step(sched, i, xg_in, x0, noises[n])
        assert maxabs(xg, ref) <= TOL_STEP, i
    ctx.close()
    nm.close()


@pytest.mark.parametrize('which', ['all_layers', 'layer0_only'])
def test_exact_score_ties_and_twin_pairs_split_by_capacity(which):
    """Degenerate gate (cosine projector with zero weight, so its output is the bias for every token): all tokens
    have the SAME score vector, pick the same two experts with exactly tied importance, batch-prioritised routing falls
    back to token order, those two experts overflow massively and the capacity cut lands BETWEEN a first-CFG-half
    token and its identical second-half twin.  Exercises the stable tie order of the radix select and the twin-dedupe
    bookkeeping of base layer 0 when twins end up with different keep flags; routing must equal the free-running oracle
    exactly.  (Ties BETWEEN experts inside one token's top-2 are left out: torch.topk's tie order is
    implementation-defined, i.e. the reference itself has no single answer there.)"""
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    dims = SMALL
    sd = W.make_state_dict(dims, SMALL_SEED)
    layers = range(dims['NL']) if which == 'all_layers' else [0]
    for l in layers:
        pre = f'temporal_decoder_blocks.{l}.ca_block.motion_moe.model.gates.0.cosine_projector.'
        sd[pre + 'weight'] = torch.zeros_like(sd[pre + 'weight'])
        sd[pre + 'bias'] = torch.randn(sd[pre + 'bias'].shape, generator=torch.Generator().manual_seed(11 + l))
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    x, xf, mask = synth_inputs(dims, 2, 24, seed=97, lengths=[24, 20])
    ctx = nm.context(2, 24, max_steps=1)
    ctx.enable_capture()
    ctx.set_timesteps([333])
    ctx.set_condition(xf.cuda(), mask.cuda())
    out2 = ctx.denoise(x.cuda(), 0)
    w = (1 - (1000 - 333) / 1000) * dims['scale'] + 1
    got = out2[:2] * w + out2[2:] * (1 - w)
    cap = {}
    ref = O.denoise(sd, dims, x, 333, xf, mask, cap=cap)                 # free-running oracle: no teacher forcing
    N = 2 * 2 * 24 * dims['H']
    for l in layers:
        idx, keep = ctx.routing(l)
        free = cap[f'layer{l}']['routing']['free']
        assert torch.equal(idx, torch.stack(free['indices'], 1)) and torch.equal(keep, torch.stack(free['keeps'], 1)), l
        assert bool((idx[:, 0] == idx[0, 0]).all()) and bool((idx[:, 1] == idx[0, 1]).all())
        assert bool(keep[:N // 2, 0].any()) and not bool(keep[N // 2:, 0].any())     # twins split by the capacity cut
    err = maxabs(got, ref)
    print(f'exact ties ({which}): |hip - oracle| {err:.2e}')
    assert err <= TOL_STEP
    # the other tie policy (a16: tutel's argsort is not stable, the order of equal-importance tokens is implementation-
    # defined): ONE switch in the kernel (mc_ctx_set_tie_policy) and one in the oracle (TIE_POLICY) -- the higher token
    # index now ranks first, so the capacity cut keeps the SECOND-half twins and drops the originals
    from oracle import tutel_restated as TR
    ctx.set_tie_policy('reverse')
    with pytest.raises(RuntimeError):          # the hoisted text K/V were routed under the old policy
        ctx.denoise(x.cuda(), 0)
    ctx.set_condition(xf.cuda(), mask.cuda())
    out2r = ctx.denoise(x.cuda(), 0)
    gotr = out2r[:2] * w + out2r[2:] * (1 - w)
    TR.TIE_POLICY = 'reverse'
    try:
        capr = {}
        refr = O.denoise(sd, dims, x, 333, xf, mask, cap=capr)
    finally:
        TR.TIE_POLICY = 'stable'
    for l in layers:
        idx, keep = ctx.routing(l)
        free = capr[f'layer{l}']['routing']['free']
        assert torch.equal(idx, torch.stack(free['indices'], 1)) and torch.equal(keep, torch.stack(free['keeps'], 1)), l
        assert bool(keep[N // 2:, 0].any()) and not bool(keep[:N // 2, 0].any())
    errr = maxabs(gotr, refr)
    print(f'exact ties ({which}), reverse tie policy: |hip - oracle| {errr:.2e}; |stable - reverse| {maxabs(got, gotr):.2e}')
    assert errr <= TOL_STEP and maxabs(got, gotr) > 1e-3
    ctx.close()
    nm.close()


def test_fused_proj_qkv_body_kernel_at_latent_64_vs_the_separate_kernels():
    """pqbody_k<64> (the M2D width: 8 dynamic heads of 8 channels, two heads per DPP row, the head walked by XOR reads instead of row
    rotations) against projqkv_k<64> + body_reg_k<8>: mf the same bits, ys within fp32 round-off (the contraction over the head's 8
    channels is summed in another order than body_reg_k's ascending-d loop).  L = 64 architecture at a small batch in the large-batch
    schedule, ragged tiles, twin aliasing on (layer 1 snapshot) and off (stop_after_layers = 1)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = W.default_dims(L=64, F=256, max_seq_len=24)
    nm = NativeModel(dims, W.make_state_dict(dims, 3), cfg_scale=dims['scale'])
    B, T = 3, 24
    x, xf, mask = synth_inputs(dims, B, T, seed=6, lengths=[24, 18, 11])
    got = {}
    for tag, chain in (('fused', DEFAULT_CHAIN), ('separate', DEFAULT_CHAIN & ~(1 << 15)), ('mlp_reg_staged', DEFAULT_CHAIN & ~(1 << 18))):
        ctx = nm.context(B, T, max_steps=1)
        ctx.set_option('big_tokens', 0)
        ctx.set_option('chain', chain)
        ctx.set_timesteps([700])
        ctx.set_condition(xf.cuda(), mask.cuda())
        out = ctx.denoise(x.cuda(), 0).clone()
        ctx.denoise(x.cuda(), 0, stop_after_layers=1)
        torch.cuda.synchronize()
        got[tag] = (out, ctx.buffer('ys').clone(), ctx.buffer('mf').clone())
        ctx.close()
    assert torch.equal(got['fused'][2], got['separate'][2])
    # mlp2d_k<64> (LDS-DMA staged expert / SFFN weight chunks) vs mlp2_k<64>: the same MFMA order, the same bits everywhere
    for k in range(3):
        assert torch.equal(got['fused'][k], got['mlp_reg_staged'][k]), k
    e_ys, e_out = maxabs(got['fused'][1], got['separate'][1]), maxabs(got['fused'][0], got['separate'][0])
    print(f'L = 64: |ys fused - separate| {e_ys:.2e} (|ys| max {float(got["separate"][1].abs().max()):.2f}), |x0| {e_out:.2e}')
    assert bool(torch.isfinite(got['fused'][0]).all()) and e_ys <= 1e-5 and e_out <= 1e-4
    # the fp16-MFMA twin (pqbody_h_k<64>) against projqkv_h_k<64> + body_reg_k<8> in the split mode
    hgot = {}
    for tag, chain in (('fused', DEFAULT_CHAIN), ('separate', DEFAULT_CHAIN & ~(1 << 15)), ('mlp_reg_staged', DEFAULT_CHAIN & ~(1 << 18))):
        ctx = nm.context(B, T, max_steps=1)
        ctx.set_option('big_tokens', 0)
        ctx.set_option('half_min_rows', 0)
        ctx.set_option('chain', chain)
        ctx.set_precision('f16x3')
        ctx.set_timesteps([700])
        ctx.set_condition(xf.cuda(), mask.cuda())
        out = ctx.denoise(x.cuda(), 0).clone()
        ctx.denoise(x.cuda(), 0, stop_after_layers=1)
        torch.cuda.synchronize()
        hgot[tag] = (out, ctx.buffer('ys').clone(), ctx.buffer('mf').clone())
        ctx.close()
    assert torch.equal(hgot['fused'][2], hgot['separate'][2])
    for k in range(3):          # mlp2hd_k<64> vs mlp2_h_k<64>
        assert torch.equal(hgot['fused'][k], hgot['mlp_reg_staged'][k]), k
    e_ys, e_out = maxabs(hgot['fused'][1], hgot['separate'][1]), maxabs(hgot['fused'][0], hgot['separate'][0])
    print(f'L = 64, f16x3: |ys fused - separate| {e_ys:.2e}, |x0| {e_out:.2e}; |x0 f16x3 - x0 f32| {maxabs(hgot["fused"][0], got["fused"][0]):.2e}')
    assert e_ys <= 1e-5 and e_out <= 1e-4 and maxabs(hgot['fused'][0], got['fused'][0]) <= 2e-4
    nm.close()


def test_temporal_attention_two_parts_per_workgroup_at_latent_64():
    """temporal_k<64, PAIR> (round 4, chain bit 22: one workgroup owns two adjacent body parts of a sample, so that all four waves carry
    MFMA tiles at L = 64) against the one-part form on the same mf / text rows: y_t of base layer 0 within fp32 round-off (the column
    statistics are combined over 8 instead of 16 row slices), ragged lengths, the masked-text half included; x0 of the step as well."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = W.default_dims(L=64, F=256, max_seq_len=24)
    nm = NativeModel(dims, W.make_state_dict(dims, 5), cfg_scale=dims['scale'])
    B, T = 3, 24
    x, xf, mask = synth_inputs(dims, B, T, seed=10, lengths=[24, 15, 7])
    got = {}
    for tag, chain in (('pair', DEFAULT_CHAIN), ('single', DEFAULT_CHAIN & ~(1 << 22))):
        ctx = nm.context(B, T, max_steps=1)
        ctx.set_option('big_tokens', 0)
        ctx.set_option('temporal_split', 0)
        ctx.set_option('chain', chain)
        ctx.set_timesteps([620])
        ctx.set_condition(xf.cuda(), mask.cuda())
        out = ctx.denoise(x.cuda(), 0).clone()
        ctx.denoise(x.cuda(), 0, stop_after_layers=1)
        torch.cuda.synchronize()
        got[tag] = (out, ctx.buffer('yt').clone(), ctx.buffer('mf').clone())
        ctx.close()
    assert torch.equal(got['pair'][2], got['single'][2])
    scale = float(got['single'][1].abs().max())
    e, e0 = maxabs(got['pair'][1], got['single'][1]), maxabs(got['pair'][0], got['single'][0])
    print(f'L = 64 temporal attention, two parts per workgroup: |dy_t| {e:.2e} (|y_t| max {scale:.2f}), |dx0| {e0:.2e}')
    assert bool(torch.isfinite(got['pair'][1]).all()) and e <= 2e-6 * max(scale, 1.0) and e0 <= 1e-4
    nm.close()


def test_twin_pairs_split_by_capacity_in_the_large_batch_schedule():
    """The same degenerate gate in base layer 0 of the 0.125b architecture, pushed into the LARGE-batch schedule (big_tokens = 0): two
    sample groups on two streams, the twin layer's front as two sample sub-groups (round 4), pqbody_k.  The capacity cut separates
    first-half tokens from their CFG twins, so the routing raises the split flag: no row may be aliased, the second CFG half runs its
    own front (pqbody_k reading its twins' expert rows through twin_from) behind the cross-join.  Routing must equal the free-running
    oracle exactly, the output within the per-call tolerance; beside it the same call with the round-4 schedule bits off."""
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    dims = FULL
    sd = W.make_state_dict(dims, 0)
    pre = 'temporal_decoder_blocks.0.ca_block.motion_moe.model.gates.0.cosine_projector.'
    sd[pre + 'weight'] = torch.zeros_like(sd[pre + 'weight'])
    sd[pre + 'bias'] = torch.randn(sd[pre + 'bias'].shape, generator=torch.Generator().manual_seed(11))
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    B, T = 3, 24
    x, xf, mask = synth_inputs(dims, B, T, seed=97, lengths=[24, 20, 9])
    w = (1 - (1000 - 333) / 1000) * dims['scale'] + 1
    cap = {}
    ref = O.denoise(sd, dims, x, 333, xf, mask, cap=cap)                 # free-running oracle
    free = cap['layer0']['routing']['free']
    N = 2 * B * T * dims['H']
    outs = []
    # (default; then with the fused front of a sub-group also covering that sub-group's twins, chain bit 25 (off by default: slower); then without the
    # event that orders the groups' FiLM blocks, bit 26; then the round-4 schedule bits off)
    for chain in (DEFAULT_CHAIN, DEFAULT_CHAIN | (1 << 25), DEFAULT_CHAIN & ~(1 << 26), DEFAULT_CHAIN & ~((1 << 15) | (1 << 16))):
        ctx = nm.context(B, T, max_steps=1)
        ctx.set_option('big_tokens', 0)
        ctx.set_option('chain', chain)
        ctx.enable_capture()
        ctx.set_timesteps([333])
        ctx.set_condition(xf.cuda(), mask.cuda())
        out2 = ctx.denoise(x.cuda(), 0)
        got = out2[:B] * w + out2[B:] * (1 - w)
        idx, keep = ctx.routing(0)
        assert torch.equal(idx, torch.stack(free['indices'], 1)) and torch.equal(keep, torch.stack(free['keeps'], 1))
        assert bool(keep[:N // 2, 0].any()) and not bool(keep[N // 2:, 0].any())     # twins split by the capacity cut
        err = maxabs(got, ref)
        print(f'chain {chain}: twins split in the large-batch schedule: |hip - oracle| {err:.2e}')
        assert err <= TOL_STEP
        outs.append(got.clone())
        ctx.close()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    nm.close()


@pytest.mark.parametrize('regime,B', [('random_scores', 3), ('exact_ties', 3), ('tiny_capacity', 3), ('reverse_ties', 3),
                                      ('random_scores', 9), ('tiny_capacity', 9), ('exact_ties', 9)])
def test_register_routing_kernel_equals_the_streaming_form(regime, B, monkeypatch):
    """route_small_k<10> / <20> (pairs in registers, selection problems resolved as soon as a bin is taken whole; B = 3: 3456
    pairs -> <10>, B = 9: 10368 pairs -> <20>) against the L2-streaming one-workgroup kernel they replace (MC_ROUTE_REG=0): identical keep flags / combine weights and
    bit-identical denoiser output, with every radix pass exercised -- random scores (resolved after the 4 score bytes), exact
    score ties (only the token-index bytes split them), capacity so small that second choices are dropped wholesale, and the
    reverse tie order."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = SMALL
    sd = W.make_state_dict(dims, SMALL_SEED)
    if regime in ('exact_ties', 'reverse_ties'):
        for l in range(dims['NL']):
            pre = f'temporal_decoder_blocks.{l}.ca_block.motion_moe.model.gates.0.cosine_projector.'
            sd[pre + 'weight'] = torch.zeros_like(sd[pre + 'weight'])
            sd[pre + 'bias'] = torch.randn(sd[pre + 'bias'].shape, generator=torch.Generator().manual_seed(21 + l))
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'], capacity_factor=0.3 if regime == 'tiny_capacity' else 1.5)
    x, xf, mask = synth_inputs(dims, B, 24, seed=5, lengths=([24, 20, 7] * 3)[:B])
    got = {}
    for reg in ('0', '1'):
        monkeypatch.setenv('MC_ROUTE_REG', reg)               # read when the context is created
        ctx = nm.context(B, 24, max_steps=1)
        if regime == 'reverse_ties':
            ctx.set_tie_policy('reverse')
        ctx.enable_capture()
        ctx.set_timesteps([500])
        ctx.set_condition(xf.cuda(), mask.cuda())
        out2 = ctx.denoise(x.cuda(), 0).clone()
        got[reg] = (out2, [ctx.routing(l) for l in range(dims['NL'])])
        ctx.close()
    assert torch.equal(got['0'][0], got['1'][0])
    dropped = 0
    for (ia, ka), (ib, kb) in zip(got['0'][1], got['1'][1]):
        assert torch.equal(ia, ib) and torch.equal(ka, kb)
        dropped += int((~ka).sum())
    assert dropped > 0, 'the capacity cut never engaged: the radix passes were not exercised'
    nm.close()


@pytest.mark.parametrize('regime,B', [('random_scores', 3), ('exact_ties', 3), ('tiny_capacity', 9), ('reverse_ties', 3),
                                      ('random_scores', 9), ('exact_ties', 9)])
def test_cooperative_routing_kernel_equals_the_launch_sequence(regime, B, monkeypatch):
    """route_coop_k (the routing of a layer as ONE launch with grid barriers: what batches above 32768 pairs run) against
    the 12-launch sequence it replaces (MC_ROUTE_COOP=0), both forced onto the small config (MC_ROUTE_SMALL_CTX=0 switches the
    one-workgroup kernels off): identical expert ids / keep flags and bit-identical denoiser output over random scores,
    exact ties (all 8 radix passes), wholesale drops and the reverse tie order, on 2 and 5 workgroups."""
    from motioncraft_amd.engine import NativeModel
    from oracle import weights as W
    dims = SMALL
    sd = W.make_state_dict(dims, SMALL_SEED)
    if regime in ('exact_ties', 'reverse_ties'):
        for l in range(dims['NL']):
            pre = f'temporal_decoder_blocks.{l}.ca_block.motion_moe.model.gates.0.cosine_projector.'
            sd[pre + 'weight'] = torch.zeros_like(sd[pre + 'weight'])
            sd[pre + 'bias'] = torch.randn(sd[pre + 'bias'].shape, generator=torch.Generator().manual_seed(21 + l))
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'], capacity_factor=0.3 if regime == 'tiny_capacity' else 1.5)
    x, xf, mask = synth_inputs(dims, B, 24, seed=6, lengths=([24, 20, 7] * 3)[:B])
    got = {}
    monkeypatch.setenv('MC_ROUTE_SMALL_CTX', '0')
    for coop in ('0', '1'):
        monkeypatch.setenv('MC_ROUTE_COOP', coop)              # read when the context is created
        ctx = nm.context(B, 24, max_steps=1)
        if regime == 'reverse_ties':
            ctx.set_tie_policy('reverse')
        ctx.enable_capture()
        ctx.set_timesteps([500])
        ctx.set_condition(xf.cuda(), mask.cuda())
        outs = [ctx.denoise(x.cuda(), 0).clone() for _ in range(3)]      # repeated: the barrier words and counts are left reusable
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        got[coop] = (outs[0], [ctx.routing(l) for l in range(dims['NL'])])
        ctx.close()
    monkeypatch.delenv('MC_ROUTE_SMALL_CTX')
    ctx = nm.context(B, 24, max_steps=1)                       # and the one-workgroup kernels the small config normally runs
    if regime == 'reverse_ties':
        ctx.set_tie_policy('reverse')
    ctx.set_timesteps([500])
    ctx.set_condition(xf.cuda(), mask.cuda())
    small = ctx.denoise(x.cuda(), 0).clone()
    ctx.close()
    assert torch.equal(got['0'][0], got['1'][0]) and torch.equal(got['1'][0], small)
    dropped = 0
    for (ia, ka), (ib, kb) in zip(got['0'][1], got['1'][1]):
        assert torch.equal(ia, ib) and torch.equal(ka, kb)
        dropped += int((~ka).sum())
    assert dropped > 0, 'the capacity cut never engaged'
    nm.close()


def test_cooperative_routing_admission_and_concurrent_contexts(full_model):
    """route_coop_k needs its whole grid resident, so the library reserves a context's workgroups out of the device's
    capacity (occupancy query x CUs) at mc_ctx_create and a context that does not fit falls back to the launch sequence
    by itself (no env var, no trap).  Four B=16 contexts (59 workgroups each) run their denoiser calls CONCURRENTLY on four
    streams, three rounds: every result equals the same context's result when run alone, and mc_ctx_check stays clean."""
    sd, nm = full_model
    B, T = 16, 196
    ctxs, inputs = [], []
    for k in range(4):
        x, xf, mask = synth_inputs(FULL, B, T, seed=300 + k, lengths=[T - 3 * j for j in range(B)])
        c = nm.context(B, T, max_steps=1)
        assert c.uses_coop_routing
        c.set_timesteps([400 + 100 * k])
        c.set_condition(xf.cuda(), mask.cuda())
        ctxs.append(c)
        inputs.append(x.cuda())
    alone = [c.denoise(x, 0).clone() for c, x in zip(ctxs, inputs)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in ctxs]
    for rnd in range(3):
        outs = []
        for c, x, st in zip(ctxs, inputs, streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(c.denoise(x, 0))
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        for k, (a, o) in enumerate(zip(alone, outs)):
            assert torch.equal(a, o), (rnd, k)
    for c in ctxs:
        c.check()
        c.close()


def test_cooperative_routing_falls_back_when_the_device_cannot_hold_the_grid(monkeypatch):
    """With the resident-workgroup capacity set to 3 (MC_ROUTE_COOP_SLOTS, read once per process -- so this runs in a child
    process), the first small-config context (2 workgroups) gets the cooperative kernel, the second does not fit and runs
    the 12-launch sequence; both give the same bits; closing the first frees its reservation for a third."""
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, 'tests')
from helpers import SMALL, SMALL_SEED, synth_inputs
from oracle import weights as W
from motioncraft_amd.engine import NativeModel
sd = W.make_state_dict(SMALL, SMALL_SEED)
nm = NativeModel(SMALL, sd, cfg_scale=SMALL['scale'])
x, xf, mask = synth_inputs(SMALL, 3, 24, seed=6, lengths=[24, 20, 7])
def run(c):
    c.set_timesteps([500]); c.set_condition(xf.cuda(), mask.cuda()); return c.denoise(x.cuda(), 0).clone()
a = nm.context(3, 24, max_steps=1); b = nm.context(3, 24, max_steps=1)
assert a.uses_coop_routing and not b.uses_coop_routing, (a.uses_coop_routing, b.uses_coop_routing)
ra, rb = run(a), run(b)
assert torch.equal(ra, rb)
a.check(); b.check(); a.close()
c = nm.context(3, 24, max_steps=1)
assert c.uses_coop_routing
assert torch.equal(run(c), ra)
print('fallback ok')
"""
    env = dict(os.environ, MC_ROUTE_COOP_SLOTS='3', MC_ROUTE_SMALL_CTX='0', PYTHONPATH=os.path.dirname(HERE))
    r = subprocess.run([sys.executable, '-c', code], cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'fallback ok' in r.stdout, r.stdout + r.stderr


def test_device_noise_stream_bits_and_statistics():
    """The fused sampler loop draws its per-step noise on the device (Philox4x32-10 + Box-Muller inside sampler_update_k).
    mc_op_philox_normal exposes the same stream: raw words bit-identical to oracle/philox_oracle.py (itself pinned by the
    published known-answer vectors), normals equal to the fp64 Box-Muller of those words within fp32 round-off, and the usual
    battery on 2^24 normals: moments, tail mass, Kolmogorov-Smirnov against the normal CDF, serial and cross-draw correlation."""
    from scipy import stats
    from motioncraft_amd import lib as L_
    from oracle import philox_oracle as P
    lib = L_.load(require_gpu=True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    seed, draw = 0x9e3779b97f4a7c15, (3 << 32) | 41
    for n in (1, 5, 4096 + 3):
        z = torch.full((n + 8,), float('nan'), device='cuda')
        bits = torch.zeros(n + 8, dtype=torch.int32, device='cuda')
        L_.check(lib.mc_op_philox_normal(ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(bits.data_ptr()), n, seed, draw, st))
        torch.cuda.synchronize()
        assert np.array_equal(bits[:n].cpu().numpy().view(np.uint32), P.draw_bits(n, seed, draw))
        assert int((bits[n:] != 0).sum()) == 0 and bool(torch.isnan(z[n:]).all())             # nothing past n
        ref = P.draw_normal(n, seed, draw, dtype=np.float64)
        assert np.abs(z[:n].cpu().numpy().astype(np.float64) - ref).max() <= 4e-6
    n = 1 << 24
    z = torch.empty(n, device='cuda')
    z2 = torch.empty(n, device='cuda')
    L_.check(lib.mc_op_philox_normal(ctypes.c_void_p(z.data_ptr()), None, n, seed, 0, st))
    L_.check(lib.mc_op_philox_normal(ctypes.c_void_p(z2.data_ptr()), None, n, seed, 1, st))
    torch.cuda.synchronize()
    a, b = z.double(), z2.double()
    se = 1.0 / n ** 0.5
    assert abs(float(a.mean())) < 5 * se and abs(float(a.var()) - 1) < 5 * 2 ** 0.5 * se
    assert abs(float((a ** 3).mean())) < 5 * 15 ** 0.5 * se and abs(float((a ** 4).mean()) - 3) < 5 * 96 ** 0.5 * se
    for k, p in ((2.0, 0.04550026), (3.0, 0.002699796), (4.0, 6.334248e-5)):
        frac = float((a.abs() > k).double().mean())
        assert abs(frac - p) < 5 * (p * (1 - p) / n) ** 0.5, (k, frac, p)
    assert float(a.abs().max()) < 6.8
    assert abs(float((a[:-1] * a[1:]).mean())) < 5 * se and abs(float((a[:-4] * a[4:]).mean())) < 5 * se      # inside / across blocks
    assert abs(float((a * b).mean())) < 5 * se                                                                  # draw 0 vs draw 1
    ks = stats.kstest(a[::16].cpu().numpy(), 'norm')
    assert ks.pvalue > 1e-3, ks
    print(f'device Philox normals, n = 2^24: mean {float(a.mean()):+.2e} var {float(a.var()):.5f} kurt {float((a ** 4).mean()):.4f} '
          f'max |z| {float(a.abs().max()):.2f} KS p = {ks.pvalue:.3f}')


@pytest.mark.parametrize('mode', ['ddpm', 'ddim'])
def test_fused_sampler_loop_equals_the_per_step_path(small_model, mode):
    """mc_sample_loop (the whole p_sample_loop / ddim_sample_loop in one C-ABI call, x updated in place) against one
    mc_sample_step per step from Python: bit-identical when both read the same noise tensors, and -- with the noise drawn
    inside the sampler kernel (Philox) -- bit-identical to the per-step path fed the draws mc_op_philox_normal writes out.
    Through the reference API: p_sample_loop / ddim_sample_loop(generator=...) run the fused loop and reproduce under the
    generator's seed; fused=False keeps the torch.randn per-step path."""
    from motioncraft_amd import lib as L_
    from motioncraft_amd.diffusion import build_diffusion
    sd, nm = small_model
    lib = L_.load(require_gpu=True)
    B, T, C = 3, 24, SMALL['input_feats']
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large',
                             **({'respace': '15,15,8,6,6'} if mode == 'ddim' else {})))
    x_T, xf, mask = synth_inputs(SMALL, B, T, seed=12, lengths=[24, 17, 9])
    ctx = nm.context(B, T, max_steps=len(d.timestep_map))
    ctx.set_timesteps(d.timestep_map)
    ctx.set_condition(xf.cuda(), mask.cuda())
    idx = list(range(len(d.timestep_map) - 1, len(d.timestep_map) - 13, -1)) if mode == 'ddpm' else list(range(49, -1, -1))
    coefs = [d.step_coefs(i, mode, SMALL['scale'], 0.0) for i in idx]
    g = torch.Generator().manual_seed(5)
    nz = torch.randn(len(idx), B, T, C, generator=g).cuda()

    def per_step(noises):
        x = x_T.cuda().clone()
        for k, i in enumerate(idx):
            x = ctx.sample_step(x, i, coefs[k], noises[k].contiguous())
        return x
    ref = per_step(nz)
    x = x_T.cuda().clone()
    x0 = torch.empty_like(x)
    ctx.sample_loop(x, idx, coefs, noise=nz, x0=x0)
    assert torch.equal(x, ref)
    assert bool(torch.isfinite(x0).all())
    # device noise: same bits as the per-step path on the written-out stream
    seed, draw0 = 77, 1000
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pz = torch.empty(len(idx), B, T, C, device='cuda')
    for k in range(len(idx)):
        L_.check(lib.mc_op_philox_normal(ctypes.c_void_p(pz[k].data_ptr()), None, B * T * C, seed, draw0 + k, st))
    x = x_T.cuda().clone()
    ctx.sample_loop(x, idx, coefs, noise=None, seed=seed, draw0=draw0)
    assert torch.equal(x, per_step(pz))
    ctx.close()

    # through the reference API
    class _M:      # the two things the loops ask of the model wrapper
        cfg_scale = SMALL['scale']

        def sampling_context(self, B_, T_, tmap, kw, dev):
            c = nm.context(B_, T_, max_steps=len(tmap))
            c.set_timesteps(tmap)
            c.set_condition(xf.cuda(), mask.cuda())
            return c
    loop = d.p_sample_loop if mode == 'ddpm' else d.ddim_sample_loop
    kw = dict(noise=x_T, num_steps=8, clip_denoised=False)
    a = loop(_M(), (B, T, C), generator=torch.Generator().manual_seed(3), **kw)
    b = loop(_M(), (B, T, C), generator=torch.Generator().manual_seed(3), **kw)
    c = loop(_M(), (B, T, C), generator=torch.Generator().manual_seed(4), **kw)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    if mode == 'ddpm':
        assert not torch.equal(a, c)                                # (eta = 0 DDIM does not use the draws)
    # fused with step_noise == per-step with step_noise
    sn = [torch.randn(B, T, C, generator=torch.Generator().manual_seed(100 + i)) for i in range(len(d.timestep_map))]
    f1 = loop(_M(), (B, T, C), step_noise=sn, **kw)
    f0 = loop(_M(), (B, T, C), step_noise=sn, fused=False, **kw)
    assert torch.equal(f1, f0)


@pytest.mark.parametrize('tag', ['hml263', 'kit251', 'smplx322'])
def test_sampler_entry_points_on_unaligned_shapes_vs_oracle(tag):
    """B*T*C not a multiple of 4 (B=1, T=25: 263 -> 6575, 251 -> 6275, 322 -> 8050 floats): the second partial product of the
    folded decoder tail (out2 + B*T*C) and the k-th noise slice (noise + k*n) are then 4- or 8-byte aligned only.  mc_sample_step,
    mc_sample_loop (host noise and device Philox noise) and the hipGraph replay must take the element-wise form of the sampler
    kernel and still equal the oracle / each other (the float4 kernel used to be the only one: ADVICE r03)."""
    from motioncraft_amd import lib as L_
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    kw = dict(max_seq_len=25, L=32, NL=2, F=64, Te=64, Dt=32, Nt=8)
    dims = {'hml263': W.humanml3d_dims(**kw), 'kit251': W.humanml3d_dims(input_feats=251, dataset='kit_ml', **kw),
            'smplx322': W.default_dims(**kw)}[tag]
    sd = W.make_state_dict(dims, SMALL_SEED)
    nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
    B, T, C = 1, 25, dims['input_feats']
    assert (B * T * C) % 4 != 0
    x_T, xf, mask = synth_inputs(dims, B, T, seed=44, lengths=[21])
    d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large'))
    ctx = nm.context(B, T, max_steps=1000)
    ctx.set_timesteps(d.timestep_map)
    ctx.set_condition(xf.cuda(), mask.cuda())
    idx = [999, 998, 997, 996, 995]
    coefs = [d.step_coefs(i, 'ddpm', dims['scale']) for i in idx]
    g = torch.Generator().manual_seed(6)
    nz = torch.randn(len(idx), B, T, C, generator=g)
    ref = O.sample_loop(sd, dims, O.Schedule(1000, None), 'ddpm', x_T, xf, mask, step_noise=lambda i: nz[999 - i], num_steps=len(idx))
    x = x_T.cuda().clone()
    for k, i in enumerate(idx):                                    # per-step entry (noise slices individually allocated: aligned)
        x = ctx.sample_step(x, i, coefs[k], nz[k].cuda().contiguous())
    assert maxabs(x, ref) <= TOL_FINAL, tag
    xl = x_T.cuda().clone()
    x0 = torch.empty_like(xl)
    ctx.sample_loop(xl, idx, coefs, noise=nz.cuda(), x0=x0)        # loop entry: noise + k*n is misaligned for k = 1, 2, 3
    assert torch.equal(xl, x) and bool(torch.isfinite(x0).all())
    # device noise == the written-out Philox stream through the per-step path
    lib = L_.load(require_gpu=True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    pz = [torch.empty(B, T, C, device='cuda') for _ in idx]
    for k in range(len(idx)):
        L_.check(lib.mc_op_philox_normal(ctypes.c_void_p(pz[k].data_ptr()), None, B * T * C, 9, 40 + k, st))
    xa = x_T.cuda().clone()
    ctx.sample_loop(xa, idx, coefs, noise=None, seed=9, draw0=40)
    xb = x_T.cuda().clone()
    for k, i in enumerate(idx):
        xb = ctx.sample_step(xb, i, coefs[k], pz[k])
    assert torch.equal(xa, xb)
    ctx.close()
    nm.close()


def test_control_branch_without_condition_cfg_vs_oracle():
    """condition_encode_cfg.condition_cfg=False: the control condition also drives the unconditional CFG half
    (controlnet.py forward_test: `c * cond_type` only when condition_cfg)."""
    from motioncraft_amd.engine import NativeModel
    from oracle import stmogen_oracle as O, weights as W
    g = load('control_small.npz')
    sd = W.make_state_dict(CTRL, SMALL_SEED, shapes=W.control_param_shapes(CTRL, CTRL_COPY, CTRL_FEATS))
    x, xf, mask, c = (T_(g[k]) for k in ('x_t', 'xf_out', 'motion_mask', 'c'))
    outs = {}
    for ccfg in (True, False):
        nm = NativeModel(CTRL, sd, cfg_scale=CTRL['scale'], condition_cfg=ccfg)
        ctx = nm.context(2, 24, max_steps=1)
        ctx.set_timesteps([640])
        ctx.set_condition(xf.cuda(), mask.cuda())
        ctx.set_control(c.cuda())
        out2 = ctx.denoise(x.cuda(), 0)
        w = (1 - (1000 - 640) / 1000) * CTRL['scale'] + 1
        outs[ccfg] = (out2[:2] * w + out2[2:] * (1 - w)).cpu()
        ref = O.denoise_control(sd, CTRL, x, 640, xf, mask, c, CTRL_COPY, condition_cfg=ccfg)
        assert maxabs(outs[ccfg], ref) <= TOL_STEP, ccfg
        ctx.close()
        nm.close()
    assert maxabs(outs[True], T_(g['x0_t640'])) <= TOL_STEP
    assert maxabs(outs[True], outs[False]) > 1e-3          # the flag matters


def _eval_model_dir(tmp_path, vocab, bert):
    import json
    (tmp_path / 'vocab.txt').write_text('\n'.join(vocab) + '\n', encoding='utf-8')
    (tmp_path / 'config.json').write_text(json.dumps(dict(bert, vocab_size=len(vocab), model_type='distilbert')))
    (tmp_path / 'tokenizer_config.json').write_text(json.dumps(dict(do_lower_case=True)))
    return str(tmp_path)


def test_evaluation_embedding_model_vs_reference_golden_and_oracle(tmp_path):
    """T2MContrastiveModel_SMPLX built through the registry with the configs' evaluator_model layout;
    encode_motion / encode_text (own WordPiece tokenizer + DistilBERT + VAE-token transformer on the device) vs the
    reference's outputs, then the shipped sizes (latent 256, 4 layers, DistilBERT-base widths, T=196) vs the oracle."""
    import motioncraft_amd as mc
    from motioncraft_amd import evaluation as E
    from helpers import EVAL_BERT, EVAL_DIMS
    from oracle import eval_encoder_oracle as EO, weights as W
    g = load('evaluator.npz')
    vocab = [str(v) for v in g['vocab']]
    shapes = W.eval_encoder_param_shapes(bert=dict(EVAL_BERT, vocab_size=len(vocab)), **EVAL_DIMS)
    sd = W.make_eval_encoder_state(shapes, seed=int(g['seed']))
    enc_cfg = {k: v for k, v in EVAL_DIMS.items() if k != 'nfeats'}
    model = mc.build_submodule(dict(type='T2MContrastiveModel_SMPLX',
                                    motion_encoder=dict(nfeats=EVAL_DIMS['nfeats'], vae=True, **enc_cfg),
                                    text_encoder=dict(modelpath=_eval_model_dir(tmp_path, vocab, EVAL_BERT), **enc_cfg),
                                    state_dict=sd))
    mu = model.encode_motion(T_(g['motion']).cuda(), torch.from_numpy(g['lengths']).cuda())
    tu = model.encode_text([str(t) for t in g['texts']], device='cuda')
    e1, e2 = maxabs(mu, T_(g['motion_mu'])), maxabs(tu, T_(g['text_mu']))
    print(f'evaluation encoders (reduced): |hip - reference| motion {e1:.2e}, text {e2:.2e}')
    assert e1 <= 5e-5 and e2 <= 5e-5
    # frames past the length and padded tokens never reach the embedding
    m2 = T_(g['motion']).clone()
    m2[2, 9:] = 123.0
    assert torch.equal(model.encode_motion(m2.cuda(), torch.from_numpy(g['lengths']).cuda())[2], mu[2])
    ids2 = torch.from_numpy(g['input_ids']).clone()
    ids2[torch.from_numpy(g['attention_mask']) == 0] = 7
    t2 = model.encode_text(None, input_ids=ids2, attention_mask=torch.from_numpy(g['attention_mask']), device='cuda')
    assert torch.equal(t2, tu)
    # motion side alone: the text entry fails loudly
    only_motion = E.NativeEvalEncoder({k: v for k, v in sd.items() if k.startswith('motionencoder.')}, **EVAL_DIMS)
    with pytest.raises(RuntimeError):
        only_motion.encode_tokens(ids2.cuda(), torch.from_numpy(g['attention_mask']).cuda())
    only_motion.close()

    # shipped sizes (motionx_bs128.py:38-51; DistilBERT-base widths with a 2000-piece vocabulary)
    bert = dict(dim=768, n_layers=6, n_heads=12, hidden_dim=3072, max_position_embeddings=512, vocab_size=2000)
    shapes = W.eval_encoder_param_shapes(bert=bert)
    sd = W.make_eval_encoder_state(shapes, seed=8)
    enc = E.NativeEvalEncoder(sd, bert=bert)
    gen = torch.Generator().manual_seed(12)
    B, T, S = 8, 196, 40
    motion = torch.randn(B, T, 322, generator=gen)
    lengths = [196, 150, 64, 63, 1, 196, 100, 12]
    ids = torch.randint(0, 2000, (B, S), generator=gen)
    tl = torch.tensor([40, 33, 5, 2, 17, 40, 21, 9])
    mask = (torch.arange(S)[None] < tl[:, None]).to(torch.uint8)
    mu = enc.encode_motion(motion.cuda(), lengths)
    tu = enc.encode_tokens(ids.cuda(), mask.cuda())
    e1 = maxabs(mu, EO.encode_motion(sd, motion, lengths))
    e2 = maxabs(tu, EO.encode_text_tokens(sd, ids, mask, 6, 12))
    print(f'evaluation encoders (shipped sizes): |hip - oracle| motion {e1:.2e}, text {e2:.2e}')
    assert e1 <= 1e-4 and e2 <= 1e-4
    enc.close()


def test_fid_and_precision_evaluators_on_device_embeddings(tmp_path):
    """The evaluators of mogen/core/evaluation driven by the device embedding model: metrics equal the same formulas
    applied to the oracle's embeddings of the same result list."""
    import motioncraft_amd as mc
    from motioncraft_amd import evaluation as E
    from helpers import EVAL_BERT, EVAL_DIMS
    from oracle import eval_encoder_oracle as EO, weights as W
    g = load('evaluator.npz')
    vocab = [str(v) for v in g['vocab']]
    sd = W.make_eval_encoder_state(W.eval_encoder_param_shapes(bert=dict(EVAL_BERT, vocab_size=len(vocab)), **EVAL_DIMS), seed=3)
    enc_cfg = {k: v for k, v in EVAL_DIMS.items() if k != 'nfeats'}
    model = mc.build_submodule(dict(type='T2MContrastiveModel_SMPLX', motion_encoder=dict(nfeats=322, vae=True, **enc_cfg),
                                    text_encoder=dict(modelpath=_eval_model_dir(tmp_path, vocab, EVAL_BERT), **enc_cfg), state_dict=sd))
    gen = torch.Generator().manual_seed(4)
    N, T = 192, 24           # more samples than embedding dimensions: the covariances stay well conditioned
    words = [w for w in vocab[57:] if not w.startswith('##') and w.isalpha()]
    results = []
    for i in range(N):
        n = int(torch.randint(8, T + 1, (1,), generator=gen))
        msk = (torch.arange(T) < n).float()
        mot = torch.randn(T, 322, generator=gen) * msk[:, None]
        text = ' '.join(words[int(j)] for j in torch.randint(0, len(words), (5,), generator=gen))
        results.append(dict(motion=mot, pred_motion=mot + 0.5 * torch.randn(T, 322, generator=gen) * msk[:, None], motion_mask=msk,
                            pred_motion_mask=msk, motion_length=torch.tensor(n), pred_motion_length=torch.tensor(n), text=text))
    common = dict(data_len=N, replication_times=1, evaluator_model=model)
    got = {}
    got.update(E.FIDEvaluator(emb_scale=1.0, **common).evaluate(results))
    got.update(E.PrecisionEvaluator(top_k=3, batch_size=32, **common).evaluate(results))
    got.update(E.MatchingScoreEvaluator(batch_size=32, **common).evaluate(results))
    lens = [int(r['motion_length']) for r in results]
    pm = EO.encode_motion(sd, torch.stack([r['pred_motion'] for r in results]), lens, 2, 2).numpy()
    gm = EO.encode_motion(sd, torch.stack([r['motion'] for r in results]), lens, 2, 2).numpy()
    ids, mask = model.tokenizer([r['text'] for r in results])
    tm = EO.encode_text_tokens(sd, torch.from_numpy(ids).long(), torch.from_numpy(mask), 2, 2, 2, 2).numpy()
    fid = EO.fid(pm, gm)
    hits = sum(EO.r_precision_counts(tm[i:i + 32], pm[i:i + 32]) for i in range(0, N, 32)) / N
    match = sum(EO.matching_score_sum(tm[i:i + 32], pm[i:i + 32]) for i in range(0, N, 32)) / N
    print(f"device evaluators: FID {got['FID (mean)']:.5f} (oracle {fid:.5f}), R-precision {[got['R_precision Top %d (mean)' % k] for k in (1, 2, 3)]}"
          f" (oracle {hits.tolist()}), matching score {got['Matching Score (mean)']:.5f} (oracle {match:.5f})")
    assert abs(got['FID (mean)'] - fid) <= 1e-3 * max(1.0, abs(fid))
    assert abs(got['Matching Score (mean)'] - match) <= 1e-3
    assert np.allclose([got['R_precision Top %d (mean)' % k] for k in (1, 2, 3)], hits, atol=1.5 / N)
    assert got['FID (conf)'] == 0


def test_t2m_bigru_evaluator_vs_reference_golden_and_oracle():
    """HumanML3D / KIT evaluator (T2MContrastiveModel, t2m_bigru.py) through mc_t2meval_*: conv-as-GEMM movement encoder and
    BiGRU heads vs the reference's outputs (reduced widths), then the shipped widths (263-d, 512 / 1024 / 512; 196 frames;
    word_size 300, hidden 512, 22 tokens) vs the oracle; word vectors past a sentence's length do not matter."""
    import motioncraft_amd as mc
    from motioncraft_amd import evaluation as E
    from helpers import T2M_DIMS, T2M_TEXT
    from oracle import t2m_eval_oracle as TO, weights as W
    g = load('t2m_evaluator.npz')
    sd = W.make_t2m_eval_state(W.t2m_eval_param_shapes(**T2M_DIMS, **T2M_TEXT), seed=int(g['seed']))
    model = mc.build_submodule(dict(type='T2MContrastiveModel', motion_encoder=dict(T2M_DIMS), text_encoder=dict(T2M_TEXT, max_text_len=10),
                                    state_dict=sd))
    me = model.encode_motion(T_(g['motion']).cuda(), torch.from_numpy(g['lengths']).cuda())
    te = model.encode_text(None, word_emb=T_(g['word_emb']), pos_onehot=T_(g['pos_onehot']), sent_len=torch.from_numpy(g['sent_len']))
    e1, e2 = maxabs(me, T_(g['motion_emb'])), maxabs(te, T_(g['text_emb']))
    print(f't2m evaluator (reduced): |hip - reference| motion {e1:.2e}, text {e2:.2e}')
    assert e1 <= 5e-5 and e2 <= 5e-5
    w2 = T_(g['word_emb']).clone()                           # (motion: the conv windows do reach past the length, as in the reference)
    w2[3, 3:] = -9.0
    assert torch.equal(model.encode_text(None, word_emb=w2, pos_onehot=T_(g['pos_onehot']), sent_len=torch.from_numpy(g['sent_len']))[3], te[3])
    with pytest.raises(ValueError):
        model.encode_motion(T_(g['motion']).cuda(), torch.tensor([3, 40, 8, 21]))
    with pytest.raises(RuntimeError):
        model.encode_text(['a'], token=['walk/VERB'])            # no word-vector lookup given
    shapes = W.t2m_eval_param_shapes()
    sd = W.make_t2m_eval_state(shapes, seed=2)
    enc = E.NativeT2MEvaluator(sd)
    gen = torch.Generator().manual_seed(6)
    B, T, S = 6, 196, 22
    motion = torch.randn(B, T, 263, generator=gen)
    lengths = torch.tensor([196, 120, 64, 7, 4, 199 - 3])
    word = torch.randn(B, S, 300, generator=gen)
    pos = torch.nn.functional.one_hot(torch.randint(0, 15, (B, S), generator=gen), 15).float()
    sent = torch.tensor([22, 1, 9, 15, 3, 22])
    me, te = enc.encode_motion(motion.cuda(), lengths), enc.encode_word_vectors(word.cuda(), pos.cuda(), sent)
    e1, e2 = maxabs(me, TO.encode_motion(sd, motion, lengths)), maxabs(te, TO.encode_text(sd, word, pos, sent))
    print(f't2m evaluator (shipped widths): |hip - oracle| motion {e1:.2e}, text {e2:.2e}')
    assert e1 <= 1e-4 and e2 <= 1e-4
    enc.close()
