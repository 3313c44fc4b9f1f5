"""Sampler host side: schedule tables + DDPM / DDIM loops driving the native denoiser.

Mirrors the sampling half of the reference's ``mogen/models/utils/gaussian_diffusion.py`` API
(``get_named_beta_schedule`` :235-260, ``space_timesteps`` :1346-1404, ``GaussianDiffusion`` tables
:336-387, ``p_sample_loop`` :698-797, ``ddim_sample_loop`` :925-1049, ``SpacedDiffusion`` :1407-1448)
so callers written against it keep working; the per-step arithmetic itself (network, CFG combine,
posterior / DDIM update) runs in libmotioncraft_amd.so.  The RePaint / outpainting mode of the DDIM loop
(``y = {gt, outpainting_mask}``: :492-501, :855-877, the resampling ``harmonize`` loop :1050-1118 and its jump
schedule ``mogen/models/utils/scheduler.py:178-208``) is covered too.  Training losses, learned variances and
cond_fn guidance are outside this path and raise loudly.
"""
import enum
import math

import numpy as np
import torch

from . import lib as _lib


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    if schedule_name == 'linear':
        k = 1000 / num_diffusion_timesteps
        return np.linspace(k * 0.0001, k * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == 'cosine':
        f = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        n = num_diffusion_timesteps
        return np.array([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)], dtype=np.float64)
    raise NotImplementedError(f'unknown beta schedule: {schedule_name}')


def space_timesteps(num_timesteps, section_counts):
    """Retained original timesteps for a respacing spec ('15,15,8,6,6', [..] or 'ddimN')."""
    if isinstance(section_counts, str):
        if section_counts.startswith('ddim'):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f'cannot create exactly {num_timesteps} steps with an integer stride')
        section_counts = [int(v) for v in section_counts.split(',')]
    base, extra = divmod(num_timesteps, len(section_counts))
    kept, start = [], 0
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f'cannot divide section of {size} steps into {count}')
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type=None, rescale_timesteps=False,
                 opt=None):
        self.opt = opt
        self.model_mean_type, self.model_var_type, self.loss_type = model_mean_type, model_var_type, loss_type
        self.rescale_timesteps = rescale_timesteps
        betas = np.array(betas, dtype=np.float64)
        if betas.ndim != 1 or not ((betas > 0).all() and (betas <= 1).all()):
            raise ValueError('betas must be a 1-D array in (0, 1]')
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        if not hasattr(self, 'timestep_map'):
            self.timestep_map = list(range(self.num_timesteps))
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        self.alphas_cumprod = ac
        self.alphas_cumprod_prev = np.append(1.0, ac[:-1])
        self.alphas_cumprod_next = np.append(ac[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(ac)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - ac)

    # ------------------------------------------------------------------------------------------
    def _check_supported(self, clip_denoised, denoised_fn, cond_fn, model_kwargs, pre_seq=None, transl_req=None):
        if self.model_mean_type != ModelMeanType.START_X:
            raise NotImplementedError('the MI355X path implements model_mean_type="start_x" (all stmogen configs)')
        if self.model_var_type != ModelVarType.FIXED_LARGE:
            raise NotImplementedError('the MI355X path implements model_var_type="fixed_large" (all stmogen configs)')
        if clip_denoised or denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError('clip_denoised / denoised_fn / cond_fn are not on the MotionDiffusion eval path '
                                      '(diffusion_architecture.py:177-191 passes clip_denoised=False)')

    def _inpaint_operands(self, mode, model_kwargs, shape, device):
        """y = {gt, outpainting_mask} of the long-sequence windows (tools/m2d_test.py:177-195) -> device operands,
        or None when the mode is off (no mask / all-False mask, like the reference's ``True in mask`` tests)."""
        y = (model_kwargs or {}).get('y', {}) or {}
        if 'outpainting_mask' not in y:
            return None
        keep = y['outpainting_mask']
        if keep.dtype != torch.bool or tuple(keep.shape) != tuple(shape):
            raise AssertionError('outpainting_mask must be a bool tensor of the shape of the sample')
        if not bool(keep.any()):
            return None
        if 'gt' not in y:
            if mode == 'ddpm':
                return None                 # p_mean_variance :493-496 needs both keys, p_sample has no other use
            raise KeyError('gt')
        if tuple(y['gt'].shape) != tuple(shape):
            raise AssertionError('gt must have the shape of the sample')
        if mode == 'ddim' and self.opt is None:
            raise ValueError("the outpainting mode of ddim_sample reads opt.overlap_len / opt.addBlend: build the "
                             "architecture with cfg.model['opt'] = args as the reference tools do")
        return dict(keep=keep.to(device).contiguous(), gt=y['gt'].to(device=device, dtype=torch.float32).contiguous())

    def step_coefs(self, i, mode, scale, eta=0.0):
        """fp64 tables -> fp32 scalars exactly like _extract_into_tensor(...).float()."""
        t_orig = self.timestep_map[i]
        w = (1 - (1000 - int(t_orig)) / 1000) * scale + 1
        c = _lib.StepCoefs()
        c.mode = 0 if mode == 'ddpm' else 1
        c.text_coef, c.none_coef = w, 1 - w
        c.c1, c.c2 = self.posterior_mean_coef1[i], self.posterior_mean_coef2[i]
        c.log_var = math.log(self.posterior_variance[1] if i == 0 else self.betas[i])
        c.sqrt_recip, c.sqrt_recipm1 = self.sqrt_recip_alphas_cumprod[i], self.sqrt_recipm1_alphas_cumprod[i]
        c.ab, c.ab_prev, c.eta = self.alphas_cumprod[i], self.alphas_cumprod_prev[i], eta
        c.nonzero = 0.0 if i == 0 else 1.0
        return c

    def _loop(self, mode, model, shape, noise, model_kwargs, device, progress, eta, step_noise, generator,
              num_steps=None, trajectory=None, pre_seq=None, transl_req=None, graph=False, fused=True):
        if model_kwargs is None:
            model_kwargs = {}
        if not isinstance(shape, (tuple, list)):
            raise AssertionError('shape must be a tuple or list')
        B, T, C = shape
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        inp = self._inpaint_operands(mode, model_kwargs, shape, device)
        ctx = model.sampling_context(B, T, self.timestep_map, model_kwargs, device)
        if noise is not None:
            img = noise.to(device=device, dtype=torch.float32).contiguous().clone()
        else:
            img = torch.randn(*shape, device=device, generator=generator)
        # the schedule: (i, True) = denoise at spaced index i, (i, False) = forward "undo" step with beta[i]
        if inp is not None and mode == 'ddim' and not getattr(self.opt, 'no_repaint', False):
            if num_steps is not None:
                raise ValueError('num_steps truncation is not defined for the resampling schedule')
            n_ddim = int(str(self.opt.timestep_respacing)[4:])
            if getattr(self.opt, 'no_resample', False):
                times = get_schedule_jump_cjm_ddim(n_ddim)
            else:
                times = get_schedule_jump_cjm_ddim(n_ddim, jump_length=self.opt.jump_length,
                                                   jump_n_sample=self.opt.jump_n_sample)
            if max(times) >= self.num_timesteps:
                raise ValueError(f'opt.timestep_respacing={self.opt.timestep_respacing!r} starts the resampling loop at '
                                 f'step {max(times)} of a {self.num_timesteps}-step schedule')
            plan = [(a, b < a) for a, b in zip(times[:-1], times[1:])]
        else:
            indices = list(range(self.num_timesteps))[::-1]
            if num_steps is not None:
                indices = indices[:num_steps]
            plan = [(i, True) for i in indices]
        if progress:
            from tqdm.auto import tqdm
            plan = tqdm(plan)
        seeded = pre_seq is not None or bool(transl_req)
        if seeded:
            # p_sample :664-674 / ddim_sample :816-820: every step re-noises pre_seq (and the requested translation
            # channels of frames 0-1) to the step's level and writes it over the first frames of x before the network call
            if inp is not None:
                raise NotImplementedError('pre_seq / transl_req together with the outpainting mask: no reference tool combines them')
            if pre_seq is not None:
                pre_seq = pre_seq.to(device=device, dtype=torch.float32).contiguous()
                if pre_seq.dim() != 3 or pre_seq.shape[0] != B or pre_seq.shape[2] != C or pre_seq.shape[1] > T:
                    raise RuntimeError(f'pre_seq of shape {tuple(pre_seq.shape)} cannot be written into x[:, :T, :] of {tuple(shape)}')
            transl_req = [list(it) for it in (transl_req or [])]
            if transl_req and B > 2:
                # _extract_into_tensor(arr, t, (2,)) expands a [B] tensor to (2,): the reference raises for B > 2 as well
                raise RuntimeError(f'transl_req: the expanded size of the tensor (2) must match the batch size ({B})')
            if any(len(it) != 3 for it in transl_req):
                raise ValueError('transl_req items are [channel, value_frame0, value_frame1]')
            if step_noise is not None and not hasattr(step_noise, '__next__'):
                raise ValueError('with pre_seq / transl_req several tensors of different shapes are drawn per step: '
                                 'step_noise must be an iterator yielding them in the order the reference draws them '
                                 '(randn_like(pre_seq), one randn(2) per transl_req item, randn_like(x))')
        draw_no = [0]
        if inp is not None and step_noise is not None and not (hasattr(step_noise, '__next__') or callable(step_noise)):
            raise ValueError('the outpainting mode draws several independent randn_like tensors per step index: '
                             'step_noise must be an iterator or a callable of the running draw number there, not a '
                             'sequence indexed by step (every draw of a step would get the same tensor)')

        def draw(i):
            """the next randn_like(x) of the reference loop; in the outpainting mode several are drawn per step,
            so there ``step_noise`` is an iterator (or a callable of the running draw number)."""
            n = draw_no[0]
            draw_no[0] += 1
            if step_noise is None:
                return torch.randn(*shape, device=device, generator=generator)
            if hasattr(step_noise, '__next__'):
                e = next(step_noise)
            elif callable(step_noise):
                e = step_noise(n if inp is not None else i)
            else:
                e = step_noise[i]
            return e.to(device=device, dtype=torch.float32).contiguous()

        blend_w = None
        if inp is not None and mode == 'ddim':
            ov = int(self.opt.overlap_len)
            if not 0 <= ov <= T:
                raise ValueError(f'opt.overlap_len={ov} outside the {T}-frame window')
            blend_w = torch.linspace(0, 1, ov, device=device) if ov > 0 else None
        nxt = torch.empty_like(img)
        x0 = torch.empty_like(img) if trajectory is not None else None
        if graph:
            # hipGraph replay (BASELINE configs[4]): one captured graph of the step for the whole schedule, x updated in
            # place, the per-step noise refilled in place; bit-identical to the eager sequence, 2-3 % faster for B <= 4
            # (profiles/r02_graph_vs_eager.txt)
            if inp is not None or seeded or trajectory is not None:
                raise ValueError('graph replay covers the plain p_sample / ddim_sample step (no outpainting, seeding or trajectory)')
            # the graph (and the x / noise buffers baked into it) lives on the context and is reused by later calls with the
            # same schedule: capture + instantiate cost a few ms, more than one small-batch loop gains
            key = (mode, float(eta), float(model.cfg_scale), tuple(int(t) for t in self.timestep_map))
            st = getattr(ctx, '_graph_state', None)
            if st is None or st['key'] != key:
                side = torch.cuda.Stream(device=device)
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    gx, nbuf = torch.empty_like(img), torch.empty_like(img)
                    coefs = [self.step_coefs(j, mode, model.cfg_scale, eta) for j in range(self.num_timesteps)]
                    ctx.graph_capture(gx, nbuf, coefs)
                st = ctx._graph_state = dict(key=key, stream=side, x=gx, noise=nbuf)
            side = st['stream']
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                st['x'].copy_(img)
                for i, _ in plan:
                    st['noise'].copy_(draw(i))
                    ctx.graph_step(i)
                img.copy_(st['x'])
            torch.cuda.current_stream(device).wait_stream(side)
            if getattr(ctx, 'uses_coop_routing', False):
                ctx.check()
            return img
        if (fused and inp is None and not seeded and trajectory is None and not progress and all(d for _, d in plan)):
            # the whole loop inside the library (mc_sample_loop): no return to Python between steps.  Without step_noise the
            # per-step randn_like is drawn INSIDE the sampler-update kernel (Philox4x32-10; the loop's 64-bit key is one draw from
            # `generator`, so a seeded generator reproduces the run and successive calls differ); with step_noise the draws are
            # gathered into [steps, B, T, C] chunks of at most ~512 MB and the same entry point reads them
            idx = [i for i, _ in plan]
            coefs = [self.step_coefs(i, mode, model.cfg_scale, eta) for i in idx]
            if step_noise is None:
                gdev = generator.device if generator is not None else torch.device('cpu')
                key = int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=gdev).item())
                ctx.sample_loop(img, idx, coefs, noise=None, seed=key, draw0=0)
            else:
                chunk = max(1, int((512 << 20) // (4 * B * T * C)))
                for k0 in range(0, len(idx), chunk):
                    part = idx[k0:k0 + chunk]
                    nz = torch.stack([draw(i) for i in part])
                    ctx.sample_loop(img, part, coefs[k0:k0 + chunk], noise=nz)
            if getattr(ctx, 'uses_coop_routing', False):
                ctx.check()
            return img
        for i, denoise in plan:
            if not denoise:                                                   # _undo (:429-435)
                beta = np.float32(self.betas[i])
                ctx.renoise(img, draw(i), np.sqrt(np.float32(1) - beta), np.sqrt(beta), out=nxt)
            elif seeded:
                a_, b_ = np.float32(self.sqrt_alphas_cumprod[i]), np.float32(self.sqrt_one_minus_alphas_cumprod[i])
                pre_noise = None
                if pre_seq is not None:
                    pre_noise = (next(step_noise).to(device=device, dtype=torch.float32).contiguous() if step_noise is not None
                                 else torch.randn(*pre_seq.shape, device=device, generator=generator))
                transl = []
                for it in transl_req:                                        # th.randn(2) on the host, then q_sample in fp32
                    n2 = (next(step_noise) if step_noise is not None else torch.randn(2, generator=generator if generator is not None and generator.device.type == 'cpu' else None))
                    n2 = n2.detach().cpu().numpy().astype(np.float32)
                    v = a_ * np.asarray(it[1:], dtype=np.float32) + b_ * n2
                    transl.append((int(it[0]), float(v[0]), float(v[1])))
                eps = draw(i)
                ctx.sample_step_seeded(img, i, self.step_coefs(i, mode, model.cfg_scale, eta), eps, a_, b_, pre_seq=pre_seq,
                                       pre_noise=pre_noise, transl=transl, x_prev=nxt, x0=x0)
            elif inp is None:
                eps = draw(i)                                                 # drawn every step, DDIM too
                ctx.sample_step(img, i, self.step_coefs(i, mode, model.cfg_scale, eta), eps, x_prev=nxt, x0=x0)
            else:
                eps = draw(i)
                gt_noise, blend_len = None, 0
                if mode == 'ddim':
                    gt_noise = draw(i)                                        # :868
                    noise_weight = np.sqrt(np.float32(1) - np.float32(self.alphas_cumprod_prev[i]))
                    if noise_weight < np.float32(0.2) and getattr(self.opt, 'addBlend', True) and blend_w is not None:
                        blend_len = int(self.opt.overlap_len)                 # :872-875
                ctx.sample_step_inpaint(img, i, self.step_coefs(i, mode, model.cfg_scale, eta), eps, inp['gt'],
                                        inp['keep'], gt_noise=gt_noise, blend_w=blend_w, blend_len=blend_len,
                                        x_prev=nxt, x0=x0)
            img, nxt = nxt, img
            if trajectory is not None:
                trajectory.append((i, img.clone(), x0.clone()))
        if getattr(ctx, 'uses_coop_routing', False):
            ctx.check()            # one sync after the last step: a timed-out grid barrier of the routing kernel raises here
        return img

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, pre_seq=None, transl_req=None, progress=False,
                      step_noise=None, generator=None, num_steps=None, trajectory=None, graph=False, fused=True):
        """gaussian_diffusion.py:698-797.  Extras beside the reference's arguments: ``step_noise`` (the per-step draws, for runs on
        the reference's seeds), ``generator``, ``num_steps``, ``trajectory``, ``graph`` (hipGraph replay) and ``fused`` (default:
        the whole loop is ONE library call, mc_sample_loop, with the per-step noise drawn on the device when no step_noise is
        given; False: one mc_sample_step per step with torch.randn draws)."""
        self._check_supported(clip_denoised, denoised_fn, cond_fn, model_kwargs, pre_seq, transl_req)
        return self._loop('ddpm', model, shape, noise, model_kwargs, device, progress, 0.0, step_noise, generator,
                          num_steps, trajectory, pre_seq=pre_seq, transl_req=transl_req, graph=graph, fused=fused)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, pre_seq=None,
                         step_noise=None, generator=None, num_steps=None, trajectory=None, graph=False, fused=True):
        self._check_supported(clip_denoised, denoised_fn, cond_fn, model_kwargs, pre_seq)
        if self.opt is not None and getattr(self.opt, 'same_overlap_noisy', False):
            raise NotImplementedError('opt.same_overlap_noisy: the reference writes self.saved_noisy_tail '
                                      '(gaussian_diffusion.py:879-881) without ever creating it, so that option has '
                                      'no defined behaviour to reproduce')
        return self._loop('ddim', model, shape, noise, model_kwargs, device, progress, float(eta), step_noise,
                          generator, num_steps, trajectory, pre_seq=pre_seq, graph=graph, fused=fused)


def get_schedule_jump_cjm_ddim(time_respacing=25, jump_length=1, jump_n_sample=1):
    """Visit order of the resampling ("harmonize") DDIM loop, mirrors mogen/models/utils/scheduler.py:178-208:
    start 60 % into the schedule (step 15 of 25), walk down to 0, and the first jump_n_sample-1 times a multiple
    of jump_length (below t_T - jump_length) is reached, walk jump_length steps back up first.  Ends with -1."""
    t_T = 15 if time_respacing == 25 else int(time_respacing * 0.6)
    revisits = dict.fromkeys(range(0, t_T - jump_length, jump_length), jump_n_sample - 1)
    ts, t = [], t_T - 1
    while t >= 0:
        ts.append(t)
        if revisits.get(t, 0) > 0:
            revisits[t] -= 1
            ts.extend(range(t + 1, t + jump_length + 1))
            t += jump_length
        t -= 1
    ts.append(-1)
    if any(abs(a - b) != 1 for a, b in zip(ts[:-1], ts[1:])) or ts[0] >= t_T:
        raise AssertionError('inconsistent jump schedule')
    return ts


class SpacedDiffusion(GaussianDiffusion):
    """Skips steps of a base process: betas re-derived from the retained alpha-bars; the network is
    fed ``timestep_map[i]`` (the original timestep), as the reference's ``_WrappedModel`` does."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs['betas'])
        base_ac = np.cumprod(1.0 - np.array(kwargs['betas'], dtype=np.float64), axis=0)
        self.timestep_map, new_betas, last = [], [], 1.0
        for i, a in enumerate(base_ac):
            if i in self.use_timesteps:
                new_betas.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        kwargs['betas'] = np.array(new_betas)
        super().__init__(**kwargs)


def build_diffusion(cfg, opt=None):
    """cfg = diffusion_test / diffusion_train dict of the configs (diffusion_architecture.py:25-54)."""
    betas = get_named_beta_schedule(cfg['beta_scheduler'], cfg['diffusion_steps'])
    mean_type = {'start_x': ModelMeanType.START_X, 'previous_x': ModelMeanType.PREVIOUS_X,
                 'epsilon': ModelMeanType.EPSILON}[cfg['model_mean_type']]
    var_type = {'learned': ModelVarType.LEARNED, 'fixed_small': ModelVarType.FIXED_SMALL,
                'fixed_large': ModelVarType.FIXED_LARGE, 'learned_range': ModelVarType.LEARNED_RANGE}[cfg['model_var_type']]
    if cfg.get('respace', None) is not None:
        return SpacedDiffusion(use_timesteps=space_timesteps(cfg['diffusion_steps'], cfg['respace']), betas=betas,
                               model_mean_type=mean_type, model_var_type=var_type, loss_type=LossType.MSE, opt=opt)
    return GaussianDiffusion(betas=betas, model_mean_type=mean_type, model_var_type=var_type, loss_type=LossType.MSE)
