"""Thin Python owner of the native handles: ``NativeModel`` (weights in HBM) and ``NativeContext``
(workspace + schedule tables + condition K/V for one (batch, frames) shape).

PyTorch is used only for device memory and the stream handle; every compute step is a call into
libmotioncraft_amd.so.
"""
import ctypes

import numpy as np
import torch

from . import lib as _lib
from .weights import control_info, pack_state_dict


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f'{name} must be a contiguous float32 tensor in device (HBM) memory')
    return t


class NativeModel:
    def __init__(self, dims, state_dict, cfg_scale=6.5, capacity_factor=1.5, dyn_heads=8, device=None,
                 condition_cfg=True):
        self.lib = _lib.load(require_gpu=True)
        if device is not None:
            torch.cuda.set_device(device)
            _lib.check(self.lib.mc_set_device(torch.cuda.current_device()), 'mc_set_device')
        self.dims = dict(dims)
        self.copy_blocks_num, self.control_cond_feats = control_info(state_dict)
        cfg = _lib.ModelConfig(
            input_feats=dims['input_feats'], max_seq_len=dims['max_seq_len'], latent_dim=dims['L'],
            num_parts=dims['H'], num_layers=dims['NL'], ffn_dim=dims['F'], time_embed_dim=dims['Te'],
            text_latent_dim=dims['Dt'], max_text_len=dims['Nt'], num_experts=dims['E'], topk=dims.get('topk', 2),
            dyn_heads=dyn_heads, capacity_factor=capacity_factor, cfg_scale=cfg_scale,
            num_ctrl_layers=self.copy_blocks_num, ctrl_cond_feats=self.control_cond_feats,
            ctrl_condition_cfg=int(bool(condition_cfg)))
        self.cfg_scale = float(cfg_scale)
        h = ctypes.c_void_p()
        _lib.check(self.lib.mc_model_create(ctypes.byref(cfg), ctypes.byref(h)), 'mc_model_create')
        self.handle = h
        for name, arr in pack_state_dict(state_dict, dims).items():
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            _lib.check(self.lib.mc_model_set_param(self.handle, name.encode(), arr.ctypes.data_as(ctypes.c_void_p),
                                                   arr.size), f'mc_model_set_param({name})')
        _lib.check(self.lib.mc_model_finalize(self.handle), 'mc_model_finalize')

    def context(self, batch, frames, max_steps=1000):
        return NativeContext(self, batch, frames, max_steps)

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.mc_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeContext:
    def __init__(self, model, batch, frames, max_steps):
        self.model, self.lib = model, model.lib
        self.B, self.T, self.C = int(batch), int(frames), model.dims['input_feats']
        h = ctypes.c_void_p()
        _lib.check(self.lib.mc_ctx_create(model.handle, self.B, self.T, int(max_steps), ctypes.byref(h)), 'mc_ctx_create')
        self.handle = h
        self.timesteps = None
        self._keep = []
        self._graph_state = None

    @property
    def workspace_bytes(self):
        return int(self.lib.mc_ctx_workspace_bytes(self.handle))

    def check(self):
        """Synchronise the current stream and raise if a kernel of this context flagged an error (the cooperative routing
        kernel's bounded grid barrier); the sampler loops call it once after the last step."""
        _lib.check(self.lib.mc_ctx_check(self.handle, _stream()), 'mc_ctx_check')

    def profile(self, on=True):
        """Bracket every FiLM out_layers GEMM launch with HIP events on its launch stream (bench.py's dominant-kernel figure)."""
        _lib.check(self.lib.mc_ctx_profile(self.handle, int(bool(on))), 'mc_ctx_profile')

    def profile_read(self, rows=0):
        """(average duration in us, launches, algorithmic GFLOP per launch) of the bracketed launches (of `rows` rows if given)."""
        us, n, gf = ctypes.c_double(), ctypes.c_int32(), ctypes.c_double()
        _lib.check(self.lib.mc_ctx_profile_read(self.handle, int(rows), ctypes.byref(us), ctypes.byref(n), ctypes.byref(gf)),
                   'mc_ctx_profile_read')
        return us.value, n.value, gf.value

    @property
    def effective_precision(self):
        """'f32' / 'f16' / 'f16x3' the per-step kernels really run in (a reduced-precision mode falls back to the fp32 small-batch
        kernels up to 512 residual rows, i.e. B = 1 at 196 frames)."""
        return {0: 'f32', 1: 'f16', 2: 'f16x3'}[int(self.lib.mc_ctx_effective_precision(self.handle))]

    @property
    def uses_coop_routing(self):
        return bool(self.lib.mc_ctx_uses_coop_routing(self.handle))

    def enable_capture(self):
        """Keep every layer's routing decisions of the last denoise call (tests)."""
        self._drop_graph()
        _lib.check(self.lib.mc_ctx_enable_capture(self.handle), 'mc_ctx_enable_capture')

    def _drop_graph(self):
        # a captured graph bakes in the kernel selection and the condition buffers of the moment of capture
        if getattr(self, '_graph_state', None) is not None:
            self.graph_release()

    def set_precision(self, precision):
        """'f32' (default, exact fp32 MFMA), 'f16' (fp16 operands, fp32 accumulate) or 'f16x3' (fp16 hi/lo split, three
        products: fp32-class) for the per-step GEMM-shaped kernels; gate / routing / normalisations stay fp32."""
        code = {'f32': 0, 'f16': 1, 'f16x3': 2}[precision]
        self._drop_graph()
        _lib.check(self.lib.mc_ctx_set_precision(self.handle, code), 'mc_ctx_set_precision')
        self.precision = precision

    def set_option(self, key, value):
        """Kernel-selection switch of THIS context (include/motioncraft_amd.h, mc_ctx_set_option): 'chain', 'big_tokens',
        'split_groups', 'gemm_tune', 'route_coop', ... -- the MC_* environment variables only seed the defaults."""
        self._drop_graph()
        _lib.check(self.lib.mc_ctx_set_option(self.handle, str(key).encode(), int(value)), 'mc_ctx_set_option')

    def set_tie_policy(self, policy):
        """'stable' (default) or 'reverse': order of equal-importance tokens at a capacity cut (tutel boundary, a16).
        Call before set_condition."""
        code = {'stable': 0, 'reverse': 1}[policy]
        self._drop_graph()             # a captured graph has the tie key and the twin-mode launch sequence baked in
        self._keep = []                # the library forgets the condition: set_condition must follow
        _lib.check(self.lib.mc_ctx_set_tie_policy(self.handle, code), 'mc_ctx_set_tie_policy')

    def routing(self, layer):
        """(expert ids [N,2] long, keep [N,2] bool) of `layer` from the last denoise call, on CPU."""
        idx = self.buffer('cap_idx', layer, dtype=torch.int32).view(-1, 2).cpu().long()
        keep = (self.buffer('cap_w', layer).view(-1, 2) != 0).cpu()
        return idx, keep

    def set_timesteps(self, t_orig):
        self._drop_graph()
        t = np.ascontiguousarray(np.asarray(t_orig, dtype=np.int32))
        _lib.check(self.lib.mc_ctx_set_timesteps(self.handle, t.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                 len(t), _stream()), 'mc_ctx_set_timesteps')
        self.timesteps = [int(v) for v in t]

    def set_condition(self, xf_out, motion_mask):
        xf = _dev_f32(xf_out, 'xf_out')
        mask = _dev_f32(motion_mask, 'motion_mask')
        d = self.model.dims
        if tuple(xf.shape) != (self.B, d['Nt'], d['Dt']):
            raise ValueError(f'xf_out shape {tuple(xf.shape)} != {(self.B, d["Nt"], d["Dt"])}')
        if mask.numel() != self.B * self.T:
            raise ValueError(f'motion_mask has {mask.numel()} elements, expected {self.B * self.T}')
        self._keep = [xf, mask]        # (the library copies the mask; a captured graph stays valid across conditions)
        _lib.check(self.lib.mc_ctx_set_condition(self.handle, _ptr(xf), _ptr(mask), _stream()), 'mc_ctx_set_condition')

    def set_control(self, c_feat):
        """c_feat [B, Tc, control_cond_feats] (output of the step-invariant condition pre-encoder) or None."""
        if (c_feat is None) != (not getattr(self, '_ctrl_on', False)):
            self._drop_graph()         # control branch on / off changes the captured launch sequence
        self._ctrl_on = c_feat is not None
        if c_feat is None:
            _lib.check(self.lib.mc_ctx_set_control(self.handle, None, 0, _stream()), 'mc_ctx_set_control')
            return
        c = _dev_f32(c_feat, 'c')
        if c.dim() != 3 or c.shape[0] != self.B or c.shape[2] != self.model.control_cond_feats:
            raise ValueError(f'c shape {tuple(c.shape)} != (B={self.B}, Tc, {self.model.control_cond_feats})')
        self._keep_c = c
        _lib.check(self.lib.mc_ctx_set_control(self.handle, _ptr(c), int(c.shape[1]), _stream()), 'mc_ctx_set_control')

    def denoise(self, x_t, step_index, out2=None, stop_after_layers=-1):
        x = _dev_f32(x_t, 'x_t')
        if tuple(x.shape) != (self.B, self.T, self.C):
            raise ValueError(f'x_t shape {tuple(x.shape)} != {(self.B, self.T, self.C)}')
        if out2 is None and stop_after_layers < 0:
            out2 = torch.empty(2 * self.B, self.T, self.C, device=x.device, dtype=torch.float32)
        _lib.check(self.lib.mc_denoise(self.handle, _ptr(x), int(step_index), _ptr(out2), int(stop_after_layers),
                                       _stream()), 'mc_denoise')
        return out2

    def sample_step(self, x_t, step_index, coefs, noise, x_prev=None, x0=None):
        x = _dev_f32(x_t, 'x_t')
        n = _dev_f32(noise, 'noise')
        if x_prev is None:
            x_prev = torch.empty_like(x)
        _lib.check(self.lib.mc_sample_step(self.handle, _ptr(x), int(step_index), ctypes.byref(coefs), _ptr(n),
                                           _ptr(x_prev), _ptr(x0), _stream()), 'mc_sample_step')
        return x_prev

    def sample_loop(self, x, step_indices, coefs, noise=None, seed=0, draw0=0, x0=None):
        """The whole sampler loop in one library call (mc_sample_loop): ``x`` [B,T,C] is updated IN PLACE through the schedule
        indices ``step_indices`` with ``coefs[k]``; ``noise`` [len, B,T,C] = the per-step draws, or None: drawn on the device
        (Philox4x32-10 keyed by ``seed``, draw index ``draw0 + k``).  Asynchronous on the current stream."""
        x = _dev_f32(x, 'x')
        if tuple(x.shape) != (self.B, self.T, self.C):
            raise ValueError(f'x shape {tuple(x.shape)} != {(self.B, self.T, self.C)}')
        n = len(step_indices)
        if len(coefs) != n:
            raise ValueError('one StepCoefs per step index')
        if noise is not None:
            noise = _dev_f32(noise, 'noise')
            if tuple(noise.shape) != (n, self.B, self.T, self.C):
                raise ValueError(f'noise shape {tuple(noise.shape)} != {(n, self.B, self.T, self.C)}')
        idx = (ctypes.c_int32 * n)(*[int(i) for i in step_indices])
        arr = (_lib.StepCoefs * n)(*coefs)
        _lib.check(self.lib.mc_sample_loop(self.handle, _ptr(x), idx, arr, n, _ptr(noise), int(seed) & (2 ** 64 - 1),
                                           int(draw0) & (2 ** 64 - 1), _ptr(x0), _stream()), 'mc_sample_loop')
        return x

    def graph_capture(self, x, noise, coefs):
        """Capture mc_sample_step into ONE hipGraph for the whole schedule.  ``x`` [B,T,C] is updated in place by every
        ``graph_step``, ``noise`` [B,T,C] is read by every replay (refill it in place between steps); both tensors must stay
        alive and keep their addresses.  ``coefs``: list of StepCoefs for every schedule index.  Call on a non-default
        torch stream (``with torch.cuda.stream(s):``)."""
        x, noise = _dev_f32(x, 'x'), _dev_f32(noise, 'noise')
        if tuple(x.shape) != (self.B, self.T, self.C) or noise.shape != x.shape:
            raise ValueError(f'x / noise shape {tuple(x.shape)} != {(self.B, self.T, self.C)}')
        arr = (_lib.StepCoefs * len(coefs))(*coefs)
        self._graph_keep = (x, noise)
        _lib.check(self.lib.mc_ctx_graph_capture(self.handle, _ptr(x), _ptr(noise), arr, len(coefs), _stream()),
                   'mc_ctx_graph_capture')

    def graph_step(self, step_index):
        _lib.check(self.lib.mc_ctx_graph_step(self.handle, int(step_index), _stream()), 'mc_ctx_graph_step')

    def graph_release(self):
        _lib.check(self.lib.mc_ctx_graph_release(self.handle), 'mc_ctx_graph_release')
        self._graph_keep = None
        self._graph_state = None

    def sample_step_seeded(self, x_t, step_index, coefs, noise, sqrt_ab, sqrt_1mab, pre_seq=None, pre_noise=None,
                           transl=(), x_prev=None, x0=None):
        """One step with the reference's pre_seq / transl_req seeding: ``x_t`` is overwritten IN PLACE on its first
        frames (q_sample of ``pre_seq`` with ``pre_noise``; ``transl`` = [(channel, v0, v1), ...] already q-sampled)."""
        x = _dev_f32(x_t, 'x_t')
        n = _dev_f32(noise, 'noise')
        sd = _lib.Seed()
        sd.pre_len = 0
        if pre_seq is not None:
            p, pn = _dev_f32(pre_seq, 'pre_seq'), _dev_f32(pre_noise, 'pre_noise')
            if p.dim() != 3 or p.shape[0] != self.B or p.shape[2] != self.C or p.shape[1] > self.T or pn.shape != p.shape:
                raise ValueError(f'pre_seq shape {tuple(p.shape)} does not fit the window {(self.B, self.T, self.C)}')
            sd.pre_seq_dev, sd.pre_noise_dev, sd.pre_len = p.data_ptr(), pn.data_ptr(), int(p.shape[1])
        sd.sqrt_ab, sd.sqrt_1mab = float(sqrt_ab), float(sqrt_1mab)
        if len(transl) > _lib.MAX_TRANSL:
            raise ValueError(f'at most {_lib.MAX_TRANSL} transl_req items')
        sd.num_transl = len(transl)
        for k, (ch, v0, v1) in enumerate(transl):
            sd.transl_channel[k] = int(ch)
            sd.transl_value[k][0], sd.transl_value[k][1] = float(v0), float(v1)
        if x_prev is None:
            x_prev = torch.empty_like(x)
        _lib.check(self.lib.mc_sample_step_seeded(self.handle, _ptr(x), int(step_index), ctypes.byref(coefs), _ptr(n),
                                                  ctypes.byref(sd), _ptr(x_prev), _ptr(x0), _stream()), 'mc_sample_step_seeded')
        return x_prev

    def sample_step_inpaint(self, x_t, step_index, coefs, noise, gt, keep, gt_noise=None, blend_w=None, blend_len=0,
                            x_prev=None, x0=None):
        """RePaint step: ``keep`` is the bool outpainting_mask, ``gt`` the kept motion (both [B,T,C] on the device)."""
        x = _dev_f32(x_t, 'x_t')
        n = _dev_f32(noise, 'noise')
        g = _dev_f32(gt, 'gt')
        if keep.dtype != torch.bool or not keep.is_cuda or not keep.is_contiguous() or keep.shape != x.shape \
                or g.shape != x.shape:
            raise ValueError('outpainting_mask must be a contiguous bool device tensor of the shape of x_t (and gt too)')
        ip = _lib.Inpaint()
        ip.gt_dev, ip.keep_dev = g.data_ptr(), keep.data_ptr()
        gn = _dev_f32(gt_noise, 'gt_noise') if gt_noise is not None else None
        ip.gt_noise_dev = gn.data_ptr() if gn is not None else None
        bw = _dev_f32(blend_w, 'blend_w') if blend_len else None
        ip.blend_w_dev, ip.blend_len = (bw.data_ptr() if bw is not None else None), int(blend_len)
        if x_prev is None:
            x_prev = torch.empty_like(x)
        _lib.check(self.lib.mc_sample_step_inpaint(self.handle, _ptr(x), int(step_index), ctypes.byref(coefs), _ptr(n),
                                                   ctypes.byref(ip), _ptr(x_prev), _ptr(x0), _stream()),
                   'mc_sample_step_inpaint')
        return x_prev

    def renoise(self, x, noise, a, b, out=None):
        """out = a x + b noise (the forward ``_undo`` step of the resampling schedule)."""
        x, noise = _dev_f32(x, 'x'), _dev_f32(noise, 'noise')
        if out is None:
            out = torch.empty_like(x)
        _lib.check(self.lib.mc_op_renoise(_ptr(x), _ptr(noise), float(a), float(b), _ptr(out), x.numel(), _stream()),
                   'mc_op_renoise')
        return out

    def buffer(self, name, layer=0, dtype=torch.float32):
        """Copy of a named workspace buffer (tests)."""
        p = ctypes.c_void_p()
        n = ctypes.c_int64()
        _lib.check(self.lib.mc_ctx_get_buffer(self.handle, name.encode(), int(layer), ctypes.byref(p), ctypes.byref(n)),
                   'mc_ctx_get_buffer')
        out = torch.empty(n.value, device='cuda', dtype=dtype)
        # stream-ordered copy on the CURRENT torch stream: a plain hipMemcpy device-to-device is enqueued on the null stream and does not
        # block the host, so under a non-default (non-blocking) torch stream the reads of `out` that follow could overtake it
        import ctypes as _c
        hip = _c.CDLL('libamdhip64.so')
        hip.hipMemcpyAsync.argtypes = [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_void_p]
        rc = hip.hipMemcpyAsync(_c.c_void_p(out.data_ptr()), p, n.value * out.element_size(), 3, _c.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f'hipMemcpyAsync failed: {rc}')
        torch.cuda.current_stream().synchronize()
        return out

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.mc_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
