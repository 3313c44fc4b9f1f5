"""Host-side mirrors of the reference classes on the hot path, registered under the SAME names so
``configs/stmogen/*.py`` build unchanged:

  MotionDiffusion      mogen/models/architectures/diffusion_architecture.py:57-204 (eval branch)
  STMoGenTransformer   mogen/models/transformers/stmogen.py:626-761
  STMA                 mogen/models/attentions/st_attention.py:64-103 (config holder; the arithmetic
                       is inside libmotioncraft_amd.so)
  MSELoss              mogen/models/losses (training only: accepted, inert)

They own no arithmetic: weights are packed once into HBM (``weights.py``), every denoiser call goes
through the C-ABI.  Training entry points raise: training is not part of this path.
"""
import torch

from .builder import ARCHITECTURES, ATTENTIONS, LOSSES, SUBMODULES, build_attention, build_loss, build_submodule
from .diffusion import build_diffusion
from .engine import NativeModel


@ATTENTIONS.register_module()
class STMA:
    """Configuration of one MC-Attn block (kwargs exactly as in ca_block_cfg of the configs)."""

    def __init__(self, latent_dim, text_latent_dim, num_heads, num_text_heads, num_experts, topk, gate_type,
                 gate_noise, ffn_dim, time_embed_dim, max_seq_len, max_text_seq_len, temporal_comb, dropout,
                 static_body=True, dynamic_body=False, patch_size=1):
        if gate_type != 'cosine_top':
            raise NotImplementedError(f"gate_type={gate_type!r}: the stmogen configs use 'cosine_top'")
        if num_text_heads != 1 or not static_body or not dynamic_body or patch_size != 1:
            raise NotImplementedError('the MI355X path covers num_text_heads=1, static_body=True, dynamic_body=True, '
                                      'patch_size=1 (every shipped stmogen motionx config)')
        if dropout != 0:
            raise NotImplementedError('dropout != 0 is a training setting')
        self.latent_dim, self.text_latent_dim, self.num_heads = latent_dim, text_latent_dim, num_heads
        self.num_experts, self.topk, self.max_seq_len, self.max_text_seq_len = num_experts, topk, max_seq_len, max_text_seq_len
        self.time_embed_dim, self.ffn_dim = time_embed_dim, ffn_dim


@LOSSES.register_module()
class MSELoss:
    def __init__(self, reduction='mean', loss_weight=1.0):
        self.reduction, self.loss_weight = reduction, loss_weight


@SUBMODULES.register_module()
class STMoGenTransformer:
    def __init__(self, input_feats, max_seq_len=240, latent_dim=512, time_embed_dim=2048, num_layers=8,
                 sa_block_cfg=None, ca_block_cfg=None, ffn_cfg=None, text_encoder=None, use_pos_embedding=True,
                 use_residual_connection=False, time_embedding_type='sinusoidal', post_process_cfg=None,
                 init_cfg=None, patch_size=1, scale_func_cfg=None, pose_encoder_cfg=None, pose_decoder_cfg=None,
                 moe_route_loss_weight=1.0, template_kl_loss_weight=0.0001):
        if sa_block_cfg is not None or not use_pos_embedding or use_residual_connection or patch_size != 1 \
                or time_embedding_type != 'sinusoidal':
            raise NotImplementedError('option outside the shipped stmogen configs')
        for c in (pose_encoder_cfg, pose_decoder_cfg):
            if c.get('dataset_name') not in ('motionx', 'human_ml3d', 'kit_ml') or c.get('joints', False) \
                    or c.get('body_graph', False) or c.get('dataset_name') != pose_encoder_cfg.get('dataset_name'):
                raise NotImplementedError("the MI355X path covers the part-wise pose layouts of dataset_name="
                                          "'motionx' (12 parts, 322-d), 'human_ml3d' (8 parts, 263-d) and 'kit_ml' "
                                          '(8 parts, 251-d); joints=True / body_graph / rot6d / openpose17 are not shipped')
        if isinstance(ffn_cfg, list):
            raise NotImplementedError('per-layer ffn_cfg lists are not used by the shipped configs')
        self.ca_block = build_attention(ca_block_cfg)
        a = self.ca_block
        if latent_dim != a.latent_dim * a.num_heads:
            raise ValueError('latent_dim must equal ca_block_cfg.latent_dim * num_heads')
        self.input_feats, self.max_seq_len, self.latent_dim = input_feats, max_seq_len, latent_dim
        self.num_layers, self.time_embed_dim = num_layers, time_embed_dim
        self.scale_func_cfg = scale_func_cfg
        self.cfg_scale = float(scale_func_cfg['scale'])
        self.post_process_cfg = post_process_cfg
        self.text_encoder_cfg = text_encoder
        self.use_text_proj = bool(text_encoder.get('use_text_proj', False)) if text_encoder else False
        if self.use_text_proj:
            raise NotImplementedError('use_text_proj=True is not used by the shipped configs')
        self.dims = dict(input_feats=input_feats, max_seq_len=max_seq_len, L=a.latent_dim, H=a.num_heads,
                         NL=num_layers, F=ffn_cfg['ffn_dim'], Te=time_embed_dim, Dt=a.text_latent_dim,
                         Nt=a.max_text_seq_len, E=a.num_experts, topk=a.topk,
                         dataset=pose_encoder_cfg.get('dataset_name'))
        self.training = False
        self._native = None
        self._ctx = {}
        self._state = None
        self._control = None            # set by ControlT2MHalf: dict(copy_blocks_num, cond_feats, raw_feats, condition_cfg, pre_encode)
        self._wavenc = None
        self._textenc = None
        self.precision = 'f32'          # 'f32' | 'f16' | 'f16x3': MFMA operand precision of the per-step GEMMs (wrap_fp16_model)

    # ---- nn.Module-ish surface used by the tools ---------------------------------------------
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('training is outside the MI355X sampling path')
        return self

    def parameters(self):
        return iter(())

    def load_state_dict(self, state_dict, strict=True):
        self._state = dict(state_dict)
        self.release()
        return self

    def to(self, device=None, *a, **k):
        if device is not None and torch.device(device).type == 'cpu':
            raise RuntimeError('motioncraft_amd has no CPU path: the denoiser runs on MI355X only')
        return self

    def cuda(self, device=None):
        return self

    def release(self):
        for c in self._ctx.values():
            c.close()
        self._ctx = {}
        if self._native is not None:
            self._native.close()
        self._native = None
        if self._wavenc is not None:
            self._wavenc.close()
        self._wavenc = None
        if self._textenc is not None:
            self._textenc.close()
        self._textenc = None

    # ---- native plumbing ------------------------------------------------------------------------
    @property
    def native(self):
        if self._native is None:
            if self._state is None:
                raise RuntimeError('load_state_dict() has not been called: no weights to run')
            ccfg = True if not self._control else self._control['condition_cfg']
            self._native = NativeModel(self.dims, self._state, cfg_scale=self.cfg_scale, condition_cfg=ccfg)
            if self._control and self._native.copy_blocks_num != self._control['copy_blocks_num']:
                raise RuntimeError(f"checkpoint has {self._native.copy_blocks_num} control blocks, the wrapper expects "
                                   f"{self._control['copy_blocks_num']}")
        return self._native

    WAV_KEY = 'condition_pre_encoder.pre_encoder.feat_extractor.'

    @property
    def wav_encoder(self):
        """ConditionEncoder -> WavEncoder (controlnet.py:90-105) built from the checkpoint's own keys, on the device."""
        if self._wavenc is None:
            from .wav_encoder import NativeWavEncoder
            if self._state is None:
                raise RuntimeError('load_state_dict() has not been called: no weights to run')
            hit = [k for k in self._state if self.WAV_KEY in k]
            if not hit:
                raise RuntimeError(f"condition_pre_encode=True but the checkpoint has no '{self.WAV_KEY}*' weights")
            prefix = hit[0][:hit[0].index(self.WAV_KEY) + len(self.WAV_KEY)]
            self._wavenc = NativeWavEncoder(self._control['cond_feats'], self._control['raw_feats'], self._state, prefix)
        return self._wavenc

    def _encode_control(self, c, dev):
        """The step-invariant condition pre-encoder (ControlT2MHalf.forward_c line 1, controlnet.py:187): WavEncoder
        for BEAT2 audio (raw [B, samples, audio_in] -> [B, frames, D]), identity for FineDance music features."""
        c = c.to(device=dev, dtype=torch.float32).contiguous()
        if self._control.get('pre_encode'):
            if c.dim() == 2:
                c = c.unsqueeze(-1)
            if c.dim() != 3 or c.shape[-1] != self._control['raw_feats']:
                raise ValueError(f"raw audio condition of shape {tuple(c.shape)}: expected [B, samples, "
                                 f"{self._control['raw_feats']}]")
            c = self.wav_encoder(c)
        want = self.native.control_cond_feats
        if c.dim() != 3 or c.shape[-1] != want:
            raise ValueError(f'control condition of shape {tuple(c.shape)}: expected [B, Tc, {want}]')
        return c

    def sampling_context(self, B, T, timestep_map, model_kwargs, device=None):
        """Context with FiLM tables for ``timestep_map`` and text K/V for ``model_kwargs['xf_out']``."""
        key = (int(B), int(T))
        ctx = self._ctx.get(key)
        if ctx is None or ctx.max_steps < len(timestep_map):
            if ctx is not None:
                ctx.close()
            ctx = self.native.context(B, T, max_steps=max(len(timestep_map), 50))
            ctx.max_steps = max(len(timestep_map), 50)
            self._ctx[key] = ctx
        if getattr(ctx, 'precision', 'f32') != self.precision:
            ctx.set_precision(self.precision)
        if ctx.timesteps != [int(t) for t in timestep_map]:
            ctx.set_timesteps(timestep_map)
        c = model_kwargs.get('c', None)
        if c is not None and not self._control:
            raise ValueError('a control condition `c` was given but the model has no control branch: wrap it with '
                             'ControlT2MHalf and load a checkpoint that has controlnet.* weights')
        xf = model_kwargs.get('xf_out', None)
        if xf is None:
            raise ValueError("model_kwargs['xf_out'] is required (see get_precompute_condition)")
        mask = model_kwargs.get('motion_mask', None)
        if mask is None:
            raise ValueError("model_kwargs['motion_mask'] is required")
        dev = device or torch.device('cuda', torch.cuda.current_device())
        xf = xf.to(device=dev, dtype=torch.float32).contiguous()
        mask = mask.to(device=dev, dtype=torch.float32).reshape(B, T).contiguous()
        ctx.set_condition(xf, mask)
        if self._control:
            ctx.set_control(None if c is None else self._encode_control(c, dev))
        return ctx

    # ---- reference API ----------------------------------------------------------------------------
    @property
    def text_encoder(self):
        """build_text_encoder (diffusion_transformer.py:109-141) from the checkpoint's own keys, on the device."""
        if self._textenc is None:
            from .text_encoder import NativeTextEncoder
            from .weights import strip_prefix
            if self.text_encoder_cfg is None:
                raise RuntimeError('the config has text_encoder=None: pass xf_out')
            if self._state is None:
                raise RuntimeError('load_state_dict() has not been called: no weights to run')
            self._textenc = NativeTextEncoder(self.text_encoder_cfg, strip_prefix(self._state), max_len=self.dims['Nt'])
        return self._textenc

    def encode_text(self, text, clip_feat, device):
        """diffusion_transformer.py:142-172: CLIP features (given, or computed from the tokenized prompts) ->
        text_pre_proj -> textTransEncoder -> text_ln."""
        enc = self.text_encoder
        dev = device if device is not None and torch.device(device).type == 'cuda' else \
            torch.device('cuda', torch.cuda.current_device())
        if clip_feat is not None:
            return enc.encode_feat(clip_feat.to(device=dev, dtype=torch.float32))
        return enc.encode_text(text, dev)

    def get_precompute_condition(self, text=None, motion_length=None, xf_out=None, re_dict=None, device=None,
                                 sample_idx=None, clip_feat=None, **kwargs):
        if xf_out is None:
            xf_out = self.encode_text(text, clip_feat, device)
        return {'xf_out': xf_out}

    def post_process(self, motion):
        if self.post_process_cfg is not None:
            import numpy as np
            mean = torch.from_numpy(np.load(self.post_process_cfg['mean_path'])).type_as(motion)
            std = torch.from_numpy(np.load(self.post_process_cfg['std_path'])).type_as(motion)
            motion = motion * std + mean
        return motion

    def scale_func(self, timestep):
        w = (1 - (1000 - timestep) / 1000) * self.cfg_scale + 1
        return {'text_coef': w, 'none_coef': 1 - w}

    def forward(self, motion, timesteps, motion_mask=None, motion_length=None, num_intervals=1, patch_size=1,
                **kwargs):
        """One denoiser evaluation ``model(x, ts, **model_kwargs)`` -> CFG-combined x0 [B,T,C]."""
        t = int(timesteps[0])
        if not bool((timesteps == t).all()):
            raise NotImplementedError('per-sample timesteps differ: the sampler always uses one t per batch')
        B, T, _ = motion.shape
        kw = dict(kwargs)
        kw['motion_mask'] = motion_mask
        ctx = self.sampling_context(B, T, [t], kw, motion.device if motion.is_cuda else None)
        dev = torch.device('cuda', torch.cuda.current_device())
        out2 = ctx.denoise(motion.to(device=dev, dtype=torch.float32).contiguous(), 0)
        c = self.scale_func(t)
        return out2[:B] * c['text_coef'] + out2[B:] * c['none_coef']

    __call__ = forward


class ControlT2MHalf:
    """Mirror of the reference's plug-and-play control wrapper (controlnet.py:107-439):
    ``model.model = ControlT2MHalf(model.model, copy_blocks_num, control_cond_feats, cfg)`` as in
    tools/s2g_test.py:592-601 and tools/m2d_test.py:372-381.  The copied DecoderLayers, the zero-init
    before/after projections and ``control_cond_input`` run inside the library; the step-invariant condition
    pre-encoder (WavEncoder) runs once per condition through ``mc_wavenc_forward``."""

    def __init__(self, base_model, copy_blocks_num=2, control_cond_feats=438, cfg=None, joint_embed_unfreeze=True,
                 unfreeze_mode='all'):
        if not isinstance(base_model, STMoGenTransformer):
            raise TypeError('base_model must be the STMoGenTransformer built from the config')
        cfg = cfg if cfg is not None else {}
        ce = cfg['condition_encode_cfg'] if 'condition_encode_cfg' in cfg else {}
        if cfg.get('patch_size', 1) != 1:
            raise NotImplementedError('patch_size > 1 is not used by the shipped configs')
        if not (1 <= copy_blocks_num < base_model.num_layers):
            raise ValueError('copy_blocks_num must be in [1, num_layers)')
        self.base_model = base_model
        self.copy_blocks_num = copy_blocks_num
        pre = bool(ce.get('condition_pre_encode', False))
        if pre and (ce.get('dataset_name') != 'beats2' or ce.get('condition_pre_encode_type') != 'wav'):
            raise NotImplementedError("condition_pre_encode: the reference only defines dataset_name='beats2' with "
                                      "condition_pre_encode_type='wav' (controlnet.py:93-99)")
        feats = ce.get('condition_latent_dim', base_model.latent_dim) if pre else control_cond_feats
        base_model._control = dict(copy_blocks_num=copy_blocks_num, cond_feats=feats, raw_feats=control_cond_feats,
                                   condition_cfg=bool(ce.get('condition_cfg', False)), pre_encode=pre)
        self.cfg = cfg
        self.training = False

    # delegate the sampler-facing surface
    cfg_scale = property(lambda self: self.base_model.cfg_scale)
    dims = property(lambda self: self.base_model.dims)

    def sampling_context(self, *a, **k):
        return self.base_model.sampling_context(*a, **k)

    def get_precompute_condition(self, **kwargs):
        return self.base_model.get_precompute_condition(**kwargs)

    def post_process(self, output):
        return self.base_model.post_process(output)

    def load_state_dict(self, state_dict, strict=True):
        self.base_model.load_state_dict(state_dict, strict=strict)
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('training is outside the MI355X sampling path')
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def release(self):
        self.base_model.release()

    def forward(self, motion, timesteps, motion_mask=None, motion_length=None, num_intervals=1, c=None, **kwargs):
        return self.base_model.forward(motion, timesteps, motion_mask=motion_mask, motion_length=motion_length,
                                       num_intervals=num_intervals, c=c, **kwargs)

    __call__ = forward


def wrap_fp16_model(model, split=True):
    """Mirror of mmcv.runner.wrap_fp16_model as the reference tools call it (``tools/test.py:95-97``: ``if cfg.get('fp16')
    is not None: wrap_fp16_model(model)``): switches the per-step GEMM-shaped kernels of the denoiser to the fp16 MFMA.
    ``split=True`` (default) keeps fp32-class results (operands split hi + lo, three products, fp32 accumulate:
    'f16x3'); ``split=False`` is the single-rounding fp16 form mmcv's wrapper produces ('f16').  The gate, routing and
    every normalisation stay fp32 either way.  Accepts a MotionDiffusion, a STMoGenTransformer or a ControlT2MHalf.

    Two differences from mmcv's wrapper to be aware of: (1) the DEFAULT here is the split form, which is NOT the numerics of
    mmcv's single-rounding fp16 -- pass ``split=False`` to validate against an fp16 reference run; (2) up to 512 residual rows
    (2 B T; B = 1 at 196 frames) a reduced-precision context runs the fp32 small-batch kernels, which are faster there, so a
    B = 1 benchmark of 'fp16' exercises the fp32 path -- ``NativeContext.effective_precision`` reports what really runs."""
    target = getattr(model, 'model', model)
    target = getattr(target, 'base_model', target)
    if not hasattr(target, 'precision'):
        raise TypeError(f'wrap_fp16_model: {type(model).__name__} has no MI355X denoiser inside')
    target.precision = 'f16x3' if split else 'f16'
    return model


def to_cpu(x):
    return x.detach().cpu() if isinstance(x, torch.Tensor) else x


@ARCHITECTURES.register_module()
class MotionDiffusion:
    def __init__(self, model=None, loss_recon=None, loss_reduction='frame', diffusion_train=None, diffusion_test=None,
                 sampler_type='uniform', init_cfg=None, inference_type='ddpm', opt=None, hand_loss_factor=1.0,
                 face_no_loss=False, hand_no_loss=False, **kwargs):
        self.inference_type = inference_type
        self.opt = opt
        if self.inference_type != 'gt':
            self.model = build_submodule(model)
        self.loss_recon = build_loss(loss_recon)
        self.diffusion_train_cfg = diffusion_train
        self.diffusion_test = build_diffusion(diffusion_test, opt=opt)
        self.training = False

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('training is outside the MI355X sampling path')
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def load_state_dict(self, state_dict, strict=True):
        if self.inference_type != 'gt':
            self.model.load_state_dict(state_dict, strict=strict)
        return self

    def forward(self, **kwargs):
        """Eval branch of the reference's MotionDiffusion.forward (diffusion_architecture.py:163-204)."""
        if self.training:
            raise NotImplementedError('training is outside the MI355X sampling path')
        motion = kwargs['motion'].float()
        motion_mask = kwargs['motion_mask'].float()
        motion_length = kwargs['motion_length']
        num_intervals = kwargs.get('num_intervals', 1)
        sample_idx = kwargs.get('sample_idx', None)
        patch_size = kwargs.get('patch_size', 1)
        c = kwargs.get('c', None)
        y = kwargs.get('y', {})
        B, T = motion.shape[:2]
        text = [kwargs['motion_metas'][i]['text'] for i in range(B)] if 'motion_metas' in kwargs else None
        dim_pose = kwargs['motion'].shape[-1]
        if self.inference_type != 'gt':
            cond_kw = {k: v for k, v in kwargs.items() if k not in ('text', 'device')}
            model_kwargs = self.model.get_precompute_condition(device=motion.device, text=text, **cond_kw)
            model_kwargs.update(motion_mask=motion_mask, sample_idx=sample_idx, motion_length=motion_length,
                                num_intervals=num_intervals, c=c, y=y, patch_size=patch_size)
            inference_kwargs = kwargs.get('inference_kwargs', {})
        if self.inference_type == 'ddpm':
            output = self.diffusion_test.p_sample_loop(self.model, (B, T, dim_pose), clip_denoised=False,
                                                       progress=False, model_kwargs=model_kwargs, **inference_kwargs)
        elif self.inference_type == 'ddim':
            output = self.diffusion_test.ddim_sample_loop(self.model, (B, T, dim_pose), clip_denoised=False,
                                                          progress=False, model_kwargs=model_kwargs, eta=0,
                                                          **inference_kwargs)
        elif self.inference_type == 'gt':
            output = motion
        else:
            raise KeyError(self.inference_type)
        results = kwargs
        if self.inference_type != 'gt':
            output = self.model.post_process(output)
        results['pred_motion'] = output
        return self.split_results(results)

    __call__ = forward

    @staticmethod
    def split_results(results):
        """base_architecture.py:112-140."""
        B = results['motion'].shape[0]
        output = []
        for i in range(B):
            o = dict()
            o['motion'] = to_cpu(results['motion'][i])
            o['pred_motion'] = to_cpu(results['pred_motion'][i])
            o['motion_length'] = to_cpu(results['motion_length'][i])
            o['motion_mask'] = to_cpu(results['motion_mask'][i])
            o['pred_motion_length'] = to_cpu(results['pred_motion_length'][i]) if 'pred_motion_length' in results \
                else to_cpu(results['motion_length'][i])
            o['pred_motion_mask'] = to_cpu(results['pred_motion_mask'][i]) if 'pred_motion_mask' in results \
                else to_cpu(results['motion_mask'][i])
            if 'motion_metas' in results:
                metas = results['motion_metas'][i]
                if 'text' in metas:
                    o['text'] = metas['text']
                if 'token' in metas:
                    o['token'] = metas['token']
            output.append(o)
        return output
