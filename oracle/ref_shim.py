"""Import the reference's OWN hot-path modules from /root/reference.  TEST INFRASTRUCTURE ONLY.

Works only in the build container (``/root/reference`` does not exist on the GPU
box).  Used by ``tests/golden/make_golden.py`` to (a) pin ``oracle/stmogen_oracle.py``
against the real reference code and (b) generate the committed golden fixtures.
Nothing is copied: the modules are imported where they lie.

Recipe: the reference's package ``__init__`` files pull in
mmcv version asserts, CLIP, pytorch3d ... so ``mogen`` and its sub-packages are
registered as empty shells whose ``__path__`` points into the reference tree, and
three third-party names are stubbed: ``mmcv`` (Registry / BaseModule only),
``clip`` (empty; the text encoder is off the per-step path) and ``tutel``
(``oracle/tutel_restated.py``).
"""
import importlib
import os
import sys
import types

import torch.nn as nn

REF = os.environ.get('MOTIONCRAFT_REFERENCE', '/root/reference')


class _Registry:
    """The slice of mmcv.utils.Registry the reference uses (mogen/models/builder.py:1-36)."""

    def __init__(self, name, parent=None, build_func=None, scope=None):
        self.name, self.parent, self._modules = name, parent, {}
        self.build_func = build_func or _build_from_cfg

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._modules[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        if key in self._modules:
            return self._modules[key]
        return self.parent.get(key) if self.parent is not None else None

    def build(self, cfg, *args, **kwargs):
        return self.build_func(cfg, self, *args, **kwargs)


def _build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop('type')
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f'{typ} is not in the {registry.name} registry')
    return cls(**args)


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not os.path.isdir(os.path.join(REF, 'mogen')):
        raise RuntimeError(f'reference tree not found at {REF} (ref_shim only works in the build container)')
    from . import tutel_restated

    mmcv = types.ModuleType('mmcv')
    mmcv.__version__ = '1.7.0'
    mmcv_cnn = types.ModuleType('mmcv.cnn')
    mmcv_cnn.MODELS = _Registry('model', build_func=_build_from_cfg)
    mmcv_cnn.build_activation_layer = lambda cfg: getattr(nn, cfg['type'])()
    mmcv_cnn.build_norm_layer = lambda cfg, n: (cfg['type'], nn.BatchNorm2d(n))
    for n in ('build_conv_layer', 'constant_init', 'kaiming_init', 'normal_init'):
        setattr(mmcv_cnn, n, None)
    mmcv_utils = types.ModuleType('mmcv.utils')
    mmcv_utils.Registry = _Registry
    mmcv_runner = types.ModuleType('mmcv.runner')
    mmcv_runner.BaseModule = _BaseModule
    mmcv_runner.load_checkpoint = None
    for name, mod in (('mmcv', mmcv), ('mmcv.cnn', mmcv_cnn), ('mmcv.utils', mmcv_utils),
                      ('mmcv.runner', mmcv_runner)):
        sys.modules[name] = mod
    clip = types.ModuleType('clip')          # encode_text only touches clip.tokenize / self.clip.dtype when clip_feat is given

    class _ClipStub(nn.Module):
        dtype = __import__('torch').float32
    clip.load = lambda name, device='cpu', **k: (_ClipStub(), None)
    clip.tokenize = lambda text, truncate=True: __import__('torch').zeros(len(text), 77, dtype=__import__('torch').long)
    sys.modules['clip'] = clip

    tutel = types.ModuleType('tutel')
    moe = types.ModuleType('tutel.moe')
    net = types.ModuleType('tutel.net')
    moe.moe_layer = tutel_restated.RestatedMoELayer

    def _no_world(group_count=1):
        raise RuntimeError('torch.distributed not initialised')  # reference catches this -> group=None
    net.create_groups_from_world = _no_world
    tutel.moe, tutel.net = moe, net
    sys.modules.update({'tutel': tutel, 'tutel.moe': moe, 'tutel.net': net})

    def shell(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__package__ = name
        sys.modules[name] = m
    shell('mogen', REF + '/mogen')
    shell('mogen.models', REF + '/mogen/models')
    for sub in ('utils', 'transformers', 'attentions', 'architectures', 'gnns', 'losses'):
        shell('mogen.models.' + sub, f'{REF}/mogen/models/{sub}')
    _installed = True


def load():
    """Returns a namespace with the reference modules of the hot path."""
    install()
    ns = types.SimpleNamespace()
    ns.stmogen = importlib.import_module('mogen.models.transformers.stmogen')
    ns.st_attention = importlib.import_module('mogen.models.attentions.st_attention')
    ns.efficient_attention = importlib.import_module('mogen.models.attentions.efficient_attention')
    ns.stylization_block = importlib.import_module('mogen.models.utils.stylization_block')
    ns.gaussian_diffusion = importlib.import_module('mogen.models.utils.gaussian_diffusion')
    ns.position_encoding = importlib.import_module('mogen.models.utils.position_encoding')
    return ns


def _shell(name, path):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__package__ = name
        sys.modules[name] = m


def load_evaluation():
    """The reference's evaluation side: the embedding model module (needs the ``transformers``
    package, present in this image) and mogen/core/evaluation's metric functions + evaluators."""
    install()
    _shell('mogen.models.rnns', REF + '/mogen/models/rnns')
    _shell('mogen.core', REF + '/mogen/core')
    _shell('mogen.core.evaluation', REF + '/mogen/core/evaluation')
    _shell('mogen.core.evaluation.evaluators', REF + '/mogen/core/evaluation/evaluators')
    ns = types.SimpleNamespace()
    ns.rnns = importlib.import_module('mogen.models.rnns.t2m_bigru_smplx')
    ns.utils = importlib.import_module('mogen.core.evaluation.utils')
    for n in ('precision', 'matching_score', 'fid', 'diversity', 'multimodality'):
        setattr(ns, n, importlib.import_module(f'mogen.core.evaluation.evaluators.{n}_evaluator'))
    return ns


def build_reference_denoiser(model_cfg):
    """STMoGenTransformer(**cfg) with text_encoder=None (SURVEY.md section 8c recipe)."""
    ref = load()
    cfg = {k: v for k, v in dict(model_cfg).items() if k != 'type'}
    cfg['text_encoder'] = None
    m = ref.stmogen.STMoGenTransformer(**cfg)
    m.use_text_proj = False  # attribute otherwise undefined (diffusion_transformer.py:118,212)
    m.eval()
    return m


def build_reference_text_encoder(model_cfg, text_encoder_cfg):
    """The reference denoiser WITH its text encoder (text_pre_proj, textTransEncoder, text_ln); the CLIP tower itself is a
    stub (un-vendored package), so only ``encode_text(text, clip_feat=...)`` is meaningful."""
    ref = load()
    cfg = {k: v for k, v in dict(model_cfg).items() if k != 'type'}
    cfg['text_encoder'] = dict(text_encoder_cfg)
    m = ref.stmogen.STMoGenTransformer(**cfg)
    m.eval()
    return m


def build_reference_control(model_cfg, copy_blocks_num, control_cond_feats, condition_cfg=True, wav_pre_encode=False):
    """ControlT2MHalf around a reference base model: condition_pre_encode=False (M2D form, raw
    feature condition) or, with ``wav_pre_encode``, the S2G form (dataset 'beats2', WavEncoder on raw audio)."""
    import torch.nn as _nn
    install()
    ctl = importlib.import_module('mogen.models.transformers.controlnet')
    base = build_reference_denoiser(model_cfg)
    for n in ('clip', 'text_pre_proj', 'textTransEncoder', 'text_ln'):
        setattr(base, n, _nn.Identity())

    class _Cfg(dict):
        __getattr__ = dict.__getitem__
    mcfg = _Cfg(model=_Cfg(model=_Cfg({k: v for k, v in dict(model_cfg).items()})),
                condition_encode_cfg=_Cfg(dataset_name='beats2' if wav_pre_encode else 'nothing',
                                          condition_pre_encode=bool(wav_pre_encode),
                                          condition_pre_encode_type='wav' if wav_pre_encode else 'nothing',
                                          control_cond_feats=control_cond_feats,
                                          condition_latent_dim=model_cfg['latent_dim'], condition_cfg=condition_cfg))
    m = ctl.ControlT2MHalf(base, copy_blocks_num=copy_blocks_num, control_cond_feats=control_cond_feats, cfg=mcfg)
    m.eval()
    return m


def build_reference_diffusion(diffusion_cfg, opt=None):
    """build_diffusion (diffusion_architecture.py:25-54) without importing that module
    (its import chain needs pytorch3d/librosa through utils/vis.py)."""
    gd = load().gaussian_diffusion
    if opt is None:
        opt = types.SimpleNamespace(same_overlap_noisy=False, no_repaint=False, addBlend=True,
                                    overlap_len=0)
    betas = gd.get_named_beta_schedule(diffusion_cfg['beta_scheduler'], diffusion_cfg['diffusion_steps'])
    mean_type = {'start_x': gd.ModelMeanType.START_X, 'previous_x': gd.ModelMeanType.PREVIOUS_X,
                 'epsilon': gd.ModelMeanType.EPSILON}[diffusion_cfg['model_mean_type']]
    var_type = {'learned': gd.ModelVarType.LEARNED, 'fixed_small': gd.ModelVarType.FIXED_SMALL,
                'fixed_large': gd.ModelVarType.FIXED_LARGE,
                'learned_range': gd.ModelVarType.LEARNED_RANGE}[diffusion_cfg['model_var_type']]
    if diffusion_cfg.get('respace', None) is not None:
        return gd.SpacedDiffusion(use_timesteps=gd.space_timesteps(diffusion_cfg['diffusion_steps'],
                                                                   diffusion_cfg['respace']),
                                  betas=betas, model_mean_type=mean_type, model_var_type=var_type,
                                  loss_type=gd.LossType.MSE, opt=opt)
    return gd.GaussianDiffusion(betas=betas, model_mean_type=mean_type, model_var_type=var_type,
                                loss_type=gd.LossType.MSE)
