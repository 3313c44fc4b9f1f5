"""Text condition encoder on the device: host side of ``mc_textenc_*``.

Mirrors ``DiffusionTransformer.build_text_encoder`` / ``encode_text``
(``mogen/models/transformers/diffusion_transformer.py:109-172``): CLIP ViT-B/32 text transformer ->
``text_pre_proj`` -> ``nn.TransformerEncoder`` (post-LN, GELU) -> ``text_ln``.  The reference accepts a precomputed
``clip_feat`` [B, 77, 512] in place of the raw prompt; that entry needs no tokenizer.  The raw-prompt entry needs
``clip.tokenize`` (BPE vocabulary of the un-vendored ``clip`` package): when that package is importable it is used on
the host; with only its vocabulary file at hand ``clip_bpe.ClipBPE`` restates the scheme; otherwise token ids must be
supplied.
"""
import ctypes

import numpy as np
import torch

from . import lib as _lib

CLIP_TEXT = dict(width=512, layers=12, heads=8, ff=2048, vocab=49408, context=77)      # ViT-B/32 text tower


def text_encoder_keys(num_layers, clip_layers=0):
    """The reference's state-dict keys (relative to the denoiser) the library consumes."""
    ks = ['text_pre_proj.weight', 'text_pre_proj.bias', 'text_ln.weight', 'text_ln.bias']
    for i in range(num_layers):
        p = f'textTransEncoder.layers.{i}.'
        ks += [p + n for n in ('self_attn.in_proj_weight', 'self_attn.in_proj_bias', 'self_attn.out_proj.weight',
                               'self_attn.out_proj.bias', 'linear1.weight', 'linear1.bias', 'linear2.weight',
                               'linear2.bias', 'norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias')]
    if clip_layers:
        ks += ['clip.token_embedding.weight', 'clip.positional_embedding', 'clip.ln_final.weight', 'clip.ln_final.bias']
        for i in range(clip_layers):
            p = f'clip.transformer.resblocks.{i}.'
            ks += [p + n for n in ('attn.in_proj_weight', 'attn.in_proj_bias', 'attn.out_proj.weight', 'attn.out_proj.bias',
                                   'mlp.c_fc.weight', 'mlp.c_fc.bias', 'mlp.c_proj.weight', 'mlp.c_proj.bias',
                                   'ln_1.weight', 'ln_1.bias', 'ln_2.weight', 'ln_2.bias')]
    return ks


class NativeTextEncoder:
    def __init__(self, text_encoder_cfg, state_dict, max_len=77, clip=None):
        """text_encoder_cfg: the configs' ``text_encoder=dict(pretrained_model='clip', latent_dim, num_layers, ff_size,
        num_heads=4, ...)``; ``clip``: dict(width, layers, heads, ff, vocab) of the text tower (default ViT-B/32).
        The tower is enabled when the state dict carries ``clip.token_embedding.weight``."""
        c = dict(text_encoder_cfg)
        if c.get('pretrained_model', 'clip') != 'clip':
            raise NotImplementedError("text_encoder.pretrained_model: the reference only defines 'clip'")
        if c.get('use_text_proj', False):
            raise NotImplementedError('use_text_proj=True is not used by the shipped configs')
        if c.get('activation', 'gelu') != 'gelu' or c.get('num_layers', 0) < 1:
            raise NotImplementedError('the shipped configs use a >=1-layer GELU nn.TransformerEncoder')
        self.lib = _lib.load(require_gpu=True)
        clip = dict(CLIP_TEXT, **(clip or {}))
        sd = {k: v for k, v in state_dict.items()}
        has_clip = 'clip.token_embedding.weight' in sd
        cfg = _lib.TextEncConfig()
        cfg.clip_dim, cfg.text_latent_dim = clip['width'], c['latent_dim']
        cfg.num_layers, cfg.ff_size, cfg.num_heads = c['num_layers'], c.get('ff_size', 2048), c.get('num_heads', 4)
        cfg.max_len = max_len
        cfg.clip_layers, cfg.clip_heads, cfg.clip_ff = (clip['layers'] if has_clip else 0), clip['heads'], clip['ff']
        cfg.vocab = int(sd['clip.token_embedding.weight'].shape[0]) if has_clip else clip['vocab']
        self.cfg, self.has_clip = cfg, has_clip
        h = ctypes.c_void_p()
        _lib.check(self.lib.mc_textenc_create(ctypes.byref(cfg), ctypes.byref(h)), 'mc_textenc_create')
        self.handle = h
        for k in text_encoder_keys(cfg.num_layers, cfg.clip_layers):
            if k not in sd:
                if k.startswith('text_pre_proj') and cfg.clip_dim == cfg.text_latent_dim:
                    continue                                       # nn.Identity (diffusion_transformer.py:124-127)
                raise KeyError(f'text encoder weight {k!r} is missing from the checkpoint')
            a = np.ascontiguousarray(sd[k].detach().cpu().float().numpy())
            _lib.check(self.lib.mc_textenc_set_param(self.handle, k.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                       f'mc_textenc_set_param({k})')
        _lib.check(self.lib.mc_textenc_finalize(self.handle), 'mc_textenc_finalize')

    def _out(self, B, dev):
        return torch.empty(B, self.cfg.max_len, self.cfg.text_latent_dim, device=dev, dtype=torch.float32)

    def encode_feat(self, clip_feat):
        """clip_feat [B, 77, 512] float32 device tensor -> xf_out [B, 77, text_latent_dim]."""
        f = clip_feat
        if not (f.is_cuda and f.dtype == torch.float32 and f.dim() == 3 and
                tuple(f.shape[1:]) == (self.cfg.max_len, self.cfg.clip_dim)):
            raise ValueError(f'clip_feat must be a float32 device tensor [B, {self.cfg.max_len}, {self.cfg.clip_dim}]')
        f = f.contiguous()
        out = self._out(f.shape[0], f.device)
        _lib.check(self.lib.mc_textenc_forward_feat(self.handle, ctypes.c_void_p(f.data_ptr()), f.shape[0],
                                                    ctypes.c_void_p(out.data_ptr()),
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   'mc_textenc_forward_feat')
        return out

    def encode_tokens(self, tokens, return_clip_feat=False):
        """tokens [B, 77] integer ids as produced by clip.tokenize -> xf_out (and the CLIP features)."""
        t = tokens.to(dtype=torch.int32).contiguous()
        if not t.is_cuda or t.dim() != 2 or t.shape[1] != self.cfg.max_len:
            raise ValueError(f'tokens must be a device tensor [B, {self.cfg.max_len}]')
        out = self._out(t.shape[0], t.device)
        feat = torch.empty(t.shape[0], self.cfg.max_len, self.cfg.clip_dim, device=t.device) if return_clip_feat else None
        _lib.check(self.lib.mc_textenc_forward_tokens(self.handle, ctypes.c_void_p(t.data_ptr()), t.shape[0],
                                                      ctypes.c_void_p(feat.data_ptr()) if feat is not None else None,
                                                      ctypes.c_void_p(out.data_ptr()),
                                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   'mc_textenc_forward_tokens')
        return (out, feat) if return_clip_feat else out

    def encode_text(self, text, device, bpe_path=None):
        """Raw prompts -> xf_out: ``clip.tokenize(text, truncate=True)`` of the reference when the ``clip`` package is
        installed, else the restated byte-pair scheme over its vocabulary file (``bpe_path`` or $MC_CLIP_BPE,
        ``bpe_simple_vocab_16e6.txt.gz``; see clip_bpe.py)."""
        import os
        try:
            import clip
            tokens = clip.tokenize(text, truncate=True)
        except ImportError as e:
            bpe_path = bpe_path or os.environ.get('MC_CLIP_BPE')
            if not bpe_path:
                raise NotImplementedError("raw prompts need clip.tokenize: neither the `clip` package nor its BPE vocabulary "
                                          '(MC_CLIP_BPE=/path/to/bpe_simple_vocab_16e6.txt.gz) is available: pass clip_feat '
                                          '[B,77,512] or token ids instead') from e
            from .clip_bpe import ClipBPE
            if getattr(self, '_bpe_path', None) != bpe_path:
                self._bpe, self._bpe_path = ClipBPE(bpe_path, vocab_size=self.cfg.vocab), bpe_path
            tokens = torch.from_numpy(self._bpe.tokenize(list(text), context_length=self.cfg.max_len, truncate=True))
        return self.encode_tokens(tokens.to(device))

    def close(self):
        if self.handle:
            self.lib.mc_textenc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
