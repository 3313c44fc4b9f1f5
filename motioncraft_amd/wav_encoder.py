"""WavEncoder on the device: host side of ``mc_wavenc_*``.

Mirrors ``mogen/models/utils/blocks.py:11-71`` (``BasicBlock`` x 6, ``WavEncoder``) as used by the speech-to-gesture
configs through ``ConditionEncoder`` (``controlnet.py:90-105``): ``forward(wav [B, samples, audio_in]) -> [B, frames,
out_dim]``.  Eval-mode ``BatchNorm1d`` (running statistics) is folded into the preceding convolution here, in fp64;
the convolutions themselves run as MFMA GEMMs in libmotioncraft_amd.so.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import lib as _lib

BLOCKS_WITH_DOWNSAMPLE = (0, 1, 3, 5)      # blocks.py:58-63
BN_EPS = 1e-5


def wav_encoder_param_shapes(out_dim, audio_in):
    """state-dict keys/shapes of the reference WavEncoder (relative to ``feat_extractor.``)."""
    D = out_dim
    spec = [(audio_in, D // 4), (D // 4, D // 4), (D // 4, D // 4), (D // 4, D // 2), (D // 2, D // 2), (D // 2, D)]
    s = OrderedDict()
    for i, (cin, planes) in enumerate(spec):
        def bn(pre):
            for n in ('weight', 'bias', 'running_mean', 'running_var'):
                s[pre + n] = (planes,)
            s[pre + 'num_batches_tracked'] = ()
        s[f'{i}.conv1.weight'], s[f'{i}.conv1.bias'] = (planes, cin, 15), (planes,)
        bn(f'{i}.bn1.')
        s[f'{i}.conv2.weight'], s[f'{i}.conv2.bias'] = (planes, planes, 15), (planes,)
        bn(f'{i}.bn2.')
        if i in BLOCKS_WITH_DOWNSAMPLE:
            s[f'{i}.downsample.0.weight'], s[f'{i}.downsample.0.bias'] = (planes, cin, 15), (planes,)
            bn(f'{i}.downsample.1.')
    return s


def _fold(w, b, bn, pre):
    """conv [Cout, Cin, K] + eval BatchNorm -> tap-major GEMM weight [Cout, ceil4(K*Cin)] and bias, fp64 -> fp32."""
    g, beta = bn[pre + 'weight'].double(), bn[pre + 'bias'].double()
    mean, var = bn[pre + 'running_mean'].double(), bn[pre + 'running_var'].double()
    scale = g / torch.sqrt(var + BN_EPS)
    wf = (w.double() * scale[:, None, None]).permute(0, 2, 1).reshape(w.shape[0], -1)      # [Cout, K*Cin], k = tap*Cin + ci
    bf = (b.double() - mean) * scale + beta
    ld = (wf.shape[1] + 3) // 4 * 4
    out = torch.zeros(wf.shape[0], ld, dtype=torch.float64)
    out[:, :wf.shape[1]] = wf
    return out.float().contiguous(), bf.float().contiguous()


def pack_wav_encoder(state_dict, prefix=''):
    """reference state dict (keys ``<prefix>{i}.conv1.weight`` ...) -> the library's parameter table."""
    sd = {k[len(prefix):]: v.detach().cpu() for k, v in state_dict.items() if k.startswith(prefix)}
    out = OrderedDict()
    for i in range(6):
        out[f'b{i}.conv1.w'], out[f'b{i}.conv1.b'] = _fold(sd[f'{i}.conv1.weight'], sd[f'{i}.conv1.bias'], sd, f'{i}.bn1.')
        out[f'b{i}.conv2.w'], out[f'b{i}.conv2.b'] = _fold(sd[f'{i}.conv2.weight'], sd[f'{i}.conv2.bias'], sd, f'{i}.bn2.')
        if i in BLOCKS_WITH_DOWNSAMPLE:
            out[f'b{i}.down.w'], out[f'b{i}.down.b'] = _fold(sd[f'{i}.downsample.0.weight'], sd[f'{i}.downsample.0.bias'],
                                                               sd, f'{i}.downsample.1.')
    return out


class NativeWavEncoder:
    def __init__(self, out_dim, audio_in, state_dict, prefix=''):
        self.lib = _lib.load(require_gpu=True)
        self.out_dim, self.audio_in = int(out_dim), int(audio_in)
        h = ctypes.c_void_p()
        _lib.check(self.lib.mc_wavenc_create(self.audio_in, self.out_dim, ctypes.byref(h)), 'mc_wavenc_create')
        self.handle = h
        for name, t in pack_wav_encoder(state_dict, prefix).items():
            a = np.ascontiguousarray(t.numpy(), dtype=np.float32)
            _lib.check(self.lib.mc_wavenc_set_param(self.handle, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                       f'mc_wavenc_set_param({name})')
        _lib.check(self.lib.mc_wavenc_finalize(self.handle), 'mc_wavenc_finalize')

    def out_len(self, samples):
        n = ctypes.c_int32()
        _lib.check(self.lib.mc_wavenc_out_len(self.handle, int(samples), ctypes.byref(n)), 'mc_wavenc_out_len')
        return int(n.value)

    def __call__(self, wav):
        """wav [B, samples] (audio_in == 1) or [B, samples, audio_in], float32 on the device."""
        if wav.dim() == 2:
            wav = wav.unsqueeze(-1)
        if not (wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 3 and wav.shape[-1] == self.audio_in):
            raise ValueError(f'wav must be a float32 device tensor [B, samples, {self.audio_in}]')
        wav = wav.contiguous()
        B, S, _ = wav.shape
        T = self.out_len(S)
        if T < 1:
            raise ValueError(f'{S} audio samples are too few for the 540x-strided encoder')
        out = torch.empty(B, T, self.out_dim, device=wav.device, dtype=torch.float32)
        _lib.check(self.lib.mc_wavenc_forward(self.handle, ctypes.c_void_p(wav.data_ptr()), B, S,
                                              ctypes.c_void_p(out.data_ptr()),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'mc_wavenc_forward')
        return out

    forward = __call__

    def close(self):
        if self.handle:
            self.lib.mc_wavenc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
