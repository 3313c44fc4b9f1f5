#!/usr/bin/env python
"""Microbenchmark of the fp32 MFMA GEMM kernel at the denoiser's shapes (B=64, T=196 sizes),
HIP-event timed on the launch stream, interleaved rounds.  Run on the MI355X box."""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd import lib as L_  # noqa: E402
from motioncraft_amd.engine import _ptr, _stream  # noqa: E402

SHAPES = [  # name, M, N, K, act, residual
    ('film 1536^2', 25088, 1536, 1536, 0, True),
    ('exp/ffn fc2 K=512', 50176, 128, 512, 0, False),
    ('gate proj', 301056, 256, 128, 0, False),
    ('dyn qkv', 301056, 384, 128, 0, False),
    ('fc1 gelu K=128', 50176, 512, 128, 1, False),
    ('decoder N=322', 25088, 322, 1536, 0, False),
    ('film tables M=1000', 1000, 3072, 2048, 0, False),
    ('4096^3 (guide reference point)', 4096, 4096, 4096, 0, False),
    ('film, exact 4 rounds', 21760 - 21760 % 128 + 128 * 0, 1536, 1536, 0, True),
    ('2048x1536 tiles=192', 2048, 1536, 1536, 0, True),
]


def main(rounds=5):
    lib = L_.load(require_gpu=True)
    bufs = {}
    for name, M, N, K, act, res in SHAPES:
        bufs[name] = (torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda') / K ** 0.5,
                      torch.randn(N, device='cuda'), torch.randn(M, N, device='cuda') if res else None,
                      torch.empty(M, N, device='cuda'))
    times = {n: [] for n, *_ in SHAPES}
    for r in range(rounds + 1):
        for name, M, N, K, act, res in SHAPES:
            a, w, b, rr, c = bufs[name]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                L_.check(lib.mc_op_gemm(_ptr(a), _ptr(w), _ptr(b), _ptr(rr), _ptr(c), M, N, K, K, act, _stream()))
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[name].append(e0.elapsed_time(e1) / 3)
    for name, M, N, K, act, res in SHAPES:
        t = sorted(times[name])
        med = t[len(t) // 2]
        print(f'{name:22s} M={M:6d} N={N:5d} K={K:5d}: median {med*1e3:8.1f} us  min {t[0]*1e3:8.1f} us  '
              f'{2.0*M*N*K/med/1e9:7.1f} TFLOP/s ({2.0*M*N*K/med/1e9/157.3*100:5.1f}% of fp32 MFMA peak)')


if __name__ == '__main__':
    main()
