#!/usr/bin/env python
"""Folded decoder tail alone (mc_op_gemm_tail): the column-tile kernel (gemm_tail_k) against the block-range kernel (gemm_tail2_k) on the
per-GPU shapes -- error against fp64 and time per launch, interleaved on one box.  usage: python tools/tail_bench.py [M ...]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd import lib as L_                      # noqa: E402
from motioncraft_amd.engine import _ptr, _stream           # noqa: E402

lib = L_.load(require_gpu=True)
N, K = 322, 1536
for M in [int(v) for v in sys.argv[1:]] or [12544, 6272, 19200, 3136]:
    g = torch.Generator(device='cuda').manual_seed(M)
    h = torch.randn(2 * M, K, device='cuda', generator=g)
    a = torch.randn(2 * M, K, device='cuda', generator=g)
    w = torch.randn(2, N, K, device='cuda', generator=g) / K ** 0.5
    b = torch.randn(2, N, device='cuda', generator=g)
    wc, wu = 5.16, -4.16
    sub = torch.arange(0, M, max(1, M // 512), device='cuda')
    ref = ((wc * h[sub].double() + wu * h[sub + M].double()) @ w[0].double().t() + (wc * a[sub].double() + wu * a[sub + M].double()) @ w[1].double().t()
           + b[0].double() + b[1].double())
    outs, times = {}, {1: [], 2: []}
    for rep in range(6):
        for v in (1, 2):
            c = torch.full((M + 1, N), 777.0, device='cuda')
            c2 = torch.full((M + 1, N), 777.0, device='cuda')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                L_.check(lib.mc_op_gemm_tail(_ptr(h), _ptr(a), _ptr(w), _ptr(b), _ptr(c), _ptr(c2), M, N, K, wc, wu, v, _stream()))
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / 10)
            outs[v] = c
    for v in (1, 2):
        e = float((outs[v][sub].double() - ref).abs().max())
        guard = bool((outs[v][M] == 777.0).all()) and bool((c2[M] == 777.0).all())
        t = sorted(times[v][1:])[2]
        fl = 2.0 * M * N * 2 * K
        print(f'M={M:6d} variant {v} ({"column tiles" if v == 1 else "block ranges"}): {t * 1e6:7.1f} us  {fl / t / 1e12:6.1f} TFLOP/s ({fl / t / 1e12 / 157.3:.3f} of peak)  '
              f'max|err vs fp64| {e:.2e}  guard row intact {guard}')
    print(f'         variants agree within {float((outs[1][:M] - outs[2][:M]).abs().max()):.2e}')
