#!/usr/bin/env python
"""A/B of the full-tile fp32 GEMM variants on the FiLM shape (and 4096^3): correctness vs a float64 product + interleaved timing.
usage: python tools/gemm_ab.py   (spawns itself once per MC_GEMM_TUNE value: the variant is chosen at library load)"""
import os
import subprocess
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)

SHAPES = [('film 25088x1536x1536', 25088, 1536, 1536), ('half 12544x1536x1536', 12544, 1536, 1536), ('4096^3', 4096, 4096, 4096)]


def child():
    from motioncraft_amd import lib as L_
    from motioncraft_amd.engine import _ptr, _stream
    lib = L_.load(require_gpu=True)
    tune = os.environ.get('MC_GEMM_TUNE', 'default')
    g = torch.Generator(device='cuda').manual_seed(0)
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device='cuda', generator=g)
        w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
        b = torch.randn(N, device='cuda', generator=g)
        r = torch.randn(M, N, device='cuda', generator=g)
        c = torch.empty(M, N, device='cuda')
        run = lambda: L_.check(lib.mc_op_gemm(_ptr(a), _ptr(w), _ptr(b), _ptr(r), _ptr(c), M, N, K, K, 0, _stream()))
        run()
        torch.cuda.synchronize()
        ref = (a[:512].double() @ w.double().T + b.double() + r[:512].double())
        ref2 = (a[-512:].double() @ w.double().T + b.double() + r[-512:].double())
        err = max(float((c[:512].double() - ref).abs().max()), float((c[-512:].double() - ref2).abs().max())) / float(ref.abs().max())
        for _ in range(40):          # steady clock: after a host-side gap (the fp64 reference above) the part re-ramps for tens of ms
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4)
        ts.sort()
        med = ts[len(ts) // 2]
        print(f'MC_GEMM_TUNE={tune:8s} {name:24s} median {med * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us  '
              f'{2.0 * M * N * K / med / 1e9:6.1f} TFLOP/s ({2.0 * M * N * K / med / 1e9 / 157.3 * 100:5.1f} % of fp32 MFMA peak)  '
              f'max rel err {err:.1e}', flush=True)


if __name__ == '__main__':
    if os.environ.get('MC_GEMM_AB_CHILD'):
        child()
    else:
        for tune in sys.argv[1:] or ['17', '49']:
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, MC_GEMM_TUNE=tune, MC_GEMM_AB_CHILD='1'))
