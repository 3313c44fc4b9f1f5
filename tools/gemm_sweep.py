#!/usr/bin/env python
"""Per-launch time of the FiLM Linear (392 B x 1536 x 1536, bias + residual) over batch sizes; MC_SMALL_TILE_N=48|64|96 forces the
small-M tile width, unset = the per-launch choice (mc_launch_gemm_small).  One line: width, then us per launch for B = 1 .. 16."""
import os, sys, torch, ctypes, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from motioncraft_amd import lib as L
lib = L.load()
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
N = K = 1536
w = torch.randn(N, K).cuda() / K ** 0.5; b = torch.randn(N).cuda()
out = []
for B in [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16]:
    M = 392 * B
    a = torch.randn(M, K).cuda(); r = torch.randn(M, N).cuda(); c = torch.empty(M, N).cuda()
    for _ in range(5): lib.mc_op_gemm(p(a), p(w), p(b), p(r), p(c), M, N, K, K, 0, st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(40): lib.mc_op_gemm(p(a), p(w), p(b), p(r), p(c), M, N, K, K, 0, st)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40)
    out.append(f'{best * 1e6:7.1f}')
print(os.environ.get('MC_SMALL_TILE_N', 'auto').rjust(5), ' '.join(out))
