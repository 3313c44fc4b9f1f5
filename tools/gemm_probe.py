import sys, ctypes, torch
import os; sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from motioncraft_amd import lib as L
lib = L.load(require_gpu=True)
def run(M, N, K, res=True):
    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda') if res else None
    c = torch.empty(M, N, device='cuda')
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    f = lambda: L.check(lib.mc_op_gemm(P(a), P(w), P(b), P(r), P(c), M, N, K, K, 0, st), 'gemm')
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f'M={M} N={N} K={K}: {us:.1f} us  ({2*M*N*K/us/1e6:.1f} TFLOP/s)')
for shp in [(392,1536,1536),(392,1536,768),(392,1536,384),(392,1536,192),(392,1536,96),(128,1536,1536),(384,1536,1536),(512,1536,1536),(3136,1536,1536)]:
    run(*shp)
