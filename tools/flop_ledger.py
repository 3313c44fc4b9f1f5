#!/usr/bin/env python
"""FLOP ledger of ONE denoising step (GPU): what every launcher of libmotioncraft_amd.so books as useful multiply-add work (x 2) while
mc_debug_flop_ledger is on, keyed "kernel@<grid work-items>" -- the key a rocprofv3 kernel trace gives a dispatch.  tools/kernel_roofline.py
prices the per-kernel table with it, so no MFMA kernel of the step is left without FLOPs and the column sums to the executed FLOPs bench.py reports.

    [MC_CHAIN=92208535] python tools/flop_ledger.py [batch, default 64] [f32|f16|f16x3] > profiles/rNN_flop_ledger[_serial].txt

Expert MLPs are booked at the routing's slot count (tokens x top-2 before capacity drops)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd import lib as L_                         # noqa: E402
from motioncraft_amd.diffusion import build_diffusion         # noqa: E402
from motioncraft_amd.engine import NativeModel                # noqa: E402
from motioncraft_amd.synthetic import default_dims, make_state_dict   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else 'f32'
T = 196
dims = default_dims()
nm = NativeModel(dims, make_state_dict(dims, 0), cfg_scale=dims['scale'])
ctx = nm.context(B, T, max_steps=1000)
ctx.set_precision(prec)
d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large'))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, T, 322, generator=g).cuda()
xf = torch.nn.functional.layer_norm(torch.randn(B, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],)).cuda()
lib = L_.load(require_gpu=True)


def dump():
    n = lib.mc_debug_flop_ledger_dump(None, 0)
    buf = ctypes.create_string_buffer(int(n))
    lib.mc_debug_flop_ledger_dump(buf, n)
    return [l.split('\t') for l in buf.value.decode().splitlines() if l]


lib.mc_debug_flop_ledger(1)
ctx.set_timesteps(d.timestep_map)                            # once-per-batch set-up: time embedding for all steps, FiLM tables, text MoE + K/V per layer
ctx.set_condition(xf, torch.ones(B, T).cuda())
torch.cuda.synchronize()
lib.mc_debug_flop_ledger(0)
setup_rows = dump()
order = [999, 998]
coefs = [d.step_coefs(i, 'ddpm', dims['scale']) for i in order]
ctx.sample_loop(x, order[:1], coefs[:1], seed=1)            # warm-up step (lazy allocations, routing buffers)
torch.cuda.synchronize()
lib.mc_debug_flop_ledger(1)
ctx.sample_loop(x, order[1:], coefs[1:], seed=1)            # the booked step
torch.cuda.synchronize()
lib.mc_debug_flop_ledger(0)
try:
    commit = subprocess.check_output(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    commit = os.environ.get('GIT_COMMIT', 'unknown')
rows = dump()
total = sum(float(r[2]) for r in rows)
print(f'# commit {commit}')
print(f'# tools/flop_ledger.py: one mc_sample_loop step, B={B} x {T} frames, precision {prec}, MC_CHAIN={os.environ.get("MC_CHAIN", "default")}')
print('# kernel@grid-work-items <TAB> launches per step <TAB> GFLOP per step (multiply-add = 2; expert MLPs at the slot count of their routing)')
for name, calls, fl in sorted(rows, key=lambda r: -float(r[2])):
    print(f'{name}\t{calls}\t{float(fl) / 1e9:.3f}')
print(f'# total {total / 1e9:.3f} GFLOP per step = {total / 1e9 / B:.3f} GFLOP per sample and step')
print('# setup (mc_ctx_set_timesteps + mc_ctx_set_condition, once per batch): kernel@grid <TAB> launches <TAB> GFLOP')
for name, calls, fl in sorted(setup_rows, key=lambda r: -float(r[2])):
    print(f'setup:{name}\t{calls}\t{float(fl) / 1e9:.3f}')
