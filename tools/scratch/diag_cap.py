import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import FULL, synth_inputs, step_noise_from_seed
from motioncraft_amd.diffusion import build_diffusion
from motioncraft_amd.engine import NativeModel
from motioncraft_amd.synthetic import control_param_shapes, make_state_dict
dims, copy, B, T = FULL, 2, 32, 196
feats = 1536
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16'
tune = int(sys.argv[2]) if len(sys.argv) > 2 else -1
sd = make_state_dict(dims, 0, shapes=control_param_shapes(dims, copy, feats))
nm = NativeModel(dims, sd, cfg_scale=dims['scale'])
g_ = torch.Generator().manual_seed(401)
lengths = [int(v) for v in torch.randint(T // 2, T + 1, (B,), generator=g_)]
x_T, xf, mask = synth_inputs(dims, B, T, seed=402, lengths=lengths)
c = torch.randn(B, T, feats)
d = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large', respace='15,15,8,6,6'))
coefs = [d.step_coefs(i, 'ddim', dims['scale']) for i in range(50)]
N2 = 2 * 2 * B * T * 12
stream = torch.cuda.Stream()
for rep in range(4):
  with torch.cuda.stream(stream):
        ctx = nm.context(B, T, max_steps=50)
        ctx.set_precision(prec)
        if tune >= 0:
            ctx.set_option('gemm_tune', tune)
        ctx.enable_capture()
        ctx.set_timesteps(d.timestep_map)
        ctx.set_condition(xf.cuda(), mask.cuda())
        ctx.set_control(c.cuda())
        x = x_T.cuda()
        noise = torch.randn(B, T, 322).cuda()
        for i in (49, 48, 47):
            ctx.sample_step(x, i, coefs[i], noise, x_prev=x)
            stream.synchronize()
            xx = x.cpu()
            for slot in range(6):
                idx = ctx.buffer('cap_idx', slot, dtype=torch.int32)
                bad = ((idx < 0) | (idx > 15)).nonzero().flatten()
                if len(bad):
                    print(f'rep {rep} step {i} slot {slot}: {len(bad)} bad entries of {idx.numel()}, first {int(bad[0])} last {int(bad[-1])}, sample {idx[bad[:4]].tolist()}')
        ctx.close()
print('done', prec, tune)
