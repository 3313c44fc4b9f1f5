import os, sys, types
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import SMALL, SMALL_SEED
import motioncraft_amd as mc
from oracle import stmogen_oracle as O, weights as W
sd = W.make_state_dict(SMALL, SMALL_SEED)
opt = types.SimpleNamespace(same_overlap_noisy=False, no_repaint=False, addBlend=True, overlap_len=6, no_resample=True,
                            jump_length=3, jump_n_sample=5, timestep_respacing='ddim50')
cfg = mc.Config.fromfile(os.path.join(ROOT, 'tests', 'configs', 'stmogen_small.py'))
cfg.model['opt'] = opt
arch = mc.build_architecture(cfg.model)
arch.load_state_dict({'model.' + k: v for k, v in sd.items()})
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 321
S, L = int(sys.argv[2]) if len(sys.argv) > 2 else 3, 24
g = torch.Generator().manual_seed(seed)
xf = torch.nn.functional.layer_norm(torch.randn(S, SMALL['Nt'], SMALL['Dt'], generator=g), (SMALL['Dt'],))
first_gt = torch.randn(S, 6, 322, generator=g)
x_T = torch.randn(S, L, 322, generator=g)
def draws(sd_):
    gen = torch.Generator().manual_seed(sd_)
    return (torch.randn(S, L, 322, generator=gen) for _ in range(10 ** 6))
keep = torch.zeros(S, L, 322, dtype=torch.bool); keep[:, :6] = True
gt = torch.zeros(S, L, 322); gt[:, :6] = first_gt
sched = O.Schedule(1000, '15,15,8,6,6')
mode = sys.argv[3] if len(sys.argv) > 3 else 'traj'
if 'pre' in mode:        # a B=6 plain loop on the same model first
    mk6 = dict(xf_out=xf.repeat(2, 1, 1).cuda(), motion_mask=torch.ones(2 * S, L).cuda())
    arch.diffusion_test.ddim_sample_loop(arch.model, (2 * S, L, 322), noise=torch.randn(2 * S, L, 322), clip_denoised=False, model_kwargs=mk6, eta=0)
outs = []
for rep in range(2):
    traj = [] if 'traj' in mode else None
    mk = dict(xf_out=xf.cuda(), motion_mask=torch.ones(S, L).cuda(), y=dict(gt=gt.cuda(), outpainting_mask=keep.cuda()))
    out = arch.diffusion_test.ddim_sample_loop(arch.model, (S, L, 322), noise=x_T, clip_denoised=False, model_kwargs=mk, eta=0,
                                               step_noise=draws(60), trajectory=traj)
    outs.append((out.cpu(), traj))
print('deterministic:', torch.equal(outs[0][0], outs[1][0]))
rt = []
ref = O.sample_loop_repaint(sd, SMALL, sched, x_T, xf, torch.ones(S, L), keep, gt, draws(60), 6, 50, no_resample=True, trajectory=rt)
for k, ((i, x, x0), (tl, tc, xr)) in enumerate(zip(outs[0][1] or [], rt)):
    if float((outs[0][0] - ref).abs().max()) > 1e-3: print(k, i, tl, 'per-sample max err', [f'{float((x.cpu()[s] - xr[s]).abs().max()):.1e}' for s in range(S)])
print(mode, 'final', float((outs[0][0] - ref).abs().max()))
