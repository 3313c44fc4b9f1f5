"""GPU (one MI355X), world_size 2: the REAL sampling path under two ranks.

Both ranks run on device 0 (a lease has one GPU; collectives go through gloo with host staging, exactly the code path
`bench.py --backend gloo` takes), each with its own NativeModel / context, through `motioncraft_amd.dist.sample_sharded`
-> MotionDiffusion.forward -> ddim_sample_loop -> libmotioncraft_amd.so.  Parity definition for W > 1: every rank equals
the CPU oracle run on ITS shard alone (the MoE capacity couples the samples of a rank's batch only, like the
reference's DDP evaluation, tools/test.py:107-113), and the all-gather equals the concatenation of the shards.
The control condition `c` travels with the broadcast (BASELINE configs[2]-[4])."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
NSTEPS = 6


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        import motioncraft_amd as mc
        from motioncraft_amd import dist as mcd
        from helpers import CTRL, CTRL_COPY, CTRL_FEATS, SMALL_SEED
        from oracle import stmogen_oracle as O, weights as W
        dims, B, T, Tc = CTRL, 4, 24, 20
        sd = W.make_state_dict(dims, SMALL_SEED, shapes=W.control_param_shapes(dims, CTRL_COPY, CTRL_FEATS))
        cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
        cfg.model.model.num_layers = 3
        cfg.merge_from_dict({'condition_encode_cfg': dict(dataset_name='nothing', condition_pre_encode=False,
                                                          condition_pre_encode_type='nothing', control_cond_feats=CTRL_FEATS,
                                                          condition_latent_dim=dims['L'] * dims['H'], condition_cfg=True)})
        arch = mc.build_architecture(cfg.model)
        arch.model = mc.ControlT2MHalf(arch.model, copy_blocks_num=CTRL_COPY, control_cond_feats=CTRL_FEATS, cfg=cfg)
        arch.load_state_dict({'model.' + k: v for k, v in sd.items()})
        g = torch.Generator().manual_seed(3)
        x_T = torch.randn(B, T, dims['input_feats'], generator=g)
        steps = [torch.randn(B, T, dims['input_feats'], generator=g) for _ in range(NSTEPS)]
        xf_src = torch.nn.functional.layer_norm(torch.randn(B, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],))
        mask_src = torch.ones(B, T)
        mask_src[1, 19:] = 0
        mask_src[2, 15:] = 0
        c_src = torch.randn(B, Tc, CTRL_FEATS, generator=g)
        # rank 0 owns the conditions (HBM); the other rank starts from garbage and must receive them
        dev = torch.device('cuda', 0)
        own = rank == 0
        xf = (xf_src if own else torch.full_like(xf_src, float('nan'))).to(dev)
        mask = (mask_src if own else torch.zeros_like(mask_src)).to(dev)
        c = (c_src if own else torch.full_like(c_src, float('nan'))).to(dev)
        lo, hi = mcd.shard_range(B)
        xs, ms, cs = mcd.broadcast_condition(xf, mask, src=0, c=c)
        assert torch.equal(xs.cpu(), xf_src[lo:hi]) and torch.equal(ms.cpu(), mask_src[lo:hi]) and torch.equal(cs.cpu(), c_src[lo:hi])
        S = 50
        out = mcd.sample_sharded(arch, torch.zeros(B, T, dims['input_feats']), mask.cpu(), xf.cpu(), noise=x_T,
                                 step_noise=lambda i: steps[S - 1 - i], c_local=cs.cpu(),
                                 motion_metas=[{'text': ''}] * (hi - lo), inference_kwargs=dict(num_steps=NSTEPS))
        # the oracle on this rank's shard alone
        sched = O.Schedule(1000, '15,15,8,6,6')
        x = x_T[lo:hi]
        for n in range(NSTEPS):
            i = S - 1 - n
            x0 = O.denoise_control(sd, dims, x, sched.timestep_map[i], xf_src[lo:hi], mask_src[lo:hi], c_src[lo:hi], CTRL_COPY)
            x = O.ddim_step(sched, i, x, x0, steps[n][lo:hi])
        err = float((out[lo:hi].cpu() - x).abs().max())
        arch.model.release()
        q.put((rank, err, tuple(out.shape), out.cpu().double().sum().item(), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_real_path_on_one_gpu():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res.sort()
    print('2 ranks on one GPU: per-shard |hip - oracle|', [f'{r[1]:.2e}' for r in res])
    assert [r[0] for r in res] == [0, 1] and [r[4] for r in res] == [(0, 2), (2, 4)]
    assert all(r[1] <= 1e-3 for r in res)                       # each rank == the oracle on its shard
    assert all(r[2] == (4, 24, 322) for r in res)
    assert res[0][3] == res[1][3]                               # both ranks hold the same gathered tensor


def _rccl_world1_worker(port, q):
    """ONE rank, backend "nccl" (= RCCL): every line of motioncraft_amd.dist / longform that only runs under a device-collective
    backend executes on the leased GPU -- broadcast of HBM tensors, scatter of the control condition, all_gather_into_tensor of
    the finished poses, the padded gather of the batched window driver -- and must leave the results of the same calls made
    WITHOUT a process group untouched, bit for bit (reference pattern: mogen/apis/test.py:36-82,141-150).  No N > 1 emulation."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    import motioncraft_amd as mc
    from motioncraft_amd import dist as mcd, longform
    from helpers import CTRL, CTRL_COPY, CTRL_FEATS, SMALL_SEED
    from oracle import weights as W
    dims, B, T, Tc = CTRL, 4, 24, 20
    sd = W.make_state_dict(dims, SMALL_SEED, shapes=W.control_param_shapes(dims, CTRL_COPY, CTRL_FEATS))
    cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
    cfg.model.model.num_layers = 3
    cfg.merge_from_dict({'condition_encode_cfg': dict(dataset_name='nothing', condition_pre_encode=False,
                                                      condition_pre_encode_type='nothing', control_cond_feats=CTRL_FEATS,
                                                      condition_latent_dim=dims['L'] * dims['H'], condition_cfg=True)})
    arch = mc.build_architecture(cfg.model)
    arch.model = mc.ControlT2MHalf(arch.model, copy_blocks_num=CTRL_COPY, control_cond_feats=CTRL_FEATS, cfg=cfg)
    arch.load_state_dict({'model.' + k: v for k, v in sd.items()})
    g = torch.Generator().manual_seed(3)
    x_T = torch.randn(B, T, dims['input_feats'], generator=g)
    steps = [torch.randn(B, T, dims['input_feats'], generator=g) for _ in range(NSTEPS)]
    xf = torch.nn.functional.layer_norm(torch.randn(B, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],))
    mask = torch.ones(B, T)
    mask[1, 19:] = 0
    c = torch.randn(B, Tc, CTRL_FEATS, generator=g)
    S = 50

    def sharded(on_device):
        xs, ms, cs = mcd.broadcast_condition(xf.to(dev), mask.to(dev), src=0, c=c.to(dev))
        assert xs.is_cuda and torch.equal(xs.cpu(), xf) and torch.equal(ms.cpu(), mask) and torch.equal(cs.cpu(), c)
        if not on_device:           # host tensors through the same entry points (RCCL only moves HBM tensors: dist.py stages them on the rank's GPU)
            xh, mh, ch = mcd.broadcast_condition(xf.clone(), mask.clone(), src=0, c=c.clone())
            assert not xh.is_cuda and torch.equal(xh, xf) and torch.equal(mh, mask) and torch.equal(ch, c)
        motion = torch.zeros(B, T, dims['input_feats'], device=dev if on_device else 'cpu')
        return mcd.sample_sharded(arch, motion, ms.cpu(), xs.cpu(), noise=x_T, step_noise=lambda i: steps[S - 1 - i], c_local=cs.cpu(),
                                  motion_metas=[{'text': ''}] * B, inference_kwargs=dict(num_steps=NSTEPS))

    # batched window driver on the same architecture: 3 sequences of different lengths, 2-3 windows each, one model call
    totals, L, pre = [42, 60, 42], 24, 6
    cw = [torch.randn(t, CTRL_FEATS, generator=g) for t in totals]
    nwin = sum(longform.window_starts(t, L, pre)[0] for t in totals)
    xw = torch.randn(nwin, L, dims['input_feats'], generator=g)
    nzw = [torch.randn(nwin, L, dims['input_feats'], generator=g) for _ in range(NSTEPS)]
    xfw = torch.nn.functional.layer_norm(torch.randn(3, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],))

    def windows():
        return longform.sample_long_batched(arch, totals, L, pre, c=cw, text=['a'] * 3, condition_kwargs=dict(xf_out=xfw.to(dev)),
                                            inference_kwargs=dict(noise=xw, step_noise=lambda i: nzw[S - 1 - i], num_steps=NSTEPS),
                                            max_batch=160, shard=True)[0]

    assert not mcd.is_dist()
    ref, ref_w = sharded(False), windows()
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        assert mcd.is_dist() and mcd._device_collectives() and mcd.world() == (0, 1)
        dist.barrier(device_ids=[0])
        out, out_h, out_w = sharded(True), sharded(False), windows()
        # the collectives themselves on HBM tensors, incl. the fp64 tensor bench.py's max-over-ranks reduce uses
        t = torch.arange(12, dtype=torch.float32, device=dev).view(3, 4)
        assert torch.equal(mcd.gather_results(t), t)
        t64 = torch.tensor([1.25, 2.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t64, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        ok = (out.is_cuda and torch.equal(out.cpu(), ref) and not out_h.is_cuda and torch.equal(out_h, ref) and all(a.shape == b.shape and bool((a == b).all()) for a, b in zip(out_w, ref_w))
              and t64.tolist() == [1.25, 2.5] and bool(torch.isfinite(out).all()))
        arch.model.release()
        q.put((ok, tuple(out.shape), [tuple(w.shape) for w in out_w], dist.get_backend()))
    finally:
        dist.destroy_process_group()


def _guarded(fn, port, q):
    try:
        fn(port, q)
    except BaseException:
        import traceback
        q.put(('error', traceback.format_exc()))
        raise


def test_rccl_branch_at_world_size_one():
    """VERDICT r05 "What's missing" 1: the `nccl` (RCCL) branch of dist.py / longform.py had never executed anywhere.  A world-size-1
    RCCL communicator on the leased MI355X runs every one of those calls (device broadcast / scatter / all_gather_into_tensor /
    barrier(device_ids) / fp64 all_reduce); results must equal the un-distributed calls bit for bit."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_guarded, args=(_rccl_world1_worker, _free_port(), q))
    p.start()
    res = q.get(timeout=600)
    assert res[0] != 'error', res[1]
    ok, shape, wshapes, backend = res
    p.join(timeout=120)
    assert p.exitcode == 0
    print('RCCL world size 1:', backend, shape, wshapes)
    assert backend == 'nccl' and ok and shape == (4, 24, 322) and wshapes == [(42, 322), (60, 322), (42, 322)]
