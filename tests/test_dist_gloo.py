"""CPU, world_size 2, gloo: the N>1 plumbing of motioncraft_amd/dist.py (sharding, condition
broadcast, result all-gather).  The denoiser itself is replaced by a per-sample function of
(noise, condition) so that the test checks exactly what distribution must guarantee: every rank
computes its own contiguous shard, the gathered result equals the single-process result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from motioncraft_amd import dist as mcd


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeArch:
    """Stands in for MotionDiffusion: pred = f(noise, xf, mask) per sample (no cross-sample coupling)."""

    def __call__(self, motion, motion_mask, motion_length, xf_out, inference_kwargs, c=None, **kw):
        x = inference_kwargs['noise']
        for n in inference_kwargs.get('step_noise', []):
            x = 0.5 * x + 0.1 * n
        pred = x * motion_mask.unsqueeze(-1) + xf_out.mean(dim=(1, 2)).view(-1, 1, 1)
        if c is not None:
            pred = pred + c.mean(dim=(1, 2)).view(-1, 1, 1)
        return [{'pred_motion': pred[i]} for i in range(pred.shape[0])]


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        B, T, C = 8, 6, 5
        g = torch.Generator().manual_seed(0)
        noise = torch.randn(B, T, C, generator=g)
        steps = [torch.randn(B, T, C, generator=g) for _ in range(3)]
        xf_src = torch.randn(B, 4, 3, generator=g)
        mask_src = (torch.rand(B, T, generator=g) > 0.3).float()
        c_src = torch.randn(B, 5, 7, generator=g)          # control condition (S2G / M2D / mixed configs)
        # rank 0 owns the condition; other ranks start from garbage and must receive it
        xf = xf_src.clone() if rank == 0 else torch.full_like(xf_src, float('nan'))
        mask = mask_src.clone() if rank == 0 else torch.zeros_like(mask_src)
        c = c_src.clone() if rank == 0 else torch.full_like(c_src, float('nan'))
        lo, hi = mcd.shard_range(B)
        assert (lo, hi) == (rank * B // ws, (rank + 1) * B // ws)
        xs, ms, cs = mcd.broadcast_condition(xf, mask, src=0, c=c)
        assert torch.equal(xs, xf_src[lo:hi]) and torch.equal(ms, mask_src[lo:hi]) and torch.equal(cs, c_src[lo:hi])
        if rank != 0:
            assert bool(torch.isnan(c).all())              # the control condition is scattered: no global copy on the other ranks
        out = mcd.sample_sharded(FakeArch(), torch.zeros(B, T, C), mask, xf, noise=noise, step_noise=steps, c_local=cs)
        ref = FakeArch()(None, mask_src, None, xf_src, dict(noise=noise, step_noise=steps), c=c_src)
        ref = torch.stack([r['pred_motion'] for r in ref])
        q.put((rank, bool(torch.equal(out, ref)), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


def test_shard_broadcast_gather_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2] == (8, 6, 5) for r in res)


def test_shard_range_errors_and_single_process_passthrough():
    assert mcd.shard_range(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        mcd.shard_range(10, 0, 4)
    x = torch.arange(6.).view(2, 3)
    assert mcd.gather_results(x) is x          # not initialised -> identity
    a, b = mcd.broadcast_condition(x, x)
    assert a is x and b is x
