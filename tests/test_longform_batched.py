"""CPU: the batched window driver (motioncraft_amd/longform.py::sample_long_batched; reference loop tools/m2d_test.py:139-232,
BASELINE configs[3]).  The denoiser is replaced by a per-window function of the window's own inputs (condition slice, RePaint gt),
so what is checked is what the driver must guarantee: the right (sequence, window) pairs per model call, RePaint chaining per
sequence, stitching identical to the one-window-at-a-time `sample_long`, and the 4-rank partition of configs[3] (128 sequences x 5
windows -> 160 windows per rank) gathered back to the single-process result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from motioncraft_amd import longform


class WindowFn:
    """pred[j] = f(c[j], xf_out[j], y.gt[j]) per window: no cross-window coupling, so batched == one at a time."""

    def __init__(self):
        self.calls = []

    def __call__(self, motion, motion_mask, motion_length, num_intervals, motion_metas, inference_kwargs, c=None, y=None,
                 xf_out=None, **kw):
        b, L, C = motion.shape
        self.calls.append(b)
        t = torch.arange(L, dtype=torch.float32).view(1, L, 1) / L
        pred = t.expand(b, L, C).clone()
        if c is not None:
            pred = pred + c.mean(dim=2, keepdim=True) + 0.01 * c[:, :, :1]
        if xf_out is not None:
            pred = pred + xf_out.mean(dim=(1, 2)).view(b, 1, 1)
        if y:
            keep = y['outpainting_mask']
            pred = torch.where(keep, y['gt'], pred + 0.1 * y['gt'][:, :1].mean(dim=(1, 2)).view(b, 1, 1))
        return [{'pred_motion': pred[j]} for j in range(b)]


def _inputs(S, total, F=7, C=5):
    g = torch.Generator().manual_seed(3)
    c = [torch.randn(total, F, generator=g) for _ in range(S)]
    xf = torch.randn(S, 4, 3, generator=g)
    first = torch.randn(S, 6, C, generator=g)
    mean, std = torch.randn(C, generator=g).numpy(), (torch.rand(C, generator=g) + 0.5).numpy()
    return c, xf, first, mean, std


def test_batched_windows_equal_one_window_at_a_time_and_stitch_the_same():
    S, total, L, pre, C = 5, 60, 24, 6, 5
    c, xf, first, mean, std = _inputs(S, total)
    dev = torch.device('cpu')
    for repaint in (False, True):
        fn = WindowFn()
        recs, wins = longform.sample_long_batched(fn, total, L, pre, c=c, text=['t'] * S, repaint=repaint, overlap_len=6, first_gt=first,
                                                  mean=mean, std=std, input_dim=C, device=dev, condition_kwargs=dict(xf_out=xf),
                                                  max_batch=4, shard=False)
        n_win, stride = longform.window_starts(total, L, pre)
        assert n_win == 3 and len(wins) == S * n_win
        # plain: all 15 windows chunked by 4 -> 4 4 4 3;  RePaint: window i of all 5 sequences -> (4, 1) x 3
        assert fn.calls == ([4, 4, 4, 3] if not repaint else [4, 1] * 3)
        for s in range(S):
            one = WindowFn()
            rec1, wins1 = longform.sample_long(one, total, L, pre, c=c[s], text='t', repaint=repaint, overlap_len=6, first_gt=first[s],
                                               mean=mean, std=std, input_dim=C, device=dev, condition_kwargs=dict(xf_out=xf[s:s + 1]))
            assert one.calls == [1] * n_win
            for w in range(n_win):
                assert np.array_equal(wins[(s, w)], wins1[w]), (repaint, s, w)
            assert recs[s].shape == rec1.shape == (stride * (n_win - 1) + L, C)
            assert np.array_equal(recs[s], rec1), (repaint, s)
            if repaint:          # window w > 0 starts with the de-normalised tail of window w - 1 (m2d_test.py:193)
                assert np.allclose(wins[(s, 1)][:6], wins[(s, 0)][-6:] * std + mean)


def test_sequences_of_different_lengths():
    L, pre, C = 24, 6, 5
    totals = [60, 42, 100]
    g = torch.Generator().manual_seed(5)
    c = [torch.randn(t, 7, generator=g) for t in totals]
    for repaint in (False, True):
        fn = WindowFn()
        recs, wins = longform.sample_long_batched(fn, totals, L, pre, c=c, text='x', repaint=repaint, overlap_len=6, fix_very_first=False,
                                                  input_dim=C, device=torch.device('cpu'), max_batch=160, shard=False)
        nw = [longform.window_starts(t, L, pre)[0] for t in totals]
        assert nw == [3, 2, 5]
        assert fn.calls == ([10] if not repaint else [3, 3, 2, 1, 1])
        for s, t in enumerate(totals):
            assert recs[s].shape == (18 * (nw[s] - 1) + 24, C)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        # BASELINE configs[3] geometry: 128 sequences of 512-frame class (5 windows of 120 frames advancing by 90: 480 frames), 4 ranks
        S, total, L, pre, C = 128, 480, 120, 30, 4
        g = torch.Generator().manual_seed(11)
        c = torch.randn(S, total, 3, generator=g)
        fn = WindowFn()
        recs, wins = longform.sample_long_batched(fn, total, L, pre, c=c, text='dance', repaint=False, input_dim=C,
                                                  device=torch.device('cpu'), max_batch=160, shard=True)
        assert fn.calls == [160], fn.calls                       # 32 sequences x 5 windows = ONE model call of 160 windows per rank
        assert sorted({s for s, _ in wins}) == list(range(rank * 32, rank * 32 + 32))
        ref, _ = longform.sample_long_batched(WindowFn(), total, L, pre, c=c, text='dance', repaint=False, input_dim=C,
                                              device=torch.device('cpu'), max_batch=160, shard=False)
        ok = len(recs) == S and all(np.array_equal(a, b) for a, b in zip(recs, ref)) and recs[0].shape == (4 * 90 + 120, C)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _worker_repaint(rank, ws, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=ws)
    try:
        # uneven shard (5 sequences over 2 ranks: 3 + 2), ragged lengths, RePaint chaining, chunks smaller than the shard
        totals, L, pre, C = [60, 42, 100, 78, 60], 24, 6, 5
        g = torch.Generator().manual_seed(21)
        c = [torch.randn(t, 7, generator=g) for t in totals]
        first = torch.randn(len(totals), 6, C, generator=g)
        kw = dict(c=c, text='x', repaint=True, overlap_len=6, first_gt=first, input_dim=C, device=torch.device('cpu'), max_batch=2)
        recs, wins = longform.sample_long_batched(WindowFn(), totals, L, pre, shard=True, **kw)
        ref, _ = longform.sample_long_batched(WindowFn(), totals, L, pre, shard=False, **kw)
        mine = longform.rank_sequences(len(totals), rank, ws)
        ok = (mine == ([0, 1, 2] if rank == 0 else [3, 4]) and sorted({s for s, _ in wins}) == mine
              and all(np.array_equal(a, b) for a, b in zip(recs, ref)) and [r.shape[0] for r in recs] == [18 * (n - 1) + 24 for n in (3, 2, 5, 4, 3)])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_repaint_mode_uneven_shards_ragged_lengths_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_repaint, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res)


def test_configs3_partition_640_windows_over_4_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1, 2, 3] and all(r[1] for r in res)
