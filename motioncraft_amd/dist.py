"""Data-parallel sampling over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The path shards by samples: rank r owns the contiguous block
[r*B/W, (r+1)*B/W) and a full weight replica.  There is NO per-step collective: one broadcast of
the frozen condition embeddings before the loop (reference analogue: every rank reads its own
batches, tools/test.py:107-113) and one all-gather of the finished [B/W, T, C] fp32 poses after
it (reference: pickled-bytes all_gather in mogen/apis/test.py:141-150).
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world():
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def shard_range(total, rank=None, world_size=None):
    """Contiguous block of `total` samples owned by `rank`; requires total % world_size == 0."""
    if rank is None:
        rank, world_size = world()
    if total % world_size != 0:
        raise ValueError(f'batch {total} is not divisible by world size {world_size}')
    per = total // world_size
    return rank * per, (rank + 1) * per


def _device_collectives():
    """RCCL ("nccl") moves HBM tensors over xGMI directly; with gloo (CPU tests, single-GPU smoke runs)
    device tensors are staged through host memory."""
    return dist.get_backend() == 'nccl'


def _staging_device(t):
    """Where tensor `t` has to live for a collective of the active backend, or None if it is already there: RCCL only takes HBM
    tensors (a host tensor -- e.g. the `motion` placeholder a caller built on the CPU -- goes to this rank's GPU and the result comes
    back to the host: found by the world-size-1 RCCL run of round 6, "No backend type associated with device type cpu"); gloo only
    takes host tensors."""
    if _device_collectives():
        return None if t.is_cuda else torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu') if t.is_cuda else None


def _bcast(t, src):
    dev = _staging_device(t)
    if dev is not None:
        h = t.to(dev)
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)


def _scatter_rows(t, src):
    """Rank `src` holds the global tensor `t` [B, ...]; every rank gets its own contiguous row block -- 1/W of the bytes per
    link instead of the whole tensor to everyone (RCCL scatter = grouped send/recv over xGMI)."""
    rank, ws = world()
    lo, hi = shard_range(t.shape[0])
    dev = _staging_device(t)
    out = torch.empty((hi - lo,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev if dev is not None else t.device)
    parts = None
    if rank == src:
        g = t.to(dev) if dev is not None else t
        parts = [g[r * (hi - lo):(r + 1) * (hi - lo)].contiguous() for r in range(ws)]
    dist.scatter(out, parts, src=src)
    return out.to(t.device) if dev is not None else out


def broadcast_condition(xf_out, motion_mask, src=0, c=None):
    """Rank `src` holds the frozen condition embeddings of the GLOBAL batch; every rank returns its own slice.
    Tensors on other ranks only need the right shape/dtype/device.  The text embeddings and the mask (79 KB per sample) are
    broadcast -- afterwards every rank holds the global tensors, which `sample_sharded` slices.  ``c``, the control condition
    of the S2G / M2D / mixed configs (encoded audio [B, Tc, D] = 1.2 MB per sample at 196 frames: 308 MB at B = 256;
    SURVEY.md section 2.2), is SCATTERED instead: each rank receives only its block (the global `c` stays valid on `src`
    only); with it the return value is (xf, mask, c_local)."""
    if is_dist():
        for t in (xf_out, motion_mask):
            _bcast(t, src)
        lo, hi = shard_range(xf_out.shape[0])
        xf_out, motion_mask = xf_out[lo:hi].contiguous(), motion_mask[lo:hi].contiguous()
        if c is not None:
            c = _scatter_rows(c, src)
    return (xf_out, motion_mask) if c is None else (xf_out, motion_mask, c)


def gather_results(local):
    """all-gather of the finished sequences: [B/W, T, C] on every rank -> [B, T, C] on every rank (on the device `local` lives on)."""
    if not is_dist():
        return local
    rank, ws = world()
    dev = _staging_device(local)
    h = (local.detach().to(dev) if dev is not None else local.detach()).contiguous()
    out = torch.empty((ws * h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype, device=h.device)
    if _device_collectives():
        dist.all_gather_into_tensor(out, h)              # one RCCL all-gather into the contiguous result
    else:
        dist.all_gather(list(out.chunk(ws, dim=0)), h)
    return out.to(local.device) if dev is not None else out


def sample_sharded(arch, motion, motion_mask, xf_out, noise=None, step_noise=None, c=None, c_local=None, **kwargs):
    """Shard a global batch over the ranks, sample each shard through `arch` (MotionDiffusion
    mirror) and all-gather the poses.  `noise` / `step_noise` are GLOBAL tensors / callables
    returning global tensors (parity definition of SURVEY.md section 8e: each rank must match the
    oracle run on its shard alone).  `c` = GLOBAL control condition (ControlT2MHalf), sharded like the batch; `c_local` = this
    rank's block of it (what `broadcast_condition(..., c=)` returns: the control condition is scattered, not broadcast)."""
    lo, hi = shard_range(motion.shape[0])
    sl = slice(lo, hi)
    if c_local is not None:
        if c_local.shape[0] != hi - lo:
            raise ValueError(f'c_local has {c_local.shape[0]} samples, this rank owns {hi - lo}')
        kwargs['c'] = c_local
    elif c is not None:
        kwargs['c'] = c[sl]
    inf = dict(kwargs.pop('inference_kwargs', {}))
    if noise is not None:
        inf['noise'] = noise[sl]
    if step_noise is not None:
        inf['step_noise'] = (lambda i: step_noise(i)[sl]) if callable(step_noise) else [n[sl] for n in step_noise]
    res = arch(motion=motion[sl], motion_mask=motion_mask[sl], motion_length=motion_mask[sl].sum(1, keepdim=True).long(),
               xf_out=xf_out[sl], inference_kwargs=inf, **kwargs)
    local = torch.stack([r['pred_motion'] for r in res])
    dev = motion.device
    return gather_results(local.to(dev))
