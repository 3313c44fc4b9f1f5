"""Long-sequence generation by overlapping windows: the caller side of the RePaint mode.

Mirrors the window loop of the reference's ``finedance_eval`` (``tools/m2d_test.py:139-225``; the same scheme is in
``tools/s2g_test.py:180-260``): a sequence of ``total_frames`` is covered by windows of ``motion_length`` frames that
advance by ``motion_length - pre_frames``; in RePaint mode every window after the first keeps its first
``overlap_len`` frames equal to the previous window's last ones (``y = {gt, outpainting_mask}``), so the windows are
sequentially dependent; otherwise they are independent and the overlap is simply cut away when stitching.

Reference quirk kept as the default: the ``gt`` handed to the next window is the previous output AFTER
``* std + mean`` (``m2d_test.py:193,206-212``), although the sampler works on normalised motion; pass
``gt_space='normalised'`` to feed the normalised frames instead.
"""
import numpy as np
import torch


def window_starts(total_frames, motion_length, pre_frames):
    """(number of windows, stride): m2d_test.py:143-145."""
    stride = motion_length - pre_frames
    if stride <= 0:
        raise ValueError('pre_frames must be smaller than motion_length')
    return (total_frames - pre_frames) // stride, stride


def sample_long(model, total_frames, motion_length, pre_frames=30, c=None, text='', repaint=False, overlap_len=30,
                fix_very_first=True, first_gt=None, mean=None, std=None, gt_space='denormalised', input_dim=322,
                device=None, condition_kwargs=None, inference_kwargs=None):
    """model: MotionDiffusion mirror (``mc.build_architecture``), opt set on it like the reference tools do.
    c: per-frame control condition [total_frames, F] (music features) or None.  condition_kwargs: ``xf_out`` /
    ``clip_feat`` for ONE window ([1, 77, *]).  inference_kwargs: dict or callable(window index) -> dict.
    Returns (stitched de-normalised motion [frames, input_dim] as float32 numpy, list of per-window outputs)."""
    dev = device or torch.device('cuda', torch.cuda.current_device())
    n_win, stride = window_starts(total_frames, motion_length, pre_frames)
    if n_win < 1:
        raise ValueError('sequence shorter than one window')
    mean = np.zeros(input_dim, np.float32) if mean is None else np.asarray(mean)
    std = np.ones(input_dim, np.float32) if std is None else np.asarray(std)
    if c is not None:
        c = torch.as_tensor(c, dtype=torch.float32)
    pieces_repaint, pieces_plain, windows, outputs = [], [], [], None
    for i in range(n_win):
        lo = i * stride
        kw = dict(motion=torch.zeros(1, motion_length, input_dim, device=dev),
                  motion_mask=torch.ones(1, motion_length, device=dev),
                  motion_length=torch.tensor([motion_length], device=dev).long(), num_intervals=1,
                  motion_metas=[{'text': text}])
        if c is not None:
            kw['c'] = c[lo:lo + motion_length].unsqueeze(0).to(dev)
        kw.update(condition_kwargs or {})
        inf = inference_kwargs(i) if callable(inference_kwargs) else dict(inference_kwargs or {})
        kw['inference_kwargs'] = inf
        if repaint:
            y = {}
            if overlap_len > 0:
                gt = torch.zeros(1, motion_length, input_dim, device=dev)
                keep = torch.zeros(1, motion_length, input_dim, dtype=torch.bool, device=dev)
                if i == 0:
                    if fix_very_first:
                        if first_gt is None:
                            raise ValueError('fix_very_first needs first_gt [>= overlap_len, input_dim]')
                        keep[:, :overlap_len] = True
                        gt[:, :overlap_len] = torch.as_tensor(first_gt, dtype=torch.float32)[:overlap_len].to(dev)
                else:
                    keep[:, :overlap_len] = True
                    gt[:, :overlap_len] = outputs[:, -overlap_len:]
                y = dict(gt=gt, outpainting_mask=keep)
            kw['y'] = y
        out = model(**kw)
        pred = out[0]['pred_motion'][:motion_length].detach().cpu().numpy()
        windows.append(pred)
        den = pred * std + mean                                   # m2d_test.py:206
        outputs = torch.as_tensor(den if gt_space == 'denormalised' else pred, dtype=torch.float32).unsqueeze(0).to(dev)
        pieces_repaint.append(den if i == n_win - 1 else den[:stride])          # :214-217
        pieces_plain.append(den if i == 0 else den[pre_frames:])                # :219-222
    rec = np.concatenate(pieces_repaint if repaint else pieces_plain, axis=0)   # :227-232
    return rec.astype(np.float32), windows


# ---- batched window driver ---------------------------------------------------------------------------------------------------
# BASELINE configs[3] ("batch=128, 512-frame long-sequence chunked attention, 4 GPUs") and SURVEY.md section 5 "Long-context": the
# reference's finedance_eval walks ONE window of ONE sequence per model call (tools/m2d_test.py:139-232, B = 1).  The windows of a
# sequence are independent when repaint=False (m2d_test.py:146-153 only rewrites `motion`, which the test branch never reads), and in
# RePaint mode window i only needs window i-1 of the SAME sequence -- so S sequences x W windows fold into the batch:
#   repaint=False   one batch of all S*W windows (chunked to `max_batch` windows per model call)
#   repaint=True    window i of all S sequences per batch, sequentially over i
# and the sequences shard over the ranks of torch.distributed (dist.py: no per-step collective, one gather of the stitched motions).
# NB the tutel MoE capacity couples the tokens of one model call (capacity = 1.5 x tokens / experts, batch-prioritised dropping), so
# a window sampled inside a batch is NOT bit-equal to the same window sampled alone: parity is "this batch on the oracle", exactly as
# for every other batched call of the path; the STITCHING is identical to sample_long's.


def stitch_windows(windows, pre_frames, repaint, mean=None, std=None):
    """De-normalise and stitch the windows of ONE sequence exactly as ``sample_long`` / m2d_test.py:206-232 do:
    RePaint mode keeps the first ``stride`` frames of every window but the last, plain mode cuts the first ``pre_frames``
    frames of every window but the first."""
    L = windows[0].shape[0]
    stride = L - pre_frames
    mean = 0.0 if mean is None else np.asarray(mean)
    std = 1.0 if std is None else np.asarray(std)
    den = [np.asarray(w) * std + mean for w in windows]
    n = len(den)
    if repaint:
        parts = [d if i == n - 1 else d[:stride] for i, d in enumerate(den)]
    else:
        parts = [d if i == 0 else d[pre_frames:] for i, d in enumerate(den)]
    return np.concatenate(parts, axis=0).astype(np.float32)


def rank_sequences(n_seq, rank, ws):
    """Contiguous block of sequence indices rank `rank` of `ws` owns (uneven counts allowed: the first ranks take one more)."""
    base, extra = divmod(n_seq, ws)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def _gather_ragged(local, lengths, owners, input_dim, device):
    """all-gather of per-sequence motions of different lengths in ONE collective (dist.gather_results): every rank pads its block to
    the longest sequence of the job.  No length exchange: a stitched motion has (n_windows - 1) * stride + window frames, which every
    rank computes from the global `total_frames` list (`lengths`: sequence -> frames)."""
    from . import dist as mcd
    if not mcd.is_dist():
        return local
    rank, ws = mcd.world()
    per = max(len(o) for o in owners)
    maxlen = max(lengths)
    buf = torch.zeros(per, maxlen, input_dim, device=device)
    for k, s in enumerate(owners[rank]):
        if local[s].shape[0] != lengths[s]:
            raise RuntimeError(f'sequence {s}: stitched {local[s].shape[0]} frames, expected {lengths[s]}')
        buf[k, :lengths[s]] = torch.as_tensor(local[s], device=device)
    allb = mcd.gather_results(buf).view(ws, per, maxlen, input_dim).cpu().numpy()
    out = {}
    for r in range(ws):
        for k, s in enumerate(owners[r]):
            out[s] = allb[r, k, :lengths[s]]
    return out


def sample_long_batched(model, total_frames, motion_length, pre_frames=30, c=None, text=None, repaint=False, overlap_len=30,
                        fix_very_first=True, first_gt=None, mean=None, std=None, gt_space='denormalised', input_dim=322,
                        device=None, condition_kwargs=None, inference_kwargs=None, max_batch=160, shard=True):
    """S sequences x their windows through as few model calls as the mode allows (see the block comment above).

    total_frames: int (every sequence) or a list of S ints.  c: list of S per-frame conditions [total_s, F] (or a tensor
    [S, total, F]) or None.  text: list of S prompts (or one string).  condition_kwargs: per-SEQUENCE tensors with leading dimension S
    (``xf_out`` [S, 77, *] / ``clip_feat``); every window of sequence s gets row s.  first_gt: [S, >= overlap_len, input_dim] (RePaint,
    fix_very_first).  inference_kwargs: dict, or callable(pairs) -> dict for one model call, ``pairs`` = [(sequence, window), ...] in
    batch order (noise [b, L, C] / step_noise for parity runs).  max_batch: windows per model call (the per-GPU batch; BASELINE
    configs[3]: 128 sequences x 5 windows / 4 GPUs = 160).  shard: split the SEQUENCES over the ranks of an initialised process group
    (contiguous blocks) and gather the stitched motions on every rank.

    Returns (list of S stitched de-normalised motions [frames_s, input_dim] float32 numpy -- ALL S sequences on every rank, gathered
    by one padded all-gather --, dict (sequence, window) -> window output of THIS RANK's sequences only: the raw windows are not
    gathered)."""
    dev = device or torch.device('cuda', torch.cuda.current_device())
    S = len(total_frames) if not isinstance(total_frames, int) else (len(c) if c is not None else (len(text) if isinstance(text, (list, tuple)) else 1))
    totals = [int(total_frames)] * S if isinstance(total_frames, int) else [int(t) for t in total_frames]
    texts = list(text) if isinstance(text, (list, tuple)) else [text or ''] * S
    if len(texts) != S or (c is not None and len(c) != S):
        raise ValueError('text / c must have one entry per sequence')
    n_wins = []
    for t in totals:
        n, stride = window_starts(t, motion_length, pre_frames)
        if n < 1:
            raise ValueError('sequence shorter than one window')
        n_wins.append(n)
    from . import dist as mcd
    rank, ws = mcd.world() if shard else (0, 1)
    owners = [rank_sequences(S, r, ws) for r in range(ws)]
    mine = owners[rank]
    stride = motion_length - pre_frames
    cs = None if c is None else [torch.as_tensor(x, dtype=torch.float32) for x in c]
    cond = {k: v for k, v in (condition_kwargs or {}).items()}
    for k, v in cond.items():
        if not torch.is_tensor(v) or v.shape[0] != S:
            raise ValueError(f'condition_kwargs[{k!r}] must be a tensor with one row per sequence ({S})')
    windows, prev = {}, {}

    def run(pairs):
        b = len(pairs)
        kw = dict(motion=torch.zeros(b, motion_length, input_dim, device=dev), motion_mask=torch.ones(b, motion_length, device=dev),
                  motion_length=torch.full((b, 1), motion_length, device=dev).long(), num_intervals=1,
                  motion_metas=[{'text': texts[s]} for s, _ in pairs])
        if cs is not None:
            kw['c'] = torch.stack([cs[s][w * stride:w * stride + motion_length] for s, w in pairs]).to(dev)
        idx = torch.tensor([s for s, _ in pairs])
        for k, v in cond.items():
            kw[k] = v[idx.to(v.device)]
        kw['inference_kwargs'] = inference_kwargs(pairs) if callable(inference_kwargs) else dict(inference_kwargs or {})
        if repaint:
            y = {}
            if overlap_len > 0:
                gt = torch.zeros(b, motion_length, input_dim, device=dev)
                keep = torch.zeros(b, motion_length, input_dim, dtype=torch.bool, device=dev)
                for j, (s, w) in enumerate(pairs):
                    if w == 0:
                        if fix_very_first:
                            if first_gt is None:
                                raise ValueError('fix_very_first needs first_gt [S, >= overlap_len, input_dim]')
                            keep[j, :overlap_len] = True
                            gt[j, :overlap_len] = torch.as_tensor(first_gt[s], dtype=torch.float32)[:overlap_len].to(dev)
                    else:
                        keep[j, :overlap_len] = True
                        gt[j, :overlap_len] = prev[s][-overlap_len:]
                y = dict(gt=gt, outpainting_mask=keep)
            kw['y'] = y
        out = model(**kw)
        for j, (s, w) in enumerate(pairs):
            pred = out[j]['pred_motion'][:motion_length].detach().cpu().numpy()
            windows[(s, w)] = pred
            if repaint:          # what the next window of this sequence keeps (m2d_test.py:193,206-212: the DE-normalised frames by default)
                den = pred * (1.0 if std is None else np.asarray(std)) + (0.0 if mean is None else np.asarray(mean))
                prev[s] = torch.as_tensor(den if gt_space == 'denormalised' else pred, dtype=torch.float32).to(dev)

    if repaint:
        for w in range(max((n_wins[s] for s in mine), default=0)):
            todo = [(s, w) for s in mine if w < n_wins[s]]
            for k in range(0, len(todo), max_batch):
                run(todo[k:k + max_batch])
    else:
        todo = [(s, w) for s in mine for w in range(n_wins[s])]
        for k in range(0, len(todo), max_batch):
            run(todo[k:k + max_batch])
    local = {s: stitch_windows([windows[(s, w)] for w in range(n_wins[s])], pre_frames, repaint, mean, std) for s in mine}
    lengths = [(n_wins[s] - 1) * stride + motion_length for s in range(S)]
    full = _gather_ragged(local, lengths, owners, input_dim, dev) if (shard and mcd.is_dist()) else local
    return [full[s] for s in range(S)], windows

