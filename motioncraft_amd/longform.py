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
