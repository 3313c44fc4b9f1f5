"""Result post-processing + on-disk format of the 322-d SMPL-X motion.

Mirrors what the reference tools do with ``output[i]['pred_motion']`` on the host
(``tools/visualize.py:217-263``, ``tools/s2g_visualize.py:236-246``, ``tools/s2g_test.py:289-297,431-448``):
de-normalise with the dataset's ``mean.npy`` / ``std.npy``, re-pack the 322 channels into SMPL-X
``poses[165] / expressions[100] / trans[3]``, smooth every channel over time with
``scipy.ndimage.gaussian_filter(sigma, mode="nearest")`` and ``np.savez`` the AMASS-style file.  Here the
arithmetic runs in one HIP kernel (``mc_postprocess_smplx``) on the sampler's output while it is still in HBM;
only the finished arrays cross PCIe.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib as _lib

# per-tool filter widths: (body + jaw, hands, trans, expressions); None = unfiltered
SIGMAS_T2M = (3.5, 3.5, 3.0, 2.0)      # tools/visualize.py:244-246 (whole poses array at 3.5)
SIGMAS_S2G = (3.5, 1.0, 3.5, None)     # tools/s2g_visualize.py:243-245


def gaussian_taps(sigma, truncate=4.0):
    """The normalised taps scipy.ndimage.gaussian_filter1d correlates with (order 0): radius int(truncate*sigma+.5)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    w = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return radius, w / w.sum()


def _filter_operands(C, mean, std, sigmas, dev):
    mean = np.zeros(C) if mean is None else np.asarray(mean)         # visualize.py:187-190 default: 0 and 1
    std = np.ones(C) if std is None else np.asarray(std)
    if mean.shape != (C,) or std.shape != (C,):
        raise ValueError('mean / std must have shape (322,)')
    stats_f32 = int(mean.dtype == np.float32 and std.dtype == np.float32)
    mean_d = torch.from_numpy(mean.astype(np.float64)).to(dev)
    std_d = torch.from_numpy(std.astype(np.float64)).to(dev)
    taps = np.zeros((4, _lib.POST_MAXTAP), np.float64)
    radius = (ctypes.c_int32 * 4)()
    for g, sg in enumerate(sigmas):
        if sg is None or sg <= 0:
            radius[g] = -1
            continue
        r, w = gaussian_taps(sg)
        if 2 * r + 1 > _lib.POST_MAXTAP:
            raise ValueError(f'sigma={sg} needs {2 * r + 1} taps (max {_lib.POST_MAXTAP})')
        radius[g] = r
        taps[g, :2 * r + 1] = w
    return mean_d, std_d, torch.from_numpy(taps).to(dev), radius, stats_f32


def _check_pred(x):
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3):
        raise ValueError('pred_motion must be a contiguous float32 [B,T,322] tensor in device (HBM) memory')
    if x.shape[2] != 322:
        raise ValueError(f'SMPL-X post-processing expects the 322-d motionx layout, got {x.shape[2]}')


_p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def postprocess_smplx(pred_motion, motion_length=None, mean=None, std=None, sigmas=SIGMAS_T2M):
    """Every sample filtered on its own: pred_motion [B,T,322] fp32 device tensor (normalised sampler output) -> dict of
    device fp64 tensors poses [B,T,165], expressions [B,T,100], trans [B,T,3]; frames >= motion_length[b] are zero.
    (The file the T2M tool saves filters AFTER stitching the intervals: ``postprocess_smplx_stitched``.)"""
    lib = _lib.load(require_gpu=True)
    x = pred_motion
    _check_pred(x)
    B, T, C = x.shape
    dev = x.device
    mean_d, std_d, taps_d, radius, stats_f32 = _filter_operands(C, mean, std, sigmas, dev)
    len_d = None
    if motion_length is not None:
        len_d = torch.as_tensor(motion_length).reshape(-1).to(device=dev, dtype=torch.int32).contiguous()
        if len_d.numel() != B:
            raise ValueError('motion_length must have one entry per sample')
    poses = torch.empty(B, T, 165, device=dev, dtype=torch.float64)
    expr = torch.empty(B, T, 100, device=dev, dtype=torch.float64)
    trans = torch.empty(B, T, 3, device=dev, dtype=torch.float64)
    _lib.check(lib.mc_postprocess_smplx(_p(x), _p(len_d), _p(mean_d), _p(std_d), _p(taps_d), ctypes.byref(radius), stats_f32,
                                        B, T, C, _p(poses), _p(expr), _p(trans),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               'mc_postprocess_smplx')
    return dict(poses=poses, expressions=expr, trans=trans, stats_f32=bool(stats_f32))


def postprocess_smplx_stitched(pred_motion, motion_length, mean=None, std=None, sigmas=SIGMAS_T2M):
    """tools/visualize.py:216-246 for ``num_intervals`` >= 1: the valid frames ``pred[b, :motion_length[b]]`` of all
    intervals are concatenated FIRST and the Gaussian filter runs over the stitched sequence (the smoothing crosses the
    interval seams; only the two ends of the whole sequence replicate their edge frame).  Returns device fp64 tensors
    poses [sum(len),165], expressions [sum(len),100], trans [sum(len),3]."""
    lib = _lib.load(require_gpu=True)
    x = pred_motion
    _check_pred(x)
    B, T, C = x.shape
    lens = [int(v) for v in torch.as_tensor(motion_length).reshape(-1)]
    if len(lens) != B or any(n < 0 or n > T for n in lens):
        raise ValueError('motion_length must have one entry in [0, T] per interval')
    dev = x.device
    rows = np.concatenate([b * T + np.arange(n, dtype=np.int32) for b, n in enumerate(lens)] or [np.zeros(0, np.int32)])
    n = int(rows.size)
    rows_d = torch.from_numpy(rows.astype(np.int32)).to(dev)
    mean_d, std_d, taps_d, radius, stats_f32 = _filter_operands(C, mean, std, sigmas, dev)
    poses = torch.empty(n, 165, device=dev, dtype=torch.float64)
    expr = torch.empty(n, 100, device=dev, dtype=torch.float64)
    trans = torch.empty(n, 3, device=dev, dtype=torch.float64)
    if n:
        _lib.check(lib.mc_postprocess_smplx_stitched(_p(x), _p(rows_d), n, _p(mean_d), _p(std_d), _p(taps_d),
                                                     ctypes.byref(radius), stats_f32, C, _p(poses), _p(expr), _p(trans),
                                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   'mc_postprocess_smplx_stitched')
    return dict(poses=poses, expressions=expr, trans=trans, stats_f32=bool(stats_f32))


def smplx_npz_dict(post):
    """Arrays of the AMASS-style file the tools save (visualize.py:247-256) from ``postprocess_smplx_stitched``: poses
    fp64; expressions / trans keep the dtype numpy would have produced (float32 when mean/std are float32 files)."""
    dt = np.float32 if post['stats_f32'] else np.float64
    return dict(betas=np.zeros(300), poses=post['poses'].cpu().numpy(), expressions=post['expressions'].cpu().numpy().astype(dt),
                trans=post['trans'].cpu().numpy().astype(dt), model='smplx2020', gender='neutral', mocap_frame_rate=30)


def result_name(text, motion_length):
    """visualize.py:247: 'res_' + caption with '/', ' ' -> '_' and '.' removed + '_<length>'."""
    return 'res_' + text.replace('/', '_').replace(' ', '_').replace('.', '') + f'_{int(motion_length)}'


def save_smplx_npz(save_path, text, pred_motion, motion_length, mean=None, std=None, sigmas=SIGMAS_T2M):
    post = postprocess_smplx_stitched(pred_motion, motion_length, mean, std, sigmas)
    d = smplx_npz_dict(post)
    lens = torch.as_tensor(motion_length).reshape(-1)
    path = os.path.join(save_path, result_name(text, lens[0]) + '.npz')
    np.savez(path, **d)
    return path
