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


def postprocess_smplx(pred_motion, motion_length=None, mean=None, std=None, sigmas=SIGMAS_T2M):
    """pred_motion [B,T,322] fp32 device tensor (normalised sampler output) -> dict of device fp64 tensors
    poses [B,T,165], expressions [B,T,100], trans [B,T,3]; frames >= motion_length[b] are zero."""
    lib = _lib.load(require_gpu=True)
    x = pred_motion
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3):
        raise ValueError('pred_motion must be a contiguous float32 [B,T,322] tensor in device (HBM) memory')
    B, T, C = x.shape
    if C != 322:
        raise ValueError(f'SMPL-X post-processing expects the 322-d motionx layout, got {C}')
    mean = np.zeros(C) if mean is None else np.asarray(mean)         # visualize.py:187-190 default: 0 and 1
    std = np.ones(C) if std is None else np.asarray(std)
    if mean.shape != (C,) or std.shape != (C,):
        raise ValueError('mean / std must have shape (322,)')
    stats_f32 = int(mean.dtype == np.float32 and std.dtype == np.float32)
    dev = x.device
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
    taps_d = torch.from_numpy(taps).to(dev)
    len_d = None
    if motion_length is not None:
        len_d = torch.as_tensor(motion_length).reshape(-1).to(device=dev, dtype=torch.int32).contiguous()
        if len_d.numel() != B:
            raise ValueError('motion_length must have one entry per sample')
    poses = torch.empty(B, T, 165, device=dev, dtype=torch.float64)
    expr = torch.empty(B, T, 100, device=dev, dtype=torch.float64)
    trans = torch.empty(B, T, 3, device=dev, dtype=torch.float64)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(lib.mc_postprocess_smplx(p(x), p(len_d), p(mean_d), p(std_d), p(taps_d), ctypes.byref(radius), stats_f32,
                                        B, T, C, p(poses), p(expr), p(trans),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               'mc_postprocess_smplx')
    return dict(poses=poses, expressions=expr, trans=trans, stats_f32=bool(stats_f32))


def smplx_npz_dict(post, motion_length):
    """Arrays of the AMASS-style file the tools save (visualize.py:247-256): the intervals' valid frames
    concatenated (:218-223), poses fp64; expressions / trans keep the dtype numpy would have produced
    (float32 when mean/std are float32 files)."""
    lens = [int(v) for v in torch.as_tensor(motion_length).reshape(-1)]
    cat = lambda t: np.concatenate([t[b, :n].cpu().numpy() for b, n in enumerate(lens)], axis=0)
    dt = np.float32 if post['stats_f32'] else np.float64
    return dict(betas=np.zeros(300), poses=cat(post['poses']), expressions=cat(post['expressions']).astype(dt),
                trans=cat(post['trans']).astype(dt), model='smplx2020', gender='neutral', mocap_frame_rate=30)


def result_name(text, motion_length):
    """visualize.py:247: 'res_' + caption with '/', ' ' -> '_' and '.' removed + '_<length>'."""
    return 'res_' + text.replace('/', '_').replace(' ', '_').replace('.', '') + f'_{int(motion_length)}'


def save_smplx_npz(save_path, text, pred_motion, motion_length, mean=None, std=None, sigmas=SIGMAS_T2M):
    post = postprocess_smplx(pred_motion, motion_length, mean, std, sigmas)
    d = smplx_npz_dict(post, motion_length)
    lens = torch.as_tensor(motion_length).reshape(-1)
    path = os.path.join(save_path, result_name(text, lens[0]) + '.npz')
    np.savez(path, **d)
    return path
