"""CPU restatement of the reference tools' result post-processing.  TEST INFRASTRUCTURE ONLY.

Follows tools/visualize.py:217-246 (T2M) and tools/s2g_visualize.py:236-245 (S2G) literally, calling the same
third-party routine the reference calls (scipy.ndimage.gaussian_filter, mode="nearest"; scipy is a dependency of the
reference and is present in this image), so the filter itself is pinned by construction.  The reference code sits
inside script ``main()`` bodies that cannot be imported (mmcv / smplx / pytorch3d imports), hence no golden file.
"""
import numpy as np
from scipy.ndimage import gaussian_filter


def _filter_cols(a, sigma):
    """motion_temporal_filter_wo_reshape (visualize.py:39-44): column by column, in place."""
    for i in range(a.shape[1]):
        a[:, i] = gaussian_filter(a[:, i], sigma=sigma, mode="nearest")
    return a


def denormalise(pred_list, mean, std):
    """visualize.py:217-223: per interval pred[:length] * std + mean, concatenated."""
    return np.concatenate([p * std + mean for p in pred_list], axis=0)


def repack(pred_motion):
    """visualize.py:236-243 / s2g_test.py:289-297."""
    T = pred_motion.shape[0]
    pose = np.zeros((T, 165))
    pose[:, :3 + 63] = pred_motion[:, :3 + 63]
    pose[:, 66:66 + 3] = pred_motion[:, 66 + 90:66 + 93]
    pose[:, 66 + 9:66 + 90 + 9] = pred_motion[:, 66:66 + 90]
    return pose, pred_motion[:, 209:209 + 100].copy(), pred_motion[:, 309:309 + 3].copy()


def t2m_result(pred_motion):
    """visualize.py:244-246 on ONE interval's de-normalised [T,322] array."""
    pose, expr, trans = repack(pred_motion)
    return _filter_cols(pose, 3.5), _filter_cols(expr, 2.0), _filter_cols(trans, 3.0)


def s2g_result(pred_motion):
    """s2g_visualize.py:243-245 (expressions unfiltered)."""
    pose, expr, trans = repack(pred_motion)
    pose[:, :66 + 3] = _filter_cols(pose[:, :66 + 3].copy(), 3.5)
    pose[:, 66 + 9:66 + 90 + 9] = _filter_cols(pose[:, 66 + 9:66 + 90 + 9].copy(), 1.0)
    return pose, expr, _filter_cols(trans, 3.5)
