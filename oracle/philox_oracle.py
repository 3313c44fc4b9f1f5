"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the library's device noise stream (mc_sample_loop with noise_dev == NULL,
mc_op_philox_normal; motioncraft_amd/csrc/mc_kernels.hip `philox_normal4`).

The reference draws the per-step noise with ``th.randn_like(x)`` (mogen/models/utils/gaussian_diffusion.py:684, 847), i.e. from
whatever generator torch has; a fused device-side loop needs a counter-based generator of its own.  The integer part is
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123) and is pinned by
the published known-answer vectors (tests/test_oracle.py); the float part is two Box-Muller pairs per 4-word block.

counter = (element // 4 [lo, hi], draw index [lo, hi]), key = (seed lo, seed hi); element 4g + j takes normal j of block g."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key [2] or [..., 2] uint32 -> [..., 4] uint32."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.broadcast_to(np.asarray(key, dtype=np.uint32), c.shape[:-1] + (2,)).copy()
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c[..., 0].astype(np.uint64)
            p1 = M1 * c[..., 2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[..., 1] ^ k[..., 0]
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[..., 3] ^ k[..., 1]
            c = np.stack([n0, (p1 & MASK).astype(np.uint32), n2, (p0 & MASK).astype(np.uint32)], axis=-1)
            k = np.stack([k[..., 0] + W0, k[..., 1] + W1], axis=-1)
    return c


def draw_bits(n, seed, draw):
    """The raw words of draw `draw`: [n] uint32."""
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    ctr = np.stack([(g & MASK).astype(np.uint32), (g >> np.uint64(32)).astype(np.uint32),
                    np.full(g.shape, draw & 0xFFFFFFFF, np.uint32), np.full(g.shape, (draw >> 32) & 0xFFFFFFFF, np.uint32)], axis=-1)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32_10(ctr, key).reshape(-1)[:n]


def draw_normal(n, seed, draw, dtype=np.float32):
    """The normals of draw `draw`: words (r0, r1), (r2, r3) of a block -> Box-Muller pairs, u = r 2^-32 + 2^-33 (fp32 like the kernel:
    (float) r rounds to 24 bits first)."""
    r = draw_bits(4 * ((n + 3) // 4), seed, draw).reshape(-1, 2, 2)
    f = r.astype(np.float32)
    u1 = (f[..., 0] * np.float32(2.3283064365386963e-10) + np.float32(1.1641532182693481e-10)).astype(dtype)
    u2 = (f[..., 1] * np.float32(2.3283064365386963e-10) + np.float32(1.1641532182693481e-10)).astype(dtype)
    rad = np.sqrt(dtype(-2.0) * np.log(u1))
    ang = dtype(2.0 * np.pi) * u2
    z = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1)
    return z.reshape(-1)[:n].astype(dtype)
