#!/usr/bin/env python
"""Headline benchmark: sampled SMPL-X frames/s of the STMoGen 0.125b denoiser, 196-frame sequences,
1000-step DDPM with classifier-free guidance, synthetic inputs (BASELINE.json configs[1]:
batch=64 per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: denoiser (CFG-doubled) + CFG combine +
p_sample update, including the per-step noise draw (on the device, inside the sampler-update
kernel: mc_sample_loop).  EXACTLY K consecutive steps of the loop, issued as one mc_sample_loop call (as a sampling run issues
them), are timed between barriers + device syncs; MAX over ranks.  `value` = whole-job frames/s of the COMPLETE 1000-step loop:

    value = N * B * T / (t_setup + 1000 * t_step + t_gather)

where t_setup = once-per-batch work that the loop needs (RCCL broadcast of the condition
embeddings, FiLM/time tables for all 1000 steps, per-layer text K/V) and t_gather = RCCL
all-gather of the finished sequences; both are measured here and reported in `config`.
Inputs are resident in HBM when the timed region starts.

Extra objects on the JSON line:
  roofline      bound "mfma": algorithmic FLOPs of one step (SURVEY.md section 8d figure x samples)
                / mean step duration from HIP events on the launch stream, vs the 157.3 TFLOP/s
                fp32-matrix peak (v_mfma_f32_32x32x2_f32; MI355X_MICROARCH.md)
  cpu_baseline  the CPU oracle (oracle/stmogen_oracle.py, kind "port") timed on this host's cores,
                rank 0, N=1 only, bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_FP16_MFMA_TFLOPS = 2500.0        # dense (MI355X_MICROARCH.md)
TOTAL_DDPM_STEPS = 1000


def measured_hbm_traffic():
    """HBM bytes of one B=64 step.  PMC counters cannot be collected from inside this process, so the number is READ from the
    newest tracked rocprofv3 summary profiles/rNN_pmc_hbm_traffic.txt (two separate --pmc passes, FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, written by tools/hbm_traffic.py; its header names the command and the commit it was taken
    at).  Returns (GB per step or None, source file)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.txt')))
    if not files:
        return None, None
    src = files[-1]
    m, commit = None, None
    for line in open(src):
        mm = re.match(r'# per denoising step .*= ([0-9.]+) GB', line)
        if mm:
            m = mm
        mc_ = re.match(r'# commit (\S+)', line)
        if mc_:
            commit = mc_.group(1)
    # (the file goes stale when kernels change: its header carries the commit it was measured at, printed in the line)
    return (float(m.group(1)) if m else None), os.path.relpath(src, ROOT) + (f' @ commit {commit}' if commit else '')

DIMS = dict(input_feats=322, max_seq_len=196, L=128, H=12, NL=4, F=512, Te=2048, Dt=256, Nt=77, E=16, topk=2,
            scale=6.5)


def algorithmic_flops_per_sample_step(d, T):
    """multiply-add = 2 flops; per CFG half per layer, then x2 x NL + enc/dec."""
    L, H, F, Nt, E = d['L'], d['H'], d['F'], d['Nt'], d['E']
    D = L * H
    TH = T * H
    layer = (TH * (2 * L * 256 + 2 * 256 * E) + TH * d['topk'] * 16 * L * L + TH * 8 * L * L + T * 2 * H * H * L
             + TH * 6 * L * L + T * 8 * (2 * 12 * (L // 8) ** 2) * 2 + H * (2 * (Nt + T) * L * L + 2 * T * L * L)
             + 2 * T * 2 * D * D + TH * 4 * L * F)
    return 2 * d['NL'] * layer + T * 2 * 644 * L + 2 * T * 2 * 644 * L


def executed_flops_per_sample_step(d, T):
    """FLOPs the kernels actually LAUNCH per sample and step (useful MFMA work, tile padding not counted), against the reference's count
    above: base layer 0 runs gate / experts / proj / q,k,v / body topology for ONE CFG half (twin dedupe), the last StylizationBlock
    Linear + pose decoder run once on the CFG-combined rows as one [322, 2D] product (folded tail), the pose encoder / decoder are dense
    [D, 324] / [322, D] products of the packed weights instead of the reference's per-part ones (more FLOPs, one GEMM)."""
    L, H, F, Nt, E = d['L'], d['H'], d['F'], d['Nt'], d['E']
    D = L * H
    TH = T * H
    front = (TH * (2 * L * 256 + 2 * 256 * E) + TH * d['topk'] * 16 * L * L + TH * 8 * L * L + T * 2 * H * H * L
             + TH * 6 * L * L + T * 8 * (2 * 12 * (L // 8) ** 2) * 2)                 # gate, experts, proj, static topology, q/k/v, dynamic topology
    temporal = H * (2 * (Nt + T) * L * L + 2 * T * L * L)
    film = T * 2 * D * D
    sffn = TH * 4 * L * F
    total = T * 2 * 324 * D                                  # pose encoder, written to both halves by one GEMM
    for i in range(d['NL']):
        total += (1 if i == 0 else 2) * front + 2 * (temporal + film + sffn)
        total += 0 if i == d['NL'] - 1 else 2 * film          # the SFFN's FiLM Linear; the last one is folded into the tail
    return total + T * 2 * 322 * 2 * D                        # folded tail on the combined rows


def csrc_digest():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'motioncraft_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def lone_dominant_kernel():
    """The dominant kernel's LONE-launch figure (serial single-stream schedule, so a launch's duration is the kernel's own): read from the
    newest tracked profiles/rNN_kernel_roofline.txt (tools/kernel_roofline.py over the rocprofv3 --pmc pass; its header carries the
    commit).  HIP events around in-step launches of the two-stream schedule measure the overlap, not the kernel."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_roofline.txt')))
    if not files:
        return None
    src, commit, row, digest = files[-1], None, None, None
    for line in open(src):
        mc_ = re.match(r'# commit (\S+)', line)
        if mc_:
            commit = mc_.group(1)
        md = re.match(r'# csrc sha256 (\S+)', line)
        if md:
            digest = md.group(1)
        if row is None and line.startswith('gemm_wp_k'):
            row = line.split()
    if not row:
        return None
    return {'kernel': 'gemm_wp_k (fp32 MFMA 32x32x2, wave-private LDS-DMA ring): FiLM out_layers GEMMs h += a W^T + b (stylization_block.py:39) + pose encoder',
            'avg_us': float(row[3]), 'gflop_per_launch': float(row[4]), 'achieved': float(row[5]), 'frac': round(float(row[6]) / 100, 4),
            'mfma_util_pct': float(row[7]), 'clock_ghz': float(row[8]),
            'source': os.path.relpath(src, ROOT) + (f' @ commit {commit}' if commit else ''),
            # the table is a tracked profile, not measured in this run: `stale` = the kernel sources it was profiled from are not the ones of
            # the library being timed now (sha256 over motioncraft_amd/csrc, written by tools/kernel_roofline.py; None = the table carries no digest)
            'stale': (digest != csrc_digest()) if digest else None,
            'how': 'lone whole-batch launches of the serial schedule, rocprofv3 --kernel-trace --pmc pass (duration and counters of the same dispatches)'}


def synth_condition(B, seed):
    g = torch.Generator().manual_seed(seed)
    xf = torch.nn.functional.layer_norm(torch.randn(B, DIMS['Nt'], DIMS['Dt'], generator=g), (DIMS['Dt'],))
    return xf


def cpu_baseline(B=8, T=196, steps=6):
    """The oracle on the host cores: `steps` DDPM steps at batch B, same synthetic inputs/weights.
    torch-CPU does not scale to every core of a many-core host (256-thread runs are >100x slower
    than 16-32 threads), so the thread count is calibrated first on one B=1 denoiser call and the
    fastest setting is used and reported as `cores`."""
    from oracle import stmogen_oracle as O, weights as W
    dims = W.default_dims()
    sd = W.make_state_dict(dims, 0)
    g = torch.Generator().manual_seed(0)
    x_T = torch.randn(B, T, 322, generator=g)
    xf = synth_condition(B, 1)
    mask = torch.ones(B, T)
    sched = O.Schedule(1000, None)
    ncpu = os.cpu_count() or 1
    best, best_t = 1, float('inf')
    tf1 = None
    for nt in sorted({min(ncpu, v) for v in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        if tf1 is None:
            tf1 = O.precompute_text(sd, xf[:1], dims)
        O.denoise(sd, dims, x_T[:1], 999, xf[:1], mask[:1], text_feats=tf1)       # warm
        t0 = time.time()
        O.denoise(sd, dims, x_T[:1], 999, xf[:1], mask[:1], text_feats=tf1)
        dt = time.time() - t0
        if dt < best_t:
            best, best_t = nt, dt
        if dt > 4 * best_t or dt > 20:
            break
    torch.set_num_threads(best)
    t0 = time.time()
    tf = O.precompute_text(sd, xf, dims)
    t_text = time.time() - t0
    x = x_T
    t0 = time.time()
    for n in range(steps):
        i = 999 - n
        x0 = O.denoise(sd, dims, x, sched.timestep_map[i], xf, mask, text_feats=tf)
        x = O.ddpm_step(sched, i, x, x0, torch.randn(x.shape, generator=g))
    t_step = (time.time() - t0) / steps
    out = dict(value=round(B * T / (t_text + TOTAL_DDPM_STEPS * t_step), 4), unit='frames/s',
               cores=best, kind='port',
               sample=f'oracle/stmogen_oracle.py (torch-CPU fp32, {best} threads = fastest of 8/16/32/64 on a '
                      f'{ncpu}-CPU host), batch {B}, {steps} of 1000 DDPM steps timed ({t_step:.2f} s/step) + text '
                      f'K/V hoist ({t_text:.2f} s), extrapolated to the full loop')
    # BASELINE configs[0] ("timed for config 1 exactly"): batch 1, 196 frames, the COMPLETE 50-step DDIM loop
    # (configs/stmogen/T2M_motionx_align.py:90-98 respace '15,15,8,6,6'; the reference's CPU entry is tools/visualize.sh --device cpu)
    sched50 = O.Schedule(1000, '15,15,8,6,6')
    x1, xf1, m1 = x_T[:1], xf[:1], mask[:1]
    t0 = time.time()
    tf1 = O.precompute_text(sd, xf1, dims)
    for i in range(49, -1, -1):
        x0 = O.denoise(sd, dims, x1, sched50.timestep_map[i], xf1, m1, text_feats=tf1)
        x1 = O.ddim_step(sched50, i, x1, x0, torch.zeros_like(x1))
    t_c0 = time.time() - t0
    out['configs0'] = dict(value=round(T / t_c0, 3), unit='frames/s', loop_s=round(t_c0, 3), cores=best,
                           sample='the complete loop, timed exactly: batch 1, 196 frames, 50-step DDIM (eta 0), text K/V hoist included')
    return out


def self_launch_argv(n, argv, port=None):
    """argv that re-runs this script as n ranks on this node (torch.distributed.run, rendezvous on 127.0.0.1)."""
    port = port or int(os.environ.get('MASTER_PORT', 29400 + os.getpid() % 2000))
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def workload_name(B, T):
    base = ('stmogen 0.125b text-to-motion (L=128, 12 parts, 4 layers, 16 experts top-2), CFG scale 6.5, '
            f'batch {B} per GPU, {T} frames, 1000-step DDPM')
    return ('configs[1]: ' if (B, T) == (64, 196) else 'configs[1] at a NON-BASELINE size: ') + base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64, help='samples per GPU (BASELINE configs[1]: 64)')
    ap.add_argument('--frames', type=int, default=196)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the side measurements (dominant-kernel probe, 2 batches in flight)')
    ap.add_argument('--no-full-loop', action='store_true', help='skip the complete 1000-step loop (measured-vs-extrapolated check, ~20 s)')
    ap.add_argument('--backend', default='nccl', help="'nccl' (= RCCL over xGMI); 'gloo' only for plumbing smoke tests")
    a = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU, mogen/apis/test.py:36-82 runs under
        # tools/dist_train.sh:8-10's launcher the same way); rank 0 of the children prints the one JSON line
        os.execv(sys.executable, self_launch_argv(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if a.gpus != world:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU')
    if os.environ.get('MC_BENCH_ALL_ON_DEVICE0'):      # plumbing smoke test of the N>1 path on a 1-GPU box (gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    # MC_BENCH_FORCE_DIST=1: initialise the process group at world size 1 too, so a 1-GPU lease executes the RCCL branch (device
    # broadcast, barrier(device_ids), all_gather_into_tensor, the fp64 max-over-ranks reduce) -- profiles/r06_bench_n1_rccl_world1.json
    use_dist = world > 1 or os.environ.get('MC_BENCH_FORCE_DIST') == '1'
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(a.backend)

    from motioncraft_amd import dist as mcd
    from motioncraft_amd.diffusion import build_diffusion
    from motioncraft_amd.engine import NativeModel
    from motioncraft_amd.synthetic import make_state_dict   # random-init weights (data only, no oracle import)

    B, T, C = a.batch, a.frames, DIMS['input_feats']
    GB = B * world
    sd = make_state_dict(DIMS, 0)
    nm = NativeModel(DIMS, sd, cfg_scale=DIMS['scale'], device=local_rank)
    del sd
    side_streams = [torch.cuda.Stream(), torch.cuda.Stream()]      # for the 2-batches-in-flight side measurement
    ctx = nm.context(B, T, max_steps=TOTAL_DDPM_STEPS)
    diff = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                                model_var_type='fixed_large'))

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local_rank]) if a.backend == 'nccl' else dist.barrier()
        torch.cuda.synchronize()

    # ---- once-per-batch setup: condition broadcast (RCCL), FiLM tables, text K/V ----
    xf_global = torch.empty(GB, DIMS['Nt'], DIMS['Dt'], device=dev)
    mask_global = torch.ones(GB, T, device=dev)
    if rank == 0:
        xf_global.copy_(synth_condition(GB, 1))
    barrier()
    t0 = time.perf_counter()
    xf, mask = mcd.broadcast_condition(xf_global, mask_global, src=0)
    ctx.set_timesteps(diff.timestep_map)
    ctx.set_condition(xf, mask)
    barrier()
    t_setup = time.perf_counter() - t0

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, T, C, device=dev, generator=gen)
    coefs = {i: diff.step_coefs(i, 'ddpm', DIMS['scale']) for i in range(TOTAL_DDPM_STEPS)}

    NOISE_KEY = 0x5EED0000 + rank          # key of the device-side Philox stream (one draw index per step)
    draw = [0]

    def one_step(i):
        # one step through the loop entry (x updated in place, the step's randn_like drawn inside the sampler-update kernel)
        ctx.sample_loop(x, [i], [coefs[i]], noise=None, seed=NOISE_KEY, draw0=draw[0])
        draw[0] += 1

    def run_steps(first, count):
        # `count` consecutive steps of the loop as ONE library call, as a sampling run issues them (the sampler update of a step
        # writes the next step's padded pose-encoder operand, so single-step calls would time a pad pass no loop performs)
        order = [(first - j) % TOTAL_DDPM_STEPS for j in range(count)]
        ctx.sample_loop(x, order, [coefs[j] for j in order], noise=None, seed=NOISE_KEY, draw0=draw[0])
        draw[0] += count
        return (first - count) % TOTAL_DDPM_STEPS

    i = TOTAL_DDPM_STEPS - 1
    if a.warmup > 0:
        i = run_steps(i, a.warmup)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    barrier()
    t0 = time.perf_counter()
    ev[0].record()
    i = run_steps(i, a.steps)          # EXACTLY K steps
    ev[1].record()
    barrier()
    t_loop = time.perf_counter() - t0
    ev_ms = ev[0].elapsed_time(ev[1]) / a.steps

    # ---- the COMPLETE loop once, outside the K-step region: x_T -> x_0 over all 1000 steps, every step index used once
    # (`value` extrapolates 1000 * t_step from the K timed steps; this is the measured counterpart) ----
    t_full = None
    if not a.no_full_loop:
        x.normal_(generator=gen)
        barrier()
        t0 = time.perf_counter()
        order = list(range(TOTAL_DDPM_STEPS - 1, -1, -1))
        ctx.sample_loop(x, order, [coefs[j] for j in order], noise=None, seed=NOISE_KEY, draw0=draw[0])      # ONE library call
        barrier()
        t_full = time.perf_counter() - t0
        assert bool(torch.isfinite(x).all()), 'the 1000-step loop produced non-finite poses'

    # ---- finish: all-gather of the finished sequences (RCCL) ----
    barrier()
    t0 = time.perf_counter()
    out = mcd.gather_results(x)
    barrier()
    t_gather = time.perf_counter() - t0
    assert out.shape[0] == GB and bool(torch.isfinite(out).all())

    # ---- dominant kernel: the lone-launch figure of the newest tracked rocprofv3 table (profiles/rNN_kernel_roofline.txt); in-step HIP events
    # would time the overlap of the two sample groups, not the kernel ----
    dom = lone_dominant_kernel() if rank == 0 else None

    # ---- side measurement (NOT `value`): two independent batches of B in flight on this GPU, one HIP stream and one
    # context each -- what a test loop over many batches of 64 can do; each batch keeps its own MoE capacity domain ----
    inflight2 = None
    if rank == 0 and world == 1 and not a.no_extras:
        try:
            s2 = side_streams
            c2, x2, n2 = [], [], []
            for j in (0, 1):
                with torch.cuda.stream(s2[j]):
                    cj = nm.context(B, T, max_steps=4)
                    cj.set_timesteps(diff.timestep_map[-4:])
                    cj.set_condition(xf, mask)
                    c2.append(cj)
                    x2.append(torch.randn(B, T, C, device=dev, generator=gen))
                    n2.append(torch.empty(B, T, C, device=dev))
            e2 = torch.randn(B, T, C, device=dev, generator=gen)
            torch.cuda.synchronize()

            def both(reps):
                for _ in range(reps):
                    for j in (0, 1):
                        with torch.cuda.stream(s2[j]):
                            c2[j].sample_step(x2[j], 1, coefs[1], e2, x_prev=n2[j])
            both(2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            both(8)
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / 8
            inflight2 = {'frames_per_s': round(2 * B * T / (TOTAL_DDPM_STEPS * dt2), 1), 'ms_per_step_pair': round(dt2 * 1e3, 3),
                         'note': 'side measurement, not `value`: 2 independent batches of 64 per GPU on 2 HIP streams fill each '
                                 "other's tile-count tails"}
            c2[0].close()
            c2[1].close()
        except Exception as e:      # a side measurement must not take the headline line down with it: recorded in the line, not hidden
            inflight2 = {'error': f'{type(e).__name__}: {e}'}

    # ---- side measurement (NOT `value`): the reduced-precision MFMA modes of BASELINE configs[4] on the same workload:
    # 'f16x3' = fp16 hi/lo split operands, three products, fp32 accumulate (fp32-class results: tests hold it to the same
    # 1e-3 lockstep bound, observed 1.3e-5); 'f16' = one fp16 rounding per operand.  Gate / routing / normalisations fp32. ----
    reduced = None
    if rank == 0 and world == 1 and not a.no_extras:
        try:
            reduced = {}
            for prec in ('f16x3', 'f16'):
                cj = nm.context(B, T, max_steps=8)
                cj.set_precision(prec)
                cj.set_timesteps(diff.timestep_map[-8:])
                cj.set_condition(xf, mask)
                xa, xb = torch.randn(B, T, C, device=dev, generator=gen), torch.empty(B, T, C, device=dev)
                e_ = torch.randn(B, T, C, device=dev, generator=gen)
                for _ in range(3):
                    cj.sample_step(xa, 5, coefs[5], e_, x_prev=xb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nrep = 20
                for _ in range(nrep):
                    cj.sample_step(xa, 5, coefs[5], e_, x_prev=xb)
                torch.cuda.synchronize()
                dtp = (time.perf_counter() - t0) / nrep
                reduced[prec] = {'ms_per_step': round(dtp * 1e3, 3),
                                 'frames_per_s': round(B * T / (t_setup + TOTAL_DDPM_STEPS * dtp + t_gather), 1)}
                if prec == 'f16x3':
                    # every fp32 product is three fp16 MFMA products: ceiling = dense fp16 MFMA peak / 3
                    ach16 = algorithmic_flops_per_sample_step(DIMS, T) * B / dtp / 1e12
                    reduced[prec]['roofline'] = {'bound': 'mfma', 'achieved': round(ach16, 2), 'peak': round(PEAK_FP16_MFMA_TFLOPS / 3, 1),
                                                 'unit': 'TFLOP/s (fp32-equivalent)', 'frac': round(ach16 / (PEAK_FP16_MFMA_TFLOPS / 3), 4)}
                cj.close()
            reduced['note'] = ('side measurement, not `value`: mc_ctx_set_precision modes (include/motioncraft_amd.h); the headline '
                               'stays the exact fp32 MFMA path')
        except Exception as e:      # a side measurement must not take the headline line down with it: recorded in the line, not hidden
            reduced = {'error': f'{type(e).__name__}: {e}'}

    # ---- BASELINE configs[0] on the GPU (NOT `value`): batch 1, 196 frames, the complete 50-step DDIM loop, one library call ----
    configs0 = None
    if rank == 0 and world == 1 and not a.no_extras:
        try:
            d50 = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                                       model_var_type='fixed_large', respace='15,15,8,6,6'))
            c1 = nm.context(1, T, max_steps=50)
            c1.set_timesteps(d50.timestep_map)
            c1.set_condition(xf[:1].contiguous(), mask[:1].contiguous())
            k50 = [d50.step_coefs(j, 'ddim', DIMS['scale'], 0.0) for j in range(49, -1, -1)]
            x1 = torch.randn(1, T, C, device=dev, generator=gen)
            ts1 = []
            for rep in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                c1.sample_loop(x1, list(range(49, -1, -1)), k50, noise=None, seed=NOISE_KEY, draw0=10 ** 6 + 50 * rep)
                torch.cuda.synchronize()
                ts1.append(time.perf_counter() - t0)
            c1.close()
            t1 = sorted(ts1[1:])[len(ts1[1:]) // 2]
            configs0 = {'loop_ms': round(t1 * 1e3, 2), 'frames_per_s': round(T / t1, 1),
                        'note': 'side measurement, not `value`: configs[0] (batch 1, 196 frames, 50-step DDIM) as one mc_sample_loop call, median of 5'}
        except Exception as e:      # a side measurement must not take the headline line down with it: recorded in the line, not hidden
            configs0 = {'error': f'{type(e).__name__}: {e}'}

    # ---- BASELINE configs[4] as a measured workload (NOT `value`): mixed text + audio plug-and-play control (0.125b base + 2
    # control copies, pre-encoded audio condition of width D through ControlT2MHalf, controlnet.py:340-424), fp16 MFMA
    # (tools/test.py:95-97 wrap_fp16_model), the COMPLETE 50-step DDIM loop as hipGraph replays (one captured step, device-side step
    # index), per-GPU batch 32 (configs[2]'s 256 / 8), 196 frames ----
    configs4 = None
    if rank == 0 and world == 1 and not a.no_extras:
        try:
            from motioncraft_amd.synthetic import control_param_shapes
            copy, feats, B4 = 2, DIMS['L'] * DIMS['H'], 32
            nm4 = NativeModel(DIMS, make_state_dict(DIMS, 0, shapes=control_param_shapes(DIMS, copy, feats)), cfg_scale=DIMS['scale'], device=local_rank)
            d50 = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                                       model_var_type='fixed_large', respace='15,15,8,6,6'))
            k50 = [d50.step_coefs(j, 'ddim', DIMS['scale'], 0.0) for j in range(50)]
            D_ = DIMS['L'] * DIMS['H']
            # algorithmic FLOPs per sample per step: 4 + 2 DecoderLayers (the formula of SURVEY.md section 8d with NL = 6) + the two
            # after_proj Linear layers of the copies on both CFG halves (before_proj(c) is hoisted: once per batch)
            fl4 = algorithmic_flops_per_sample_step(dict(DIMS, NL=DIMS['NL'] + copy), T) + copy * 2 * T * 2 * D_ * D_
            configs4 = {'workload': f'configs[4]: 0.125b + {copy} control copies, text + audio condition, batch {B4} per GPU, {T} frames, complete 50-step DDIM '
                                    'loop, hipGraph replay', 'algorithmic_gflop_per_sample_step': round(fl4 / 1e9, 3)}
            gs = torch.cuda.Stream()
            with torch.cuda.stream(gs):
                for prec, peak in (('f16x3', PEAK_FP16_MFMA_TFLOPS / 3), ('f16', PEAK_FP16_MFMA_TFLOPS)):
                    c4 = nm4.context(B4, T, max_steps=50)
                    c4.set_precision(prec)
                    c4.set_timesteps(d50.timestep_map)
                    c4.set_condition(xf[:B4].contiguous(), mask[:B4].contiguous())
                    c4.set_control(torch.randn(B4, T, feats, device=dev, generator=gen))
                    x4 = torch.randn(B4, T, C, device=dev, generator=gen)
                    n4 = torch.zeros_like(x4)                   # eta = 0: the draws are not used
                    c4.graph_capture(x4, n4, k50)
                    ts4 = []
                    for rep in range(4):
                        x4.normal_(generator=gen)
                        gs.synchronize()
                        t0 = time.perf_counter()
                        for j in range(49, -1, -1):
                            c4.graph_step(j)
                        gs.synchronize()
                        ts4.append(time.perf_counter() - t0)
                    assert bool(torch.isfinite(x4).all()), 'configs[4] loop produced non-finite poses'
                    # what the replays compute is checked where it is measured: one more replayed step against an EAGER mc_sample_step from the
                    # same x_t (same large-batch two-stream schedule) must give the same bits (tests/test_gpu_parity.py::
                    # test_configs4_as_benched_large_batch_graph_replay_and_lockstep holds the same shape against the oracle)
                    x4.normal_(generator=gen)
                    xs4 = x4.clone()
                    c4.graph_step(7)
                    gs.synchronize()
                    xe4 = c4.sample_step(xs4, 7, k50[7], n4)
                    gs.synchronize()
                    assert torch.equal(x4, xe4), 'configs[4]: hipGraph replay differs from the eager step'
                    t4 = sorted(ts4[1:])[1]
                    ach4 = fl4 * B4 * 50 / t4 / 1e12
                    configs4[prec] = {'loop_ms': round(t4 * 1e3, 2), 'ms_per_step': round(t4 * 20, 3), 'frames_per_s': round(B4 * T / t4, 1),
                                      'roofline': {'bound': 'mfma', 'achieved': round(ach4, 1), 'peak': round(peak, 1),
                                                   'unit': 'TFLOP/s' + (' (fp32-equivalent: three fp16 products per fp32 product)' if prec == 'f16x3' else ''),
                                                   'frac': round(ach4 / peak, 4)}}
                    c4.graph_release()
                    c4.close()
            nm4.close()
            configs4['replay_equals_eager_step'] = True      # asserted above, per mode
            configs4['plain_f16_caveat'] = ('`f16` (one fp16 rounding per operand) is OUTSIDE the north-star 1e-3 tolerance on the x0 prediction (8e-3 at the '
                                            'CFG weights of t = 640; include/motioncraft_amd.h) and meets it per sampler step only; `f16x3` meets every fp32 bound')
            configs4['note'] = ('side measurement, not `value`: median of 3 complete loops after a warm-up loop; gate / routing / normalisations / softmaxes '
                                'stay fp32 in both modes, so the fp16 MFMA ceiling bounds only the GEMM-shaped ~95 % of the FLOPs')
        except Exception as e:      # a side measurement must not take the headline line down with it: recorded in the line, not hidden
            configs4 = {'error': f'{type(e).__name__}: {e}'}

    # ---- BASELINE configs[2] / configs[3] at their per-GPU shares (NOT `value`): the plug-and-play control branch at its real widths,
    # complete 50-step DDIM loops (one mc_sample_loop call each), exact fp32 MFMA.
    #   configs[2]  S2G_Beats2_no_face_loss_025b: L=128, 8 base layers + 2 control copies, audio condition of width D (the WavEncoder output,
    #               encoded once per batch: tools/control_bench.py times it), batch 256 over 8 GPUs = 32 x 196 frames per GPU
    #   configs[3]  M2D_finedance: L=64, F=256, 4 + 3 layers, 35-d music features, 128 sequences x 5 windows of 120 frames over 4 GPUs = 160 windows
    control_cfgs = None
    if rank == 0 and world == 1 and not a.no_extras:
        try:
            from motioncraft_amd.synthetic import control_param_shapes
            control_cfgs = {}
            d50c = build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                                        model_var_type='fixed_large', respace='15,15,8,6,6'))
            for key, over, copy, feats, Bc, Tc in (('configs2_s2g', dict(NL=8), 2, DIMS['L'] * DIMS['H'], 32, 196),
                                                   ('configs3_m2d', dict(L=64, F=256), 3, 35, 160, 120)):
                dm = dict(DIMS, **over)
                nmc = NativeModel(dm, make_state_dict(dm, 0, shapes=control_param_shapes(dm, copy, feats)), cfg_scale=dm['scale'], device=local_rank)
                cc = nmc.context(Bc, Tc, max_steps=50)
                cc.set_timesteps(d50c.timestep_map)
                cc.set_condition(torch.nn.functional.layer_norm(torch.randn(Bc, dm['Nt'], dm['Dt'], device=dev, generator=gen), (dm['Dt'],)),
                                 torch.ones(Bc, Tc, device=dev))
                cc.set_control(torch.randn(Bc, Tc, feats, device=dev, generator=gen))
                kc = [d50c.step_coefs(j, 'ddim', dm['scale'], 0.0) for j in range(49, -1, -1)]
                xc = torch.randn(Bc, Tc, C, device=dev, generator=gen)
                tsc = []
                for rep in range(3):
                    xc.normal_(generator=gen)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    cc.sample_loop(xc, list(range(49, -1, -1)), kc, noise=None, seed=NOISE_KEY, draw0=2 * 10 ** 6 + 50 * rep)
                    torch.cuda.synchronize()
                    tsc.append(time.perf_counter() - t0)
                assert bool(torch.isfinite(xc).all())
                tcf = min(tsc[1:])
                Dm = dm['L'] * dm['H']
                flc = algorithmic_flops_per_sample_step(dict(dm, NL=dm['NL'] + copy), Tc) + copy * 2 * Tc * 2 * Dm * Dm
                achc = flc * Bc * 50 / tcf / 1e12
                control_cfgs[key] = {'batch_per_gpu': Bc, 'frames': Tc, 'layers': f"{dm['NL']}+{copy}", 'latent_dim': dm['L'], 'loop_ms': round(tcf * 1e3, 1),
                                     'ms_per_step': round(tcf * 20, 3), 'frames_per_s_per_gpu': round(Bc * Tc / tcf, 1),
                                     'roofline': {'bound': 'mfma', 'achieved': round(achc, 1), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                                  'frac': round(achc / PEAK_FP32_MFMA_TFLOPS, 4)}}
                cc.close()
                nmc.close()
                if key == 'configs3_m2d':
                    # the same 160 windows through the PRODUCT entry point: longform.sample_long_batched (the batched form of the reference's
                    # one-window-at-a-time loop, tools/m2d_test.py:139-232) on the registry-built MotionDiffusion + ControlT2MHalf -- 32 sequences
                    # of 480 frames = 5 windows of 120 advancing by 90 each, repaint off: ONE model call of 160 windows, stitched per sequence
                    import motioncraft_amd as mc
                    from motioncraft_amd import longform
                    from motioncraft_amd.synthetic import reference_model_cfg
                    sched = dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large')
                    cfgm = mc.Config(dict(model=dict(type='MotionDiffusion', model=reference_model_cfg(dict(dm, max_seq_len=max(Tc, dm['max_seq_len']))),
                                                     loss_recon=dict(type='MSELoss', loss_weight=1, reduction='none'), diffusion_train=sched,
                                                     diffusion_test=dict(sched, respace='15,15,8,6,6'), inference_type='ddim'),
                                          condition_encode_cfg=dict(condition_cfg=True)))
                    arch = mc.build_architecture(cfgm.model)
                    arch.model = mc.ControlT2MHalf(arch.model, copy_blocks_num=copy, control_cond_feats=feats, cfg=cfgm)
                    arch.load_state_dict({'model.' + k: v for k, v in make_state_dict(dm, 0, shapes=control_param_shapes(dm, copy, feats)).items()})
                    S_, tot_, pre_ = Bc // 5, 480, 30
                    cseq = torch.randn(S_, tot_, feats, generator=torch.Generator().manual_seed(5))
                    xfs = torch.nn.functional.layer_norm(torch.randn(S_, dm['Nt'], dm['Dt'], device=dev, generator=gen), (dm['Dt'],))
                    tdr = []
                    for rep in range(2):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        recs, wins = longform.sample_long_batched(arch, tot_, Tc, pre_, c=cseq, text=[''] * S_, repaint=False, condition_kwargs=dict(xf_out=xfs),
                                                                  max_batch=Bc, shard=False, device=dev)
                        torch.cuda.synchronize()
                        tdr.append(time.perf_counter() - t0)
                    assert len(recs) == S_ and recs[0].shape == (4 * 90 + 120, C) and len(wins) == Bc and all(bool((r == r).all()) for r in recs)
                    control_cfgs[key]['through_sample_long_batched'] = {
                        'sequences': S_, 'frames_per_sequence': tot_, 'windows': Bc, 'model_calls': 1, 'loop_ms': round(min(tdr) * 1e3, 1),
                        'note': 'motioncraft_amd.longform.sample_long_batched on the registry-built architecture: condition set-up + the 50-step loop of '
                                '160 windows in one model call + download and stitching of the 32 sequences (host side included)'}
                    arch.model.release()
            control_cfgs['note'] = ('side measurements, not `value`: per-GPU share of BASELINE configs[2] / configs[3], complete 50-step DDIM loop (best of 2 after a '
                                    'warm-up loop), condition features resident in HBM')
        except Exception as e:      # a side measurement must not take the headline line down with it: recorded in the line, not hidden
            control_cfgs = {'error': f'{type(e).__name__}: {e}'}

    t = torch.tensor([t_loop, t_setup, t_gather, ev_ms, t_full or 0.0], device=dev if a.backend == 'nccl' else 'cpu', dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_loop, t_setup, t_gather, ev_ms, t_full_max = (float(v) for v in t)
    t_step = t_loop / a.steps
    value = GB * T / (t_setup + TOTAL_DDPM_STEPS * t_step + t_gather)

    if rank == 0:
        flops_step = algorithmic_flops_per_sample_step(DIMS, T) * B
        traffic_gb, traffic_src = measured_hbm_traffic()
        full_loop = None
        if t_full is not None:
            v_meas = GB * T / (t_setup + t_full_max + t_gather)
            full_loop = {'loop_s': round(t_full_max, 3), 'value_measured': round(v_meas, 2), 'value_extrapolated': round(value, 2),
                         'measured_over_extrapolated': round(v_meas / value, 4),
                         'note': 'complete 1000-step loop run once after the timed region (x_T -> x_0, max over ranks); '
                                 '`value` uses the K timed steps as the contract asks'}
        ach = flops_step / (ev_ms * 1e-3) / 1e12
        D_ = DIMS['L'] * DIMS['H']
        alg_gb = (127.9e6 * 4 + DIMS['NL'] * 6 * (2 * B * T * D_ * 4) + 5 * B * T * C * 4) / 1e9
        line = {
            'metric': 'sampled SMPL-X frames/sec (196-frame seq, 1000-step DDPM)',
            'value': round(value, 2), 'unit': 'frames/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(t_step * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload_name(B, T),
                       'batch_per_gpu': B, 'global_batch': GB, 'frames': T, 'parallelism': f'dp{world}',
                       'weights': 'random-init (name-keyed deterministic), no checkpoint offline',
                       'setup_s': round(t_setup, 4), 'gather_s': round(t_gather, 4),
                       'process_group': (f'{a.backend} (RCCL), world size {world}' + (' -- forced at one rank by MC_BENCH_FORCE_DIST=1' if world == 1 else '')
                                         if a.backend == 'nccl' else f'{a.backend}, world size {world}') if use_dist else None,
                       'frames_per_s_formula': 'N*B*T / (setup_s + 1000*ms_per_step/1e3 + gather_s)',
                       'full_loop': full_loop, 'reduced_precision_modes': reduced, 'configs0_gpu': configs0, 'configs4': configs4, 'control_configs': control_cfgs,
                       'batches_in_flight': 1, 'two_batches_in_flight': inflight2,
                       'exact_reductions': 'results equal the unreduced computation (tests/test_gpu_parity.py): CFG twins of base layer 0 '
                                           'share gate / expert / proj / qkv / body work (identical inputs); the last StylizationBlock '
                                           'Linear + affine pose decoder run once on the CFG-combined rows; FLOPs in `roofline` are '
                                           'counted as the reference performs them (DESIGN.md section 4)'},
            'roofline': {'bound': 'mfma', 'achieved': round(ach, 3), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                         'executed_frac': round(executed_flops_per_sample_step(DIMS, T) * B / (ev_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                         'executed_gflop_per_sample_step': round(executed_flops_per_sample_step(DIMS, T) / 1e9, 3),
                         'frac_note': '`frac` counts FLOPs as the reference performs them (SURVEY.md section 8d); `executed_frac` counts what the kernels '
                                      'launch after the exact reductions (CFG twin dedupe of base layer 0, folded decoder tail)',
                         'traffic': traffic_gb if (B, T) == (64, 196) else None,
                         'algorithmic_bytes': round(alg_gb, 3),
                         'algorithmic_bytes_unit': 'GB per step (SURVEY.md section 8d): fp32 weights streamed once (0.512) + the residual stream h '
                                                   '[2B,T,D] once per fused kernel, 6 per layer + the sampler update 5 B T 322 floats',
                         'traffic_over_algorithmic': round(traffic_gb / alg_gb, 2) if (traffic_gb and (B, T) == (64, 196)) else None,
                         'traffic_unit': f'GB per step, read from {traffic_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 gfx950 correction)',
                         'kernel': 'one denoising step = all kernels of one mc_sample_loop step (dominant: gemm_wp_k fp32 MFMA GEMMs)',
                         'algorithmic_gflop_per_sample_step': round(algorithmic_flops_per_sample_step(DIMS, T) / 1e9, 3),
                         'event_ms_per_step': round(ev_ms, 4), 'dominant_kernel': dom},
        }
        if world == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
