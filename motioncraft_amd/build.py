"""Build libmotioncraft_amd.so in-tree with hipcc for gfx950 (no torch dependency in the library)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmotioncraft_amd.so')
SOURCES = ['mc_error.cpp', 'mc_gemm.hip', 'mc_kernels.hip', 'mc_route.hip', 'mc_attn.hip', 'mc_chain.hip', 'mc_half.hip', 'mc_post.hip', 'mc_wavenc.hip', 'mc_textenc.hip', 'mc_evalenc.hip', 'mc_t2meval.hip', 'mc_model.hip']


def _hipcc():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, '..', 'include', 'motioncraft_amd.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src.rsplit('.', 1)[0] + '.o')
        objs.append(obj)
        sp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                os.path.getmtime(sp), *(os.path.getmtime(os.path.join(CSRC, h)) for h in os.listdir(CSRC) if h.endswith('.h')),
                os.path.getmtime(os.path.join(HERE, '..', 'include', 'motioncraft_amd.h'))):
            continue
        cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-c', sp, '-o', obj]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed for {src}:\n{out.decode()}')
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
