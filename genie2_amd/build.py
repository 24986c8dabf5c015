"""Builds libgenie_hip.so (gfx950) in-tree with hipcc.

    python -m genie2_amd.build [--force]

Objects and the .so land in genie2_amd/lib/ (git-ignored; they travel to the
GPU box with the gpurun snapshot).  Cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libgenie_hip.so')
SOURCES = ['pair_kernels.hip', 'pair_wl_kernels.hip', 'pair_hx_kernels.hip', 'pair_fused_kernels.hip', 'single_kernels.hip', 'train_kernels.hip', 'train_layout_kernels.hip', 'probe_kernels.hip', 'genie_train.hip', 'genie_api.hip']
HEADERS = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'hx.h'), os.path.join(CSRC, 'hx_pair.h'), os.path.join(CSRC, 'hx_fused.h'), os.path.join(CSRC, 'train.h'), os.path.join(HERE, '..', 'include', 'genie_hip.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function', '-Wno-unused-value',
         '-DGENIE_BUILD'] + os.environ.get('GENIE_EXTRA_FLAGS', '').split()


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _compile(src):
    obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + '.o')
    path = os.path.join(CSRC, src)
    if _newer(obj, [path] + HEADERS):
        return obj
    cmd = [HIPCC] + FLAGS + ['-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(LIBDIR):
            if os.path.isfile(os.path.join(LIBDIR, f)):      # (lib/abl/ holds developer variant builds)
                os.remove(os.path.join(LIBDIR, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if not _newer(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
