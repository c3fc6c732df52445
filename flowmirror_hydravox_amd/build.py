"""Build libhvx.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python -m flowmirror_hydravox_amd.build [--force] [-j N]

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under csrc/_build/.
"""
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
OUT = os.path.join(HERE, 'libhvx.so')
ARCH = 'gfx950'
# No packed fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere in libhvx.  Measured on MI355X (tools/mfma_interference.py,
# docs/history/DESIGN_rounds1-4.md §8): a wave executing v_pk_*_f32 returns wrong low halves in lanes 48-63 now and then while ANOTHER wave on its SIMD streams bf16
# MFMAs — e.g. hift_stft_kernel beside the split-bf16 vocoder convolutions of a second stream: 1941 of 3000 launches differ with the packed
# forms, 0 of 3000 without.  hipcc's SLP vectoriser forms them from any two adjacent fp32 operations, so the feature is switched off for the
# device pass (the host pass does not know the feature and says so; harmless).  Beside MFMAs the packed forms are slower than two scalar ops
# anyway (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
NO_PACKED_F32 = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=' + ARCH, '-Wall', '-Wno-unused-function', '-Wno-unused-variable', '-Wno-pass-failed',
         '-I', CSRC, '-I', INCLUDE] + NO_PACKED_F32


# per-file additions: the attention softmax takes maxima of values that are never NaN (scores, -inf masks)
FILE_FLAGS = {'attention.hip': ['-fno-honor-nans']}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers_mtime():
    m = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith('.h'):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, obj, verbose):
    # HVX_EXTRA_FLAGS: lab builds only (e.g. -DHVX_ATTN_LAB adds timing-only variants of the DiT attention loop); never set for the shipped library
    cmd = [_hipcc()] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + os.environ.get('HVX_EXTRA_FLAGS', '').split() + ['-c', src, '-o', obj]
    t0 = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        sys.stderr.write('[hvx build] %s  (%.1fs)\n%s' % (os.path.basename(src), time.time() - t0, r.stdout))
    if r.returncode != 0:
        raise RuntimeError('hipcc failed on %s' % src)
    return obj


def build(force=False, jobs=None, verbose=False):
    bdir = os.path.join(CSRC, '_build')
    os.makedirs(bdir, exist_ok=True)
    hm = _headers_mtime()
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm):
            todo.append((src, obj))
    if todo:
        jobs = jobs or min(6, os.cpu_count() or 2)
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(lambda so: _compile(so[0], so[1], verbose), todo))
    if todo or not os.path.exists(OUT):
        cmd = [_hipcc(), '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', OUT] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout)
            raise RuntimeError('link failed')
    return OUT


if __name__ == '__main__':
    force = '--force' in sys.argv
    jobs = None
    if '-j' in sys.argv:
        jobs = int(sys.argv[sys.argv.index('-j') + 1])
    t0 = time.time()
    print(build(force=force, jobs=jobs, verbose=True), '(%.1fs)' % (time.time() - t0))
