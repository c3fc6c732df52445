"""Build libhvx.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python -m flowmirror_hydravox_amd.build [--force] [-j N]
    python -m flowmirror_hydravox_amd.build --lab NAME -- -DHVX_LAB [-DHVX_LAB_GEMM_EPI=1 ...]      # a LAB library, never the product's

hipcc cross-compiles without a GPU.  Objects are cached under csrc/_build/ by source mtime AND by a stamp of the exact compiler command: objects built with
other flags are rebuilt, never reused.  A lab build (extra flags) goes to its own directory and its own file, csrc/_build/lab_NAME/libhvx_lab_NAME.so, and is
loaded only through HVX_LIB_PATH (flowmirror_hydravox_amd._lib refuses a -DHVX_LAB library found at the product's path); it never touches libhvx.so.
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


def _command(src, obj, extra):
    flags_text = ' '.join(extra)
    define = ['-DHVX_BUILD_FLAGS="%s"' % flags_text.replace('"', "'")] if extra else []
    return [_hipcc()] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + list(extra) + define + ['-c', src, '-o', obj]


def _compile(src, obj, verbose, extra=()):
    cmd = _command(src, obj, extra)
    t0 = time.time()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        sys.stderr.write('[hvx build] %s  (%.1fs)\n%s' % (os.path.basename(src), time.time() - t0, r.stdout))
    if r.returncode != 0:
        raise RuntimeError('hipcc failed on %s' % src)
    with open(obj + '.cmd', 'w') as f:                 # the stamp: what this object was built with
        f.write(' '.join(cmd))
    return obj


def _stale(src, obj, hm, extra):
    if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm):
        return True
    try:
        with open(obj + '.cmd') as f:
            return f.read() != ' '.join(_command(src, obj, extra))
    except OSError:
        return True                                    # an object without a stamp (an older build, a lab build of unknown flags) is never trusted


def build(force=False, jobs=None, verbose=False, lab=None, extra=()):
    """lab = NAME + extra flags: a separate library under csrc/_build/lab_NAME/ (returned path -> HVX_LIB_PATH); the product build takes no extra flags"""
    extra = tuple(extra)
    if extra and not lab:
        raise ValueError('extra compiler flags make a LAB library: pass lab=NAME (the in-tree libhvx.so is only ever built with the product flags)')
    if os.environ.get('HVX_EXTRA_FLAGS'):
        raise RuntimeError('HVX_EXTRA_FLAGS is gone (it rebuilt the product library in place): use `python -m flowmirror_hydravox_amd.build --lab NAME -- FLAGS`')
    bdir = os.path.join(CSRC, '_build') if not lab else os.path.join(CSRC, '_build', 'lab_' + lab)
    out = OUT if not lab else os.path.join(bdir, 'libhvx_lab_%s.so' % lab)
    os.makedirs(bdir, exist_ok=True)
    hm = _headers_mtime()
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _stale(src, obj, hm, extra):
            todo.append((src, obj))
    if todo:
        jobs = jobs or min(6, os.cpu_count() or 2)
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(lambda so: _compile(so[0], so[1], verbose, extra), todo))
    if todo or not os.path.exists(out):
        cmd = [_hipcc(), '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', out] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout)
            raise RuntimeError('link failed')
    return out


if __name__ == '__main__':
    argv = sys.argv[1:]
    extra = []
    if '--' in argv:
        extra = argv[argv.index('--') + 1:]
        argv = argv[:argv.index('--')]
    force = '--force' in argv
    jobs = int(argv[argv.index('-j') + 1]) if '-j' in argv else None
    lab = argv[argv.index('--lab') + 1] if '--lab' in argv else None
    t0 = time.time()
    path = build(force=force, jobs=jobs, verbose=True, lab=lab, extra=extra)
    print(path, '(%.1fs)' % (time.time() - t0))
    if lab:
        print('lab library: export HVX_LIB_PATH=%s' % path)
