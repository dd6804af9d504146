"""Build the HIP libraries in-tree (gfx950 only).

    python -m sda_amd.build            # libsda_hip.so  (the product: device kernels + C ABI)
    python -m sda_amd.build --emu      # libsda_emu.so  (tests only: host replay of the conv tile algorithm)

The objects are compiled with hipcc and linked WITHOUT an rpath to /opt/rocm: at run time the library must bind
to the HIP runtime that PyTorch-ROCm already loaded (same soname libamdhip64.so.7), so that torch's streams and
device pointers are valid inside these kernels.  `import torch` therefore always precedes loading (sda_amd/_lib.py).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.environ.get('SDA_LIBDIR') or os.path.join(HERE, 'lib')      # (SDA_LIBDIR: tooling builds beside the product one)
ARCH = 'gfx950'
SOURCES = ['conv_igemm.hip', 'conv_wino.hip', 'conv_wino4.hip', 'conv_small1d.hip', 'conv_few.hip', 'conv_par4.hip', 'conv_h2.hip', 'conv3d.hip', 'block1d.hip', 'net1d.hip', 'step1d.hip', 'norm.hip', 'elementwise.hip', 'linear.hip', 'mlp1d.hip', 'observe.hip', 'metrics.hip', 'noise.hip', 'probe.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# extra compile flags (e.g. SDA_EXTRA_HIPCC_FLAGS=-DSDA_W4_VARIANTS builds the tuning variants tools/wino4_check.py compares)
EXTRA_FLAGS = os.environ.get('SDA_EXTRA_HIPCC_FLAGS', '').split()
CONV_PARTS = 4                                       # see SDA_CONV_PART in csrc/conv_igemm.hip


def _digest(deps, flags=()):
    h = hashlib.sha256()
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(flags).encode())
    return h.hexdigest()


def _newer(target, deps, flags=()):
    """Is `target` out of date?  By CONTENT (a digest of the inputs kept beside the target), not by mtime: file times are not
    trustworthy everywhere this runs (snapshot copies, containers with a coarse or non-monotonic clock)."""
    stamp = target + '.stamp'
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != _digest(deps, flags)


def _stamp(target, deps, flags=()):
    with open(target + '.stamp', 'w') as f:
        f.write(_digest(deps, flags))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp')]
    hs.append(os.path.join(os.path.dirname(HERE), 'include', 'sda_hip.h'))
    return hs


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    hdrs = _headers()
    procs = []
    units = []                                       # (source, object suffix, extra flags)
    for src in SOURCES:
        if src == 'conv_igemm.hip':                  # its kernel families compile as separate, parallel units
            units += [(src, f'_p{k}', [f'-DSDA_CONV_PART={k}']) for k in range(CONV_PARTS)]
        else:
            units.append((src, '', []))
    for src, suffix, flags in units:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(LIBDIR, src.replace('.hip', suffix + '.o'))
        objs.append(obj)
        cflags = ['-O3', '-std=c++17', '-fPIC', '-ffp-contract=off'] + flags + EXTRA_FLAGS
        if force or _newer(obj, [sp] + hdrs, cflags):
            cmd = [HIPCC, f'--offload-arch={ARCH}'] + cflags + ['-c', sp, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            procs.append((src + suffix, obj, [sp] + hdrs, cflags, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, obj, deps, cflags, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out.decode()}')
        _stamp(obj, deps, cflags)
    lib = os.path.join(LIBDIR, 'libsda_hip.so')
    if force or procs or _newer(lib, objs):
        cmd = [HIPCC, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', lib] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        _stamp(lib, objs)
    return lib


def build_emu(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(CSRC, 'conv_igemm.hip')
    lib = os.path.join(LIBDIR, 'libsda_emu.so')
    if force or _newer(lib, [src] + _headers()):
        subprocess.check_call([HIPCC, '--offload-host-only', '-DSDA_HOST_EMU', '-O2', '-std=c++17', '-fPIC',
                               '-ffp-contract=off', '-shared', src, '-o', lib])
        _stamp(lib, [src] + _headers())
    return lib


if __name__ == '__main__':
    force = '--force' in sys.argv
    if '--emu' in sys.argv:
        print(build_emu(force))
    else:
        print(build(force, verbose=True))
