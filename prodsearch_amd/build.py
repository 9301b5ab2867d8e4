"""Build libprodsearch_hip.so for gfx950 with hipcc (in-tree, no torch headers).

``python -m prodsearch_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU, so this also runs in the CPU container.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libprodsearch_hip.so')
DIAG_LIB = os.path.join(LIBDIR, 'libprodsearch_hip_diag.so')      # -DPS_DIAG: tuning knobs, stamps, timing-only variants (tools/)
SOURCES = ['gemm.hip', 'rowwise.hip', 'attn_sq1.hip', 'mlp_fused.hip', 'optim.hip', 'optim_rows.hip', 'rank.hip', 'tem.hip', 'rtm.hip']
HEADERS = ['common.h', 'rowwise.h', 'encoder.h', 'optim_core.h', 'graph.h', 'x3frag.h', os.path.join('..', '..', 'include', 'prodsearch_hip.h')]


DATA_LIB = os.path.join(LIBDIR, 'libprodsearch_data.so')
DATA_SOURCES = ['collate.cpp']
DATA_HEADERS = [os.path.join('..', '..', 'include', 'prodsearch_data.h')]


def build_data(force=False, verbose=False):
    """Host-only batch builder (include/prodsearch_data.h): plain g++, no GPU runtime, loads anywhere."""
    deps = [os.path.join(CSRC, s) for s in DATA_SOURCES + DATA_HEADERS]
    if not force and os.path.exists(DATA_LIB) and all(os.path.getmtime(p) <= os.path.getmtime(DATA_LIB) for p in deps):
        return DATA_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-fPIC', '-shared', '-pthread', '-Wall', '-o', DATA_LIB] + \
          [os.path.join(CSRC, s) for s in DATA_SOURCES]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed:\n" + res.stdout)
    return DATA_LIB


OBJDIR = os.path.join(HERE, '_obj')


def _newer(path, deps):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(p) > t for p in deps)


def _stale(lib):
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return _newer(lib, deps)


def build(force=False, verbose=False, diag=False):
    """Compile every HIP source for gfx950 — one object per source, in parallel, only the stale ones — and link
    them into one shared library (plus the host batch builder); returns its path.  diag=True builds the diagnostic
    library beside it (same sources, -DPS_DIAG; see csrc/common.h, ps_diag_int) — never loaded unless PS_DIAG_LIB=1."""
    build_data(force, verbose)
    LIB = DIAG_LIB if diag else globals()['LIB']
    OBJDIR = globals()['OBJDIR'] + ('_diag' if diag else '')
    if not force and not _stale(LIB):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        hipcc = 'hipcc'
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.replace('.hip', '.o'))
        objs.append(obj)
        if force or _newer(obj, [os.path.join(CSRC, src)] + hdrs):
            cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + (['-DPS_DIAG'] if diag else []) + \
                  ['-c', '-o', obj, os.path.join(CSRC, src)]
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    errs = []
    for src, pr in jobs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            errs.append("hipcc failed on %s:\n%s" % (src, out))
    if errs:
        raise RuntimeError('\n'.join(errs))
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, diag='--diag' in sys.argv))
