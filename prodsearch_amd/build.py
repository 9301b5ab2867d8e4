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
SOURCES = ['gemm.hip', 'rowwise.hip', 'attn_sq1.hip', 'mlp_fused.hip', 'optim.hip', 'optim_rows.hip', 'rank.hip', 'tem.hip', 'rtm.hip']
HEADERS = ['common.h', 'rowwise.h', 'encoder.h', 'optim_core.h', 'graph.h', os.path.join('..', '..', 'include', 'prodsearch_hip.h')]


DATA_LIB = os.path.join(LIBDIR, 'libprodsearch_data.so')
DATA_SOURCES = ['collate.cpp']
DATA_HEADERS = [os.path.join('..', '..', 'include', 'prodsearch_data.h')]


def build_data(force=False, verbose=False):
    """Host-only batch builder (include/prodsearch_data.h): plain g++, no GPU runtime, loads anywhere."""
    deps = [os.path.join(CSRC, s) for s in DATA_SOURCES + DATA_HEADERS]
    if not force and os.path.exists(DATA_LIB) and all(os.path.getmtime(p) <= os.path.getmtime(DATA_LIB) for p in deps):
        return DATA_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-o', DATA_LIB] + \
          [os.path.join(CSRC, s) for s in DATA_SOURCES]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ failed:\n" + res.stdout)
    return DATA_LIB


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    """Compile every HIP source into one shared library (and the host batch builder); returns its path."""
    build_data(force, verbose)
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        hipcc = 'hipcc'
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
