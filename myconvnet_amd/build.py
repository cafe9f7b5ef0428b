"""Build recipe for libmcn_hip.so (gfx950 only).  hipcc cross-compiles without a GPU.

    python myconvnet_amd/build.py [--force]     (run as a script: importing the package needs the built library)

Objects are cached by mtime under myconvnet_amd/csrc/_obj; the shared library is written in-tree
(myconvnet_amd/libmcn_hip.so) so that it travels with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libmcn_hip.so')
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))      # every translation unit under csrc/ (globbed: see _headers)


def _headers():
    """Every header a translation unit can see: all of csrc/*.h (globbed, so a new kernel header is part of the build identity the
    moment it exists) plus the C-ABI header.  tests/test_host_logic.py checks that every file under csrc/ is covered and that every
    `#include "..."` of the sources resolves into this list."""
    hs = sorted(f for f in os.listdir(CSRC) if f.endswith('.h'))
    return hs + [os.path.join('..', '..', 'include', 'mcn.h')]


HEADERS = _headers()
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function',
         '-Wno-unused-variable', '-ffp-contract=fast']
# MCN_KERNEL_PROBES=1 (set for the build AND for the run): compile the timing probes into the kernels (MCN_NT_EPI_FLAGS / MCN_TN_DBG skip a kind of
# instruction — wrong results, timing only).  Off by default: as run-time branches they cost the K loops 0.6-0.8 % of the step.  Own object directory
# and a different source digest, so that the two builds never mix.
if os.environ.get('MCN_EXTRA_CFLAGS'):                   # experiment builds (e.g. -DMCN_NT_STAGE_ST=2; set for the build AND the run): own objects, own build identity
    import hashlib as _h
    FLAGS += os.environ['MCN_EXTRA_CFLAGS'].split()
    OBJ = os.path.join(CSRC, '_obj_x' + _h.sha1(os.environ['MCN_EXTRA_CFLAGS'].encode()).hexdigest()[:8])
PROBES = os.environ.get('MCN_KERNEL_PROBES') == '1'
if PROBES:
    FLAGS.append('-DMCN_KERNEL_PROBES=1')
    OBJ = os.path.join(CSRC, '_obj_probes')


def source_digest():
    """sha256 over the kernel sources and headers, in a fixed order: the identity of what libmcn_hip.so must be built from."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read() + b'\0')
    if PROBES:
        h.update(b'MCN_KERNEL_PROBES=1')
    if os.environ.get('MCN_EXTRA_CFLAGS'):
        h.update(os.environ['MCN_EXTRA_CFLAGS'].encode())
    return h.hexdigest()


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = _newest([os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)])
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([HIPCC] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return cmd, r.returncode, r.stdout

    # the build id travels in a generated one-line translation unit, recompiled whenever the digest changes
    digest = source_digest()
    id_src, id_obj = os.path.join(OBJ, 'build_id.cpp'), os.path.join(OBJ, 'build_id.o')
    line = 'extern "C" const char* mcn_build_id(void) { return "%s"; }\n' % digest
    if force or not os.path.exists(id_obj) or not os.path.exists(id_src) or open(id_src).read() != line:
        with open(id_src, 'w') as fh:
            fh.write(line)
        jobs.append([os.environ.get('CXX', 'g++'), '-O1', '-fPIC', '-c', id_src, '-o', id_obj])
    objs.append(id_obj)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for cmd, rc, out in ex.map(run, jobs):
                if verbose and out.strip():
                    print(out)
                if rc != 0:
                    raise RuntimeError('hipcc failed: {}\n{}'.format(' '.join(cmd), out))
    if jobs or force or not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest(objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        cmd, rc, out = run(cmd)
        if rc != 0:
            raise RuntimeError('link failed: {}\n{}'.format(' '.join(cmd), out))
    with open(LIB + '.id', 'w') as fh:                   # what _ffi.py compares with the tree before loading the library
        fh.write(digest + '\n')
    return LIB


CPU_SRC = os.path.join(HERE, 'csrc_cpu', 'mcn_cpu.cpp')
CPU_LIB = os.path.join(HERE, 'libmcn_cpu.so')


def build_cpu(force=False):
    """libmcn_cpu.so: the same C-ABI as plain C++ / OpenMP loops (csrc_cpu/mcn_cpu.cpp).  Test infrastructure for the host code — the
    binding loads it only when MCN_LIB_PATH names it (never as a fallback)."""
    deps = [CPU_SRC, os.path.join(CSRC, '..', '..', 'include', 'mcn.h')]
    if force or not os.path.exists(CPU_LIB) or os.path.getmtime(CPU_LIB) < _newest(deps):
        cmd = [os.environ.get('CXX', 'g++'), '-O2', '-fopenmp', '-fPIC', '-shared', '-std=c++17', '-Wall', '-Wno-unused-variable', '-o', CPU_LIB, CPU_SRC]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0 and '-fopenmp' in cmd:          # no libgomp: the loops are still correct single-threaded
            cmd.remove('-fopenmp')
            cmd.insert(1, '-Wno-unknown-pragmas')
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('g++ failed: {}\n{}'.format(' '.join(cmd), r.stdout))
    return CPU_LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    print(build_cpu(force='--force' in sys.argv))
