"""Builds gennet_amd/lib/libgennet_hip.so from gennet_amd/csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU).

The library links against the HIP runtime only (no torch): python reaches it through ctypes (gennet_amd/_lib.py).
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libgennet_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function']
# development only (timing-ablation builds, -DGN_ABLATION: kernels that skip work and return wrong results); never set for the shipped library
FLAGS += os.environ.get('GENNET_HIPCC_EXTRA', '').split()


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Incremental by default (a source or header newer than its object is recompiled); force=True compiles every translation unit and
    relinks -- the "does it build from scratch" check (__graft_entry__.build() forces it).  Returns the library path; what was done is
    recorded in lib/build_info.json: {"mode": "clean" | "incremental", "compiled": [...], "seconds": ...}."""
    import json
    import time
    t0 = time.time()
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, 'obj')
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdrs = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(HERE, '..', 'include', 'gennet_hip.h')]
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return r.returncode, r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for rc, out in ex.map(run, jobs):
                if out.strip() and verbose:
                    print(out)
                if rc:
                    raise RuntimeError('hipcc failed:\n' + out)
    if jobs or force or _stale(LIB, objs):
        rc, out = run([HIPCC, '-shared', '-fPIC', '--offload-arch=gfx950', '-o', LIB] + objs)
        if rc:
            raise RuntimeError('link failed:\n' + out)
    info = {'mode': 'clean' if force else 'incremental', 'compiled': [os.path.basename(j[-3]) for j in jobs], 'sources': len(srcs),
            'seconds': round(time.time() - t0, 1), 'hipcc': HIPCC, 'flags': FLAGS}
    with open(os.path.join(LIBDIR, 'build_info.json'), 'w') as f:
        json.dump(info, f)
    if verbose:
        print('build_mode=%s compiled=%d/%d translation units in %.1f s' % (info['mode'], len(jobs), len(srcs), info['seconds']), flush=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
