import os
import sys


def _host_cores():
    """Cores this process may use: min(affinity mask, cgroup CPU quota).  The GPU box shows 256 logical CPUs but grants a 16-CPU quota:
    a BLAS pool of 256 threads there runs the numpy oracle ~10x slower than 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (IOError, OSError, ValueError):
        pass
    return n


for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, str(_host_cores()))

# the opt-in split math is gated by launch size (capi.hip: split_worth_it, 50 GFLOP): the parity tests run its kernels on small shapes on purpose
os.environ.setdefault('GN_BF16X3_MIN_GFLOP', '0')

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    try:                                   # numpy may have been imported (with its own default pool size) before this file ran
        import threadpoolctl
        session.config._gn_tp = threadpoolctl.threadpool_limits(limits=_host_cores())
    except Exception:
        pass
