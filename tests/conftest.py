import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def _usable_cores() -> int:
    """cores this process may really use: affinity mask capped by the cgroup CPU quota (inside a container os.cpu_count() reports
    the host's cores and torch sizes its OpenMP pool by it -- the CPU oracle then runs oversubscribed and several times slower)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    import torch
    torch.set_num_threads(_usable_cores())


# `pytest -m gpu -x` (the driver's round-end run) stops at the first failure: the oracle-parity files of the hot path run FIRST, the
# bf16 self-consistency file after them and the multi-process / infrastructure file LAST, so that a stop in infrastructure can never
# hide a parity result.  Files not named keep their alphabetical place between the two groups.
_GPU_ORDER = ['test_gpu_kernels', 'test_gpu_model', 'test_gpu_ref_golden', 'test_gpu_full_size', 'test_gpu_gradients',
              'test_gpu_postproc', 'test_gpu_eval']
_GPU_LAST = ['test_gpu_bf16', 'test_gpu_dist']


def _file_rank(item) -> int:
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _GPU_ORDER:
        return _GPU_ORDER.index(name)
    if name in _GPU_LAST:
        return 1000 + _GPU_LAST.index(name)
    return 500


def pytest_collection_modifyitems(config, items):
    items.sort(key=_file_rank)                 # stable: order inside a file (and among unnamed files) is kept
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)
