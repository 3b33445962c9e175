"""Build libu3d_hip.so (gfx950) in-tree with hipcc.  `python -m unidet3d_amd.csrc.build`."""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ['misc.hip', 'voxelize.hip', 'rulebook.hip', 'spconv.hip', 'spconv_wg.hip', 'spconv_ts.hip', 'spconv_wgrad_rows.hip', 'bn.hip', 'pool.hip', 'attn.hip', 'attn_x3.hip', 'gemm.hip', 'gemm_b16.hip', 'norm.hip', 'postproc.hip', 'criterion.hip', 'hashidx.hip', 'radix.hip']
LIB = os.path.join(HERE, 'libu3d_hip.so')
# (no -munsafe-fp-atomics: since round 5 no kernel of the library issues a floating-point atomic -- every reduction has a fixed order)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value',
         '-I', os.path.join(ROOT, 'include'), '-I', HERE]


# per-file code generation options
EXTRA = {'spconv.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'],
         'spconv_wg.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'],     # MFMA results are consumed by VALU/LDS right away
         'attn_x3.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'],
         'postproc.hip': ['-ffp-contract=off']}                     # bit-exact against the oracle's operation order


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _stale(src, obj):
    if not os.path.exists(obj):
        return True
    deps = [src, os.path.join(HERE, 'u3d_common.h'), os.path.join(HERE, 'spconv_gmm.h'), os.path.join(ROOT, 'include', 'u3d.h'), __file__]
    return any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(src, obj):
            jobs.append([hipcc, *FLAGS, *EXTRA.get(s, []), '-c', src, '-o', obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if verbose and (res.stdout or res.stderr):
                    sys.stderr.write(res.stdout + res.stderr)
                if res.returncode != 0:
                    raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    if jobs or not os.path.exists(LIB):
        res = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs],
                             capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('link failed: ' + res.stdout + res.stderr)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
