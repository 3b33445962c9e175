"""Which Python lines launch the small torch kernels of a training step (298 torch launches / step).

Runs the cfg2 step under torch.profiler with Python stacks and prints, per aten operator that reaches the GPU, the call sites inside
unidet3d_amd/ (forward ops) or the autograd node that issued it (backward ops), with launch counts per step.
usage: python tools/launch_sites.py [scenes] [points]      |  python tools/launch_sites.py cfg4   (the joint six-dataset batch of bench.py --config cfg4)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cfg4 = len(sys.argv) > 1 and sys.argv[1] == 'cfg4'
    B = int(sys.argv[1]) if len(sys.argv) > 1 and not cfg4 else 8
    pts = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd.config import build_model, scannet_model_cfg
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.dist import FlatGradBucket
    from unidet3d_amd.synthetic import make_scene
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    if cfg4:
        from bench import CFG4_SCENES
        from unidet3d_amd.config import joint_model_cfg
        from unidet3d_amd.data import make_joint_batch
        model = build_model(joint_model_cfg()).to(dev).train()
    else:
        model = build_model(scannet_model_cfg(voxel_size=0.02)).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    bucket = FlatGradBucket(params, attach=False)
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.05, fused=True)
    if cfg4:
        _, _, _, inputs, samples = make_joint_batch(joint_model_cfg(), [(n, p, p / 100_000) for n, p in CFG4_SCENES], dev, seed0=200)
    else:
        inputs, samples = make_batch_inputs([make_scene(i, n_points=pts) for i in range(B)], dev)

    def step():
        bucket.clear_grads()
        loss = model.loss(inputs, samples)['det_loss']
        loss.backward()
        bucket.sync()
        bucket.clip_grad_norm_(10.0)
        opt.step()
        model.prefetch(inputs, samples)
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    LAUNCHING = ('copy_', 'fill_', 'zero_', 'add', 'add_', 'sub', 'sub_', 'mul', 'mul_', 'div', 'div_', 'cat', 'where', 'ge', 'gt', 'lt', 'le', 'eq', 'ne',
                 'index', 'index_put_', 'index_select', '_to_copy', 'clone', 'zeros', 'ones', 'full', 'zeros_like', 'ones_like', 'full_like', 'sum', 'cumsum',
                 'constant_pad_nd', 'neg', 'rsqrt', 'sqrt', 'exp', 'log', 'clamp', 'clamp_min', 'maximum', 'minimum', 'arange', 'remainder', 'reciprocal',
                 'linalg_vector_norm', '_foreach_copy_', '_foreach_add_', '_fused_adamw_', 'sigmoid', 'softmax', '_softmax', 'topk', 'sort', 'nonzero', 'any', 'all',
                 'max', 'min', 'mean', 'bitwise_and', 'bitwise_or', 'logical_and', 'logical_not', 'masked_fill_', 'masked_fill', 'scatter_', 'gather', 'stack',
                 'slice_backward', 'select_backward', 'embedding', 'bmm', 'mm', 'addmm', 'matmul', 'contiguous', 'abs', 'pow', 'item', '_local_scalar_dense')
    sites = collections.Counter()
    shapes = {}

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func.__name__.split('.')[0]
            if name in LAUNCHING:
                cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or \
                    any(isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor) and a[0].is_cuda for a in args) or \
                    str((kwargs or {}).get('device', '')).startswith('cuda')
                if cuda:
                    fr = [f for f in traceback.extract_stack() if ('unidet3d_amd' in f.filename or 'tools/' in f.filename) and 'launch_sites' not in f.name]
                    where = ' <- '.join(f'{os.path.relpath(f.filename, ROOT)}:{f.lineno}' for f in fr[-2:][::-1]) if fr else 'native autograd node / torch internals'
                    key = (name, where)
                    sites[key] += 1
                    shapes.setdefault(key, str([tuple(a.shape) for a in args if isinstance(a, torch.Tensor)])[:80])
            return func(*args, **(kwargs or {}))

    with Mode():
        step()
        torch.cuda.synchronize()
    per_op = collections.Counter()
    for (name, _), n in sites.items():
        per_op[name] += n
    print('aten ops on CUDA tensors in one step (views / allocations excluded):', sum(per_op.values()))
    for name, n in per_op.most_common():
        print(f'  {n:4d}  {name}')
    print()
    for (name, where), n in sites.most_common(120):
        print(f'{n:4d}  {name:22s} {where[:120]}   {shapes[(name, where)]}')


if __name__ == '__main__':
    main()
