"""Find the backward op(s) that are not run-to-run deterministic.

Every torch.autograd.Function of the package gets its ``backward`` wrapped: the wrapper records a bitwise checksum of the gradients that
arrive and of the gradients the op returns.  The same training step (fresh model, same weights, same scenes) is traced REPS times; an
op whose INPUT checksums equal those of repetition 0 while its OUTPUT checksums differ is a source of nondeterminism (ops downstream of
one merely inherit different inputs).  usage: python tools/grad_trace.py [n_points] [reps] [bf16]"""
import collections
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

TRACE = []


def _h(t):
    if not isinstance(t, torch.Tensor):
        return None
    t = t.detach().contiguous()
    if t.numel() == 0:
        return 0
    if t.element_size() == 4:
        return int(t.view(torch.int32).to(torch.int64).sum().item())
    if t.element_size() == 8:
        return int(t.view(torch.int64).sum().item())
    if t.element_size() == 2:
        return int(t.view(torch.int16).to(torch.int64).sum().item())
    return int(t.to(torch.int64).sum().item())


def _all_functions():
    out, todo = [], list(torch.autograd.Function.__subclasses__())
    while todo:
        c = todo.pop()
        todo.extend(c.__subclasses__())
        if c.__module__.startswith('unidet3d_amd') and 'backward' in c.__dict__:
            out.append(c)
    return out


def _wrap(cls):
    orig = cls.__dict__['backward'].__func__ if isinstance(cls.__dict__['backward'], staticmethod) else cls.__dict__['backward']

    def traced(ctx, *grads):
        res = orig(ctx, *grads)
        outs = res if isinstance(res, tuple) else (res,)
        saved = tuple(_h(t) for t in getattr(ctx, 'saved_tensors', ()))
        TRACE.append((cls.__name__, tuple(_h(g) for g in grads), tuple(_h(o) for o in outs), saved,
                      tuple(tuple(o.shape) if isinstance(o, torch.Tensor) else None for o in outs)))
        return res
    cls.backward = staticmethod(traced)


def main():
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    mode = 'bf16' if len(sys.argv) > 3 and sys.argv[3] == 'bf16' else 'fp32'
    from _detw import fill_state_dict
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd import precision as P
    from unidet3d_amd import criterion, dense, encoder, ops, sparse, unidet3d  # noqa: F401  (define the Functions)
    from unidet3d_amd.config import build_model, scannet_model_cfg
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    fns = _all_functions()
    for c in fns:
        _wrap(c)
    print('traced Functions:', sorted(c.__name__ for c in fns))
    cfg = scannet_model_cfg(voxel_size=0.05 if n_points < 50_000 else 0.02)
    cfg['decoder']['num_layers'] = 2
    inputs, samples = make_batch_inputs([make_scene(70, n_points=n_points), make_scene(71, n_points=n_points)], 'cuda:0')
    traces = []
    for _ in range(reps):
        model = fill_state_dict(build_model(cfg), tag0=3000, scale=0.06).to('cuda:0').train()
        TRACE.clear()
        with P.operands(mode):
            loss = model.loss(inputs, copy.deepcopy(samples))['det_loss']
            loss.backward()
        torch.cuda.synchronize()
        traces.append(list(TRACE))
    print('loss', float(loss), 'backward ops traced per step:', len(traces[0]))
    for r in range(1, reps):
        a, b = traces[0], traces[r]
        assert [x[0] for x in a] == [x[0] for x in b], 'different op sequences'
        src = collections.Counter()
        first = None
        n_diff = 0
        for i, (x, y) in enumerate(zip(a, b)):
            same_in = x[1] == y[1] and x[3] == y[3]
            same_out = x[2] == y[2]
            if not same_out:
                n_diff += 1
            if same_in and not same_out:
                which = [j for j, (p, q) in enumerate(zip(x[2], y[2])) if p != q]
                src[(x[0], tuple(which), tuple(x[4][j] for j in which))] += 1
                if first is None:
                    first = (i, x[0], which, [x[4][j] for j in which])
            if x[3] != y[3] and first is None:
                first = (i, x[0], 'SAVED TENSORS DIFFER (forward nondeterminism)', None)
        print(f'rep {r} vs rep 0: {n_diff} of {len(a)} ops return different gradients; first source: {first}')
        for k, n in src.most_common(25):
            print(f'    source x{n}: {k[0]} outputs {list(k[1])} shapes {list(k[2])}')


if __name__ == '__main__':
    main()
