"""Run-to-run determinism of the training step on the GPU box.

The same step (fresh model from the same deterministic weights, same scenes) is run REPS times inside one process for every
arithmetic mode and for both batch-norm launch sequences (fused; separate statistics / finalize / apply as under data
parallelism), and the loss, the flat gradient and two running statistics are compared BITWISE between repetitions.  Between
launch sequences only the relative distance is reported.  usage: python tools/determinism.py [n_points] [reps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    quick = len(sys.argv) > 3 and sys.argv[3] == 'quick'          # default fp32 math, fused launch sequence only
    from _detw import fill_state_dict
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd import dist as D
    from unidet3d_amd import precision as P
    from unidet3d_amd.config import build_model, scannet_model_cfg
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    import copy
    os.environ.setdefault('MASTER_PORT', '29517')
    D.init_from_env('nccl', force=True)            # one RCCL rank: force_collectives() then drives the data-parallel launch sequence
    cfg = scannet_model_cfg(voxel_size=0.05 if n_points < 50_000 else 0.02)
    cfg['decoder']['num_layers'] = 2
    inputs, samples = make_batch_inputs([make_scene(70, n_points=n_points), make_scene(71, n_points=n_points)], 'cuda:0')
    out = {}
    for math in ('bf16x3',) if quick else ('bf16x3', 'mfma'):
        P.set_fp32_math(math)
        for mode in ('fp32',) if quick else ('fp32', 'bf16'):
            if mode == 'bf16' and math == 'mfma':
                continue
            per_seq = {}
            for forced in (False,) if quick else (False, True):
                D.force_collectives(forced)
                runs = []
                for _ in range(reps):
                    model = fill_state_dict(build_model(cfg), tag0=3000, scale=0.06).to('cuda:0').train()
                    with P.operands(mode):
                        loss = model.loss(inputs, copy.deepcopy(samples))['det_loss']
                        loss.backward()
                    torch.cuda.synchronize()
                    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
                    named = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
                    runs.append((loss.detach().clone(), flat, model.output_layer[0].running_mean.clone(),
                                 model.unet.u.u.blocks[0].conv_branch[0].running_var.clone(), named))
                same = dict(loss=all(torch.equal(runs[0][0], r[0]) for r in runs[1:]), grads=all(torch.equal(runs[0][1], r[1]) for r in runs[1:]),
                            stats=all(torch.equal(runs[0][2], r[2]) and torch.equal(runs[0][3], r[3]) for r in runs[1:]))
                worst = max(float((runs[0][1] - r[1]).abs().max() / runs[0][1].abs().max()) for r in runs[1:])
                per_seq[forced] = runs[0]
                differing = {}
                for k, g0 in runs[0][4].items():
                    d = max(float((g0 - r[4][k]).abs().max() / g0.abs().max().clamp_min(1e-30)) for r in runs[1:])
                    if d > 0:
                        differing[k] = (d, list(g0.shape), [int((g0 != r[4][k]).sum()) for r in runs[1:]])
                out[f'{math}/{mode}/{"separate" if forced else "fused"}'] = dict(bitwise_equal_over_reps=same, worst_grad_rel=worst, reps=reps, tensors_that_differ=differing)
            D.force_collectives(False)
            if quick:
                continue
            a, b = per_seq[False], per_seq[True]
            out[f'{math}/{mode}/fused_vs_separate'] = dict(loss_rel=float((a[0] - b[0]).abs() / a[0].abs()), loss_equal=bool(torch.equal(a[0], b[0])),
                                                           grad_rel=float((a[1] - b[1]).abs().max() / a[1].abs().max()),
                                                           stats_rel=float((a[3] - b[3]).abs().max() / a[3].abs().max()))
    P.set_fp32_math('bf16x3')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
