"""CPU-only analysis (run as a script; test infrastructure, imports the oracle): why the backbone's parameter gradients of ANY fp32
evaluation differ from an fp64 evaluation by ~5e-3 (median over tensors) / 14 % (worst tensor) on cfg1, and what is left when the
ReLU on/off decisions are held fixed.  Output kept as profiles/round3_relu_flip_analysis.txt.

  python tests/analysis_relu_flips.py [scene_seed=40]
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import _parity as PA                                   # noqa: E402
from _detw import fill_state_dict                      # noqa: E402
from oracle import model as om                         # noqa: E402
from unidet3d_amd.config import build_model, scannet_model_cfg      # noqa: E402
from unidet3d_amd.synthetic import make_scene          # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cfg = scannet_model_cfg(voxel_size=0.05)
    prod = fill_state_dict(build_model(cfg), tag0=3000, scale=0.06)
    orac = om.ODetector(backbone=cfg['backbone'], decoder=cfg['decoder'], voxel_size=cfg['voxel_size'])
    orac.load_state_dict(prod.state_dict(), strict=True)
    scenes = [make_scene(seed, n_points=10_000)]
    run = lambda m: PA.oracle_forward(m, scenes, ['scannet'])  # noqa: E731
    print(f'cfg1: scene seed {seed}, 10 k points, 5 cm voxels, the test weights (tests/_detw.py tag0=3000); oracle fp32 vs oracle fp64')
    o64 = copy.deepcopy(orac).double().train()
    m64, O64 = PA.oracle_relu_masks(o64, run)
    O64['loss'].backward()
    g64 = {k: p.grad for k, p in o64.named_parameters() if p.grad is not None}
    n_units = sum(v.numel() for v in m64.values())
    print(f'{len(m64)} BatchNorm+ReLU layers, {n_units} units; rows per level: {sorted({v.shape[0] for v in m64.values()}, reverse=True)}')
    bb = [k for k in g64 if not k.startswith('decoder.')]
    for thr in (1, 4, 8, 16):
        torch.set_num_threads(thr)
        o32 = copy.deepcopy(orac).train()
        vals, hooks, mods = {}, [], dict(o32.named_modules())
        for name, _, _ in PA.bn_relu_pairs(o32):
            hooks.append(mods[name].register_forward_hook(lambda m, i, o, name=name: vals.__setitem__(name, o.detach().clone())))
        O = run(o32)
        for h in hooks:
            h.remove()
        O['loss'].backward()
        g32 = {k: p.grad for k, p in o32.named_parameters() if p.grad is not None}
        m32 = {k: v > 0 for k, v in vals.items()}
        flips = [(k, int((m32[k] != m64[k]).sum()), m32[k].shape[0]) for k in m32 if bool((m32[k] != m64[k]).any())]
        g64m, _ = PA.oracle_fp64_grads_same_activation_pattern(orac, run, m32)
        print(f'--- torch CPU threads = {thr}: loss fp32 {float(O["loss"].detach()):.7f} fp64 {float(O64["loss"].detach()):.7f}; '
              f'units whose ReLU decision differs from the fp64 run: {sum(f[1] for f in flips)}  {[(k, n, f"{rows} rows") for k, n, rows in flips]}')
        for tag, ref in (('vs fp64, own decisions       ', g64), ('vs fp64, the fp32 run\'s decisions', g64m)):
            e = np.array([PA.rel(g32[k], ref[k]) for k in bb])
            st = PA.flat_gradient_stats({'o32': g32}, ref, bb)['o32']
            worst = max(bb, key=lambda k: PA.rel(g32[k], ref[k]))
            print(f'    backbone gradients {tag}: per-tensor max-norm rel. error median {np.median(e):.2e} p90 {np.percentile(e, 90):.2e} '
                  f'max {e.max():.2e} ({worst}) | flat: 1-cos {1 - st["cos"]:.1e}, L2 rel {st["l2_rel"]:.1e}, directional {[f"{x:.1e}" for x in st["dir_rel"]]}')


if __name__ == '__main__':
    main()
