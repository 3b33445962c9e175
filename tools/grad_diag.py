"""Per-parameter gradient error of the product (GPU) and of the fp32 CPU oracle against the fp64 oracle, one small scene batch.
usage (GPU box): python tools/grad_diag.py [n_scenes n_points voxel_size]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import _parity as PA  # noqa: E402
from unidet3d_amd.config import scannet_model_cfg  # noqa: E402
from unidet3d_amd.data import make_batch_inputs  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_points = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
vs = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
cfg = scannet_model_cfg(voxel_size=vs)
prod, orac = PA.build_pair(cfg)
scenes = [make_scene(40 + i, n_points=n_points) for i in range(n_scenes)]
names = ['scannet'] * n_scenes
O = PA.oracle_forward(orac, scenes, names)
g64 = PA.oracle_fp64_grads(orac, lambda m: PA.oracle_forward(m, scenes, names))
inputs, samples = make_batch_inputs(scenes, PA.DEV)
P = PA.product_forward(prod, inputs, samples)
P['loss'].backward(); O['loss'].backward()
og, pg = dict(orac.named_parameters()), dict(prod.named_parameters())
rows = []
for k, g in g64.items():
    rows.append((k, PA.rel(pg[k].grad, g), PA.rel(og[k].grad, g), float(g.abs().max()), tuple(g.shape)))
rows.sort(key=lambda r: -r[1])
print('loss', float(P['loss']), float(O['loss']))
print(f'{"parameter":60s} {"prod/64":>9s} {"o32/64":>9s} {"|g|max":>9s} shape')
for r in rows:
    print(f'{r[0]:60s} {r[1]:9.2e} {r[2]:9.2e} {r[3]:9.2e} {r[4]}')
# also: decoder input gradient (d loss / d pooled features) -- where backbone errors start
