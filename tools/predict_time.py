"""Inference latency of UniDet3D.predict (forward + top-k + NMS + superpoint trimming) on one 100k-point scene."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidet3d_amd.config import build_model, scannet_model_cfg
from unidet3d_amd.data import make_batch_inputs
from unidet3d_amd.synthetic import make_scene
dev = 'cuda:0'
torch.manual_seed(0)
model = build_model(scannet_model_cfg()).to(dev).eval()
inputs, samples = make_batch_inputs([make_scene(3)], dev)
with torch.no_grad():
    for _ in range(3):
        res = model.predict(inputs, samples)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 20
    for _ in range(n):
        res = model.predict(inputs, samples)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    p = res[0].pred_instances_3d
print(f'predict: {dt * 1e3:.2f} ms/scene ({1 / dt:.1f} scenes/s), {len(p.labels_3d)} boxes after NMS + trimming, '
      f'{model._vb.coords.shape[0]} voxels')
