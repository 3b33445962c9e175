"""Host-side (Python launch) time vs GPU time of one bench step."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd.config import build_model, scannet_model_cfg
from unidet3d_amd.data import make_batch_inputs
from unidet3d_amd.synthetic import make_scene
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_model(scannet_model_cfg()).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.05, fused=True)
inputs, samples = make_batch_inputs([make_scene(i) for i in range(8)], dev)
def step(parts):
    t = [time.perf_counter()]
    for p in params: p.grad = None
    loss = model.loss(inputs, samples)['det_loss']; t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    torch.nn.utils.clip_grad_norm_(params, 10, foreach=True); opt.step(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    parts.append([b - a for a, b in zip(t[:-1], t[1:])])
for _ in range(3): step([])
torch.cuda.synchronize()
parts = []
for _ in range(5): step(parts)
import numpy as np
m = np.array(parts).mean(0) * 1e3
print('host ms: forward+loss %.1f  backward %.1f  clip+adamw %.1f  | tail sync %.1f  total %.1f' % (m[0], m[1], m[2], m[3], m.sum()))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step([]); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
