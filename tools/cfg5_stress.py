"""BASELINE configs[4] probe: ONE scene of 1 M points (a large room, ~10x the ScanNet footprint) through a full training step on cuda:0 --
memory and time of the path far outside the benchmark shape.  usage (GPU box): python tools/cfg5_stress.py"""
import sys, time, torch
sys.path.insert(0, '.')
from unidet3d_amd.config import build_model, scannet_model_cfg
from unidet3d_amd.data import make_batch_inputs
from unidet3d_amd.synthetic import make_scene
dev = 'cuda:0'
torch.manual_seed(0)
model = build_model(scannet_model_cfg()).to(dev).train()
sc = make_scene(500, n_points=1_000_000, area_scale=10.0, n_furniture=40)
inputs, samples = make_batch_inputs([sc], dev)
for it in range(3):
    for p in model.parameters(): p.grad = None
    torch.cuda.synchronize(); t = time.perf_counter()
    loss = model.loss(inputs, samples)['det_loss']; loss.backward()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f'iter {it}: loss {loss.item():.4f} voxels {model._vb.coords.shape[0]} superpoints {int(sc.superpoints.max()) + 1} '
          f'{dt * 1e3:.1f} ms  peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB', flush=True)
assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
print('finite grads ok')
