"""Host-side cost of one training step: small scenes make the GPU work negligible, so wall time ~ the time the host needs to
issue a step (launch overheads, autograd, allocator, read-backs).  cProfile of 5 steps.  usage (GPU box): python tools/cpu_profile.py [points]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from unidet3d_amd.config import build_model, scannet_model_cfg  # noqa: E402
from unidet3d_amd.data import make_batch_inputs  # noqa: E402
from unidet3d_amd.dist import FlatGradBucket  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

pts = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = build_model(scannet_model_cfg(voxel_size=0.02)).to(dev)
model.train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = FlatGradBucket(params, attach=False)
opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.05, fused=True)
inputs, samples = make_batch_inputs([make_scene(i, n_points=pts) for i in range(8)], dev)


def step():
    bucket.clear_grads()
    loss = model.loss(inputs, samples)['det_loss']
    loss.backward()
    bucket.sync()
    bucket.clip_grad_norm_(10.0)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f'{pts} points/scene: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per step (host-bound)')
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(60)
