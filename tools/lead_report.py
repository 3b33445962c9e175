"""How far does the host run ahead of the GPU inside a training step?  At each phase boundary the host time of issuing an event and
the GPU time of reaching it are recorded (no synchronisation inside the step); lead = GPU time - host time.  A lead near zero means
the GPU is waiting for the host there (launch-bound); a large lead means the queue is full.  usage (GPU box): python tools/lead_report.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from unidet3d_amd.config import build_model, scannet_model_cfg  # noqa: E402
from unidet3d_amd.data import make_batch_inputs  # noqa: E402
from unidet3d_amd.dist import FlatGradBucket  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = build_model(scannet_model_cfg(voxel_size=0.02)).to(dev)
model.train()
params = [p for p in model.parameters() if p.requires_grad]
bucket = FlatGradBucket(params, attach=False)
opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.05, fused=True)
inputs, samples = make_batch_inputs([make_scene(i, n_points=100000) for i in range(8)], dev)
prefetch = '--no-prefetch' not in sys.argv
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, time.perf_counter(), e))


def step():
    bucket.clear_grads()
    mark('step start')
    pre, model._prefetched = model._prefetched, None
    if pre is not None and pre[0] is inputs:
        torch.cuda.current_stream().wait_event(pre[3])
        prep = pre[2]
    else:
        prep = model._prepare_train(inputs, samples)
    mark('front ready')
    vb, plan, offs, names = prep['vb'], prep['plan'], prep['batch_offsets'], prep['names']
    model._vb = vb
    feats = model.extract_feat(prep['x'], plan, vb.inverse, offs)
    mark('backbone+pool fwd issued')
    q, c, gts = model._select_queries(feats, prep['sp_gt_instances'], None)
    out = model.decoder(q, c, names)
    mark('decoder fwd issued')
    loss = model.criterion(out, gts, names)['det_loss']
    mark('criterion fwd issued')
    loss.backward()
    mark('backward issued')
    bucket.sync()
    bucket.clip_grad_norm_(10.0)
    opt.step()
    mark('clip+adamw issued')
    if prefetch:
        model.prefetch(inputs, samples)
        mark('prefetch issued (side stream)')


for _ in range(4):
    step()
torch.cuda.synchronize()
marks.clear()
base = torch.cuda.Event(enable_timing=True)
base.record()
torch.cuda.synchronize()
t_base = time.perf_counter()
N = 4
for _ in range(N):
    step()
torch.cuda.synchronize()
t_end = time.perf_counter()
print(f'prefetch={prefetch}: {(t_end - t_base) / N * 1e3:.2f} ms/step')
print(f'{"phase boundary":34s} {"host ms":>9s} {"gpu ms":>9s} {"lead ms":>8s}   (host / gpu time since the previous boundary)')
ph, pg = 0.0, 0.0
for name, th, e in marks:
    h = (th - t_base) * 1e3
    g = base.elapsed_time(e)
    print(f'{name:34s} {h:9.2f} {g:9.2f} {g - h:8.2f}   (+{h - ph:6.2f} / +{g - pg:6.2f})')
    ph, pg = h, g
