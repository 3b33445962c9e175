"""GPU time per phase of one training step (phases are synchronised, so the sum exceeds the pipelined step)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unidet3d_amd import ops
from unidet3d_amd.config import build_model, scannet_model_cfg
from unidet3d_amd.data import make_batch_inputs
from unidet3d_amd.synthetic import make_scene
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_model(scannet_model_cfg()).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.05, fused=True)
inputs, samples = make_batch_inputs([make_scene(i) for i in range(8)], dev)
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
def step():
    for p in params: p.grad = None
    torch.cuda.synchronize(); t = time.perf_counter()
    vb, plan, offs, cent, names = model._front(inputs, samples, True); t = tick('front: voxelise+csr+centres', t)
    for i, ds in enumerate(samples):
        inst = ds.gt_instances_3d
        pts = inputs['points'][i][:, :3] - vb.stats[i, :3]
        inst.bboxes_3d = model.get_bboxes_by_masks(ds.gt_pts_seg.pts_instance_mask, len(inst.labels_3d), pts)
        inst.sp_centers = cent[i]
    t = tick('gt boxes', t)
    x = model._sparse_input(8); model.unet.prepare_geometry(x); t = tick('rulebooks (all levels)', t)
    x = model.input_conv(x); x, _ = model.unet(x); x = model.output_layer(x); t = tick('backbone fwd', t)
    pooled = ops.superpoint_pool(x.features, plan); feats = [pooled[offs[i]:offs[i + 1]] for i in range(8)]; t = tick('pool fwd', t)
    q, c, g = model._select_queries(feats, [s.gt_instances_3d for s in samples])
    out = model.decoder(q, c, names); t = tick('decoder fwd', t)
    loss = model.criterion(out, g, names)['det_loss']; t = tick('criterion fwd', t)
    loss.backward(); t = tick('backward (all)', t)
    torch.nn.utils.clip_grad_norm_(params, 10, foreach=True); opt.step(); t = tick('clip+adamw', t)
for _ in range(2): step()
T.clear()
N = 5
for _ in range(N): step()
tot = sum(T.values())
for k, v in T.items(): print(f'{k:32s} {v / N * 1e3:7.2f} ms')
print(f'{"sum (synchronised phases)":32s} {tot / N * 1e3:7.2f} ms')
# criterion alone (forward + backward to the decoder outputs), kernel count from the torch profiler
vb, plan, offs, cent, names = model._front(inputs, samples, True)
x = model._sparse_input(8); x = model.input_conv(x); x, _ = model.unet(x); x = model.output_layer(x)
pooled = ops.superpoint_pool(x.features, plan); feats = [pooled[offs[i]:offs[i + 1]] for i in range(8)]
for i, ds in enumerate(samples):
    ds.gt_instances_3d.sp_centers = cent[i]
q, c, g = model._select_queries(feats, [s.gt_instances_3d for s in samples])
with torch.no_grad():
    out = model.decoder(q, c, names)
pk = out['_packed']
leaves = [t.detach().clone().requires_grad_() for t in pk['cls'] + pk['box']]
L_ = len(pk['cls'])
out2 = dict(out, _packed=dict(cls=leaves[:L_], box=leaves[L_:], sizes=pk['sizes']))
def crit():
    loss = model.criterion(out2, g, names)['det_loss']
    loss.backward()
for _ in range(3): crit()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): crit()
torch.cuda.synchronize()
print(f'criterion fwd+bwd alone: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms')
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    crit(); torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0]
print('criterion: %d kernel launches, %.2f ms of GPU time' % (sum(e.count for e in ev), sum(e.device_time_total for e in ev) / 1e3))
