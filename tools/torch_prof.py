"""GPU time of one training step grouped by the launching aten / autograd op (torch.profiler), to find the torch-side
elementwise tail.  usage: python tools/torch_prof.py [rows]"""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
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
def step():
    for p in params: p.grad = None
    loss = model.loss(inputs, samples)['det_loss']
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 10, foreach=True); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 45
ev = [e for e in prof.key_averages() if e.self_device_time_total > 0]
ev.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in ev)
print(f'device time in torch-launched kernels: {tot / N / 1e3:.2f} ms/step')
for e in ev[:rows]:
    print(f'{e.self_device_time_total / N / 1e3:8.3f} ms/step {e.count / N:7.0f} calls  {e.key[:90]}')

# shapes behind the small aten ops
by = {}
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ('aten::add_', 'aten::copy_', 'aten::fill_', 'aten::mul', 'aten::sum') and e.self_device_time_total > 0:
        by.setdefault(e.key, []).append((e.self_device_time_total / N / 1e3, e.count / N, str(e.input_shapes)[:90]))
for k, v in by.items():
    v.sort(reverse=True)
    print(k)
    for t, c, sh in v[:7]:
        print(f'   {t:7.3f} ms/step {c:6.0f} calls  {sh}')
