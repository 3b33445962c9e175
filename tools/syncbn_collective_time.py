"""What the SyncBatchNorm exchange costs per step when every collective really goes through RCCL: ONE RCCL rank on
cuda:0 under dist.force_collectives() runs the cfg2 training step (8 scenes x 100k points); every dist.all_reduce is bracketed by HIP
events on the stream it is issued from, and the whole step is timed with the exchange forced and without.  A one-rank all-reduce moves
no data over xGMI -- what this measures is the FIXED cost of the 90 fp64 exchanges on the critical path (launch of the RCCL kernel,
stream hand-over, the separate statistics / finalize / apply launches the exchange path needs instead of the fused call): the floor
an N-GPU run adds the link latency to.  usage: python tools/syncbn_collective_time.py [steps=10] [scenes=8] [points=100000]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
points = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
torch.cuda.set_device(0)
import torch.distributed as dist  # noqa: E402
import unidet3d_amd  # noqa: E402,F401
from unidet3d_amd import dist as D  # noqa: E402
from unidet3d_amd import sparse  # noqa: E402
from unidet3d_amd.config import build_model, scannet_model_cfg  # noqa: E402
from unidet3d_amd.data import make_batch_inputs  # noqa: E402
from unidet3d_amd.synthetic import make_scene  # noqa: E402

D.init_from_env('nccl', force=True)
dev = torch.device('cuda:0')
inputs, samples = make_batch_inputs([make_scene(i, n_points=points) for i in range(n_scenes)], dev)
real = dist.all_reduce
events = []


def timed_all_reduce(t, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real(t, *a, **kw)
    e1.record()
    events.append((str(t.dtype), t.numel(), kw.get('group') is not None, e0, e1))
    return r


def run(forced, instrument):
    D.force_collectives(forced)
    sparse.set_wgrad_overlap(2)
    torch.manual_seed(0)
    model = build_model(scannet_model_cfg()).to(dev).train()
    D.broadcast_params(model)
    params = [p for p in model.parameters() if p.requires_grad]
    bucket = D.FlatGradBucket(params, attach=False)
    if forced:
        bucket.enable_overlap()
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.05, fused=True)

    def step():
        bucket.clear_grads()
        loss = model.loss(inputs, samples)['det_loss']
        loss.backward()
        bucket.sync()
        bucket.clip_grad_norm_(10.0)
        opt.step()
        model.prefetch(inputs, samples)
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    if instrument:
        dist.all_reduce = timed_all_reduce
    events.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    dist.all_reduce = real
    return dt


ms_plain = run(False, False)
ms_forced = run(True, False)
ms_forced_i = run(True, True)
torch.cuda.synchronize()
f64 = [(e0.elapsed_time(e1) * 1e3, n) for dt_, n, g, e0, e1 in events if dt_ == 'torch.float64']
f32 = [(e0.elapsed_time(e1) * 1e3, n) for dt_, n, g, e0, e1 in events if dt_ == 'torch.float32']
print(f'RCCL {torch.cuda.nccl.version()} one rank on {torch.cuda.get_device_name(0)}; cfg2 step: {n_scenes} scenes x {points} points, {steps} steps per arm')
print(f'step without a process-group exchange (fused batch-norm calls):        {ms_plain:7.3f} ms')
print(f'step with every collective forced through RCCL:                        {ms_forced:7.3f} ms  (+{ms_forced - ms_plain:.3f} ms)')
print(f'  same, with HIP events around every all_reduce:                       {ms_forced_i:7.3f} ms')
if f64:
    us = sorted(u for u, _ in f64)
    print(f'SyncBatchNorm exchanges (fp64 [sum x, sum x^2, n] / [sum dy, sum dy x^]): {len(f64) / steps:.0f} per step, '
          f'GPU time per all_reduce: median {us[len(us) // 2]:.1f} us, p10 {us[len(us) // 10]:.1f}, p90 {us[len(us) * 9 // 10]:.1f}, '
          f'sum {sum(us) / steps / 1e3:.3f} ms / step')
if f32:
    us = [u for u, _ in f32]
    print(f'gradient buckets (fp32, own communicator): {len(f32) / steps:.0f} per step, {sum(n for _, n in f32) / steps * 4 / 1e6:.1f} MB / step, '
          f'GPU time {sum(us) / steps / 1e3:.3f} ms / step')
dist.destroy_process_group()
