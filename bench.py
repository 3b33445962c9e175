#!/usr/bin/env python
"""Headline benchmark: scenes/sec forward+backward of UniDet3D's detection hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): 8 synthetic ScanNet-shape scenes per GPU, 100k points each,
0.02 m voxels, fp32, the ScanNet model config (5-level sparse U-Net 32..160 ch, 6-layer 256-d decoder).
One step = one pass of the hot path over the batch with the points already resident in HBM:
voxelise -> rulebooks -> backbone -> superpoint pooling -> decoder -> matcher/loss -> backward ->
(gradient all-reduce when N > 1) -> grad clip + AdamW.  Weak scaling: every rank runs its own 8 scenes.
Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak (= the fp32 vector rate)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA peak (AMD's 5 PF figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0
METRIC = 'scenes/sec fwd+bwd, 100k-pt ScanNet voxel grid, 1/2/4/8 MI355X'      # BASELINE.json's metric, verbatim


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=None, help='scenes per GPU (cfg2: 8; cfg3 = --dtype bf16: 16)')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                    help="MFMA operand precision: fp32 = BASELINE configs[1] (the headline line), bf16 = configs[2] "
                         "(bf16 operands, fp32 accumulate / statistics / optimizer; unidet3d_amd/precision.py)")
    ap.add_argument('--points', type=int, default=100_000)
    ap.add_argument('--voxel-size', type=float, default=0.02)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: nccl (= RCCL over xGMI); 'gloo' lets two ranks "
                    "share one GPU to exercise the N>1 code path on a single-GPU box")
    ap.add_argument('--no-optimizer', action='store_true', help='time fwd+bwd only (diagnostic, not the reported config)')
    ap.add_argument('--no-prefetch', action='store_true',
                    help="build the next step's voxel grid / rulebooks at the start of that step instead of on a side stream")
    return ap.parse_args()


def usable_cores() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (os.cpu_count()
    reports the host's cores inside a container and oversubscribes the OpenMP pool)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, 64))


def _cpu_baseline_child(points: int, voxel_size: float, cores: int):
    """Runs in a child process (bounded by a timeout in the parent)."""
    from oracle import criterion as oc
    from oracle import model as om
    from unidet3d_amd.config import scannet_model_cfg
    from unidet3d_amd.synthetic import make_scene
    torch.set_num_threads(cores)
    cfg = scannet_model_cfg(voxel_size=voxel_size)
    torch.manual_seed(0)
    det = om.ODetector(backbone=cfg['backbone'], decoder=cfg['decoder'], voxel_size=voxel_size)
    det.train()

    def run(sc):
        p = [torch.from_numpy(sc.points)]
        s = [torch.from_numpy(sc.superpoints)]
        feats, _ = det.extract_feat(p, s)
        out = det.decoder(feats, det.sp_centers(p, s), ['scannet'])
        inst = oc.gt_from_scene(p[0][:, :3] - p[0][:, :3].min(0)[0], torch.from_numpy(sc.instance_mask),
                                torch.from_numpy(sc.labels), s[0])
        loss = oc.criterion(out, [inst])
        det.zero_grad()
        loss.backward()
        return float(loss.detach())

    run(make_scene(900, n_points=max(points // 20, 2000)))      # warm the thread pool / allocator
    scenes = [make_scene(i, n_points=points) for i in range(8)]  # the bench batch; as many of them as fit ~12 s of CPU work
    t0, k = time.perf_counter(), 0
    while k < len(scenes) and (k == 0 or time.perf_counter() - t0 < 12.0):
        run(scenes[k])
        k += 1
    print(json.dumps({'seconds': time.perf_counter() - t0, 'scenes': k}))


def cpu_baseline(points: int, voxel_size: float):
    """The oracle (a PyTorch-CPU restatement of the algorithm spconv's CPU path uses) timed on this
    box's host cores on a bounded sample: scenes of the same batch, forward + loss + backward each, for about 12 s."""
    import subprocess
    cores = usable_cores()
    for pts, limit in ((points, 150), (max(points // 5, 2000), 90)):
        code = f'import sys; sys.path.insert(0, {ROOT!r}); import bench; bench._cpu_baseline_child({pts}, {voxel_size}, {cores})'
        try:
            res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=limit,
                                 env=dict(os.environ, OMP_NUM_THREADS=str(cores), HIP_VISIBLE_DEVICES=''))
            rec = json.loads(res.stdout.strip().splitlines()[-1])
            dt, k = rec['seconds'], rec['scenes']
        except Exception as e:  # noqa: BLE001  (timeout / parse error -> try the smaller sample)
            log(f'cpu_baseline sample of {pts} pts failed: {type(e).__name__}')
            continue
        return dict(value=k / dt, unit='scenes/s', cores=cores, kind='port',
                    sample=f'{k} scene(s) ({pts} pts, {voxel_size} m voxels) of the bench batch, fwd+loss+bwd each, fp32, '
                           f'{dt:.1f} s with torch CPU threads={cores}' +
                           ('' if pts == points else f' (reduced from {points} pts to stay inside the time bound)'))
    return dict(value=None, unit='scenes/s', cores=cores, kind='port', sample='timed out')


def log(msg: str):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.batch is None:
        args.batch = 16 if args.dtype == 'bf16' else 8
    from unidet3d_amd import _lib as L
    from unidet3d_amd import account, precision
    from unidet3d_amd import sparse
    from unidet3d_amd.config import build_model, scannet_model_cfg
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.dist import FlatGradBucket, broadcast_params, init_from_env
    from unidet3d_amd.synthetic import make_scene

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    L.lib()                                           # fail loudly if the HIP library is missing
    if args.backend != 'nccl':
        os.environ['LOCAL_RANK_DEVICE'] = str(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    local_dev = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    rank, world, local = init_from_env(args.backend)
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    dev = torch.device('cuda', local_dev)

    precision.set_operand_dtype(args.dtype)
    torch.manual_seed(0)
    model = build_model(scannet_model_cfg(voxel_size=args.voxel_size)).to(dev)
    model.train()
    broadcast_params(model)
    params = [p for p in model.parameters() if p.requires_grad]
    bucket = FlatGradBucket(params, attach=False)
    if world > 1:
        bucket.enable_overlap()              # 16 MB buckets, all-reduced over RCCL as backward completes them (a collective
                                             # call: every rank takes the same path, a failure is fatal on all of them)
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.05, fused=True)      # configs/unidet3d_1xb8_scannet.py:712

    log(f'rank {rank}/{world}: model built, generating {args.batch} scenes')
    scenes = [make_scene(rank * args.batch + i, n_points=args.points) for i in range(args.batch)]
    inputs, samples = make_batch_inputs(scenes, dev)                              # resident in HBM before timing

    inflight = collections.deque()

    def step():
        # with the read-backs of the voxeliser on the side stream nothing in a step makes the host wait for the GPU any more
        # (it issues a step in ~18 ms, the GPU needs ~33): keep at most two steps queued
        if len(inflight) >= 2:
            inflight.popleft().synchronize()
        bucket.clear_grads()                 # backward writes fresh grads: no accumulate kernels
        loss = model.loss(inputs, samples)['det_loss']
        loss.backward()                      # N > 1: hooks copy finished buckets into the flat buffer and start their all-reduce
        bucket.sync()                        # N > 1: wait + average; N = 1: one multi-tensor copy into the flat buffer
        if not args.no_optimizer:
            bucket.clip_grad_norm_(10.0)     # clip_grad max_norm=10, norm_type=2 (configs :713) on the flat buffer
            opt.step()
        if not args.no_prefetch:
            # the batch-only part of the NEXT step (voxelisation, superpoint CSR, ground-truth boxes, rulebooks: integer kernels
            # and host read-backs) goes to a side stream now, while this step's backward is still executing; every step
            # still does this work exactly once inside the timed region (K steps issue K of them)
            model.prefetch(inputs, samples)
        ev = torch.cuda.Event()
        ev.record()
        inflight.append(ev)
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sparse.set_profile_flops(True)      # launches carry their algorithmic flops (pair counts cached during warm-up)
    log('inputs resident, warm-up')
    for i in range(max(args.warmup, 1)):
        loss = step()
        torch.cuda.synchronize()
        log(f'warm-up step {i} done, loss {float(loss.detach()):.4f}')
    assert bucket.check_views(), 'p.grad does not alias the flat gradient buffer'
    FAMILIES = (('conv_gmm', L.K_CONV_FWD), ('conv_wgrad', L.K_CONV_WGRAD), ('attn_fwd', L.K_ATTN_FWD), ('attn_bwd', L.K_ATTN_BWD),
                ('gemm', L.K_GEMM))
    # ---- the timed region: exactly K steps, nothing instrumented -------------------------------------------------------
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    log(f'timed region done: {dt / args.steps * 1e3:.2f} ms/step')
    # ---- per-family kernel times: HIP events around every launch of a family, on the stream it runs on, over extra steps
    # OUTSIDE the timed region (recording ~300 event pairs per step costs ~1 ms/step of host time) ----------------------
    prof_steps = max(1, min(args.steps, 5))
    for _, c in FAMILIES:
        L.prof_enable(c, True)
    account.reset()
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof = {}
    acc = account.snapshot()
    for name, c in FAMILIES:
        ms, n, work = L.prof_collect(c)
        L.prof_enable(c, False)
        prof[name] = dict(ms=ms, launches=n, flops=work, bytes=acc.get(name, {}).get('bytes', 0.0),
                          flops_booked=acc.get(name, {}).get('flops', 0.0))
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    loss_val = float(loss.item())

    if rank == 0:
        bf = args.dtype == 'bf16'
        # which MFMA peak prices a family: the bf16-operand kernels exist for the sparse conv forward / input gradient, the
        # decoder's NT GEMMs and attention; weight gradients (sparse and dense) keep fp32 operands
        peak = {k: (PEAK_BF16_MFMA_TFLOPS if bf and k in ('conv_gmm', 'attn_fwd', 'attn_bwd') else PEAK_F32_MFMA_TFLOPS) for k in prof}
        kernels = {}
        for k, v in prof.items():
            t = v['ms'] * 1e-3
            tf = v['flops'] / t / 1e12 if t > 0 and v['flops'] > 0 else None
            gbs = v['bytes'] / t / 1e9 if t > 0 and v['bytes'] > 0 else None
            kernels[k] = {'ms_per_step': v['ms'] / prof_steps, 'launches_per_step': v['launches'] / prof_steps,
                          'algorithmic_gflop_per_step': v['flops'] / prof_steps / 1e9, 'algorithmic_MB_per_step': v['bytes'] / prof_steps / 1e6,
                          'tflops': tf, 'mfma_peak': peak[k] if k != 'gemm' else ('mixed' if bf else PEAK_F32_MFMA_TFLOPS),
                          'frac_mfma': (tf / peak[k] if tf and (k != 'gemm' or not bf) else None),
                          'hbm_gbs': gbs, 'frac_hbm': gbs / PEAK_HBM_GBS if gbs else None}
        g = prof['conv_gmm']
        ach = g['flops'] / (g['ms'] * 1e-3) / 1e12 if g['ms'] > 0 else 0.0
        n_vox = int(model._vb.coords.shape[0])
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, 'profiles', 'round2_pmc_traffic.json')
        if os.path.exists(tpath) and not bf:   # PMC pass (FETCH_SIZE x2 + WRITE_SIZE, KB) of this same command, tools/pmc_bench.sh
            import hashlib
            rec = json.load(open(tpath))
            sha = hashlib.sha256(open(os.path.join(ROOT, 'unidet3d_amd', 'csrc', 'spconv.hip'), 'rb').read()).hexdigest()[:16]
            if rec.get('_meta', {}).get('spconv_hip_sha16') == sha:        # stale counters are not reported
                traffic = rec['_spconv_gmm_k_all']['hbm_MB_per_launch'] * 1e6
                traffic_src = (f"profiles/round2_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this "
                               f"command at commit {rec['_meta'].get('git_head', '?')}; spconv.hip sha256/16 {sha} matches the kernel timed here)")
            else:
                traffic_src = 'profiles/round2_pmc_traffic.json was measured on a different spconv.hip: not reported'
        out = {
            'metric': METRIC,
            'value': args.batch * world * args.steps / dt,
            'unit': 'scenes/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if bf else 'f32', 'data': 'synthetic',
            'config': {'workload': f'{"cfg3" if bf else "cfg2"}: {args.batch} synthetic ScanNet-shape scenes/GPU x {args.points} pts, '
                                   f'{args.voxel_size} m voxels, unidet3d_1xb8_scannet model, '
                                   + ('bf16 MFMA operands (sparse conv fwd/dgrad, Linear fwd/dX, attention), fp32 accumulate/BN/softmax/optimizer; '
                                      if bf else 'fp32; ') +
                                   'step = voxelise+rulebook+fwd+loss+bwd' + ('' if args.no_optimizer else '+clip+AdamW')
                                   + ('' if args.no_prefetch else "; each step's voxelise+rulebook part is queued on a side stream during "
                                      "the previous step's backward"),
                       'front_prefetch': not args.no_prefetch,
                       'global_batch': args.batch * world, 'points_per_scene': args.points,
                       'active_voxels_per_gpu': n_vox, 'parallelism': f'dp{world}', 'loss': loss_val},
            'roofline': {'kernel': 'spconv_gmm_k (sparse conv forward + input-gradient, all levels)',
                         'bound': 'mfma', 'achieved': ach, 'peak': peak['conv_gmm'], 'unit': 'TFLOP/s',
                         'frac': ach / peak['conv_gmm'], 'traffic': traffic, 'traffic_unit': 'bytes/launch (HBM, PMC)',
                         'traffic_source': traffic_src,
                         'algorithmic_bytes_per_launch': g['bytes'] / max(g['launches'], 1),
                         'hbm_achieved_gbs': kernels['conv_gmm']['hbm_gbs'], 'hbm_frac': kernels['conv_gmm']['frac_hbm'],
                         'launches': g['launches'], 'avg_launch_us': g['ms'] * 1e3 / max(g['launches'], 1),
                         'algorithmic_gflop_per_launch': g['flops'] / max(g['launches'], 1) / 1e9,
                         'share_of_step': (g['ms'] / prof_steps) / (dt / args.steps * 1e3),
                         'timing': f'HIP events around each launch over {prof_steps} instrumented steps run after the timed region'},
            'kernels': kernels,
            'step_roofline': {
                'note': 'all five timed families: sum of algorithmic flops / (sum of their time); HBM side: sum of algorithmic bytes / time',
                'families_ms_per_step': sum(v['ms'] for v in prof.values()) / prof_steps,
                'tflops': sum(v['flops'] for v in prof.values()) / max(sum(v['ms'] for v in prof.values()) * 1e-3, 1e-12) / 1e12,
                'hbm_gbs': sum(v['bytes'] for v in prof.values()) / max(sum(v['ms'] for v in prof.values()) * 1e-3, 1e-12) / 1e9,
                'share_of_step': (sum(v['ms'] for v in prof.values()) / prof_steps) / (dt / args.steps * 1e3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.points, args.voxel_size)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
