#!/usr/bin/env python
"""Headline benchmark: scenes/sec forward+backward of UniDet3D's detection hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a torchrun environment (WORLD_SIZE unset): the script re-executes itself under
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`
  (tools/train.py:49-52,73 of the reference selects its launcher the same way); launched BY torchrun it reads
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

Workload (BASELINE.json configs[1]): 8 synthetic ScanNet-shape scenes per GPU, 100k points each,
0.02 m voxels, fp32, the ScanNet model config (5-level sparse U-Net 32..160 ch, 6-layer 256-d decoder).
One step = one pass of the hot path over the batch with the points already resident in HBM:
voxelise -> rulebooks -> backbone -> superpoint pooling -> decoder -> matcher/loss -> backward ->
(gradient all-reduce when N > 1) -> grad clip + AdamW.  Weak scaling: every rank runs its own 8 scenes.
Prints ONE JSON line (rank 0).  The line's headline is the fp32 cfg2 measurement; measured right after it in the same process ride
along (N = 1 only): `fp32_native_mfma` (the same workload on the native fp32 MFMA kernels), `cfg3` (BASELINE.json configs[2]: bf16
MFMA operands, 16 scenes per GPU), `cfg4` and `cfg5` (configs[3] / [4], one GPU's share each: the six-dataset joint batch, one 1 M-point
room) and `cpu_baseline`; --no-mfma-line / --no-cfg3 / --no-extra-configs / --no-cpu-baseline skip them.  `tree_hash` stamps the line
with a hash of the shipped sources (`python bench.py --tree-hash` prints it without a GPU).
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
    ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg4', 'cfg5'],
                    help="workload: cfg2 = BASELINE.json configs[1] (the metric's configuration; default; the line also carries cfg3); "
                         "cfg4 = configs[3]'s per-GPU share: the six-dataset joint model, 8 mixed scenes of 100k / 180k / 200k points by "
                         "dataset; cfg5 = configs[4]'s per-GPU share: ONE S3DIS-shape room of 1 M points.  cfg4 / cfg5 print the same "
                         "line shape (value in scenes/s of THAT workload) and never stand in for the headline")
    ap.add_argument('--batch', type=int, default=None, help='scenes per GPU (cfg2: 8; cfg3 = --dtype bf16: 16)')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                    help="MFMA operand precision of the HEADLINE: fp32 = BASELINE configs[1] (default), bf16 = configs[2] "
                         "(bf16 operands, fp32 accumulate / statistics / optimizer; unidet3d_amd/precision.py)")
    ap.add_argument('--fp32-math', default=None, choices=['bf16x3', 'mfma'],
                    help="how the fp32 path forms its products: 'bf16x3' (library default) = three exact bf16 planes per operand, six bf16 "
                         "MFMAs per product, fp32-level error; 'mfma' = native v_mfma_f32_* (round 1-2 headline kernels)")
    ap.add_argument('--no-mfma-line', action='store_true',
                    help='skip the `fp32_native_mfma` block (the same fp32 workload on the native fp32 MFMA kernels) appended to the default line')
    ap.add_argument('--no-cfg3', action='store_true', help='skip the cfg3 block (bf16 operands, 16 scenes/GPU) appended to the fp32 line')
    ap.add_argument('--no-extra-configs', action='store_true',
                    help="skip the cfg4 / cfg5 blocks (BASELINE.json configs[3] / [4] per-GPU shares, fp32) appended to the default line")
    ap.add_argument('--tree-hash', action='store_true', help='print the hash of the shipped source tree (the stamp of every line) and exit; needs no GPU')
    ap.add_argument('--points', type=int, default=100_000)
    ap.add_argument('--voxel-size', type=float, default=0.02)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: nccl (= RCCL over xGMI); 'gloo' lets several ranks "
                    "share one GPU to exercise the N>1 code path on a single-GPU box")
    ap.add_argument('--no-optimizer', action='store_true', help='time fwd+bwd only (diagnostic, not the reported config)')
    ap.add_argument('--optimizer', default='adamw', choices=['adamw', 'sgd'],
                    help='adamw = configs/unidet3d_1xb8_scannet.py:710-713 (the reported config); sgd = diagnostic (DESIGN.md section 2: '
                         "AdamW's first updates are lr*sign(g), which turns rounding-level gradient differences into different trajectories)")
    ap.add_argument('--wgrad-overlap', type=int, default=None, choices=[0, 1, 2],
                    help="weight gradients on a side stream (unidet3d_amd/sparse.py set_wgrad_overlap): 0 off, 1 next to the layer's "
                         'input gradient, 2 (default here) a chain of their own joined at the end of backward')
    ap.add_argument('--no-prefetch', action='store_true',
                    help="build the next step's voxel grid / rulebooks at the start of that step instead of on a side stream")
    return ap.parse_args()


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(gpus: int, argv, port: int):
    """The torchrun command `python bench.py --gpus N` turns itself into (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__), *argv]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: become the launcher (the reference picks its
    launcher in tools/train.py:49-52,73; here the only multi-process form is one rank per GPU on one node)."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return
    cmd = launch_command(args.gpus, sys.argv[1:], _free_port())
    log('self-launch: ' + ' '.join(cmd))
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', str(max(1, usable_cores() // args.gpus))))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


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
    """Runs in a child process (bounded by a timeout in the parent).  SURVEY.md 8(d) protocol: 2 warm-up iterations, then
    >= 5 timed ones, median; an iteration = one scene of the bench batch, forward + loss + backward."""
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
        t = time.perf_counter()
        p = [torch.from_numpy(sc.points)]
        s = [torch.from_numpy(sc.superpoints)]
        feats, _ = det.extract_feat(p, s)
        out = det.decoder(feats, det.sp_centers(p, s), ['scannet'])
        inst = oc.gt_from_scene(p[0][:, :3] - p[0][:, :3].min(0)[0], torch.from_numpy(sc.instance_mask),
                                torch.from_numpy(sc.labels), s[0])
        loss = oc.criterion(out, [inst])
        det.zero_grad()
        loss.backward()
        return time.perf_counter() - t

    scenes = [make_scene(i, n_points=points) for i in range(8)]  # the bench batch
    warm = [run(scenes[6]), run(scenes[7])]
    times, t0, k = [], time.perf_counter(), 0
    while k < 6 and (k < 5 or time.perf_counter() - t0 < 14.0):
        times.append(run(scenes[k]))
        k += 1
    times.sort()
    print(json.dumps({'median_s': times[len(times) // 2], 'times': times, 'warm': warm, 'seconds': time.perf_counter() - t0}))


def cpu_baseline(points: int, voxel_size: float):
    """The oracle (a PyTorch-CPU restatement of the algorithm spconv's CPU path uses) timed on this box's host cores on a
    bounded sample of the bench batch: 2 warm-up scenes, then >= 5 timed scenes (forward + loss + backward each), median."""
    import subprocess
    cores = usable_cores()
    for pts, limit in ((points, 200), (max(points // 5, 2000), 90)):
        code = f'import sys; sys.path.insert(0, {ROOT!r}); import bench; bench._cpu_baseline_child({pts}, {voxel_size}, {cores})'
        try:
            res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=limit,
                                 env=dict(os.environ, OMP_NUM_THREADS=str(cores), HIP_VISIBLE_DEVICES=''))
            rec = json.loads(res.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001  (timeout / parse error -> try the smaller sample)
            log(f'cpu_baseline sample of {pts} pts failed: {type(e).__name__}')
            continue
        return dict(value=1.0 / rec['median_s'], unit='scenes/s', cores=cores, kind='port',
                    sample=f'median of {len(rec["times"])} timed scenes after 2 warm-up scenes ({pts} pts, {voxel_size} m voxels; scenes of the '
                           f'bench batch, fwd+loss+bwd each, fp32), {rec["seconds"]:.1f} s of timed CPU work with torch CPU threads={cores}; '
                           f'per-scene seconds min/median/max = {rec["times"][0]:.2f}/{rec["median_s"]:.2f}/{rec["times"][-1]:.2f}' +
                           ('' if pts == points else f' (reduced from {points} pts to stay inside the time bound)'))
    return dict(value=None, unit='scenes/s', cores=cores, kind='port', sample='timed out')


def log(msg: str):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


FAMILY_NAMES = ('conv_gmm', 'conv_wgrad', 'attn_fwd', 'attn_bwd', 'gemm')
# bench.py --config cfg4: (dataset, points) of the 8 scenes of one GPU's batch of the joint six-dataset config
CFG4_SCENES = [('scannet', 100_000), ('arkitscenes', 100_000), ('s3dis', 200_000), ('multiscan', 100_000), ('3rscan', 100_000),
               ('scannetpp', 180_000), ('scannet', 100_000), ('arkitscenes', 100_000)]


# A PMC file describes every kernel of the step, so it is only quoted when it was taken on exactly THIS library: every .hip / .h under
# unidet3d_amd/csrc (round 4 stamped the sparse-convolution sources only, and the file went on listing GEMM kernels that no longer
# existed -- VERDICT r4 weak #12)
def _all_csrc():
    import glob
    d = os.path.join(ROOT, 'unidet3d_amd', 'csrc')
    return tuple(sorted(os.path.basename(f) for f in glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h'))))


PMC_SOURCES = _all_csrc()


def csrc_hashes(names=None):
    """sha256/16 of kernel sources (default: every .hip / .h under unidet3d_amd/csrc) -- the stamp tools/pmc_bench.sh writes into a
    PMC summary and ``_pmc_traffic`` checks before quoting it."""
    import glob
    import hashlib
    d = os.path.join(ROOT, 'unidet3d_amd', 'csrc')
    if names is None:
        names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h')))
    return {n: hashlib.sha256(open(os.path.join(d, n), 'rb').read()).hexdigest()[:16] for n in names}


def tree_hash() -> str:
    """sha256/16 over (relative path, content hash) of every source file that decides a measurement: unidet3d_amd/**.{py,hip,h,json},
    include/u3d.h and this file -- a stamp the shipped tree computes itself, on the GPU box and in any checkout alike (VERDICT r5 weak
    #6: `.git_head` is written at the builder's last GPU run and goes stale with the next commit).  `python bench.py --tree-hash`."""
    import glob
    import hashlib
    files = [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, 'include', 'u3d.h')]
    for ext in ('py', 'hip', 'h', 'json'):
        files += glob.glob(os.path.join(ROOT, 'unidet3d_amd', '**', f'*.{ext}'), recursive=True)
    h = hashlib.sha256()
    for f in sorted(set(files)):
        h.update(os.path.relpath(f, ROOT).replace(os.sep, '/').encode())
        h.update(hashlib.sha256(open(f, 'rb').read()).digest())
    return h.hexdigest()[:16]


def _bf16_act() -> bool:
    """under bf16 operands the decoder keeps bf16 tensors in HBM (unidet3d_amd.precision.bf16_act's switch)"""
    from unidet3d_amd import precision as P
    return bool(P._BF16_ACT)


def git_head() -> str:
    """Commit of this tree: `git rev-parse` where .git exists (the build container), else the .git_head file tools/grun.sh writes
    into the snapshot that travels to the GPU box."""
    import subprocess
    try:
        r = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short=12', 'HEAD'], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return r.stdout.strip()
    except Exception:  # noqa: BLE001
        pass
    try:
        return open(os.path.join(ROOT, '.git_head')).read().strip() or '?'
    except OSError:
        return '?'


def _pmc_traffic(bf: bool, workload: str = 'cfg2'):
    """HBM bytes per sparse-convolution (forward / input-gradient) launch from the newest committed PMC pass of this command whose
    source hashes match the library timed here (tools/pmc_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    FETCH x2 on gfx950; a cfg2 pass and a cfg3 = --dtype bf16 pass).  A file taken on other sources is refused, not quoted."""
    import glob
    if workload != 'cfg2':       # the PMC passes are passes of the cfg2 / cfg3 commands: their per-launch bytes do not describe another workload's launches
        return None, 'PMC traffic is collected for the cfg2 / cfg3 commands only (tools/pmc_bench.sh)'
    want = csrc_hashes(PMC_SOURCES)
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*_pmc_traffic.json')), reverse=True)
    seen_other = 0
    for f in files:
        rec = json.load(open(f))
        have = rec.get('_meta', {}).get('csrc_sha16', {})
        blk = rec.get('cfg3', {}) if bf else rec
        if all(have.get(n) == h for n, h in want.items()) and '_spconv_gmm_all' in blk:
            return (blk['_spconv_gmm_all']['hbm_MB_per_launch'] * 1e6,
                    f"profiles/{os.path.basename(f)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of the {'cfg3' if bf else 'cfg2'} command at "
                    f"commit {rec['_meta'].get('git_head', '?')}; sha256/16 of every source under unidet3d_amd/csrc matches the library timed here)")
        seen_other += 1
    if bf and not seen_other:
        return None, None
    return None, (f'{len(files)} PMC file(s) under profiles/ were measured on different kernel sources (or hold no pass of this command): not reported' if files else None)


def measure(args, dtype: str, batch: int, rank: int, world: int, dev, fp32_math=None):
    """W untimed + K timed steps of one configuration (model, optimizer, scenes built here); returns the fields of the JSON line
    that describe it.  Collective when world > 1 (every rank calls it with the same arguments)."""
    from unidet3d_amd import _lib as L
    from unidet3d_amd import account, precision
    from unidet3d_amd import sparse
    from unidet3d_amd.config import build_model, scannet_model_cfg
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.dist import FlatGradBucket, broadcast_params
    from unidet3d_amd.synthetic import make_scene

    precision.set_operand_dtype(dtype)
    # weight gradients (sparse conv + Linear) as a stream chain of their own, joined when backward ends: the training loop's choice,
    # like the front-end prefetch; same kernels, bit-identical gradients (tests/test_gpu_gradients.py).  Same-box A/B: +2.7 % (r5)
    sparse.set_wgrad_overlap(2 if getattr(args, 'wgrad_overlap', None) is None else args.wgrad_overlap)
    if fp32_math or getattr(args, 'fp32_math', None):
        precision.set_fp32_math(fp32_math or args.fp32_math)
    x3 = dtype == 'fp32' and precision.get_fp32_math() == 'bf16x3'
    torch.manual_seed(0)
    wl = getattr(args, 'config', 'cfg2')
    if wl == 'cfg4':
        from unidet3d_amd.config import joint_model_cfg
        model = build_model(joint_model_cfg()).to(dev)
    else:
        model = build_model(scannet_model_cfg(voxel_size=args.voxel_size)).to(dev)
    model.train()
    broadcast_params(model)
    params = [p for p in model.parameters() if p.requires_grad]
    bucket = FlatGradBucket(params, attach=False)
    if world > 1:
        bucket.enable_overlap()              # 16 MB buckets, all-reduced over RCCL as backward completes them (a collective
                                             # call: every rank takes the same path, a failure is fatal on all of them)
    if args.optimizer == 'sgd':
        opt = torch.optim.SGD(params, lr=1e-3)
    else:
        opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.05, fused=True)      # configs/unidet3d_1xb8_scannet.py:712

    log(f'rank {rank}/{world}: [{dtype}] model built, generating {batch} scenes ({wl})')
    if wl == 'cfg4':
        # 100k - 200k points per scene by dataset; the scene's surface grows with its point count so that the
        # voxel density stays ScanNet's (S3DIS rooms and ScanNet++ laser scans are the large ones)
        from unidet3d_amd.config import joint_model_cfg
        from unidet3d_amd.data import make_joint_batch
        specs = [(n, p, p / 100_000) for n, p in CFG4_SCENES][:batch]
        scenes, _names, _gtb, inputs, samples = make_joint_batch(joint_model_cfg(), specs, dev, seed0=200 + rank * batch)
    elif wl == 'cfg5':
        scenes = [make_scene(500 + rank * batch + i, n_points=1_000_000, area_scale=10.0, n_furniture=40) for i in range(batch)]
        inputs, samples = make_batch_inputs(scenes, dev)
    else:
        scenes = [make_scene(rank * batch + i, n_points=args.points) for i in range(batch)]
        inputs, samples = make_batch_inputs(scenes, dev)                          # resident in HBM before timing
    n_points_total = int(sum(len(sc.points) for sc in scenes))

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
    log(f'[{dtype}] inputs resident, warm-up')
    warm_losses = []
    for i in range(max(args.warmup, 1)):
        loss = step()
        torch.cuda.synchronize()
        warm_losses.append(float(loss.detach()))
        log(f'[{dtype}] warm-up step {i} done, loss {warm_losses[-1]:.4f}')
    assert bucket.check_views(), 'p.grad does not alias the flat gradient buffer'
    families = tuple(zip(FAMILY_NAMES, (L.K_CONV_FWD, L.K_CONV_WGRAD, L.K_ATTN_FWD, L.K_ATTN_BWD, L.K_GEMM)))
    # ---- the timed region: exactly K steps, nothing instrumented -------------------------------------------------------
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    log(f'[{dtype}] timed region done: {dt / args.steps * 1e3:.2f} ms/step')
    # ---- per-family kernel times: HIP events around every launch of a family, on the stream it runs on, over extra steps
    # OUTSIDE the timed region (recording ~300 event pairs per step costs ~1 ms/step of host time) ----------------------
    prof_steps = max(1, min(args.steps, 5))
    overlap_was = sparse.set_wgrad_overlap(0)       # per-family times are those of kernels running alone (overlapped kernels stretch each other)
    for _, c in families:
        L.prof_enable(c, True)
    account.reset()
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof = {}
    acc = account.snapshot()
    for name, c in families:
        ms, n, work = L.prof_collect(c)
        L.prof_enable(c, False)
        prof[name] = dict(ms=ms, launches=n, flops=work, bytes=acc.get(name, {}).get('bytes', 0.0),
                          flops_booked=acc.get(name, {}).get('flops', 0.0))
    sparse.set_profile_flops(False)
    sparse.set_wgrad_overlap(overlap_was)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    loss_val = float(loss.item())
    n_vox = int(model._vb.coords.shape[0])
    inflight.clear()
    del model, opt, bucket, params, inputs, samples
    torch.cuda.empty_cache()

    bf = dtype == 'bf16'
    # which MFMA peak prices a family = the matrix instruction its kernels ISSUE.  bf16 operands (cfg3): every family multiplies on
    # v_mfma_f32_16x16x32_bf16 -- sparse conv forward / input gradient, the weight gradient (u3d_spconv_wgrad_rows on the bf16 shadows, or
    # u3d_spconv_wgrad_bf16), Linear fwd / dX / dW and attention; only the 16-channel input convolution stays on fp32 MFMAs (round 4
    # priced conv_wgrad against the fp32 peak and reported 0.82 -- VERDICT r4 weak #11).  fp32: native fp32 MFMAs, or see below.
    peak = {k: (PEAK_BF16_MFMA_TFLOPS if bf else PEAK_F32_MFMA_TFLOPS) for k in prof}
    mfma_instr = {k: ('v_mfma_f32_16x16x32_bf16' if bf else
                      ('v_mfma_f32_16x16x4_f32 (below 160 channels) / 6 x v_mfma_f32_16x16x32_bf16' if k == 'conv_wgrad' and x3 else
                       ('6 x v_mfma_f32_16x16x32_bf16 per product (three exact bf16 planes per operand)' if x3 else 'v_mfma_f32_*_f32')))
                  for k in prof}
    # fp32 math from three bf16 planes: the instructions that run are bf16 MFMAs, six per fp32-equivalent product, so the hardware
    # ceiling of those kernels is the bf16 dense peak / 6.  The roofline block's `frac` is priced against THAT ceiling (a correct
    # kernel can never exceed 1 there); the fraction of the dtype's own fp32 MFMA peak -- the figure rounds 1-3 quoted, comparable
    # across modes -- rides along as `frac_of_dtype_peak` (weight gradients below 160 channels and the 16-channel input convolution
    # stay on fp32 MFMAs)
    x3_ceiling = PEAK_BF16_MFMA_TFLOPS / 6.0
    kernels = {}
    for k, v in prof.items():
        t = v['ms'] * 1e-3
        tf = v['flops'] / t / 1e12 if t > 0 and v['flops'] > 0 else None
        gbs = v['bytes'] / t / 1e9 if t > 0 and v['bytes'] > 0 else None
        kernels[k] = {'ms_per_step': v['ms'] / prof_steps, 'launches_per_step': v['launches'] / prof_steps,
                      'algorithmic_gflop_per_step': v['flops'] / prof_steps / 1e9, 'algorithmic_MB_per_step': v['bytes'] / prof_steps / 1e6,
                      'tflops': tf, 'mfma_peak': peak[k], 'mfma_instr': mfma_instr[k],
                      'frac_mfma': (tf / peak[k] if tf else None),
                      **({'bf16x3_ceiling': x3_ceiling, 'frac_bf16x3_ceiling': tf / x3_ceiling} if x3 and tf and k != 'conv_wgrad' else {}),
                      'hbm_gbs': gbs, 'frac_hbm': gbs / PEAK_HBM_GBS if gbs else None}
    g = prof['conv_gmm']
    ach = g['flops'] / (g['ms'] * 1e-3) / 1e12 if g['ms'] > 0 else 0.0
    traffic, traffic_src = _pmc_traffic(bf, wl)
    fam_ms = sum(v['ms'] for v in prof.values())
    return {
        'value': batch * world * args.steps / dt,
        'unit': 'scenes/s',
        'ms_per_step': dt / args.steps * 1e3,
        'dtype': 'bf16' if bf else 'f32',
        'config': {'workload': ((f'cfg4 (BASELINE.json configs[3], one GPU\'s share): {batch} mixed synthetic scenes/GPU over the six datasets '
                                 f'({", ".join(f"{n} {p // 1000}k" for n, p in CFG4_SCENES[:batch])} pts), 0.02 m voxels, the joint '
                                 'unidet3d_1xb8_scannet_s3dis_multiscan_3rscan_scannetpp_arkitscenes model (7-dof head, rotated DIoU), ') if wl == 'cfg4' else
                                (f'cfg5 (BASELINE.json configs[4], one GPU\'s share): {batch} synthetic S3DIS-shape room/GPU x 1M pts, '
                                 f'{args.voxel_size} m voxels, unidet3d_1xb8_scannet model, ') if wl == 'cfg5' else
                                f'{"cfg3" if bf else "cfg2"}: {batch} synthetic ScanNet-shape scenes/GPU x {args.points} pts, '
                                f'{args.voxel_size} m voxels, unidet3d_1xb8_scannet model, ')
                               + ('bf16 MFMA operands (sparse conv fwd/dgrad/wgrad, Linear fwd/dX/dW, attention), fp32 accumulate/BN/softmax/optimizer; '
                                  + ('decoder activations (q/k/v, attention output, MLP hidden tensors, their gradients, LayerNorm copies) are bf16 '
                                     'tensors in HBM, residual stream / LayerNorm / parameters fp32 (the data flow of the reference\'s autocast); '
                                     if _bf16_act() else 'fp32 tensors rounded in flight (U3D_BF16_ACT=0); ')
                                  if bf else ('fp32 (products from three exact bf16 planes per operand on the bf16 matrix pipe, fp32 accumulation, '
                                              'fp32-level error; --fp32-math mfma = native fp32 MFMAs); ' if x3 else 'fp32 (native fp32 MFMAs); ')) +
                               'step = voxelise+rulebook+fwd+loss+bwd' + ('' if args.no_optimizer else f'+clip+{args.optimizer}')
                               + ('' if args.no_prefetch else "; each step's voxelise+rulebook part is queued on a side stream during "
                                  "the previous step's backward")
                               + ({0: '', 1: "; conv weight gradients on a side stream next to each layer's input gradient",
                                   2: '; weight-gradient kernels (sparse conv + Linear) run as a side-stream chain joined at the end of backward'}[overlap_was]),
                   'front_prefetch': not args.no_prefetch, 'wgrad_overlap': overlap_was, 'fp32_math': ('bf16x3' if x3 else 'mfma') if not bf else None,
                   'bf16_activations': _bf16_act() if bf else None,
                   'global_batch': batch * world, 'points_per_scene': args.points if wl == 'cfg2' else n_points_total // max(batch, 1),
                   'points_per_gpu': n_points_total,
                   'active_voxels_per_gpu': n_vox, 'parallelism': f'dp{world}', 'loss': loss_val, 'warmup_losses': warm_losses},
        'roofline': {'kernel': 'spconv_gmm_k (sparse conv forward + input-gradient, all levels)',
                     'bound': 'mfma', 'achieved': ach, 'peak': x3_ceiling if x3 else peak['conv_gmm'], 'unit': 'TFLOP/s',
                     'frac': ach / (x3_ceiling if x3 else peak['conv_gmm']),
                     **({'math': 'bf16x3: six v_mfma_f32_16x16x32_bf16 per fp32-equivalent product; peak = the ceiling of the instructions that '
                                 'run (bf16 dense MFMA peak / 6); dtype_peak = the fp32 dense MFMA peak (what rounds 1-3 priced frac against)',
                         'dtype_peak': peak['conv_gmm'], 'frac_of_dtype_peak': ach / peak['conv_gmm']} if x3 else {}),
                     'traffic': traffic, 'traffic_unit': 'bytes/launch (HBM, PMC)',
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
            'families_ms_per_step': fam_ms / prof_steps,
            'tflops': sum(v['flops'] for v in prof.values()) / max(fam_ms * 1e-3, 1e-12) / 1e12,
            'hbm_gbs': sum(v['bytes'] for v in prof.values()) / max(fam_ms * 1e-3, 1e-12) / 1e9,
            'share_of_step': (fam_ms / prof_steps) / (dt / args.steps * 1e3)},
    }


def main():
    args = parse()
    if args.tree_hash:
        print(tree_hash())
        return
    self_launch(args)                                  # N > 1 outside torchrun: does not return
    from unidet3d_amd import _lib as L
    from unidet3d_amd.dist import init_from_env

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the product path has no CPU fallback')
    L.lib()                                           # fail loudly if the HIP library is missing
    n_dev = torch.cuda.device_count()
    if args.backend == 'nccl' and args.gpus > n_dev:
        raise SystemExit(f'--gpus {args.gpus} with the RCCL backend needs {args.gpus} GPUs, this node shows {n_dev} '
                         "(--backend gloo lets ranks share a GPU to exercise the code path)")
    local_dev = int(os.environ.get('LOCAL_RANK', '0')) % n_dev
    torch.cuda.set_device(local_dev)
    rank, world, local = init_from_env(args.backend)
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    dev = torch.device('cuda', local_dev)

    head_batch = args.batch if args.batch is not None else (1 if args.config == 'cfg5' else (16 if args.dtype == 'bf16' else 8))
    if args.config != 'cfg2':      # the extra blocks (native MFMAs, cfg3, cfg4, cfg5, CPU baseline) belong to the headline line
        args.no_mfma_line = args.no_cfg3 = args.no_cpu_baseline = args.no_extra_configs = True
    head = measure(args, args.dtype, head_batch, rank, world, dev)
    # The two extra blocks below belong to the single-GPU line only: at N > 1 every further configuration builds a second model,
    # gradient bucket and communicator inside the same job -- measured with 2 gloo ranks: an ~11 s one-time stall lands somewhere in
    # the second configuration's first steps (3.7 s/step "timed") -- and the scaling runs need the headline only.
    native = None
    if world == 1 and args.dtype == 'fp32' and head['config'].get('fp32_math') == 'bf16x3' and not args.no_mfma_line:
        # the same workload, steps and protocol on the native fp32 MFMA kernels: the headline forms its fp32 products from bf16 pieces
        #  -- a reader who wants the number of the plain fp32 matrix instructions finds it in the same line
        native = measure(args, 'fp32', head_batch, rank, world, dev, fp32_math='mfma')
        from unidet3d_amd import precision as _P
        _P.set_fp32_math('bf16x3')
    cfg3 = None
    if world == 1 and args.dtype == 'fp32' and not args.no_cfg3:
        # BASELINE.json configs[2] right behind the headline, same process, same protocol (W warm-up + K timed steps)
        cfg3 = measure(args, 'bf16', 16, rank, world, dev)

    extra = {}
    if world == 1 and args.dtype == 'fp32' and not args.no_extra_configs:
        # BASELINE.json configs[3] / [4], one GPU's share each, behind the headline in the same process so that the driver's default run
        # measures all four GPU configurations.  Same protocol with their own, smaller step counts (stated in the block)
        import copy as _copy
        for name, b in (('cfg4', 8), ('cfg5', 1)):
            a2 = _copy.copy(args)
            a2.config, a2.steps, a2.warmup = name, min(args.steps, 10), min(args.warmup, 2)
            blk = measure(a2, 'fp32', b, rank, world, dev)
            extra[name] = dict(note=f"BASELINE.json configs[{3 if name == 'cfg4' else 4}], one GPU's share, measured after the headline in the same process: "
                                    f"{a2.warmup} warm-up + {a2.steps} timed steps, fp32 (bf16x3 products); PMC traffic is collected for the cfg2 / cfg3 commands only",
                               steps=a2.steps, warmup=a2.warmup, **blk)

    if rank == 0:
        out = {'metric': METRIC, 'value': head['value'], 'unit': head['unit'], 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': head['dtype'], 'data': 'synthetic', 'config': head['config'], 'roofline': head['roofline'],
               'kernels': head['kernels'], 'step_roofline': head['step_roofline']}
        if native is not None:
            out['fp32_native_mfma'] = dict(note='same workload / steps / protocol with U3D_FP32_MATH=mfma (v_mfma_f32_* instead of six bf16 MFMAs per '
                                                'product), measured right after the headline in the same process',
                                           value=native['value'], unit=native['unit'], ms_per_step=native['ms_per_step'],
                                           roofline=native['roofline'], kernels=native['kernels'],
                                           warmup_losses=native['config']['warmup_losses'])
        if cfg3 is not None:
            out['cfg3'] = dict(note='BASELINE.json configs[2] measured after the headline in the same process: bf16 MFMA operands, 16 scenes/GPU; '
                                    'families priced against the bf16 dense MFMA peak where their operands are bf16', **cfg3)
        for name, blk in extra.items():
            out[name] = blk
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.points, args.voxel_size)
        out['git_head'] = git_head()
        out['tree_hash'] = tree_hash()
        # the line is long and a log keeps its TAIL: the few numbers a reader wants first are repeated here, last
        out['summary'] = {'scenes_per_s': round(head['value'], 2), 'ms_per_step': round(head['ms_per_step'], 3), 'workload': args.config,
                          'dtype': head['dtype'], 'n_gpus': world,
                          'roofline_frac': round(head['roofline']['frac'], 4), 'roofline_frac_of_dtype_peak': round(head['roofline'].get('frac_of_dtype_peak', head['roofline']['frac']), 4),
                          'roofline_traffic_bytes_per_launch': head['roofline']['traffic'],
                          'fp32_native_mfma_scenes_per_s': round(native['value'], 2) if native is not None else None,
                          'cfg3_bf16_scenes_per_s': round(cfg3['value'], 2) if cfg3 is not None else None,
                          'cfg4_joint_scenes_per_s': round(extra['cfg4']['value'], 2) if 'cfg4' in extra else None,
                          'cfg5_1m_point_rooms_per_s': round(extra['cfg5']['value'], 2) if 'cfg5' in extra else None,
                          'tree_hash': out['tree_hash'],
                          'cpu_baseline_scenes_per_s': (out.get('cpu_baseline') or {}).get('value')}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
