"""N>1 path on the GPU box: two gloo ranks share cuda:0 (RCCL needs one GPU per rank; the box has one),
one scene per rank, SyncBatchNorm statistics all-reduced between the HIP kernels, gradients packed and
all-reduced by FlatGradBucket.  The averaged gradients must equal those of ONE process running both
scenes as a batch -- the property the reference gets from DDP + nn.SyncBatchNorm."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _build():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _detw import fill_state_dict
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd.config import build_model, scannet_model_cfg
    cfg = scannet_model_cfg(voxel_size=0.05)
    cfg['decoder']['num_layers'] = 2
    return fill_state_dict(build_model(cfg), tag0=3000, scale=0.06).to('cuda:0').train()


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    torch.cuda.set_device(0)
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.dist import FlatGradBucket, broadcast_params, init_from_env
    from unidet3d_amd.synthetic import make_scene
    init_from_env('gloo')
    from unidet3d_amd import sparse
    sparse.set_wgrad_overlap(2)        # the bench's setting: weight gradients on their own stream chain; the bucket hooks must join it
    model = _build()
    broadcast_params(model)
    params = [p for p in model.parameters() if p.requires_grad]
    bucket = FlatGradBucket(params, attach=False).enable_overlap(bucket_bytes=8 << 20)      # the bench's exchange: buckets reduced during backward
    assert len(bucket.buckets) >= 4
    inputs, samples = make_batch_inputs([make_scene(70 + rank, n_points=8000)], 'cuda:0')
    bucket.clear_grads()
    loss = model.loss(inputs, samples)['det_loss']
    loss.backward()
    bucket.finish()
    assert bucket.check_views()
    if rank == 0:
        torch.save(dict(flat=bucket.flat.cpu(), rm=model.output_layer[0].running_mean.cpu(),
                        rv=model.unet.u.u.blocks[0].conv_branch[0].running_var.cpu()), out_path)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_equal_one_process_with_both_scenes():
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    out_path = os.path.join(tempfile.mkdtemp(), 'ddp.pt')
    port = _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out_path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = torch.load(out_path)
    model = _build()
    inputs, samples = make_batch_inputs([make_scene(70, n_points=8000), make_scene(71, n_points=8000)], 'cuda:0')
    loss = model.loss(inputs, samples)['det_loss']
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad]).cpu()
    n_dec = sum(p.numel() for p in model.decoder.parameters())            # the decoder's parameters come last in the flat buffer
    rel_dec = float((got['flat'][-n_dec:] - ref[-n_dec:]).abs().max() / ref[-n_dec:].abs().max())
    rel_bb = float((got['flat'][:-n_dec] - ref[:-n_dec]).abs().max() / ref[:-n_dec].abs().max())
    import _parity as PA
    PA.log_errors('ddp_2ranks_vs_1process', dict(decoder_grad_rel=rel_dec, backbone_grad_rel=rel_bb))
    # the two ranks' partial batch-norm sums are fp64 and every other kernel sees the same rows: measured 1.3e-7 (round 3)
    assert rel_dec < 1e-4, rel_dec
    assert rel_bb < 1e-4, rel_bb
    # synchronized statistics: running stats after one step equal the single-process ones
    assert torch.allclose(got['rm'], model.output_layer[0].running_mean.cpu(), rtol=1e-3, atol=1e-5)
    assert torch.allclose(got['rv'], model.unet.u.u.blocks[0].conv_branch[0].running_var.cpu(), rtol=1e-3, atol=1e-5)


def _rccl_worker(port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    torch.cuda.set_device(0)
    import torch.distributed as dist
    from unidet3d_amd import dist as D
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    D.init_from_env('nccl', force=True)                  # a ONE-rank RCCL communicator on cuda:0
    assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    calls = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **kw):
        calls.append((t.dtype, t.numel(), kw.get('group') is not None, t.is_cuda))
        return real_all_reduce(t, *a, **kw)

    import copy
    inputs, samples = make_batch_inputs([make_scene(70, n_points=8000), make_scene(71, n_points=8000)], 'cuda:0')
    res = {}
    for forced in (True, False):
        D.force_collectives(forced)
        dist.all_reduce = counting_all_reduce
        try:
            model = _build()
            D.broadcast_params(model)
            params = [p for p in model.parameters() if p.requires_grad]
            bucket = D.FlatGradBucket(params, attach=False).enable_overlap(bucket_bytes=8 << 20)     # forced: dist.new_group() -> second communicator
            assert (bucket.group is not None) == forced
            bucket.clear_grads()
            loss = model.loss(inputs, samples)['det_loss']
            loss.backward()
            bucket.finish()
            torch.cuda.synchronize()
            res[forced] = dict(flat=bucket.flat.clone(), loss=loss.detach().clone(), rm=model.output_layer[0].running_mean.clone(),
                               rv=model.unet.u.u.blocks[0].conv_branch[0].running_var.clone(), n_buckets=len(bucket.buckets))
        finally:
            dist.all_reduce = real_all_reduce
        if forced:
            res['calls'] = list(calls)
        else:
            assert len(calls) == len(res['calls']), 'the non-distributed step issued a collective'
    D.force_collectives(False)
    # the same pair in bf16-operand mode: the exchange path's separate statistics / apply launches must write the bf16 shadows the
    # sparse convolutions gather (precision.bf16_rows) exactly like the fused call does
    from unidet3d_amd import precision as P
    from unidet3d_amd import sparse
    bres = {}
    for forced in (True, False):
        D.force_collectives(forced)
        model = _build()
        sparse.SHADOW_STATS.update(hit=0, miss=0)
        with P.operands('bf16'):
            loss = model.loss(inputs, copy.deepcopy(samples))['det_loss']
            loss.backward()
        torch.cuda.synchronize()
        bres[forced] = (loss.detach().clone(), torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]), dict(sparse.SHADOW_STATS))
    D.force_collectives(False)
    f64 = [c for c in res['calls'] if c[0] == torch.float64]
    f32 = [c for c in res['calls'] if c[0] == torch.float32]
    ok = dict(n_f64=len(f64), n_f32=len(f32), n_buckets=res[True]['n_buckets'],
              f32_on_own_group=all(c[2] for c in f32), all_cuda=all(c[3] for c in res['calls']),
              flat_equal=bool(torch.equal(res[True]['flat'], res[False]['flat'])), loss_equal=bool(torch.equal(res[True]['loss'], res[False]['loss'])),
              stats_equal=bool(torch.equal(res[True]['rm'], res[False]['rm']) and torch.equal(res[True]['rv'], res[False]['rv'])),
              loss_rel=float((res[True]['loss'] - res[False]['loss']).abs() / res[False]['loss'].abs()),
              max_rel=float((res[True]['flat'] - res[False]['flat']).abs().max() / res[False]['flat'].abs().max()),
              stats_rel=float((res[True]['rv'] - res[False]['rv']).abs().max() / res[False]['rv'].abs().max()),
              bf16_loss_rel=float((bres[True][0] - bres[False][0]).abs() / bres[False][0].abs()),
              bf16_grad_rel=float((bres[True][1] - bres[False][1]).abs().max() / bres[False][1].abs().max()),
              bf16_shadow_hits=int(bres[True][2]['hit']), bf16_shadow_hits_plain=int(bres[False][2]['hit']))
    torch.save(ok, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_one_rank_runs_the_whole_collective_sequence():
    """the data-parallel step's collectives have only ever met gloo.  One RCCL rank on cuda:0 with
    dist.force_collectives(): every SyncBatchNorm exchange (fp64 [sum x, sum x^2, n], forward and backward) and every gradient
    bucket (fp32, on the communicator dist.new_group() made for them, launched from the backward hooks) goes through librccl;
    a one-rank all-reduce is the identity, so gradients, loss and running statistics must equal the non-distributed step's to the
    rounding of the separate statistics / apply launches the exchange needs (bit equality is logged, never asserted)."""
    out_path = os.path.join(tempfile.mkdtemp(), 'rccl1.pt')
    ctx = mp.get_context('spawn')
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), out_path))
    p.start()
    p.join(timeout=600)
    assert p.exitcode == 0
    ok = torch.load(out_path)
    import _parity as PA
    PA.log_errors('rccl_one_rank_vs_no_group', {k: (float(v) if not isinstance(v, bool) else v) for k, v in ok.items()})
    full = '; '.join(f'{k}={v!r}' for k, v in sorted(ok.items()))          # the whole record, untruncated, in every assertion message
    assert ok['n_f64'] >= 40, full            # SyncBatchNorm layers x (forward + backward)
    assert ok['n_f32'] == ok['n_buckets'] >= 4 and ok['f32_on_own_group'] and ok['all_cuda'], full
    # The exchange path runs batch-norm statistics, finalize and apply as separate launches (fp64 sums handed to the collective), the
    # non-distributed path the fused call: two DIFFERENT launch sequences, each deterministic, whose results differ by fp32 rounding of
    # the normalised activations.  Bit equality of loss / gradients / statistics between them is therefore a coincidence of rounding
    # (seen true for the loss on some trees and false on others: it flipped when an unrelated GEMM tile rule changed the summation
    # order downstream) and is only LOGGED (flat_equal / loss_equal / stats_equal above).  Asserted: the same bounds the two-rank test
    # uses.  Measured on MI355X / RCCL 2.26.6 over rounds 4-5: loss 0 ... 1e-7, gradients 5.5e-8 ... 4.3e-6, statistics 1.1e-7.
    assert ok['loss_rel'] <= 1e-5, full
    assert ok['max_rel'] <= 1e-4, full
    assert ok['stats_rel'] <= 1e-5, full
    # bf16 operands: the shadows are written and used on the exchange path as on the fused one (same hit count), results agree to the
    # rounding of the separate launches seen through bf16 operands
    assert ok['bf16_shadow_hits'] == ok['bf16_shadow_hits_plain'] >= 80, full
    assert ok['bf16_loss_rel'] < 1e-3 and ok['bf16_grad_rel'] < 5e-2, full


def test_bench_gpus_2_launches_itself():
    """`python bench.py --gpus 2` exactly as the driver would call it for N > 1 -- no outer torchrun; gloo lets the two ranks share
    this box's single GPU -- must come back with ONE JSON line for n_gpus = 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1',
                          '--points', '20000', '--no-cpu-baseline', '--no-cfg3', '--no-mfma-line'], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 16
    assert d['steps'] == 2 and d['value'] > 0 and d['scaling'] == 'weak'
    assert abs(d['value'] - 16 / d['ms_per_step'] * 1e3) < 1e-6 * d['value']
