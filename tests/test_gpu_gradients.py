"""Training-dynamics parity on the GPU: a multi-step SGD trajectory of the product against the fp32 CPU
oracle, and the weight-pack cache against optimizers that do not bump ``Tensor._version``.

Background: round 2's bench trajectory differed from round 1's from step 1 on (14.3895 -> 14.4689) although
init, data and optimizer were the same.  Cause: ``WeightPacks`` reused the MFMA-order copies of the convolution weights while
``(data_ptr, _version)`` of the parameters was unchanged, and ``torch.optim.AdamW(fused=True)`` updates parameters WITHOUT bumping
``_version`` -- every step after the first ran the backbone's forward / input-gradient convolutions on the step-0 weights.
The tests below fail on that bug."""
import copy

import pytest
import torch

import _parity as PA
from _detw import fill_state_dict

pytestmark = pytest.mark.gpu
DEV = PA.DEV


def _small_cfg():
    from unidet3d_amd.config import scannet_model_cfg
    cfg = scannet_model_cfg(voxel_size=0.05)
    return cfg


def test_sgd_trajectory_matches_the_fp32_oracle():
    """cfg1 (one 10 k-point scene, 5 cm voxels, the full 6-layer model): 5 plain-SGD steps on the same scene, product on the GPU
    vs the oracle on the CPU, identical initial weights.  The loss of every step must agree to 1e-4 relative (measured: logged),
    and the run must be a real optimisation (the loss falls by > 5 % over the 5 steps;
    the fp32 and fp64 CPU oracles stay within 6e-7 of each other on this trajectory), so a backward kernel with a systematic
    error, a stale weight copy or a missed parameter shows up as a diverging trajectory.  SGD, not Adam: Adam's first updates are
    lr * sign(g), which amplifies rounding-level differences of near-zero gradient entries into different trajectories."""
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    cfg = _small_cfg()
    prod, orac = PA.build_pair(cfg)
    scenes = [make_scene(40, n_points=10_000)]
    inputs, samples0 = make_batch_inputs(scenes, DEV)
    lr = 3e-4          # the loss falls by 10 % in 5 steps; at 2e-3 (27 % in one step) the fp32 and fp64 CPU oracles already part by 5e-4
    po = torch.optim.SGD(prod.parameters(), lr=lr)
    oo = torch.optim.SGD(orac.parameters(), lr=lr)
    lp, lo = [], []
    for it in range(5):
        samples = copy.deepcopy(samples0)                    # loss() rewrites the GT boxes of its samples in the training frame
        po.zero_grad(set_to_none=True)
        loss = prod.loss(inputs, samples)['det_loss']
        loss.backward()
        po.step()
        lp.append(float(loss.detach()))
        oo.zero_grad(set_to_none=True)
        O = PA.oracle_forward(orac, scenes, ['scannet'])
        O['loss'].backward()
        oo.step()
        lo.append(float(O['loss'].detach()))
    rels = [abs(a - b) / abs(b) for a, b in zip(lp, lo)]
    PA.log_errors('sgd_trajectory_cfg1', dict(lr=lr, product=lp, oracle=lo, rel=rels))
    print('sgd trajectory product', lp, 'oracle', lo, 'rel', rels)
    assert lo[-1] < 0.95 * lo[0], lo                        # the steps are large enough to matter
    assert max(rels) < 1e-4, (lp, lo)
    # and the weights themselves after the 5 steps (fp32 vs fp64 CPU oracle: 4.2e-5 on the worst tensor)
    og = dict(orac.named_parameters())
    worst = max((PA.rel(p, og[k]), k) for k, p in prod.named_parameters())
    print('worst weight tensor after 5 steps', worst)
    assert worst[0] < 3e-4, worst


@pytest.mark.parametrize('fused', [True, False])
def test_convolutions_see_the_weights_an_optimizer_step_wrote(fused):
    """After ``optimizer.step()`` the next forward must use the NEW convolution weights, whatever the optimizer does to
    ``Tensor._version`` (``AdamW(fused=True)`` leaves it unchanged): the loss of the second step equals the loss
    of a freshly built model that loaded the updated state_dict (up to the 1-ulp run-to-run noise of a step) (and therefore packs its weights from scratch)."""
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd.config import build_model
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    cfg = _small_cfg()
    cfg['decoder']['num_layers'] = 2
    model = fill_state_dict(build_model(cfg), tag0=3300, scale=0.06).to(DEV).train()
    inputs, samples0 = make_batch_inputs([make_scene(41, n_points=9000)], DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.05, fused=fused)
    w_before = model.unet.blocks[0].conv_branch[2].weight.detach().clone()
    v_before = model.unet.blocks[0].conv_branch[2].weight._version
    l0 = model.loss(inputs, copy.deepcopy(samples0))['det_loss']
    l0.backward()
    opt.step()
    assert not torch.equal(w_before, model.unet.blocks[0].conv_branch[2].weight.detach())
    print('fused', fused, '_version before / after the step:', v_before, model.unet.blocks[0].conv_branch[2].weight._version)
    l1 = float(model.loss(inputs, copy.deepcopy(samples0))['det_loss'].detach())
    fresh = build_model(cfg).to(DEV).train()
    fresh.load_state_dict(model.state_dict(), strict=True)
    # (the first forward updated the running statistics of `model`; the fresh copy starts from them, the loss does not depend on them)
    l1_fresh = float(fresh.loss(inputs, copy.deepcopy(samples0))['det_loss'].detach())
    assert abs(l1 - l1_fresh) <= 2e-6 * abs(l1_fresh), (l1, l1_fresh, float(l0.detach()))      # run-to-run noise of a step: ~1 ulp of the loss
    assert abs(l1 - float(l0.detach())) > 1e-3              # lr 1e-2 moves the loss visibly: a stale forward would repeat l0


def test_eval_mode_packs_follow_data_writes_after_invalidate():
    """Eval mode reuses the packed weights while (data_ptr, _version) is unchanged; a write through ``.data`` (no version bump:
    EMA parameter swaps) needs ``invalidate_weight_packs()`` -- and the first eval forward after training repacks on its own."""
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd.config import build_model
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    cfg = _small_cfg()
    cfg['decoder']['num_layers'] = 1
    model = fill_state_dict(build_model(cfg), tag0=3400, scale=0.06).to(DEV)
    sc = make_scene(42, n_points=8000)
    inputs, samples = make_batch_inputs([sc], DEV)
    sp = torch.from_numpy(sc.superpoints).to(DEV)
    S = int(sc.superpoints.max()) + 1

    def run(m):
        """per-superpoint backbone features (eval mode: batch norm on the running statistics)"""
        with torch.no_grad():
            m.collate(inputs['points'])
            return m.extract_feat(m._sparse_input(1), sp, m._vb.inverse, [0, S])[0].clone()
    # training step with a fused optimizer, then eval: the eval forward must see the stepped weights
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, fused=True)
    model.loss(inputs, copy.deepcopy(samples))['det_loss'].backward()
    opt.step()
    model.eval()
    a = run(model)
    fresh = build_model(cfg).to(DEV).eval()
    fresh.load_state_dict(model.state_dict(), strict=True)
    b = run(fresh)
    assert PA.rel(a, b) < 1e-5
    assert PA.rel(a, run(model)) < 1e-6                  # unchanged weights, unchanged (data_ptr, _version): the packs are reused
    # .data write in eval mode
    w = model.unet.blocks[0].conv_branch[2].weight
    v = w._version
    w.data.mul_(1.5)
    assert w._version == v
    model.invalidate_weight_packs()
    c = run(model)
    assert PA.rel(a, c) > 1e-3, PA.rel(a, c)
    fresh.load_state_dict(model.state_dict(), strict=True)
    assert PA.rel(c, run(fresh)) < 1e-5


def test_eval_mode_fine_tuning_with_grads_set_to_none_sees_every_step():
    """a root kept in eval() (frozen batch norm) and stepped by ``AdamW(fused=True)`` in the STANDARD loop order --
    zero_grad(set_to_none=True) -> forward -> backward -> step -- has every ``.grad`` None and every ``_version`` unchanged when the
    next forward refreshes the packs; the refresh must repack all the same (it keys on autograd being enabled over trainable weights).
    The loss of step 3 equals that of a fresh model loaded from the stepped state_dict; a stale pack would repeat the step-1 backbone."""
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd.config import build_model
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    cfg = _small_cfg()
    cfg['decoder']['num_layers'] = 2
    model = fill_state_dict(build_model(cfg), tag0=3500, scale=0.06).to(DEV).eval()
    inputs, samples0 = make_batch_inputs([make_scene(43, n_points=9000)], DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.05, fused=True)
    losses = []
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        assert all(p.grad is None for p in model.parameters())
        loss = model.loss(inputs, copy.deepcopy(samples0))['det_loss']
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    opt.zero_grad(set_to_none=True)
    l2 = float(model.loss(inputs, copy.deepcopy(samples0))['det_loss'].detach())
    fresh = build_model(cfg).to(DEV).eval()
    fresh.load_state_dict(model.state_dict(), strict=True)
    l2_fresh = float(fresh.loss(inputs, copy.deepcopy(samples0))['det_loss'].detach())
    print('eval-mode fine-tuning losses', losses, l2, 'fresh', l2_fresh)
    assert abs(l2 - l2_fresh) <= 1e-5 * abs(l2_fresh), (losses, l2, l2_fresh)
    assert abs(losses[1] - losses[0]) > 1e-3 and abs(l2 - losses[1]) > 1e-3, (losses, l2)       # the steps are visible in the loss
    # inference afterwards (no autograd) reuses the packs: two forwards, same result, one pack launch at most
    with torch.no_grad():
        a = float(model.loss(inputs, copy.deepcopy(samples0))['det_loss'])
        b = float(model.loss(inputs, copy.deepcopy(samples0))['det_loss'])
    assert abs(a - b) <= 2e-6 * abs(a), (a, b)


@pytest.mark.parametrize('operands', ['fp32', 'bf16'])
def test_training_step_is_bit_reproducible(operands):
    """Run-to-run determinism: the same training step -- fresh model from the same weights, same two scenes -- run
    three times gives the SAME BITS for the loss, every parameter gradient and the running statistics.  Rounds 1-4 did not: the
    superpoint CSR was filled through an atomic cursor, so the fp32 sums of the pooling kernel (and everything downstream of it,
    amplified by ~90 training-mode batch norms on the way back) changed in the last bits from run to run (tools/grad_trace.py;
    profiles/round5_determinism_*.json).  No kernel of the step uses floating-point atomics; every reduction has a fixed order."""
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd import precision as P
    from unidet3d_amd.config import build_model
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    cfg = _small_cfg()
    cfg['decoder']['num_layers'] = 2
    inputs, samples0 = make_batch_inputs([make_scene(70, n_points=8000), make_scene(71, n_points=8000)], DEV)
    runs = []
    for _ in range(3):
        model = fill_state_dict(build_model(cfg), tag0=3000, scale=0.06).to(DEV).train()
        with P.operands(operands):
            loss = model.loss(inputs, copy.deepcopy(samples0))['det_loss']
            loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                     {k: b.clone() for k, b in model.named_buffers()}))
    for r in runs[1:]:
        assert torch.equal(runs[0][0], r[0]), (float(runs[0][0]), float(r[0]))
        diff = [k for k, g in runs[0][1].items() if not torch.equal(g, r[1][k])]
        assert not diff, f'{len(diff)} gradient tensors differ between two runs of the same step, e.g. {diff[:5]}'
        assert all(torch.equal(b, r[2][k]) for k, b in runs[0][2].items())


@pytest.mark.parametrize('operands', ['fp32', 'bf16'])
def test_weight_gradients_on_the_side_stream_give_the_same_bits(operands):
    """sparse.set_wgrad_overlap(1 | 2): the sparse convolutions' weight gradients run on a side stream -- next to the layer's input
    gradient (1) or as a chain of their own that is only joined when the backward pass ends (2).  Same kernels, same inputs: loss and
    EVERY gradient must equal the single-stream step bit for bit (the step is bit-reproducible, so any difference is a race: a buffer
    recycled under a running kernel, a gradient read before its kernel finished).  Also with gradients ACCUMULATED into existing .grad
    tensors (mode 2 must then join before autograd adds), and through FlatGradBucket's copy."""
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd import precision as P
    from unidet3d_amd import sparse
    from unidet3d_amd.config import build_model
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.dist import FlatGradBucket
    from unidet3d_amd.synthetic import make_scene
    cfg = _small_cfg()
    cfg['decoder']['num_layers'] = 2
    inputs, samples0 = make_batch_inputs([make_scene(80, n_points=20000), make_scene(81, n_points=20000)], DEV)

    def run(mode, accumulate=False, bucket=False):
        prev = sparse.set_wgrad_overlap(mode)
        try:
            model = fill_state_dict(build_model(cfg), tag0=3000, scale=0.06).to(DEV).train()
            params = [p for p in model.parameters() if p.requires_grad]
            fb = FlatGradBucket(params, attach=False) if bucket else None
            if accumulate:
                for p in params:
                    p.grad = torch.full_like(p, 0.25)
            with P.operands(operands):
                loss = model.loss(inputs, copy.deepcopy(samples0))['det_loss']
                loss.backward()
            if fb is not None:
                fb.sync()
            torch.cuda.synchronize()
            return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        finally:
            sparse.set_wgrad_overlap(prev)

    ref_loss, ref = run(0)
    for mode in (1, 2, 2, 2):
        loss, g = run(mode)
        assert torch.equal(loss, ref_loss)
        bad = [k for k in ref if not torch.equal(ref[k], g[k])]
        assert not bad, (mode, len(bad), bad[:5])
    _, acc0 = run(0, accumulate=True)
    _, acc2 = run(2, accumulate=True)
    bad = [k for k in acc0 if not torch.equal(acc0[k], acc2[k])]
    assert not bad, ('accumulate', len(bad), bad[:5])
    assert any(not torch.equal(acc0[k], ref[k]) for k in ref)             # the 0.25 really was added to
    _, b2 = run(2, bucket=True)
    bad = [k for k in ref if not torch.equal(ref[k], b2[k])]
    assert not bad, ('bucket', len(bad), bad[:5])


def test_side_stream_weight_gradients_with_shared_weights_and_hooks():
    """in mode 2 a weight gradient may stay on the side stream only when autograd's AccumulateGrad is its sole consumer.
    (i) A Linear weight AND a sparse-convolution weight used by two layers of one graph (autograd sums the two dWs on the main stream),
    (ii) a parameter with a tensor hook that reads the gradient at once: gradients must equal the single-stream ones bit for bit."""
    from unidet3d_amd import dense, ops, sparse
    from unidet3d_amd.synthetic import make_scene
    torch.manual_seed(0)
    vb = ops.voxelize([torch.from_numpy(make_scene(60, n_points=30000).points).to(DEV)], 0.02, 128)
    rb = sparse.build_subm_rulebook(vb.coords, vb.index)
    n = vb.coords.shape[0]
    x0 = torch.randn(n, 32, device=DEV)
    wc0 = torch.randn(32, 3, 3, 3, 32, device=DEV) * 0.05
    wl0 = torch.randn(32, 32, device=DEV) * 0.1
    seen = []

    def run(mode, hook):
        prev = sparse.set_wgrad_overlap(mode)
        try:
            wc, wl = wc0.clone().requires_grad_(), wl0.clone().requires_grad_()
            if hook:
                wl.register_hook(lambda g: seen.append(float(g.abs().sum())) or g * 2.0)
            x = x0.clone().requires_grad_()
            h = sparse.sparse_conv(x, wc, rb)
            h = dense.linear(h, wl)
            h = sparse.sparse_conv(h, wc, rb)           # the same convolution weight a second time
            h = dense.linear(h, wl)                     # and the same Linear weight
            (h * h).sum().backward()
            torch.cuda.synchronize()
            return wc.grad.clone(), wl.grad.clone(), x.grad.clone()
        finally:
            sparse.set_wgrad_overlap(prev)

    for hook in (False, True):
        ref = run(0, hook)
        for _ in range(3):
            got = run(2, hook)
            for name, a, b in zip(('conv weight', 'linear weight', 'input'), ref, got):
                assert torch.equal(a, b), (name, hook, float((a - b).abs().max()))
    assert len(seen) == 4 and len(set(seen)) == 1          # the hook saw the same (summed) gradient in every run


def test_presplit_weight_planes_and_batched_transposes_change_no_bit():
    """dense.transposed_weights(): every Linear weight's [K, N] copy (one u3d_transpose_batch launch) and, for the three-plane fp32
    products, the three bf16 planes of every weight and transposed copy (one u3d_weight_planes_batch launch; the NT kernels then load
    their W operand pre-split, u3d_gemm_w_planes) -- against the same step with neither (per-launch transposes, in-kernel splits):
    loss and every gradient identical to the bit, in both fp32 math modes."""
    import unidet3d_amd  # noqa: F401
    from unidet3d_amd import dense
    from unidet3d_amd import precision as P
    from unidet3d_amd.config import build_model
    from unidet3d_amd.data import make_batch_inputs
    from unidet3d_amd.synthetic import make_scene
    cfg = _small_cfg()
    cfg['decoder']['num_layers'] = 2
    inputs, samples0 = make_batch_inputs([make_scene(90, n_points=9000), make_scene(91, n_points=7000)], DEV)

    def run(planes, batch):
        was = dense._W_PLANES
        dense._W_PLANES = planes
        model = fill_state_dict(build_model(cfg), tag0=3000, scale=0.06).to(DEV).train()
        ctx_cls = dense.transposed_weights
        if not batch:                                   # no context: every backward transposes for itself, no planes
            class _Off:
                def __init__(self, m): pass
                def __enter__(self): return self
                def __exit__(self, *a): return False
            dense.transposed_weights = _Off
        try:
            loss = model.loss(inputs, copy.deepcopy(samples0))['det_loss']
            loss.backward()
            torch.cuda.synchronize()
        finally:
            dense.transposed_weights = ctx_cls
            dense._W_PLANES = was
        return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    for math in ('bf16x3', 'mfma'):
        with P.fp32_math(math):
            ref_loss, ref = run(False, False)
            for planes, batch in ((True, True), (False, True)):
                loss, g = run(planes, batch)
                assert torch.equal(loss, ref_loss), (math, planes, batch, float(loss), float(ref_loss))
                bad = [k for k in ref if not torch.equal(ref[k], g[k])]
                assert not bad, (math, planes, batch, len(bad), bad[:5])
