"""Host logic of the bf16 shadows (precision.bf16_rows) that needs no GPU: the fragment-order layout the batch-norm
kernels write (csrc/bn.hip store_shadow; restated by sparse.to_shadow), how a shadow follows its tensor, and the mode switches."""
import torch


def test_to_shadow_is_the_documented_fragment_order():
    from unidet3d_amd import sparse
    n, C = 5, 96
    t = torch.arange(n * C, dtype=torch.float32).reshape(n, C) * 0.25
    sh = sparse.to_shadow(t)
    assert sh.dtype == torch.bfloat16 and sh.shape == (n, C) and sh.is_contiguous()
    for r in (0, 4):
        for p in range(C):                                   # position p of a row holds channel (p & ~31) | ((p >> 2) & 1) << 4 | ((p >> 3) & 3) << 2 | (p & 3)
            ch = (p & ~31) | (((p >> 2) & 1) << 4) | (((p >> 3) & 3) << 2) | (p & 3)
            assert float(sh[r, p]) == float(t[r, ch].to(torch.bfloat16)), (r, p, ch)
    # the 16 bytes at byte 16 q of a 32-channel group: channels 4q..4q+3 and 16+4q..16+4q+3 (what MFMA lane group q multiplies)
    g = sparse.to_shadow(torch.arange(32, dtype=torch.float32)[None])[0].float().tolist()
    assert g[8:16] == [4, 5, 6, 7, 20, 21, 22, 23]


def test_shadow_follows_its_tensor_and_is_dropped_when_the_tensor_changes():
    from unidet3d_amd import sparse
    t = torch.randn(7, 32)
    assert sparse.shadow_of(t) is None
    sparse.attach_shadow(t, sparse.to_shadow(t))
    assert sparse.shadow_of(t) is not None
    u = t                                                    # the same tensor object through another name (what autograd hands on)
    assert sparse.shadow_of(u) is sparse.shadow_of(t)
    assert sparse.shadow_of(t + 0) is None and sparse.shadow_of(t.clone()) is None        # a new tensor (e.g. an accumulated gradient) has none
    t.mul_(2.0)                                              # in-place change: the shadow no longer describes the tensor
    assert sparse.shadow_of(t) is None


def test_bf16_rows_follow_the_operand_mode():
    from unidet3d_amd import precision as P
    assert not P.bf16_rows()                                 # fp32 operands: never
    with P.operands('bf16'):
        assert P.bf16_rows() == P._BF16_ROWS
        with P.bf16_rows_mode(False):
            assert not P.bf16_rows()
        with P.bf16_rows_mode(True):
            assert P.bf16_rows()
    assert not P.bf16_rows()
