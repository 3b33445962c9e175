"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU, exports every
symbol include/u3d.h declares, the ctypes table mirrors the header, and the product path refuses
to run without the HIP kernels (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'u3d.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(u3d_[a-z0-9_]+)\s*\(', src)) - {'u3d_stream_t'})


def test_library_exports_every_declared_symbol():
    from unidet3d_amd.csrc.build import build
    path = build(verbose=False)
    lib = ctypes.CDLL(path)
    names = _header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/u3d.h but not exported'
    lib.u3d_version.restype = ctypes.c_int
    from unidet3d_amd import _lib
    hdr = int(re.search(r'#define\s+U3D_ABI_VERSION\s+(\d+)', open(os.path.join(ROOT, 'include', 'u3d.h')).read()).group(1))
    assert lib.u3d_version() == hdr == _lib.ABI_VERSION


def test_stale_library_is_refused(monkeypatch):
    """a library built against another argument list must not load silently."""
    from unidet3d_amd import _lib
    _lib.lib()
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'ABI_VERSION', _lib.ABI_VERSION + 1)
    import pytest
    with pytest.raises(_lib.U3DError, match='ABI version'):
        _lib.lib()


def test_conv_kernel_switch_is_host_state():
    from unidet3d_amd import _lib
    l = _lib.lib()
    prev = l.u3d_conv_kernel(-1)
    try:
        assert prev in (0, 1)
        assert l.u3d_conv_kernel(0) == prev and l.u3d_conv_kernel(-1) == 0 and l.u3d_conv_kernel(5) == 0
        assert l.u3d_conv_kernel(1) == 0 and l.u3d_conv_kernel(-1) == 1
        assert l.u3d_conv_kernel(2) == 1 and l.u3d_conv_kernel(-1) == 2
    finally:
        l.u3d_conv_kernel(prev)


def test_ctypes_table_mirrors_header():
    from unidet3d_amd import _lib
    assert sorted(_lib.PROTOTYPES) == _header_symbols()
    src = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'u3d.h')).read(), flags=re.S)
    for name, (_, args) in _lib.PROTOTYPES.items():
        m = re.search(r'\b' + name + r'\s*\((.*?)\)\s*;', src, flags=re.S)
        assert m, name
        params = [p for p in m.group(1).split(',') if p.strip() and p.strip() != 'void']
        assert len(params) == len(args), f'{name}: header has {len(params)} params, ctypes table {len(args)}'


def test_pure_size_queries_run_without_gpu():
    from unidet3d_amd import _lib
    l = _lib.lib()
    assert l.u3d_index_words(2, 128, 130, 65) == 2 * 128 * 130 * 2
    R, G = ctypes.c_int(0), ctypes.c_int(0)
    assert l.u3d_spconv_plan(32, 32, 27, 400000, ctypes.byref(R), ctypes.byref(G)) == 0 and (R.value, G.value) == (64, 1)
    # a level that cannot fill the chip: 27 offsets -> 64-row tiles, nine offset groups; 8 offsets (strided / inverse) -> 32-row tiles
    assert l.u3d_spconv_plan(160, 160, 27, 1400, ctypes.byref(R), ctypes.byref(G)) == 0 and (R.value, G.value) == (64, 9)
    assert l.u3d_spconv_plan(128, 160, 8, 1400, ctypes.byref(R), ctypes.byref(G)) == 0 and (R.value, G.value) == (32, 1)
    assert l.u3d_spconv_plan(24, 32, 27, 1000, ctypes.byref(R), ctypes.byref(G)) < 0     # unsupported channel count is refused
    assert l.u3d_subm_rulebook_ws_bytes(1000) >= 2 * 27 * 4 * 4      # block counts + bases of 4 blocks (no dense neighbour matrix since round 3)
    assert l.u3d_down_rulebook_ws_bytes(1000) >= 8 * 1000 * 4


def test_gemm_tn_workspace_covers_every_kernel_plan():
    """u3d_gemm_tn writes one [N*K + N] block of partial sums per row split, grid padded to whole groups of 8 splits; the split
    count depends on the kernel that runs (fp32 MFMA tiles 128 / 64, bf16, three-plane 128x64 / 64x64 tiles with 32-row trips,
    gemm.hip tn_splits / tn_x3_splits).  The size query must cover the largest of them for the decoder's shapes and the edge cases."""
    from unidet3d_amd import _lib
    l = _lib.lib()

    def cdiv(a, b):
        return -(-a // b)

    def x3_splits(M, N, K):      # gemm.hip tn_x3_splits
        ta = 128 if cdiv(N, 128) * cdiv(K, 128) >= 16 else 64
        tiles = cdiv(N, ta) * cdiv(K, 64)
        return max(1, min(256, cdiv(768, tiles), cdiv(M, 4 * 32)))

    def fp32_splits(M, N, K, T):  # gemm.hip tn_splits (default target of 768 workgroups, 16-row steps)
        tiles = cdiv(N, T) * cdiv(K, T)
        return max(1, min(256, cdiv(768, tiles), cdiv(M, 4 * 16)))

    for M, N, K in [(12300, 1024, 256), (12300, 256, 1024), (12300, 768, 256), (12300, 256, 256), (24600, 256, 32), (2500, 20, 256),
                    (1, 256, 256), (333, 8, 256), (1_000_000, 256, 256)]:
        need = max(x3_splits(M, N, K), fp32_splits(M, N, K, 128), fp32_splits(M, N, K, 64))
        assert l.u3d_gemm_tn_ws_bytes(M, N, K) >= (need + 8) * (N * K + N) * 4, (M, N, K)


@pytest.mark.skipif(torch.cuda.is_available(), reason='CPU-only check')
def test_no_cpu_fallback():
    from unidet3d_amd import _lib, ops
    with pytest.raises(_lib.U3DError):
        ops.voxelize([torch.rand(10, 6)], 0.02, 128)
    with pytest.raises(_lib.U3DError):
        _lib.ptr(torch.zeros(4))


def test_index_words_bounds():
    """The direct-address occupancy index refuses extents it cannot hold instead of allocating (no GPU needed)."""
    from unidet3d_amd import _lib as L
    lib = L.lib()
    assert lib.u3d_index_words(8, 512, 512, 256) == 8 * 512 * 512 * 4
    assert lib.u3d_index_words(1, 128, 128, 65) == 128 * 128 * 2
    assert lib.u3d_index_words(0, 128, 128, 128) < 0
    assert lib.u3d_index_words(8, 100_000, 100_000, 4096) < 0 and b'extent too large' in lib.u3d_last_error()


def test_fp32_math_switch_is_host_state_and_drives_the_conv_format():
    """u3d_fp32_math is pure host state (no GPU needed): default = three-plane bf16 products unless U3D_FP32_MATH=mfma, any other
    argument only queries, and the Python layer picks the sparse-conv entry point / packed-weight layout from it
    (precision.conv_format; the packed layout belongs to the entry point, so the library does not switch it silently)."""
    from unidet3d_amd import _lib as L
    from unidet3d_amd import precision as P
    lib = L.lib()
    default = 0 if os.environ.get('U3D_FP32_MATH') == 'mfma' else 1
    prev = P.get_fp32_math()
    try:
        P.set_fp32_math('bf16x3' if default else 'mfma')
        assert lib.u3d_fp32_math(-1) == default and lib.u3d_fp32_math(7) == default       # queries
        with P.fp32_math('mfma'):
            assert lib.u3d_fp32_math(-1) == 0 and P.get_fp32_math() == 'mfma' and P.conv_format() == P.FMT_FP32
            with P.fp32_math('bf16x3'):
                assert lib.u3d_fp32_math(-1) == 1 and P.conv_format() == P.FMT_X3
                with P.operands('bf16'):
                    assert P.conv_format() == P.FMT_BF16                                   # bf16 operands win over the fp32 math mode
            assert P.get_fp32_math() == 'mfma'
        with pytest.raises(ValueError):
            P.set_fp32_math('tf32')
    finally:
        P.set_fp32_math(prev)
    from unidet3d_amd import sparse
    assert [sparse._pack_floats(96, f) for f in (0, 1, 2)] == [96, 48, 144]               # fp32 | bf16 | three bf16 planes
