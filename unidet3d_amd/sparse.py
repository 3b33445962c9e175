"""Sparse-voxel tensor + layers over the gfx950 kernels (host side of the boundary).

Mirrors the subset of ``spconv.pytorch`` the reference uses
(unidet3d/spconv_unet.py:5-7,34-72,146-192; unidet3d/unidet3d.py:96-111,353-354):
``SparseConvTensor`` (features / indices / spatial_shape / batch_size /
replace_feature / per-forward ``indice_dict`` rulebook cache keyed by
``indice_key``), ``SparseSequential``, ``SubMConv3d``, ``SparseConv3d``,
``SparseInverseConv3d``; plus the BatchNorm the reference takes from torch
(``nn.SyncBatchNorm`` / ``nn.BatchNorm1d``).  Same constructor arguments,
same parameter names and weight layout ``[C_out, k, k, k, C_in]``.

Every compute step is a call through the C ABI (include/u3d.h); there is no
PyTorch fallback.  Rows must be in canonical order (ascending
((b*X+x)*Y+y)*Z+z) -- which is what ``ops.voxelize`` / ``UniDet3D.collate``
produce.
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from . import dense
from . import precision as P
from . import account


# ----------------------------------------------------------------------------------------
# geometry: occupancy index + rulebooks
# ----------------------------------------------------------------------------------------
# A grid whose direct-address table would exceed this many bytes (or that u3d_index_words refuses) gets the hashed index
# (csrc/hashidx.hip: memory follows occupancy).  The six indoor configs stay far below (cfg2: 100 MB); U3D_INDEX=hash forces the
# hashed form everywhere (tests), U3D_INDEX=bitmap forbids it.
_INDEX_MODE = os.environ.get('U3D_INDEX', 'auto')
_HASH_ABOVE_BYTES = int(float(os.environ.get('U3D_INDEX_HASH_ABOVE_MB', '4096')) * (1 << 20))


class OccupancyIndex:
    """Cell -> canonical row of one level: bitmap + popcount rank over the grid extent, or -- large extents -- a hash table over the
    occupied cells.  ``table()`` gives the (pointer, pointer, hash_slots) triple every C entry point takes."""

    def __init__(self, bitmap, rank, B, shape, slots=0, ukeys=None, n_dev=None):
        self.bitmap, self.rank, self.B = bitmap, rank, B          # hashed form: table keys (int64 view of uint64) / table values
        self.shape = [int(s) for s in shape]
        self.slots, self.ukeys, self.n_dev = int(slots), ukeys, n_dev

    @property
    def hashed(self) -> bool:
        return self.slots > 0

    def table(self):
        return L.ptr(self.bitmap), L.ptr(self.rank), self.slots

    @staticmethod
    def wants_hash(B, shape) -> bool:
        if _INDEX_MODE == 'hash':
            return True
        nw = L.lib().u3d_index_words(B, *[int(s) for s in shape])
        if _INDEX_MODE == 'bitmap':
            return False
        return nw < 0 or nw * 12 > _HASH_ABOVE_BYTES

    @staticmethod
    def alloc(B, shape, device):
        nw = L.lib().u3d_index_words(B, *[int(s) for s in shape])
        if nw < 0:
            raise L.U3DError(f'occupancy index refused (code {nw}): {L.lib().u3d_last_error().decode()}')
        bitmap = torch.zeros(nw, dtype=torch.int64, device=device)
        rank = torch.empty(nw + 1, dtype=torch.int32, device=device)
        return OccupancyIndex(bitmap, rank, B, shape)

    @staticmethod
    def from_cells(cells: torch.Tensor, B: int, shape) -> 'OccupancyIndex':
        """Hashed index of the cells in ``cells`` (int64 cell ids, duplicates allowed, INT64_MAX = none)."""
        n, dev = cells.shape[0], cells.device
        slots = L.lib().u3d_hash_index_slots(n)
        ukeys = torch.empty(n, dtype=torch.int64, device=dev)
        n_dev = torch.empty(1, dtype=torch.int32, device=dev)
        vals = torch.empty(slots, dtype=torch.int32, device=dev)
        if n == 0:                 # an empty level / point set: a table of free slots, zero rows (the C entry points take n > 0 only)
            keys = torch.full((slots,), -1, dtype=torch.int64, device=dev)
            return OccupancyIndex(keys, vals, B, shape, slots, ukeys, n_dev.zero_())
        keys = torch.empty(slots, dtype=torch.int64, device=dev)
        w = L.ws(L.lib().u3d_hash_index_ws_bytes(n), dev)
        n_cells = int(B) * int(shape[0]) * int(shape[1]) * ((int(shape[2]) + 63) // 64) * 64       # cell ids are < n_cells: bounds the sort's passes
        L.call('u3d_hash_index_build', L.ptr(cells), n, n_cells if n_cells < (1 << 62) else 0, L.ptr(ukeys), L.ptr(n_dev), L.ptr(keys), L.ptr(vals),
               slots, L.ptr(w), L.stream())
        return OccupancyIndex(keys, vals, B, shape, slots, ukeys, n_dev)

    def build_rank(self):
        nw = self.bitmap.numel()
        w = L.ws(L.lib().u3d_index_rank_ws_bytes(nw), self.bitmap.device)
        L.call('u3d_index_rank', L.ptr(self.bitmap), nw, L.ptr(self.rank), L.ptr(w), L.stream())

    def count(self) -> int:
        return int((self.n_dev if self.hashed else self.rank[-1]).item())       # documented read-back (one host sync)

    def coords(self, n: int) -> torch.Tensor:
        c = torch.empty(n, 4, dtype=torch.int32, device=self.bitmap.device)
        if n and self.hashed:
            L.call('u3d_hash_index_coords', L.ptr(self.ukeys), n, *self.shape, L.ptr(c), L.stream())
        elif n:
            L.call('u3d_index_coords', L.ptr(self.bitmap), L.ptr(self.rank), self.B, *self.shape, L.ptr(c), L.stream())
        return c

    @staticmethod
    def from_coords(coords: torch.Tensor, B: int, shape, shift: int = 0) -> 'OccupancyIndex':
        shape = [int(s) for s in shape]
        if OccupancyIndex.wants_hash(B, shape):
            cells = torch.empty(coords.shape[0], dtype=torch.int64, device=coords.device)
            if coords.shape[0] == 0:
                return OccupancyIndex.from_cells(cells, B, shape)
            L.call('u3d_cells_of_coords', L.ptr(coords), coords.shape[0], shift, B, *shape, L.ptr(cells), L.stream())
            return OccupancyIndex.from_cells(cells, B, shape)
        ix = OccupancyIndex.alloc(B, shape, coords.device)
        L.call('u3d_index_mark', L.ptr(coords), coords.shape[0], shift, *ix.shape, L.ptr(ix.bitmap), L.stream())
        ix.build_rank()
        return ix


_PAIR_CACHE: Dict = {}
# rulebook tag (its indice_key) -> the (role, tile height) lists the convolutions asked of that rulebook in earlier steps:
# ``Rulebook.precompute_tiles`` builds them with the rulebook (on the side stream of ``UniDet3D.prefetch``) instead of on first use
# in the middle of the forward / backward pass (34 launches of ~6.6 us on the main stream per step otherwise)
_TILE_HISTORY: Dict = {}
_TILE_PRECOMPUTE = os.environ.get('U3D_TILE_PRECOMPUTE', '1') != '0'


class Rulebook:
    """pair_in / pair_out int32 [K, cap] grouped by offset, ascending inside an offset."""

    def __init__(self, pair_in, pair_out, counts, K, n_in, n_out):
        self.pair_in, self.pair_out, self.counts = pair_in, pair_out, counts
        self.K, self.cap, self.n_in, self.n_out = K, pair_in.shape[1], n_in, n_out
        self._tiles: Dict = {}
        self._total = None
        self.tag = None             # set by the layer that caches it (indice_key): lets the tile lists follow last step's use
        self.coords = self.index = None      # SubM rulebooks: the level's voxels and occupancy index (tile-stationary kernel's tables)
        self._halo: Dict = {}

    def _tile_starts(self, role: str, T: int) -> torch.Tensor:
        key = (role, T)
        if key not in self._tiles:
            rows = self.pair_out if role == 'out' else self.pair_in
            n_dst = self.n_out if role == 'out' else self.n_in
            nt = (n_dst + T - 1) // T
            ts = torch.empty(self.K, nt + 1, dtype=torch.int32, device=rows.device)
            L.call('u3d_tile_starts', L.ptr(rows), L.ptr(self.counts), self.K, self.cap, T, nt, L.ptr(ts), L.stream())
            self._tiles[key] = ts
        return self._tiles[key]

    def tile_starts(self, role: str, T: int, wgrad=None) -> torch.Tensor:
        """``wgrad`` = (C_in, C_out) when T is a weight-gradient tile height: that height follows the row count (u3d_spconv_wgrad_tile_rows),
        so the history remembers the layer shape and the next rulebook recomputes ITS height instead of replaying this one."""
        if self.tag is not None:
            _TILE_HISTORY.setdefault(self.tag, set()).add((role, T) if wgrad is None else (role, 'wgrad', int(wgrad[0]), int(wgrad[1])))
        return self._tile_starts(role, T)

    def halo(self, T: int, H: int):
        """Tables of the tile-stationary SubM kernel (csrc/spconv_ts.hip, u3d_subm_halo) for tiles of T rows and passes of H halo
        rows: (nhalo, halo, loc, pmask).  Built once per rulebook and (T, H); remembered like the tile lists so that the next step's
        rulebook builds them on the side stream."""
        if self.tag is not None:
            _TILE_HISTORY.setdefault(self.tag, set()).add(('halo', int(T), int(H)))
        return self._halo_tables(int(T), int(H))

    def _halo_tables(self, T: int, H: int):
        key = (T, H)
        if key not in self._halo:
            if self.coords is None:
                raise L.U3DError('halo tables need a SubM rulebook built by build_subm_rulebook')
            n, dev = self.n_out, self.coords.device
            nt = (n + T - 1) // T
            pmax = L.lib().u3d_subm_halo_pmax(T, H)
            nhalo = torch.empty(nt, dtype=torch.int32, device=dev)
            halo = torch.empty(nt, 27 * T, dtype=torch.int32, device=dev)
            loc = torch.empty(nt, 27, T, dtype=torch.int16, device=dev)
            pmask = torch.empty(nt, pmax, dtype=torch.int32, device=dev)
            L.call('u3d_subm_halo', L.ptr(self.coords), n, *self.index.table(), self.index.B, *self.index.shape, T, H,
                   L.ptr(nhalo), L.ptr(halo), L.ptr(loc), L.ptr(pmask), L.stream())
            self._halo[key] = (nhalo, halo, loc, pmask)
        return self._halo[key]

    def precompute_tiles(self):
        """Build the tile lists the PREVIOUS rulebook with this tag was asked for: fixed heights (the forward / input-gradient kernels'
        32 / 64 rows) as they were, weight-gradient heights recomputed for THIS rulebook's row count.  Only this step's real requests
        are remembered for the next."""
        if self.tag is None or not _TILE_PRECOMPUTE:
            return
        for key in sorted(_TILE_HISTORY.pop(self.tag, ()), key=str):
            if key[0] == 'halo':
                if self.coords is not None and self.n_out > 0:
                    self._halo_tables(key[1], key[2])
            elif len(key) == 2:
                self._tile_starts(*key)
            else:
                role, _, cin, cout = key
                n_dy = self.n_out if role == 'out' else self.n_in
                if n_dy > 0:
                    self._tile_starts(role, int(L.lib().u3d_spconv_wgrad_tile_rows(self.K, n_dy, cin, cout)))

    @property
    def total_pairs(self) -> int:
        """Number of rulebook pairs (flops accounting).  Host read-back, cached by geometry so a bench
        that replays the same scenes does not synchronise inside its timed region."""
        if self._total is None:
            key = (self.K, self.n_in, self.n_out)
            if key not in _PAIR_CACHE:
                _PAIR_CACHE[key] = int(self.counts.sum().item())
            self._total = _PAIR_CACHE[key]
        return self._total

    def lists(self):
        """Host copy as 27/8 (in_rows, out_rows) arrays -- the canonical rulebook (tests)."""
        c = self.counts.cpu().tolist()
        pi, po = self.pair_in.cpu(), self.pair_out.cpu()
        return [(pi[k, :c[k]].numpy(), po[k, :c[k]].numpy()) for k in range(self.K)]


def build_subm_rulebook(coords: torch.Tensor, index: OccupancyIndex) -> Rulebook:
    n = coords.shape[0]
    dev = coords.device
    pin = torch.empty(27, n, dtype=torch.int32, device=dev)
    pout = torch.empty(27, n, dtype=torch.int32, device=dev)
    cnt = torch.empty(27, dtype=torch.int32, device=dev)
    w = L.ws(L.lib().u3d_subm_rulebook_ws_bytes(n), dev)
    L.call('u3d_subm_rulebook', L.ptr(coords), n, *index.table(), index.B, *index.shape,
           L.ptr(pin), L.ptr(pout), L.ptr(cnt), L.ptr(w), L.stream())
    rb = Rulebook(pin, pout, cnt, 27, n, n)
    rb.coords, rb.index = coords, index
    return rb


def build_down_rulebook(coords: torch.Tensor, B: int, shape):
    """Returns (out_coords, out_shape, out_index, Rulebook)."""
    n = coords.shape[0]
    dev = coords.device
    oshape = [int(s) // 2 for s in shape]
    ix2 = OccupancyIndex.from_coords(coords, B, oshape, shift=1)
    n2 = ix2.count()
    oc = ix2.coords(n2)
    pin = torch.empty(8, n, dtype=torch.int32, device=dev)
    pout = torch.empty(8, n, dtype=torch.int32, device=dev)
    cnt = torch.empty(8, dtype=torch.int32, device=dev)
    w = L.ws(L.lib().u3d_down_rulebook_ws_bytes(n), dev)
    L.call('u3d_down_rulebook', L.ptr(coords), n, *ix2.table(), B, *oshape,
           L.ptr(pin), L.ptr(pout), L.ptr(cnt), L.ptr(w), L.stream())
    return oc, oshape, ix2, Rulebook(pin, pout, cnt, 8, n, n2)


# ----------------------------------------------------------------------------------------
# convolution (forward / dgrad / wgrad through the C ABI)
# ----------------------------------------------------------------------------------------
def _plan(Cs, Cd, K, n_dst, rows_kernel=False):
    """(rows per wave-tile, offset groups) the kernel wants for this shape (``rows_kernel``: u3d_spconv_gmm_bf16a's own plan)."""
    import ctypes
    R, G = ctypes.c_int(0), ctypes.c_int(0)
    fn = L.lib().u3d_spconv_plan_bf16a if rows_kernel else L.lib().u3d_spconv_plan
    if n_dst <= 0 or fn(Cs, Cd, K, n_dst, ctypes.byref(R), ctypes.byref(G)) != 0:
        raise L.U3DError(f'sparse conv: channel combination {Cs}->{Cd} unsupported by the gfx950 kernels')
    return R.value, G.value


# ---- bf16 shadows of the rows the sparse convolutions gather (precision.bf16_rows) ------------------------------------------
# A shadow is an attribute of the fp32 tensor it was rounded from: it travels with that tensor object through autograd (torch keeps
# a tensor's Python object, attributes included, alive as long as the tensor itself) and dies with it; a tensor autograd built by
# ADDING two gradients is a new object without one, and its consumer then gathers the fp32 rows -- same values, same results.
SHADOW_STATS = {'hit': 0, 'miss': 0}


def to_shadow(t: torch.Tensor) -> torch.Tensor:
    """The bf16 shadow of an fp32 [n, C] tensor (C % 32 == 0) as the batch-norm kernels write it: rounded to nearest even, each
    32-channel group in MFMA fragment order -- 8-byte pieces (quad q, half h) at position 2q + h, i.e. the 16 bytes at byte 16 q hold
    channels 4q..4q+3 and 16+4q..16+4q+3 (csrc/bn.hip store_shadow).  Torch ops, for tests and tools."""
    n, C = t.shape
    return t.detach().reshape(n, C // 32, 2, 4, 4).permute(0, 1, 3, 2, 4).to(torch.bfloat16).reshape(n, C).contiguous()


def attach_shadow(t: torch.Tensor, shadow: torch.Tensor):
    t._u3d_shadow = (shadow, t._version)


def shadow_of(t: torch.Tensor):
    e = getattr(t, '_u3d_shadow', None)
    if e is not None and e[1] == t._version and e[0].shape == t.shape and e[0].device == t.device:
        SHADOW_STATS['hit'] += 1
        return e[0]
    SHADOW_STATS['miss'] += 1
    return None


# ---- all weight packs of a model in one launch ------------------------------------------------------------------------------
_PACKED: Dict = {}          # (weight data_ptr, transposed, bf16) -> (buffer, weight tensor, version at pack time)


class WeightPacks:
    """MFMA-fragment-order copies (u3d_weight_pack[_bf16]) of every convolution weight of a module tree, both orientations,
    refreshed by ONE launch (u3d_weight_pack_batch) instead of one pack launch in front of each of the ~90 convolution launches
    of a step.

    Validity contract: in TRAINING mode -- and in eval mode whenever autograd is enabled and a convolution weight requires a gradient
    (fine-tuning with frozen batch norm) -- every ``refresh()`` repacks (one launch per step -- what a changed weight costs anyway),
    and the first eval-mode ``refresh()`` after a training-mode one repacks too, so no optimizer can leave a stale pack behind.
    Otherwise (eval mode) a pack is reused while ``(data_ptr, Tensor._version)`` of every weight is unchanged.  ``_version`` is
    bumped by torch's for-loop / foreach optimizers, ``load_state_dict`` and in-place ops on the parameter, but NOT by
    ``torch.optim.AdamW(fused=True)`` (``_fused_adamw_`` leaves it alone -- round 2's version-keyed cache therefore ran the
    backbone of every bench step after the first on the step-0 weights) and NOT by writes through ``.data``
    (``p.data.copy_()``, mmengine's EMAHook parameter swap): a writer of that kind in eval mode must call ``invalidate()``
    (``UniDet3D.invalidate_weight_packs()``) afterwards."""

    def __init__(self, root: nn.Module):
        self.root = root
        self.convs = [m for m in root.modules() if isinstance(m, _ConvBase) and m.in_channels % 16 == 0 and m.kernel_size != 1]
        self.state = None
        self.bufs: Dict = {}
        self.desc = None
        self.blocks = 0
        self.dirty = False          # a training-mode refresh happened: an optimizer may have stepped since

    def __del__(self):
        for key in list(getattr(self, 'bufs', {})):
            _PACKED.pop(key, None)

    def invalidate(self):
        """Forget the packs (for writers that bypass ``Tensor._version``, see the class docstring)."""
        self.dirty = True
        if self.state is not None:
            self.state = self.state[:2] + (tuple((a, -1) for a, _ in self.state[2]),)
        for key, (buf, w) in self.bufs.items():
            _PACKED[key] = (buf, w, -1)

    def refresh(self):
        if not self.convs:
            return
        bf = P.conv_format()           # 0 fp32 fragments, 1 bf16, 2 three bf16 planes (precision.conv_format)
        dev = self.convs[0].weight.device
        state = (bf, str(dev), tuple((m.weight.data_ptr(), m.weight._version) for m in self.convs))
        # a root kept in eval() (frozen batch norm) while an optimizer steps it: a trainable weight can change behind _version's back
        # (fused AdamW), so every GRAD-ENABLED forward over trainable convolution weights repacks, exactly like training mode.  The
        # test is on requires_grad, not on .grad: zero_grad(set_to_none=True) -- torch's default, and mmengine's order step ->
        # zero_grad -- leaves every .grad None at the time the next forward runs.  Inference (torch.no_grad() /
        # inference_mode, or frozen weights) keeps the version-keyed reuse.
        tuned = torch.is_grad_enabled() and any(m.weight.requires_grad for m in self.convs)
        if state == self.state and not self.root.training and not self.dirty and not tuned:
            return
        self.dirty = bool(self.root.training or tuned)
        rebuild = self.state is None or self.state[0] != bf or self.state[1] != str(dev) or \
            [a for a, _ in self.state[2]] != [a for a, _ in state[2]]
        if rebuild:
            rows, blocks = [], 0
            for key in [k for k in _PACKED if k in self.bufs]:
                _PACKED.pop(key, None)
            self.bufs = {}
            for m in self.convs:
                w = m.weight
                K = m.kernel_size ** 3
                for transposed, (Cd, Cs) in ((0, (m.out_channels, m.in_channels)), (1, (m.in_channels, m.out_channels))):
                    if Cd % 32 or Cs % 16:
                        continue
                    use_bf = bf if Cs % 32 == 0 else 0
                    buf = torch.empty(_pack_floats(w.numel(), use_bf), dtype=torch.float32, device=dev)
                    nvec = w.numel() // (8 if use_bf else 4)          # threads of the pack kernel (x3: three vectors per thread)
                    rows.append([w.data_ptr(), buf.data_ptr(), Cd, K, Cs, transposed, int(use_bf), blocks])
                    blocks += (nvec + 255) // 256
                    self.bufs[(w.data_ptr(), transposed, use_bf)] = (buf, w)
            self.desc = L.h2d(rows, torch.int64, dev)
            self.blocks = blocks
        L.call('u3d_weight_pack_batch', L.ptr(self.desc), self.desc.shape[0], self.blocks, L.stream())
        for key, (buf, w) in self.bufs.items():
            _PACKED[key] = (buf, w, w._version)
        self.state = state


_GMM_ENTRY = {0: ('u3d_weight_pack', 'u3d_spconv_gmm'), 1: ('u3d_weight_pack_bf16', 'u3d_spconv_gmm_bf16'), 2: ('u3d_weight_pack_x3', 'u3d_spconv_gmm_x3')}


def _pack_floats(numel: int, fmt: int) -> int:
    """size of a packed weight buffer in floats: fp32 fragments, bf16 (half), three bf16 planes (one and a half)"""
    return {0: numel, 1: numel // 2, 2: numel * 3 // 2}[fmt]


# Tile-stationary SubM kernel (csrc/spconv_ts.hip; round 6) for the three-plane fp32 path wherever
# u3d_spconv_ts_plan has a shape for it.  OFF by default: correct (tables bit-exact, fp32-level errors, tests/test_gpu_kernels.py) but
# measured BEHIND the pair-list kernels at every real layer shape of cfg2 (level 1, 32 -> 32: 148 us against 128; level 2, 64 -> 64:
# 165 against 112; DESIGN.md 4.15 has the ablations and the cycle trace that say why).  U3D_CONV_TS=1 / set_conv_ts(True) turns it on.
_CONV_TS = os.environ.get('U3D_CONV_TS', '0') == '1'


def set_conv_ts(on: bool) -> bool:
    global _CONV_TS
    prev, _CONV_TS = _CONV_TS, bool(on)
    return prev


@contextlib.contextmanager
def conv_ts(on: bool):
    prev = set_conv_ts(on)
    try:
        yield
    finally:
        set_conv_ts(prev)


# Register-stationary form (spconv_rs_k: weights of a 32 x 32 block in registers, persistent workgroups): U3D_CONV_RS=1 / set_conv_rs
_CONV_RS = os.environ.get('U3D_CONV_RS', '0') == '1'
_RS_MIN_ROWS = int(os.environ.get('U3D_CONV_RS_MIN_ROWS', '40000'))      # below that a level cannot feed one workgroup per CU
_RS_MAX_BLOCKS = int(os.environ.get('U3D_CONV_RS_MAX_BLOCKS', '8'))      # Cs/32 x Cd/32 block launches per convolution at most


def set_conv_rs(on: bool) -> bool:
    global _CONV_RS
    prev, _CONV_RS = _CONV_RS, bool(on)
    return prev


@contextlib.contextmanager
def conv_rs(on: bool):
    prev = set_conv_rs(on)
    try:
        yield
    finally:
        set_conv_rs(prev)


# bf16 operands from bf16 rows (BASELINE configs[2]): the one-plane form spconv_rsb_k; U3D_CONV_RS_BF16=1 / set_conv_rs_bf16
_CONV_RS_BF16 = os.environ.get('U3D_CONV_RS_BF16', '0') == '1'


def set_conv_rs_bf16(on: bool) -> bool:
    global _CONV_RS_BF16
    prev, _CONV_RS_BF16 = _CONV_RS_BF16, bool(on)
    return prev


@contextlib.contextmanager
def conv_rs_bf16(on: bool):
    prev = set_conv_rs_bf16(on)
    try:
        yield
    finally:
        set_conv_rs_bf16(prev)


def _rs_ok(Cs, Cd, n, rb, bf):
    return (_CONV_RS and rb.coords is not None and int(bf) == P.FMT_X3 and Cs % 32 == 0 and Cd % 32 == 0 and n >= _RS_MIN_ROWS
            and (Cs // 32) * (Cd // 32) <= _RS_MAX_BLOCKS)


def _ts_plan(Cs, Cd, n):
    import ctypes
    T, H = ctypes.c_int(0), ctypes.c_int(0)
    if L.lib().u3d_spconv_ts_plan(Cs, Cd, n, ctypes.byref(T), ctypes.byref(H)) != 0:
        return None
    return T.value, H.value


def _gmm(src, weight, transposed, rb, gather, scatter, role, n_dst, addend, flops, bf=0, stats_out=None, src_rows_bf16=None):
    """weight: the layer's [C_out, K, C_in] tensor; transposed=True runs the input-gradient (dst channels = C_in).
    ``stats_out`` (a dict, or None): asks the kernel's epilogue for the per-tile column sums of dst that the batch norm behind
    this convolution needs (``partial`` float [n_tiles, 2, Cd], ``n_tiles``); left empty when the launch splits the kernel
    offsets over groups (deep levels: a few thousand rows, the norm then makes its own pass).
    ``bf``: operand format (precision.conv_format: 0 fp32 MFMAs, 1 bf16 operands, 2 fp32 products from three bf16 planes);
    source channel counts that are not a multiple of 32 (the 6 -> 32 input convolution, padded to 16) stay on the fp32 kernel.
    ``src_rows_bf16``: the bf16 shadow of ``src`` (``shadow_of``) -- the launch then gathers those rows (u3d_spconv_gmm_bf16a)."""
    Cs, Cd = src.shape[1], (weight.shape[2] if transposed else weight.shape[0])
    dst = torch.empty(n_dst, Cd, dtype=torch.float32, device=src.device)
    ts = _ts_plan(Cs, Cd, n_dst) if (_CONV_TS and n_dst and rb.coords is not None and int(bf) == P.FMT_X3 and Cs % 32 == 0
                                       and not (stats_out is not None and _EPILOGUE_STATS)) else None
    if (n_dst and _CONV_RS_BF16 and src_rows_bf16 is not None and int(bf) == P.FMT_BF16 and rb.coords is not None and Cs % 32 == 0
            and Cd % 32 == 0 and n_dst >= _RS_MIN_ROWS and (Cs // 32) * (Cd // 32) <= _RS_MAX_BLOCKS):
        H = int(os.environ.get('U3D_RSB_H', '448'))
        if _PROFILE_FLOPS:
            account.add('conv_gmm', flops, 4.0 * (src.shape[0] * Cs + n_dst * Cd) + 8.0 * rb.total_pairs + 4.0 * rb.K * Cs * Cd)
        nhalo, halo, loc, _pm = rb.halo(64, H)
        hit = _PACKED.get((weight.data_ptr(), int(transposed), P.FMT_BF16))
        if hit is not None and hit[2] == weight._version and hit[1].device == src.device:
            wp = hit[0]
        else:
            wp = torch.empty(_pack_floats(weight.numel(), P.FMT_BF16), dtype=torch.float32, device=src.device)
            L.call('u3d_weight_pack_bf16', L.ptr(weight), L.ptr(wp), Cd, rb.K, Cs, int(transposed), L.stream())
        if stats_out is not None:
            stats_out.clear()
        L.call('u3d_spconv_rs_bf16a', L.ptr(src_rows_bf16), n_dst, L.ptr(wp), L.ptr(nhalo), L.ptr(halo), L.ptr(loc), H, int(transposed),
               Cs, Cd, L.ptr(addend), L.ptr(dst), int(os.environ.get('U3D_RS_WGS', '0')), float(flops), L.stream())
    elif n_dst and _rs_ok(Cs, Cd, n_dst, rb, bf) and not (stats_out is not None and _EPILOGUE_STATS):
        H = int(os.environ.get('U3D_RS_H', '320'))
        if _PROFILE_FLOPS:
            account.add('conv_gmm', flops, 4.0 * (src.shape[0] * Cs + n_dst * Cd) + 8.0 * rb.total_pairs + 4.0 * rb.K * Cs * Cd)
        nhalo, halo, loc, _pm = rb.halo(64, H)
        hit = _PACKED.get((weight.data_ptr(), int(transposed), P.FMT_X3))
        if hit is not None and hit[2] == weight._version and hit[1].device == src.device:
            wp = hit[0]
        else:
            wp = torch.empty(_pack_floats(weight.numel(), P.FMT_X3), dtype=torch.float32, device=src.device)
            L.call('u3d_weight_pack_x3', L.ptr(weight), L.ptr(wp), Cd, rb.K, Cs, int(transposed), L.stream())
        L.call('u3d_spconv_rs_x3', L.ptr(src), n_dst, L.ptr(wp), L.ptr(nhalo), L.ptr(halo), L.ptr(loc), H, int(transposed),
               Cs, Cd, L.ptr(addend), L.ptr(dst), int(os.environ.get('U3D_RS_WGS', '0')), float(flops), L.stream())
    elif ts is not None:
        T, H = ts
        if _PROFILE_FLOPS:
            account.add('conv_gmm', flops, 4.0 * (src.shape[0] * Cs + n_dst * Cd) + 8.0 * rb.total_pairs + 4.0 * rb.K * Cs * Cd)
        nhalo, halo, loc, pmask = rb.halo(T, H)
        hit = _PACKED.get((weight.data_ptr(), int(transposed), P.FMT_X3))
        if hit is not None and hit[2] == weight._version and hit[1].device == src.device:
            wp = hit[0]
        else:
            wp = torch.empty(_pack_floats(weight.numel(), P.FMT_X3), dtype=torch.float32, device=src.device)
            L.call('u3d_weight_pack_x3', L.ptr(weight), L.ptr(wp), Cd, rb.K, Cs, int(transposed), L.stream())
        L.call('u3d_spconv_ts_x3', L.ptr(src), n_dst, L.ptr(wp), L.ptr(nhalo), L.ptr(halo), L.ptr(loc), L.ptr(pmask), T, H, int(transposed),
               Cs, Cd, L.ptr(addend), L.ptr(dst), float(flops), L.stream())
    elif n_dst:
        R, G = _plan(Cs, Cd, rb.K, n_dst, src_rows_bf16 is not None and int(bf) == P.FMT_BF16 and Cs % 32 == 0)
        if _PROFILE_FLOPS:      # N(Cs+Cd)s + 2P*idx + K*Cs*Cd*s  (s = 4 B, idx = 4 B)
            account.add('conv_gmm', flops, 4.0 * (src.shape[0] * Cs + n_dst * Cd) + 8.0 * rb.total_pairs + 4.0 * rb.K * Cs * Cd)
        ws = torch.empty(G * n_dst * Cd, dtype=torch.float32, device=src.device) if G > 1 else None
        partial = None
        if stats_out is not None and G == 1 and _EPILOGUE_STATS:
            n_tiles = (n_dst + R - 1) // R
            partial = torch.empty(n_tiles, 2, Cd, dtype=torch.float32, device=src.device)
            stats_out.update(partial=partial, n_tiles=n_tiles)
        bf = int(bf) if Cs % 32 == 0 else 0
        rows = src_rows_bf16 is not None and bf == P.FMT_BF16       # gather the bf16 shadow: same packed weights, u3d_spconv_gmm_bf16a
        if rows:
            src, partial = src_rows_bf16, None
            if stats_out is not None:
                stats_out.clear()
        pack_fn, gmm_fn = _GMM_ENTRY[bf]
        hit = _PACKED.get((weight.data_ptr(), int(transposed), bf))
        if hit is not None and hit[2] == weight._version and hit[1].device == src.device:
            wp = hit[0]                                   # packed with all the model's weights by WeightPacks.refresh()
        else:
            wp = torch.empty(_pack_floats(weight.numel(), bf), dtype=torch.float32, device=src.device)       # MFMA-fragment order
            L.call(pack_fn, L.ptr(weight), L.ptr(wp), Cd, rb.K, Cs, int(transposed), L.stream())
        if rows:
            L.call('u3d_spconv_gmm_bf16a', L.ptr(src), src.shape[0], L.ptr(wp), L.ptr(gather), L.ptr(scatter), L.ptr(rb.tile_starts(role, R)),
                   rb.K, rb.cap, Cs, Cd, n_dst, R, G, L.ptr(addend), L.ptr(dst), L.ptr(ws), float(flops), L.stream())
        else:
            L.call(gmm_fn, L.ptr(src), src.shape[0], L.ptr(wp), L.ptr(gather), L.ptr(scatter), L.ptr(rb.tile_starts(role, R)),
                   rb.K, rb.cap, Cs, Cd, n_dst, R, G, L.ptr(addend), L.ptr(dst), L.ptr(ws), L.ptr(partial), float(flops), L.stream())
    return dst


# Weight gradients on a side stream (U3D_WGRAD_SIDE_STREAM / set_wgrad_overlap):
#   0  off: every kernel of the backward pass on the one stream;
#   1  the weight gradient of a layer runs next to that layer's input gradient and is joined before the layer's backward returns
#      (round 2: +0.9 % step throughput);
#   2  decoupled: the weight gradients form their own chain on the side stream -- each waits for the gradient it reads, nothing on the
#      main stream waits for them until the backward pass ends (an autograd-engine callback joins the streams; FlatGradBucket joins
#      before it copies a bucket).  Nothing downstream of a layer needs its dW, so the dgrad / batch-norm chain never stalls on a
#      weight-gradient kernel, and the small-level kernels of both chains (a few dozen workgroups each) share the chip.
# Overlapped kernels stretch each other, so the bench's per-family HIP-event timings are taken with the overlap off.
_WGRAD_OVERLAP = int(os.environ.get('U3D_WGRAD_SIDE_STREAM', '0') or 0)
_SIDE = {}
_JOIN_PENDING = {}


def set_wgrad_overlap(mode: int) -> int:
    global _WGRAD_OVERLAP
    prev, _WGRAD_OVERLAP = _WGRAD_OVERLAP, int(mode)
    return prev


_N_SIDE = max(1, int(os.environ.get('U3D_SIDE_STREAMS', '1') or 1))      # weight-gradient chains (round-robin); A/B: tools, DESIGN.md 4.14
_SIDE_RR = {}


def _side_stream(device):
    """next weight-gradient stream of ``device`` (round-robin over U3D_SIDE_STREAMS streams)"""
    if device not in _SIDE:
        _SIDE[device] = [torch.cuda.Stream(device=device) for _ in range(_N_SIDE)]
        _SIDE_RR[device] = 0
    i = _SIDE_RR[device]
    _SIDE_RR[device] = (i + 1) % len(_SIDE[device])
    return _SIDE[device][i]


_ASYNC_DW_SEEN: Dict = {}       # id(parameter) -> True for parameters whose dW of the running backward pass went to the side stream


def async_dw_ok(*params) -> bool:
    """May the weight gradients of ``params`` (a layer's weight and bias) stay on the side stream until the backward pass ends (mode 2)?
    Only when the ONLY consumer of each of them is autograd's AccumulateGrad storing the tensor: a leaf without an existing
    ``.grad``, without tensor hooks or post-accumulate-grad hooks (an optimizer-in-backward would read dW on the main stream at once),
    outside ``create_graph`` -- and used for the FIRST time in this backward pass: a parameter shared by two layers gets its two
    gradients summed by autograd on the main stream as soon as the second arrives.  In that case (and in every other refused one) the
    caller computes in line AND the main stream first waits for everything already queued on the side stream, so that the first
    gradient is complete before autograd adds to it."""
    ok = not torch.is_grad_enabled()
    for w in params:
        if w is None:
            continue
        ok = ok and w.is_leaf and w.grad is None and not getattr(w, '_backward_hooks', None) and \
            not getattr(w, '_post_accumulate_grad_hooks', None) and id(w) not in _ASYNC_DW_SEEN
    if ok:
        for w in params:
            if w is not None:
                _ASYNC_DW_SEEN[id(w)] = True
    elif any(w is not None and id(w) in _ASYNC_DW_SEEN for w in params):
        join_wgrad_stream()
    return ok


def join_wgrad_stream(device=None):
    """Make the current stream wait for the weight-gradient kernels queued on the side stream(s) (mode 2).  Called by the autograd
    callback at the end of a backward pass and by anything that reads ``.grad`` of a convolution weight earlier than that
    (``dist.FlatGradBucket`` before it copies a bucket).  A no-op when nothing is pending."""
    for dev in ([device] if device is not None else list(_JOIN_PENDING)):
        if _JOIN_PENDING.pop(dev, None):
            cur = torch.cuda.current_stream(dev)
            for st in _SIDE[dev]:
                cur.wait_stream(st)


def _queue_join(device):
    """Ask the autograd engine to join the side stream(s) when the running backward pass ends.  A callback is queued for EVERY side-stream
    launch (the first one to run joins and clears the pending flag, the rest are no-ops): a flag-guarded single callback would be lost
    for good if a backward pass died between queueing and running it."""
    _JOIN_PENDING[device] = True
    torch.autograd.Variable._execution_engine.queue_callback(lambda: end_of_backward(device))


def end_of_backward(device=None):
    """The autograd engine's callback when a backward pass ends: join the side stream(s) and forget which parameters sent their
    gradient there (``async_dw_ok`` counts uses per pass; a join in the middle of a pass must NOT reset that)."""
    join_wgrad_stream(device)
    _ASYNC_DW_SEEN.clear()


_PROFILE_FLOPS = False      # bench.py turns this on so that launches carry exact algorithmic flops
# Batch-norm statistics from the convolution epilogue (per-tile column sums, no pass over the conv output): built and tested
# (tests/test_gpu_kernels.py), but OFF by default -- measured on MI355X at cfg2 (round 3, visit C): no time gained (32.60 vs 32.62
# ms/step: 21 statistics passes of ~5 us saved, paid for in the convolution epilogues), 0.9 GB/step less HBM traffic, and the
# fp32 per-tile sums of x^2 make the backbone gradients 35x less accurate (L2 error vs the fp64 oracle on the same activation
# pattern 6.2e-4 instead of 1.8e-5: var = E[x^2] - mean^2 is taken through ~90 ill-conditioned batch-norm backwards).  The norm's
# own pass accumulates in fp64.  U3D_EPILOGUE_STATS=1 turns it on.
_EPILOGUE_STATS = os.environ.get('U3D_EPILOGUE_STATS', '0') == '1'


def set_profile_flops(on: bool):
    """Launches carry their algorithmic flops to the C side (HIP-event timing per kernel family) and are booked in
    ``account`` (sparse conv here, dense GEMMs and attention in dense.py / encoder.py)."""
    global _PROFILE_FLOPS
    _PROFILE_FLOPS = bool(on)
    dense.set_profile_flops(on)
    account.enable(on)


class _SparseConvFn(torch.autograd.Function):
    """mode: 'fwd' (src rows = pair_in, dst rows = pair_out; SubM and strided conv) or
    'inv' (roles swapped; SparseInverseConv3d)."""

    @staticmethod
    def forward(ctx, src, weight, rb: Rulebook, mode: str, addend, stats_out=None):
        cout, cin = weight.shape[0], weight.shape[-1]
        w = weight.reshape(cout, rb.K, cin)
        src = src.contiguous()
        if mode == 'fwd':
            g, s, role, n_dst = rb.pair_in, rb.pair_out, 'out', rb.n_out
        else:
            g, s, role, n_dst = rb.pair_out, rb.pair_in, 'in', rb.n_in
        flops = 2.0 * rb.total_pairs * cin * cout if _PROFILE_FLOPS else 0.0
        ctx.bf = P.conv_format()
        ctx.rows = ctx.bf == P.FMT_BF16 and P.bf16_rows() and cin % 32 == 0
        ctx.src_shadow = shadow_of(src) if ctx.rows else None          # bf16 rows of src: gathered here and by the weight gradient
        dst = _gmm(src, w.contiguous(), False, rb, g, s, role, n_dst, None if addend is None else addend.contiguous(), flops, ctx.bf,
                   stats_out, ctx.src_shadow)
        ctx.save_for_backward(src, weight)
        ctx.rb, ctx.mode, ctx.has_addend = rb, mode, addend is not None
        return dst

    @staticmethod
    def backward(ctx, dout):
        src, weight = ctx.saved_tensors
        rb, mode = ctx.rb, ctx.mode
        cout, cin = weight.shape[0], weight.shape[-1]
        dout = dout.contiguous()
        dsrc = dw = None
        flops = 2.0 * rb.total_pairs * cin * cout if _PROFILE_FLOPS else 0.0
        side = None
        dout_shadow = shadow_of(dout) if ctx.bf == P.FMT_BF16 and P.bf16_rows() and cout % 32 == 0 else None
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            if mode == 'fwd':
                rx, rg, role, n_dy = rb.pair_in, rb.pair_out, 'out', rb.n_out
            else:
                rx, rg, role, n_dy = rb.pair_out, rb.pair_in, 'in', rb.n_in
            if _PROFILE_FLOPS:
                account.add('conv_wgrad', flops, 4.0 * (src.shape[0] * cin + n_dy * cout) + 8.0 * rb.total_pairs + 4.0 * rb.K * cin * cout)
            Tw = L.lib().u3d_spconv_wgrad_tile_rows(rb.K, n_dy, cin, cout)
            ws_bytes = L.lib().u3d_spconv_wgrad_ws_bytes(rb.K, n_dy, cin, cout)
            ts = rb.tile_starts(role, Tw, wgrad=(cin, cout))
            # the weight-gradient walk is bound by its row gathers: bf16 operands pay off only where the matrix
            # work is a visible share -- measured (tools/prof_wgrad.py): 32x32 channels 179 us fp32 vs 221 us bf16, 64x64 150 vs 104
            wg = 'u3d_spconv_wgrad_bf16' if ctx.bf == P.FMT_BF16 and cin * cout >= 64 * 64 else 'u3d_spconv_wgrad'
            xw, gw = src, dout
            if ctx.src_shadow is not None and dout_shadow is not None and L.lib().u3d_spconv_wgrad_rows_supported(cin, cout):
                # both operands exist as bf16 rows: whole-row gathers, LDS transpose reads, bf16 MFMAs over 32 pairs (spconv_wgrad_rows.hip)
                wg, xw, gw = 'u3d_spconv_wgrad_rows', ctx.src_shadow, dout_shadow

            overlap = _WGRAD_OVERLAP if ctx.needs_input_grad[0] else 0       # (the first convolution has no input gradient to run next to)
            if overlap == 2 and not async_dw_ok(weight):
                overlap = 1      # autograd will ADD dw to an existing .grad (or feed it to another node / a hook) on this stream right away: join first
            if overlap:
                side = _side_stream(weight.device)
                side.wait_stream(torch.cuda.current_stream())                 # dout (and everything before it) is complete
                with torch.cuda.stream(side):
                    ws = L.scratch(ws_bytes, weight.device)                   # the side stream's own workspace (keyed by stream)
                    L.call(wg, L.ptr(xw), xw.shape[0], L.ptr(gw), L.ptr(rx), L.ptr(rg), L.ptr(ts),
                           rb.K, rb.cap, n_dy, Tw, cin, cout, L.ptr(dw), L.ptr(ws), float(flops), L.stream())
                for t in (xw, gw, dw, ts, rx, rg):                             # blocks must not be recycled while the side kernel uses them
                    t.record_stream(side)
                if overlap == 2:
                    _queue_join(weight.device)
                    side = None
            else:
                ws = L.scratch(ws_bytes, weight.device)
                L.call(wg, L.ptr(xw), xw.shape[0], L.ptr(gw), L.ptr(rx), L.ptr(rg), L.ptr(ts),
                       rb.K, rb.cap, n_dy, Tw, cin, cout, L.ptr(dw), L.ptr(ws), float(flops), L.stream())
        if ctx.needs_input_grad[0]:
            if mode == 'fwd':
                g, s, role, n_dst = rb.pair_out, rb.pair_in, 'in', rb.n_in
            else:
                g, s, role, n_dst = rb.pair_in, rb.pair_out, 'out', rb.n_out
            dsrc = _gmm(dout, weight.reshape(cout, rb.K, cin).contiguous(), True, rb, g, s, role, n_dst, None, flops, ctx.bf,
                        None, dout_shadow)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        return dsrc, dw, None, None, (dout if ctx.has_addend else None), None


def sparse_conv(src, weight, rb, mode='fwd', addend=None, stats_out=None):
    dst = _SparseConvFn.apply(src, weight, rb, mode, addend, stats_out)
    if P.bf16_rows():
        dst._u3d_from_conv = True      # the batch norm behind this output hands its gradient back with a bf16 shadow (bf16_rows)
    return dst


# ----------------------------------------------------------------------------------------
# batch norm (+ReLU)
# ----------------------------------------------------------------------------------------
def _dist_on():
    from .dist import collectives_on          # > 1 rank, or any initialised group under dist.force_collectives()
    return collectives_on()


def allreduce_bn_sums(sums: torch.Tensor, group=None):
    """SyncBatchNorm exchange: one all-reduce of the fp64 [.., count] vector (RCCL over xGMI
    on the GPU box, gloo in the CPU tests)."""
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


class _BNReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, relu, training, sync, nbt=None, want_skip=False,
                stats=None, yb=None, dx_shadow=False):
        """``stats``: dict(partial, n_tiles) from the epilogue of the convolution that produced x (sparse._gmm), or None.
        ``yb`` (bf16 [n, C] or None): receives y rounded to bf16 in the same pass (precision.bf16_rows); ``dx_shadow``: the backward
        writes such a copy of dx too and attaches it to the gradient it returns (x came out of a sparse convolution)."""
        x = x.contiguous()
        n, C = x.shape
        dev = x.device
        ctx.dx_shadow = bool(dx_shadow)
        st = torch.empty(4, C, dtype=torch.float32, device=dev)     # mean, invstd, scale, shift
        y = torch.empty_like(x)
        ws = L.scratch(L.lib().u3d_bn_ws_bytes(C), dev)
        part, n_tiles = (stats['partial'], stats['n_tiles']) if stats else (None, 0)
        sums = None
        sync_on = sync and _dist_on()
        if training and (n or sync_on):           # a rank without rows still joins the exchange (zero sums, zero count), like nn.SyncBatchNorm
            sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)   # [sum x, sum x^2, rows]
            if sync_on:
                if n:
                    L.call('u3d_bn_stats', L.ptr(x), n, C, L.ptr(part), n_tiles, L.ptr(sums), L.ptr(ws), L.stream())
                else:
                    sums.zero_()
                allreduce_bn_sums(sums)          # rows ride along: no host read-back on the critical path
                L.call('u3d_bn_finalize', L.ptr(sums), -1.0, L.ptr(gamma), L.ptr(beta), eps, momentum,
                       L.ptr(running_mean), L.ptr(running_var), C, L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]),
                       L.ptr(st[3]), L.ptr(nbt), L.stream())
                if n:
                    L.call('u3d_bn_apply', L.ptr(x), L.ptr(st[2]), L.ptr(st[3]), int(relu), n, C, L.ptr(y), L.ptr(yb), L.stream())
            else:                                # one call: stats -> finalize -> apply
                L.call('u3d_bn_forward', L.ptr(x), n, C, L.ptr(part), n_tiles, L.ptr(gamma), L.ptr(beta), eps, momentum, L.ptr(running_mean),
                       L.ptr(running_var), L.ptr(nbt), int(relu), L.ptr(y), L.ptr(yb), L.ptr(st), L.ptr(sums), L.ptr(ws), L.stream())
        else:
            st[0] = running_mean
            st[1] = torch.rsqrt(running_var + eps)
            st[2] = gamma * st[1]
            st[3] = beta - running_mean * st[2]
            if n:
                L.call('u3d_bn_apply', L.ptr(x), L.ptr(st[2]), L.ptr(st[3]), int(relu), n, C, L.ptr(y), L.ptr(yb), L.stream())
        ctx.save_for_backward(x, st, sums)
        ctx.relu, ctx.training, ctx.sync, ctx.want_skip = relu, training, sync, want_skip
        ctx.set_materialize_grads(False)          # an unused skip output sends None, not a zero tensor
        if want_skip:
            # second output = x itself, for the identity branch that leaves the block input next to this norm (residual skip,
            # U-Net concat): its gradient then arrives HERE, together with dy, and is added inside the backward kernel
            # instead of by autograd's accumulation kernel
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, st, fsums = ctx.saved_tensors
        if dy is None:                                 # only the identity branch carried a gradient
            return (dskip,) + (None,) * 14
        dy = dy.contiguous()
        dskip = None if dskip is None else dskip.contiguous()
        n, C = x.shape
        dev = x.device
        dx = torch.empty_like(x)
        dxb = torch.empty(n, C, dtype=torch.bfloat16, device=dev) if ctx.dx_shadow and n else None
        # dgamma, dbeta (local sums: DDP averages later); separate tensors so that autograd can adopt them as .grad without a copy
        dgb = [torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)]
        if not n:
            if ctx.training and fsums is not None and ctx.sync and _dist_on():     # keep the collective sequence of the other ranks
                dist.all_reduce(torch.zeros(2 * C, dtype=torch.float64, device=dev), op=dist.ReduceOp.SUM)
            return dx, dgb[0].zero_(), dgb[1].zero_(), None, None, None, None, None, None, None, None, None, None, None, None
        ws = L.scratch(L.lib().u3d_bn_ws_bytes(C), dev)
        sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
        if ctx.training and fsums is not None:
            if ctx.sync and _dist_on():
                sums[2 * C:] = fsums[2 * C:]         # global row count of the forward pass
                L.call('u3d_bn_bwd_stats', L.ptr(x), L.ptr(dy), L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(st[3]),
                       int(ctx.relu), n, C, L.ptr(sums), L.ptr(ws), L.stream())
                dgb[1] = sums[:C].to(torch.float32)
                dgb[0] = sums[C:2 * C].to(torch.float32)
                dist.all_reduce(sums[:2 * C], op=dist.ReduceOp.SUM)
                L.call('u3d_bn_bwd_apply', L.ptr(x), L.ptr(dy), L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(st[3]),
                       int(ctx.relu), L.ptr(sums), -1.0, n, C, L.ptr(dx), L.ptr(dxb), None, None, L.ptr(dskip), L.stream())
            else:                                    # one call: bwd_stats -> bwd_apply (+ dgamma / dbeta)
                L.call('u3d_bn_backward', L.ptr(x), L.ptr(dy), L.ptr(st), int(ctx.relu), L.ptr(fsums), L.ptr(sums), n, C, L.ptr(dx), L.ptr(dxb),
                       L.ptr(dgb[0]), L.ptr(dgb[1]), L.ptr(dskip), L.ptr(ws), L.stream())
        else:                                        # eval: statistics are constants -> dx = scale * dy'
            L.call('u3d_bn_bwd_stats', L.ptr(x), L.ptr(dy), L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(st[3]),
                   int(ctx.relu), n, C, L.ptr(sums), L.ptr(ws), L.stream())
            dgb[1] = sums[:C].to(torch.float32)
            dgb[0] = sums[C:2 * C].to(torch.float32)
            sums.zero_()
            sums[2 * C] = 1.0
            L.call('u3d_bn_bwd_apply', L.ptr(x), L.ptr(dy), L.ptr(st[0]), L.ptr(st[1]), L.ptr(st[2]), L.ptr(st[3]),
                   int(ctx.relu), L.ptr(sums), -1.0, n, C, L.ptr(dx), L.ptr(dxb), None, None, L.ptr(dskip), L.stream())
        if dxb is not None:
            attach_shadow(dx, dxb)
        return dx, dgb[0], dgb[1], None, None, None, None, None, None, None, None, None, None, None, None


class SparseBatchNorm(nn.Module):
    """BatchNorm over voxel rows with the parameter/buffer names of nn.BatchNorm1d /
    nn.SyncBatchNorm (weight, bias, running_mean, running_var, num_batches_tracked).
    ``sync=True`` all-reduces the statistics across ranks when a process group exists
    (nn.SyncBatchNorm degenerates to plain batch norm without one -- SURVEY.md 2.3)."""

    def __init__(self, num_features, eps=1e-4, momentum=0.1, sync=True):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.sync = num_features, eps, momentum, sync
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def forward(self, x: torch.Tensor, relu: bool = False, skip: bool = False, stats=None, shadow: bool = True):
        """``skip=True`` -> (y, x_id): ``x_id`` is x for a second consumer (the identity branch next to this norm); the gradient that
        consumer sends back is added to dx inside this layer's backward kernel.
        ``stats``: the per-tile column sums the producing convolution's epilogue wrote for exactly this ``x``
        (``SparseConvTensor.stats_for``): the statistics then cost no pass over x."""
        # bf16 operands with bf16 rows in HBM (precision.bf16_rows): y (when a ReLU follows -- the input of a sparse convolution -- and
        # the channel count suits the bf16 kernels) and the gradient handed back to a producing convolution get a bf16 shadow
        rows = P.bf16_rows() and x.is_cuda and x.shape[0] > 0 and x.shape[1] % 32 == 0
        yb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if rows and relu and shadow else None
        dx_shadow = rows and getattr(x, '_u3d_from_conv', False)
        # num_batches_tracked is incremented by the statistics kernel (one launch less per layer)
        out = _BNReLUFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                              self.momentum, relu, self.training, self.sync,
                              self.num_batches_tracked if self.training and (x.shape[0] or (self.sync and _dist_on())) else None, skip,
                              stats if self.training else None, yb, dx_shadow)
        if yb is not None:
            attach_shadow(out[0] if skip else out, yb)
        return out


# ----------------------------------------------------------------------------------------
# SparseConvTensor + layers
# ----------------------------------------------------------------------------------------
class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, indice_dict=None, index=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {} if indice_dict is None else indice_dict
        self._index = index
        self.stats = None           # set by the convolution that produced ``features`` (training): dict(partial, n_tiles, owner)

    def replace_feature(self, new_features, stats=None):
        t = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size,
                             self.indice_dict, self._index)
        if stats:
            t.stats = dict(stats, owner=new_features)
        return t

    def stats_for(self, features):
        """The convolution-epilogue statistics, if they were produced for exactly this feature tensor."""
        return self.stats if self.stats is not None and self.stats.get('owner') is features and 'partial' in self.stats else None

    @property
    def index(self) -> OccupancyIndex:
        if self._index is None:
            key = ('__index__', tuple(self.spatial_shape), self.indices.data_ptr())
            if key not in self.indice_dict:
                self.indice_dict[key] = OccupancyIndex.from_coords(self.indices, self.batch_size, self.spatial_shape)
            self._index = self.indice_dict[key]
        return self._index


class SparseModule(nn.Module):
    pass


class SparseSequential(SparseModule):
    """spconv.SparseSequential: sparse layers see the tensor, dense layers its features;
    a BatchNorm directly followed by ReLU runs as one fused kernel."""

    def __init__(self, *args):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], dict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def __len__(self):
        return len(self._modules)

    def forward(self, x: SparseConvTensor) -> SparseConvTensor:
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(m, SparseBatchNorm):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                feeds_conv = fuse and i + 2 < len(mods) and isinstance(mods[i + 2], _ConvBase)      # only then a bf16 shadow of y is of use
                x = x.replace_feature(m(x.features, relu=fuse, stats=x.stats_for(x.features), shadow=feeds_conv))
                i += 1 if fuse else 0
            elif isinstance(m, nn.Identity):
                pass
            elif isinstance(m, nn.ReLU):
                x = x.replace_feature(_relu(x.features))
            else:
                raise L.U3DError(f'SparseSequential: no gfx950 kernel for dense layer {type(m).__name__}')
            i += 1
        return x


def _relu(f):
    # stand-alone ReLU (not preceded by BN) does not occur on the hot path; identity-BN kernel reuse
    C = f.shape[1]
    one = torch.ones(C, device=f.device)
    zero = torch.zeros(C, device=f.device)
    y = torch.empty_like(f)
    L.call('u3d_bn_apply', L.ptr(f.contiguous()), L.ptr(one), L.ptr(zero), 1, f.shape[0], C, L.ptr(y), None, L.stream())
    return y


def _pad16(f):
    c = f.shape[1]
    return f if c % 16 == 0 else F.pad(f, (0, 16 - c % 16))


class _ConvBase(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, bias=False, indice_key=None):
        super().__init__()
        if bias:
            raise L.U3DError('bias=True is not used by the reference configs and not built')
        k = kernel_size
        self.in_channels, self.out_channels, self.kernel_size, self.indice_key = in_channels, out_channels, k, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def _w16(self):
        c = self.in_channels
        return self.weight if c % 16 == 0 else F.pad(self.weight, (0, 16 - c % 16))


class SubMConv3d(_ConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, bias, indice_key)
        assert kernel_size in (1, 3)

    def forward(self, x: SparseConvTensor, addend: Optional[torch.Tensor] = None) -> SparseConvTensor:
        if self.kernel_size == 1:
            # 1x1 skip convolution = [N, Cin] x [Cin, Cout] GEMM, no rulebook (K5): the decoder's fp32 MFMA GEMM kernels (the
            # library picked a 0.6 ms kernel for the level-1 weight gradient, a [64 x 356k] x [356k x 32] product)
            y = dense.linear(x.features, self.weight.view(self.out_channels, self.in_channels))
            return x.replace_feature(y if addend is None else y + addend)
        st = {} if self.training else None
        return x.replace_feature(sparse_conv(_pad16(x.features), self._w16(), self.geometry(x), 'fwd', addend, st), st)

    def geometry(self, x: SparseConvTensor) -> Rulebook:
        key = self.indice_key if self.indice_key is not None else ('__subm__', id(self))
        rb = x.indice_dict.get(key)
        if rb is None:
            rb = build_subm_rulebook(x.indices, x.index)
            rb.tag = ('subm', key) if isinstance(key, str) else None
            x.indice_dict[key] = rb
        return rb


class SparseConv3d(_ConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, bias, indice_key)
        assert kernel_size == 2 and stride == 2 and padding == 0, 'only k=2,s=2 is on the hot path'

    def forward(self, x: SparseConvTensor) -> SparseConvTensor:
        oc, oshape, ix2, rb = self.geometry(x)
        st = {} if self.training else None
        f = sparse_conv(_pad16(x.features), self._w16(), rb, 'fwd', None, st)
        out = SparseConvTensor(f, oc, oshape, x.batch_size, x.indice_dict, ix2)
        if st:
            out.stats = dict(st, owner=f)
        return out

    def geometry(self, x: SparseConvTensor):
        """Coarser level + rulebook for this conv; cached in ``indice_dict`` (the inverse conv reads the
        same entry), so it can be built ahead of the feature pass (``prepare_geometry``)."""
        key = ('__down__', self.indice_key)
        if key not in x.indice_dict:
            oc, oshape, ix2, rb = build_down_rulebook(x.indices, x.batch_size, x.spatial_shape)
            rb.tag = key
            x.indice_dict[key] = (oc, oshape, ix2, rb)
            x.indice_dict[self.indice_key] = (rb, x.indices, x.spatial_shape, x._index)
        return x.indice_dict[key]


class SparseInverseConv3d(_ConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=False):
        super().__init__(in_channels, out_channels, kernel_size, bias, indice_key)

    def forward(self, x: SparseConvTensor) -> SparseConvTensor:
        if self.indice_key not in x.indice_dict:
            raise L.U3DError(f'SparseInverseConv3d: no rulebook saved under {self.indice_key!r}')
        rb, idx, shape, index = x.indice_dict[self.indice_key]
        st = {} if self.training else None
        f = sparse_conv(_pad16(x.features), self._w16(), rb, 'inv', None, st)
        out = SparseConvTensor(f, idx, shape, x.batch_size, x.indice_dict, index)
        if st:
            out.stats = dict(st, owner=f)
        return out
