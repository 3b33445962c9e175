/*
 * u3d.h -- C ABI of the MI355X (gfx950) kernels behind UniDet3D's detection hot path.
 *
 * Every entry point replaces an operator the reference reaches through a third-party
 * CUDA package (call sites cited per function, file:line relative to the reference repo).
 * Conventions:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless
 *     the name ends in _host;
 *   - returns 0 on success, a negative U3D_E* code otherwise; never throws;
 *   - the caller owns every buffer; scratch is passed explicitly (`ws`, sized by the
 *     matching *_ws_bytes() query); no hidden allocation, no device synchronisation:
 *     work is enqueued on `stream`; counts the host needs (voxel counts) are read back
 *     by the caller from the documented device words;
 *   - thread-safe for distinct streams.
 * Rows of feature matrices are contiguous fp32 ([N, C] row-major, C % 16 == 0 for the
 * convolution operands).  Voxel rows are in CANONICAL order: ascending
 * key = ((b*X + x)*Y + y)*Z + z.  Rulebook pair lists are grouped by kernel offset and
 * ascending in both columns inside an offset.
 */
#ifndef U3D_H_
#define U3D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* u3d_stream_t; /* hipStream_t */

#define U3D_OK 0
#define U3D_EINVAL (-1)   /* bad argument / unsupported size */
#define U3D_ELAUNCH (-2)  /* HIP launch error (hipGetLastError != success) */
#define U3D_EUNSUPPORTED (-3) /* channel combination not instantiated */

/* Bumped with every change of an entry point's argument list; unidet3d_amd/_lib.py refuses a library whose version differs from
 * the one it was written against (a stale .so would misread shifted arguments instead of failing). */
#define U3D_ABI_VERSION 114
int u3d_version(void);
const char* u3d_last_error(void);
/* How the fp32 matrix kernels (decoder GEMMs, attention, sparse convolutions without U3D_BF16_OPERANDS) multiply:
 *   1 (default; env U3D_FP32_MATH=bf16x3): each fp32 operand is split exactly into three bf16 pieces and a product is six
 *     bf16 MFMAs with fp32 accumulation -- fp32-level error (the dropped terms are below 2^-23 of a product), 2.7x less matrix
 *     pipe time than
 *   0 (env U3D_FP32_MATH=mfma): the native v_mfma_f32_* instructions.
 * Returns the previous mode; any other argument only queries.  Process-wide, not thread-safe against running launches. */
int u3d_fp32_math(int mode);

/* Which output-stationary sparse-convolution kernel serves u3d_spconv_gmm_bf16 / _x3:
 *   1 (default; env U3D_GMM_WG unset or != 0): workgroup tiles -- four waves share 4 x tile_rows dst rows, the packed weights of
 *     an offset are staged once per workgroup in LDS and the offset's pairs are dealt evenly to the waves (csrc/spconv_wg.hip) --
 *     where they measured ahead: the three-plane entry point at k_groups = 1; 2: wherever that kernel is instantiated (tests, A/B);
 *   0: wave-private tiles, every item reads its weight fragments through the vector-memory path (csrc/spconv.hip; also what
 *     u3d_spconv_gmm -- native fp32 MFMAs -- and launches with bn_partial always use).
 * Same arguments, same tile_starts, bit-identical pair arithmetic; results differ only in fp32 summation order across offsets
 * of one row (none: both accumulate offsets in ascending order).  Returns the previous mode; other arguments only query. */
int u3d_conv_kernel(int mode);

/* ---- kernel timing (HIP events on the launch stream; used by bench.py's roofline) ---- */
enum { U3D_K_CONV_FWD = 0, U3D_K_CONV_WGRAD = 1, U3D_K_BN = 2, U3D_K_POOL = 3, U3D_K_ATTN_FWD = 4,
       U3D_K_ATTN_BWD = 5, U3D_K_RULEBOOK = 6, U3D_K_VOXELIZE = 7, U3D_K_GEMM = 8, U3D_K_COUNT = 9 };
int u3d_prof_enable(int kernel_class, int on);           /* record start/stop events around each launch of the class */
int u3d_prof_collect(int kernel_class, double* total_ms, int64_t* launches, double* work); /* syncs the events, then resets */

/* =====================================================================================
 * R1  voxelisation -- replaces ME.utils.batch_sparse_collate + ME.TensorField(...).sparse()
 *     + field.inverse_mapping (unidet3d/unidet3d.py:158-174).
 * ===================================================================================== */
/* Per-scene min/max of the coordinate source and mean of xyz; also the batch-wide maximum
 * cell index per axis.  points [n_pts, 6] (xyz rgb); coord_src == NULL -> xyz of `points`
 * scaled by 1/voxel_size (unidet3d.py:159), else coord_src [n_pts,3] already in voxel units
 * (elastic path, :164, pass voxel_size = 1).  pt_offsets int64 [B+1].
 * div_mode 0: IEEE divide (what torch's CPU kernel does -- the oracle); 1: multiply by the
 * fp32 reciprocal (what torch's CUDA div-by-scalar kernel does).
 * stats float [B,12] = min[3] max[3] mean_xyz[3] pad[3]; grid_max int32[3]. */
int u3d_vox_scene_stats(const float* points, const float* coord_src, const int64_t* pt_offsets, int B,
                        int64_t max_pts_per_scene, float voxel_size, int div_mode, float* stats,
                        int32_t* grid_max, void* ws, u3d_stream_t stream);
int64_t u3d_vox_scene_stats_ws_bytes(int B);

/* Occupancy index of one level: bitmap uint64 [B*X*Y*Zw] (Zw = ceil(Z/64); bit z&63 of word
 * ((b*X+x)*Y+y)*Zw + (z>>6)) and word_rank int32 [n_words+1] = exclusive popcount prefix, so
 * row(b,x,y,z) = word_rank[w] + popc(bitmap[w] & ((1<<bit)-1)) IS the canonical row.
 * This is a direct-address (perfect-hash) table over the grid EXTENT -- north_star's "hash-built rulebook" with the hash replaced
 * by the cell address, which is what makes the rows come out in canonical order without a sort.  Its size does not depend on
 * occupancy: 12 bytes per 64 z-cells; large extents use the hashed form further down.  u3d_index_words returns the word count, or a negative code when the grid is invalid or
 * would need more than U3D_INDEX_MAX_WORDS words (refused, never allocated). */
#define U3D_INDEX_MAX_WORDS (1LL << 31)
int64_t u3d_index_words(int B, int X, int Y, int Z);
/* sets the bit of every point's cell; writes pt_cell int64 [n_pts] = word*64 + bit (the cell id).  bitmap must be zeroed; bitmap ==
 * NULL: only the cell ids are written (hashed index, below). */
int u3d_vox_mark(const float* points, const float* coord_src, const int64_t* pt_offsets, int B,
                 int64_t max_pts_per_scene, const float* stats, float voxel_size, int div_mode, int X, int Y,
                 int Z, uint64_t* bitmap, int64_t* pt_cell, u3d_stream_t stream);
/* word_rank[0..n_words] ; word_rank[n_words] = number of active voxels (host reads it back). */
int u3d_index_rank(const uint64_t* bitmap, int64_t n_words, int32_t* word_rank, void* ws, u3d_stream_t stream);
int64_t u3d_index_rank_ws_bytes(int64_t n_words);
/* coords int32 [n_vox,4] (b,x,y,z) in canonical order from the index. */
int u3d_index_coords(const uint64_t* bitmap, const int32_t* word_rank, int B, int X, int Y, int Z,
                     int32_t* coords, u3d_stream_t stream);
/* inverse int64 [n_pts] (point -> voxel row); CSR of points per voxel (vox_offsets int32 [n_vox+1],
 * vox_points int32 [n_pts]); feats [n_vox,6] = unweighted mean of [rgb, xyz - mean_xyz(scene)]. */
int u3d_vox_finalize(const float* points, const int64_t* pt_offsets, int B, int64_t n_pts, const float* stats,
                     const int64_t* pt_cell, const uint64_t* bitmap, const int32_t* word_rank, int64_t hash_slots /* 0: bitmap index */, int64_t n_vox,
                     int64_t* inverse, int32_t* vox_offsets, int32_t* vox_points, float* feats, int feat_ld,
                     void* ws, u3d_stream_t stream);
int64_t u3d_vox_finalize_ws_bytes(int64_t n_pts, int64_t n_vox);

/* Hashed form of the index ("hash-built rulebook in HBM", csrc/hashidx.hip) for grids whose extent makes the direct-address
 * table impractical: memory follows OCCUPANCY.  Same contract (cell -> canonical row, rows ascending in the cell id):
 *   u3d_hash_index_build: cells int64 [n] (cell ids as u3d_vox_mark / u3d_cells_of_coords write them; duplicates allowed; entries
 *     equal to INT64_MAX are ignored) -> radix sort (csrc/radix.hip: hand-written stable LSD sort, no library) -> unique: ukeys
 *     int64 [n] (the first *n_unique entries = the occupied cells in canonical order), n_unique device int32 (host reads it back,
 *     like word_rank[n_words]) -> open-addressing table table_keys uint64 [slots] / table_vals int32 [slots] by 64-bit atomicCAS
 *     insertion, slots = u3d_hash_index_slots(n) (power of two >= 2 n).  n_cells: every valid cell id is < n_cells (the grid's
 *     B X Y ceil(Z/64) 64) -- the sort then only runs over the bits the grid needs; 0 = unknown (63 bits).
 *   u3d_hash_index_coords: coords int32 [n,4] of the occupied cells; u3d_cells_of_coords: the (parent) cell of every voxel of a
 *     level (shift = 1: next coarser level, parents outside the halved grid -> INT64_MAX).
 * Every entry point that takes (bitmap, word_rank) also takes `hash_slots`: 0 = bitmap form; > 0 = the two pointers are
 * (table_keys, table_vals) of a table with that many slots.  The rulebook / voxel kernels are the same for both forms. */
int64_t u3d_hash_index_slots(int64_t n);
int64_t u3d_hash_index_ws_bytes(int64_t n);
int u3d_hash_index_build(const int64_t* cells, int64_t n, int64_t n_cells, int64_t* ukeys, int32_t* n_unique, uint64_t* table_keys,
                         int32_t* table_vals, int64_t slots, void* ws, u3d_stream_t stream);
/* The sort itself (keys only, or keys + permutation): keys_in uint64 [n] -> keys_out [n] ascending, STABLE; perm_out (nullable)
 * int32 [n] = position in keys_in of every sorted key.  key_bits: number of significant low bits (passes = ceil(key_bits / 8)).
 * keys_out must not alias keys_in.  Deterministic: no global atomics, ranks by wave ballots in index order. */
int u3d_sort_u64(const uint64_t* keys_in, int64_t n, int key_bits, uint64_t* keys_out, int32_t* perm_out, void* ws, u3d_stream_t stream);
int64_t u3d_sort_ws_bytes(int64_t n, int with_values);
int u3d_hash_index_coords(const int64_t* ukeys, int64_t n, int X, int Y, int Z, int32_t* coords, u3d_stream_t stream);
int u3d_cells_of_coords(const int32_t* coords, int64_t n, int shift, int B, int X2, int Y2, int Z2, int64_t* cells, u3d_stream_t stream);

/* =====================================================================================
 * R2  SubMConv3d rulebook (spconv indice pairs, k=3) -- first conv of each indice_key
 *     subm1..5 (unidet3d/unidet3d.py:97-103, unidet3d/spconv_unet.py:43-56,138).
 *     pair_in/pair_out int32 [27, n]; counts int32 [27]; offset k=(dx+1)*9+(dy+1)*3+(dz+1),
 *     input = output + (dx,dy,dz).
 * ===================================================================================== */
int u3d_subm_rulebook(const int32_t* coords, int64_t n, const uint64_t* bitmap, const int32_t* word_rank, int64_t hash_slots,
                      int B, int X, int Y, int Z, int32_t* pair_in, int32_t* pair_out, int32_t* counts,
                      void* ws, u3d_stream_t stream);
int64_t u3d_subm_rulebook_ws_bytes(int64_t n);

/* =====================================================================================
 * R3  SparseConv3d(k=2,s=2) rulebook, reused by SparseInverseConv3d
 *     (unidet3d/spconv_unet.py:148-154,178-183).  out = in>>1, k=(x&1)*4+(y&1)*2+(z&1),
 *     out dims = floor(in/2), inputs whose parent is out of range are dropped.
 * ===================================================================================== */
/* sets the bit of (b, x>>shift, y>>shift, z>>shift) for every row whose shifted cell is inside (X,Y,Z);
 * shift = 1 builds the next (coarser) level, shift = 0 indexes a caller-supplied coordinate list. bitmap must be zeroed. */
int u3d_index_mark(const int32_t* coords, int64_t n, int shift, int X, int Y, int Z, uint64_t* bitmap,
                   u3d_stream_t stream);
int u3d_down_rulebook(const int32_t* coords, int64_t n, const uint64_t* bitmap2, const int32_t* word_rank2, int64_t hash_slots,
                      int B, int X2, int Y2, int Z2, int32_t* pair_in, int32_t* pair_out, int32_t* counts,
                      void* ws, u3d_stream_t stream);
int64_t u3d_down_rulebook_ws_bytes(int64_t n);

/* tile_starts int32 [K, n_tiles+1]: lower bound of t*tile_rows in the (ascending) list rows[k][0..counts[k]). */
int u3d_tile_starts(const int32_t* rows, const int32_t* counts, int K, int64_t cap, int tile_rows,
                    int64_t n_tiles, int32_t* tile_starts, u3d_stream_t stream);

/* =====================================================================================
 * K4-K8  sparse convolution, all variants through one gather-MFMA-scatter kernel:
 *   dst[scatter[k][p]] (+)= W_k . src[gather[k][p]]      for p < counts[k], k < K
 * w_packed: the weights in the kernel's MFMA-fragment order, produced by u3d_weight_pack() from spconv's native
 * [C_out,k0,k1,k2,C_in] tensor (transposed = 0 for forward, 1 for the input gradient, where dst = C_in).
 * scatter lists must be ascending (they are, in both columns); tile_starts from u3d_tile_starts
 * on the scatter lists with the tile height u3d_spconv_plan() returns.  addend (nullable, [n_dst,Cd]) initialises the accumulator
 * (fuses the residual add of ResidualBlock.forward, spconv_unet.py:88-89).
 * Replaces SubMConv3d / SparseConv3d / SparseInverseConv3d forward and their input-gradients.
 * ===================================================================================== */
int u3d_spconv_gmm(const float* src, int64_t n_src, const float* w_packed, const int32_t* gather, const int32_t* scatter,
                   const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst,
                   int tile_rows, int k_groups, const float* addend, float* dst, void* ws,
                   float* bn_partial /* nullable, k_groups == 1 only: float [ceil(n_dst / tile_rows)][2][Cd], per-tile sum x | sum x^2 of
                                        dst for the batch norm behind this convolution (u3d_bn_forward / u3d_bn_stats take it) */,
                   double flops_hint, u3d_stream_t stream);
/* bf16-operand form (BASELINE configs[2], "MFMA bf16 on rule GEMM"): same arguments; src / dst / addend stay fp32, gathered
 * rows are rounded to bf16 as the MFMA operand is formed, weights come pre-rounded from u3d_weight_pack_bf16 (Cd*K*Cs*2 bytes,
 * fragment order of v_mfma_f32_16x16x32_bf16); fp32 accumulation.  Cs % 32 == 0. */
int u3d_spconv_gmm_bf16(const float* src, int64_t n_src, const void* w_rows_bf16, const int32_t* gather, const int32_t* scatter,
                        const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                        int k_groups, const float* addend, float* dst, void* ws, float* bn_partial, double flops_hint, u3d_stream_t stream);
int u3d_weight_pack_bf16(const float* w, void* wp, int Cd, int K, int Cs, int transposed, u3d_stream_t stream);
/* bf16 SOURCE ROWS (BASELINE configs[2] with bf16 copies of the gathered activations in HBM): src_bf16 is the shadow a
 * batch-norm call wrote next to its fp32 output (u3d_bn_apply / u3d_bn_bwd_apply, y_bf16 / dx_bf16): [n_src][Cs] bf16, rounded to
 * nearest even, each 32-channel group in MFMA fragment order (16 bytes at byte 16 q = channels 4q..4q+3, 16+4q..16+4q+3).  A lane
 * loads its fragment straight from the row: half the gathered bytes, no LDS transposition, no rounding arithmetic.  Weights from
 * u3d_weight_pack_bf16 (the same pack as u3d_spconv_gmm_bf16); dst / addend fp32, fp32 accumulation; results are bit-identical to
 * u3d_spconv_gmm_bf16 on the fp32 tensor the shadow was rounded from.  Always the workgroup-tile kernel (csrc/spconv_wg.hip). */
int u3d_spconv_gmm_bf16a(const void* src_bf16, int64_t n_src, const void* w_rows_bf16, const int32_t* gather, const int32_t* scatter,
                         const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                         int k_groups, const float* addend, float* dst, void* ws, double flops_hint, u3d_stream_t stream);
/* fp32 products on the bf16 matrix pipe (see u3d_fp32_math): same arguments and results at fp32-level error; the gathered rows are
 * split exactly into three bf16 planes as the MFMA operand is formed, the weights come pre-split from u3d_weight_pack_x3
 * (Cd*K*Cs*6 bytes: three consecutive 1 KB plane blocks per (32-channel group, 16-column block) of the bf16 form's order), a
 * fragment pair takes six v_mfma_f32_16x16x32_bf16.  Cs % 32 == 0.  The host picks this form over u3d_spconv_gmm when
 * u3d_fp32_math(-1) == 1 -- the packed-weight layout belongs to the entry point, so the library does not switch it silently. */
int u3d_spconv_gmm_x3(const float* src, int64_t n_src, const void* w_rows_x3, const int32_t* gather, const int32_t* scatter,
                      const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                      int k_groups, const float* addend, float* dst, void* ws, float* bn_partial, double flops_hint, u3d_stream_t stream);
int u3d_weight_pack_x3(const float* w, void* wp, int Cd, int K, int Cs, int transposed, u3d_stream_t stream);
/* all packs of a model in one launch: desc = device array of n_desc records of eight int64
 * {src pointer, dst pointer, Cd, K, Cs, transposed, format (0 fp32, 1 bf16, 2 x3), first block}; record i owns blocks
 * [first_i, first_{i+1}) of 256 threads: one 16-byte output vector per thread in the fp32 (Cd*K*Cs/4 threads) and bf16 (Cd*K*Cs/8)
 * forms, the three plane vectors of one bf16-form position per thread in the x3 form (Cd*K*Cs/8 threads). */
int u3d_weight_pack_batch(const void* desc, int n_desc, int64_t total_blocks, u3d_stream_t stream);
/* Launch plan for a shape: tile_rows (rows per wave-tile = the tile height to pass to u3d_tile_starts) and
 * k_groups (kernel offsets are split into that many groups when the level has too few rows to fill the chip;
 * the groups' partial sums go through ws = k_groups*n_dst*Cd*4 bytes and a fixed-order reduce).
 * Returns U3D_EUNSUPPORTED for channel counts that are not instantiated. */
int u3d_spconv_plan(int Cs, int Cd, int K, int64_t n_dst, int* tile_rows, int* k_groups);
/* the plan u3d_spconv_gmm_bf16a wants (its light items prefer 32-row tiles and fewer offset groups at the small levels) */
int u3d_spconv_plan_bf16a(int Cs, int Cd, int K, int64_t n_dst, int* tile_rows, int* k_groups);
/* ---- tile-stationary SubMConv3d(k=3) (csrc/spconv_ts.hip; unidet3d/spconv_unet.py:43-56): accumulators of a row tile in registers
 * for all 27 offsets and all C_out columns, every source row of a tile fetched once.
 * u3d_subm_halo builds, per tile of tile_rows consecutive rows (64 / 128 / 256) of a level, from the same occupancy index the
 * rulebook uses:  nhalo int32 [n_tiles]; halo int32 [n_tiles][27*tile_rows] -- the sorted unique source rows of the tile's
 * (row, offset) neighbours, the first nhalo[t] entries valid;  loc uint16 [n_tiles][27][tile_rows] -- position of neighbour k of
 * row r in that list, 0xFFFF = none;  pmask uint32 [n_tiles][pmax] -- bit k of word p: offset k has a neighbour among entries
 * [p*halo_rows, (p+1)*halo_rows) of the list (pmax = u3d_subm_halo_pmax(tile_rows, halo_rows) <= 64).  Integer-only, deterministic.
 * u3d_spconv_ts_x3: dst[r] = addend[r] + sum_k W_k' . src[neighbour_k(r)] with fp32 products from three bf16 planes (weights from
 * u3d_weight_pack_x3, K = 27); flip = 0: k' = k (forward); flip = 1: k' = 26 - k with the TRANSPOSED pack -- the input gradient
 * (SubM pairs are symmetric).  tile_rows / halo_rows must be the pair the tables were built for (u3d_spconv_ts_plan gives the
 * measured choice per shape, U3D_EUNSUPPORTED for shapes the kernel is not instantiated for -- the caller then uses u3d_spconv_gmm_x3). */
/* Register-stationary form of the same convolution (csrc/spconv_ts.hip spconv_rs_k): persistent workgroups (`workgroups` of them, <= 0:
 * one per CU) keep the weights of all 27 offsets of a 32 -> 32 channel block in registers for the whole launch -- each of the four
 * waves the fragments of its 6-7 offsets -- and walk contiguous ranges of 64-row tiles: no weight traffic and no barrier per offset;
 * the waves' partial tiles are added in wave order (deterministic).  Wider layers run as Cs/32 x Cd/32 launches of the block kernel
 * inside this call (accumulating over the source blocks).  Tables: u3d_subm_halo(..., tile_rows = 64, halo_rows) with halo_rows in
 * {256, 320, 416} (tiles with more unique source rows take further passes); pmask is not read.  Same arguments otherwise. */
int u3d_spconv_rs_x3(const float* src, int64_t n, const void* w_rows_x3, const int32_t* nhalo, const int32_t* halo, const uint16_t* loc,
                     int halo_rows, int flip, int Cs, int Cd, const float* addend, float* dst, int workgroups, double flops_hint, u3d_stream_t stream);
/* bf16-operand form of u3d_spconv_rs_x3 on bf16 ROWS (the shadows of u3d_spconv_gmm_bf16a; weights from u3d_weight_pack_bf16): one plane,
 * no split, 56 weight registers per wave -- three workgroups per CU (`workgroups` <= 0: 3 per CU).  halo_rows in {256, 320, 448}. */
int u3d_spconv_rs_bf16a(const void* src_bf16, int64_t n, const void* w_rows_bf16, const int32_t* nhalo, const int32_t* halo, const uint16_t* loc,
                        int halo_rows, int flip, int Cs, int Cd, const float* addend, float* dst, int workgroups, double flops_hint, u3d_stream_t stream);
int u3d_spconv_ts_plan(int Cs, int Cd, int64_t n, int* tile_rows, int* halo_rows);
int u3d_subm_halo_pmax(int tile_rows, int halo_rows);
int u3d_subm_halo(const int32_t* coords, int64_t n, const uint64_t* bitmap, const int32_t* word_rank, int64_t hash_slots, int B,
                  int X, int Y, int Z, int tile_rows, int halo_rows, int32_t* nhalo, int32_t* halo, uint16_t* loc, uint32_t* pmask,
                  u3d_stream_t stream);
int u3d_spconv_ts_x3(const float* src, int64_t n, const void* w_rows_x3, const int32_t* nhalo, const int32_t* halo, const uint16_t* loc,
                     const uint32_t* pmask, int tile_rows, int halo_rows, int flip, int Cs, int Cd, const float* addend, float* dst,
                     double flops_hint, u3d_stream_t stream);
/* dW[(n*K+k)*Cs + c] = sum_p dy[rows_dy[k][p]][n] * x[rows_x[k][p]][c]   (dW is overwritten).
 * The pairs of offset k are processed per tile of dy rows: tile_starts = u3d_tile_starts(rows_dy, ..., tile_rows =
 * u3d_spconv_wgrad_tile_rows(...)); per-tile partial blocks go through ws and are summed in a fixed order
 * (deterministic, no atomics).
 * Both conv entry points address through 32-bit buffer offsets: row counts < 2^24, feature matrices and pair lists < 2 GiB
 * (U3D_EUNSUPPORTED otherwise). */
int u3d_spconv_wgrad(const float* x, int64_t n_rows_x, const float* dy, const int32_t* rows_x, const int32_t* rows_dy,
                     const int32_t* tile_starts, int K, int64_t cap, int64_t n_rows_dy, int tile_rows, int Cs, int Cd,
                     float* dW, void* ws, double flops_hint, u3d_stream_t stream);
/* bf16-operand form (BASELINE configs[2]): x and dy rows are rounded to bf16 as the v_mfma_f32_16x16x32_bf16 operands are
 * formed (32 pairs per instruction); fp32 accumulation, partials and reduce.  Same arguments / workspace. */
int u3d_spconv_wgrad_bf16(const float* x, int64_t n_rows_x, const float* dy, const int32_t* rows_x, const int32_t* rows_dy,
                          const int32_t* tile_starts, int K, int64_t cap, int64_t n_rows_dy, int tile_rows, int Cs, int Cd,
                          float* dW, void* ws, double flops_hint, u3d_stream_t stream);
/* Weight gradient from bf16 ROWS: x_bf16 / dy_bf16 are the shadows u3d_bn_apply (y_bf16) and u3d_bn_bwd_apply (dx_bf16) wrote --
 * [n][C] bf16 in fragment order (see u3d_spconv_gmm_bf16a).  Whole rows are gathered, turned in LDS with ds_read_b64_tr_b16 and
 * multiplied by v_mfma_f32_16x16x32_bf16 (32 pairs per instruction), fp32 accumulation, fixed-order reductions; dW comes out in
 * the natural [C_out][K][C_in] layout.  Same tile plan and workspace as u3d_spconv_wgrad.  Instantiated for 32 / 64 channels (one wave per tile)
 * and the twelve 64 ... 256-channel combinations of levels 3-5 and the tail blocks (four waves share a tile): ask
 * u3d_spconv_wgrad_rows_supported; the operands are the bf16 values, so the result equals u3d_spconv_wgrad on the rounded tensors
 * up to fp32 summation order. */
int u3d_spconv_wgrad_rows(const void* x_bf16, int64_t n_rows_x, const void* dy_bf16, const int32_t* rows_x, const int32_t* rows_dy,
                          const int32_t* tile_starts, int K, int64_t cap, int64_t n_rows_dy, int tile_rows, int Cs, int Cd,
                          float* dW, void* ws, double flops_hint, u3d_stream_t stream);
int u3d_spconv_wgrad_rows_supported(int Cs, int Cd);
int u3d_spconv_wgrad_tile_rows(int K, int64_t n_rows_dy, int Cs, int Cd);
int64_t u3d_spconv_wgrad_ws_bytes(int K, int64_t n_rows_dy, int Cs, int Cd);
/* wp = fragment-ordered copy of the weights for u3d_spconv_gmm ([Cd*K*Cs] floats):
 * transposed = 0: logical W(n,k,c) = w[(n*K+k)*Cs + c]; transposed = 1: W(n,k,c) = w[(c*K+k)*Cd + n]. */
int u3d_weight_pack(const float* w, float* wp, int Cd, int K, int Cs, int transposed, u3d_stream_t stream);
/* wt[(c*K + k)*Cd + n] = w[(n*K + k)*Cs + c] */
int u3d_weight_transpose(const float* w, float* wt, int Cd, int K, int Cs, u3d_stream_t stream);

/* =====================================================================================
 * K9  BatchNorm1d / SyncBatchNorm (train) + ReLU over voxel rows
 *     (unidet3d/spconv_unet.py:42,49,119-124,147,177; unidet3d/unidet3d.py:104-111).
 *     Split so that the caller can all-reduce the fp64 statistics between the two phases
 *     (SyncBatchNorm across data-parallel ranks).
 * ===================================================================================== */
/* sums[0..C) = sum x, sums[C..2C) = sum x^2, sums[2C] = n (fp64): per-workgroup partial rows in ws, added in a fixed order by a
 * second launch -> deterministic.  partial != NULL: the statistics come from the per-tile sums a convolution epilogue wrote
 * (u3d_spconv_gmm bn_partial, float [n_tiles][2][C]) and x is not read.  ws: u3d_bn_ws_bytes(C). */
int u3d_bn_stats(const float* x, int64_t n, int C, const float* partial, int64_t n_tiles, double* sums, void* ws, u3d_stream_t stream);
int64_t u3d_bn_ws_bytes(int C);
/* mean/var from sums/count; scale = gamma*invstd, shift = beta - mean*scale; running stats updated in place
 * (momentum, unbiased var) when running_mean != NULL.  count <= 0: the row count is read from sums[2C]
 * (it then travels through the SyncBatchNorm all-reduce with the sums: no host read-back).
 * num_batches_tracked (nullable, device int64 scalar) is incremented by one. */
int u3d_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, int C, float* mean, float* invstd,
                    float* scale, float* shift, int64_t* num_batches_tracked, u3d_stream_t stream);
/* y_bf16 (nullable, [n][C] bf16, C % 32 == 0): a copy of y rounded to nearest even, fragment order inside each 32-channel group
 * (see u3d_spconv_gmm_bf16a), written in the same pass -- the source rows of u3d_spconv_gmm_bf16a. */
int u3d_bn_apply(const float* x, const float* scale, const float* shift, int relu, int64_t n, int C, float* y, void* y_bf16,
                 u3d_stream_t stream);
/* backward of y = relu(x*scale+shift): sums[0..C) = sum dy', sums[C..2C) = sum dy'*xhat  (dy' = dy*[y>0]);
 * sums[2C] is left untouched (the caller keeps the forward row count there). */
int u3d_bn_bwd_stats(const float* x, const float* dy, const float* mean, const float* invstd,
                     const float* scale, const float* shift, int relu, int64_t n, int C, double* sums, void* ws, u3d_stream_t stream);
/* dx = scale*(dy' - sum_dy/count - xhat*sum_dyxhat/count); dgamma = sum_dyxhat, dbeta = sum_dy (fp32 out);
 * count <= 0: read from sums[2C]. */
int u3d_bn_bwd_apply(const float* x, const float* dy, const float* mean, const float* invstd,
                     const float* scale, const float* shift, int relu, const double* sums, double count,
                     int64_t n, int C, float* dx, void* dx_bf16 /* nullable [n][C] bf16: dx rounded to nearest even (as y_bf16 of u3d_bn_apply) */,
                     float* dgamma, float* dbeta, const float* addend /* nullable [n][C]: added to dx (a second gradient of x, e.g. the residual identity branch) */,
                     u3d_stream_t stream);

/* Single-call forms for the non-distributed case: forward = statistics -> sum + finalize -> apply; backward = bwd_stats -> sum ->
 * bwd_apply.  st float [4C] = mean, invstd, scale, shift (saved for backward); sums double [2C+1]; partial / n_tiles as in
 * u3d_bn_stats. */
int u3d_bn_forward(const float* x, int64_t n, int C, const float* partial, int64_t n_tiles, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, int relu, float* y, void* y_bf16 /* nullable */,
                   float* st, double* sums, void* ws, u3d_stream_t stream);
/* fwd_sums: the forward call's sums vector (its entry [2C] is the row count the backward divides by) */
int u3d_bn_backward(const float* x, const float* dy, const float* st, int relu, const double* fwd_sums, double* sums, int64_t n,
                    int C, float* dx, void* dx_bf16 /* nullable */, float* dgamma, float* dbeta, const float* addend /* nullable, as in u3d_bn_bwd_apply */,
                    void* ws, u3d_stream_t stream);

/* =====================================================================================
 * K11/K12  superpoint pooling -- replaces x.features[inverse_mapping] + torch_scatter.scatter_mean
 *     (unidet3d/unidet3d.py:130) and the superpoint centres (:332-333, :446-447).
 * ===================================================================================== */
/* CSR of element ids per segment: offsets int32 [S+1], list int32 [L], element ids ASCENDING inside every segment (stable radix
 * sort of the ids by segment id, csrc/radix.hip): the sums the pooling kernels take over a segment then have one order, so pooled
 * features -- and with them the whole training step -- are bit-reproducible from run to run. */
int u3d_csr_build(const int64_t* seg_ids, int64_t L, int64_t S, int32_t* offsets, int32_t* list, void* ws,
                  u3d_stream_t stream);
int64_t u3d_csr_build_ws_bytes(int64_t L, int64_t S);
/* out[j] = map[list[j]]  (int64 map -> int32) : composes the CSR with inverse_mapping / superpoint ids */
int u3d_gather_i64_to_i32(const int64_t* map, const int32_t* list, int64_t L, int32_t* out, u3d_stream_t stream);
/* out[s] = out_scale(s) * sum_{j in [offsets[s],offsets[s+1])} src_scale(rows[j]) * src[rows[j]]
 * mean_mode 1: out_scale = 1/max(len,1) (scatter_mean semantics: empty segment -> 0).
 * src_inv_count (nullable, int32 [n_src rows' segment sizes]) : src_scale(r) = 1/max(src_inv_count[r+1]-src_inv_count[r],1)
 * (backward of the mean: rows are superpoints, scaled by their own 1/count). */
int u3d_segment_gather_sum(const float* src, const int32_t* rows, const int32_t* offsets, int64_t S, int C,
                           int mean_mode, const int32_t* src_seg_offsets, float* out, u3d_stream_t stream);
/* centres[s] = mean over the segment's points of (xyz - sub[scene]) ; points row stride pt_ld floats.
 * sub (nullable) [B, sub_ld] holds the per-scene shift (the voxelizer's stats rows: min xyz), the scene
 * of a segment is found from its first point via pt_offsets int64 [B+1]. */
int u3d_segment_mean_xyz(const float* points, int pt_ld, const int32_t* list, const int32_t* offsets, int64_t S,
                         const float* sub, int sub_ld, const int64_t* pt_offsets, int B, float* out,
                         u3d_stream_t stream);

/* out[s] = [min xyz, max xyz] of (xyz - sub[scene]) over the points with ids[p] == s (ids < 0 are skipped):
 * the axis-aligned GT boxes of UniDet3D.get_bboxes_by_masks (unidet3d/unidet3d.py:220-256) for a whole batch in one
 * pass.  Segments without points return [+FLT_MAX-like, -FLT_MAX-like] sentinels.  ws: n_seg*6*4 bytes. */
int u3d_segment_minmax_xyz(const float* points, int pt_ld, const int64_t* ids, int64_t n, int n_seg, const float* sub,
                           int sub_ld, const int64_t* pt_offsets, int B, float* out /*[n_seg,6]*/, void* ws,
                           u3d_stream_t stream);

/* =====================================================================================
 * K13  self-attention core of nn.MultiheadAttention(256, 8, batch_first) per scene
 *     (unidet3d/encoder.py:19-20,36-37): softmax(Q K^T / sqrt(hd)) V over packed variable
 *     length scenes.  qkv [n_total, 3*H*hd] (the in_proj output), cu_seqlens int32 [B+1],
 *     out [n_total, H*hd], lse [H, n_total] (log-sum-exp, saved for backward).  hd == 32.
 * ===================================================================================== */
int u3d_attn_varlen_fwd(const float* qkv, const int32_t* cu_seqlens, int B, int max_len, int64_t n_total,
                        int H, int hd, float scale, float* out, float* lse, double flops_hint,
                        u3d_stream_t stream);
int u3d_attn_varlen_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                        const int32_t* cu_seqlens, int B, int max_len, int64_t n_total, int H, int hd,
                        float scale, float* dqkv, float* delta_ws /*[H,n_total]*/, double flops_hint,
                        u3d_stream_t stream);
/* bf16-operand form of the two calls above (BASELINE configs[2]): identical arguments and results layout; Q/K/V/dO tiles and
 * the probabilities are rounded to bf16 for v_mfma_f32_16x16x32_bf16, softmax statistics and all accumulators stay fp32. */
int u3d_attn_varlen_fwd_bf16(const float* qkv, const int32_t* cu_seqlens, int B, int max_len, int64_t n_total, int H, int hd,
                             float scale, float* out, float* lse, double flops_hint, u3d_stream_t stream);
int u3d_attn_varlen_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse, const int32_t* cu_seqlens,
                             int B, int max_len, int64_t n_total, int H, int hd, float scale, float* dqkv, float* delta_ws,
                             double flops_hint, u3d_stream_t stream);

/* bf16 TENSORS (K14b, csrc/gemm_b16.hip): qkv [n,3D], out [n,D], dout and dqkv are bf16 in HBM (one plane, no conversion while they
 * are staged; the score scale is applied to the scores), lse / delta_ws fp32 -- the tensors the reference's autocast hands to and
 * takes from nn.MultiheadAttention (tools/train.py:86-99). */
int u3d_attn_varlen_fwd_b16(const void* qkv, const int32_t* cu_seqlens, int B, int max_len, int64_t n_total, int H, int hd,
                            float scale, void* out, float* lse, double flops_hint, u3d_stream_t stream);
int u3d_attn_varlen_bwd_b16(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                            int B, int max_len, int64_t n_total, int H, int hd, float scale, void* dqkv, float* delta_ws,
                            double flops_hint, u3d_stream_t stream);

/* =====================================================================================
 * K14  dense fp32 GEMMs of the decoder's nn.Linear layers (unidet3d/encoder.py:19-21,55-61,138-140,
 *      153-155,163): forward C = A W^T + bias, input-gradient (the same call on the transposed weight),
 *      weight-gradient C = A^T B with the row reduction split over workgroups (ws, fixed-order sum).
 * ===================================================================================== */
int u3d_gemm_nt(const float* A /*[M,K]*/, const float* W /*[N,K]*/, const float* bias /*[N] or NULL*/, float* C /*[M,N]*/,
                int64_t M, int N, int K /* % 16 == 0 */, double flops_hint, u3d_stream_t stream);
/* OR-ed into `act` of u3d_linear_act / u3d_linear_dact / u3d_ffn_fwd: the MFMA operands are rounded to bf16 (round to nearest
 * even) on their way into LDS -- data stays fp32 in HBM, accumulation and epilogues stay fp32 (BASELINE configs[2]; the
 * reference trains with `--amp`, tools/train.py:86-99).  K must then be a multiple of 32. */
#define U3D_BF16_OPERANDS 16
/* Linear + activation in the GEMM epilogue (no separate elementwise kernel): Y = act(X W^T + bias), act: 0 none, 1 ReLU,
 * 2 GELU (erf form, unidet3d/encoder.py:58-59 `nn.GELU()`).  For GELU `pre` [M,N] receives X W^T + bias (kept for backward). */
int u3d_linear_act(const float* X /*[M,K]*/, const float* W /*[N,K]*/, const float* bias, int act, float* pre, float* Y /*[M,N]*/,
                   int64_t M, int N, int K, double flops_hint, u3d_stream_t stream);
/* input gradient THROUGH the activation: dX = (dY Wt^T) * act'(aux), Wt [N,K] = the next layer's weight transposed;
 * aux [M,N] = the ReLU output (act 1) or the GELU pre-activation (act 2). */
int u3d_linear_dact(const float* dY /*[M,K]*/, const float* Wt /*[N,K]*/, const float* aux, int act, float* dX /*[M,N]*/,
                    int64_t M, int N, int K, double flops_hint, u3d_stream_t stream);
/* C = A W^T + addend: a second gradient contribution folded into the GEMM that produces the first (flags: 0 or U3D_BF16_OPERANDS) */
int u3d_gemm_nt_add(const float* A /*[M,K]*/, const float* W /*[N,K]*/, const float* addend /*[M,N]*/, int flags, float* C, int64_t M,
                    int N, int K, double flops_hint, u3d_stream_t stream);
/* SURVEY.md 8(b) `ln_linear`: NQ = LayerNorm(X (+ RES)) (u3d_layer_norm_fwd: SUM receives X + RES, STATS the row statistics), then
 * Y = act(NQ W^T + bias) (u3d_linear_act) -- the head of the decoder (unidet3d/encoder.py:187-196: out_norm -> out_bboxes / outs_cls).
 * Two launches: the decoder is post-norm and NQ has further consumers, so the normalised rows are written in any case. */
int u3d_ln_linear(const float* X, const float* RES, const float* gamma, const float* beta, float eps, float* SUM, float* NQ, float* STATS,
                  const float* W /*[N,C]*/, const float* bias, int act, float* PRE, float* Y /*[M,N]*/, int64_t M, int C, int N,
                  double flops_hint, u3d_stream_t stream);
/* SURVEY.md 8(b) `ffn`: Z = act(X W1^T + b1) W2^T + b2 in two launches -- the FFN block (encoder.py:55-61, act 2), input_proj
 * (:138-140, act 1) and the class head (:153-155, act 1).  A [M,hid] receives the activation, H [M,hid] the GELU
 * pre-activation (NULL for ReLU); both are what the backward pass needs (u3d_linear_dact, u3d_gemm_tn). */
int u3d_ffn_fwd(const float* X, const float* W1 /*[hid,d_in]*/, const float* b1, const float* W2 /*[d_out,hid]*/, const float* b2,
                int act, float* H, float* A, float* Z /*[M,d_out]*/, int64_t M, int d_in, int hid, int d_out, double flops_hint,
                u3d_stream_t stream);
/* stand-alone erf-GELU passes (the unfused alternative to act = 2 above; n % 4 == 0): a = gelu(h); dh = da * gelu'(h) */
int u3d_gelu_fwd(const float* h, float* a, int64_t n, u3d_stream_t stream);
int u3d_gelu_bwd(const float* da, const float* h, float* dh, int64_t n, u3d_stream_t stream);
/* colsum_A (nullable, [N]): column sums of A -- the bias gradient of the Linear layer -- produced by the same pass */
int u3d_gemm_tn(const float* A /*[M,N]*/, const float* B /*[M,K]*/, float* C /*[N,K] = A^T B*/, float* colsum_A, int64_t M, int N, int K,
                void* ws, double flops_hint, u3d_stream_t stream);
/* bf16-operand form (BASELINE configs[2]): A and B are rounded to bf16 as they are staged, fp32 accumulation; colsum_A is
 * summed from the unrounded fp32 values.  Same workspace. */
int u3d_gemm_tn_bf16(const float* A, const float* B, float* C, float* colsum_A, int64_t M, int N, int K, void* ws, double flops_hint,
                     u3d_stream_t stream);
int64_t u3d_gemm_tn_ws_bytes(int64_t M, int N, int K);
int u3d_transpose(const float* in /*[R,C]*/, float* out /*[C,R]*/, int R, int C, u3d_stream_t stream);
/* All transposed copies of a step in one launch: desc int64 [n_desc][5] = {src ptr ([R][C] floats), dst ptr ([C][R]), R, C, first
 * block}; a matrix takes ceil(R/32) * ceil(C/32) blocks, total_blocks = their sum (descriptors in ascending first-block order). */
int u3d_transpose_batch(const void* desc, int n_desc, int64_t total_blocks, u3d_stream_t stream);
/* Pre-split W operands for the three-plane ("bf16x3") NT products.  u3d_weight_planes_batch: desc int64 [n_desc][4] = {src ptr (fp32,
 * n8 * 8 values), dst ptr (bf16 [3][n8 * 8]: the three exact planes of every value, 8-element groups split exactly as the GEMM
 * kernels split them in flight), n8, first block}; a matrix takes ceil(n8 / 256) blocks; one launch for every weight (and transposed
 * copy) of a training step.  u3d_gemm_w_planes(w, p, w2, p2): the NEXT NT entry point called on this host thread (u3d_gemm_nt,
 * u3d_linear_act, u3d_linear_dact, u3d_gemm_nt_add, u3d_ln_linear; u3d_ffn_fwd takes p for W1 and p2 for W2) reads its W operand from
 * these planes ([3][N][K] bf16, same N, K as its fp32 W, which must still be passed) when it runs the three-plane kernel AND its W
 * argument is the pointer w (w2) the planes were registered for -- planes of another matrix are ignored; consumed (cleared) by
 * that call whatever kernel it picks; NULL = none.  Results are bit-identical with and without. */
int u3d_weight_planes_batch(const void* desc, int n_desc, int64_t total_blocks, u3d_stream_t stream);
int u3d_gemm_w_planes(const void* w, const void* planes, const void* w2, const void* planes2);

/* ---- K14b: the same Linear layers with bf16 ACTIVATIONS in HBM (csrc/gemm_b16.hip; BASELINE configs[2] -- the reference's `--amp`
 * autocasts the Linear / MultiheadAttention activations, tools/train.py:86-99 around unidet3d/encoder.py:19-21,55-61,138-163).
 * `flags` says which tensors ARE bf16 (row-major, 2 bytes an element); everything else is fp32.  Products run on bf16 MFMAs with
 * fp32 accumulation exactly as under U3D_BF16_OPERANDS (an fp32 operand is rounded to nearest even while it is staged), so a bf16
 * tensor that holds the RNE rounding of an fp32 one gives bit-identical products. */
#define U3D_A_BF16 1   /* the first streamed operand (A) */
#define U3D_B_BF16 2   /* u3d_gemm_tn_b16: the second streamed operand (B) */
#define U3D_C_BF16 4   /* u3d_gemm_nt_b16: the result C -- and `aux` of epi 3 */
/* C[M,N] = epi(A[M,K] W[N,K]^T): epi 0: + bias | 1: relu(+ bias) | 3: aux > 0 ? . : 0 (aux [M,N] = the ReLU output, C's dtype) |
 * 5: + aux (aux and C fp32).  W and bias are fp32.  K % 32 == 0; a bf16 C needs an even N. */
int u3d_gemm_nt_b16(const void* A, const float* W, const float* bias, int epi, const void* aux, void* C, int flags, int64_t M, int N, int K,
                    double flops_hint, u3d_stream_t stream);
/* C[N,K] = A[M,N]^T B[M,K] (fp32), colsum_A (nullable, [N]) = column sums of A; flags: U3D_A_BF16 | U3D_B_BF16.  N (K) % 8 == 0 for a
 * bf16 A (B), % 4 otherwise.  ws: u3d_gemm_tn_b16_ws_bytes.  Fixed-order reduction over the row splits (deterministic). */
int u3d_gemm_tn_b16(const void* A, const void* B, float* C, float* colsum_A, int flags, int64_t M, int N, int K, void* ws, double flops_hint,
                    u3d_stream_t stream);
int64_t u3d_gemm_tn_b16_ws_bytes(int64_t M, int N, int K);
/* erf GELU between bf16 tensors (n % 8 == 0): a = gelu(h); dh = da * gelu'(h) */
int u3d_gelu_fwd_b16(const void* h, void* a, int64_t n, u3d_stream_t stream);
int u3d_gelu_bwd_b16(const void* da, const void* h, void* dh, int64_t n, u3d_stream_t stream);

/* =====================================================================================
 * K15 LayerNorm of the decoder (unidet3d/encoder.py:21,38-40,61,78-79,140,167) with the preceding residual add fused in.
 *      forward: s = x (+ res), y = (s - mean) * rstd * gamma + beta; sum_out receives s when res != NULL (the tensor
 *      the backward needs), stats [M][2] = (mean, rstd).  backward: dx (= gradient of both x and res), dgamma, dbeta
 *      (per-workgroup partials in ws, fixed-order sum).  C % 4 == 0, C <= 1024.
 * ===================================================================================== */
int u3d_layer_norm_fwd(const float* x, const float* res, const float* gamma, const float* beta, int64_t M, int C, float eps,
                       float* sum_out, float* y, float* stats, u3d_stream_t stream);
int u3d_layer_norm_bwd(const float* s, const float* dy, const float* gamma, const float* stats, int64_t M, int C, float* dx,
                       float* dgamma, float* dbeta, void* ws, u3d_stream_t stream);
/* the same passes, additionally writing y (dx) rounded to bf16 into y16 (dx16) [M,C] -- the copy the next GEMM streams (K14b);
 * y16 / dx16 NULL = the calls above */
int u3d_layer_norm_fwd_b16(const float* x, const float* res, const float* gamma, const float* beta, int64_t M, int C, float eps,
                           float* sum_out, float* y, void* y16, float* stats, u3d_stream_t stream);
int u3d_layer_norm_bwd_b16(const float* s, const float* dy, const float* gamma, const float* stats, int64_t M, int C, float* dx, void* dx16,
                           float* dgamma, float* dbeta, void* ws, u3d_stream_t stream);
/* the backward pass of a LayerNorm whose RESULT had up to three consumers (the decoder: the next Linear, the residual into the next
 * LayerNorm, the prediction head -- unidet3d/encoder.py:21,38-40,221-239): the incoming gradient is dy + dy2 + dy3 (dy2 / dy3 nullable),
 * summed as the rows are read instead of by two elementwise passes in front of this call; otherwise u3d_layer_norm_bwd_b16 */
int u3d_layer_norm_bwd_sum(const float* s, const float* dy, const float* dy2, const float* dy3, const float* gamma, const float* stats,
                           int64_t M, int C, float* dx, void* dx16, float* dgamma, float* dbeta, void* ws, u3d_stream_t stream);
int64_t u3d_layer_norm_ws_bytes(int64_t M, int C);

/* ---- inference post-processing of one scene ------------------------------------------------
 * u3d_nms_bev: replaces UniDet3D._single_scene_multiclass_nms with fast_nms=True (unidet3d/unidet3d.py:595-650, which
 * calls mmcv.ops.nms3d_normal per class: greedy suppression by the IoU of the (x, y, dx, dy) rectangles).  boxes [n][6]
 * = (cx, cy, cz, dx, dy, dz) and labels [n] must be ordered by (label ascending, score descending) -- the order the
 * reference visits them in; keep[n] receives 1 for surviving boxes.  n <= 4400 (one workgroup holds the boxes in LDS; the configs use top-k = 1000). */
int u3d_nms_bev(const float* boxes, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream);
/* fast_nms=False branch (unidet3d.py:634-636): mmdet3d aligned_3d_nms on corner boxes (x1,y1,z1,x2,y2,z2) = _bbox_to_loss(boxes);
 * same ordering contract; a box survives only while its 3-D IoU with every kept box of its class is <= iou_thr. */
int u3d_nms_aligned3d(const float* corners, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream);
/* rotated boxes (unidet3d.py:625-626, mmcv.ops.nms3d): boxes [n][7] = (cx, cy, cz, dx, dy, dz, heading); suppression by the BEV IoU
 * of the rotated rectangles; same ordering contract, n <= 3600. */
int u3d_nms_rotated(const float* boxes, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream);
/* u3d_trim_boxes: replaces UniDet3D.trim_bboxes_by_superpoints + get_face_distances (unidet3d/unidet3d.py:540-593, :652-677)
 * points: rows of >= 3 floats with leading dimension pt_ld; (sp_list, sp_offsets[S+1]): CSR of point rows per superpoint
 * (u3d_csr_build).  boxes [nb][box_dim], box_dim = 6 (cx, cy, cz, dx, dy, dz) or 7 (+ heading: the point shift is rotated by
 * -heading about z before the six face tests, as get_face_distances does).  minmax [nb][6] receives (min xyz, max xyz) of the points selected for each box
 * (+inf / -inf when none, as in the reference); centre = (max+min)/2 and size = max-min are left to the caller. */
int u3d_trim_boxes(const float* points, int64_t pt_ld, const int32_t* sp_list, const int32_t* sp_offsets, int S,
                   const float* boxes, int nb, int box_dim, float low_thr, float up_thr, float* minmax, u3d_stream_t stream);

/* =====================================================================================
 * R12  matcher + losses of a batch on the device: forward value AND gradients in five launches
 *      (unidet3d/criterion.py:44-178, :200-320; unidet3d/axis_aligned_iou_loss.py:14-53; unidet3d/rotated_iou_loss.py:14-82).
 *      Single-dataset AND mixed batches of the joint config: every scene has its own dataset (class columns, top-k, weight)
 *      and box parametrisation (6-dof axis-aligned, or 7-dof with a heading -> rotated DIoU in the matcher cost and the loss).
 *  cls [L][n_tot][CU]: logits of the L decoder heads, scenes packed, CU columns per row; box [L][n_tot][BD], BD = 6 (centre, size)
 *  or 7 (+ heading; a yaw-free scene of a 7-column batch ignores its heading column and receives a zero gradient there);
 *  cu int32 [B+1] first query of a scene; gt_off int32 [B+1] first GT of a scene; gt_labels int64 [G] (index into the scene's class
 *  list); gt_boxes [G][BD]; qmask uint8: scene b's [g_b][n_b] query mask at qm_off[b] (int64 [B+1] = prefix sums of n_b g_b,
 *  P = qm_off[B]); scene_meta int32 [B][4] = {classes + 1 of the scene's dataset (the last one is "no object"), top-k,
 *  1 if the scene's boxes carry a heading, offset of the scene's class-column list in cidx}; scene_w float [B] dataset weight;
 *  cidx int32 (nullable): concatenated class-column lists (logit of class c of scene b = cls[..][cidx[off_b + c]]); NULL = the
 *  scene's classes are columns 0 .. C1-1.  max_gt = max g_b (<= 64); min_query_slack = min over scenes with g_b > 0 of
 *  n_b - (topk_b + 1) (must be >= 0, as torch.topk requires in the reference).
 *  loss [1] = sum over layers of lw_cls * mean_b(w_b CE_b) + lw_box * mean over scenes with matches (w_b DIoU_b);
 *  dcls [L][n_tot][CU] / dbox [L][n_tot][BD] receive d loss / d cls (zero in columns outside the scene's class list), d loss / d box.
 *  ws: u3d_criterion_ws_bytes. */
int u3d_criterion_packed(const float* cls, const float* box, const int32_t* cu, const int32_t* gt_off, const int64_t* gt_labels,
                         const float* gt_boxes, const uint8_t* qmask, const int64_t* qm_off, const int32_t* scene_meta,
                         const float* scene_w, const int32_t* cidx, int L, int B, int64_t n_tot, int CU, int BD, int64_t G, int64_t P,
                         int max_gt, int min_query_slack, float w_cls, float w_box, float non_obj_w, float lw_cls, float lw_box,
                         float* loss, float* dcls, float* dbox, void* ws, u3d_stream_t stream);
int64_t u3d_criterion_ws_bytes(int L, int B, int64_t n_tot, int64_t G, int64_t P);
/* box decode of a yaw-free head in one pass each way: PredBBox's exp of the six face distances + _bbox_pred_to_bbox
 * (unidet3d/encoder.py:99-111, :241-271): raw [M][8] (Linear output), centers [M][3] -> box [M][6] (centre, size);
 * backward: draw [M][8] from dbox [M][6] (the angle columns receive 0). */
int u3d_box_decode_fwd(const float* raw, const float* centers, int64_t M, float* box, u3d_stream_t stream);
int u3d_box_decode_bwd(const float* raw, const float* dbox, int64_t M, float* draw, u3d_stream_t stream);
/* the same for a head with a heading / a mixed batch (encoder.py:241-283 incl. the rotated branch): box [M][7] = (centre, w, l, size_z,
 * alpha) on rows with a heading -- yaw_rows[i] != 0, or every row when yaw_rows == NULL -- and (centre, size, 0) on the others, whose
 * raw heading columns get a zero gradient (the reference never evaluates them for scenes of yaw-free datasets, encoder.py:186-199). */
int u3d_box_decode7_fwd(const float* raw, const float* centers, const uint8_t* yaw_rows, int64_t M, float* box, u3d_stream_t stream);
int u3d_box_decode7_bwd(const float* raw, const float* dbox, const uint8_t* yaw_rows, int64_t M, float* draw, u3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* U3D_H_ */
