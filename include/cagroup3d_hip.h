/*
 * cagroup3d_hip.h -- C-ABI of the MI355X (gfx950) hot-path library `libcagroup3d_hip.so`.
 *
 * Every entry point takes raw DEVICE pointers + sizes + a `hipStream_t` (passed as void*),
 * allocates nothing (the caller owns outputs and workspaces), never calls exit(), and
 * returns a status code (CG3D_OK == 0, negative == error).  No torch types appear here.
 *
 * The entry points are exactly what the reference's extension modules bind for this path
 * (citations are into /root/reference):
 *
 *   reference pybind module `iou3d_nms_cuda`  (pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17)
 *       boxes_overlap_bev_gpu -> cg3d_boxes_overlap_bev   (iou3d_nms.cpp:49-66,  kernel .cu:236-249)
 *       boxes_iou_bev_gpu     -> cg3d_boxes_iou_bev       (iou3d_nms.cpp:68-88,  kernel .cu:251-265)
 *       nms_gpu               -> cg3d_nms (rotated=1)     (iou3d_nms.cpp:90-136, kernel .cu:267-311)
 *       nms_normal_gpu        -> cg3d_nms (rotated=0)     (iou3d_nms.cpp:139-186,kernel .cu:328-372)
 *       (literal forms with a HOST keep list and the count as return value: cg3d_nms_gpu / cg3d_nms_normal_gpu)
 *   reference pybind module `KNN_OP`          (pcdet/ops/knn/src/knn.cpp:28-45)
 *       knn_wrapper           -> cg3d_knn                 (knn_cuda.cu:58-115)
 *   reference pybind module `sort_vertices`   (pcdet/ops/rotated_iou/cuda_op/sort_vert.cpp:6-33)
 *       sort_vertices_forward -> cg3d_sort_vertices       (sort_vert_kernel.cu:42-139)
 *   MinkowskiEngine v0.5.4 (README.md:45; un-vendored) operators used by the four hot-path
 *   modules -- coordinate maps, kernel maps, sparse convolution
 *   forward/backward, trilinear feature interpolation, strided average pooling and
 *   quantise-average:
 *       ME.SparseTensor(coordinates=...)             -> cg3d_coord_map_build
 *       CoordinateManager stride / kernel_map        -> cg3d_coord_map_build(qstride) / cg3d_kernel_map
 *       MinkowskiConvolution{,Transpose} fwd / bwd   -> cg3d_spconv_fwd / cg3d_spconv_wgrad
 *       SparseTensor.features_at_coordinates         -> cg3d_interp_map / _fwd / _bwd
 *       MinkowskiAvgPooling, UNWEIGHTED_AVERAGE      -> cg3d_pool_map + cg3d_scatter_mean_fwd / _bwd
 *
 * The CPU oracle (`oracle/liboracle.so`, test infrastructure only) exports the SAME symbols on
 * HOST pointers so the parity tests drive both through one binding.
 */
#ifndef CAGROUP3D_HIP_H
#define CAGROUP3D_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CG3D_OK            0
#define CG3D_ERR_ARG      -1   /* bad size / null pointer / unsupported parameter   */
#define CG3D_ERR_LAUNCH   -2   /* HIP launch or runtime error                      */
#define CG3D_ERR_RANGE    -3   /* coordinate outside the packable range            */

/* coordinate packing limits: |x|,|y|,|z| < 2^14, 0 <= batch < 2^19 */
#define CG3D_COORD_LIMIT  16384
#define CG3D_BATCH_LIMIT  524288

typedef void *cg3d_stream_t; /* hipStream_t; ignored by the oracle */

/* 1 for the HIP library, 0 for the CPU oracle. */
int cg3d_is_device_library(void);
/* ABI version, bumped when a signature changes. */
int cg3d_abi_version(void);
/* Small host table -> device on `stream` (hipMemcpyAsync through the runtime this library is linked to; `src` should be
 * pinned for the copy to be asynchronous).  The host mirror stages its segment / chunk / tile tables through this instead of
 * a chain of framework ops per table. */
int cg3d_h2d_async(void *dst, const void *src, int64_t nbytes, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Coordinate maps (hash-built voxel grid).
 *
 * A coordinate row is int32 (batch, x, y, z).  The hash table is open-addressed:
 * `keys` uint64[cap] (packed coordinate, ~0 == empty) and `vals` int32[cap]; cap is a power of
 * two >= 2*n (use cg3d_hash_capacity).
 *
 * cg3d_coord_map_build: insert n rows; rows are first quantised to the lattice
 *   x' = floor(x / qstride) * qstride (qstride >= 1; 1 == as is).  Duplicates are merged.
 *   Representative of a voxel = its FIRST occurrence (lowest input row).  Output rows are
 *   ordered by ascending representative row.
 *     out_coords   int32 [>=n,4] : unique quantised coordinates (first *n_out rows valid)
 *     unique_index int32 [>=n]   : representative input row of each output row
 *     inverse      int32 [n]     : output row of each input row
 *     n_out        int32 [2]     : [0] number of unique voxels, [1] status: CG3D_OK, or CG3D_ERR_RANGE when a
 *                                  coordinate / batch index does not fit the packed key (the map is unusable then);
 *                                  device memory -- the caller reads both with the one copy that sizes its tensors
 *   After the call the table maps coordinate -> output row.
 *   `ws` needs cg3d_coord_map_ws_bytes(n) bytes.
 *
 * cg3d_morton_order: order[i] = input row of the i-th row in (batch, Morton(x, y, z)) order (15-bit interleave of the
 *   biased coordinates; rows of one voxel keep their input order; out-of-range rows last).  The host engine inserts every
 *   map in this order (me.MORTON_ROWS): 128 consecutive rows are then a spatially compact patch -- what the tile plans of
 *   cg3d_tile_plan_build rely on -- and the strided maps derived from it inherit the order.  ME's own row order is its
 *   hash-map iteration order (unspecified); no output of the path depends on it beyond fp32 summation order.
 * ---------------------------------------------------------------------------------------- */
int64_t cg3d_hash_capacity(int64_t n);
int64_t cg3d_coord_map_ws_bytes(int64_t n);
int64_t cg3d_morton_order_ws_bytes(int64_t n);
int cg3d_morton_order(const int32_t *coords, int64_t n, int32_t *order, void *ws, cg3d_stream_t stream);
int cg3d_coord_map_build(const int32_t *coords, int64_t n, int32_t qstride,
                         uint64_t *keys, int32_t *vals, int64_t cap, void *ws,
                         int32_t *out_coords, int32_t *unique_index, int32_t *inverse,
                         int32_t *n_out, cg3d_stream_t stream);

/* cg3d_kernel_map: nbr[k*nq + q] = table row of (q.batch, q.xyz + offsets[k]) or -1.
 *   q_coords int32 [nq,4]; offsets int32 [K,3] (already scaled by tensor stride/dilation). */
int cg3d_kernel_map(const int32_t *q_coords, int64_t nq, const int32_t *offsets, int32_t K,
                    const uint64_t *keys, const int32_t *vals, int64_t cap,
                    int32_t *nbr, cg3d_stream_t stream);
/* The same map when the queries ARE the table's rows in table order (a coordinate map onto itself) and the offsets are a
 * centred odd kernel (K odd, offsets[K-1-k] == -offsets[k], offsets[K/2] == 0): nbr[K-1-k][j] == i <=> nbr[k][i] == j, so
 * only K/2 offsets are looked up and every hit writes its mirror entry.  Bit-identical output, half the hash probes. */
int cg3d_kernel_map_self(const int32_t *coords, int64_t n, const int32_t *offsets, int32_t K,
                         const uint64_t *keys, const int32_t *vals, int64_t cap,
                         int32_t *nbr, cg3d_stream_t stream);
/* The transposed map (data gradient of a strided / transposed convolution) from the map itself, without the hash table:
 * nbrT int32 [K, n_in], nbrT[k][i] = o  <=>  nbr[k][o] = i, -1 elsewhere (a kernel map is injective per offset). */
int cg3d_kernel_map_transpose(const int32_t *nbr, int32_t K, int64_t n_out, int64_t n_in, int32_t *nbrT,
                              cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse convolution, output-stationary implicit GEMM:
 *   Y[o, :] = bias + sum_k  X[nbr[k, o], :] @ W[k]         (rows with nbr < 0 contribute 0)
 *   X float32 [n_in, cin], W float32 [K, cin, cout], nbr int32 [K, n_out], Y float32 [n_out, cout]
 *   precision: 0 = fp32 operands (v_mfma_f32_32x32x2_f32, exact fp32 products),
 *              1 = bf16 operands, fp32 accumulate: X is RNE-rounded on the fly and `W` must point to the
 *                  buffer written by cg3d_spconv_prep_weights_bf16 (uint16 [K, cout, cin]); cin % 8 == 0,
 *                  16-byte aligned X.  No atomics: every output row is stored once (deterministic).
 *              2 = as 1, but X already holds bf16 rows (uint16 [n_in, cin], written by cg3d_to_bf16): half the
 *                  gather traffic, bit-identical results (the same round-to-nearest-even, done once up front).
 *                  The same value selects bf16-stored rows in cg3d_spconv_pairs_fwd (X) and
 *                  cg3d_spconv_pairs_wgrad (X and dY).
 * cg3d_to_bf16: Xb[i] = bf16(X[i]) (RNE) for n elements (n % 4 == 0).
 *   The data gradient is the same call with (dY, W^T[k] as [K,cout,cin], transposed map).
 * cg3d_spconv_wgrad:  dW[k] = sum_o X[nbr[k,o], :]^T (outer) dY[o, :]   -> float32 [K,cin,cout]
 *   dW is overwritten (the callee zero-fills it when it accumulates with atomics).
 * ---------------------------------------------------------------------------------------- */
int cg3d_to_bf16(const float *X, uint16_t *Xb, int64_t n, cg3d_stream_t stream);
/* X[i] = float(Xb[i]) for n elements (n % 8 == 0, 16-byte aligned): the fp32 copy of rows stored as bf16, for the few
 * consumers of a bf16-stored pass that compute on fp32 rows (interpolation, pooling, the pass's fp32 output). */
int cg3d_from_bf16(const uint16_t *Xb, float *X, int64_t n, cg3d_stream_t stream);
int cg3d_spconv_fwd(const float *X, const float *W, const int32_t *nbr, const float *bias,
                    float *Y, int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout,
                    int32_t precision, cg3d_stream_t stream);
/* Row groups with their own weights (the 18 class branches): `tiles` int32 [ntile,3] = (group, first output row, row
 * count <= 128), a tile never straddles two groups; W holds the G weight sets stacked ([G*K, ...], prepared bf16);
 * precision 1 / 2 only.  tiles == NULL: cg3d_spconv_fwd. */
int cg3d_spconv_fwd_tiled(const float *X, const float *W, const int32_t *nbr, const int32_t *tiles, int64_t ntile,
                          const float *bias, float *Y, int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout,
                          int32_t precision, cg3d_stream_t stream);
int cg3d_spconv_wgrad(const float *X, const float *dY, const int32_t *nbr, float *dW,
                      int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout,
                      int32_t precision, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pair-compacted kernel maps and the gather -> MFMA -> atomic-scatter convolution built on them.
 *
 * cg3d_pairs_count: flags nbr >= 0 over the k-major table, exclusive-scans them into `ws`
 *   (int32 [K*n_out + K*n_out/1024 + 64]) and writes pair_off int32 [K*G+1].  Output rows may be
 *   split into G contiguous groups by row_bounds int32 [G+1] (device; NULL == one group): the pairs
 *   of (offset k, group g) are [pair_off[k*G+g], pair_off[k*G+g+1]); pair_off[K*G] = P.  Groups
 *   let one launch convolve many independent maps with different weights (the 18 class branches of
 *   cagroup_head.py:227-282).  The host reads pair_off to size the lists.
 * cg3d_pairs_fill: pair_in[p] = nbr[k,o], pair_out[p] = o, ordered by (k, o) -- deterministic.
 *   Within one offset every output row (and every input row) appears at most once.
 *
 * cg3d_spconv_pairs_fwd: Y = bias (or 0) -- skipped when `accumulate` != 0 (the caller initialised Y,
 *   only the MFMA kernel is launched) --, then for every segment s = (k, start, count<=128) of
 *   `seg` int32 [nseg,3] (all pairs of a segment share the offset k):
 *       Y[pair_out[p], :] += X[pair_in[p], :] @ W[k]          p in [start, start+count)
 *   accumulated with fp32 global atomics (order is not deterministic; fp32 tolerance applies).
 *   The data gradient is the same call with the two pair lists swapped and W^T.
 * cg3d_spconv_pairs_wgrad: dW[k] = sum_p X[pair_in[p]]^T (outer) dY[pair_out[p]] over the
 *   segments (any count); dW [K,cin,cout] is overwritten.
 *   precision == 1 (bf16 operands, fp32 accumulate): X stays float32 (rounded to bf16 on the fly), and `W`
 *   must point to the buffer written by cg3d_spconv_prep_weights_bf16: uint16 [slots, cout, cin] =
 *   bf16(W[slot, ci, co]) transposed per slot; cin % 8 == 0.
 * ---------------------------------------------------------------------------------------- */
int cg3d_spconv_prep_weights_bf16(const float *W, uint16_t *Wb, int64_t slots, int32_t cin, int32_t cout,
                                  cg3d_stream_t stream);
/* General form: the G*slots_per weight slots come from ONE tensor (W0, Ws == NULL) or from G tensors of slots_per
 * slots each (Ws = device array of G pointers, e.g. the per-class weights of cagroup_head.py:183-188, never stacked
 * in fp32); writes the transposed copy Wb_t [G*slots_per, cout, cin] and/or the plain copy Wb [G*slots_per, cin, cout]
 * (either may be NULL).  The plain copy is the prepared buffer of the data gradient (the swapped problem). */
int cg3d_spconv_prep_weights_bf16_multi(const float *W0, const float *const *Ws, uint16_t *Wb_t, uint16_t *Wb,
                                        int32_t G, int64_t slots_per, int32_t cin, int32_t cout, cg3d_stream_t stream);
/* Table form: the bf16 copies of ANY set of weight slots in one launch.  `table` = device int64 [nrows, 6], one row per
 * 64 x 64 tile of one slot: { address of the float32 slot [cin][cout], address of its transposed bf16 copy [cout][cin]
 * or 0, address of its plain bf16 copy [cin][cout] or 0, cin, cout, tile = ci_tile * ceil(cout / 64) + co_tile }.
 * Same values as cg3d_spconv_prep_weights_bf16_multi writes.  Bit 30 of the tile field: the transposed copy is written in
 * MFMA fragment order (the Wf_t layout of cg3d_spconv_prep_weights_frag; cin % 16 == 0, cout % 32 == 0), bit 29: the plain
 * copy is (the Wf layout; cout % 16 == 0, cin % 32 == 0) -- the operands of cg3d_spconv_tile_fwd. */
int cg3d_spconv_prep_weights_bf16_table(const int64_t *table, int64_t nrows, cg3d_stream_t stream);
int64_t cg3d_pairs_ws_bytes(int64_t total /* K*n_out */);
int cg3d_pairs_count(const int32_t *nbr, int32_t K, int64_t n_out, const int32_t *row_bounds, int32_t G,
                     void *ws, int32_t *pair_off, cg3d_stream_t stream);
int cg3d_pairs_fill(const int32_t *nbr, int32_t K, int64_t n_out, const void *ws, int32_t *pair_in,
                    int32_t *pair_out, cg3d_stream_t stream);
int cg3d_spconv_pairs_fwd(const float *X, const float *W, const int32_t *pair_in, const int32_t *pair_out,
                          const int32_t *seg, int64_t nseg, const float *bias, float *Y, int64_t n_out,
                          int32_t cin, int32_t cout, int32_t precision, int32_t accumulate,
                          cg3d_stream_t stream);
/* precision | CG3D_WGRAD_ACCUMULATE: dW is NOT zero-filled, the products are added to what the caller put there (a pass that
 * zero-fills all of its weight gradients with one memset, or a gradient accumulated over several calls). */
#define CG3D_WGRAD_ACCUMULATE 0x100
int cg3d_spconv_pairs_wgrad(const float *X, const float *dY, const int32_t *pair_in, const int32_t *pair_out,
                            const int32_t *seg, int64_t nseg, float *dW, int32_t K, int32_t cin, int32_t cout,
                            int32_t precision, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Split operands ("bf16x3"): fp32-accurate products on the bf16 matrix pipe, for the parts of the path that keep the
 * reference's fp32 arithmetic (the two heads: cagroup_head.py:227-282, cagroup_roi_head.py:69-91) while BASELINE.json
 * configs[1] runs the backbone in bf16.  With hi = bf16(x) and lo = bf16(x - hi),
 *     x w = xhi whi + xlo whi + xhi wlo + O(2^-16 |x w|),
 * three bf16 products accumulated in fp32.  The split lives in the OPERANDS: every bf16 kernel of this library then runs
 * unchanged on a three times longer contraction (pass 3 cin where it asks for cin).
 *
 * cg3d_to_bf16_split: Xs uint16 [n_rows, 3 c] = [ bf16(X) | bf16(X - bf16(X)) | bf16(X) ] row by row (c % 4 == 0).
 * cg3d_spconv_prep_weights_split: float32 [slot][cin][cout] (W0, or G tensors Ws as in ..._bf16_multi) ->
 *     W_t: the forward operand over the contraction index ci' = part * cin + ci in [0, 3 cin): parts 0 and 1 hold
 *          bf16(W), part 2 holds bf16(W - bf16(W)); uint16 [slot][cout][3 cin], or (frag != 0) in the MFMA fragment order
 *          of cg3d_spconv_prep_weights_frag with kdim = 3 cin;
 *     W_p: the data gradient's operand over co' = part * cout + co in [0, 3 cout): uint16 [slot][cin][3 cout] or fragment
 *          order with kdim = 3 cout.  Either may be NULL.
 *   Bit 28 of the tile field of a cg3d_spconv_prep_weights_bf16_table row writes the same two operands (the row's
 *   destination addresses are those of the slot's 3 cin cout-element blocks).
 * cg3d_spconv_pairs_wgrad with precision 3: X and dY are split rows (uint16 [n, 3 cin] / [n, 3 cout]);
 *     dW = Xhi^T dYhi + Xlo^T dYhi + Xhi^T dYlo     (cin % 8 == 0, cout % 8 == 0).
 * ---------------------------------------------------------------------------------------- */
int cg3d_to_bf16_split(const float *X, uint16_t *Xs, int64_t n_rows, int32_t c, cg3d_stream_t stream);
int cg3d_spconv_prep_weights_split(const float *W0, const float *const *Ws, uint16_t *W_t, uint16_t *W_p, int32_t G,
                                   int64_t slots_per, int32_t cin, int32_t cout, int32_t frag, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse convolution on LDS-staged neighbour tiles (forward and data gradient of every MinkowskiConvolution with
 * K > 1, cin % 64 == 0, cout % 64 == 0 in the bf16 mode; ME ConvolutionForwardGPU / BackwardGPU, call sites
 * backbones_3d/biresnet.py:358-406, dense_heads/cagroup_head.py:259-275).
 *
 * A TILE is 128 consecutive output rows (tile t = rows [128 t, 128 t + 128), or a row of `tiles` int32 [ntile,3] =
 * (weight group, first row, row count <= 128) for grouped convolutions).  A tile PLAN re-encodes the kernel map
 * nbr[K, n_out] per tile so that the distinct input rows a tile touches are staged ONCE into LDS:
 *   pass_tab int32 [ntile, maxpass, 4] = (k0, k1, u_off, u_cnt): the offsets [k0,k1) of the pass touch the u_cnt
 *            distinct input rows ulist[u_off .. u_off+u_cnt) (u_cnt <= ucap, the LDS row capacity); npass int32 [ntile]
 *   slots    uint16 [ntile, K, 128]: 0 = no neighbour, s > 0 = input row ulist[u_off + s - 1] of the pass holding k
 *   live     uint8  [ntile, K]: bit m set = some row of the 32-row block m of the tile has a neighbour at offset k
 *   ulist    int32  [ulist_cap]; ulist_cap >= the number of pairs of the map is always enough; cursor int32 [2]:
 *            [0] = entries used, [1] != 0 = ulist / pass_tab overflowed (the plan is unusable)
 * The slot numbering inside a pass is the implementation's choice (tests check the plan by decoding it).
 * 128 <= ucap <= 1023; maxpass >= K is always enough.
 * `order` (optional, tiles == NULL only): int32 [n_out], a permutation of the output rows; position p of tile t is then
 *   output row order[128 t + p] -- in the plan's slot table and where cg3d_spconv_tile_fwd stores its rows.
 *
 * cg3d_tile_row_order: the permutation that sorts the rows inside every window of `window` consecutive rows (a power of
 *   two, 128 <= window <= CG3D_TILE_WINDOW) by their set of live offsets (bit k = nbr[k][row] >= 0; K <= 32), ties in row
 *   order.  window 1024, sparse maps: for the transposed map of a strided convolution and the map of a transposed convolution
 *   (a row has neighbours only at the offsets of its parity class) the tiles cut from it multiply 2-3 x the rows they need
 *   instead of 7-8 x.  window 128 (= one tile): the tiles keep their rows, but rows with the same live offsets share 32-row
 *   blocks, whose dead (offset, block) pairs cg3d_spconv_tile_fwd skips (the plan's `live` bits).
 *
 * cg3d_spconv_prep_weights_frag: fp32 [slot][cin][cout] (one tensor W0, or G tensors Ws like
 *   cg3d_spconv_prep_weights_bf16_multi) -> bf16 in MFMA B-fragment order,
 *     Wf_t (forward operand):        [slot][co/32][ci/16][(ci/8 & 1)*32 + co%32][ci%8]   cin % 16 == 0, cout % 32 == 0
 *     Wf   (data-gradient operand):  [slot][ci/32][co/16][(co/8 & 1)*32 + ci%32][co%8]   cout % 16 == 0, cin % 32 == 0
 *   either may be NULL.
 * cg3d_spconv_tile_fwd: Y[o,:] = bias + sum_k X[nbr[k,o],:] @ W[k] with X bf16 rows uint16 [n_in, cin] (cg3d_to_bf16),
 *   Wf the fragment-ordered weights, fp32 accumulation, every output row stored once.  ksplit > 1 splits the live
 *   offsets over that many workgroups per tile (small maps); their partial sums meet in Y through fp32 atomics (the
 *   callee zero-fills Y).  The data gradient is the same call on the plan of the transposed map with the plain
 *   fragment copy; for a map of a coordinate map ONTO ITSELF with a centred odd kernel (nbrT[k] == nbr[K-1-k]) it is
 *   the call on the FORWARD plan with wrev = 1: offset k then reads weight slot K-1-k (per group when `tiles` is
 *   given), so neither the transposed map nor its plan is ever built.  The slot-table blocks (32 offsets each) of one
 *   pass of a cin == 64 layer share the rows staged for the pass (K = 125 / 729 class convolutions).
 *   Launch needs cg3d_spconv_tile_lds_bytes(ucap) <= 160 KB of LDS per workgroup.
 * ---------------------------------------------------------------------------------------- */
#define CG3D_TILE_ROWS 128
#define CG3D_TILE_WINDOW 1024
int cg3d_tile_row_order(const int32_t *nbr, int32_t K, int64_t n_out, int32_t window, int32_t *order,
                        cg3d_stream_t stream);
int cg3d_tile_plan_build(const int32_t *nbr, int32_t K, int64_t n_out, const int32_t *tiles, int64_t ntile,
                         int32_t ucap, int32_t maxpass, uint16_t *slots, uint8_t *live, int32_t *pass_tab,
                         int32_t *npass, int32_t *ulist, int64_t ulist_cap, int32_t *cursor, const int32_t *order,
                         cg3d_stream_t stream);
/* cg3d_spconv_tile_fwd, `wrev` argument: bit 0 = the weight slots are walked in reverse (data gradient of a symmetric map on
 * the forward plan); bit 1 (CG3D_TILE_OUT_BF16, needs ksplit == 1) = Y is uint16 [n_out, cout]: the fp32 sums (+ bias)
 * rounded to bf16 on the store (half the output stream; `stats` still sums the fp32 values).
 * cg3d_linear_fwd, `ksplit` argument: bit 16 (CG3D_LINEAR_OUT_BF16, with ksplit == 1) = the same for its Y. */
#define CG3D_TILE_OUT_BF16 2
#define CG3D_LINEAR_OUT_BF16 0x10000
int cg3d_spconv_prep_weights_frag(const float *W0, const float *const *Ws, uint16_t *Wf_t, uint16_t *Wf, int32_t G,
                                  int64_t slots_per, int32_t cin, int32_t cout, cg3d_stream_t stream);
int64_t cg3d_spconv_tile_lds_bytes(int32_t ucap);
int cg3d_spconv_tile_fwd(const uint16_t *X, const uint16_t *Wf, const uint16_t *slots, const uint8_t *live,
                         const int32_t *pass_tab, const int32_t *npass, const int32_t *ulist, int32_t maxpass,
                         int32_t ucap, const int32_t *tiles, int64_t ntile, const int32_t *order, const float *bias,
                         float *Y, int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout, int32_t ksplit,
                         int32_t wrev, float *stats, cg3d_stream_t stream);
/* stats (optional; needs ksplit == 1, tiles == NULL, cout <= 512): float32 [CG3D_BN_SLOTS][2][cout], ZERO-FILLED by the
 * caller: the sum of every output channel and of its square over the rows stored are ADDED to it (fp32 atomics) -- the
 * statistics table cg3d_bn_sums would fill in a pass of its own over Y; hand it to cg3d_bn_apply_sums.
 * (cg3d_spconv_tile_grid = its number of [2][cout] rows, CG3D_BN_SLOTS.) */
int32_t cg3d_spconv_tile_grid(int64_t ntile, int32_t cout, int32_t ksplit);

/* ------------------------------------------------------------------------------------------
 * Y = X @ W (+ bias) on bf16 rows, fp32 accumulation: the kernel-size-1 convolutions (ME: a plain matrix product per
 * sparse tensor; call sites backbones_3d/biresnet.py:270-280,308-315, dense_heads/cagroup_head.py:163-188) forward and
 * data gradient, and with ksplit the per-RoI 7^3 -> centre contraction (roi_heads/cagroup_roi_head.py:74-91).
 *   X uint16 [n, cin] bf16 rows (cg3d_to_bf16 / the BatchNorm kernels' Y16); Wf = ONE slot of
 *   cg3d_spconv_prep_weights_frag: the transposed copy Wf_t of W [cin, cout] for Y = X @ W, the plain copy Wf for the data
 *   gradient dX = dY @ W^T (then cin / cout below are the gradient's: contraction = W's cout, outputs = W's cin);
 *   cin % 64 == 0, cout % 64 == 0; bias float32 [cout] or NULL; Y float32 [n, cout].
 *   ksplit > 1: the contraction is cut into that many ranges of 64-channel chunks (ksplit <= cin / 64).  With `partials`
 *   (caller-owned scratch, float32 [ksplit, n, cout], 16-byte aligned) every range stores its product there and a second
 *   launch of the same call sums them (+ bias) into Y; with partials == NULL the ranges meet in Y through fp32 atomics
 *   (the callee zero-fills Y) -- many ranges on few rows serialise on the same addresses, so give the scratch then.
 *   stats (optional, ksplit == 1): the layer's BatchNorm statistics table, float32 [CG3D_BN_SLOTS][2][cout], zero-filled by
 *   the caller, filled like `stats` of cg3d_spconv_tile_fwd.
 * ---------------------------------------------------------------------------------------- */
int cg3d_linear_fwd(const uint16_t *X, const uint16_t *Wf, const float *bias, float *Y, int64_t n, int32_t cin, int32_t cout,
                    int32_t ksplit, float *stats, float *partials, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Trilinear interpolation of a tensor-stride-`ts` map at continuous coordinates
 * (SparseTensor.features_at_coordinates; reference call sites biresnet.py:182-197,376,389,394).
 *   q float32 [nq,4] (batch, x, y, z) in input-grid units.
 *   idx int32 [nq,8] source rows (-1 = absent corner, contributes 0, no renormalisation),
 *   w float32 [nq,8] weights prod_d (1 - |q_d - c_d| / ts).
 *   fwd: out[q,:] = sum_j w[q,j] F[idx[q,j],:];  bwd: dF[idx[q,j],:] += w[q,j] dout[q,:]
 *   (dF must be zero-filled by the caller).
 * ---------------------------------------------------------------------------------------- */
int cg3d_interp_map(const float *q, int64_t nq, int32_t ts,
                    const uint64_t *keys, const int32_t *vals, int64_t cap,
                    int32_t *idx, float *w, cg3d_stream_t stream);
int cg3d_interp_fwd(const float *F, const int32_t *idx, const float *w, float *out,
                    int64_t nq, int32_t c, cg3d_stream_t stream);
int cg3d_interp_bwd(const float *dout, const int32_t *idx, const float *w, float *dF,
                    int64_t nq, int32_t c, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Row gather / scatter-add (the forward / backward of every `features[index]` on the path:
 * cagroup_head.py:234-252, cagroup_roi_head.py:72).
 *   cg3d_gather_rows:      out[i,:]  = F[idx[i],:]            idx int32 [n] (all >= 0)
 *   cg3d_scatter_add_rows: dF[idx[i],:] += dout[i,:]          dF zero-filled by the caller
 * ---------------------------------------------------------------------------------------- */
int cg3d_gather_rows(const float *F, const int32_t *idx, float *out, int64_t n, int32_t c,
                     cg3d_stream_t stream);
int cg3d_scatter_add_rows(const float *dout, const int32_t *idx, float *dF, int64_t n, int32_t c,
                          cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Average pooling / quantise-average as "scatter mean".
 * cg3d_pool_map: for every input row i and each of the 27 lattice candidates
 *   o = (floor(c_i / out_stride) + d) * out_stride, d in {-1,0,1}^3, pmap[j*n_in + i] = table
 *   row of o if it exists and |o - c_i| <= half_extent on every axis, else -1.
 *   (MinkowskiAvgPooling(kernel k, stride s) on a tensor-stride-ts map: out_stride = s*ts,
 *    half_extent = (k-1)/2 * ts; reference call sites biresnet.py:109-127.)
 * cg3d_scatter_mean_fwd: out[m,:] = mean over {(j,i): map[j,i]==m} of F[i,:];
 *   cnt float32 [n_out] receives the member count (0 rows stay 0).  J maps of n_in rows.
 *   The quantise-average of ME.SparseTensor(..., UNWEIGHTED_AVERAGE) (cagroup_head.py:257-271)
 *   is J == 1 with map == `inverse` from cg3d_coord_map_build.
 * cg3d_scatter_mean_bwd: dF[i,:] = sum_j dout[map[j,i],:] / cnt[map[j,i]].
 * ---------------------------------------------------------------------------------------- */
int cg3d_pool_map(const int32_t *in_coords, int64_t n_in, int32_t out_stride, int32_t half_extent,
                  const uint64_t *keys, const int32_t *vals, int64_t cap,
                  int32_t *pmap, cg3d_stream_t stream);
int cg3d_scatter_mean_fwd(const float *F, const int32_t *map, int32_t J, float *out, float *cnt,
                          int64_t n_in, int64_t n_out, int32_t c, cg3d_stream_t stream);
int cg3d_scatter_mean_bwd(const float *dout, const float *cnt, const int32_t *map, int32_t J,
                          float *dF, int64_t n_in, int64_t n_out, int32_t c, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Grouped BatchNorm (+ residual) (+ activation) over the rows of a sparse tensor
 * (ME.MinkowskiBatchNorm + MinkowskiReLU/ELU + the residual add of biresnet.py:33-50,78-103; one group
 * per class branch in cagroup_head.py:117-127).  Rows are split into contiguous groups by the chunk
 * table `chunks` int32 [nchunk,3] = (group, first row, row count); a chunk never straddles two groups.
 *   act: 0 none, 1 ReLU, 2 ELU(alpha=1).
 *   Statistics live in ONE zero-based fp32 table per layer, sums float32 [CG3D_BN_SLOTS][2][G][C]; every producer ADDS
 *   its partial sums to one of the slots with fp32 atomics (a workgroup picks slot = its index % CG3D_BN_SLOTS: with a
 *   single slot the ~1 200 workgroups of a layer queue on 2 C addresses), the consumers add the slots up, and the caller
 *   zero-fills the table (round 3: the per-chunk partials + fp64 finalise kernel of rounds 1-2 were 130 launches of ~10 us
 *   per step).  Below, sums[0] / sums[1] stand for the slot sums:
 * cg3d_bn_sums:       sums[0][g] += sum of the rows of group g, sums[1][g] += sum of their squares.  For G == 1 the producing
 *                     convolution fills the same table in its epilogue (`stats` of cg3d_spconv_tile_fwd).
 * cg3d_bn_apply_sums: cg3d_bn_apply with mean = sums[0]/n, var (biased) = sums[1]/n - mean^2 (derived in fp64 by every
 *                     thread for its own channels; n = group_n[g]); also WRITES mean / var float32 [G,C] (the backward pass
 *                     reads them) and, when `running_mean`/`running_var` (float32 [G,C]) are not NULL, updates them like
 *                     nn.BatchNorm1d: r = (1-momentum)*r + momentum*stat, the variance unbiased (n/(n-1));
 *                     `num_batches_tracked` (int64 [G], may be NULL) is incremented.
 * cg3d_bn_apply:  y = act(gamma[g]*(x - mean[g])*rsqrt(var[g]+eps) + beta[g] + residual)   (residual may be NULL), mean /
 *                 var given (evaluation: the running statistics);
 *                 Y16 / dX16 (may be NULL): the same rows again as bf16 (RNE) -- the copy the next convolution (forward:
 *                 its input, backward: its output gradient) gathers from at precision 2, written while the fp32 values
 *                 are in registers instead of by a separate cg3d_to_bf16 pass
 * cg3d_bn_bwd_sums: with dz = dy * act'(y), xhat = (x-mean)*rsqrt(var+eps):
 *                 dsums[0][g] += sum(dz) (= dbeta), dsums[1][g] += sum(dz * xhat) (= dgamma); the same slot table layout,
 *                 zero-filled by the caller
 * cg3d_bn_bwd_apply:  dx = gamma*invstd*(dz - (dbeta + xhat*dgamma)/n[g])  (batch statistics;
 *                 use_batch_stats == 0: dx = gamma*invstd*dz), dres (may be NULL) = dz;  group_n float32 [G]
 * cg3d_bn_bwd_apply_sums: the same with dbeta / dgamma taken from the table `dsums` (slot sums) and WRITTEN to dbeta /
 *                 dgamma float32 [G,C] (the parameters' gradients)
 * ---------------------------------------------------------------------------------------- */
#define CG3D_BN_SLOTS 16
/* Activations stored as bf16 (the BiResNet launch program of BASELINE.json configs[1], "bf16 backbone"): with this bit set in
 * the `act` argument of cg3d_bn_apply / _apply_sums / _bwd_sums / _bwd_apply / _bwd_apply_sums EVERY row matrix of the call
 * -- X, residual, Y, dY, dX, dRes -- is uint16 [n, c] bf16 rows instead of float32 (c % 8 == 0; Y16 / dX16 must be NULL:
 * Y / dX are the bf16 rows themselves).  The arithmetic is unchanged: operands widened to fp32, statistics and per-channel
 * constants in fp32 / fp64, results rounded to nearest even on the store.  A layer then moves 6-8 bytes per element in the
 * forward (x, residual in; y out) instead of 14-18, and 6 + 8-10 in the backward instead of 12 + 18-22. */
#define CG3D_BN_STORE_BF16 0x100
int cg3d_bn_sums(const float *X, const int32_t *chunks, int64_t nchunk, int32_t G, int32_t c, float *sums,
                 cg3d_stream_t stream);
int cg3d_bn_apply_sums(const float *X, const float *residual, const int32_t *chunks, int64_t nchunk, int32_t G, int32_t c,
                       const float *sums, const float *group_n, float eps, const float *gamma, const float *beta,
                       int32_t act, float *Y, uint16_t *Y16, float *mean, float *var, float *running_mean,
                       float *running_var, int64_t *num_batches_tracked, float momentum, cg3d_stream_t stream);
int cg3d_bn_apply(const float *X, const float *residual, const int32_t *chunks, int64_t nchunk, int32_t c,
                  const float *mean, const float *var, float eps, const float *gamma, const float *beta, int32_t act,
                  float *Y, uint16_t *Y16, cg3d_stream_t stream);
int cg3d_bn_bwd_sums(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk, int32_t G,
                     int32_t c, const float *mean, const float *var, float eps, int32_t act, float *dsums,
                     cg3d_stream_t stream);
int cg3d_bn_bwd_apply(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk,
                      int32_t c, const float *mean, const float *var, float eps, const float *gamma,
                      const float *dbeta, const float *dgamma, const float *group_n, int32_t act,
                      int32_t use_batch_stats, float *dX, uint16_t *dX16, float *dRes, cg3d_stream_t stream);
int cg3d_bn_bwd_apply_sums(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk,
                           int32_t G, int32_t c, const float *mean, const float *var, float eps, const float *gamma,
                           const float *dsums, const float *group_n, int32_t act, int32_t use_batch_stats, float *dX,
                           uint16_t *dX16, float *dRes, float *dbeta, float *dgamma, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * iou3d_nms (boxes are float32 [n,7] = x,y,z,dx,dy,dz,heading, contiguous).
 *   cg3d_boxes_overlap_bev: out[a,b] = rotated-rectangle BEV intersection AREA.
 *   cg3d_boxes_iou_bev:     out[a,b] = overlap / max(sa + sb - overlap, 1e-8).
 *   cg3d_nms: boxes sorted by descending score.  rotated=1 uses the rotated BEV IoU, rotated=0
 *     the axis-aligned BEV IoU (x,y,dx,dy only).  The 64x64 suppression tiles go to
 *     `mask_ws` uint64 [n * ceil(n/64)]; the greedy scan runs ON DEVICE (the reference copies
 *     the mask to the host and scans there, iou3d_nms.cpp:111-132); `keep` int64 [n] receives
 *     the kept indices (ascending), `num_keep` int32 [1] their count -- both device memory.
 *   cg3d_nms_batched: `nseg` independent sorted segments, segment s = boxes
 *     [seg_off[s], seg_off[s+1]); keep holds per-segment LOCAL indices at the segment's offset,
 *     num_keep int32 [nseg].  mask_ws uint64 [sum_s n_s*ceil(n_s/64)] laid out by mask_off int64 [nseg].
 * ---------------------------------------------------------------------------------------- */
int cg3d_boxes_overlap_bev(const float *boxes_a, int64_t na, const float *boxes_b, int64_t nb,
                           float *out, cg3d_stream_t stream);
int cg3d_boxes_iou_bev(const float *boxes_a, int64_t na, const float *boxes_b, int64_t nb,
                       float *out, cg3d_stream_t stream);
/* boxes_iou_bev_cpu (iou3d_nms_api.cpp:16, iou3d_cpu.cpp:232-252): the same IoU on HOST pointers, no stream. */
int cg3d_boxes_iou_bev_cpu(const float *boxes_a, int64_t na, const float *boxes_b, int64_t nb, float *out);
int cg3d_nms(const float *boxes, int64_t n, float thresh, int32_t rotated,
             uint64_t *mask_ws, int64_t *keep, int32_t *num_keep, cg3d_stream_t stream);
int cg3d_nms_batched(const float *boxes, const int64_t *seg_off, const int64_t *mask_off,
                     int32_t nseg, int64_t max_seg, float thresh, int32_t rotated,
                     uint64_t *mask_ws, int64_t *keep, int32_t *num_keep, cg3d_stream_t stream);
/* The reference's two literal entry points (pcdet/ops/iou3d_nms/src/iou3d_nms.h:9-12, iou3d_nms.cpp:90-186):
 *   int nms_gpu(at::Tensor boxes [n,7] sorted by descending score, at::Tensor keep [n] int64 ON THE HOST, float thresh)
 *   int nms_normal_gpu(same)                                   both return the number of kept boxes.
 * Same argument order (boxes, keep, thresh), `keep` is HOST memory, the return value is the count (>= 0) or a negative
 * CG3D_ERR_*; the call blocks the host until `keep` is filled, like the reference (which copies the mask to the host and
 * scans there).  The reference cudaMallocs its mask per call (iou3d_nms.cpp:103,151); here the caller lends
 * `ws`: cg3d_nms_gpu_ws_bytes(n) bytes of DEVICE scratch (mask tiles + device keep list + count).  Thin wrappers over
 * cg3d_nms (rotated = 1 / 0); the product itself calls cg3d_nms / cg3d_nms_batched and leaves `keep` on the device. */
int64_t cg3d_nms_gpu_ws_bytes(int64_t n);
int cg3d_nms_gpu(const float *boxes, int64_t n, int64_t *keep_host, float thresh, void *ws, cg3d_stream_t stream);
int cg3d_nms_normal_gpu(const float *boxes, int64_t n, int64_t *keep_host, float thresh, void *ws, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * kNN (brute force; reference knn_cuda.cu:58-94): for every query new_xyz[b,m,:] the k nearest
 *   of xyz[b,n,:] by squared distance, ascending; a candidate replaces the current worst only
 *   if strictly closer (ties keep the lower index).  1 <= k <= 100.
 *   idx int32 [b,m,k], dist2 float32 [b,m,k].
 *   `ws`: cg3d_knn_ws_bytes(b, n, m, k) bytes of scratch (k == 1: uniform-grid search, exact -- cell sort buffers;
 *   small problems merge partial winners of reference-point
 *   splits through 64-bit atomicMin keys); may be NULL (single pass per query block).
 * ---------------------------------------------------------------------------------------- */
int64_t cg3d_knn_ws_bytes(int32_t b, int32_t n, int32_t m, int32_t k);
int cg3d_knn(int32_t b, int32_t n, int32_t m, int32_t k, const float *xyz, const float *new_xyz,
             int32_t *idx, float *dist2, void *ws, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ball query (reference pybind `pointnet2_batch_cuda.ball_query_wrapper`, pointnet2_api.cpp / ball_query.cpp:20-37,
 *   kernel ball_query_gpu.cu:15-52; Python caller pointnet2_utils.py:205-228): for every query new_xyz[b,m,:] the first
 *   `nsample` rows of xyz[b,n,:] (ascending row index) with squared distance < radius^2; unfilled slots repeat the first
 *   hit; a query without any hit gets zeros.  idx int32 [b,m,nsample].  Same argument order as the reference wrapper.
 * ---------------------------------------------------------------------------------------- */
int cg3d_ball_query(int32_t b, int32_t n, int32_t m, float radius, int32_t nsample, const float *new_xyz,
                    const float *xyz, int32_t *idx, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * sort_vertices (reference sort_vert_kernel.cu:42-134): per box pair, order the <= 8 valid
 *   polygon vertices (of m candidates, m == 24 in the reference) anticlockwise.
 *   vertices float32 [b,n,m,2], mask uint8/bool [b,n,m], num_valid int32 [b,n] -> idx int32 [b,n,9].
 * ---------------------------------------------------------------------------------------- */
int cg3d_sort_vertices(int32_t b, int32_t n, int32_t m, const float *vertices, const uint8_t *mask,
                       const int32_t *num_valid, int32_t *idx, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Points strictly inside (rotated) boxes -- find_points_in_boxes of
 * pcdet/models/dense_heads/target_assigner/cagroup3d_assigner.py:9-36 (used by assign_semantic :132-152, the vote
 * targets cagroup_head.py:418-452 and the bench's forced selection).
 *   points float32 [n,3], boxes float32 [g,7] (x,y,z,dx,dy,dz,heading) -> inside uint8 [n,g].
 *   point_seg int32 [n] / box_seg int32 [g] (both or neither NULL): a point only counts for boxes of its segment (scene).
 * ---------------------------------------------------------------------------------------- */
int cg3d_points_in_boxes(const float *points, int64_t n, const float *boxes, int32_t g, const int32_t *point_seg,
                         const int32_t *box_seg, uint8_t *inside, cg3d_stream_t stream);

/* ----------------------------------------------------------------------------------------
 * FCOS-style target assignment of the class-map points, all classes (and scenes) at once.
 * Replaces: CAGroup3DAssigner.assign, dense_heads/target_assigner/cagroup3d_assigner.py:62-130 with compute_centerness
 *   :39-46 (the per-class loop of [n, m, 7] face-distance tensors and their [n, m] companions: ~70 tensor launches).
 *   points float32 [n,3]; pt_cls / pt_scene int64 [n] (class map / scene of every point; pt_scene and gt_scene both NULL =
 *   one scene); gt float32 [m,7] boxes, gt_cls / gt_scene int64 [m].
 * cg3d_fcos_centerness: cness float32 [n,m] = centerness of point i in box j -- sqrt(min/max * min/max * min/max) of the
 *   face distances in the box frame, the reference's operation order -- where the pair competes (strictly inside, same
 *   class, same scene), -1 elsewhere.
 * cg3d_fcos_assign: with kth float32 [m] = the k-th largest value of every column of `cness` (k = min(points on the box's
 *   map, TOPK + 1); taken by the caller), point i is positive for the smallest-volume box j with cness[i,j] > kth[j]
 *   (ties -> the lower j): labels int64 [n] (class of the box, -1 = negative), ctr_t float32 [n] and box_t float32 [n,7]
 *   its centerness / box targets (unspecified for negatives).
 * ---------------------------------------------------------------------------------------- */
int cg3d_fcos_centerness(const float *points, const int64_t *pt_cls, const int64_t *pt_scene, int64_t n, const float *gt,
                         const int64_t *gt_cls, const int64_t *gt_scene, int32_t m, float *cness, cg3d_stream_t stream);
int cg3d_fcos_assign(const float *cness, const float *kth, int64_t n, const float *gt, const int64_t *gt_cls, int32_t m,
                     float *ctr_t, float *box_t, int64_t *labels, cg3d_stream_t stream);

/* ----------------------------------------------------------------------------------------
 * Fused sigmoid focal loss with per-row weights.
 * Replaces: py_sigmoid_focal_loss, pcdet/utils/loss_utils.py:903-961 (the element-wise torch chain FocalLoss :964-1040
 *   runs for cagroup_head.py:520-531), including the -1 -> background rewrite of FocalLoss.forward :1024.
 *   pred float32 [n,c] logits; label int32 [n] (class id in [0,c) = foreground, anything else = background row);
 *   row_w float32 [n] (the avg_factor / per-scene normaliser folded into a row weight).
 *   fwd: partial float32 [cg3d_focal_loss_nblocks(n,c)] per-block sums of row_w[i] * loss[i,a] -- the caller adds them.
 *   bwd: dpred[i,a] = gscale[0] * row_w[i] * d loss[i,a] / d pred[i,a]; gscale = device pointer to the upstream scalar.
 * ---------------------------------------------------------------------------------------- */
int32_t cg3d_focal_loss_nblocks(int64_t n, int32_t c);
int cg3d_focal_loss_fwd(const float *pred, const int32_t *label, const float *row_w, int64_t n, int32_t c, float gamma,
                        float alpha, float *partial, cg3d_stream_t stream);
int cg3d_focal_loss_bwd(const float *pred, const int32_t *label, const float *row_w, const float *gscale, int64_t n,
                        int32_t c, float gamma, float alpha, float *dpred, cg3d_stream_t stream);

/* ----------------------------------------------------------------------------------------
 * Centerness BCE + axis-aligned IoU loss over the positive points of the class maps, fused.
 * Replaces: dense_heads/cagroup_head.py:532-546 (loss_centerness :532-536, loss_bbox :537-546) with `_bbox_pred_to_bbox`
 *   :654-668, iou3d_loss.py:14-95 (axis-aligned form) and axis_aligned_bbox_overlaps_3d, loss_utils.py:419-538 -- ~45
 *   element-wise launches forward, ~80 backward.  ScanNet form (no yaw: 6 face distances per point).
 *   centerness float32 [N], bbox_pred float32 [N,6] = (dx-,dx+,dy-,dy+,dz-,dz+), points float32 [N,3],
 *   ctr_t float32 [N] centerness targets, bbox_t float32 [N,tstride] target boxes (x,y,z,w,l,h,...), scene int64 [N],
 *   n_pos / ctr_denorm float32 [B] per-scene normalisers, pos int64 [npos] rows of the positives.
 *   fwd: partial float32 [cg3d_pos_loss_nblocks(npos)][2]: per block
 *        [0] = sum wc / (n_pos[scene] + eps) * BCE(centerness, ctr_t),
 *        [1] = sum wb * ctr_t / ctr_denorm[scene] * (1 - IoU(box(points, bbox_pred), bbox_t))      (the caller adds the blocks)
 *   bwd: dcenterness[r], dbbox_pred[r, 0:6] for r in pos only (the caller zero-fills both); gscale float32 [2] = the two
 *        upstream scalars (device).
 * cg3d_smooth_l1_rows: sum_i w[i] * sum_j smooth_l1(pred[i,j] - target[i,j]; beta) (SmoothL1Loss reduction 'sum',
 *   loss_utils.py:1042-1123: the vote loss cagroup_head.py:512-519); partial float32 [cg3d_focal_loss_nblocks(n,d)].
 * ---------------------------------------------------------------------------------------- */
int32_t cg3d_pos_loss_nblocks(int64_t npos);
int cg3d_pos_loss_fwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                      const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                      const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                      float *partial, cg3d_stream_t stream);
int cg3d_pos_loss_bwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                      const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                      const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                      const float *gscale, float *dcenterness, float *dbbox_pred, cg3d_stream_t stream);
int cg3d_smooth_l1_rows_fwd(const float *pred, const float *target, const float *w, int64_t n, int32_t d, float beta,
                            float *partial, cg3d_stream_t stream);
int cg3d_smooth_l1_rows_bwd(const float *pred, const float *target, const float *w, const float *gscale, int64_t n,
                            int32_t d, float beta, float *dpred, cg3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser step of the training loop (reference tools/train_utils/train_utils.py:40-47: clip_grad_norm_ + AdamW.step()):
 * gradient scaling by the clip coefficient and the AdamW update of every parameter in one launch.
 *   table int64 [nrows,5] = (param address, exp_avg address, exp_avg_sq address, first element, element count <= 2^31) --
 *   one row per chunk of a contiguous fp32 parameter; pid int32 [nrows] = parameter id of the row; grads int64 [nparams] =
 *   this step's gradient addresses; clip: device scalar or NULL (g <- g * *clip, not written back).
 *   p -= lr*wd*p;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps),
 *   bc1 = 1 - beta1^t, bc2 = 1 - beta2^t supplied by the caller. */
/* clip_grad_norm_ (tools/train_utils/train_utils.py:40-47) over the same chunk table, before cg3d_adamw_step:
 *   *norm = sqrt(sum over every gradient element of g^2)  (chunk sums in double),  *coef = min(max_norm / (*norm + 1e-6), 1)
 *   -- the `clip` argument of cg3d_adamw_step.  scratch: one caller-owned double (zeroed by the callee). */
int cg3d_grad_norm_clip(const int64_t *table, const int32_t *pid, int64_t nrows, const int64_t *grads, float max_norm,
                        double *scratch, float *norm, float *coef, cg3d_stream_t stream);
int cg3d_adamw_step(const int64_t *table, const int32_t *pid, int64_t nrows, const int64_t *grads, const float *clip,
                    float lr, float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                    float bias_correction2, cg3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CAGROUP3D_HIP_H */
