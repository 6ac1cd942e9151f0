/*
 * dgdm_hip.h -- C ABI of libdgdm_hip.so: the MI355X (gfx950) kernels under the DGDM hot path.
 *
 * The reference (danieleschmidt/dgdm-histopath-lab) has no FFI layer: its boundary is the Python
 * class DGDMModel (models/dgdm_model.py:37) and every device kernel it runs comes from
 * torch / torch-geometric.  This library is the build's own layer *below* that class; each entry
 * point cites the reference code whose arithmetic it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers to DEVICE memory (hipMalloc'ed or torch CUDA tensors), sizes as integers,
 *     trailing `void* stream` = hipStream_t (NULL = default stream);
 *   - returns DGDM_OK (0) or a negative DGDM_ERR_* code; never allocates, never synchronises;
 *     scratch memory is passed in by the caller and its size is given by the matching
 *     *_workspace_bytes() query.  The library keeps ONE piece of state: the dropout seed epoch, a
 *     uint32 counter in device memory, one per device (see dgdm_seed_epoch_advance).  Everything
 *     else is re-entrant per stream;
 *   - float data is fp32 row-major; node/edge ids inside CSR structures are int32; the edge list
 *     at the boundary is int64 [2,E] exactly as the reference holds it (graph_layers.py:77-81);
 *   - arguments are checked on the host (null pointers, negative sizes, unsupported widths) before
 *     any launch.
 */
#ifndef DGDM_HIP_H
#define DGDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGDM_API __attribute__((visibility("default")))

enum {
  DGDM_OK = 0,
  DGDM_ERR_INVALID_ARG = -1,   /* null pointer, negative size, bad enum */
  DGDM_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels are built for */
  DGDM_ERR_WORKSPACE = -3,     /* workspace too small */
  DGDM_ERR_LAUNCH = -4         /* hipGetLastError() != hipSuccess after a launch */
};

/* An "amax slot" (operand maxima of the fp16 hi+lo GEMMs, K3'') is a group of DGDM_AMAX_WAYS uint32 words spaced
 * DGDM_AMAX_STRIDE words apart: producers spread their atomic maxima over the ways, consumers take the maximum of the ways.
 * A group therefore spans DGDM_AMAX_WAYS * DGDM_AMAX_STRIDE words; all of them zero before the first producer runs. */
#define DGDM_AMAX_WAYS 32
#define DGDM_AMAX_STRIDE 64

/* activation ids shared by the fused row kernels */
enum { DGDM_ACT_NONE = 0, DGDM_ACT_GELU = 1, DGDM_ACT_RELU = 2, DGDM_ACT_SILU = 3, DGDM_ACT_ELU = 4 };   /* ELU: alpha = 1 (nn.ELU(), encoders.py:62,207) */

/* Bumped whenever an exported signature changes (round 5 changed dgdm_segment_mse_bwd / dgdm_pool_score_bwd without doing so): a binding
 * written for one version must refuse a library that reports another -- a shifted argument list is an invalid-stream launch, not an
 * error code.  _lib.load() / open_library() compare it with _lib.ABI_VERSION. */
#define DGDM_ABI_VERSION 2
DGDM_API int dgdm_abi_version(void);
DGDM_API const char* dgdm_error_string(int code);

/* Dropout seed epoch.  Every dropout entry point takes its seed by value; forward and backward of one
 * site agree because they are given the same value.  A launch recorded in a HIP graph replays with the
 * recorded value, so the kernels additionally XOR in a per-device counter held in device memory:
 *   effective seed = seed ^ (epoch * 0x9E3779B9).
 * dgdm_seed_epoch_advance enqueues epoch += 1 (record it once per training step, before the forward, in
 * the captured graph); dgdm_seed_epoch_set enqueues epoch = value.  The counter starts at 0, where the
 * effective seed equals the seed: callers that never touch it see no change.
 * This counter is the library's only state: one uint32 per device, shared by every stream and every caller of
 * that device (two models trained in one process advance the same epoch: their dropout streams stay distinct
 * through their seeds, but neither replays bit-identically if the other runs in between). */
DGDM_API int dgdm_seed_epoch_advance(void* stream);
DGDM_API int dgdm_seed_epoch_set(uint32_t value, void* stream);

/* Input checks of DGDMModel.forward (models/dgdm_model.py:646-690): one pass over the node features x (x_numel
 * contiguous floats, 16-byte aligned) and one over edge_index (edge_numel int64 ids).  flags4[0]: NaN in x, [1]: inf
 * in x, [2]: an edge id > num_nodes - 1, [3]: an edge id < 0 (each 0 or 1); the host reads the four words back once. */
DGDM_API int dgdm_validate_inputs(const float* x, int64_t x_numel, const int64_t* edge_index, int64_t edge_numel,
                                  int64_t num_nodes, uint32_t* flags4, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K1  edge list -> CSR.   Replaces the index preparation of GraphConvolution.forward
 * (core/graph_layers.py:76-84: add_self_loops, degree) and fixes the scatter-add order of
 * MessagePassing.propagate (graph_layers.py:92) to "ascending edge id inside each row".
 *
 * edge_index: int64 [2,E] row-major (row 0 = source, row 1 = destination).
 * by_src = 0: rows are destinations, col = source ids   (forward aggregation)
 * by_src = 1: rows are sources,      col = destination ids (backward aggregation)
 * add_loops != 0: a loop edge (i,i) with edge id E+i is appended for every node (always the last
 *   entry of row i).  n_entries = E + (add_loops ? N : 0).
 * Outputs: rowptr int32 [N+1], col int32 [n_entries], eid int32 [n_entries] (original edge id of
 *   each entry).  Bit-exact against oracle/csr_oracle.py::csr_by_key.
 * Edges with an endpoint outside [0,N) are ignored (never dereferenced): the sync-free top-k
 * pooling marks dropped edges that way instead of compacting the list (graph_layers.py:322-324),
 * so col/eid are allocated for n_entries but only the first rowptr[N] entries are defined.
 */
DGDM_API size_t dgdm_csr_build_workspace_bytes(int64_t E, int32_t N, int32_t add_loops);
DGDM_API int dgdm_csr_build(const int64_t* edge_index, int64_t E, int32_t N, int32_t add_loops, int32_t by_src,
                            int32_t* rowptr, int32_t* col, int32_t* eid,
                            void* workspace, size_t workspace_bytes, void* stream);

/* GCN symmetric normalisation (graph_layers.py:80-84): deg = in-degree from the by-destination
 * rowptr (self loops included when they were added), dinv = deg^-1/2 (0 where deg == 0). */
DGDM_API int dgdm_gcn_dinv(const int32_t* rowptr_dst, int32_t N, float* dinv, void* stream);
/* w[p] = dinv[row(p)] * dinv[col[p]] for every CSR entry p (works for either orientation). */
DGDM_API int dgdm_csr_edge_weights(const int32_t* rowptr, const int32_t* col, const float* dinv, int32_t N,
                                   float* w, void* stream);

/* Everything one edge list needs, in one pipeline of five launches: both CSR orientations (_dst = by_src 0, _src =
 * by_src 1), dinv and the GCN weight w of every entry -- bit for bit what two dgdm_csr_build calls, dgdm_gcn_dinv and
 * two dgdm_csr_edge_weights calls produce (seventeen launches; a training step builds seven such sets). */
DGDM_API size_t dgdm_csr_build_pair_workspace_bytes(int64_t E, int32_t N, int32_t add_loops);
/* Byte offset, inside the workspace of dgdm_csr_build_pair, of one int32 status word the call zeroes and its scatter kernels
 * OR into: bit 0 = an edge's slot fell outside its row (cursor >= count), bit 1 = a row's extent fell outside the arrays.  The
 * offending writes are skipped, so inconsistent counters (e.g. a fill that did not re-execute in a graph replay) surface as
 * this flag instead of an out-of-bounds write.  0 after a healthy build; read it whenever a host sync is acceptable. */
DGDM_API size_t dgdm_csr_build_pair_status_offset(int64_t E, int32_t N, int32_t add_loops);
/* Long rows.  dgdm_spmm* gives every row to one wavefront, which walks the row's entries four at a time: fine for tissue graphs
 * (kNN: <= ~20 neighbours), a 40x slowdown for a hub of 5000 neighbours (profiles/r03_gather_skew.txt; the builder's own
 * per-row ordering pass is quadratic in the row length and was 400x slower on that graph).  dgdm_csr_build_pair can
 * therefore leave, per orientation, a table of the rows longer than DGDM_SPMM_LONG_ROW entries; dgdm_spmm* (argument `long_rows`)
 * then skips them in the row pass and hands every DGDM_SPMM_SEGMENT entries of such a row to a wave of its own; the waves leave
 * partial sums, and the LAST one to arrive (an arrival counter per row) adds them in segment order and applies the epilogue --
 * one launch, fixed summation order, bitwise reproducible.  A table is dgdm_spmm_long_table_words(n_entries) int32 words:
 * [count, slots, {row, first slot} x item_cap, arrival counters x item_cap]; partial sums need dgdm_spmm_long_slot_cap(n_entries)
 * slots of `ld` floats (ld >= the widest C used with the table).  n_entries = E + (add_loops ? N : 0). */
#define DGDM_SPMM_LONG_ROW 128
#define DGDM_SPMM_SEGMENT 64
typedef struct DgdmLongRows {
  int32_t* table;   /* device, written by dgdm_csr_build_pair (one orientation) */
  float* partial;   /* device, slot_cap * ld floats of scratch */
  int64_t ld;
  int32_t item_cap, slot_cap;
} DgdmLongRows;
DGDM_API int32_t dgdm_spmm_long_item_cap(int64_t n_entries);
DGDM_API int32_t dgdm_spmm_long_slot_cap(int64_t n_entries);
DGDM_API size_t dgdm_spmm_long_table_words(int64_t n_entries);
DGDM_API int dgdm_csr_build_pair(const int64_t* edge_index, int64_t E, int32_t N, int32_t add_loops,
                                 int32_t* rowptr_dst, int32_t* col_dst, int32_t* eid_dst, float* w_dst,
                                 int32_t* rowptr_src, int32_t* col_src, int32_t* eid_src, float* w_src, float* dinv,
                                 void* workspace, size_t workspace_bytes, int32_t* long_table_dst, int32_t* long_table_src,
                                 int32_t long_item_cap, void* stream);   /* long_table_*: nullable pair, see "Long rows" */
/* The index set of the SAME edge list (add_loops = 1, every edge id < n_old) over n_new >= n_old nodes, derived from the set
 * built for n_old nodes by ONE copying launch instead of a build: rows < n_old keep their entries, degrees and weights; a node
 * in [n_old, n_new) has its self loop only (eid E + node, dinv 1, weight 1), appended in node order.  Bit for bit what
 * dgdm_csr_build_pair(edge_index, E, n_new, 1, ...) writes into the entries in use.  This is the reference's decoder (D10,
 * core/graph_layers.py:420,453: level j convolves its n_j nodes with the edge list of level j + 1, whose ids are < n_{j+1});
 * ea_hat (nullable, [n_old, ea_dim] aggregated edge attributes) is padded with zero rows into ea_hat_out [n_new, ea_dim].
 * The output arrays hold entry_capacity >= E + n_new entries; long-row tables of the n_old set stay valid for the extended
 * one (no appended row is long). */
DGDM_API int dgdm_csr_extend(const int32_t* rowptr_dst, const int32_t* col_dst, const int32_t* eid_dst, const float* w_dst,
                             const int32_t* rowptr_src, const int32_t* col_src, const int32_t* eid_src, const float* w_src,
                             const float* dinv, const float* ea_hat, int32_t ea_dim, int64_t E, int32_t n_old, int32_t n_new,
                             int64_t entry_capacity, int32_t* rowptr_dst_out, int32_t* col_dst_out, int32_t* eid_dst_out,
                             float* w_dst_out, int32_t* rowptr_src_out, int32_t* col_src_out, int32_t* eid_src_out,
                             float* w_src_out, float* dinv_out, float* ea_hat_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2  CSR segmented gather-reduce  Y[r,:] = sum_{p in row r} w[p] * X[col[p],:]  (+ bias).
 * Replaces MessagePassing.propagate's gather by edge_index[0] + `norm * msg` + scatter-add by
 * edge_index[1] (graph_layers.py:92,99-110); with the by-source CSR it is the backward pass.
 * No atomics: one wavefront (or a sub-wave lane group) owns a destination row and reduces its
 * entries in CSR order, so results are bitwise reproducible.
 *   X [table_rows, C] with leading dimension ldx (floats), Y [N, C] with ldy; C % 4 == 0, C <= 1024,
 *   ldx/ldy % 4 == 0, 16-byte aligned bases.
 *   entries whose col[p] >= table_rows contribute zero (used to aggregate edge attributes by
 *   `eid`, where the appended self-loop entries have no attribute row: repair R1).
 *   bias: nullable [C].   accumulate != 0: Y += result instead of Y = result.
 */
DGDM_API int dgdm_spmm(const int32_t* rowptr, const int32_t* col, const float* w,
                       const float* X, int64_t ldx, int32_t table_rows,
                       float* Y, int64_t ldy, int32_t N, int32_t C,
                       const float* bias, int32_t accumulate, const DgdmLongRows* long_rows, void* stream);
/* long_rows (all three entry points): nullable HOST struct for the orientation of `rowptr` (see "Long rows" above).
 * Y[r, 0:C) as dgdm_spmm (no bias, no accumulate) and Y[r, C:C+Ct) = tail[r, :] in the same pass: the operand
 * [A_hat x | EA_hat] of a graph convolution's single contraction (graph_layers.py:99-110) without a separate copy of the
 * per-graph edge-attribute aggregate.  Ct % 4 == 0, ldt % 4 == 0, ldy >= C + Ct. */
/* Y = dgdm_spmm(...) + addend (addend [N, C], leading dimension lda; may not alias Y): the backward of a graph convolution
 * whose input also feeds a residual connection (DynamicGraphLayer, graph_layers.py:233-245) adds the residual's gradient
 * while it writes the scattered one, instead of leaving the sum to a separate element-wise pass. */
DGDM_API int dgdm_spmm_add(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx,
                           int32_t table_rows, const float* addend, int64_t lda, float* Y, int64_t ldy, int32_t N, int32_t C,
                           const DgdmLongRows* long_rows, void* stream);
DGDM_API int dgdm_spmm_concat(const int32_t* rowptr, const int32_t* col, const float* w, const float* X, int64_t ldx,
                              int32_t table_rows, const float* tail, int64_t ldt, int32_t Ct, float* Y, int64_t ldy,
                              int32_t N, int32_t C, uint32_t* amax, const DgdmLongRows* long_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4  fused variable-length spatial attention (head dim 16), forward.
 * Replaces SpatialAttention.compute_spatial_bias + MultiHeadAttention.forward's
 * QK^T/sqrt(d) + bias -> softmax -> .V (core/attention.py:261-283,135-157) for a whole batch in
 * one launch; the [H,N,N] score tensor is never materialised.
 *   Q,K,V: [N_tot, H*16] fp32 with row stride ld (floats) -- e.g. three column slices of one
 *          [N_tot, 3*H*16] projection output; head h occupies columns h*16..h*16+15.
 *   pos:   [N_tot, 2] raw coordinates; bias = -|pos_q - pos_k| * inv_tau  (attention.py:274-281).
 *   ptr:   int32 [B+1] DEVICE array of per-graph node offsets; attention never crosses graphs.
 *   num_q_tiles = sum_g ceil(n_g / dgdm_spatial_attn_q_tile_rows()), computed by the host.
 *   O:     [N_tot, H*16] (row stride ldo).   lse2: [H, N_tot] log2-domain log-sum-exp of the
 *          scaled+biased scores (m + log2 l), consumed by the backward kernels.
 *   drop_p, seed: dropout on the attention weights (attention.py:154), applied to softmax(S) before
 *          the product with V; the mask is a counter hash of (seed, graph, head, q, k) that the
 *          backward kernels regenerate (pass the same drop_p/seed).  drop_p = 0 disables it.
 */
DGDM_API int32_t dgdm_spatial_attn_q_tile_rows(void);
DGDM_API int dgdm_spatial_attn_fwd(const float* Q, const float* K, const float* V, int64_t ld, const float* pos,
                                   const int32_t* ptr, int32_t B, int32_t num_q_tiles, int32_t N_tot, int32_t H,
                                   float scale, float inv_tau, float drop_p, uint32_t seed, float* O, int64_t ldo,
                                   float* lse2, void* stream);
/* same, with an explicit tiling variant (0 = default) -- tuning/bench use only */
DGDM_API int dgdm_spatial_attn_fwd_variant(const float* Q, const float* K, const float* V, int64_t ld, const float* pos,
                                           const int32_t* ptr, int32_t B, int32_t num_q_tiles, int32_t N_tot, int32_t H,
                                           float scale, float inv_tau, float drop_p, uint32_t seed, float* O, int64_t ldo,
                                           float* lse2, int32_t variant, void* stream);

/* K4 backward: dQ, dK, dV of the fused spatial attention (what autograd derives from
 * core/attention.py:135-157 in the reference, with P recomputed per tile instead of stored).
 *   O, dO: [N_tot, H*16] (row stride ldo); lse2 from the forward; dQ/dK/dV: [N_tot, H*16] with
 *   row stride ldg (e.g. three column slices of one [N_tot, 3*H*16] gradient buffer).
 *   delta_ws: float [H, N_tot] scratch (rowsum(dO*O), written by the dQ pass, read by the dK/dV pass).
 * Two launches, no atomics: results are bitwise reproducible. */
DGDM_API int dgdm_spatial_attn_bwd(const float* Q, const float* K, const float* V, int64_t ld, const float* O,
                                   const float* dO, int64_t ldo, const float* pos, const int32_t* ptr, int32_t B,
                                   int32_t num_q_tiles, int32_t N_tot, int32_t H, float scale, float inv_tau,
                                   const float* lse2, float drop_p, uint32_t seed, float* dQ, float* dK, float* dV, int64_t ldg,
                                   float* delta_ws, void* stream);
/* the two passes of dgdm_spatial_attn_bwd as separate entry points (pass 2 must follow pass 1 on
 * the same stream): lets a caller time or overlap them individually. */
DGDM_API int dgdm_spatial_attn_bwd_dq(const float* Q, const float* K, const float* V, int64_t ld, const float* O,
                                      const float* dO, int64_t ldo, const float* pos, const int32_t* ptr, int32_t B,
                                      int32_t num_q_tiles, int32_t N_tot, int32_t H, float scale, float inv_tau,
                                      const float* lse2, float drop_p, uint32_t seed, float* dQ, int64_t ldg, float* delta_ws,
                                      void* stream);
DGDM_API int dgdm_spatial_attn_bwd_dkv(const float* Q, const float* K, const float* V, int64_t ld, const float* dO,
                                       int64_t ldo, const float* pos, const int32_t* ptr, int32_t B, int32_t num_q_tiles,
                                       int32_t N_tot, int32_t H, float scale, float inv_tau, const float* lse2,
                                       const float* delta_ws, float drop_p, uint32_t seed, float* dK, float* dV, int64_t ldg,
                                       void* stream);

/* Head-mean attention weights per graph (what MultiHeadAttention returns with need_weights,
 * core/attention.py:171-173; DGDMModel's `attention_weights` output, dgdm_model.py:360-361).
 * W is one float buffer holding B dense [n_g, n_g] matrices, graph g at element offset
 * w_offsets[g] (int64 DEVICE array [B]).  Uses lse2 from the forward. */
DGDM_API int dgdm_spatial_attn_mean_weights(const float* Q, const float* K, int64_t ld, const float* pos, const int32_t* ptr,
                                            int32_t B, int32_t num_q_tiles, int32_t N_tot, int32_t H, float scale,
                                            float inv_tau, const float* lse2, float* W, const int64_t* w_offsets,
                                            void* stream);

/* K4, split-fp16 path.  Same math as dgdm_spatial_attn_fwd/_bwd, but the products run on the 16-bit
 * matrix pipe with every fp32 operand -- Q', K, V, dO AND the probabilities P and dS the kernels form --
 * carried as hi+lo halfs (csrc/attn_h.hpp: ~21 significand bits, fp32 accumulation): on gfx950 the fp32
 * MFMA cannot overlap with the softmax's fp32 VALU work, the fp16 MFMA can.  Operands are packed
 * once per launch into graph-block-aligned images (64 rows per block, zero padded; block count =
 * num_q_tiles of the fp32 path) that the kernels stage with direct-to-LDS DMA:
 *   row image        R[blk][H][4 tiles][4 chunks][16 rows][8] halfs: per row 16 hi | 16 lo halfs, chunk-major inside a
 *                    16-row tile (csrc/attn_h.hpp r_off: the order that makes row reads AND transposed reads of the LDS copy
 *                    free of bank conflicts).  Operands needed transposed are read with ds_read_b64_tr_b16: no second image.
 * dgdm_attn_pack_bytes(num_blocks, H, which): buffer sizes (which: 0 R, 2 positions
 * [blk][2][64] fp32 (planar x | y), 3 per-row scalars [blk][H][64] fp32).
 * dgdm_attn_pack: tensor z (z < ntensors) = columns [col0 + z*cstride, +H*16) of X [N_tot, *]; tensor 0
 * is scaled by scale0 (Q: log2(e)/sqrt(d)) times *scale_dev (nullable device scalar); R holds ntensors row images back to
 * back.  pos / pos_b (nullable pair): block-aligned positions TIMES pos_scale (= log2(e)/tau: the kernels
 * add the plain Euclidean distance of these to the negated log2-domain scores).  O / ndelta_b (nullable pair): ndelta =
 * -rowsum(X_0 * O) for the backward (X_0 = dO); lse_in / lse_out (nullable pair, only with O): lse_out = 8 - lse_in, minus the
 * log-sum-exp the backward kernels subtract (they carry P' = 2^8 P so that the weights of a near-uniform row over 10^4..10^5
 * keys stay inside fp16's normal range; the factor leaves with the final scale). */
DGDM_API size_t dgdm_attn_pack_bytes(int32_t num_blocks, int32_t H, int32_t which);
/* fp16 range guard: out2 = {alpha, 1/alpha}, alpha = 2^k with alpha*max|x| in (target/2, target] (x: n
 * contiguous floats, n % 4 == 0).  The backward is linear in dO, so dO is packed as alpha*dO (scale_dev =
 * out2) and the kernels multiply their results by out2[1]: gradients of 1e-6 would otherwise sit below
 * fp16's normal range.  Device-side only, no host round trip. */
DGDM_API size_t dgdm_amax_scale_workspace_bytes(void);
DGDM_API int dgdm_amax_pow2_scale(const float* x, int64_t n, float target, float* out2, void* workspace, size_t workspace_bytes,
                                  void* stream);
DGDM_API int dgdm_attn_pack(const float* X, int64_t ld, int32_t col0, int32_t cstride, int32_t ntensors, float scale0,
                            const float* scale_dev, const int32_t* ptr, int32_t B, int32_t num_blocks, int32_t H, void* R,
                            const float* pos, float pos_scale, float* pos_b, const float* O, int64_t ldo,
                            float* ndelta_b, const float* lse_in, float* lse_out, void* stream);
/* forward: Rq / Rk / Rv = row images of Q', K, V; pos_b as packed (pre-scaled);
 * O [N_tot, H*16] fp32 (row stride ldo); lse2_b [blk][H][64] (log2-domain log-sum-exp, block layout). */
DGDM_API int dgdm_spatial_attn_h_fwd(const void* Rq, const void* Rk, const void* Rv, const float* pos_b, const int32_t* ptr,
                                     int32_t B, int32_t num_blocks, int32_t H, float drop_p, uint32_t seed, float* O,
                                     int64_t ldo, float* lse2_b, int32_t variant, void* stream);
/* backward: Rg = row image of dO, ndelta_b and lse_adj_b (= lse_out) from a second dgdm_attn_pack call
 * (ntensors = 1, scale0 = 1, O and the forward's lse2_b given).  dQ/dK/dV fp32 [N_tot, H*16], row
 * stride ldg.  Same drop_p/seed as the forward.  grad_scale2 = the {alpha, 1/alpha} pair dO was packed
 * with (dgdm_amax_pow2_scale, target 0.25).  variant: tiling selector (0 = default).  Two launches, no atomics. */
DGDM_API int dgdm_spatial_attn_h_bwd_dq(const void* Rq, const void* Rk, const void* Rv, const void* Rg,
                                        const float* pos_b, const float* lse_adj_b, const float* ndelta_b, const int32_t* ptr, int32_t B,
                                        int32_t num_blocks, int32_t H, float scale, float drop_p, uint32_t seed,
                                        const float* grad_scale2, float* dQ, int64_t ldg, int32_t variant, void* stream);
DGDM_API int dgdm_spatial_attn_h_bwd_dkv(const void* Rq, const void* Rk, const void* Rv, const void* Rg,
                                         const float* pos_b, const float* lse_adj_b, const float* ndelta_b,
                                         const int32_t* ptr, int32_t B, int32_t num_blocks, int32_t H, float drop_p,
                                         uint32_t seed, const float* grad_scale2, float* dK, float* dV, int64_t ldg, int32_t variant,
                                         void* stream);

/* backward in ONE pass (round 4): the key-stationary pass also produces dQ.  A workgroup owns a key SUPER-block -- 4 consecutive
 * 64-key blocks of one graph, one per wave -- and one head; per query block of its graph it writes its 64 x 16 fp32 share of dQ
 * as a partial tile, and a second launch sums the super-blocks' tiles of every query block in order (no atomics, bitwise
 * repeatable) and applies the scale.  The scores, the distance, exp2 and the dropout word are evaluated once instead of twice.
 * Super-blocks are numbered graph by graph (dgdm_spatial_attn_h_bwd_fused_superblocks gives their count); a call processes
 * [sb_first, sb_first + sb_count) (callers cut the range so that the scratch -- 4 KiB per (super-block, query block of its graph,
 * head) -- stays within their budget and call in ascending order; a later call ADDS to the dQ rows of a graph an earlier call has
 * started); dK / dV rows of those keys are final after the call.  ptr_host: the B + 1 graph offsets in HOST memory (the same
 * values `ptr` holds on the device). */
DGDM_API int32_t dgdm_spatial_attn_h_bwd_fused_superblocks(const int32_t* ptr_host, int32_t B);
DGDM_API size_t dgdm_spatial_attn_h_bwd_fused_workspace_bytes(const int32_t* ptr_host, int32_t B, int32_t H, int32_t sb_first,
                                                              int32_t sb_count);
DGDM_API int dgdm_spatial_attn_h_bwd_fused(const void* Rq, const void* Rk, const void* Rv, const void* Rg, const float* pos_b,
                                           const float* lse_adj_b, const float* ndelta_b, const int32_t* ptr, const int32_t* ptr_host,
                                           int32_t B, int32_t num_blocks, int32_t H, float drop_p, uint32_t seed,
                                           const float* grad_scale2, float* dK, float* dV, int64_t ldg,
                                           int32_t sb_first, int32_t sb_count, void* workspace, size_t workspace_bytes, void* stream);
/* ... and its second stage (same range, same workspace, right behind it on the stream): the partial tiles summed into dQ */
DGDM_API int dgdm_spatial_attn_h_bwd_fused_reduce(const int32_t* ptr, const int32_t* ptr_host, int32_t B, int32_t num_blocks, int32_t H,
                                                  float scale, const float* grad_scale2, float* dQ, int64_t ldg, int32_t sb_first,
                                                  int32_t sb_count, const void* workspace, size_t workspace_bytes, void* stream);

/* Zero-block map (round 5).  The reference feeds raw slide coordinates into -distance / temperature (core/attention.py:261-283 on
 * positions from preprocessing/tissue_graph_builder.py:381-384, pixels): all but a band of (query block, key block) pairs then have
 * weights that are 0.0f in fp32.  dgdm_attn_skip_map_build finds the pairs whose every score lies more than 200 (log2 units) below
 * every row maximum of the query block, from per-block bounds of the packed operands (two small launches); the `_sparse` forms of the
 * forward and of the one-pass backward walk those pairs over -- same bits in every output as the plain forms (the skipped products are
 * exact zeros), a band's worth of work on real slides, nothing skipped for positions in [0, 1).  skip_map == NULL: the plain forms.
 * The map of a forward call must be handed unchanged to its backward and to the backward's reduction.
 * amax_out (nullable, these three calls): a zeroed operand-maximum slot group (the layout dgdm_amax_bits fills) that receives
 * max |O| (forward) / max |dK|, |dV| (backward) and max |dQ| (reduction): the projections that consume these tensors then need no
 * reduction launch of their own. */
DGDM_API size_t dgdm_attn_skip_map_bytes(int32_t num_blocks, int32_t H);
DGDM_API size_t dgdm_attn_skip_map_workspace_bytes(int32_t num_blocks, int32_t H);
DGDM_API int dgdm_attn_skip_map_build(const void* Rq, const void* Rk, const float* pos_b, const int32_t* ptr, int32_t B,
                                      int32_t num_blocks, int32_t H, void* workspace, size_t workspace_bytes, uint32_t* map,
                                      size_t map_bytes, void* stream);
/* Measurement only (bench.py): the scores the kernels evaluate under a map.  counts (3 x uint64, device, ZERO on entry):
 * [0] forward (unmarked query-block x key-block pairs), [1] one-pass backward (unmarked key-super-block x query-block pairs: a
 * super-block runs all its key tiles), [2] every score (sum over graphs of n_g^2 x H). */
DGDM_API int dgdm_attn_skip_map_count(const uint32_t* map, const int32_t* ptr, int32_t B, int32_t num_blocks, int32_t H, uint64_t* counts,
                                      void* stream);
DGDM_API int dgdm_spatial_attn_h_fwd_sparse(const void* Rq, const void* Rk, const void* Rv, const float* pos_b, const int32_t* ptr,
                                            int32_t B, int32_t num_blocks, int32_t H, float drop_p, uint32_t seed, float* O,
                                            int64_t ldo, float* lse2_b, int32_t variant, const uint32_t* skip_map, uint32_t* amax_out,
                                            void* stream);
DGDM_API int dgdm_spatial_attn_h_bwd_fused_sparse(const void* Rq, const void* Rk, const void* Rv, const void* Rg, const float* pos_b,
                                                  const float* lse_adj_b, const float* ndelta_b, const int32_t* ptr,
                                                  const int32_t* ptr_host, int32_t B, int32_t num_blocks, int32_t H, float drop_p,
                                                  uint32_t seed, const float* grad_scale2, float* dK, float* dV, int64_t ldg,
                                                  int32_t sb_first, int32_t sb_count, void* workspace, size_t workspace_bytes,
                                                  const uint32_t* skip_map, uint32_t* amax_out, void* stream);
DGDM_API int dgdm_spatial_attn_h_bwd_fused_reduce_sparse(const int32_t* ptr, const int32_t* ptr_host, int32_t B, int32_t num_blocks,
                                                         int32_t H, float scale, const float* grad_scale2, float* dQ, int64_t ldg,
                                                         int32_t sb_first, int32_t sb_count, const void* workspace,
                                                         size_t workspace_bytes, const uint32_t* skip_map, uint32_t* amax_out,
                                                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5  out = x + sinusoidal_2d_posenc(pos)  for a whole batch.  Replaces
 * SpatialAttention.get_positional_encoding + the add (core/attention.py:225-259,306): positions
 * are min-max normalised with ONE global min/max per graph over both coordinates (+1e-8), C/4
 * frequencies, channels [sin x, cos x, sin y, cos y] interleaved with stride 4.
 *   x: nullable [N, C] (row stride ldx; NULL -> out = pe); pos [N,2]; ptr int32 [B+1] device;
 *   minmax_ws: float [2*B] scratch; out [N, C] (row stride ldo).  C % 4 == 0.
 */
DGDM_API int dgdm_add_posenc(const float* x, int64_t ldx, const float* pos, const int32_t* ptr, int32_t B, int32_t N,
                             int32_t C, float* minmax_ws, float* out, int64_t ldo, uint32_t* amax, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K6/K7  fused row normalisation  y = dropout(act(norm_G(x [+ res]) * gamma + beta)).
 * G = 1: LayerNorm (models/encoders.py:73-83,267-269; LayerNorm(out + residual) at
 * core/graph_layers.py:245 and core/attention.py:325).  G = 8: GroupNorm(8, C) on 2-D [N, C] rows
 * + SiLU + dropout (core/diffusion.py:96-102).  x, res (nullable), y: contiguous [N, C];
 * gamma, beta [C]; mean, rstd: [N*G] saved for the backward.  (C/G) % 4 == 0, C/G <= 1024.
 * Dropout: element e is dropped iff hash(seed, e) < drop_p (16-bit threshold), kept values are
 * scaled by 1/(1-p); the backward recomputes the mask from the same seed.
 * Backward: dx (gradient wrt x and, identically, wrt res), dgamma, dbeta via a fixed-order
 * two-stage reduction through `workspace` (no atomics).
 * `amax` (nullable; also on dgdm_act_dropout_*, dgdm_spmm_concat, dgdm_qsample ...): an amax slot group (see
 * DGDM_AMAX_WAYS) that receives max|out| -- the kernel that produces a GEMM operand keeps its maximum, so the fp16 hi+lo
 * GEMMs (K3'') need no separate reduction launch.  The group must be zero at launch. */
DGDM_API int dgdm_rownorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, int32_t N, int32_t C,
                              int32_t G, float eps, int32_t act, float drop_p, uint32_t seed, float* y, float* mean,
                              float* rstd, uint32_t* amax, void* stream);
DGDM_API size_t dgdm_rownorm_bwd_workspace_bytes(int32_t N, int32_t C, int32_t G);
/* dgamma == dbeta == NULL: only the row partials are written -- `workspace` then starts with [dgdm_rownorm_bwd_slots(N,C,G)][2C]
 * floats whose column sums are dgamma | dbeta, for the caller to reduce later (dgdm_gemm_tn_reduce_many with N = 1, K = 2C,
 * K0 = C takes them in the same launch as a backward pass's weight gradients). */
DGDM_API int64_t dgdm_rownorm_bwd_slots(int32_t N, int32_t C, int32_t G);
DGDM_API int dgdm_rownorm_bwd(const float* x, const float* res, const float* gamma, const float* beta, const float* mean,
                              const float* rstd, const float* dy, int32_t N, int32_t C, int32_t G, int32_t act, float drop_p,
                              uint32_t seed, float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                              uint32_t* amax, void* stream);

/* y = dropout(act(x)) and dx = dy * mask * act'(x) over n contiguous floats (n % 4 == 0):
 * the GELU+dropout after each graph convolution (core/graph_layers.py:233-239) and the ReLU between
 * the levels of the graph U-Net (core/graph_layers.py:418,434).
 * `decide` (nullable; ReLU sites only -- here, dgdm_pool_score_* and dgdm_unpool_add_relu_*): one byte
 * per element, contiguous [N, C]; non-zero = the element passes.  When given, the kernels take the side
 * of the ReLU kink from it instead of from the sign of the pre-activation, forward and backward.  It
 * exists for the parity tests: an element within rounding of zero may fall on either side, which makes
 * gradients incomparable; handing the reference's decisions to the kernels makes both sides
 * differentiate the same piecewise-linear function.  NULL (every product call) = sign of the value. */
DGDM_API int dgdm_act_dropout_fwd(const float* x, int64_t n, int32_t act, float drop_p, uint32_t seed, float* y,
                                  const uint8_t* decide, uint32_t* amax, void* stream);
DGDM_API int dgdm_act_dropout_bwd(const float* x, const float* dy, int64_t n, int32_t act, float drop_p, uint32_t seed,
                                  float* dx, const uint8_t* decide, uint32_t* amax, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense layers with few rows (one row per graph or per timestep): the denoiser's time-embedding MLP
 * (core/diffusion.py:87-91,147-163: Linear -> SiLU -> Linear), the time half of its first Linear folded
 * into a per-graph bias (:165-170), GlobalAttentionPool's query / output projections
 * (models/dgdm_model.py:596-615).  Exact fp32 on the VALU, fixed-order reductions, any M, N; K <= 2048.
 *   dgdm_linear_small_fwd: y[m,n] = act(pre), pre = sum_k x[m,k] w[n,k] + b[n] (b nullable); `pre`
 *                          (nullable, [M, N] row stride ldp) receives the pre-activation for the backward.
 *   dgdm_linear_small_bwd: g = gy * act'(pre) (act = NONE: g = gy, pre may be NULL);
 *                          dx[m,k] = sum_n g[m,n] w[n,k]; dw[n,k] = sum_m g[m,n] x[m,k]; db[n] = sum_m g[m,n];
 *                          each of dx / dw / db is skipped when NULL; lddw lets dw be a column block of a
 *                          larger gradient matrix.
 * K8  dgdm_ddpm_step: one update of DiffusionLayer.sample (core/diffusion.py:255-273) over n floats
 *   (n % 4 == 0):  x0 = (x - sqrt_one_minus_ac * eps) / sqrt_ac;  out = last ? x0 : sqrt_alpha * x0 + sqrt_var * z. */
DGDM_API int dgdm_linear_small_fwd(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* b, int32_t M, int32_t N,
                                   int32_t K, int32_t act, float* y, int64_t ldy, float* pre, int64_t ldp, void* stream);
DGDM_API int dgdm_linear_small_bwd(const float* gy, int64_t ldg, const float* pre, int64_t ldp, int32_t act, const float* x,
                                   int64_t ldx, const float* w, int64_t ldw, int32_t M, int32_t N, int32_t K, float* dx,
                                   int64_t lddx, float* dw, int64_t lddw, float* db, void* stream);
DGDM_API int dgdm_ddpm_step(const float* x, const float* eps, const float* z, int64_t n, float sqrt_one_minus_ac, float sqrt_ac,
                            float sqrt_alpha, float sqrt_var, int32_t last, float* out, void* stream);

/* K8-fused (round 6): ONE step of DiffusionLayer.sample as ONE launch (reference: core/diffusion.py:214-275, the T-step denoise loop):
 *   eps = denoise_net([x | t_emb]);  out = last ? x0 : sqrt_alpha * x0 + sqrt_var * z,  x0 = (x - sqrt_one_minus_ac * eps) / sqrt_ac.
 * Every operation of the step is row-local, so a workgroup takes 32 rows through the three Linear layers, the two GroupNorm(8) + SiLU
 * and the update with the activations in LDS (csrc/sample_step.hip).  x / z / out [N, C] fp32 (row strides ldx / ldz / ldo; z may be
 * NULL when last != 0; out may alias x only if no other launch reads x).  img0 / img1 / img2: the weight images (dgdm_gemm_image_build,
 * kind "forward") of denoise_net[0].weight[:, :C] ([4C, C]: tiles0 = 4C / 32), denoise_net[4].weight ([2C, 4C]: tiles1 = 2C / 32),
 * denoise_net[8].weight ([C, 2C]: tiles2 = C / 32).  bias0 [4C] = the time half of denoise_net[0] applied to this step's t_emb plus its
 * bias (what dgdm_linear_small_fwd gives for the timestep); gn*_w / gn*_b / eps*: the two GroupNorm(8) layers; bias1 [2C], bias2 [C].
 * Eval mode only (no dropout).  C in {128, 256} (dgdm_denoise_ddpm_step_supported), else DGDM_ERR_UNSUPPORTED. */
DGDM_API int32_t dgdm_denoise_ddpm_step_supported(int32_t C);
DGDM_API int dgdm_denoise_ddpm_step(const float* x, int64_t ldx, const float* z, int64_t ldz, int32_t N, int32_t C, const void* img0,
                                    int32_t tiles0, const void* img1, int32_t tiles1, const void* img2, int32_t tiles2, const float* bias0,
                                    const float* gn1_w, const float* gn1_b, float eps1, const float* bias1, const float* gn2_w,
                                    const float* gn2_b, float eps2, const float* bias2, float sqrt_one_minus_ac, float sqrt_ac,
                                    float sqrt_alpha, float sqrt_var, int32_t last, float* out, int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-graph (segment) primitives; graph g owns the contiguous rows [ptr[g], ptr[g+1]) (ptr: int32
 * DEVICE array [B+1]).  They replace the reference's per-graph Python loops with boolean masks
 * (models/dgdm_model.py:419-431, 607-613).
 *   dgdm_segment_bcast_add: out[n,:] = x[n,:] (x nullable) + src[g(n),:]      src [B,C], out [N,C]
 *   dgdm_segment_sum      : out[g,:] = sum_{n in g} x[n,:]  (two fixed-order stages, no atomics)
 */
DGDM_API int dgdm_segment_bcast_add(const float* x, const float* src, const int32_t* ptr, int32_t B, int32_t N, int32_t C,
                                    float* out, void* stream);
DGDM_API size_t dgdm_segment_sum_workspace_bytes(int32_t B, int32_t C);
DGDM_API int dgdm_segment_sum(const float* x, const int32_t* ptr, int32_t B, int32_t C, float* out, void* workspace,
                              size_t workspace_bytes, void* stream);
/* Column norm = nn.BatchNorm1d over the NODES of a batch, fused with the activation + dropout behind it (csrc/colnorm.hip):
 *   y = dropout(act((x - mean_c) rstd_c gamma_c + beta_c)),  x, y [N, C] contiguous, C % 4 == 0.
 * The reference builds it for normalization="batch" (models/encoders.py:95-100,211-219).  training != 0: statistics of the batch
 * (biased variance), running_mean / running_var (nullable) updated in place with `momentum` (unbiased variance), as torch does;
 * training == 0: the running averages are the statistics.  mean / rstd [C] are outputs (the backward reads them).  The backward
 * writes dx [N, C], dgamma, dbeta [C].  workspace: dgdm_colnorm_workspace_bytes(N, C) for both.  Fixed summation order
 * (bitwise repeatable); the dropout mask is the (seed, element index) function of dgdm_act_dropout_*. */
DGDM_API size_t dgdm_colnorm_workspace_bytes(int32_t N, int32_t C);
DGDM_API int dgdm_colnorm_fwd(const float* x, int32_t N, int32_t C, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, int32_t training, float momentum, float eps, int32_t act, float drop_p, uint32_t seed,
                              float* y, float* mean, float* rstd, void* workspace, size_t workspace_bytes, uint32_t* amax, void* stream);
DGDM_API int dgdm_colnorm_bwd(const float* x, const float* dy, int32_t N, int32_t C, const float* gamma, const float* beta, const float* mean,
                              const float* rstd, int32_t training, int32_t act, float drop_p, uint32_t seed, float* dx, float* dgamma,
                              float* dbeta, void* workspace, size_t workspace_bytes, uint32_t* amax, void* stream);

/* Segment max (GlobalMaxPool, models/dgdm_model.py:570-585: out[g] = x[batch == g].max(dim=0)[0]): out [B, C] and arg [B, C] (the
 * row that attains the maximum, the first one on ties; -1 and out = 0 for a graph without rows).  The backward writes the whole
 * dx [N, C]: gout[g][c] at row arg[g][c], zero elsewhere (torch.max(dim)'s gradient).  Any C; x rows at stride ldx. */
DGDM_API size_t dgdm_segment_max_workspace_bytes(int32_t B, int32_t C);
DGDM_API int dgdm_segment_max_fwd(const float* x, int64_t ldx, const int32_t* ptr, int32_t B, int32_t C, float* out, int32_t* arg,
                                  void* workspace, size_t workspace_bytes, void* stream);
DGDM_API int dgdm_segment_max_bwd(const float* gout, const int32_t* arg, const int32_t* ptr, int32_t B, int32_t N, int32_t C, float* dx,
                                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * K7 / K11  element-wise pieces of the diffusion objective and of entity masking, one launch for the whole batch
 * (the reference: framework-op chains inside Python loops over graphs, models/dgdm_model.py:405-445,482-506;
 * core/diffusion.py:123-145).  ptr: int32 DEVICE [B+1] graph offsets; timesteps: int64 DEVICE [B].
 *   dgdm_qsample : out[n,:] = tab_a[t_g] * x[n,:] + tab_b[t_g] * eps[n,:]  (tab_a = sqrt(alphas_cumprod),
 *                  tab_b = sqrt(1 - alphas_cumprod), DEVICE tables [T]); eps = NULL: out = tab_a[t_g] * x (the backward).
 *   dgdm_segment_mse_fwd : loss[0] = (1/B) sum_g mse(pred_g, target_g) (dgdm_model.py:430-433), two fixed-order stages.
 *   dgdm_segment_mse_bwd : dpred[n,:] = gloss[0] * 2 / (B n_g C) * (pred[n,:] - target[n,:])   (gloss: DEVICE scalar); amax (nullable):
 *                  a zeroed operand-maximum slot group that receives max |dpred| (as dgdm_mask_rows' does for its output).
 *   dgdm_mask_rows : out[n,:] = node_map[n] >= 0 ? token[:] : x[n,:]  (entity masking, dgdm_model.py:494-503, with the
 *                  node_map of dgdm_topk_perm over N uniform variates = a uniformly random subset of the masked size).
 * C, F % 4 == 0; float pointers 16-byte aligned. */
DGDM_API int dgdm_qsample(const float* x, const float* eps, const float* tab_a, const float* tab_b, const int64_t* timesteps,
                          const int32_t* ptr, int32_t B, int32_t N, int32_t C, float* out, uint32_t* amax, void* stream);
DGDM_API size_t dgdm_segment_mse_workspace_bytes(int32_t B);
DGDM_API int dgdm_segment_mse_fwd(const float* pred, const float* target, const int32_t* ptr, int32_t B, int32_t N, int32_t C,
                                  float* loss, void* workspace, size_t workspace_bytes, void* stream);
DGDM_API int dgdm_segment_mse_bwd(const float* pred, const float* target, const float* gloss, const int32_t* ptr, int32_t B,
                                  int32_t N, int32_t C, float* dpred, uint32_t* amax, void* stream);
DGDM_API int dgdm_mask_rows(const float* x, const int32_t* node_map, const float* token, int32_t N, int32_t F, float* out,
                            uint32_t* amax, void* stream);

/* K10  GlobalAttentionPool (models/dgdm_model.py:588-615): per graph, ONE query (the projected,
 * 1/sqrt(D)-scaled global token, q_scaled [H*D]) attends over the graph's nodes:
 *   out[g,h,:] = sum_n dropout(softmax_n(q_h . K[n,h,:]))[n] * V[n,h,:]
 * K, V: [N, H*D] with row stride ld (two column blocks of one fused projection buffer);
 * P [N,H]: pre-dropout probabilities saved for the backward; out [B, H*D].  D in {4,8,16,32}.
 * The forward runs over (head, graph, 512-node chunk) in two launches (chunk records, then their fixed-order
 * combination); max_rows = node count of the largest graph of the batch (known to the caller on the host) sizes the
 * grid and the workspace.
 * Backward: dK, dV [N, H*D] (row stride ldg), dq_partial [B, H*D] (sum over graphs = d q_scaled). */
DGDM_API size_t dgdm_attn_pool_fwd_workspace_bytes(int32_t B, int32_t H, int32_t D, int32_t max_rows);
DGDM_API int dgdm_attn_pool_fwd(const float* K, const float* V, int64_t ld, const float* q_scaled, const int32_t* ptr, int32_t B,
                                int32_t H, int32_t D, int32_t max_rows, float drop_p, uint32_t seed, float* P, float* out,
                                void* workspace, size_t workspace_bytes, void* stream);
DGDM_API int dgdm_attn_pool_bwd(const float* K, const float* V, int64_t ld, const float* q_scaled, const int32_t* ptr, int32_t B,
                                int32_t H, int32_t D, float drop_p, uint32_t seed, const float* P, const float* out,
                                const float* dout, float* dK, float* dV, int64_t ldg, float* dq_partial, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3  fp32-MFMA GEMMs for the dense node-feature x weight contractions: every nn.Linear on the
 * path (models/encoders.py:73-91,215; core/graph_layers.py:45-49,146; core/attention.py:44-49;
 * core/diffusion.py:94-104) and its autograd.  Tall-skinny: M = nodes of the batch, N, K <= ~1k.
 * Exact fp32 (v_mfma_f32_32x32x2_f32).  All matrices row-major with explicit row strides.
 *   dgdm_gemm_nt: C[M,N] (+)= A[M,K] . W[N,K]^T + bias[N]      y = Linear(x)       K % 4 == 0
 *   dgdm_gemm_nn: C[M,K] (+)= A[M,N] . W[N,K]                  dx = dy . W         N % 4 == 0
 *   dgdm_gemm_tn: dW[N,K] = dY[M,N]^T . X[M,K]; db[N] = column sums of dY (db nullable)
 *                 split over M into chunks whose partial tiles are summed in chunk order
 *                 (workspace from dgdm_gemm_tn_workspace_bytes; no atomics).  N % 4 == K % 4 == 0.
 */
DGDM_API int dgdm_gemm_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                          int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream);
DGDM_API int dgdm_gemm_nn(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int32_t M, int32_t N,
                          int32_t Kout, int32_t accumulate, void* stream);
DGDM_API size_t dgdm_gemm_tn_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t with_bias);
DGDM_API int dgdm_gemm_tn(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, float* db, int32_t M,
                          int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream);
/* dgdm_gemm_tn with the K columns of dW delivered to two matrices: [0, K0) -> dW0 (leading dimension ld0), [K0, K) -> dW1
 * (ld1).  The graph convolution contracts [A_hat x | EA_hat] with [W | W_e] in one GEMM (graph_layers.py:99-110 has two
 * Linear layers); this hands each of the two parameters a contiguous gradient without a copy.  Same workspace. */
DGDM_API int dgdm_gemm_tn_split(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW0, int64_t ld0, int32_t K0,
                                float* dW1, int64_t ld1, float* db, int32_t M, int32_t N, int32_t K, void* workspace,
                                size_t workspace_bytes, void* stream);

/* The same three contractions on the 16-bit matrix pipe with fp32-level accuracy: every fp32 operand
 * is split exactly into three bf16 values (x = h + m + l) on its way into LDS and a product is the
 * six bf16 MFMAs whose terms are not below 2^-26 of it, accumulated in fp32 (csrc/gemm3.hip).
 * Same arguments, constraints, determinism and error codes as the fp32-MFMA entry points above;
 * results agree with them to fp32 rounding (a few 1e-7 relative to sum |a.b|).  2-2.5x faster. */
DGDM_API int dgdm_gemm_nt_bf16x3(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                                 int32_t M, int32_t N, int32_t K, int32_t accumulate, void* stream);
DGDM_API int dgdm_gemm_nn_bf16x3(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int32_t M, int32_t N,
                                 int32_t K, int32_t accumulate, void* stream);
/* dgdm_gemm_nt_bf16x3 with the weight given as two matrices side by side in K: C = A . [W0 | W1]^T (+ bias), W0 [N, K0]
 * (leading dimension ldw0), W1 [N, K - K0] (ldw1); K0 % 4 == 0, 0 < K0 < K.  The graph convolution contracts
 * [A_hat x | EA_hat] with node_lin.weight and edge_lin.weight (graph_layers.py:44,48,99-105) without concatenating them. */
DGDM_API int dgdm_gemm_nt_split_bf16x3(const float* A, int64_t lda, const float* W0, int64_t ldw0, int32_t K0, const float* W1,
                                       int64_t ldw1, const float* bias, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                                       int32_t accumulate, void* stream);
DGDM_API size_t dgdm_gemm_tn_bf16x3_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t with_bias);
DGDM_API int dgdm_gemm_tn_bf16x3(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, float* db,
                                 int32_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream);
DGDM_API int dgdm_gemm_tn_split_bf16x3(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW0, int64_t ld0,
                                       int32_t K0, float* dW1, int64_t ld1, float* db, int32_t M, int32_t N, int32_t K,
                                       void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3''  the same three contractions with every fp32 operand carried as fp16 hi + lo (22 significand bits) and three
 * v_mfma_f32_32x32x16_f16 per product term (csrc/gemm_h.hip): half the matrix work of the bf16x3 kernels at the same
 * fp32-level accuracy.  fp16 has no range to spare, so every operand comes with `amax_*`: a DEVICE amax slot (a group of
 * DGDM_AMAX_WAYS words, see above) whose maximum is the float bits of an UPPER BOUND of max|x| over the operand (the exact
 * maximum, or that of a tensor the operand is a slice of); the kernel scales the operand by the power of two that puts that
 * bound in [2^14, 2^15) and undoes it in the epilogue (exact).
 * A bound that is too SMALL overflows fp16: callers without a bound use the bf16x3 entry points.
 *   dgdm_amax_bits : atomic max of the float bits of |x| over a [rows, cols] matrix (row stride ld) into the slot group
 *                    (zero it first; non-negative floats order like integers, so the result does not depend on the order).
 *   dgdm_amax_table: the same for many contiguous tensors in one launch; table = DEVICE array of `count` records
 *                    {const float* p; int64 n; int64 group;} (a large tensor is cut into several records with the same
 *                    group index), groups = base of the zeroed slot groups (all weights of a model, once per step).
 * Shapes / alignment / workspace as the bf16x3 entry points (dgdm_gemm_tn_f16x2_workspace_bytes == the bf16x3 size). */
DGDM_API int dgdm_fill_u32(uint32_t* p, int64_t n, uint32_t value, void* stream);   /* p[0..n) <- value, as a kernel (graph-replay safe) */
DGDM_API int dgdm_amax_bits(const float* x, int64_t ld, int64_t rows, int32_t cols, uint32_t* group, void* stream);
DGDM_API int dgdm_amax_table(const void* table, int32_t count, uint32_t* groups, void* stream);
DGDM_API int dgdm_gemm_nt_f16x2(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                                int32_t M, int32_t N, int32_t K, int32_t accumulate, const uint32_t* amax_a, const uint32_t* amax_w,
                                void* stream);
DGDM_API int dgdm_gemm_nt_split_f16x2(const float* A, int64_t lda, const float* W0, int64_t ldw0, int32_t K0, const float* W1,
                                      int64_t ldw1, const float* bias, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                                      int32_t accumulate, const uint32_t* amax_a, const uint32_t* amax_w0, const uint32_t* amax_w1,
                                      void* stream);
DGDM_API int dgdm_gemm_nn_f16x2(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int32_t M, int32_t N,
                                int32_t K, int32_t accumulate, const uint32_t* amax_a, const uint32_t* amax_w, void* stream);
DGDM_API size_t dgdm_gemm_tn_f16x2_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t with_bias);
DGDM_API int dgdm_gemm_tn_f16x2(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW, int64_t lddw, float* db,
                                int32_t M, int32_t N, int32_t K, void* workspace, size_t workspace_bytes, const uint32_t* amax_dy,
                                const uint32_t* amax_x, void* stream);
DGDM_API int dgdm_gemm_tn_split_f16x2(const float* dY, int64_t ldy, const float* X, int64_t ldx, float* dW0, int64_t ld0,
                                      int32_t K0, float* dW1, int64_t ld1, float* db, int32_t M, int32_t N, int32_t K,
                                      void* workspace, size_t workspace_bytes, const uint32_t* amax_dy, const uint32_t* amax_x,
                                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4-dense  MultiHeadAttention.forward as the reference exposes it (reference: core/attention.py:73-181; csrc/attn_dense.hip): a dense
 * batch of B sequences, Lq queries against Lk keys / values each, with everything the reference adds to its score tensor:
 *   S[b,h,q,k] = Q[b,q,h] . K[b,k,h] * scale + bias - |posq[b,q] - posk[b,k]| * inv_tau ;  -inf where bmask or kpm[b,k]
 *   O = dropout(softmax_k S) V
 * Q [B * Lq, >= H * D] (row stride ldq), K / V [B * Lk, >= H * D] (common row stride ldk), D in {16, 32, 64, 128} (narrower heads
 * zero-padded by the caller).  bias (float attn_mask, attention.py:131-135: added) and bmask (bool attn_mask: -inf where nonzero) are
 * mutually exclusive, either may be NULL; both are addressed through the strides (sb, sh, sq, sk), in elements, of the mask's
 * broadcast view [B, H, Lq, Lk] (0 for a broadcast dimension: a 2-D [Lq, Lk] mask has sb = sh = 0).  kpm = key_padding_mask [B, Lk]
 * (attention.py:137-142; nonzero = key ignored) or NULL.  posq [B * Lq, 2] / posk [B * Lk, 2] (both or neither): the spatial bias
 * of SpatialAttention.forward called with a mask (attention.py:311-314).  lse [B, H, Lq] NATURAL-log sum-exp, delta [B, H, Lq] scratch
 * of the backward.  A row whose keys are all masked is NaN in O / lse and in the gradients (softmax of all -inf in the reference).
 * dgdm_attn_dense_weights: the weights AFTER dropout (attention.py:145-146), head mean W [B, Lq, Lk] (per_head = 0, attention.py:172)
 * or W [B * H, Lq, Lk] (per_head = 1, attention.py:174-176); with drop_p > 0 pass the forward's seed.  The mask gets no gradient.
 * The dropout mask is a hash of (seed, b, head, query, key) private to this kernel family.  fp32 on the vector units: off the DGDM
 * hot path (DGDMModel's two uses of the class run on K4 / K10), built for parity with the class's own forward. */
DGDM_API int dgdm_attn_dense_fwd(const float* Q, int64_t ldq, const float* K, const float* V, int64_t ldk, int32_t B, int32_t Lq, int32_t Lk,
                                 int32_t H, int32_t D, float scale, const float* bias, const uint8_t* bmask, int64_t sb, int64_t sh,
                                 int64_t sq, int64_t sk, const uint8_t* kpm, const float* posq, const float* posk, float inv_tau,
                                 float drop_p, uint32_t seed, float* O, int64_t ldo, float* lse, void* stream);
DGDM_API int dgdm_attn_dense_bwd(const float* Q, int64_t ldq, const float* K, const float* V, int64_t ldk, int32_t B, int32_t Lq, int32_t Lk,
                                 int32_t H, int32_t D, float scale, const float* bias, const uint8_t* bmask, int64_t sb, int64_t sh,
                                 int64_t sq, int64_t sk, const uint8_t* kpm, const float* posq, const float* posk, float inv_tau,
                                 float drop_p, uint32_t seed, const float* O, const float* dO, int64_t ldo, const float* lse, float* delta,
                                 float* dQ, int64_t ldgq, float* dK, float* dV, int64_t ldgk, void* stream);
DGDM_API int dgdm_attn_dense_weights(const float* Q, int64_t ldq, const float* K, int64_t ldk, int32_t B, int32_t Lq, int32_t Lk, int32_t H,
                                     int32_t D, float scale, const float* bias, const uint8_t* bmask, int64_t sb, int64_t sh, int64_t sq,
                                     int64_t sk, const uint8_t* kpm, const float* posq, const float* posk, float inv_tau, float drop_p,
                                     uint32_t seed, const float* lse, int32_t per_head, float* W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4-gen  spatial attention for 16 < head_dim <= 64 (csrc/attn_gen.hip).  The reference accepts any embed_dim % num_heads == 0
 * (core/attention.py:36-40); the MFMA kernels above are tiled for head_dim 16 (smaller heads are zero-padded).  Heads of 32 / 64
 * channels (hidden_dims[-1] = 128 with attention_heads 4 / 2) run on these fp32 vector-unit kernels: same mathematics
 * (softmax(Q K^T scale - dist inv_tau), dropout on the weights, row sums before the mask), Q / K / V [N_tot, H * D] views with a
 * common row stride ld, D in {32, 64} (other head dims: zero-pad to the next one), num_tiles = sum over graphs of ceil(n_g / 64)
 * (= dgdm_spatial_attn_q_tile_rows() rows per tile), lse [H, N_tot] NATURAL-log sum-exp, delta [H, N_tot] scratch of the backward.
 * The dropout mask is a hash of (seed, graph, head, query, key) private to this kernel family. */
DGDM_API int dgdm_spatial_attn_gen_fwd(const float* Q, const float* K, const float* V, int64_t ld, const float* pos, const int32_t* ptr,
                                       int32_t B, int32_t num_tiles, int32_t N_tot, int32_t H, int32_t D, float scale, float inv_tau,
                                       float drop_p, uint32_t seed, float* O, int64_t ldo, float* lse, void* stream);
DGDM_API int dgdm_spatial_attn_gen_bwd(const float* Q, const float* K, const float* V, int64_t ld, const float* pos, const int32_t* ptr,
                                       int32_t B, int32_t num_tiles, int32_t N_tot, int32_t H, int32_t D, float scale, float inv_tau,
                                       float drop_p, uint32_t seed, const float* O, const float* dO, int64_t ldo, const float* lse,
                                       float* delta, float* dQ, float* dK, float* dV, int64_t ldg, void* stream);
DGDM_API int dgdm_spatial_attn_gen_mean_weights(const float* Q, const float* K, int64_t ld, const float* pos, const int32_t* ptr, int32_t B,
                                                int32_t num_tiles, int32_t N_tot, int32_t H, int32_t D, float scale, float inv_tau,
                                                const float* lse, float* W, const int64_t* offsets, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3-img  weight images (csrc/gemm_img.hip).  The two contractions whose second operand is a WEIGHT -- y = x W^T + b
 * (core/graph_layers.py:45-49,141-150, models/encoders.py:73-91, core/attention.py:60-63, core/diffusion.py:87-104 of the
 * reference: every nn.Linear forward) and dx = dy W (its input gradient) -- take the weight pre-split: an "image" holds the
 * fp16 hi+lo halves of B(col, k), scaled by the power of two of the weight's amax slot, in MFMA fragment order, so the GEMM
 * copies it global -> LDS without touching a register and converts only the activation operand (once per wave).
 *   image of B [cols, k]: dgdm_gemm_image_bytes(cols, k) bytes, 16-byte aligned; dgdm_gemm_image_blocks(cols, k) build blocks.
 *   dgdm_gemm_image_build_many: ONE launch builds `count` images.  table = DEVICE array of records
 *       { const float* w0; const float* w1; int64 ld0, ld1; const uint32* amax0; const uint32* amax1; void* image;
 *         int32 rows, cols0, cols1, transposed, block0; }           (80 bytes, 8-byte aligned)
 *     transposed = 0: B(col, k) = [W0 | W1][col][k], W0 [rows, cols0], W1 [rows, cols1] or NULL (forward; a weight given as two
 *                     matrices side by side in k);   transposed = 1: B(col, k) = W0[k][col] (dx = dy . W0; W1 unused).
 *     block0 = running sum of dgdm_gemm_image_blocks over the preceding records; total_blocks = the sum over all records.
 *     cols0, cols1, ld0, ld1 multiples of 4, pointers 16-byte aligned.
 *   dgdm_gemm_rows_img: C[M, ncols] (+)= A[M, K] . B[tile_begin*32 .. +ncols, 0..K)^T + bias[ncols]; `image_tiles` = number
 *     of 32-column tiles of the WHOLE image (its column count / 32 rounded up): a column range of an image is an operand too
 *     (the slice of a weight).  K % 16 == 0, lda % 4 == 0; amax_a as for dgdm_gemm_nt_f16x2.  Arithmetic: that of K3''. */
DGDM_API size_t dgdm_gemm_image_bytes(int32_t cols, int32_t k);
DGDM_API int32_t dgdm_gemm_image_blocks(int32_t cols, int32_t k);
DGDM_API int dgdm_gemm_image_build_many(const void* table, int32_t count, int32_t total_blocks, void* stream);
/* one image, the record passed by value (no device table: usable inside a stream capture) */
DGDM_API int dgdm_gemm_image_build(const float* w0, int64_t ld0, const float* w1, int64_t ld1, const uint32_t* amax0,
                                   const uint32_t* amax1, void* image, int32_t rows, int32_t cols0, int32_t cols1, int32_t transposed,
                                   void* stream);
DGDM_API int dgdm_gemm_rows_img(const float* A, int64_t lda, int32_t M, int32_t K, const void* image, int32_t image_tiles,
                                int32_t tile_begin, int32_t ncols, const float* bias, float* C, int64_t ldc, int32_t accumulate,
                                const uint32_t* amax_a, void* stream);

/* K3-img with fused epilogues (round 5).  The layer that follows a Linear on the path is finished in the GEMM's registers
 * instead of by a kernel of its own (SURVEY.md 2c, K3/K6 "fused epilogues"); operands and arithmetic as dgdm_gemm_rows_img,
 * ncols % 4 == 0, every matrix 16-byte aligned with a leading dimension that is a multiple of 4.  Dropout masks are the function
 * of (seed, row * ncols + col) that dgdm_act_dropout_* / dgdm_rownorm_* use on a contiguous [M, ncols] tensor (csrc/rowmath.hpp),
 * so either side can regenerate the other's mask.  amax_y / amax_g: nullable amax slot that receives max|output|.
 *   dgdm_gemm_rows_img_act:      pre = A.B + bias (stored when `pre` is not NULL: the backward needs it),
 *                                Y = dropout(act(pre))                     core/graph_layers.py:233-239  dropout(GELU(conv(x)))
 *   dgdm_gemm_rows_img_act_bwd:  G = (A.B) * act'(pre) * mask              the backward of that layer, as the epilogue of the GEMM
 *                                                                          that forms its incoming gradient (A = dY_next, B = W_next)
 *   dgdm_gemm_rows_img_norm:     S = dropout_pre(A.B + bias) [+ res] (stored when `sum` is not NULL: it is the input
 *                                dgdm_rownorm_bwd reads); pre_drop_p > 0: the dropout between a projection and the residual it
 *                                is added to (core/attention.py:176-181, resid dropout), mask of (pre_seed, element index);
 *                                res [M, ncols], or with res_ptr (int32 [res_segments + 1], ascending row offsets) ONE row per
 *                                segment of rows -- the per-graph time bias of the denoiser's first layer (core/diffusion.py:165-170);
 *                                Y = dropout(act(norm_groups(S) * gamma + beta)), mean / rstd [M * groups]
 *                                core/graph_layers.py:241-245 norm1(output_proj(h) + x); models/encoders.py:267-269;
 *                                core/diffusion.py:94-102 GroupNorm(8) + SiLU + dropout behind the denoiser's Linears.
 *     A (row, group) must lie inside one wave's columns: dgdm_gemm_rows_img_norm_supported(ncols, groups) says whether the
 *     shape is taken (ncols / groups a power-of-two multiple of 32, at most 256; LayerNorm: ncols <= 256). */
DGDM_API int dgdm_gemm_rows_img_act(const float* A, int64_t lda, int32_t M, int32_t K, const void* image, int32_t image_tiles,
                                    int32_t tile_begin, int32_t ncols, const float* bias, float* pre, int64_t ldp, float* Y,
                                    int64_t ldy, int32_t act, float drop_p, uint32_t seed, const uint32_t* amax_a, uint32_t* amax_y,
                                    void* stream);
DGDM_API int dgdm_gemm_rows_img_act_bwd(const float* A, int64_t lda, int32_t M, int32_t K, const void* image, int32_t image_tiles,
                                        int32_t tile_begin, int32_t ncols, const float* pre, int64_t ldp, float* G, int64_t ldg,
                                        int32_t act, float drop_p, uint32_t seed, const uint32_t* amax_a, uint32_t* amax_g,
                                        void* stream);
DGDM_API int32_t dgdm_gemm_rows_img_norm_supported(int32_t ncols, int32_t groups);
DGDM_API int dgdm_gemm_rows_img_norm(const float* A, int64_t lda, int32_t M, int32_t K, const void* image, int32_t image_tiles,
                                     int32_t tile_begin, int32_t ncols, const float* bias, float pre_drop_p, uint32_t pre_seed,
                                     const float* res, int64_t ldr, const int32_t* res_ptr, int32_t res_segments,
                                     const float* gamma, const float* beta, int32_t groups, float eps, float* sum, int64_t lds,
                                     float* Y, int64_t ldy, float* mean, float* rstd, int32_t act, float drop_p, uint32_t seed,
                                     const uint32_t* amax_a, uint32_t* amax_y, void* stream);

/* Deferred reduction of the split-M weight-gradient GEMMs.  dgdm_gemm_tn_partial_* run only the first half of dgdm_gemm_tn_*
 * (chunk partials into `workspace`, [dgdm_gemm_tn_chunks(M,N,K)][N*K (+N when with_bias)] floats); dgdm_gemm_tn_reduce_many then
 * reduces up to DGDM_TN_REDUCE_MAX such workspaces in ONE launch, with the arithmetic (fixed order) of the immediate reduction.
 * `descs` is a HOST array (copied into the kernel arguments): partial/slots/N/K as above; dW0 (ld0) receives columns [0, K0) of
 * dW, dW1 (ld1) columns [K0, K) (K0 == K: one matrix); db nullable. */
#define DGDM_TN_REDUCE_MAX 24
typedef struct DgdmTnReduce {
  const float* partial;
  float* dW0;
  float* dW1;
  float* db;
  int64_t ld0, ld1;
  int32_t slots, N, K, K0;
} DgdmTnReduce;
DGDM_API int32_t dgdm_gemm_tn_chunks(int32_t M, int32_t N, int32_t K);
DGDM_API int dgdm_gemm_tn_partial_bf16x3(const float* dY, int64_t ldy, const float* X, int64_t ldx, int32_t with_bias, int32_t M,
                                         int32_t N, int32_t K, void* workspace, size_t workspace_bytes, void* stream);
DGDM_API int dgdm_gemm_tn_partial_f16x2(const float* dY, int64_t ldy, const float* X, int64_t ldx, int32_t with_bias, int32_t M,
                                        int32_t N, int32_t K, void* workspace, size_t workspace_bytes, const uint32_t* amax_dy,
                                        const uint32_t* amax_x, void* stream);
DGDM_API int dgdm_gemm_tn_reduce_many(const DgdmTnReduce* descs, int32_t count, void* stream);
/* Up to DGDM_TN_PARTIAL_MAX weight-gradient GEMMs as ONE launch (the dW GEMMs of small layers are start-up bound as launches of
 * their own, and two dozen problems fill the chip together, so each is cut into fewer, longer row chunks than a launch of its own
 * would be).  `descs` is a HOST array; each entry has the arguments of dgdm_gemm_tn_partial_f16x2 (M, N, K > 0) and leaves
 * [dgdm_gemm_tn_chunks_grouped(M,N,K)][N*K (+N when with_bias)] floats in its workspace for dgdm_gemm_tn_reduce_many (slots =
 * that chunk count).  Fixed summation order: repeatable bit for bit; equal to the single launch up to fp32 rounding (other chunk
 * boundaries). */
#define DGDM_TN_PARTIAL_MAX 24
typedef struct DgdmTnPartial {
  const float* dY;
  const float* X;
  void* workspace;
  const uint32_t* amax_dy;
  const uint32_t* amax_x;
  int64_t ldy, ldx;
  size_t workspace_bytes;
  int32_t M, N, K, with_bias;
} DgdmTnPartial;
DGDM_API int32_t dgdm_gemm_tn_chunks_grouped(int32_t M, int32_t N, int32_t K);
DGDM_API int dgdm_gemm_tn_partial_many_f16x2(const DgdmTnPartial* descs, int32_t count, void* stream);


/* ---------------------------------------------------------------------------------------------
 * K9  top-k node pooling and unpooling of the graph U-Net
 * replaces: AdaptiveGraphPooling.forward -- core/graph_layers.py:285-329 (score MLP tail, topk,
 *           mask.nonzero, x[perm] * score, edge filter + relabel) and the unpool / skip / activation
 *           of GraphUNet.forward -- core/graph_layers.py:441-448.
 *   dgdm_pool_score_fwd : s[i] = f(w2 . relu(h[i]) + b2[0]);  h [N, C] = first score layer's output; f by `nonlinearity`
 *                         (graph_layers.py:276-283): 0 tanh (DGDMModel's pools), 1 sigmoid, 2 identity -- the logit, for
 *                         nonlinearity='softmax', whose softmax runs over ALL nodes of the batch: dgdm_vec_softmax_fwd/bwd
 *   dgdm_pool_score_bwd : dh, dw2 [C], db2 [1] from ds (fixed-order reductions)
 *   dgdm_vec_softmax_fwd/bwd : s = softmax(z) over a vector of N floats; dz = s * (ds - <s, ds>)   (one workgroup, fixed order)
 *   dgdm_count_ge       : out[0] = #{i : s[i] >= threshold}  (min_score pooling, graph_layers.py:302-303: the kept count is
 *                         data dependent -- the caller reads it back, ONE host sync, and selects the top `count`)
 *   dgdm_topk_perm      : exact top-k of s.  perm [k] int64 = kept node ids in ascending order,
 *                         node_map [N] int32 = new id or -1.  Ties at the k-th value keep the lowest
 *                         ids.  Integer work: bit-exact and deterministic.  0 <= k <= N.
 *   dgdm_pool_gather_fwd: out[j] = x[perm[j]] * s[perm[j]] * mult
 *   dgdm_pool_gather_bwd: dx [N, C] (zero rows for dropped nodes) and ds [N] from g [k, C]
 *   dgdm_edge_relabel   : out[:, e] = node_map[edge_index[:, e]] if both ends are kept, else (-1, -1);
 *                         ids outside [0, N) (dropped at an earlier level) stay dropped.  int64 [2, E].
 *   dgdm_unpool_add_relu_fwd : out[i] = relu(skip[i] + (node_map[i] >= 0 ? xc[node_map[i]] : 0))
 *   dgdm_unpool_add_relu_bwd : dskip[i] = g[i] * [out[i] > 0];  dxc[node_map[i]] = dskip[i]
 * All row pointers: C % 4 == 0, row strides % 4 == 0, 16-byte aligned. */
DGDM_API int dgdm_pool_score_fwd(const float* h, int64_t ldh, const float* w2, const float* b2, int32_t N, int32_t C, float* s,
                                 const uint8_t* decide, int32_t nonlinearity, void* stream);
DGDM_API int dgdm_vec_softmax_fwd(const float* z, int32_t N, float* s, void* stream);
DGDM_API int dgdm_vec_softmax_bwd(const float* s, const float* ds, int32_t N, float* dz, void* stream);
DGDM_API int dgdm_count_ge(const float* s, int32_t N, float threshold, int32_t* out, void* stream);
DGDM_API size_t dgdm_pool_score_bwd_workspace_bytes(int32_t N, int32_t C);
DGDM_API int dgdm_pool_score_bwd(const float* h, int64_t ldh, const float* w2, const float* s, const float* ds, int32_t N, int32_t C,
                                 float* dh, int64_t lddh, float* dw2, float* db2, const uint8_t* decide, int32_t nonlinearity,
                                 void* workspace, size_t workspace_bytes, uint32_t* amax, void* stream);
DGDM_API size_t dgdm_topk_perm_workspace_bytes(int32_t N);
DGDM_API int dgdm_topk_perm(const float* s, int32_t N, int32_t k, int64_t* perm, int32_t* node_map, void* workspace,
                            size_t workspace_bytes, void* stream);
DGDM_API int dgdm_pool_gather_fwd(const float* x, int64_t ldx, const float* s, const int64_t* perm, int32_t k, int32_t C, float mult,
                                  float* out, int64_t ldo, void* stream);
DGDM_API int dgdm_pool_gather_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* s, const int32_t* node_map,
                                  int32_t N, int32_t C, float mult, float* dx, int64_t lddx, float* ds, void* stream);
DGDM_API int dgdm_edge_relabel(const int64_t* edge_index, int64_t E, const int32_t* node_map, int32_t N, int64_t* out, void* stream);
DGDM_API int dgdm_unpool_add_relu_fwd(const float* xc, int64_t ldxc, const float* skip, int64_t lds, const int32_t* node_map, int32_t N,
                                      int32_t C, float* out, int64_t ldo, const uint8_t* decide, void* stream);
DGDM_API int dgdm_unpool_add_relu_bwd(const float* g, int64_t ldg, const float* out, int64_t ldo, const int32_t* node_map, int32_t N,
                                      int32_t C, float* dskip, int64_t ldds, float* dxc, int64_t lddxc, const uint8_t* decide,
                                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * K11  tissue-graph edge construction (the step upstream of the model)
 * replaces: TissueGraphBuilder._create_spatial_edges / _create_morphological_edges /
 *           _remove_duplicate_edges and the edge part of _to_pytorch_geometric --
 *           preprocessing/tissue_graph_builder.py:286-357, 384-402 (scikit-learn kNN + Python loops).
 *   dgdm_knn2d       : for every 2-D point the K nearest points (itself included), ascending by
 *                      (distance, index): idx int32 [N, K], dist float [N, K].  1 <= K <= min(N, 64).
 *                      d^2 = fl(fl(dx*dx) + fl(dy*dy)), d = sqrt correctly rounded: a float32 CPU
 *                      restatement reproduces both outputs bit for bit.
 *   dgdm_row_sqnorm  : sq[i] = |x_i|^2
 *   dgdm_knn_gram    : feature-space kNN of the query rows [q0, q0+B) from their Gram rows
 *                      G[q][j] = x_(q0+q) . x_j  (float [B, ldg >= N], from dgdm_gemm_nt*):
 *                      writes rows q0..q0+B-1 of idx int32 [N, K] and of sim float [N, K]
 *                      (cosine similarity of the pair, from the Gram value).  Same ordering rule as
 *                      dgdm_knn2d.
 *   dgdm_pair_cosine : sim[i][p] = cos(x_i, x_idx[i][p]) recomputed from a direct dot product in a
 *                      fixed order: bitwise symmetric in the pair (the Gram value is not), which the
 *                      duplicate rule of dgdm_edge_dedup_count relies on.
 *   dgdm_edge_dedup_count : thresholded spatial (weight exp(-10 d)) and morphological (cosine)
 *                      candidates in the reference's order -> duplicates removed per undirected pair
 *                      (heaviest wins, earliest on ties; output order = first occurrence of the pair)
 *                      -> *n_edges (device int64) = number of kept pairs U.  Column 0 of both
 *                      neighbour tables is skipped as "self" exactly as the reference does.
 *   dgdm_edge_emit   : after the caller has read U: edge_index int64 [2, 2U] (both directions,
 *                      consecutive), edge_attr float [2U, edge_dim] ([d, w, 0..] / [cos, 0..]),
 *                      edge_type int64 [2U] (0 spatial, 1 morphological), edge_weight [2U] or NULL.
 *                      Must follow dgdm_edge_dedup_count on the same, untouched workspace. */
DGDM_API int dgdm_knn2d(const float* coords, int32_t N, int32_t K, int32_t* idx, float* dist, void* stream);
DGDM_API int dgdm_row_sqnorm(const float* X, int64_t ldx, int32_t N, int32_t F, float* sq, void* stream);
DGDM_API size_t dgdm_knn_gram_workspace_bytes(int32_t B, int32_t K);
DGDM_API int dgdm_knn_gram(const float* G, int64_t ldg, const float* sq, int32_t N, int32_t q0, int32_t B, int32_t K, int32_t* idx,
                           float* sim, void* workspace, size_t workspace_bytes, void* stream);
DGDM_API int dgdm_pair_cosine(const float* X, int64_t ldx, const float* sq, const int32_t* idx, int32_t N, int32_t K, int32_t F,
                              float* sim, void* stream);
DGDM_API size_t dgdm_edge_dedup_workspace_bytes(int32_t N, int32_t Ks1, int32_t Km1);
DGDM_API int dgdm_edge_dedup_count(const int32_t* sidx, const float* sdist, int32_t Ks1, const int32_t* midx, const float* msim,
                                   int32_t Km1, int32_t N, float threshold, void* workspace, size_t workspace_bytes, int64_t* n_edges,
                                   void* stream);
DGDM_API int dgdm_edge_emit(const int32_t* sidx, const float* sdist, int32_t Ks1, const int32_t* midx, const float* msim, int32_t Km1,
                            int32_t N, float threshold, const void* workspace, int64_t U, int32_t edge_dim, int64_t* edge_index,
                            float* edge_attr, int64_t* edge_type, float* edge_weight, void* stream);

/* ---- optimizer step (reference training/trainer.py:217-226: torch.optim.AdamW(lr, weight_decay)) ------------
 * One launch per 96 tensors for ALL live parameters of the model.  `tensors`: HOST array of descriptors holding DEVICE
 * pointers (fp32, 4-byte aligned; 16-byte aligned tensors take the vector path), numel elements each; descriptors travel in
 * the kernel arguments, so a launch recorded in a HIP graph replays on the same addresses.  lr_dev (nullable): device fp32
 * learning rate (takes precedence over `lr`, for schedulers that write it without a host sync).  step_dev: device fp32 count
 * of the steps taken so far by this set of tensors -- read by every workgroup (t = step + 1 enters the bias corrections),
 * advanced by the launch itself; ticket_dev: one device uint32, zero before the first call, left zero.
 * Arithmetic of torch/optim/adamw.py (decoupled weight decay; no amsgrad, no maximize). */
typedef struct DgdmAdamTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
} DgdmAdamTensor;
DGDM_API int dgdm_adamw_step(const DgdmAdamTensor* tensors, int32_t count, const float* lr_dev, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float* step_dev, uint32_t* ticket_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DGDM_HIP_H */
