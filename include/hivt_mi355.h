/* libhivt_mi355 -- C ABI of the MI355X (gfx950) kernels behind HiVT-GAN's edge attention.
 *
 * Every entry point replaces a piece of the reference's Python / torch_geometric hot path; the
 * comment above each one cites the reference lines (paths relative to the reference repository).
 *
 * Conventions
 *   - plain C: raw device pointers, explicit sizes and strides (strides are in ELEMENTS of the
 *     pointed-to type), no C++ or torch types;
 *   - return value 0 = ok, negative = error (HIVT_ERR_*); hivt_last_error() returns the text of the
 *     calling thread's last error (thread-local, valid until that thread's next failing call);
 *   - all buffers are caller-owned; the library never allocates device memory.  Scratch comes from
 *     a caller-provided workspace sized by the matching *_workspace_bytes() query (256-B aligned);
 *   - every launch goes to the passed hipStream_t (as void*), is asynchronous, and performs no
 *     host synchronisation.  Calls are re-entrant; the only process-wide state is the operand-precision
 *     mask of hivt_set_matmul_bf16() (read once per call) and the static scratch of
 *     hivt_csr_build_static() (one array per GPU: calls that overlap on one GPU race);
 *   - row indices are int32; edge lists handed over by the framework (int64 [2,E] COO,
 *     row 0 = source j, row 1 = target i) are converted by the edge-construction entry points;
 *   - "CSR order" = edges sorted by target, ties in original list order (stable), which is the
 *     accumulation order of the reference's CPU scatter_add_.
 */
#ifndef HIVT_MI355_H
#define HIVT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIVT_OK 0
#define HIVT_ERR_ARG (-1)     /* invalid argument (null pointer, bad size, unsupported D/H) */
#define HIVT_ERR_WORKSPACE (-2) /* workspace missing or too small */
#define HIVT_ERR_LAUNCH (-3)  /* HIP runtime reported a launch error */
#define HIVT_ERR_UNSUPPORTED (-4) /* dtype or mode not built */

#define HIVT_F32 0
#define HIVT_BF16 1

typedef void* hivt_stream_t; /* hipStream_t */

/* Library version: major*10000 + minor*100 + patch. */
int hivt_version(void);

/* Text of the calling thread's last error ("" if none). */
const char* hivt_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Edge construction (bit-exact edge sets, order preserving)
 * ------------------------------------------------------------------------------------------ */

/* Workspace for hivt_edges_temporal / hivt_edges_radius over `items` candidate edges. */
size_t hivt_edges_workspace_bytes(int64_t items);

/* Per-timestep agent-agent edges for timesteps t0 .. t0+T-1 in ONE launch sequence.
 * Replaces, for every t: subgraph(~padding_mask[:, t], edge_index) + positions[src,t]-positions[dst,t]
 * (models/local_encoder.py:58-61) + DistanceDropEdge (utils.py:69-78, strict ||.||_2 < radius;
 * radius <= 0 disables the filter) and the snapshot batching with node id (t-t0)*N + n
 * (models/local_encoder.py:62-71).  With t0 = T_hist-1, T = 1, radius <= 0 it is the edge set of
 * GlobalInteractor.forward (models/global_interactor.py:48-50).
 * Output lists are t-major and keep the input edge order inside a timestep.
 *   edge_index    int64 [2,E0] (row stride E0)       padding_mask uint8/bool [N, pm_stride]
 *   positions     float [N, *, 2] with element strides pos_stride_n / pos_stride_t
 *   src_out/dst_out int32 [T*E0], attr_out float [T*E0,2], count_out int32 [1] (kept edges). */
int hivt_edges_temporal(const int64_t* edge_index, int64_t E0, const uint8_t* padding_mask, int64_t pm_stride,
                        const float* positions, int64_t pos_stride_n, int64_t pos_stride_t, int32_t N,
                        int32_t t0, int32_t T, float radius, int32_t* src_out, int32_t* dst_out,
                        float* attr_out, int32_t* count_out, void* workspace, size_t workspace_bytes,
                        hivt_stream_t stream);

/* Radius filter of a given edge list with given attributes (lane -> actor edges):
 * DistanceDropEdge.__call__ (utils.py:69-78) as used at models/local_encoder.py:80. */
int hivt_edges_radius(const int64_t* edge_index, int64_t E0, const float* edge_attr, float radius,
                      int32_t* src_out, int32_t* dst_out, float* attr_out, int32_t* count_out,
                      void* workspace, size_t workspace_bytes, hivt_stream_t stream);

/* Workspace for hivt_edges_all_pairs_csr over rows = N * T target rows. */
size_t hivt_edges_all_pairs_workspace_bytes(int64_t rows);

/* Per-timestep agent-agent edges of timesteps t0 .. t0+T-1 produced DIRECTLY in CSR order, for batches
 * whose candidate list is every ordered pair of agents of a scene, which is what the reference's
 * preprocessor writes (nuscenes_dataset.py:139-141).  Same edge set, attributes and order as
 * hivt_edges_temporal followed by hivt_csr_build over the targets -- i.e. as subgraph + positions
 * difference + DistanceDropEdge (models/local_encoder.py:58-61, utils.py:69-78; t0 = T_hist-1, T = 1,
 * radius <= 0: models/global_interactor.py:48-50) followed by torch_geometric's grouping by target --
 * without reading a candidate list and without a sort: row (t, i) lists the sources j of i's scene in
 * ascending order.
 *   scene_range int32 [N,2]: first agent of agent i's scene and one past its last;
 *   rowptr int32 [N*T + 1]; src_out / dst_out int32 [capacity] (node ids (t-t0)*N + n), attr_out float
 *   [capacity,2]; count_out int32 [1] = number of kept edges BEFORE clamping.  If it exceeds `capacity`
 *   the lists hold the first `capacity` edges, rowptr is clamped to it, and the caller discards the batch. */
int hivt_edges_all_pairs_csr(const int32_t* scene_range, const uint8_t* padding_mask, int64_t pm_stride,
                             const float* positions, int64_t pos_stride_n, int64_t pos_stride_t, int32_t N,
                             int32_t t0, int32_t T, float radius, int64_t capacity, int32_t* rowptr,
                             int32_t* src_out, int32_t* dst_out, float* attr_out, int32_t* count_out,
                             void* workspace, size_t workspace_bytes, hivt_stream_t stream);

/* Workspace for hivt_csr_build. */
size_t hivt_csr_workspace_bytes(int64_t E, int32_t num_rows);

/* Stable counting sort of an edge list by `key` (target for the softmax/aggregate side, source
 * for the node-gradient side): rowptr[r]..rowptr[r+1] index `perm`, and perm lists original edge
 * positions in ascending order inside a row.  Replaces the implicit grouping done by
 * torch_geometric's scatter in MessagePassing.propagate(aggr='add') and utils.softmax
 * (call sites models/local_encoder.py:98,179,292,369; models/global_interactor.py:75,122).
 *   key int32 [E] with values in [0,num_rows); rowptr int32 [num_rows+1]; perm int32 [E].
 * `count` (device int32, may be NULL) gives the number of meaningful entries when the list is
 * padded to a fixed capacity E: entries at or beyond *count are ignored, rowptr[num_rows] = *count
 * and perm[*count .. E) is left untouched.  This lets a captured HIP graph be replayed on batches
 * with different edge counts.
 * `row_order` (int32 [num_rows], may be NULL) receives the rows ranked by length, longest first (classes of two
 * lengths; the order inside a class is unspecified): a schedule for kernels that give every row a workgroup, see
 * hivt_seg_attn_fwd.  It costs no launch: the ranking is part of the scan of the row lengths. */
int hivt_csr_build(const int32_t* key, int64_t E, const int32_t* count, int32_t num_rows, int32_t* rowptr,
                   int32_t* perm, int32_t* row_order, void* workspace, size_t workspace_bytes, hivt_stream_t stream);

/* hivt_csr_build without a workspace and without any allocation: the scratch (row cursors and the
 * unordered placement) is a static device array inside the library, HIVT_CSR_STATIC_MAX_ROWS +
 * HIVT_CSR_STATIC_MAX_EDGES int32 words (8 MiB) held on every GPU the library runs on until the process
 * ends.  No counterpart in the reference: it exists for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same results
 * as hivt_csr_build.  Sizes beyond the maxima are rejected with HIVT_ERR_ARG.  The device must be
 * current and own `stream`.  All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (two streams, two threads) race; calls on one stream are ordered and safe.  The first call on a device looks
 * the array's address up (hipGetSymbolAddress, which may load the code object and cannot be captured): make one
 * call per device outside any stream capture first; every later call is kernel launches only and records into a
 * HIP graph like them. */
#define HIVT_CSR_STATIC_MAX_EDGES (1 << 20)
#define HIVT_CSR_STATIC_MAX_ROWS (1 << 20)
int hivt_csr_build_static(const int32_t* key, int64_t E, const int32_t* count, int32_t num_rows, int32_t* rowptr,
                          int32_t* perm, hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Row bookkeeping of a batch (index arithmetic, masks and token rows that the reference writes as
 * dozens of small tensor operations per step)
 * ------------------------------------------------------------------------------------------ */

/* Everything the step derives from the raw batch before the first layer runs, in one launch:
 *   rotate_mat[n] = [[cos a, -sin a], [sin a, cos a]], y_rot = y @ rotate_mat (models/hivt.py:153-162;
 *   y may be NULL), x_rot[t*N + n] = x[n, t] @ rotate_mat[n] for t < T (models/local_encoder.py:135-149,
 *   step-major rows as the batched agent-agent stage numbers its nodes) and x_steps[t*N + n] = x[n, t] (the
 *   un-rotated rows in the same order, which the edge gather looks its sources up in),
 *   lane_code[l] = (is_intersections[l] * 3 + turn_directions[l]) * 2 + traffic_controls[l] (one row of the
 *   12-row table that sums the three categorical embeddings, models/local_encoder.py:332-334,361-363),
 *   keep_w[n] = 1 if agent n has a valid future step (padding_mask[n, T .. T+Hf) not all set), else 0 (the
 *   weights of the masked means that stand in for the boolean indexing of models/hivt.py:227-231).
 * x float [N, *, 2] with element strides; padding_mask uint8 [N, pm_stride >= T + Hf]; lane attributes uint8 [L]. */
int hivt_batch_prepare(const float* rotate_angles, const float* y, const float* x, int64_t x_stride_n,
                       int64_t x_stride_t, const uint8_t* padding_mask, int64_t pm_stride,
                       const uint8_t* is_intersections, const uint8_t* turn_directions,
                       const uint8_t* traffic_controls, int32_t N, int32_t T, int32_t Hf, int32_t L,
                       float* rotate_mat, float* y_rot, float* x_rot, float* x_steps, int32_t* lane_code,
                       float* keep_w, hivt_stream_t stream);

/* The (step, agent) nodes t*N + n with padding_mask[n][t] == 0, packed in ascending order into `capacity` rows:
 * the agent-agent stage's result at a padded node is overwritten by the padding token before anything reads it
 * (models/local_encoder.py:230), so the stage runs on these rows only.
 *   rows[capacity]        node id of every packed row; rows at or beyond the live count repeat the first live node
 *   new_id[T*N]           packed position of a node, clamped into [0, capacity)
 *   live[capacity]        1 for packed rows below the live count
 *   count[2]              live count (unclamped), id of the first live node
 *   packed_rowptr[capacity + 1]  rowptr (int32 [T*N + 1], a CSR over all nodes whose padded nodes own no
 *                         edges) re-indexed to packed rows; both NULL to skip
 *   overflow[0]           1 if the live count exceeds capacity or any *edge_countK exceeds edge_capK (K = 0..2,
 *                         NULL counters are skipped): the flag a fixed-capacity step discards its batch by. */
size_t hivt_row_subset_workspace_bytes(int32_t N, int32_t T);
int hivt_row_subset(const uint8_t* padding_mask, int64_t pm_stride, int32_t N, int32_t T, int32_t capacity,
                    const int32_t* rowptr, const int32_t* edge_count0, int64_t edge_cap0, const int32_t* edge_count1,
                    int64_t edge_cap1, const int32_t* edge_count2, int64_t edge_cap2, int32_t* rows, int32_t* new_id,
                    uint8_t* live, int32_t* count, int32_t* packed_rowptr, uint8_t* overflow, void* workspace,
                    size_t workspace_bytes, hivt_stream_t stream);

/* Per packed row r (node m = rows[r], step t = m / N, agent n = m % N): xc[r] = x_rot[m] and
 * code[r] = bos_mask[n][t] ? t : -1 -- the inputs of center_embed and of the begin-of-sequence substitution
 * (models/local_encoder.py:149-150) for the packed nodes. */
int hivt_center_rows(const float* x_rot, const uint8_t* bos_mask, int64_t bos_stride, const int32_t* rows, int32_t N,
                     int32_t T, int32_t capacity, float* xc, int32_t* code, hivt_stream_t stream);

/* out[m] = code[m] >= 0 ? token[code[m]] : x[m]  (torch.where(bos_mask, bos_token[t], center_embed),
 * models/local_encoder.py:150).  With token == NULL it is the gradient's mask: out[m] = x[m] where
 * (code[m] >= 0) == (keep_coded != 0), zero elsewhere.  Rows of D floats, D % 4 == 0. */
int hivt_rows_put_tokens(const float* x, int64_t x_ld, const int32_t* code, const float* token, int64_t M, int32_t D,
                         int32_t keep_coded, float* out, hivt_stream_t stream);

/* For each of `count` pairs of layers: out[r] = [a[r] | b[r]] (rows of ka and kb floats side by side) and
 * bias_out = bias_a + bias_b, all pairs in one launch.  The second Linear of the two branches of a
 * MultipleInputEmbedding are summed (models/embedding.py:49-52): the sum is one product of the concatenated
 * hidden rows with the weights side by side, and this call lays them side by side once per forward pass.
 * items: 9 x int64 per pair = { a, b, out, rows, ka, kb, bias_a, bias_b, bias_out } (pointers as integers; the
 * three bias pointers all null or all set). */
int hivt_stack_pairs(const int64_t* items, int32_t count, hivt_stream_t stream);

/* Input of the temporal encoder (models/local_encoder.py:230-232): for s < T
 *   out[s, n] = (padding_mask[n][s] ? padding_token[s] : x[new_id[s*N + n]]) + pos_embed[s]
 * and out[T, n] = cls_token + pos_embed[T].  new_id == NULL: x holds all T*N rows.  pad_code[(T+1)*N] receives s
 * for rows that took the padding token and -1 elsewhere (the row codes of the token's gradient). */
int hivt_temporal_input_fwd(const float* x, const int32_t* new_id, const uint8_t* padding_mask, int64_t pm_stride,
                            const float* padding_token, const float* cls_token, const float* pos_embed, int32_t T,
                            int32_t N, int32_t D, float* out, int32_t* pad_code, hivt_stream_t stream);

/* dx[r] = r < *count ? dy[rows[r]] : 0 for r < capacity: the gradient of the gather new_id above. */
int hivt_gather_live_rows(const float* dy, const int32_t* rows, const int32_t* count, int32_t capacity, int32_t D,
                          float* dx, hivt_stream_t stream);

/* Decoder output (models/decoder.py:131-139): out[o, h] = (loc[m, 2h], loc[m, 2h+1], s(raw[m, 2h]), s(raw[m, 2h+1]))
 * with s(v) = max(elu(v) + 1, min_scale), and the gradient with respect to loc and raw (clamp passes the
 * gradient where elu(v) + 1 >= min_scale).  Output rows o are mode-major (f * N + n, the reference's
 * [F, N, H, 4]); with F <= 1 the input rows are in the same order (m = o), with F > 1 they are agent-major
 * (m = n * F + f, the order in which multihead_proj leaves the modes of an agent, models/global_interactor.py:63-65). */
int hivt_decoder_output_fwd(const float* loc, int64_t loc_ld, const float* scale_raw, int64_t raw_ld, int64_t M,
                            int32_t H, int32_t F, float min_scale, float* out, hivt_stream_t stream);
int hivt_decoder_output_bwd(const float* d_out, const float* scale_raw, int64_t raw_ld, int64_t M, int32_t H, int32_t F,
                            float min_scale, float* d_loc, float* d_scale_raw, hivt_stream_t stream);

/* out[(i * nb + j) * nc + k] = (a[i] + b[j]) + c[k]: the table of summed categorical embeddings that
 * models/local_encoder.py:361-363 adds per edge (is_intersection, turn_direction, traffic_control), and the
 * gradient of the three small tables from the gradient of that table (rows added in ascending order). */
int hivt_category_table_fwd(const float* a, const float* b, const float* c, int32_t na, int32_t nb, int32_t nc, int32_t D,
                            float* out, hivt_stream_t stream);
int hivt_category_table_bwd(const float* g, int32_t na, int32_t nb, int32_t nc, int32_t D, float* da, float* db, float* dc,
                            hivt_stream_t stream);

/* out[n, h, :] = y_hat[best_mode[n], n, h, 0:2] for y_hat [F, N, H, C] (models/hivt.py:221-224), and its
 * gradient ADDED in place to d_y_hat (which already holds the gradient of the other uses of y_hat). */
int hivt_select_mode_fwd(const float* y_hat, const int64_t* best_mode, int32_t F, int32_t N, int32_t H, int32_t C,
                         float* out, hivt_stream_t stream);
int hivt_select_mode_bwd(const float* g, const int64_t* best_mode, int32_t F, int32_t N, int32_t H, int32_t C,
                         float* d_y_hat, hivt_stream_t stream);

/* out[b][c] = sum over parts p (in order) with c < widths[p] of parts[p][b * widths[p] + c]: the input gradients
 * of critics that read the first widths[p] columns of the same rows (the horizon wrappers under models/critics slice
 * their horizon out of one trajectory tensor).  parts: HOST array of 1..4 device pointers. */
int hivt_sum_padded_rows(const float* const* parts, const int32_t* widths, int32_t count, int64_t B, int32_t W,
                         float* out, hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Kernel (a): edge gather + rotation into the target's frame
 * ------------------------------------------------------------------------------------------ */

/* out[e] = [ x_src[src[p]] @ R[dst[p] % rot_mod],  attr[p] @ R[dst[p] % rot_mod] ],  p = perm[e]
 * (perm == NULL: p = e).  Replaces rotate_mat[edge_index[1]] + two torch.bmm + the x_j lift
 * (models/local_encoder.py:169-173, 358-360).  rot == NULL copies without rotating
 * (models/local_encoder.py:165-166, 352-356).  Also emits src/dst in the output order when
 * src_csr / dst_csr are non-NULL.
 *   x_src float [Ns, x_stride] (2 used), attr float [E,2], rot float [Nr,2,2] row-major,
 *   out float [E,4].  With `count` (device int32, may be NULL) rows e >= *count are padding of a
 *   fixed-capacity list and receive zeros / index 0. */
int hivt_edge_gather_rotate_fwd(const float* x_src, int64_t x_stride, const float* attr, const float* rot,
                                const int32_t* src, const int32_t* dst, const int32_t* perm, int32_t rot_mod,
                                int64_t E, const int32_t* count, float* out, int32_t* src_csr, int32_t* dst_csr,
                                hivt_stream_t stream);

/* Gradient of the above w.r.t. attr (d_attr float [E,2], original edge order, may be NULL) and, per edge,
 * w.r.t. the gathered source row (d_x_edge float [E,4] in OUTPUT order: the two components, then two zeros;
 * may be NULL).  The gradient of x_src is the sum of the d_x_edge rows of each source, which the caller
 * takes with hivt_segment_sum_rows over a CSR by source (width 4): a fixed summation order, no float
 * atomics.  Needed only when positions / features require grad (never in the reference's step). */
int hivt_edge_gather_rotate_bwd(const float* d_out, const float* rot, const int32_t* src, const int32_t* dst,
                                const int32_t* perm, int32_t rot_mod, int64_t E, const int32_t* count,
                                float* d_attr, float* d_x_edge, hivt_stream_t stream);

/* Relative pose features of GlobalInteractor.forward (models/global_interactor.py:54-58):
 * out[e] = [ attr[p] @ R[dst[p]],  cos(theta[src[p]]-theta[dst[p]]),  sin(...) ], p = perm[e]. */
int hivt_edge_relpose_fwd(const float* attr, const float* rot, const float* angles, const int32_t* src,
                          const int32_t* dst, const int32_t* perm, int64_t E, const int32_t* count, float* out,
                          int32_t* src_csr, int32_t* dst_csr, hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Kernels (b)+(c): per-edge QK dot, segment softmax over CSR rows, weighted sum of V
 * ------------------------------------------------------------------------------------------ */

/* Layout flag of the `dtype` argument of hivt_seg_attn_fwd / hivt_seg_attn_bwd (bit 16): the rows of k, v, dk and
 * dv are interleaved -- the j-th edge of target n is row j * N + n, every target has E / N rows.  That is a
 * [steps, targets, width] tensor whose targets attend over their own steps: the last position of the temporal
 * encoder's final layer, which is all the encoder's caller reads of it (models/local_encoder.py:241).  No node
 * terms and no padded tail in this layout. */
#define HIVT_ROWS_INTERLEAVED (1 << 16)
/* Flag of hivt_seg_attn_bwd's `dtype` argument (bit 17): dk and dv point at bfloat16 rows (dkv_stride counts
 * bfloat16 elements, rows on 8-byte boundaries); the gradients are rounded to nearest even where they are stored.
 * For key / value rows that a projection made whose two gradient products run in bf16 mode: those products would
 * round the rows anyway (see hivt_linear_bwd_data_half), and the kernel writes half the bytes. */
#define HIVT_DKV_BF16 (1 << 17)

/* out[n,h,:] = sum_{e in row n} drop(softmax_e(q[n,h,:] . key_e[h,:] / sqrt(D/H))) * val_e[h,:]
 * with key_e = k[e] (+ k_node[src[e]] when k_node != NULL), val_e likewise; softmax is
 * exp(s - max) / (sum + 1e-16) per (row, head) exactly as torch_geometric.utils.softmax; rows
 * without edges give zeros.  Replaces message()+aggregate of the three modules
 * (models/local_encoder.py:174-181, 364-371 and :98, :292; models/global_interactor.py:115-124, :75).
 * k / v rows are in CSR order with element stride kv_stride (so a fused [E,2D] K|V buffer works); the
 * query rows have element stride q_stride (>= D, a multiple of 4: the queries may be a column block of a wider
 * product that also holds other projections of the same rows).
 * Saves the per-(row,head) max `m_out` and denominator `den_out` (float [N,H]) for backward.
 * Dropout (p > 0) uses a counter-based generator keyed on (seed + *seed_ptr, CSR position, head);
 * seed_ptr (device uint64, may be NULL) lets a captured graph draw fresh masks on every replay.
 * E is the number of k / v rows (a fixed capacity is fine: rows beyond rowptr[N] are never read).
 * `row_order` (int32 [N], a permutation of the rows, may be NULL) is a schedule, not data: the g-th workgroup slot of
 * the launch serves row row_order[g], so with hivt_csr_build's ranking the long rows start first and the launch does
 * not end on a few of them.  Results do not depend on it.  Launch shapes that share a wave among rows ignore it. */
int hivt_seg_attn_fwd(const float* q, int64_t q_stride, const float* k, const float* v, int64_t kv_stride, const float* k_node,
                      const float* v_node, int64_t node_stride, const int32_t* src, const int32_t* rowptr,
                      const int32_t* row_order, int32_t N, int64_t E, int32_t D, int32_t H, float dropout_p,
                      uint64_t seed, const uint64_t* seed_ptr, float* out, float* m_out, float* den_out, int dtype,
                      hivt_stream_t stream);

/* Backward of hivt_seg_attn_fwd: dq float [N,D] with element stride dq_stride; dk / dv float rows in CSR order with element
 * stride dkv_stride (gradient of the per-edge key/value, i.e. also of k_node[src[e]] before it is
 * summed per source -- use hivt_segment_sum_rows over a CSR-by-source for that).  The max is
 * treated as a constant, as torch_geometric detaches it.  With zero_tail != 0 the rows
 * [rowptr[N], E) of dk / dv (padding of a fixed-capacity list) are set to zero. */
int hivt_seg_attn_bwd(const float* d_out, const float* q, int64_t q_stride, const float* k, const float* v, int64_t kv_stride,
                      const float* k_node, const float* v_node, int64_t node_stride, const int32_t* src,
                      const int32_t* rowptr, const int32_t* row_order, const float* out, const float* m,
                      const float* den, int32_t N, int64_t E, int32_t D, int32_t H, float dropout_p, uint64_t seed,
                      const uint64_t* seed_ptr, float* dq, int64_t dq_stride, float* dk, float* dv, int64_t dkv_stride, int32_t zero_tail,
                      int dtype, hivt_stream_t stream);

/* hivt_seg_attn_fwd / _bwd with in-application timing: identical computation; additionally the
 * launch records the kernel's own start and stop time stamps into the two hipEvent_t handles
 * (created by the caller with timing enabled), so a step can price this kernel without the
 * dispatch gaps that events recorded around a launch include.  No counterpart in the reference. */
int hivt_seg_attn_fwd_timed(const float* q, int64_t q_stride, const float* k, const float* v, int64_t kv_stride, const float* k_node,
                            const float* v_node, int64_t node_stride, const int32_t* src, const int32_t* rowptr,
                            const int32_t* row_order, int32_t N, int64_t E, int32_t D, int32_t H, float dropout_p,
                            uint64_t seed, const uint64_t* seed_ptr, float* out, float* m_out, float* den_out, int dtype,
                            hivt_stream_t stream, void* start_event, void* stop_event);
int hivt_seg_attn_bwd_timed(const float* d_out, const float* q, int64_t q_stride, const float* k, const float* v, int64_t kv_stride,
                            const float* k_node, const float* v_node, int64_t node_stride, const int32_t* src,
                            const int32_t* rowptr, const int32_t* row_order, const float* out, const float* m,
                            const float* den, int32_t N, int64_t E, int32_t D, int32_t H, float dropout_p,
                            uint64_t seed, const uint64_t* seed_ptr, float* dq, int64_t dq_stride, float* dk, float* dv,
                            int64_t dkv_stride, int32_t zero_tail, int dtype, hivt_stream_t stream, void* start_event,
                            void* stop_event);

/* out[r, 0:width] = sum_{i in rowptr[r]..rowptr[r+1]} rows[perm[i], 0:width] in ascending i
 * (deterministic scatter-add without atomics; perm == NULL reads rows[i]).  Used for the
 * gradient of x_j lifts (index_select backward), e.g. lin_k_node / lin_v_node outputs
 * (models/global_interactor.py:116,118).  width must be a multiple of 4. */
int hivt_segment_sum_rows(const float* rows, int64_t row_stride, int32_t width, const int32_t* rowptr,
                          const int32_t* perm, int32_t num_rows, float* out, int64_t out_stride,
                          hivt_stream_t stream);

/* hivt_segment_sum_rows for input rows of bfloat16 values (row_stride counts them; rows on 8-byte boundaries), summed
 * in fp32: the per-source sums of a dK | dV gradient that hivt_seg_attn_bwd wrote under HIVT_DKV_BF16
 * (models/global_interactor.py:116,118). */
int hivt_segment_sum_rows_half(const void* rows, int64_t row_stride, int32_t width, const int32_t* rowptr,
                               const int32_t* perm, int32_t num_rows, float* out, int64_t out_stride,
                               hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Gated update (elementwise tail of update())
 * ------------------------------------------------------------------------------------------ */

/* out = agg + sigmoid(ih + hh) * (self_ - agg)  for float [N,D] operands with element row strides.
 * Replaces the elementwise part of update() (models/local_encoder.py:187-188, 378-379;
 * models/global_interactor.py:130-131); ih = lin_ih(agg), hh = lin_hh(c), self_ = lin_self(c). */
int hivt_gated_update_fwd(const float* agg, int64_t agg_stride, const float* ih, int64_t ih_stride,
                          const float* hh, int64_t hh_stride, const float* self_, int64_t self_stride,
                          int64_t N, int32_t D, float* out, hivt_stream_t stream);

/* Gradients of the above: d_agg (direct term only, contiguous [N,D]), the gate pre-activation
 * gradient written twice -- d_ih (contiguous [N,D]) and d_hh (row stride d_hh_stride, may be NULL) --
 * and d_self (row stride d_self_stride), so that hh | self may live in one fused [N,2D] buffer. */
int hivt_gated_update_bwd(const float* d_out, const float* agg, int64_t agg_stride, const float* ih,
                          int64_t ih_stride, const float* hh, int64_t hh_stride, const float* self_,
                          int64_t self_stride, int64_t N, int32_t D, float* d_agg, float* d_ih, float* d_hh,
                          int64_t d_hh_stride, float* d_self, int64_t d_self_stride, hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense projections on the fp32 matrix cores (exact fp32: an fmaf chain over k)
 * ------------------------------------------------------------------------------------------ */

#define HIVT_EPI_NONE 0
#define HIVT_EPI_RELU 1
#define HIVT_EPI_LN 2
#define HIVT_EPI_LN_RELU 3
/* or-ed into HIVT_EPI_RELU: `stats` then points at uint32 [M, ceil(N / 32)] and receives one bit per stored element,
 * bit (n & 31) of word [m][n >> 5] = (y[m][n] > 0) after ReLU and dropout -- all that the gradient through the
 * activation needs of y (hivt_linear_bwd_data_half with HIVT_MASK_BITS reads it instead of the [M, N] tensor). */
#define HIVT_EPI_SIGN_BITS 0x100

/* y = epilogue(x w^T + bias [+ add]) for x float [M,K] (row stride x_ld), w float [N,K] (nn.Linear
 * layout, row stride w_ld), bias [N] or NULL, add [M,N] or NULL; with add_index (int32 [M], optional)
 * row m adds add[clamp(add_index[m], 0, add_rows-1)] instead -- the rows of a small embedding table, which is how the summed
 * categorical lane embeddings enter (models/local_encoder.py:349-357).  Epilogues: none, ReLU,
 * LayerNorm(gamma, beta, eps) and LayerNorm + ReLU (N <= 128); the LayerNorm epilogues also write
 * the pre-normalisation values z (optional) and per-row (mean, rstd) into stats [M,2].
 * The plain epilogue can apply a LeakyReLU (neg_slope != 1: negative values are multiplied by it; the
 * critics' activation, models/critics/base_critic.py:28-31).
 * The plain epilogues (none, ReLU) can finish with y = residual + dropout(.) (dropout_p in [0,1), mask
 * from a counter-based generator keyed on (seed [+ *seed_ptr], row * N + col); residual [M,N] or NULL),
 * which is the "x + Dropout(Linear(.))" tail of every attention and feed-forward block
 * (models/local_encoder.py:168-169,193-202,268-280).
 * Replaces nn.Linear / nn.LayerNorm / nn.ReLU chains of the per-edge embedding MLPs
 * (models/embedding.py:15-23,36-46), the K|V / Q / update / output projections
 * (models/local_encoder.py:106-113, models/global_interactor.py:79-88) and the feed-forward blocks
 * (models/local_encoder.py:117-122). */
int hivt_linear_fwd(const float* x, int64_t x_ld, const float* w, int64_t w_ld, const float* bias,
                    const float* add, int64_t add_ld, const int32_t* add_index, int32_t add_rows,
                    const float* gamma, const float* beta, float eps, int epilogue, float neg_slope, float dropout_p,
                    uint64_t seed, const uint64_t* seed_ptr, const float* residual, int64_t res_ld, int64_t M,
                    int32_t N, int32_t K, float* y, int64_t y_ld, float* z, int64_t z_ld, float* stats,
                    hivt_stream_t stream);

/* Gradient through the dropout (and ReLU) of hivt_linear_fwd's plain epilogues:
 * dz = dy * keep / (1 - dropout_p) [* (y > 0) when relu != 0; y is then the forward output and the mask
 * is not regenerated], keep = the forward's mask recomputed from (seed [+ *seed_ptr], row * N + col). */
int hivt_dropout_bwd(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, int relu, float dropout_p,
                     uint64_t seed, const uint64_t* seed_ptr, int64_t M, int32_t N, float* dz, int64_t dz_ld,
                     hivt_stream_t stream);
/* The same with dz stored as bfloat16 [M,N] (nearest even; dz_ld in elements): for a dz that only bf16 products
 * read (see hivt_linear_bwd_data_half).  N and the leading dimensions must be multiples of 4, rows aligned. */
int hivt_dropout_bwd_half(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, int relu, float dropout_p,
                          uint64_t seed, const uint64_t* seed_ptr, int64_t M, int32_t N, void* dz, int64_t dz_ld,
                          hivt_stream_t stream);

/* dx[M,K] = dy[M,N] w[N,K]  (gradient of hivt_linear_fwd w.r.t. x for the plain epilogue).
 * mask (may be NULL, [M,K] with row stride mask_ld): x was itself the output of a ReLU [+ dropout] epilogue,
 * and the gradient is carried through it in the same pass: dx = mask > 0 ? dx * mask_scale : 0, with
 * mask = that output and mask_scale = 1 / (1 - its dropout probability) -- the step from the second to the
 * first Linear of a feed-forward block (models/local_encoder.py:117-122,279).
 * add (may be NULL, [M,K] with row stride add_ld; not together with mask): dx += add, for an input that
 * also reaches the loss by another route (the aggregated messages feed both the gate's lin_ih and the
 * update itself, models/local_encoder.py:186-188). */
int hivt_linear_bwd_data(const float* dy, int64_t dy_ld, const float* w, int64_t w_ld, int64_t M, int32_t N,
                         int32_t K, const float* mask, int64_t mask_ld, float mask_scale, const float* add,
                         int64_t add_ld, float* dx, int64_t dx_ld, hivt_stream_t stream);

/* Workspace for hivt_linear_wgrad. */
size_t hivt_linear_wgrad_workspace_bytes(int64_t M, int32_t N, int32_t K);

/* dw[N,K] = dy^T x and dbias[N] = column sums of dy (dbias may be NULL), reduced over M in a fixed
 * order (row chunks, then chunk order): bitwise reproducible. */
int hivt_linear_wgrad(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, int64_t M, int32_t N,
                      int32_t K, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                      hivt_stream_t stream);

/* Several weight gradients in one call (models' backward passes hold dozens of them at a few thousand rows,
 * where a launch per product and per sum is all latency).  `items` holds `count` rows of ten int64:
 *   dy, dy_ld, x, x_ld, M, N, K, dw, dbias (0: none), flags
 * flags bit 0: the sums are added to what dw / dbias hold; bit 1: the term joins the item before it (same dw,
 * N, K; its sum is added to that item's, terms in table order; up to three terms may join one item); bit 3,
 * on a joining term: the column sums of its dy are added to the item's dbias as well (without it a joining
 * term has no bias term); bit 4: dw points into a [N, 2K] matrix whose left or right half the item fills (rows 2K
 * floats apart: the two halves of a Linear over concatenated inputs, models/decoder.py:119-126, are two items).
 * Results are those of hivt_linear_wgrad on the same
 * item (torch.nn.Linear's weight.grad / bias.grad, as `loss.backward()` fills them in
 * models/hivt.py:283,312,325): bit-identical for terms that call cuts into fewer than 64 row chunks, equal to
 * rounding for longer ones (summed in chunks of about 2048 rows).  Sixteen terms per launch. */
size_t hivt_linear_wgrad_group_workspace_bytes(const int64_t* items, int32_t count);
int hivt_linear_wgrad_group(const int64_t* items, int32_t count, void* workspace, size_t workspace_bytes,
                            hivt_stream_t stream);

/* Stand-alone LayerNorm forward over the last dimension (N <= 128): y = gamma * (x - mean) * rstd + beta,
 * stats float [M,2] = (mean, rstd) per row for hivt_ln_bwd (called with z = x, relu = 0).  Replaces the
 * nn.LayerNorm in front of the attention and feed-forward blocks (models/local_encoder.py:119-120,
 * 166-167, 254-255; models/global_interactor.py:92-93,107-108). */
int hivt_ln_fwd(const float* x, int64_t x_ld, const float* gamma, const float* beta, float eps, int64_t M, int32_t N,
                float* y, int64_t y_ld, float* stats, hivt_stream_t stream);

/* Workspace for hivt_ln_bwd. */
size_t hivt_ln_bwd_workspace_bytes(int64_t M, int32_t N);

/* Backward of the LayerNorm (+ReLU when relu != 0) epilogue: from dy, the saved pre-normalisation
 * z and stats, computes dz [M,N], dgamma [N], dbeta [N] (N <= 128).  add (may be NULL, [M,N] with
 * row stride add_ld) is added to dz: the gradient that reaches the un-normalised tensor through the
 * residual connection around the block (x + block(norm(x)), models/local_encoder.py:151-153,266-267;
 * models/global_interactor.py:104-105), so that the two routes are summed without another pass.
 * With dgamma == dbeta == NULL the kernel's per-workgroup partial sums are left in `workspace`
 * ([hivt_ln_bwd_partial_rows(M)][2][N] floats: the dgamma terms, then the dbeta terms of each workgroup) and
 * no reduction is launched: the caller sums them later, e.g. all LayerNorms of a backward pass in one
 * hivt_reduce_rows_group call (same summation order, same bits). */
int hivt_ln_bwd(const float* dy, int64_t dy_ld, const float* z, int64_t z_ld, const float* stats,
                const float* gamma, const float* beta, int relu, int64_t M, int32_t N, const float* add,
                int64_t add_ld, float* dz, int64_t dz_ld, float* dgamma, float* dbeta, void* workspace,
                size_t workspace_bytes, hivt_stream_t stream);
/* The same with dz stored as bfloat16 [M,N] (nearest even, after `add`; dz_ld in elements; N and dz_ld multiples
 * of 4): for a dz that only bf16 products read, see below. */
int hivt_ln_bwd_half(const float* dy, int64_t dy_ld, const float* z, int64_t z_ld, const float* stats,
                     const float* gamma, const float* beta, int relu, int64_t M, int32_t N, const float* add,
                     int64_t add_ld, void* dz, int64_t dz_ld, float* dgamma, float* dbeta, void* workspace,
                     size_t workspace_bytes, hivt_stream_t stream);

/* Gradients that only bf16 products consume may live in memory as bfloat16: those products round their operands to
 * bfloat16 when they stage them, so storing the rounded values changes no product and halves the bytes.  Valid
 * only while the data-gradient (weight-gradient) products are in bf16 mode (hivt_set_matmul_bf16).
 * hivt_linear_bwd_data_half: hivt_linear_bwd_data with dy read as bfloat16 [M,N] (HIVT_HALF_DY) and / or dx written
 * as bfloat16 [M,K] (HIVT_HALF_DX; rounded to nearest even after the mask / add epilogue); leading dimensions
 * count elements, rows start on 8-byte boundaries. */
#define HIVT_HALF_DY 1
#define HIVT_HALF_DX 2
/* hivt_linear_bwd_data_half: `mask` points at the bits written under HIVT_EPI_SIGN_BITS (uint32 words, mask_ld
 * words per row): dx *= mask_scale where the bit is set, dx = 0 elsewhere. */
#define HIVT_MASK_BITS 4
int hivt_linear_bwd_data_half(const void* dy, int64_t dy_ld, const float* w, int64_t w_ld, int64_t M, int32_t N,
                              int32_t K, const float* mask, int64_t mask_ld, float mask_scale, const float* add,
                              int64_t add_ld, void* dx, int64_t dx_ld, int32_t half_flags, hivt_stream_t stream);
/* hivt_linear_wgrad with dy read as bfloat16 [M,N] (half_flags = HIVT_HALF_DY); the bias gradient is then the sum
 * of the rounded values.  In hivt_linear_wgrad_group the same is bit 2 of an item's flags. */
int hivt_linear_wgrad_half(const void* dy, int64_t dy_ld, const float* x, int64_t x_ld, int64_t M, int32_t N,
                           int32_t K, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                           int32_t half_flags, hivt_stream_t stream);

/* Rows of partial sums hivt_ln_bwd leaves for M input rows (0 for M == 0). */
size_t hivt_ln_bwd_partial_rows(int64_t M);

/* count reductions over rows in shared launches (48 per launch): item i = items[8 i .. 8 i + 7] =
 * (partial float* [R, pitch], R, pitch, C, out float*, split, out1 float* or 0, accumulate):
 * out[c] (c < split) or out1[c - split] (c >= split) (+)= sum over r < R of partial[r * pitch + c], c < C, summed
 * in a fixed order (16 row slices for R >= 64, 4 below; ascending inside a slice, slices in order): the
 * parameter gradients of the LayerNorms (models/local_encoder.py:114-115, models/embedding.py:16-23) from the
 * partial sums of hivt_ln_bwd, issued once per backward pass instead of once per LayerNorm. */
int hivt_reduce_rows_group(const int64_t* items, int32_t count, hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Temporal encoder attention core (next-row component: TemporalEncoder)
 * ------------------------------------------------------------------------------------------ */

/* Causal multi-head self-attention over the time axis for all agents:
 *   out[s,n,h,:] = sum_{j<=s} drop(softmax_j(q[s,n,h,:].k[j,n,h,:] / sqrt(D/H))) v[j,n,h,:]
 * for the fused projection qkv float [S,N,3D] (q | k | v, each D wide, head h at columns h*D/H..),
 * S <= 32.  Replaces the attention inside nn.MultiheadAttention(x, x, x, attn_mask=causal) of
 * TemporalEncoderLayer._sa_block (models/local_encoder.py:271-276, mask :237-241) including its
 * dropout on the attention weights.  Saves lse float [N,H,S] for the backward.
 * Arithmetic: fp32 throughout, except for D = 128, H = 8 while hivt_set_matmul_bf16 has HIVT_BF16_FWD_SPLIT set --
 * then the two products of a head run on the matrix cores with every operand entering as a (hi, lo) pair of
 * bfloat16 numbers (sixteen mantissa bits; outputs within 1e-4 of fp64), and in the backward the rebuilt scores
 * and dP likewise while dQ, dK, dV take operands rounded to bfloat16, as the dense gradient products of that mode
 * do (gradients within 5e-3 normwise).  The dropout decisions are the same on both routes. */
int hivt_temporal_attn_fwd(const float* qkv, int32_t S, int32_t N, int32_t D, int32_t H, float dropout_p,
                           uint64_t seed, const uint64_t* seed_ptr, float* out, float* lse, hivt_stream_t stream);

/* Backward of the above: d_qkv float [S,N,3D] from d_out [S,N,D], the forward inputs / outputs. */
int hivt_temporal_attn_bwd(const float* d_out, const float* qkv, const float* out, const float* lse, int32_t S,
                           int32_t N, int32_t D, int32_t H, float dropout_p, uint64_t seed, const uint64_t* seed_ptr,
                           float* d_qkv, hivt_stream_t stream);
/* The same with d_qkv stored as bfloat16 [S,N,3D] (nearest even): its only readers are the input projection's
 * two gradient products, which round it to bfloat16 anyway while they run in bf16 mode. */
int hivt_temporal_attn_bwd_half(const float* d_out, const float* qkv, const float* out, const float* lse, int32_t S,
                                int32_t N, int32_t D, int32_t H, float dropout_p, uint64_t seed,
                                const uint64_t* seed_ptr, void* d_qkv, hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused regression / classification loss (min over modes + Laplace NLL + soft-target CE)
 * ------------------------------------------------------------------------------------------ */

/* Workspace for hivt_laplace_minmode_fwd. */
size_t hivt_laplace_workspace_bytes(int32_t N);

/* Replaces HiVT._supervised_losses (models/hivt.py:174-209) with LaplaceNLLLoss.forward
 * (losses/laplace_nll_loss.py:12-36, reduction 'mean') and SoftTargetCrossEntropyLoss.forward
 * (losses/soft_target_cross_entropy_loss.py:12-23, reduction 'mean'):
 *   l2[f,n]   = sum_h ||y_hat[f,n,h,0:2] - y[n,h]||_2 * valid[n,h]
 *   best[n]   = argmin_f l2[f,n]   (first minimum)
 *   reg_loss  = mean over valid (n,h) and 2 channels of  s + ln2 + |y - loc| / (exp(s) + 1e-6),
 *               s = clamp(y_hat[best,n,h,2:4], -10, 10)
 *   cls_loss  = mean over n with >=1 valid step of  -sum_f softmax_f(-l2/(cnt+1e-6)) * log_softmax(pi[n])_f
 * valid[n,h] = !padding_mask[n, t_hist + h].  With no valid step at all both losses are 0.
 *   y_hat float [F,N,Hf,4]; pi float [N,F]; y float [N,Hf,2]; padding_mask uint8 [N, pm_stride];
 *   losses_out float [2] = {reg, cls}; best_out int64 [N]; stats_out float [4] =
 *   {valid (n,h) count, rows with a valid step, 0, 0} (kept for backward). */
int hivt_laplace_minmode_fwd(const float* y_hat, const float* pi, const float* y, const uint8_t* padding_mask,
                             int64_t pm_stride, int32_t t_hist, int32_t F, int32_t N, int32_t Hf,
                             float* losses_out, int64_t* best_out, float* soft_out, float* stats_out,
                             void* workspace, size_t workspace_bytes, hivt_stream_t stream);

/* d_y_hat float [F,N,Hf,4] (written in full: zeros outside the best mode) and d_pi float [N,F]
 * for upstream gradients g[0] (reg) and g[1] (cls) read from device memory `grad_losses`. */
int hivt_laplace_minmode_bwd(const float* grad_losses, const float* y_hat, const float* pi, const float* y,
                             const uint8_t* padding_mask, int64_t pm_stride, int32_t t_hist, int32_t F,
                             int32_t N, int32_t Hf, const int64_t* best, const float* soft, const float* stats,
                             float* d_y_hat, float* d_pi, hivt_stream_t stream);

size_t hivt_rowsum_by_code_workspace_bytes(int64_t M, int32_t N, int32_t R);

/* out[r][c] = sum over rows m with index[m] == r of x[m][c], for R <= 32 codes and N <= 128 columns,
 * summed in a fixed order; rows whose index is outside [0, R) (e.g. -1) are left out.  It is the gradient
 * of the table handed to hivt_linear_fwd through add / add_index (x = gradient at the pre-normalisation
 * sum), replacing the scatter of per-edge gradients into the categorical embedding parameters, and the
 * gradient of per-timestep tokens broadcast over agents (bos_token, padding_token, pos_embed, cls_token:
 * models/local_encoder.py:150,230-232). */
int hivt_rowsum_by_code(const float* x, int64_t x_ld, const int32_t* index, int64_t M, int32_t N, int32_t R,
                        float* out, void* workspace, size_t workspace_bytes, hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * First stage of the embedding MLPs (two-component inputs)
 * ------------------------------------------------------------------------------------------ */

/* h[m, b*D : (b+1)*D] = ReLU(LayerNorm_b(W_b f[m, 2b:2b+2] + bias_b)) for `branches` (1 or 2) inputs of two
 * components each, D in {64, 128}.  Replaces the Linear(2, D) + LayerNorm + ReLU that opens
 * SingleInputEmbedding and every branch of MultipleInputEmbedding, and the concatenation of the branch
 * outputs (models/embedding.py:12-19,36-44,49-52).  w / bias / gamma / beta are HOST arrays of `branches`
 * device pointers (w[b] float [D,2] row-major, the others float [D], all 16-byte aligned); f rows are
 * f_ld floats apart (even, 8-byte aligned); h is float [M, branches*D] contiguous. */
int hivt_embed2_fwd(const float* f, int64_t f_ld, int32_t branches, const float* const* w, const float* const* bias,
                    const float* const* gamma, const float* const* beta, float eps, int64_t M, int32_t D, float* h,
                    hivt_stream_t stream);

size_t hivt_embed2_bwd_workspace_bytes(int64_t M, int32_t D, int32_t branches);

/* Gradients of the above from dh float [M, branches*D]; the pre-norm values are recomputed from f.
 * dparams float [branches][5*D] receives, per branch: dW (D x 2, row-major) | dbias | dgamma | dbeta.
 * With dparams NULL (M > 0) the workgroups' partial sums stay in workspace as
 * [hivt_embed2_bwd_partial_rows(M, D, branches)][branches*5*D] for the caller's own reduction (one
 * hivt_reduce_rows_group over all such buffers of a backward pass gives the same sums bit for bit).
 * df (optional, float [M, 2*branches]) receives the input gradient. */
size_t hivt_embed2_bwd_partial_rows(int64_t M, int32_t D, int32_t branches);
int hivt_embed2_bwd(const float* dh, const float* f, int64_t f_ld, int32_t branches, const float* const* w,
                    const float* const* bias, const float* const* gamma, const float* const* beta, float eps,
                    int64_t M, int32_t D, float* dparams, float* df, void* workspace, size_t workspace_bytes,
                    hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Spectral normalisation of the critics' linear layers
 * ------------------------------------------------------------------------------------------ */

#define HIVT_SPECTRAL_MAX_LAYERS 16 /* layers per call (all layers of the three critics fit one) */

/* One power iteration and the division by the singular-value estimate for `layers` weight matrices
 * in ONE launch (one workgroup per layer).  Replaces, per layer and per forward call, the hook that
 * torch.nn.utils.spectral_norm installs on the critics' linears (models/critics/base_critic.py:26-38):
 *     v = normalize(W^T u); u = normalize(W v)      (only if power_iteration != 0; u, v updated in place)
 *     sigma = u . (W v);  wn = W / sigma
 * normalize(x) = x / max(||x||_2, eps).  w[l] is float [rows[l], cols[l]] row-major, u[l] [rows[l]],
 * v[l] [cols[l]].  u_used / v_used receive the vectors sigma was formed with; they are the constants of
 * the gradient (the hook clones u and v for the same purpose).  The pointer arrays and rows / cols are
 * HOST arrays of `layers` entries; each side of a matrix is at most 512. */
int hivt_spectral_norm_fwd(int32_t layers, const float* const* w, float* const* u, float* const* v,
                           const int32_t* rows, const int32_t* cols, int32_t power_iteration, float eps,
                           float* const* wn, float* const* u_used, float* const* v_used, float* sigma,
                           hivt_stream_t stream);

/* Gradient of the above with u_used, v_used constant:
 *     d_w = d_wn / sigma - (<d_wn, w> / sigma^2) * u_used v_used^T. */
int hivt_spectral_norm_bwd(int32_t layers, const float* const* d_wn, const float* const* w,
                           const float* const* u_used, const float* const* v_used, const float* sigma,
                           const int32_t* rows, const int32_t* cols, float* const* d_w, hivt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A whole trajectory critic per call
 * ------------------------------------------------------------------------------------------ */

#define HIVT_CRITIC_MAX_LAYERS 8 /* hidden layers */

/* A critic is the MLP of models/critics/base_critic.py:19-44 over B flattened trajectories:
 *     h_l = Dropout(LeakyReLU_{neg_slope}(h_{l-1} W_l^T + b_l)), l = 1..layers;   score = h_L w_o^T + b_o.
 * widths[0..layers] = input width, hidden widths.  w / bias are HOST arrays of layers+1 device pointers
 * (W_l float [widths[l], widths[l-1]], then w_o float [1, widths[layers]]; b_l [widths[l]], b_o [1]) --
 * the spectrally normalised weights of hivt_spectral_norm_fwd.
 *
 * hivt_critic_fwd writes scores float [B] and keeps the hidden activations in `saved`
 * (hivt_critic_saved_floats floats).  With with_input_grad it also evaluates g = d score / d x per sample
 * by the chain G_L = w_o * c_L, g_{l-1} = G_l W_l, G_{l-1} = g_{l-1} * c_{l-1} (c_l = the activation /
 * dropout factor, read off the stored h_l) and writes grad_norm2[i] = ||g_i||^2 -- the quantity the R1
 * penalty averages (losses/adv_losses.py:36-47, models/hivt.py:244-252) -- without a double backward.
 * Dropout masks come from the counter-based generator keyed on (seed [+ *seed_ptr], layer, element). */
size_t hivt_critic_saved_floats(int64_t B, int32_t layers, const int32_t* widths, int32_t with_input_grad);
size_t hivt_critic_workspace_bytes(int64_t B, int32_t layers, const int32_t* widths);
int hivt_critic_fwd(const float* x, int64_t x_ld, int64_t B, int32_t layers, const int32_t* widths,
                    const float* const* w, const float* const* bias, float neg_slope, float dropout_p, uint64_t seed,
                    const uint64_t* seed_ptr, int32_t with_input_grad, float* saved, float* scores,
                    float* grad_norm2, hivt_stream_t stream);

/* Gradients of a loss with d loss / d scores = g_scores [B] (or NULL) and d loss / d grad_norm2 = g_norm2
 * [B] (or NULL) with respect to every weight (dw[l], same shapes as w[l]) and bias, and optionally the
 * input (dx float [B, widths[0]] contiguous, from the score path only: the factors c_l are constants).
 * The grad_norm2 path is the second chain e_0 = 2 g_norm2 g, dW_l += G_l^T e_{l-1},
 * e_l = (e_{l-1} W_l^T) * c_l, dw_o += colsum(e_L).
 * dw = dbias = NULL (with g_norm2 = NULL and dx given): the parameters are frozen -- the generator's pass
 * through the critics (models/hivt.py:317-327) -- and only dx is computed. */
int hivt_critic_bwd(const float* g_scores, const float* g_norm2, const float* x, int64_t x_ld, int64_t B,
                    int32_t layers, const int32_t* widths, const float* const* w, float neg_slope, float dropout_p,
                    const float* saved, float* const* dw, float* const* dbias, float* dx, void* workspace,
                    size_t workspace_bytes, hivt_stream_t stream);

/* The same passes, several per call: models/hivt.py:244-252 and :317-323 evaluate the three critics on the real
 * and on the generated trajectories one after the other; here up to six such passes share every launch (a
 * workgroup carries 64 rows of one pass through all of its layers).
 *   hivt_critics_fwd   passes: count rows of 12 int64 -- x, x_ld, B, layers, widths (host int32*), w (host
 *                      float**), bias (host float**), seed, with_input_grad, saved, scores, grad_norm2; the
 *                      meaning of every field is that of the hivt_critic_fwd argument of the same name.
 *   hivt_critics_bwd   rows of 12 int64 -- g_scores, g_norm2, x, x_ld, B, layers, widths, w, saved, dw (host
 *                      float** or 0), dbias (host float** or 0), dx; as hivt_critic_bwd.
 * Backward passes that are handed the SAME dw / dbias arrays (the passes of one critic over real and generated
 * rows, whose weight gradients the optimizer wants summed) leave the sum of their gradients there; such passes
 * must be built for the layer chain (see the last sentence).
 * Results are those of one hivt_critic_fwd / hivt_critic_bwd call per pass (scores: an fp32 dot product in
 * place of the matrix-core product, so equal to rounding).  Passes with more than four hidden layers, widths
 * that are not multiples of 4 or operands that are not 16-byte aligned take that single-pass route. */
int hivt_critics_fwd(const int64_t* passes, int32_t count, float neg_slope, float dropout_p, const uint64_t* seed_ptr,
                     hivt_stream_t stream);
size_t hivt_critics_bwd_workspace_bytes(const int64_t* passes, int32_t count);
int hivt_critics_bwd(const int64_t* passes, int32_t count, float neg_slope, float dropout_p, void* workspace,
                     size_t workspace_bytes, hivt_stream_t stream);

/* Critic loss of one scale (losses/adv_losses.py:36-58 with losses/r1_regularization.py:27-36), over all
 * B rows with per-row weights w (0/1 keep mask) instead of boolean-indexed subsets:
 *   out[0] = sum_i w_i relu(1 - s_real_i) / count + sum_i w_i relu(1 + s_fake_i) / count + lambda_r1 * out[1]
 *   out[1] = sum_i w_i grad_norm2_i / count                       (the R1 penalty)
 * count is a device scalar (number of kept rows, at least 1).  One block, fixed summation order. */
int hivt_hinge_r1_fwd(const float* s_real, const float* grad_norm2, const float* s_fake, const float* w,
                      const float* count, int64_t B, float lambda_r1, float* out, hivt_stream_t stream);

/* Gradient of out[0] above with respect to s_real, grad_norm2 and s_fake; g_loss is a device scalar. */
int hivt_hinge_r1_bwd(const float* g_loss, const float* s_real, const float* s_fake, const float* w,
                      const float* count, int64_t B, float lambda_r1, float* g_real, float* g_norm2, float* g_fake,
                      hivt_stream_t stream);

/* The critic loss of ALL scales in one launch (models/hivt.py:305-310: the per-scale terms above, summed in scale
 * order): out[0] = sum_k out[1 + 2k], out[1 + 2k] = loss of scale k, out[2 + 2k] = its R1 penalty, out[1 + 2 scales] =
 * the number of kept rows; s_real / grad_norm2 / s_fake are HOST arrays of `scales` (1..4) device pointers to B
 * rows each.  count == NULL in the forward call: the number of kept rows is max(sum_i w_i, 1), computed here (the
 * backward call takes it from out).  The gradient of out[0] goes to g_real[k], g_norm2[k], g_fake[k] (HOST arrays of
 * device pointers). */
int hivt_hinge_r1_multi_fwd(const float* const* s_real, const float* const* grad_norm2, const float* const* s_fake,
                            int32_t scales, const float* w, const float* count, int64_t B, float lambda_r1, float* out,
                            hivt_stream_t stream);
int hivt_hinge_r1_multi_bwd(const float* g_loss, const float* const* s_real, const float* const* s_fake, int32_t scales,
                            const float* w, const float* count, int64_t B, float lambda_r1, float* const* g_real,
                            float* const* g_norm2, float* const* g_fake, hivt_stream_t stream);

/* The generator's objective in one launch (models/hivt.py:317-324, losses/adv_losses.py:91-98):
 *   out[2 + k] = -sum_i w_i s_fake[k][i] / count,  out[1] = lambda_adv * sum_k out[2 + k] (scale order),
 *   out[0] = (losses[0] + losses[1]) + out[1]   with losses = (regression, classification) loss,
 *   out[2 + scales] = count (count == NULL in the forward call: max(sum_i w_i, 1), computed here).
 * Gradient of out[0]: g_losses[0] = g_losses[1] = g_out, g_s[k][i] = -lambda_adv * g_out * w_i / count. */
int hivt_generator_total_fwd(const float* losses, const float* const* s_fake, int32_t scales, const float* w,
                             const float* count, int64_t B, float lambda_adv, float* out, hivt_stream_t stream);
int hivt_generator_total_bwd(const float* g_out, int32_t scales, const float* w, const float* count, int64_t B,
                             float lambda_adv, float* g_losses, float* const* g_s, hivt_stream_t stream);

/* Guards of the training step in one pass (models/hivt.py:264,272,294): flag[0] = 1 if nan_check holds a
 * NaN, finite_check holds a NaN or an infinity, or *extra != 0 (extra: optional device byte, e.g. an
 * overflow flag); else 0.  flag is a device int32 written in full by every call (its first byte read as a
 * bool is the guard).  Either array may be NULL / empty. */
int hivt_step_guards(const float* nan_check, int64_t n_nan, const float* finite_check, int64_t n_finite,
                     const uint8_t* extra, int32_t* flag, hivt_stream_t stream);

/* AdamW update of `count` float32 tensors (torch.optim.AdamW as the reference configures it,
 * models/hivt.py:444,453: decoupled weight decay, no amsgrad):
 *   p *= 1 - lr * weight_decay;  m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * params / grads / exp_avgs / exp_avg_sqs are HOST arrays of `count` device addresses, numels the element
 * counts; `step` is a device scalar holding the number of this update (>= 1, advanced by the caller);
 * found_inf (optional device scalar) == 1 leaves every tensor untouched.  Up to 96 tensors travel in one
 * launch's argument block, cut into 4096-element pieces, one workgroup each. */
int hivt_adamw_step(const uint64_t* params, const uint64_t* grads, const uint64_t* exp_avgs,
                    const uint64_t* exp_avg_sqs, const int64_t* numels, int32_t count, const float* step,
                    const float* found_inf, float lr, float beta1, float beta2, float eps, float weight_decay,
                    hivt_stream_t stream);

/* The same update for a caller that keeps `step` = number of updates done SO FAR: the update uses *step + 1 and the
 * call then advances *step by one (one extra one-thread launch), unless found_inf == 1.  lr_ptr (optional device
 * scalar) replaces lr, so that a learning-rate schedule needs no new launch arguments (a recorded graph keeps
 * replaying while the schedule writes the scalar). */
int hivt_adamw_update(const uint64_t* params, const uint64_t* grads, const uint64_t* exp_avgs,
                      const uint64_t* exp_avg_sqs, const int64_t* numels, int32_t count, float* step,
                      const float* found_inf, const float* lr_ptr, float lr, float beta1, float beta2, float eps,
                      float weight_decay, hivt_stream_t stream);

/* Generator's adversarial term of one scale (losses/adv_losses.py:91-98): out = scale * sum_i w_i s_i / count
 * (scale = -1 for -mean D(fake)), and its gradient g_s_i = scale * g_out * w_i / count. */
int hivt_weighted_mean_fwd(const float* s, const float* w, const float* count, int64_t B, float scale, float* out,
                           hivt_stream_t stream);
int hivt_weighted_mean_bwd(const float* g_out, const float* w, const float* count, int64_t B, float scale, float* g_s,
                           hivt_stream_t stream);

/* Operand precision of hivt_linear_fwd / hivt_linear_bwd_data / hivt_linear_wgrad (and of everything built
 * on them), one bit per product: a product whose bit is clear (default: all clear) takes fp32 operands on the
 * fp32 matrix instructions, exact; with its bit set the operands are rounded to bfloat16 (nearest even) on
 * their way into LDS and multiplied by the bf16 matrix instructions with fp32 accumulation -- inputs,
 * outputs, epilogues, bias sums and reductions stay fp32.  Process-wide; returns the previous mask.  No
 * counterpart in the reference, which trains in fp32 throughout. */
#define HIVT_BF16_FWD 1   /* y = x w^T */
#define HIVT_BF16_DGRAD 2 /* dx = dy w */
#define HIVT_BF16_WGRAD 4 /* dw = dy^T x */
/* Forward product on split operands: every value enters as hi = bf16(x), lo = bf16(x - hi) and the product is
 * hi*hi + hi*lo + lo*hi on the bf16 matrix instructions (16 significant bits per operand, results within about
 * 1e-5 of the fp32 ones).  Overrides HIVT_BF16_FWD.  Rounding the forward operands to 8 bits moves the
 * parameter gradients of the network by about 1e-1 (they depend on the saved activations); rounding only the
 * gradient products moves them by under 1e-2. */
#define HIVT_BF16_FWD_SPLIT 8
int hivt_set_matmul_bf16(int mask);

#ifdef __cplusplus
}
#endif
#endif /* HIVT_MI355_H */
