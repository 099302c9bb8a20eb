/*
 * mecha_hip.h -- C ABI of libmecha_hip.so, the MI355X (gfx950) backend for
 * sd-mecha's per-key merge hot path.
 *
 * The reference (ljleb/sd-mecha v1.1.5) has no FFI of its own: the boundary it
 * offers is the Python call MergeMethod.merge_key -> fn(*tensors, **hyper)
 * (sd_mecha/extensions/merge_methods.py:304-337, called from
 * sd_mecha/merging.py:418-426).  Every entry point below replaces the torch op
 * sequence of one builtin method body reached through that call; the comment on
 * each cites it (paths relative to /root/reference/sd_mecha).  INTEGRATION.md
 * shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types, no exceptions.
 *  - All tensor pointers are DEVICE pointers to contiguous data on the current
 *    HIP device; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *    Work is only enqueued: nothing here synchronises the device or the stream
 *    unless its comment says so.  The library never frees or retains caller memory.
 *  - dtypes: MECHA_F32 / MECHA_F16 / MECHA_BF16.  Arithmetic is IEEE fp32
 *    ("merge_dtype", merging.py:116-117); inputs are upcast exactly
 *    (logistics.py:53), outputs rounded once, to nearest even, to `out_dtype`
 *    (root cast node, merging.py:157-158).
 *  - `flags` (may be NULL): device uint32 word(s), OR-ed into, never cleared here.
 *    bit 0 (MECHA_FLAG_INPUT_NONFINITE): some input element was inf/NaN -- the
 *    caller turns it into NonFiniteStateDictKeyError = "skip this key"
 *    (logistics.py:132-142).  bit 1 (MECHA_FLAG_OUTPUT_NONFINITE): some output
 *    element is inf/NaN -- warning only (merging.py:297-308).
 *  - Return value: 0 on success, negative on error; mecha_last_error() returns
 *    the message of the calling thread's last error.
 *  - Workspaces are caller-provided device memory sized by the *_workspace_bytes
 *    query next to each entry point; they need no initialisation.
 *  - Re-entrant: no global mutable state besides the per-thread error string.
 */
#ifndef MECHA_HIP_H
#define MECHA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MECHA_ABI_VERSION 1

#define MECHA_F32 0
#define MECHA_F16 1
#define MECHA_BF16 2

#define MECHA_FLAG_INPUT_NONFINITE 1u
#define MECHA_FLAG_OUTPUT_NONFINITE 2u

#define MECHA_MAX_MODELS 32

#define MECHA_OK 0
#define MECHA_ERR_ARG (-1)
#define MECHA_ERR_HIP (-2)
#define MECHA_ERR_WORKSPACE (-3)

int mecha_abi_version(void);
const char *mecha_last_error(void);

/* Per-kernel timers (no counterpart in the reference, which has no timers or profiler: SURVEY.md §5).
 * Off by default.  While enabled, every kernel launch of the library is bracketed by HIP events recorded
 * on the launch's own stream; mecha_profile_read(id) waits for the recorded launches and returns the summed
 * elapsed time and the launch count of kernel family `id` (0 <= id < mecha_profile_kernel_count()).
 * bench.py derives `roofline.achieved` from these. */
int mecha_profile_enable(int on);
int mecha_profile_kernel_count(void);
const char *mecha_profile_kernel_name(int id);
int mecha_profile_read(int id, double *total_ms, long long *launches, int reset);

/* A run of elements that belongs to one state-dict key inside per-model arenas
 * (the key scheduler's HBM layout, DESIGN.md "data layout").  `offset` is in
 * ELEMENTS from each arena base and must be a multiple of 8; `numel` is the
 * key's element count.  Batched entry points take a DEVICE array of these. */
typedef struct mecha_segment {
    int64_t offset;
    int64_t numel;
} mecha_segment;

/* ------------------------------------------------------------------ per key
 * One call = one state-dict key = one call of the reference method body. */

/* cast (extensions/builtin/merge_methods/logistics.py:38-53) fused with the
 * finite check of omit_non_finite (logistics.py:132-142). out may be NULL
 * (check only). */
int mecha_cast(const void *a, void *out, int64_t numel, int in_dtype, int out_dtype,
               uint32_t *flags, void *stream);

/* weighted_sum (extensions/builtin/merge_methods/linear.py:7-23): torch.lerp(a, b, alpha)
 * with the alpha==0 -> a, alpha==1 -> b shortcuts of linear.py:18-21 (the shortcut
 * paths read only that one input, as the lazy StateDict inputs of the reference do). */
int mecha_weighted_sum(const void *a, const void *b, void *out, int64_t numel,
                       int in_dtype, int out_dtype, float alpha, uint32_t *flags, void *stream);

/* weighted_sum / add_difference with a FULL alpha tensor (linear.py:16, :23, :71, :76: `alpha.numel() > 1`, no 0 / 1
 * shortcut -- block weights, per-row or per-element weights).  `alpha`: DEVICE fp32, one weight per element of `a` (the shim
 * broadcasts the reference's tensor to the shape of `a`), 32-byte aligned for the vector path.  The same single fma per
 * element as the scalar forms (torch's lerp_vec blends coefficient and base per element; addcmul = fma(b, alpha, a)).
 * The weights are hyper-parameters: they take no part in the omit_non_finite flag.  Algorithmic bytes: + 4 per element. */
int mecha_weighted_sum_tensor(const void *a, const void *b, const float *alpha, void *out, int64_t numel,
                              int in_dtype, int out_dtype, uint32_t *flags, void *stream);
int mecha_add_difference_tensor(const void *a, const void *b, const float *alpha, void *out, int64_t numel,
                                int in_dtype, int out_dtype, uint32_t *flags, void *stream);

/* n_average (linear.py:26-40): running mean res = lerp(res, m_i, 1/i), i = 2..N.
 * `models` is a HOST array of n_models device pointers (1 <= n_models <= MECHA_MAX_MODELS). */
int mecha_n_average(const void *const *models, int n_models, void *out, int64_t numel,
                    int in_dtype, int out_dtype, uint32_t *flags, void *stream);

/* add_difference (linear.py:64-76): a.addcmul(b, alpha) = fma(b, alpha, a); alpha==0 -> a. */
int mecha_add_difference(const void *a, const void *b, void *out, int64_t numel,
                         int in_dtype, int out_dtype, float alpha, uint32_t *flags, void *stream);

/* subtract (linear.py:96-101): a - b. */
int mecha_subtract(const void *a, const void *b, void *out, int64_t numel,
                   int in_dtype, int out_dtype, uint32_t *flags, void *stream);

/* scale (linear.py:79-93): a * alpha (the alpha==0 -> python 0.0 case stays in the host shim). */
int mecha_scale(const void *a, void *out, int64_t numel, int in_dtype, int out_dtype,
                float alpha, uint32_t *flags, void *stream);

/* slerp (linear.py:43-61), perpendicular_component (linear.py:104-113) and the
 * add_perpendicular graph (wrappers.py:42-64: a + alpha * perp(a - c, b - c)).
 * Two-pass: block-partial reductions into `workspace`, then the elementwise apply. */
size_t mecha_reduce_workspace_bytes(int64_t numel);
int mecha_slerp(const void *a, const void *b, void *out, int64_t numel, int in_dtype, int out_dtype,
                float alpha, void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream);
int mecha_perpendicular_component(const void *a, const void *b, void *out, int64_t numel,
                                  int in_dtype, int out_dtype, void *workspace, size_t workspace_bytes,
                                  uint32_t *flags, void *stream);
int mecha_add_perpendicular(const void *a, const void *b, const void *c, void *out, int64_t numel,
                            int in_dtype, int out_dtype, float alpha, void *workspace,
                            size_t workspace_bytes, uint32_t *flags, void *stream);

/* ties_sum (extensions/builtin/merge_methods/ties.py:98-125) over n_models deltas:
 * per-delta magnitude threshold = kth-smallest |delta| with kth = max(int((1-k)*numel), 1)
 * (ties.py:128-132, exact radix select on the fp32 bit pattern), sign election
 * (sign of the sum of signs, or of the sum of deltas if vote_sgn), disjoint mean,
 * nan_to_num.  `thresholds_out` (may be NULL): device float[n_models], receives the
 * selected thresholds.  `deltas` is a HOST array of device pointers. */
size_t mecha_ties_workspace_bytes(int n_models, int64_t numel);
int mecha_ties_sum(const void *const *deltas, int n_models, void *out, int64_t numel,
                   int in_dtype, int out_dtype, double k, int vote_sgn, float *thresholds_out,
                   void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream);

/* The TIES recipe graph of wrappers.py:70-94 in one fused pipeline:
 *   out = add_difference(base, ties_sum(m_0 - base, ..., m_{n-1} - base, k), alpha)
 * (subtract linear.py:101, ties_sum ties.py:98-125, add_difference linear.py:76).
 * Reads base and every model once in the selection pass (three more times for a (key, delta)
 * pair whose bracket misses) and once in the merge pass; no fp32 delta is ever materialised.
 * Environment variable MECHA_TIES_SELECT=radix|fail (tests): force the radix-select pipeline /
 * force every bracket to miss. */
int mecha_add_difference_ties(const void *base, const void *const *models, int n_models, void *out,
                              int64_t numel, int in_dtype, int out_dtype, double k, float alpha,
                              int vote_sgn, float *thresholds_out, void *workspace,
                              size_t workspace_bytes, uint32_t *flags, void *stream);

/* The same with NO caller workspace and no allocation of any kind.  No counterpart in the reference: it is for callers
 * that cannot allocate -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  The kernels
 * use a scratch array that is part of the library's device image (8 MiB per GPU the library runs on, held until the process
 * ends).  Limits: 1..8 models, numel <= 2^26 (every SDXL / FLUX key); beyond them the call fails with MECHA_ERR_ARG and a
 * message naming the limit.  Calls no hipMalloc / hipFree or allocating routine; the first call on a device looks the
 * array's address up (hipGetSymbolAddress: not a stream operation, so make that first call OUTSIDE graph capture -- one warm-up
 * call per device; later calls only enqueue kernels and memsets, and with mecha_set_profiling off nothing else).  Every call on
 * a GPU shares that GPU's array (the call clears what must start
 * at zero): calls that OVERLAP on one GPU -- two streams, two threads -- race; serialise them or use the workspace form. */
int mecha_add_difference_ties_noalloc(const void *base, const void *const *models, int n_models, void *out,
                                      int64_t numel, int in_dtype, int out_dtype, double k, float alpha,
                                      int vote_sgn, float *thresholds_out, uint32_t *flags, void *stream);

/* Diagnostics of the last TIES call that used `workspace` (sized for the same n_models,
 * n_segments (1 for the per-key entry points), total_elems (numel for per-key)): how many
 * (key, delta) pairs fell back to the radix select, and how many elements went through the
 * ambiguous lists.  Synchronises `stream`.  No counterpart in the reference (torch.kthvalue,
 * ties.py:130, has no such state); used by bench.py and the tests. */
int mecha_ties_select_stats(const void *workspace, int n_models, int n_segments, int64_t total_elems,
                            unsigned long long *failed_pairs, unsigned long long *ambiguous_elems,
                            void *stream);

/* DARE: ties_sum_with_dropout (ties.py:10-50) with della_eps == 0, apply_median == False and
 * apply_stock == True (t = 1, ties.py:90).  `keep_bits`: device uint32[n_models][ceil(numel/32)],
 * bit (i & 31) of word (i >> 5) = Bernoulli keep mask of element i of that delta.  For results
 * identical to the reference's CPU generator stream, draw them with mecha_dare_draw_masks_host.
 * out = ties_sum(delta_m * mask_m ...) / (rescale ? 1 - probability : 1). */
int mecha_ties_sum_with_dropout(const void *const *deltas, int n_models, const uint32_t *keep_bits,
                                void *out, int64_t numel, int in_dtype, int out_dtype, double k,
                                int vote_sgn, float probability, int rescale, float *thresholds_out,
                                void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream);

/* ties_sum_extended (ties.py:68-94) with apply_median == False, optionally behind the DARE masks of
 * ties_sum_with_dropout (ties.py:10-50; keep_bits == NULL: no dropout, probability / rescale ignored).
 * model_stock != 0 applies the Model-Stock factor of ties.py:146-159 -- what the reference does for
 * apply_stock == False, its default (ties.py:90): per slice of the key's last dimension (`row_len`
 * elements; numel for 1-D keys) t = n c / (1 + (n - 1) c), c = mean cosine of consecutive filtered deltas
 * (CosineSimilarity(dim=-1, eps=cos_eps)); out = nan_to_num(sum * t / count) [/ (1 - probability)].
 * 2..8 deltas with the factor.  model_stock == 0 is mecha_ties_sum / mecha_ties_sum_with_dropout. */
size_t mecha_ties_extended_workspace_bytes(int n_models, int64_t numel, int64_t row_len);
int mecha_ties_sum_extended(const void *const *deltas, int n_models, const uint32_t *keep_bits, void *out,
                            int64_t numel, int64_t row_len, int in_dtype, int out_dtype, double k,
                            int vote_sgn, int model_stock, float cos_eps, float probability, int rescale,
                            float *thresholds_out, void *workspace, size_t workspace_bytes,
                            uint32_t *flags, void *stream);

/* model_stock (ties.py:137-159): out = nan_to_num(t * (sum(deltas) / n)) with the Model-Stock factor t of ties.py:146-159 per slice
 * of the key's last dimension (`row_len` elements): t = n c / (1 + (n - 1) c), c = mean cosine of consecutive deltas
 * (CosineSimilarity(dim=-1, eps=cos_eps)).  nan_to_num with torch's defaults: nan -> 0, +-inf -> +-FLT_MAX.  2..8 deltas. */
size_t mecha_model_stock_workspace_bytes(int n_models, int64_t numel, int64_t row_len);
int mecha_model_stock(const void *const *deltas, int n_models, void *out, int64_t numel, int64_t row_len, int in_dtype,
                      int out_dtype, float cos_eps, void *workspace, size_t workspace_bytes, uint32_t *flags, void *stream);

/* dropout, "n-supermario" (ties.py:202-238) at odd whole `overlap` (its default): delta m keeps the elements whose bit is set in
 * ITS mask (`keep_bits`: device uint32[n_models][ceil(numel/32)], drawn by the host shim with numpy's default_rng(seed).binomial
 * exactly as the reference draws them, :214-217);  out = sum_m masked delta_m / max(number of masks set, 1) / rescalar,
 * rescalar = (1 - probability) ** rescale (the shim applies :231-235). */
int mecha_dropout(const void *const *deltas, int n_models, const uint32_t *keep_bits, void *out, int64_t numel, int in_dtype,
                  int out_dtype, float rescalar, uint32_t *flags, void *stream);

/* HOST function: the Bernoulli keep masks the reference draws on CPU (ties.py:29-33, :63-64):
 * torch.Generator().manual_seed(seed), then torch.bernoulli(1 - probability) per delta in order
 * (MT19937, one 24-bit uniform per element).  Writes bit-packed masks, layout as `keep_bits`
 * above, into HOST memory `bits_out` (n_models * ceil(numel/32) words). */
int mecha_dare_draw_masks_host(uint64_t seed, float probability, int n_models, int64_t numel,
                               uint32_t *bits_out);
/* HOST function: the same generator as ONE stream: keep bit of draw i = bit (i & 31) of word (i >> 5), for the first n_bits
 * draws.  `bits_out`: ceil(n_bits / 32) + 1 words (the last one is a zero pad word the kernels may read).  The reference
 * creates and seeds a fresh generator for EVERY key (ties.py:29-31), so every key's masks are cut from the same stream:
 * delta m of a key of n elements uses draws [m n, (m + 1) n).  One stream of n_models * (largest key) bits therefore
 * serves a whole checkpoint (mecha_batched_add_difference_dare). */
int mecha_dare_draw_stream_host(uint64_t seed, float probability, int64_t n_bits, uint32_t *bits_out);

/* LoRA / LyCORIS reconstruction (extensions/builtin/lycoris.py:226-263, LoraAlgorithm.compose / compose_alpha) fused
 * with add_difference (linear.py:64-76) for every network of one key:
 *   out = base; for each network i in order: out = fma((up_i @ down_i) * scales[i], weights[i], out)
 * up_i is [rows, ranks[i]] row-major (up.view(out, -1)), down_i is [ranks[i], cols] (down.view(r, -1)); both of dtype
 * `lora_dtype` (MECHA_F16 or MECHA_BF16: the MFMA path multiplies 16-bit operands exactly, accumulates in fp32).
 * scales[i] = alpha_i / r_i (compose_alpha, lycoris.py:258-263) or 1 (compose); weights[i] = the add_difference alpha
 * (a network with weight == 0 is skipped, linear.py:72-73).  base == NULL computes the (sum of) deltas only.
 * Consecutive networks with bit-equal (scale, weight) are accumulated in one fp32 MFMA chain and folded once:
 * fma((sum_i up_i @ down_i) * scale, weight, out) -- against a fold per network this moves results by fp32 rounding of the
 * partial sums only (the reference's product is BLAS-ordered: parity is <= 1 ulp of the output dtype).
 * ups / downs / ranks / scales / weights are HOST arrays of n_networks entries (<= MECHA_MAX_MODELS). */
int mecha_lora_accumulate(const void *base, void *out, int64_t rows, int64_t cols, int base_dtype, int out_dtype,
                          const void *const *ups, const void *const *downs, const int32_t *ranks, const float *scales,
                          const float *weights, int n_networks, int lora_dtype, uint32_t *flags, void *stream);

/* The same for every key of a batch in ONE launch (the reference evaluates one key per future,
 * merging.py:223-229; a launch per key costs more than the key's tiles).  Base and output keys sit in
 * two arenas at the same element offsets.  `keys` (DEVICE) / `keys_host` (HOST copy) hold n_keys
 * entries; key i uses nets[net_begin .. net_begin + net_count) of the DEVICE table `nets`
 * (net_count <= MECHA_MAX_MODELS; networks with weight 0 must be left out by the caller);
 * tile_start = running sum of ceil(rows/MECHA_LORA_TILE) * ceil(cols/MECHA_LORA_TILE) over the keys before it.
 * offset % 4 == 0 and cols % 4 == 0 enable the vector path; with cols % 8 == 0, every rank a multiple of 32
 * and 16-byte aligned factors a key takes the fast kernel.  `nets_host` (HOST copy of `nets`, may be NULL) lets
 * the call launch only the kernel(s) the batch needs; without it both are launched (each skips the other's tiles).
 * `key_flags` (may be NULL): DEVICE uint32[n_keys]. */
#define MECHA_LORA_TILE 128
typedef struct mecha_lora_key {
    int64_t offset;       /* first element of the key inside the base / output arenas */
    int64_t rows, cols;   /* the key viewed as [out, in_flat] */
    int64_t tile_start;
    int32_t net_begin, net_count;
} mecha_lora_key;
typedef struct mecha_lora_net {
    const void *up;       /* [rows, rank] */
    const void *down;     /* [rank, cols] */
    int32_t rank;
    float scale, weight;
    int32_t reserved;
} mecha_lora_net;
int mecha_batched_lora_accumulate(const void *base_arena, void *out_arena, const mecha_lora_key *keys,
                                  const mecha_lora_key *keys_host, int n_keys, const mecha_lora_net *nets,
                                  const mecha_lora_net *nets_host, int base_dtype, int out_dtype, int lora_dtype,
                                  uint32_t *key_flags, void *stream);

/* rotate (extensions/builtin/merge_methods/svd.py:10-86) at alignment = 1, alpha = 0 for one key already reshaped to
 * n_rows x n_cols (svd.py:49-53): centre both matrices by their column means * centralization (svd.py:67-70),
 * orthogonal_procrustes (svd.py:157-169; FullRankOrthogonalMatmul :194-206 for n_rows >= n_cols, LowRankOrthogonalMatmul
 * :172-191 otherwise) with every contraction on f32 MFMA; U V^T of the reference's SVD is computed as the orthogonal
 * polar factor by a fixed schedule of 14 odd-quintic iteration steps (GEMMs only, no SVD; the optimal composition from
 * [1e-7, 1], DESIGN.md "rotate"); apply the transform to the centred `a`, add b's mean (svd.py:84-85).  The early-outs of
 * svd.py:37-47 stay in the host shim (mecha_allclose_half implements the torch.allclose(a.half(), b.half()) test of :46).
 * Only enqueues on `stream`: nothing is read back, keys on different streams overlap.  A singular value below 1e-7 of
 * the largest (a centred square key has one) ends between 0 and 1 instead of LAPACK's noise-determined +-1.
 * `sweeps_out` (HOST int, may be NULL) receives the number of iteration steps ENQUEUED (14 for the one polar solve of a
 * tall key; 2 x (6 + 14) for the two solves of a wide key, whose last 14 exit on the device when the first 6 converged;
 * 0 when `transform_in` is given). */
size_t mecha_rotate_workspace_bytes(int64_t n_rows, int64_t n_cols);
int mecha_rotate(const void *a, const void *b, void *out, int64_t n_rows, int64_t n_cols, int in_dtype, int out_dtype,
                 float centralization, void *workspace, size_t workspace_bytes, int *sweeps_out, uint32_t *flags, void *stream);
/* rotate with the reference's other whole-number settings and its factor cache (svd.py:55-85):
 *   alignment in {-1, 0, 1}: the power of the transform applied to `a` (svd.py:84; FullRankOrthogonalMatmul :198-203 with
 *     fractional_orthogonal_matrix_power :209-215, LowRankOrthogonalMatmul :177-183); the mean added back is
 *     lerp(a_mean, b_mean, alignment) (svd.py:85).  Fractional alignment (eigendecomposition / Stiefel log, :216-219,
 *     :184-188) is outside this path and is refused.
 *   alpha (scalar): a_2d = lerp(a_2d, transform(b_2d, -1), alpha) before the transform is applied (svd.py:81-82);
 *     fused into the epilogue of the GEMM that forms transform(b_2d, -1).  alignment == 0 with alpha == 0 is refused
 *     (the shim returns `a`, svd.py:39-40).
 *   transform_out (DEVICE fp32, may be NULL): receives the transform, mecha_rotate_transform_floats(n_rows, n_cols)
 *     floats: R (n_cols x n_cols) for n_rows >= n_cols, else Qb^T then Z (n_rows x n_cols each; U V^T = Z^T Qb^T).
 *   transform_in (DEVICE fp32, 16-byte aligned, may be NULL): a transform saved earlier for the same key and
 *     centralization; the Procrustes solve is skipped (svd.py:74-75 `cache["transform"]`).  The shim keeps the saved
 *     copy on the host in fp16 as the reference does (svd.py:79). */
int64_t mecha_rotate_transform_floats(int64_t n_rows, int64_t n_cols);
int mecha_rotate_ex(const void *a, const void *b, void *out, int64_t n_rows, int64_t n_cols, int in_dtype, int out_dtype,
                    float centralization, int alignment, float alpha, const float *transform_in, float *transform_out,
                    void *workspace, size_t workspace_bytes, int *sweeps_out, uint32_t *flags, void *stream);
int mecha_allclose_half(const void *a, const void *b, int64_t numel, int in_dtype, uint32_t *not_close, void *stream);
/* truncate_rank, exact mode (reference sd_mecha/extensions/builtin/merge_methods/svd.py:89-143 with
 * use_approximate_basis=False: u[:, :r] * s[:r] @ vh[:r] of torch.linalg.svd(a.flatten(start_dim=1))): the best rank-`rank`
 * approximation of one key, n_rows x n_cols row-major.  By Eckart-Young that matrix is A P with P the orthogonal projector on the
 * `rank` leading eigenvectors of the smaller Gram matrix W; P = (I + sign(W - tau I)) / 2 for any tau between the rank-th and the
 * next eigenvalue, sign() = the polar factor rotate's solver computes from matrix products alone, and tau comes from a bisection on
 * the eigenvalue count (q + trace sign) / 2, one polar solve per trial (DESIGN.md "truncate_rank").  0 < rank < min(n_rows, n_cols):
 * the reference's early-outs (svd.py:99-100 keys of < 2 dimensions, :111-116 rank >= max -> a, rank 0 -> zeros) stay in the host
 * shim.  The approximate-basis mode (svd.py:126-127, a seeded random sketch) approximates this same matrix and has no parity
 * target of its own; the host shim runs this entry point for both.  SYNCHRONISES `stream` once per trial (two doubles come back);
 * *trials_out (HOST int, may be NULL) = polar solves run (typically 8-16).  Singular values below ~3e-4 of the largest are under the
 * fp32 resolution of W: a cut among them keeps the largest count <= rank that could be resolved. */
size_t mecha_truncate_rank_workspace_bytes(int64_t n_rows, int64_t n_cols);
int mecha_truncate_rank(const void *a, void *out, int64_t n_rows, int64_t n_cols, int64_t rank, int in_dtype, int out_dtype, void *workspace,
                        size_t workspace_bytes, int *trials_out, uint32_t *flags, void *stream);
/* rotate for every key of ONE 2-D shape in one launch sequence (svd.py:10-86 per key; replaces one future per key,
 * merging.py:223-229, for a recipe rotate(model, model)): the keys lie in the per-model arenas at key_offsets[i] elements
 * (DEVICE int64[n_keys]); every GEMM, norm and centring launch covers all keys (the key index is a grid dimension), each
 * key with its own slice of `workspace`.  Wide keys (n_rows < n_cols) first run the 6-step polar schedule made for
 * [5e-3, 1]; a Gram matrix of the result decides per key, on the device, whether all singular values have arrived, and
 * only the keys that have not go on through the 14-step schedule (their blocks exit on a per-key word; nothing is read
 * back).  svd.py:46's allclose(a.half(), b.half()) early-out is evaluated on the device too: such keys get
 * lerp(a, b, alpha) (svd.py:47).  `seg_flags`: DEVICE uint32[n_keys].  The factor cache is a per-key feature
 * (mecha_rotate_ex). */
size_t mecha_batched_rotate_workspace_bytes(int64_t n_rows, int64_t n_cols, int n_keys);
int mecha_batched_rotate(const void *a_arena, const void *b_arena, void *out_arena, const int64_t *key_offsets, int n_keys,
                         int64_t n_rows, int64_t n_cols, int in_dtype, int out_dtype, float centralization, int alignment,
                         float alpha, void *workspace, size_t workspace_bytes, uint32_t *seg_flags, void *stream);

/* ------------------------------------------------------------------ batched
 * The key scheduler's form (replaces the per-key futures loop of merging.py:223-229
 * for recognised recipe graphs): every model's shard lives in one arena with
 * identical key offsets; ONE launch sequence merges n_segments keys.
 * `arenas`: HOST array of device arena base pointers; `segments`: DEVICE array;
 * `seg_flags` (may be NULL): DEVICE uint32[n_segments], per-key flag words. */
int mecha_batched_weighted_sum(const void *a_arena, const void *b_arena, void *out_arena,
                               const mecha_segment *segments, int n_segments, int64_t total_elems,
                               int in_dtype, int out_dtype, float alpha, uint32_t *seg_flags, void *stream);
int mecha_batched_n_average(const void *const *arenas, int n_models, void *out_arena,
                            const mecha_segment *segments, int n_segments, int64_t total_elems,
                            int in_dtype, int out_dtype, uint32_t *seg_flags, void *stream);
int mecha_batched_add_difference(const void *a_arena, const void *b_arena, void *out_arena,
                                 const mecha_segment *segments, int n_segments, int64_t total_elems,
                                 int in_dtype, int out_dtype, float alpha, uint32_t *seg_flags, void *stream);

/* add_difference(a, subtract(b, c), alpha) (wrappers.py:9-39 with `c`; linear.py:101, :76): fma(b - c, alpha, a) in one pass,
 * per key and for an arena of keys.  alpha == 0 is refused (the reference returns `a` without loading b or c). */
int mecha_add_difference_sub(const void *a, const void *b, const void *c, void *out, int64_t numel, int in_dtype,
                             int out_dtype, float alpha, uint32_t *flags, void *stream);
int mecha_batched_add_difference_sub(const void *a_arena, const void *b_arena, const void *c_arena, void *out_arena,
                                     const mecha_segment *segments, int n_segments, int64_t total_elems, int in_dtype,
                                     int out_dtype, float alpha, uint32_t *seg_flags, void *stream);

/* slerp (linear.py:43-61), perpendicular_component (linear.py:104-113) and the add_perpendicular graph (wrappers.py:42-64)
 * for every key of an arena in one launch sequence.  The norms and dot products are per key (whole-tensor reductions in
 * the reference): the arena is cut into 2048-element tiles that never straddle a key, a flat grid writes one fp64 partial
 * per tile, one wave per key folds its tiles in order (no atomics: the result does not depend on block scheduling), a
 * second flat grid applies the formula.  Keys whose result holds a NaN are rewritten with the reference's fallback
 * (lerp / zeros) by a kernel that exits at once when no key needs it.  Two reads of each input, one write. */
size_t mecha_batched_reduce_workspace_bytes(int n_segments, int64_t total_elems);
int mecha_batched_slerp(const void *a_arena, const void *b_arena, void *out_arena, const mecha_segment *segments,
                        const mecha_segment *segments_host, int n_segments, int64_t total_elems, int in_dtype,
                        int out_dtype, float alpha, void *workspace, size_t workspace_bytes, uint32_t *seg_flags,
                        void *stream);
int mecha_batched_perpendicular_component(const void *a_arena, const void *b_arena, void *out_arena,
                                          const mecha_segment *segments, const mecha_segment *segments_host,
                                          int n_segments, int64_t total_elems, int in_dtype, int out_dtype,
                                          void *workspace, size_t workspace_bytes, uint32_t *seg_flags, void *stream);
int mecha_batched_add_perpendicular(const void *a_arena, const void *b_arena, const void *c_arena, void *out_arena,
                                    const mecha_segment *segments, const mecha_segment *segments_host, int n_segments,
                                    int64_t total_elems, int in_dtype, int out_dtype, float alpha, void *workspace,
                                    size_t workspace_bytes, uint32_t *seg_flags, void *stream);

/* Batched add_difference(base, ties_sum(m_i - base ..., k), alpha): per-key thresholds for every
 * (key, model) pair, all keys in one launch sequence.  `segments_host` is the HOST copy of
 * `segments` (needed to size the launches); `thresholds_out` (may be NULL): device
 * float[n_segments][n_models]. */
size_t mecha_batched_ties_workspace_bytes(int n_models, int n_segments, int64_t total_elems);
int mecha_batched_add_difference_ties(const void *base_arena, const void *const *model_arenas, int n_models,
                                      void *out_arena, const mecha_segment *segments,
                                      const mecha_segment *segments_host, int n_segments,
                                      int64_t total_elems, int in_dtype, int out_dtype, double k,
                                      float alpha, int vote_sgn, float *thresholds_out, void *workspace,
                                      size_t workspace_bytes, uint32_t *seg_flags, void *stream);

/* Batched DARE (wrappers.py:231-273, della_eps == 0, apply_median == False):
 *   add_difference(base, ties_sum_with_dropout(m_i - base ..., probability, rescale, k, vote_sgn, seed, apply_stock), alpha)
 * for every key of the arenas.  `mask_stream`: DEVICE copy of mecha_dare_draw_stream_host's output, at least
 * n_models * (largest key) bits.  Thresholds are those of the MASKED deltas (ties.py:33-34), by the radix-select passes.
 * model_stock != 0 (the reference's apply_stock == False, its default, ties.py:90) applies the Model-Stock factor per slice of
 * each key's last dimension: `row_lens` (DEVICE) / `row_lens_host` (HOST) int64[n_segments] hold that dimension per key
 * (numel for 1-D keys; 0-d keys have none -- the reference raises, the caller must keep them out), 2..8 models.
 * Workspace: mecha_batched_dare_workspace_bytes (total_rows = sum of numel / row_len over the keys, 0 without the factor). */
size_t mecha_batched_dare_workspace_bytes(int n_models, int n_segments, int64_t total_elems, int64_t total_rows);
int mecha_batched_add_difference_dare(const void *base_arena, const void *const *model_arenas, int n_models, void *out_arena,
                                      const mecha_segment *segments, const mecha_segment *segments_host, int n_segments,
                                      int64_t total_elems, int in_dtype, int out_dtype, double k, float alpha, int vote_sgn,
                                      float probability, int rescale, const uint32_t *mask_stream, int64_t mask_stream_bits,
                                      int model_stock, float cos_eps, const int64_t *row_lens, const int64_t *row_lens_host,
                                      float *thresholds_out, void *workspace, size_t workspace_bytes, uint32_t *seg_flags,
                                      void *stream);

/* HOST functions: the file traffic of one wave of keys, between safetensors files and pinned
 * staging arenas, moved by up to `threads` workers in pieces of 4 MiB.
 * mecha_host_pread_many replaces InSafetensorsDict._load_tensor's locked readinto window + clone
 * (streaming.py:151-178) for the batched path: entry i reads nbytes[i] bytes at file offset
 * offsets[i] of fds[i] into dst[i]; fds[i] < 0 zero-fills dst[i] instead (arena gaps).
 * mecha_host_pwrite_many replaces OutSafetensorsDict's per-thread buffer flushed under one lock
 * (streaming.py:277-308): entry i writes nbytes[i] bytes from src[i] at file offset offsets[i].
 * All arrays are host arrays of length n.  Return 0, or < 0 with mecha_last_error() set. */
int mecha_host_pread_many(const int *fds, void *const *dst, const size_t *nbytes,
                          const int64_t *offsets, int64_t n, int threads);
int mecha_host_pwrite_many(int fd, const void *const *src, const size_t *nbytes,
                           const int64_t *offsets, int64_t n, int threads);
/* The same list stored through a MAP_SHARED mapping of the file (buffered pwrites to one file
 * serialise on its inode lock; page faults of a mapping do not).  The file must already be long
 * enough for the furthest entry. */
int mecha_host_mmap_write_many(int fd, const void *const *src, const size_t *nbytes,
                               const int64_t *offsets, int64_t n, int threads);

/* n host-to-device copies on one stream, in order: dst_dev[i] <- src_host[i] (nbytes[i] bytes).  For inputs that already sit
 * in pinned host memory (the reference has no counterpart: its cast node copies from pageable memory, logistics.py:53). */
int mecha_host_h2d_many(void *const *dst_dev, const void *const *src_host, const size_t *nbytes, int64_t n, void *stream);

/* The JSON header of a .safetensors file as flat tables (replaces the json.loads of reference sd_mecha/streaming.py:141-149 on the
 * read side: 10-17 ms of interpreter time per LoRA file, 17 files per LoRA job).  json / len: the header text after the 8-byte
 * length.  Tensors come out ordered by data offset (as the reference orders them): the names NUL-terminated one after the other in
 * `names` (room for `len` bytes, *names_bytes = bytes used), dtype_code[i] = index into {F64, I64, F32, I32, F16, BF16, I16, I8,
 * U8, BOOL}, ndim[i] (<= 8), shape[8 i ..], offsets[2 i], [2 i + 1] = data_offsets.  metadata_span[0], [1] = offset and length of
 * the "__metadata__" value (0, 0 when absent).  Returns the number of tensors, or -1 for a header it does not read (another dtype
 * tag, more than 8 dimensions, more than max_entries tensors, a name with a JSON escape, malformed text): fall back to a JSON
 * reader then.  No GPU needed. */
int64_t mecha_host_parse_safetensors_header(const char *json, int64_t len, int64_t max_entries, char *names, int64_t *names_bytes,
                                            int32_t *dtype_code, int32_t *ndim, int64_t *shape, int64_t *offsets, int64_t *metadata_span);

#ifdef __cplusplus
}
#endif
#endif /* MECHA_HIP_H */
