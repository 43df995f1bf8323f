/*
 * mvg_decoder.h -- C ABI of libmvgformer_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary of the MVGFormer decoder hot path.  Plain pointers and sizes,
 * no torch / ATen types.  All pointers are DEVICE pointers unless marked "host".
 * Every entry point enqueues work on `stream` (a hipStream_t passed as void*;
 * NULL = the default stream), never synchronises, never allocates, never
 * mutates its inputs, and returns 0 (hipSuccess) or a hipError_t / MVG_E_* code.
 * No exception crosses this boundary.
 *
 * Reference interfaces replaced (paths relative to the MVGFormer reference tree):
 *   - pybind module `Deformable` : lib/models/ops/src/vision.cpp:24-27
 *       deform_forward  : lib/models/ops/src/deform.h:32-50  -> deform_cuda_forward,
 *                         lib/models/ops/src/cuda/deform_cuda.cu:31-91,
 *                         kernel lib/models/ops/src/cuda/deform_im2col_cuda.cuh:248-309
 *       deform_backward : lib/models/ops/src/deform.h:53-72  -> deform_cuda_backward,
 *                         lib/models/ops/src/cuda/deform_cuda.cu:94-164, kernels cuh:312-930
 *   - the ATen op chains of ProjAttn.forward (lib/models/ops/modules/projattn.py:115-204),
 *     DQDecoderLayer.forward (lib/models/dq_decoder.py:850-1045) and the DLT
 *     triangulation (lib/mvn/utils/multiview.py:170-269) -> the mvg_* stage entry points.
 *
 * Layout conventions (see DESIGN.md "Data layout in HBM"):
 *   image index n = v*B + b (view-major, like dq_decoder.py:560);  Lq = NQ*J joint tokens;
 *   value / feat : (N_img, S, C) channels-last, level l at rows [start_l, start_l+H_l*W_l);
 *   dtype code   : MVG_F32 = 0, MVG_BF16 = 1 (bf16 storage, fp32 accumulation).
 */
#ifndef MVG_DECODER_H
#define MVG_DECODER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVG_F32 0
#define MVG_BF16 1

#define MVG_E_BADARG 10001   /* shape / alignment / dtype not supported            */
#define MVG_E_NOGPU 10002    /* no gfx950 device visible                            */

/* floats per packed camera record, one record per image n = v*B + b            */
#define MVG_CAM_STRIDE 48
/* record layout (float index):
 *   0..8  R (row-major 3x3)       9..11 T (camera centre, mm)   12 fx 13 fy 14 cx 15 cy
 *   16..18 k1 k2 k3               19,20 p1 p2
 *   21..26 crop affine 2x3 (orig px -> network px; lib/utils/transforms.py:72-112)
 *   27..32 inverse crop affine 2x3 (meta['inv_affine_trans'][:2])
 *   33,34 wh = 2*center (orig image size)   35 clamp_max (max of wh over the batch)
 *   36,37 network image size (w,h)          38..47 reserved (0)
 */

/* Library / device identification.  Returns 0 and fills arch (e.g. "gfx950") if a
 * usable device is present; MVG_E_NOGPU otherwise.  Host-only. */
int mvg_device_info(char* arch_out, int arch_len, int* cu_count);
const char* mvg_version(void);
/* Kernel-variant knobs for A/B measurements.  PROCESS-GLOBAL and not thread-safe: a value set here applies to every later call
 * of every entry point from every thread, whatever the stream or device; set knobs before the first launch (or from one thread
 * while nothing is in flight), never as per-request state.  Host-only.  Env MVG_TUNE="key=value,..." sets them at load time.
 *   "gsamp_threads" = 128 | 256 | 512 | 1024, "gsamp_map" = 0 | n : workgroup size and XCD block mapping of the G-sampling kernel
 *       (chunks of n slot blocks per XCD, 0 = one head per XCD);  "gsamp_pipe" = 0 | 1 | 2 : its gather loop double-buffered from 32 768 pairs per launch on / always / never;
 *       "gsamp_lds_pad" / "chain_a_lds_pad" = bytes (probes): unused dynamic LDS per sampler / chain-A workgroup, caps the workgroups per
 *       CU (occupancy sweeps: tools/r06_sampler_probe.py, tools/r06_fusion_emul.py);
 *   "gfused_chunk" = 0 | n : the same mapping for the fp32 G-sampling kernel;  "fwd_map" = 0 | 1 | 2 : decomposition of mvg_msda_forward;
 *   "auto_small" = 1 | 0 : launches with few rows (a rank's shard of a query-sharded run) use smaller workgroups / tiles in the
 *       sampler, chain A (64-row tiles up to 320 tiles of 128 rows) and chain B (32-row tiles up to 128 tiles of 64 rows);
 *   "chain_rm" = 64 | 128 | 256 : rows per tile of mvg_chain_attn_pose;  "f32h_rows" = 0 | 32 | 64 : rows per tile of the fp32 chains;
 *   "wreg_grid" = n : persistent workgroups of mvg_pyramid_group_ws;  "bin_multi" = 1 | 0 : multi-workgroup binning for large Lq;
 *   "linear_tiles" = 0 | 1 | 2 : tile shapes of mvg_linear*;
 *   "f32_split" = 1 | 0 : fp32 GEMMs of mvg_linear* as six bf16 MFMA products on operands split into three bf16 parts (default) or
 *       as v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain); with 0 the Python layer also stands the fused fp32 kernels down.
 * Every knob but f32_split, auto_small (chain B's 32-row tiles sum the FFN in another order) and chain_rm = 256 selects
 * bit-identical results (tests/test_hip_parity.py: test_every_kernel_variant_behind_a_tuning_knob). */
int mvg_set_tuning(const char* key, int value);

/* ---- Deformable.deform_forward / deform_backward (deform.h:32-72) -------------------
 * value (N,S,M,D); spatial_shapes (L,2) int64 (H,W); level_start_index (L,) int64;
 * sampling_loc (N,Lq,M,L,P,2) (x,y) in [0,1]; attn_weight (N,Lq,M,L,P); out (N,Lq,M*D).
 * f32: every tensor float32 (AT_DISPATCH_FLOATING_TYPES' float case, deform_cuda.cu:75).
 * bf16: value/out bf16, sampling_loc/attn_weight float32, fp32 accumulation. */
int mvg_msda_forward_f32(const float* value, const int64_t* spatial_shapes,
                         const int64_t* level_start_index, const float* sampling_loc,
                         const float* attn_weight, float* out,
                         int N, int S, int M, int D, int L, int Lq, int P, void* stream);
int mvg_msda_forward_bf16(const void* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const float* sampling_loc,
                          const float* attn_weight, void* out,
                          int N, int S, int M, int D, int L, int Lq, int P, void* stream);
/* f64: every tensor float64 -- the `double` case of AT_DISPATCH_FLOATING_TYPES (deform_cuda.cu:75,145), for gradcheck-style
 * calls; plain one-thread-per-output-channel kernels, no fast path.  mvg_msda_backward_f64: grad_value must be zero-filled
 * by the caller like the f32 form; grad_sampling_loc / grad_attn_weight are zeroed and accumulated by the entry point. */
int mvg_msda_forward_f64(const double* value, const int64_t* spatial_shapes,
                         const int64_t* level_start_index, const double* sampling_loc,
                         const double* attn_weight, double* out,
                         int N, int S, int M, int D, int L, int Lq, int P, void* stream);
int mvg_msda_backward_f64(const double* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const double* sampling_loc,
                          const double* attn_weight, const double* grad_output,
                          double* grad_value, double* grad_sampling_loc, double* grad_attn_weight,
                          int N, int S, int M, int D, int L, int Lq, int P, void* stream);
/* grad_value (N,S,M,D) must be zero-filled by the caller (at::zeros_like, deform_cuda.cu:132);
 * grad_sampling_loc / grad_attn_weight are fully overwritten.  Deterministic for a fixed
 * launch geometry only up to the atomic-add order in grad_value (as the reference, cuh:135-162). */
int mvg_msda_backward_f32(const float* value, const int64_t* spatial_shapes,
                          const int64_t* level_start_index, const float* sampling_loc,
                          const float* attn_weight, const float* grad_output,
                          float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                          int N, int S, int M, int D, int L, int Lq, int P, void* stream);

/* Deterministic, atomics-free-in-fp32 form of mvg_msda_backward_f32 for D = 32 (csrc/msda_bwd.hip): the samples are binned
 * by destination tile, a workgroup per tile gathers its value patch into LDS and accumulates grad_value there as 64-bit
 * fixed-point integers (integer addition is associative: no dependence on the order of arrival), grad_sampling_loc /
 * grad_attn_weight are reduced over the 32 channels with a fixed shuffle tree.  Bit-reproducible run to run; every output is
 * fully overwritten (grad_value need not be zero-filled).  workspace: device scratch of
 * mvg_msda_backward_det_workspace(...) bytes (0 = shape not supported: D != 32, L*P > 256, Lq >= 2^24 -- use the
 * atomic form).  shapes_host / starts_host: HOST int64 arrays (the level table is a kernel argument here).
 * Non-finite entries of grad_output are not propagated into grad_value (the fixed-point scale is the largest finite |g|). */
size_t mvg_msda_backward_det_workspace(int N, int S, int M, int D, int L, int Lq, int P, const int64_t* shapes_host);
int mvg_msda_backward_det_f32(const float* value, const int64_t* shapes_host, const int64_t* starts_host,
                              const float* sampling_loc, const float* attn_weight, const float* grad_output,
                              float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                              int N, int S, int M, int D, int L, int Lq, int P,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ---- stage entry points of the decoder layer ------------------------------------------ */


/* all L levels in one launch: src_nchw_host = HOST array of L device pointers to the (N_img,C,H_l,W_l) maps */
int mvg_pack_pyramid(const float* const* src_nchw_host, void* feat, int dtype, int N_img, int C, const int64_t* shapes_host,
                     const int64_t* starts_host, int L, int S, void* stream);

/* A.1 + A.2 projection (dq_decoder.py:331-397,570-573; cameras.py:167-217): X (B,Lq,3) mm,
 * cams (V*B, MVG_CAM_STRIDE) -> r (V*B,Lq,2) normalised network-image coords,
 * ref_lvl (V*B,Lq,L,2) = r * (W_l,H_l)/(W_l-1,H_l-1), inside (V*B,Lq) uint8.
 * shapes_host: L x (H,W) int64 on the host. */
int mvg_project(const float* X, const float* cams, const int64_t* shapes_host, int L,
                float* r, float* ref_lvl, uint8_t* inside, int V, int B, int Lq, void* stream);

/* A.3 step 1 (projattn.py:134,148-153,180): bilinear ref-point features of every level
 * + (tgt+query_pos) -> ain (V*B*Lq*L, C) in `dtype`.  feat (V*B,S,C) `dtype`;
 * ref_lvl (V*B,Lq,L,2); x (B,Lq,C) f32; shapes/starts host int64 arrays (L). */
int mvg_gather_ref(const void* feat, int dtype, const float* ref_lvl, const float* x,
                   const int64_t* shapes_host, const int64_t* starts_host, void* ain,
                   int V, int B, int Lq, int L, int S, int C, void* stream);

/* Dense projection out[M,N] = act(A[M,K] @ W[N,K]^T + bias[N]) (* rowmask[M]) on MFMA.
 * a_dtype / w_dtype / out_dtype: MVG_F32 or MVG_BF16 (weights must match the compute
 * dtype: bf16 weights -> bf16 MFMA, A converted on load; f32 weights -> fp32 products, formed either as six bf16 MFMAs on
 * operands split into three bf16 parts (default; fp32 accuracy, not bitwise an fmaf chain; Inf / > 3.39e38 inputs give NaN)
 * or as v_mfma_f32_32x32x2_f32 (mvg_set_tuning("f32_split", 0): bitwise an fp32 fmaf chain)).
 * relu: 0/1.  rowmask: NULL or uint8 (M).  lda/ldc in elements. */
int mvg_linear(const void* A, int a_dtype, int lda, const void* W, int w_dtype,
               const float* bias, void* out, int out_dtype, int ldc,
               const uint8_t* rowmask, int relu, int M, int N, int K, void* stream);

/* mvg_linear (fp32) over rows in a processing order, skipping tiles the consumer masks: tile row i works on row order[m0 + i] of
 * A / out; a tile none of whose rows has inside[row] != 0 writes `masked_row` (N floats, what the same kernel computes for such
 * a row) to all its rows without arithmetic.  The per-view output projection and pose MLP of the fp32 path (projattn.py:203,
 * dq_decoder.py:585-586, 673-690): pairs outside the image -- a third of them at cfg-2 -- cost a broadcast instead of three GEMMs. */
int mvg_linear_ordered(const float* A, int lda, const float* W, const float* bias, float* out, int ldc,
                       const uint8_t* rowmask, int relu, int M, int N, int K, const int32_t* order,
                       const uint8_t* inside, const float* masked_row, void* stream);



/* The same with the bias gradient and the sum over the slices: dW (N, K) = dY^T X and, when db is not NULL, db (N) = the column sums
 * of dY (what autograd's Linear backward returns for weight and bias, lib/models/dq_decoder.py:659-717 under run/train_3d.py).
 * `partial` (splits, N, K) and `partial_db` (splits, N) are workspaces; a second small launch adds them in slice order
 * (deterministic; dW equals mvg_linear_wgrad_f32's partials summed slice 0 first). */
int mvg_linear_wgrad_bias_f32(const float* dY, int ldy, const float* X, int ldx, float* partial, float* partial_db, float* dW, float* db,
                              int rows, int N, int K, int splits, void* stream);

/* mvg_linear with the activation formed as A + A2 on load (A2: fp32, same shape and leading dimension as A, or NULL):
 * the query term of the first layer, Linear(tgt + query_pos) (dq_decoder.py:580 `with_pos_embed` + projattn.py:180-181),
 * without a separate elementwise pass over the two (B*Lq, 256) tensors. */
int mvg_linear_sum(const void* A, const void* A2, int a_dtype, int lda, const void* W, int w_dtype, const float* bias,
                   void* out, int out_dtype, int ldc, const uint8_t* rowmask, int relu, int M, int N, int K, void* stream);

/* A.3 steps 3-6 fused (projattn.py:180-200): oa (V*B*Lq*L, 192) f32 = Linear outputs
 * [sampling_offsets(128) | attention_weights(64)] per level row; reinterpretation,
 * softmax(L*P), locations (ref_lvl (V*B*Lq,L,2) + offset/(W,H)) and multi-scale sampling of
 * value (V*B,S,256).  M=8, D=32, P=8.
 * samp (V*B*Lq, 256) in `dtype`. */
int mvg_msda_fused(const void* value, int dtype, const float* oa, const float* ref_lvl,
                   const int64_t* shapes_host, const int64_t* starts_host, void* samp,
                   int N_img, int Lq, int L, int S, void* stream);

/* G-sampling in fp32 (the reference's arithmetic; replaces mvg_gather_ref + the (V*Lq*L x 256 x 192) mvg_linear +
 * mvg_msda_fused of the fp32 path): value (V*B,S,256) f32 pixel-major; G (V*B*S, 192) f32 = feat @ [Woff; Wattn]^T with the
 * columns in mvgformer_amd.ops.gsamp_column_order (no bias); xw (B*Lq, 192) f32 = (tgt+query_pos) @ W^T + b, same column
 * order; samp (V*B*Lq, 256) f32.  Bilinear sampling commutes with the Linear, so the results equal the gather-then-Linear
 * form to fp32 rounding.  pair_mask / order (or NULL) as in mvg_msda_gsamp: masked pairs are zero-filled without being
 * sampled (the consumer multiplies exactly these rows by the in-image mask, dq_decoder.py:585-586), slot i of the launch
 * computes pair order[i].  M=8, D=32, P=8, L<=4. */
int mvg_msda_gfused_f32(const float* value, const float* G, const float* xw, const float* ref_lvl,
                        const int64_t* shapes_host, const int64_t* starts_host, float* samp,
                        const uint8_t* pair_mask, const int32_t* order,
                        int N_img, int Lq, int L, int S, int B, void* stream);

/* Several of the two products above over the SAME packed pyramid in ONE launch (round 5): job j is a value projection into head
 * planes (planes[j] != 0: N[j] = 256, bias[j] required, out[j] = vh) or a G product (planes[j] == 0: N[j] = 192, bias ignored,
 * out[j] = G row-major).  The host arrays hold njobs (1..8) entries.  An XCD's workgroups are divided among the jobs and sweep
 * its row tiles together, so the pyramid is fetched through the fabric once per launch instead of once per product; every
 * output is bit-identical to the single-product entry points (same kernel body).  slots_per_xcd: persistent workgroups per
 * XCD (0 = the default: all 64 resident ones; fewer leave CU room for kernels that run next to the launch).
 * projattn.py:169,180-181 for all layers. */
int mvg_pyramid_group_ws(const void* feat, int n_img, int S, int njobs, const void* const* Wf, const float* const* bias,
                         void* const* out, const int* N, const int* planes, int slots_per_xcd, void* stream);
int mvg_msda_gsamp(const void* vh, const void* G, const float* xw, const float* ref_lvl,
                   const int64_t* shapes_host, const int64_t* starts_host, void* samp,
                   const uint8_t* pair_mask, const int32_t* order,
                   int N_img, int Lq, int L, int S, int B, void* stream);

/* Processing order for mvg_msda_gsamp: per image, the Lq (image, query) pairs counting-sorted by the Morton code of
 * the level-0 cell block of their reference point, pairs with inside == 0 last.  order (N_img*Lq) int32 holds
 * global pair indices (n*Lq + q), a permutation of all of them; it changes where a pair is computed, never its result.
 * inside may be NULL.
 * workspace: device scratch of at least mvg_bin_pairs_workspace(N_img, Lq) bytes (0 for small Lq), or NULL: with it,
 * images with many pairs are sorted by 8 workgroups each (two kernels) instead of one.
 * Layout: with the workspace (>= 8192 pairs per image) the in-image pairs of ALL images come first, image by image, then the
 * pairs with inside == 0, image by image -- consumers that skip the latter then find them in one run at the end of their launch
 * instead of one run per image; without it every image's Lq entries are [its in-image pairs | its other pairs]. */
size_t mvg_bin_pairs_workspace(int N_img, int Lq);
int mvg_bin_pairs(const float* ref_lvl, const uint8_t* inside, const int64_t* shapes_host, int L, int32_t* order,
                  int N_img, int Lq, void* workspace, size_t workspace_bytes, void* stream);

/* A.4 (dq_decoder.py:770): mean over views of attn (V,B*Lq,C) `dtype` -> (B*Lq,C) `dtype`. */
int mvg_mean_views(const void* attn, int dtype, void* out, int V, int rows, int C, void* stream);

/* y = LayerNorm(res + h) * gamma + beta, eps 1e-5 (norm2 / norm3, dq_decoder.py:776-778,
 * mvp_decoder.py:94-98).  res f32 (rows,C); h `h_dtype`; y f32. */
int mvg_add_layernorm(const float* res, const void* h, int h_dtype, const float* gamma,
                      const float* beta, float* y, int rows, int C, void* stream);

/* A.5 (dq_decoder.py:889-908,596-623): class head + validity.  tgt (B,NQ*J,C) f32,
 * Wc (2,C), bc (2) -> prob (B,NQ,2); valid (B,NQ) uint8 = prob[...,1] > threshold
 * (or the caller-provided `forced_valid` mask when not NULL: training indices);
 * any_valid: 1 int, must be zeroed by the caller; set to 1 if any query is valid. */
int mvg_class_head(const float* tgt, const float* Wc, const float* bc, float threshold,
                   const uint8_t* forced_valid, float* prob, uint8_t* valid, int* any_valid,
                   int B, int NQ, int J, int C, void* stream);

/* last pose_embed layer (N=3): o (rows,3) f32 = h (rows,C) @ W3 (3,C)^T + b3. */
int mvg_rowdot3(const void* h, int h_dtype, const float* W3, const float* b3, float* o,
                int rows, int C, void* stream);

/* Fused bf16 chain per (image, query) row (dq_decoder.py:585-588,659-690):
 *   attn = inside * (samp @ Wp^T + bp)                       (stored: input of the view mean)
 *   o    = W2 relu(W1 relu(W0 attn + b0) + b1) + b2          (dx, dy, confidence logit)
 * samp/attn (rows,256) bf16; Wp,W0,W1 (256,256) bf16; W2 (3,256) f32; biases f32; o (rows,3) f32.
 * Replaces mvg_linear x3 + mvg_rowdot3 of the unfused path; activations stay in LDS.
 * order (rows) i32 or NULL: tile row i of the launch works on row order[i] (mvg_bin_pairs: masked rows last).
 * o_masked (3) f32 or NULL: o of a row with inside == 0 (the MLP of a zero row; obtain it by running this entry
 * point on one masked row).  When given, 64-row tiles without a single in-image row only write attn = 0 and
 * o = o_masked instead of running the chain. */
int mvg_chain_attn_pose(const void* samp, const uint8_t* inside, const void* Wp, const float* bp,
                        const void* W0, const float* b0, const void* W1, const float* b1,
                        const float* W2, const float* b2, void* attn, float* o,
                        const int32_t* order, const float* o_masked, int rows, void* stream);

/* ---- fp32 path as fused kernels (fp32 storage, fp32-accurate products on the fp16 matrix pipe; csrc/f32s.hip) ----------------
 * The reference's arithmetic is fp32 (lib/models/ops/src/cuda/deform_cuda.cu:75 dispatches float / double only; the Linears of
 * lib/models/dq_decoder.py:763-848,659-717 and lib/models/ops/modules/projattn.py:169,180-181,203 are fp32 nn.Linear).  These
 * entry points keep fp32 tensors at every boundary and form every product as three fp16 MFMAs (l*h + h*l + h*h, fp32 accumulate)
 * on operands scaled by a power of two and split into two fp16 parts (x 2^s = h + l: 22 bits of each operand; error against the
 * fp64 product not above an fp32 fmaf chain's, tests/test_hip_parity.py).  Weight operands W*_planes: the weight times 2^scale as
 * two fp16 planes (h, l), each in the MFMA-fragment order of mvg_chain_attn_pose (mvgformer_amd.ops.swizzle_weight), plane p at
 * element offset p * N * K, N padded to a multiple of 256 with zero rows (mvgformer_amd.ops.split_swizzle_weight_h2, which also
 * returns the scale); activation rows are scaled by the kernels (a power of two per row from the row's maximum).  A non-finite
 * input value makes its output row NaN.  The range-safe alternative is the unfused path on mvg_linear's three-part split form. */

/* chain B in fp32 (lib/models/dq_decoder.py:770-778, lib/models/mvp_decoder.py:94-98, dq_decoder.py:889-908): attn (V, B*NQ*J, 256)
 * f32; Wu (256,256), W1 (1024,256), W2 (256,1024), W_next (n_next padded to 256, 256) as planes with their scales; everything else
 * as in mvg_chain_update_ffn_class (n_next: multiple of 32).  32- or 64-row tiles by the row count (both sum a row identically).
 * The FFN's hidden activations carry a scale per (row, 256-column chunk); the residual t1 stays in fp32 registers.  Replaces
 * mvg_mean_views + mvg_linear x 3 (+ the next layer's query-term GEMM) + mvg_add_layernorm x 2 + mvg_class_head of the unfused path. */
int mvg_chain_update_ffn_class_f32h(const float* attn, int V, const float* tgt, const void* Wu, int wu_scale, const float* bu,
                                    const float* g2, const float* be2, const void* W1, int w1_scale, const float* b1, const void* W2,
                                    int w2_scale, const float* b2, const float* g3, const float* be3, const float* Wc, const float* bc,
                                    float threshold, const uint8_t* forced_valid, float* tgt_out, float* prob, uint8_t* valid,
                                    int* any_valid, const float* query_pos, const void* W_next, int wn_scale, const float* b_next,
                                    float* xw_next, int n_next, int B, int NQ, int J, int has_ffn, void* stream);

/* chain A in fp32 (lib/models/dq_decoder.py:585-588,659-690): samp / attn (rows, 256) f32, o (rows, 3) f32; Wp, W0, W1 planes of the
 * (256, 256) weights with their scales; W2 (3, 256) f32; order / o_masked as in mvg_chain_attn_pose (o_masked: this entry point run
 * on one masked row).  Activation rows are scaled between the stages (row maximum over the 8 wavefronts through LDS); attn is
 * stored as the exact fp32 result of its stage.  Replaces mvg_linear_ordered x 3 + mvg_rowdot3 of the unfused fp32 path. */
int mvg_chain_attn_pose_f32h(const float* samp, const uint8_t* inside, const void* Wp, int wp_scale, const float* bp, const void* W0,
                             int w0_scale, const float* b0, const void* W1, int w1_scale, const float* b1, const float* W2,
                             const float* b2, float* attn, float* o, const int32_t* order, const float* o_masked, int rows,
                             void* stream);

/* value = feat @ Wv^T + bv  (rows, 256)  and  G = feat @ Wg^T  (rows, n_g)  in one pass over the packed fp32 pyramid feat
 * (rows, 256): the value projection of projattn.py:169 and the pyramid side of the offsets / logits Linear (projattn.py:180-181
 * through Linear(bilinear(feat) + x) = bilinear(feat W^T) + (x W^T + b), see mvg_msda_gfused_f32).  n_g: multiple of 32, <= 256. */
int mvg_pyramid_f32h(const float* feat, const void* Wv_planes, int wv_scale, const float* bv, const void* Wg_planes, int wg_scale,
                     float* value, float* G, int64_t rows, int n_g, void* stream);

/* Fused bf16 chain per joint token (dq_decoder.py:770-778, mvp_decoder.py:94-98, dq_decoder.py:889-908):
 *   t1 = LN2(tgt + Wu mean_v(attn_v) + bu);  tgt' = LN3(t1 + W2 relu(W1 t1 + b1) + b2) (has_ffn) else t1;
 *   prob = mean_j sigmoid(Wc tgt' + bc);  valid = prob[...,1] > threshold (or forced_valid).
 * attn (V, B*NQ*J, 256) bf16; tgt/tgt_out (B*NQ*J, 256) f32; Wu (256,256), W1 (1024,256), W2 (256,1024) bf16
 * in the fragment order of csrc/chain.hip (mvgformer_amd.ops.swizzle_weight); LN params / biases f32;
 * any_valid must be zeroed by the caller.  Replaces mean_views + 3 linears + 2 add_layernorm + class_head.
 * Optional tail (W_next != NULL): xw_next (B*NQ*J, n_next) f32 = (tgt' + query_pos) @ W_next^T + b_next, the query
 * term of the NEXT layer's offsets/logits Linear (the xw operand of mvg_msda_gsamp), computed while the rows are in
 * LDS.  W_next: (256,256) bf16 fragment order, rows >= n_next zero; b_next: 256 f32 (zero padded); query_pos
 * (B*NQ*J, 256) f32 or NULL; n_next <= 256, multiple of 4.
 * attn_inside (V, B*NQ*J) u8 or NULL: the in-image flags mvg_chain_attn_pose was given.  Rows of attn with flag 0 are zero
 * by construction; with the flags the view mean does not read them (same sums, a third fewer bytes at cfg-2). */
int mvg_chain_update_ffn_class(const void* attn, int V, const float* tgt, const void* Wu, const float* bu,
                               const float* g2, const float* be2, const void* W1, const float* b1,
                               const void* W2, const float* b2, const float* g3, const float* be3,
                               const float* Wc, const float* bc, float threshold,
                               const uint8_t* forced_valid, float* tgt_out, float* prob, uint8_t* valid,
                               int* any_valid, const float* query_pos, const void* W_next, const float* b_next,
                               float* xw_next, int n_next, int B, int NQ, int J, int has_ffn, const uint8_t* attn_inside, void* stream);

/* A.6-A.8 (dq_decoder.py:659-717,399-461,119-246,1013-1029; multiview.py:170-269):
 * 2D refinement, view-softmax confidence, un-crop, 5-iteration undistortion, DLT rows,
 * smallest right singular vector, masked scatter.
 * r (V*B,Lq,2); o (V*B,Lq,3) = (dx,dy,conf logit); cams (V*B,STRIDE); valid (B,NQ);
 * any_valid (1 int: if 0, query (0,0) is forced valid, dq_decoder.py:620-623).
 * Outputs: new_ref (B,Lq,3) mm; ref2d, proj2d (B,V,Lq,2) network-image px; zeros for
 * non-valid queries. */
int mvg_triangulate(const float* r, const float* o, const float* cams, const uint8_t* valid,
                    const int* any_valid, float* new_ref, float* ref2d, float* proj2d,
                    int V, int B, int NQ, int J, void* stream);

/* mvg_triangulate + the NEXT layer's mvg_project in one launch: the lanes that solved a (query, joint) problem project
 * its new 3D point (zeros for queries that did not pass, exactly what the next layer receives as reference_points) into
 * all V views -> r_next (V*B,Lq,2), ref_lvl_next (V*B,Lq,L,2), inside_next (V*B,Lq), bit-identical to mvg_project(new_ref). */
int mvg_triangulate_project(const float* r, const float* o, const float* cams, const uint8_t* valid,
                            const int* any_valid, float* new_ref, float* ref2d, float* proj2d,
                            int V, int B, int NQ, int J, const int64_t* shapes_host, int L,
                            float* r_next, float* ref_lvl_next, uint8_t* inside_next, void* stream);

/* Training path: the refined 2D points ref2d (B, V, Lq, 2; network-image px) -> un-cropped, undistorted
 * original-image px ud (B, V, Lq, 2) (lib/models/dq_decoder.py:414-420, 119-204: inverse crop affine, K^-1, 5 fixed-point
 * iterations, K) and jac (B, V, Lq, 4) = the row-major 2 x 2 Jacobian d(ud) / d(ref2d) of every point (forward mode through the
 * same iterations): the backward of this step is grad_ref2d = jac^T grad_ud.  cams: the packed records, image n = v * B + b. */
int mvg_uncrop_undistort_jac(const float* ref2d, const float* cams, float* ud, float* jac, int V, int B, int Lq, void* stream);

/* The differentiable triangulation of the training path (lib/models/multiview.py:170-228 as called by
 * lib/models/dq_decoder.py:929-967 under autograd), dense over the (B, NQ * J) tokens: X (B, NQ*J, 3) = v0[:3] / v0[3], v0 the
 * eigenvector of the smallest eigenvalue of A^T A, A's rows conf[b,v,t] * (P[b,v,2] * ud[b,v,t,c] - P[b,v,c]) (rows, Gram
 * matrix and Jacobi eigen-solve in fp64).  ud (B, V, Lq, 2), conf (B, V, Lq), Pm (B, V, 3, 4) fp32; valid (B, NQ) uint8: tokens of
 * queries with valid == 0 are skipped and get X = 0 (the reference triangulates the matched queries only).  V <= 32. */
int mvg_dlt_forward(const float* ud, const float* conf, const float* Pm, const uint8_t* valid, float* X, int V, int B, int NQ, int J,
                    void* stream);

/* Its backward: g_ud (B, V, Lq, 2), g_conf (B, V, Lq) from gX (B, Lq, 3); the decomposition is recomputed from the inputs
 * (dL/dG = sym(m v0^T), m = sum_{i != 0} v_i (v_i^T g) / (l0 - l_i); pairs with |l0 - l_i| <= 1e-14 max|l| contribute nothing);
 * zero gradients for skipped tokens. */
int mvg_dlt_backward(const float* ud, const float* conf, const float* Pm, const uint8_t* valid, const float* gX, float* g_ud,
                     float* g_conf, int V, int B, int NQ, int J, void* stream);

/* Batched eigen-decomposition of n symmetric 4x4 fp64 matrices G (n,4,4): evals (n,4) in no particular order, evecs
 * (n,4,4) with the eigenvectors as columns (G v_k = evals_k v_k, v_k = evecs[:, :, k]).  fp64 cyclic Jacobi, one lane
 * per matrix.  Used by the differentiable DLT of the training path (multiview.py:170-228 under autograd): smallest
 * eigenvector of A^T A in the forward, all pairs in the backward. */
int mvg_sym4_eigh(const double* G, double* evals, double* evecs, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVG_DECODER_H */
