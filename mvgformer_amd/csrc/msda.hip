// Multi-scale deformable sampling for gfx950 (MI355X): forward (plain + fused), backward.
//
// Semantics follow the reference op `Deformable.deform_forward`
// (lib/models/ops/src/cuda/deform_im2col_cuda.cuh:248-309, bilinear helper :44-94) but the
// decomposition is wave64-first:
//   * one 64-lane wavefront owns one (image, query) pair = 8 heads x 32 channels; lane
//     (m = lane>>3, sub = lane&7) accumulates channels [4*sub, 4*sub+4) of head m in
//     registers, so every corner read is one 16-byte (fp32) / 8-byte (bf16) vector load and
//     the 8 lanes of a head cover one contiguous 128-B / 64-B row segment of value[s, m, :];
//   * no atomics, no shared memory round trip in the forward: the (L*P) attention logits and
//     offsets of a head are loaded as 16-byte vectors (all 8 lanes of the head hit the same
//     addresses -> one broadcast transaction) and the softmax is evaluated in registers;
//   * workgroup -> (image, query) mapping is XCD-aware: the 8 XCDs of the chip each walk a
//     contiguous range of query blocks, so queries that project next to each other (the
//     15 joints of a person, neighbouring persons of the query grid) share one XCD's L2.
#include "common.h"

// ------------------------------------------------------------------------------------------
// bilinear sample of 4 consecutive channels with the reference's zero padding
// (cuh:44-94): corners outside [0,H-1]x[0,W-1] contribute 0; the sample is skipped unless
// h_im > -1 && w_im > -1 && h_im < H && w_im < W (cuh:298).  Branch-free: invalid corners
// get weight 0 and a clamped (legal) address.
template <typename T>
__device__ __forceinline__ f32x4 bilinear4(const T* __restrict__ lvl_base, int H, int W, long row_stride,
                                           float h_im, float w_im, float scale) {
  const float hl_f = floorf(h_im), wl_f = floorf(w_im);
  const int h_low = (int)hl_f, w_low = (int)wl_f;
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - hl_f, lw = w_im - wl_f;
  const float hh = 1.f - lh, hw = 1.f - lw;
  const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
  const float s = inside ? scale : 0.f;
  const bool hl_ok = h_low >= 0, hh_ok = h_high <= H - 1, wl_ok = w_low >= 0, wh_ok = w_high <= W - 1;
  const float w1 = (hl_ok && wl_ok) ? hh * hw : 0.f;
  const float w2 = (hl_ok && wh_ok) ? hh * lw : 0.f;
  const float w3 = (hh_ok && wl_ok) ? lh * hw : 0.f;
  const float w4 = (hh_ok && wh_ok) ? lh * lw : 0.f;
  const int hl_c = min(max(h_low, 0), H - 1), hh_c = min(max(h_high, 0), H - 1);
  const int wl_c = min(max(w_low, 0), W - 1), wh_c = min(max(w_high, 0), W - 1);
  const f32x4 v1 = Vec4<T>::load(lvl_base + ((long)hl_c * W + wl_c) * row_stride);
  const f32x4 v2 = Vec4<T>::load(lvl_base + ((long)hl_c * W + wh_c) * row_stride);
  const f32x4 v3 = Vec4<T>::load(lvl_base + ((long)hh_c * W + wl_c) * row_stride);
  const f32x4 v4 = Vec4<T>::load(lvl_base + ((long)hh_c * W + wh_c) * row_stride);
  f32x4 val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
  return val * s;
}

template <typename T>
__device__ __forceinline__ float bilinear1(const T* __restrict__ lvl_base, int H, int W, long row_stride,
                                           float h_im, float w_im, float scale) {
  const float hl_f = floorf(h_im), wl_f = floorf(w_im);
  const int h_low = (int)hl_f, w_low = (int)wl_f;
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - hl_f, lw = w_im - wl_f;
  const float hh = 1.f - lh, hw = 1.f - lw;
  if (!((h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W))) return 0.f;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = load1<T>(lvl_base + ((long)h_low * W + w_low) * row_stride);
  if (h_low >= 0 && w_high <= W - 1) v2 = load1<T>(lvl_base + ((long)h_low * W + w_high) * row_stride);
  if (h_high <= H - 1 && w_low >= 0) v3 = load1<T>(lvl_base + ((long)h_high * W + w_low) * row_stride);
  if (h_high <= H - 1 && w_high <= W - 1) v4 = load1<T>(lvl_base + ((long)h_high * W + w_high) * row_stride);
  return (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * scale;
}

// ------------------------------------------------------------------------------------------
// Plain forward = the drop-in for Deformable.deform_forward: any M, D, L, P.
// VEC = 4: D % 4 == 0, D/4 adjacent lanes own one head (D=32 -> 8 lanes, one wave = one query).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void msda_fwd_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const float* __restrict__ loc, const float* __restrict__ wgt, T* __restrict__ out,
    int N, int S, int M, int D, int L, int Lq, int P) {
  const int dv_per = D / VEC;
  const long total = (long)N * Lq * M * dv_per;
  const long row_stride = (long)M * D;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int dv = (int)(idx % dv_per);
    long t = idx / dv_per;
    const int m = (int)(t % M);
    t /= M;
    const int q = (int)(t % Lq);
    const int n = (int)(t / Lq);
    const long qm = ((long)n * Lq + q) * M + m;
    const float* locp = loc + qm * L * P * 2;
    const float* wp = wgt + qm * L * P;
    const T* vbase = value + (long)n * S * row_stride + (long)m * D + dv * VEC;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    float acc1 = 0.f;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T* lvl = vbase + (long)starts[l] * row_stride;
      for (int p = 0; p < P; ++p) {
        const float lx = locp[(l * P + p) * 2], ly = locp[(l * P + p) * 2 + 1];
        const float aw = wp[l * P + p];
        const float h_im = ly * (float)H - 0.5f;   // cuh:295
        const float w_im = lx * (float)W - 0.5f;   // cuh:296
        if (VEC == 4) acc4 += bilinear4<T>(lvl, H, W, row_stride, h_im, w_im, aw);
        else acc1 += bilinear1<T>(lvl, H, W, row_stride, h_im, w_im, aw);
      }
    }
    T* op = out + qm * D + dv * VEC;
    if (VEC == 4) Vec4<T>::store(op, acc4);
    else store1<T>(op, acc1);
  }
}

template <typename T>
static int launch_msda_fwd(const T* value, const int64_t* shapes, const int64_t* starts, const float* loc,
                           const float* wgt, T* out, int N, int S, int M, int D, int L, int Lq, int P,
                           hipStream_t st) {
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return MVG_E_BADARG;
  if (Lq == 0) return 0;
  const bool vec = (D % 4 == 0) && ((reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
  const long total = (long)N * Lq * M * (vec ? D / 4 : D);
  const int block = 256;
  long grid = (total + block - 1) / block;
  if (grid > (1L << 22)) grid = 1L << 22;
  if (vec)
    hipLaunchKernelGGL((msda_fwd_kernel<T, 4>), dim3((unsigned)grid), dim3(block), 0, st, value, shapes, starts, loc,
                       wgt, out, N, S, M, D, L, Lq, P);
  else
    hipLaunchKernelGGL((msda_fwd_kernel<T, 1>), dim3((unsigned)grid), dim3(block), 0, st, value, shapes, starts, loc,
                       wgt, out, N, S, M, D, L, Lq, P);
  MVG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Fused forward for the decoder (M=8, D=32, P=8, C=256): reinterpretation of the Linear
// outputs + softmax + locations + sampling (projattn.py:180-200) in one pass.
//   oa  : (pairs*L, 192) fp32, cols [0,128) = sampling_offsets, [128,192) = attention_weights
//   ref : (pairs, L, 2) per-level reference point r * (W,H)/(W-1,H-1) (dq_decoder.py:570-573)
// The module is built with n_levels=1 but applied to L levels, and the outputs are VIEWED,
// not transposed: flat offset index f_O = ((m*L+l)*P+p)*2+xy lives in level row
// f_O/128, column f_O%128; flat logit index f_A = m*L*P + l*P + p in row f_A/64, column f_A%64.
template <typename T, int L>
__global__ __launch_bounds__(256) void msda_fused_kernel(const T* __restrict__ value, const float* __restrict__ oa,
                                                         const float* __restrict__ r, LevelTable lv,
                                                         T* __restrict__ samp, int n_pairs, int Lq, int S) {
  constexpr int D = 32, P = 8, C = 256, LP = L * P;   // M = 8 heads = 64 lanes / 8
  // XCD-aware block remap (bijective): XCD x = blockIdx % 8 walks a contiguous block range
  const int nb = gridDim.x;
  const int q8 = nb >> 3, r8 = nb & 7;
  const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
  const int lblock = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pair = lblock * 4 + wave;
  if (pair >= n_pairs) return;
  const int n = pair / Lq;
  const int m = lane >> 3, sub = lane & 7;

  // ---- logits -> softmax weights (in registers, redundantly on the 8 lanes of a head)
  float aw[LP];
  const float* oa_q = oa + (long)pair * L * 192;
#pragma unroll
  for (int i = 0; i < LP / 4; ++i) {
    const int fa = m * LP + 4 * i;
    const f32x4 v = *reinterpret_cast<const f32x4*>(oa_q + (fa >> 6) * 192 + 128 + (fa & 63));
    aw[4 * i] = v[0]; aw[4 * i + 1] = v[1]; aw[4 * i + 2] = v[2]; aw[4 * i + 3] = v[3];
  }
  float mx = aw[0];
#pragma unroll
  for (int i = 1; i < LP; ++i) mx = fmaxf(mx, aw[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LP; ++i) {
    aw[i] = __expf(aw[i] - mx);
    sum += aw[i];
  }
  const float inv_sum = 1.f / sum;

  const T* vbase = value + (long)n * S * C + m * D + sub * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const float Wf = (float)W, Hf = (float)H;
    const float refx = r[((long)pair * L + l) * 2], refy = r[((long)pair * L + l) * 2 + 1];
    const float invW = 1.f / Wf, invH = 1.f / Hf;
    const T* lvl = vbase + (long)lv.start[l] * C;
#pragma unroll
    for (int pp = 0; pp < P / 2; ++pp) {
      const int fo = ((m * L + l) * P + 2 * pp) * 2;  // 2 points = 4 floats, 16-B aligned
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(oa_q + (fo >> 7) * 192 + (fo & 127));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float lx = refx + o4[2 * h] * invW;       // projattn.py:186-191
        const float ly = refy + o4[2 * h + 1] * invH;
        const float h_im = ly * Hf - 0.5f;
        const float w_im = lx * Wf - 0.5f;
        acc += bilinear4<T>(lvl, H, W, C, h_im, w_im, aw[l * P + 2 * pp + h] * inv_sum);
      }
    }
  }
  Vec4<T>::store(samp + (long)pair * C + m * D + sub * 4, acc);
}

template <typename T>
static int launch_msda_fused(const T* value, const float* oa, const float* r, const LevelTable& lv, T* samp,
                             int n_pairs, int Lq, int S, hipStream_t st) {
  if (n_pairs <= 0) return 0;
  const int grid = (n_pairs + 3) / 4;
  switch (lv.L) {
    case 1: hipLaunchKernelGGL((msda_fused_kernel<T, 1>), dim3(grid), dim3(256), 0, st, value, oa, r, lv, samp, n_pairs, Lq, S); break;
    case 2: hipLaunchKernelGGL((msda_fused_kernel<T, 2>), dim3(grid), dim3(256), 0, st, value, oa, r, lv, samp, n_pairs, Lq, S); break;
    case 3: hipLaunchKernelGGL((msda_fused_kernel<T, 3>), dim3(grid), dim3(256), 0, st, value, oa, r, lv, samp, n_pairs, Lq, S); break;
    case 4: hipLaunchKernelGGL((msda_fused_kernel<T, 4>), dim3(grid), dim3(256), 0, st, value, oa, r, lv, samp, n_pairs, Lq, S); break;
    default: return MVG_E_BADARG;
  }
  MVG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Backward (training drop-in; deform_cuda.cu:94-164, cuh:98-169,312-413).
// One thread per (n, q, m, channel); the D lanes of a head reduce grad_sampling_loc /
// grad_attn_weight with wavefront shuffles (no shared-memory serial reduce as in
// cuh:312-413), then one lane writes.  grad_value uses fp32 atomics like the reference.
template <int D>
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const float* __restrict__ loc, const float* __restrict__ wgt, const float* __restrict__ gout,
    float* __restrict__ gvalue, float* __restrict__ gloc, float* __restrict__ gwgt,
    int N, int S, int M, int L, int Lq, int P, int Druntime) {
  // D > 0: compile-time power-of-two head width <= 64 (shuffle reduce); D == 0: generic (atomics)
  const int Dd = D > 0 ? D : Druntime;
  const long total = (long)N * Lq * M * Dd;
  const long row_stride = (long)M * Dd;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = idx < total;
  const long cidx = active ? idx : total - 1;
  const int c = (int)(cidx % Dd);
  long t = cidx / Dd;
  const int m = (int)(t % M);
  t /= M;
  const int q = (int)(t % Lq);
  const int n = (int)(t / Lq);
  const long qm = ((long)n * Lq + q) * M + m;
  const float go = active ? gout[qm * Dd + c] : 0.f;
  const float* vb = value + (long)n * S * row_stride + (long)m * Dd + c;
  float* gvb = gvalue + (long)n * S * row_stride + (long)m * Dd + c;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long lbase = (long)starts[l] * row_stride;
    for (int p = 0; p < P; ++p) {
      const long sidx = qm * L * P + l * P + p;
      const float lx = loc[sidx * 2], ly = loc[sidx * 2 + 1];
      const float aw = wgt[sidx];
      const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
      float g_w = 0.f, g_h = 0.f, g_a = 0.f;
      if (active && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const float hl_f = floorf(h_im), wl_f = floorf(w_im);
        const int h_low = (int)hl_f, w_low = (int)wl_f, h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - hl_f, lw = w_im - wl_f, hh = 1.f - lh, hw = 1.f - lw;
        const float top = go * aw;
        float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        if (h_low >= 0 && w_low >= 0) {
          const long o = lbase + ((long)h_low * W + w_low) * row_stride;
          v1 = vb[o];
          atomicAdd(gvb + o, hh * hw * top);
        }
        if (h_low >= 0 && w_high <= W - 1) {
          const long o = lbase + ((long)h_low * W + w_high) * row_stride;
          v2 = vb[o];
          atomicAdd(gvb + o, hh * lw * top);
        }
        if (h_high <= H - 1 && w_low >= 0) {
          const long o = lbase + ((long)h_high * W + w_low) * row_stride;
          v3 = vb[o];
          atomicAdd(gvb + o, lh * hw * top);
        }
        if (h_high <= H - 1 && w_high <= W - 1) {
          const long o = lbase + ((long)h_high * W + w_high) * row_stride;
          v4 = vb[o];
          atomicAdd(gvb + o, lh * lw * top);
        }
        // d(val)/d(w_im), d(val)/d(h_im)  (cuh:128-160), scaled by W / H (cuh:166-167)
        g_w = (hh * (v2 - v1) + lh * (v4 - v3)) * top * (float)W;
        g_h = (hw * (v3 - v1) + lw * (v4 - v2)) * top * (float)H;
        g_a = go * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
      }
      if (D > 0) {
#pragma unroll
        for (int off = D / 2; off > 0; off >>= 1) {
          g_w += __shfl_xor(g_w, off, D);
          g_h += __shfl_xor(g_h, off, D);
          g_a += __shfl_xor(g_a, off, D);
        }
        if (active && c == 0) {
          gloc[sidx * 2] = g_w;
          gloc[sidx * 2 + 1] = g_h;
          gwgt[sidx] = g_a;
        }
      } else if (active) {
        atomicAdd(gloc + sidx * 2, g_w);
        atomicAdd(gloc + sidx * 2 + 1, g_h);
        atomicAdd(gwgt + sidx, g_a);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
extern "C" {

int mvg_msda_forward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const float* sampling_loc, const float* attn_weight, float* out, int N, int S, int M, int D,
                         int L, int Lq, int P, void* stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out) return MVG_E_BADARG;
  return launch_msda_fwd<float>(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, N, S, M, D,
                                L, Lq, P, (hipStream_t)stream);
}

int mvg_msda_forward_bf16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const float* sampling_loc, const float* attn_weight, void* out, int N, int S, int M, int D,
                          int L, int Lq, int P, void* stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out) return MVG_E_BADARG;
  return launch_msda_fwd<bf16_t>((const bf16_t*)value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                 (bf16_t*)out, N, S, M, D, L, Lq, P, (hipStream_t)stream);
}

int mvg_msda_backward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const float* sampling_loc, const float* attn_weight, const float* grad_output,
                          float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, int N, int S, int M,
                          int D, int L, int Lq, int P, void* stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_output ||
      !grad_value || !grad_sampling_loc || !grad_attn_weight)
    return MVG_E_BADARG;
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return MVG_E_BADARG;
  if (Lq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)N * Lq * M * D;
  const int block = 256;
  const unsigned grid = (unsigned)((total + block - 1) / block);
#define MVG_BWD(DD)                                                                                              \
  hipLaunchKernelGGL((msda_bwd_kernel<DD>), dim3(grid), dim3(block), 0, st, value, spatial_shapes,               \
                     level_start_index, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,    \
                     grad_attn_weight, N, S, M, L, Lq, P, D)
  switch (D) {
    case 4: MVG_BWD(4); break;
    case 8: MVG_BWD(8); break;
    case 16: MVG_BWD(16); break;
    case 32: MVG_BWD(32); break;
    case 64: MVG_BWD(64); break;
    default: {
      // generic head width: atomically accumulated -> needs zeroed outputs
      hipError_t e = hipMemsetAsync(grad_sampling_loc, 0, sizeof(float) * (size_t)N * Lq * M * L * P * 2, st);
      if (e != hipSuccess) return (int)e;
      e = hipMemsetAsync(grad_attn_weight, 0, sizeof(float) * (size_t)N * Lq * M * L * P, st);
      if (e != hipSuccess) return (int)e;
      MVG_BWD(0);
    }
  }
#undef MVG_BWD
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_msda_fused(const void* value, int dtype, const float* oa, const float* r, const int64_t* shapes_host,
                   const int64_t* starts_host, void* samp, int N_img, int Lq, int L, int S, void* stream) {
  if (!value || !oa || !r || !shapes_host || !starts_host || !samp) return MVG_E_BADARG;
  LevelTable lv;
  int e = mvg_fill_levels(&lv, shapes_host, starts_host, L);
  if (e) return e;
  if (L > 4) return MVG_E_BADARG;
  const long pairs = (long)N_img * Lq;
  if (pairs > 0x7fffffffL / 4) return MVG_E_BADARG;
  if (dtype == MVG_F32)
    return launch_msda_fused<float>((const float*)value, oa, r, lv, (float*)samp, (int)pairs, Lq, S, (hipStream_t)stream);
  if (dtype == MVG_BF16)
    return launch_msda_fused<bf16_t>((const bf16_t*)value, oa, r, lv, (bf16_t*)samp, (int)pairs, Lq, S, (hipStream_t)stream);
  return MVG_E_BADARG;
}

}  // extern "C"
