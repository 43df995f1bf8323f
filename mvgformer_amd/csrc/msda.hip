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
// msda_fwd_kernel = the drop-in for Deformable.deform_forward; msda_fused_kernel = generic fused form
// (fp32 / bf16, pixel-major value); msda_gsamp_kernel = the benchmarked bf16 kernel (see its header).
#include <string.h>

#include "common.h"
#include "gsamp_dev.h"


// ------------------------------------------------------------------------------------------
// bilinear sample of 4 consecutive channels with the reference's zero padding
// (cuh:44-94): corners outside [0,H-1]x[0,W-1] contribute 0; the sample is skipped unless
// h_im > -1 && w_im > -1 && h_im < H && w_im < W (cuh:298).  Branch-free: invalid corners
// get weight 0 and a clamped (legal) address.
template <typename T>
__device__ __forceinline__ f32x4 bilinear4(const T* __restrict__ lvl_base, int H, int W, long row_stride,
                                           float h_im, float w_im, float scale) {
  const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
  h_im = index_safe(h_im, (float)H);
  w_im = index_safe(w_im, (float)W);
  const float hl_f = floorf(h_im), wl_f = floorf(w_im);
  const int h_low = (int)hl_f, w_low = (int)wl_f;
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - hl_f, lw = w_im - wl_f;
  const float hh = 1.f - lh, hw = 1.f - lw;
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
  if (!((h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W))) return 0.f;   // also NaN
  const float hl_f = floorf(h_im), wl_f = floorf(w_im);
  const int h_low = (int)hl_f, w_low = (int)wl_f;
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h_im - hl_f, lw = w_im - wl_f;
  const float hh = 1.f - lh, hw = 1.f - lw;
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

// The same units of work (one (n, q, m) on D / 4 lanes, identical operations in identical order: bit-identical outputs) with a
// workgroup on ONE head: its 256 lanes are 256 / (D / 4) consecutive queries of head m = block % M (round 4).  In the mapping above a
// wavefront is one query's M heads -- M different sampling patterns, M different lines per gather instruction, nothing for the 32-KB
// L1 to reuse; consecutive queries of one head (the joints of a person, neighbouring persons) sample the same or adjacent pixels.
// With M = 8 a head's blocks all land on one XCD (block % 8), so each L2 serves one head's channels.  Needs 256 % (D / 4) == 0.
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_hp_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const float* __restrict__ loc, const float* __restrict__ wgt, T* __restrict__ out,
    int N, int S, int M, int D, int L, int Lq, int P, int n_qblocks) {
  const int dv_per = D / 4, qpb = 256 / dv_per;
  const long row_stride = (long)M * D;
  const long n_blocks = (long)N * n_qblocks * M;
  const int dv = threadIdx.x % dv_per, qs = threadIdx.x / dv_per;
  for (long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int m = (int)(blk % M);
    const long t = blk / M;
    const int q = (int)(t % n_qblocks) * qpb + qs;
    const int n = (int)(t / n_qblocks);
    if (q >= Lq) continue;
    const long qm = ((long)n * Lq + q) * M + m;
    const float* locp = loc + qm * L * P * 2;
    const float* wp = wgt + qm * L * P;
    const T* vbase = value + (long)n * S * row_stride + (long)m * D + dv * 4;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    // batches of NB samples: coordinates, zero padding and weights of all NB first, their 4 * NB corner loads issued back to back,
    // then the blends in bilinear4's order (as a plain loop the compiler keeps 4-6 loads in flight; the kernel is latency-bound)
    constexpr int NB = 4;
    const int LP = L * P;
    for (int s0 = 0; s0 < LP; s0 += NB) {
      f32x4 raw[NB][4];
      float cw[NB][4], sc[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int sidx = min(s0 + k, LP - 1);
        const int l = sidx / P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const T* lvl = vbase + (long)starts[l] * row_stride;
        const float2 lxy = *reinterpret_cast<const float2*>(locp + sidx * 2);
        const float aw = wp[sidx];
        float h_im = lxy.y * (float)H - 0.5f;   // cuh:295
        float w_im = lxy.x * (float)W - 0.5f;   // cuh:296
        const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W) && (s0 + k < LP);
        h_im = index_safe(h_im, (float)H);
        w_im = index_safe(w_im, (float)W);
        const float hl_f = floorf(h_im), wl_f = floorf(w_im);
        const int h_low = (int)hl_f, w_low = (int)wl_f;
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - hl_f, lw = w_im - wl_f;
        const float hh = 1.f - lh, hw = 1.f - lw;
        sc[k] = inside ? aw : 0.f;
        const bool hl_ok = h_low >= 0, hh_ok = h_high <= H - 1, wl_ok = w_low >= 0, wh_ok = w_high <= W - 1;
        cw[k][0] = (hl_ok && wl_ok) ? hh * hw : 0.f;
        cw[k][1] = (hl_ok && wh_ok) ? hh * lw : 0.f;
        cw[k][2] = (hh_ok && wl_ok) ? lh * hw : 0.f;
        cw[k][3] = (hh_ok && wh_ok) ? lh * lw : 0.f;
        const int hl_c = min(max(h_low, 0), H - 1), hh_c = min(max(h_high, 0), H - 1);
        const int wl_c = min(max(w_low, 0), W - 1), wh_c = min(max(w_high, 0), W - 1);
        raw[k][0] = Vec4<T>::load(lvl + ((long)hl_c * W + wl_c) * row_stride);
        raw[k][1] = Vec4<T>::load(lvl + ((long)hl_c * W + wh_c) * row_stride);
        raw[k][2] = Vec4<T>::load(lvl + ((long)hh_c * W + wl_c) * row_stride);
        raw[k][3] = Vec4<T>::load(lvl + ((long)hh_c * W + wh_c) * row_stride);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const f32x4 val = cw[k][0] * raw[k][0] + cw[k][1] * raw[k][1] + cw[k][2] * raw[k][2] + cw[k][3] * raw[k][3];
        acc4 += val * sc[k];
      }
    }
    Vec4<T>::store(out + qm * D + dv * 4, acc4);
  }
}

// D = 32 (the decoder's head width): msda_fwd_hp_kernel with the per-sample arithmetic shared out.  There all 8 lanes of a (query, head)
// unit compute every sample's coordinates, zero padding and weights; here lane `sub` prepares sample s0 + sub of a batch of 8 (one
// location / weight load per lane instead of eight identical ones) and the lanes fetch each other's results with ds_swizzle (crossbar
// only), as msda_gfused_f32_hp_kernel's second pass does.  Same operations in the same order per sample: identical outputs.
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_hp8_kernel(
    const T* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const float* __restrict__ loc, const float* __restrict__ wgt, T* __restrict__ out,
    int N, int S, int M, int L, int Lq, int P, int n_qblocks) {
  constexpr int D = 32, QPB = 32;
  const long row_stride = (long)M * D;
  const long n_blocks = (long)N * n_qblocks * M;
  const int sub = threadIdx.x & 7, qs = threadIdx.x >> 3;
  const int LP = L * P;
#define MVG_SW8(K, X) __builtin_amdgcn_ds_swizzle((X), 24 | ((K) << 5))      /* value of lane K of every group of 8 lanes */
  for (long blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int m = (int)(blk % M);
    const long t = blk / M;
    const int q = (int)(t % n_qblocks) * QPB + qs;
    const int n = (int)(t / n_qblocks);
    if (q >= Lq) continue;                               // whole 8-lane groups
    const long qm = ((long)n * Lq + q) * M + m;
    const float* locp = loc + qm * LP * 2;
    const float* wp = wgt + qm * LP;
    const T* vbase = value + (long)n * S * row_stride + (long)m * D + sub * 4;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < LP; s0 += 8) {
      int my_w[4], my_s, my_t, my_b, my_x;
      {
        const int sidx = min(s0 + sub, LP - 1);
        const int l = sidx / P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const float2 lxy = *reinterpret_cast<const float2*>(locp + sidx * 2);
        const float aw = wp[sidx];
        float h_im = lxy.y * (float)H - 0.5f;   // cuh:295
        float w_im = lxy.x * (float)W - 0.5f;   // cuh:296
        const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W) && (s0 + sub < LP);
        h_im = index_safe(h_im, (float)H);
        w_im = index_safe(w_im, (float)W);
        const float hl_f = floorf(h_im), wl_f = floorf(w_im);
        const int h_low = (int)hl_f, w_low = (int)wl_f;
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - hl_f, lw = w_im - wl_f;
        const float hh = 1.f - lh, hw = 1.f - lw;
        my_s = __float_as_int(inside ? aw : 0.f);
        const bool hl_ok = h_low >= 0, hh_ok = h_high <= H - 1, wl_ok = w_low >= 0, wh_ok = w_high <= W - 1;
        my_w[0] = __float_as_int((hl_ok && wl_ok) ? hh * hw : 0.f);
        my_w[1] = __float_as_int((hl_ok && wh_ok) ? hh * lw : 0.f);
        my_w[2] = __float_as_int((hh_ok && wl_ok) ? lh * hw : 0.f);
        my_w[3] = __float_as_int((hh_ok && wh_ok) ? lh * lw : 0.f);
        const int hl_c = min(max(h_low, 0), H - 1), hh_c = min(max(h_high, 0), H - 1);
        const int wl_c = min(max(w_low, 0), W - 1), wh_c = min(max(w_high, 0), W - 1);
        const int st = (int)starts[l];
        my_t = (st + hl_c * W + wl_c) * (int)row_stride;
        my_b = (st + hh_c * W + wl_c) * (int)row_stride;
        my_x = (wh_c - wl_c) * (int)row_stride;
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float cw[4][4], sc[4];
        f32x4 raw[4][4];
#define MVG_FW8(S_, K)                                                                   \
        {                                                                                \
          const int et = MVG_SW8(K, my_t), eb = MVG_SW8(K, my_b), ex = MVG_SW8(K, my_x); \
          cw[S_][0] = __int_as_float(MVG_SW8(K, my_w[0]));                               \
          cw[S_][1] = __int_as_float(MVG_SW8(K, my_w[1]));                               \
          cw[S_][2] = __int_as_float(MVG_SW8(K, my_w[2]));                               \
          cw[S_][3] = __int_as_float(MVG_SW8(K, my_w[3]));                               \
          sc[S_] = __int_as_float(MVG_SW8(K, my_s));                                     \
          raw[S_][0] = Vec4<T>::load(vbase + et);                                        \
          raw[S_][1] = Vec4<T>::load(vbase + (et + ex));                                 \
          raw[S_][2] = Vec4<T>::load(vbase + eb);                                        \
          raw[S_][3] = Vec4<T>::load(vbase + (eb + ex));                                 \
        }
        if (half == 0) {
          MVG_FW8(0, 0) MVG_FW8(1, 1) MVG_FW8(2, 2) MVG_FW8(3, 3)
        } else {
          MVG_FW8(0, 4) MVG_FW8(1, 5) MVG_FW8(2, 6) MVG_FW8(3, 7)
        }
#undef MVG_FW8
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 val = cw[k][0] * raw[k][0] + cw[k][1] * raw[k][1] + cw[k][2] * raw[k][2] + cw[k][3] * raw[k][3];
          acc4 += val * sc[k];
        }
      }
    }
    Vec4<T>::store(out + qm * D + sub * 4, acc4);
  }
#undef MVG_SW8
}

// fp64 drop-in (the `double` case of AT_DISPATCH_FLOATING_TYPES, deform_cuda.cu:75: gradcheck-style calls): one thread
// per (n, q, m, channel), everything in double.  No fast path -- nothing on the decoder's inference path is fp64.
__global__ __launch_bounds__(256) void msda_fwd_f64_kernel(
    const double* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const double* __restrict__ loc, const double* __restrict__ wgt, double* __restrict__ out,
    int N, int S, int M, int D, int L, int Lq, int P) {
  const long total = (long)N * Lq * M * D;
  const long row_stride = (long)M * D;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % D);
    long t = idx / D;
    const int m = (int)(t % M);
    t /= M;
    const int q = (int)(t % Lq);
    const int n = (int)(t / Lq);
    const long qm = ((long)n * Lq + q) * M + m;
    const double* vb = value + (long)n * S * row_stride + (long)m * D + c;
    double acc = 0.0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const double* lvl = vb + (long)starts[l] * row_stride;
      for (int p = 0; p < P; ++p) {
        const long sidx = qm * L * P + l * P + p;
        const double h_im = loc[sidx * 2 + 1] * (double)H - 0.5, w_im = loc[sidx * 2] * (double)W - 0.5;   // cuh:295-296
        if (!(h_im > -1.0 && w_im > -1.0 && h_im < (double)H && w_im < (double)W)) continue;              // cuh:298 (also NaN)
        const double hl_f = floor(h_im), wl_f = floor(w_im);
        const int h_low = (int)hl_f, w_low = (int)wl_f, h_high = h_low + 1, w_high = w_low + 1;
        const double lh = h_im - hl_f, lw = w_im - wl_f, hh = 1.0 - lh, hw = 1.0 - lw;
        double v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
        if (h_low >= 0 && w_low >= 0) v1 = lvl[((long)h_low * W + w_low) * row_stride];
        if (h_low >= 0 && w_high <= W - 1) v2 = lvl[((long)h_low * W + w_high) * row_stride];
        if (h_high <= H - 1 && w_low >= 0) v3 = lvl[((long)h_high * W + w_low) * row_stride];
        if (h_high <= H - 1 && w_high <= W - 1) v4 = lvl[((long)h_high * W + w_high) * row_stride];
        acc += wgt[sidx] * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
      }
    }
    out[qm * D + c] = acc;
  }
}

// fp64 backward: same decomposition as msda_bwd_kernel<0> (atomics on all three gradients, outputs zeroed by the entry point)
__global__ __launch_bounds__(256) void msda_bwd_f64_kernel(
    const double* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ starts,
    const double* __restrict__ loc, const double* __restrict__ wgt, const double* __restrict__ gout,
    double* __restrict__ gvalue, double* __restrict__ gloc, double* __restrict__ gwgt,
    int N, int S, int M, int D, int L, int Lq, int P) {
  const long total = (long)N * Lq * M * D;
  const long row_stride = (long)M * D;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % D);
  long t = idx / D;
  const int m = (int)(t % M);
  t /= M;
  const int q = (int)(t % Lq);
  const int n = (int)(t / Lq);
  const long qm = ((long)n * Lq + q) * M + m;
  const double go = gout[qm * D + c];
  const double* vb = value + (long)n * S * row_stride + (long)m * D + c;
  double* gvb = gvalue + (long)n * S * row_stride + (long)m * D + c;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long lbase = (long)starts[l] * row_stride;
    for (int p = 0; p < P; ++p) {
      const long sidx = qm * L * P + l * P + p;
      const double aw = wgt[sidx];
      const double h_im = loc[sidx * 2 + 1] * (double)H - 0.5, w_im = loc[sidx * 2] * (double)W - 0.5;
      if (!(h_im > -1.0 && w_im > -1.0 && h_im < (double)H && w_im < (double)W)) continue;
      const double hl_f = floor(h_im), wl_f = floor(w_im);
      const int h_low = (int)hl_f, w_low = (int)wl_f, h_high = h_low + 1, w_high = w_low + 1;
      const double lh = h_im - hl_f, lw = w_im - wl_f, hh = 1.0 - lh, hw = 1.0 - lw;
      const double top = go * aw;
      double v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0;
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
      atomicAdd(gloc + sidx * 2, (hh * (v2 - v1) + lh * (v4 - v3)) * top * (double)W);     // cuh:128-167
      atomicAdd(gloc + sidx * 2 + 1, (hw * (v3 - v1) + lw * (v4 - v2)) * top * (double)H);
      atomicAdd(gwgt + sidx, go * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4));
    }
  }
}

static int g_fwd_map = 1;          // tuning knob "fwd_map": mvg_msda_forward, 0 = a wavefront takes the M heads of a query, 1 = one head per workgroup (D = 32: per-sample arithmetic shared out over the unit's 8 lanes), 2 = one head per workgroup, every lane computes every sample
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
  if (vec && g_fwd_map >= 1 && 256 % (D / 4) == 0) {
    const int qpb = 256 / (D / 4), n_qblocks = (Lq + qpb - 1) / qpb;
    long blocks = (long)N * n_qblocks * M;
    if (blocks > (1L << 22)) blocks = (1L << 22) / M * M;        // grid-stride, a whole number of heads per stride
    if (D == 32 && g_fwd_map == 1 && (long)S * M * D < (1L << 31))
      hipLaunchKernelGGL((msda_fwd_hp8_kernel<T>), dim3((unsigned)blocks), dim3(block), 0, st, value, shapes, starts, loc, wgt, out, N, S,
                         M, L, Lq, P, n_qblocks);
    else
      hipLaunchKernelGGL((msda_fwd_hp_kernel<T>), dim3((unsigned)blocks), dim3(block), 0, st, value, shapes, starts, loc, wgt, out, N, S,
                         M, D, L, Lq, P, n_qblocks);
  } else if (vec)
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
// Work decomposition: CPL channels per lane (4: 8 lanes per head, one (image,query) pair per
// wavefront; 8: 4 lanes per head, two pairs per wavefront -- bf16 only, 16-byte loads).  The
// sampling loop is software-pipelined by hand: the corner addresses / weights of NB samples are
// computed first, their 4*NB vector loads are issued back to back (16 loads in flight per
// wavefront instead of the 4-6 the compiler schedules on its own -- the kernel is latency-bound
// on L2/MALL gathers otherwise), and only then are they blended into the accumulators.
template <typename T, int CPL> struct RawVec;
template <> struct RawVec<float, 4> {
  typedef f32x4 type;
  static __device__ __forceinline__ type load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void fma(float (&acc)[4], const type& v, float w) {
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = fmaf(w, v[c], acc[c]);
  }
};
template <> struct RawVec<bf16_t, 4> {
  typedef uint2 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ void fma(float (&acc)[4], const type& v, float w) {
    acc[0] = fmaf(w, __uint_as_float(v.x << 16), acc[0]);
    acc[1] = fmaf(w, __uint_as_float(v.x & 0xffff0000u), acc[1]);
    acc[2] = fmaf(w, __uint_as_float(v.y << 16), acc[2]);
    acc[3] = fmaf(w, __uint_as_float(v.y & 0xffff0000u), acc[3]);
  }
};
template <> struct RawVec<bf16_t, 8> {
  typedef uint4 type;
  static __device__ __forceinline__ type load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void fma(float (&acc)[8], const type& v, float w) {
    acc[0] = fmaf(w, __uint_as_float(v.x << 16), acc[0]);
    acc[1] = fmaf(w, __uint_as_float(v.x & 0xffff0000u), acc[1]);
    acc[2] = fmaf(w, __uint_as_float(v.y << 16), acc[2]);
    acc[3] = fmaf(w, __uint_as_float(v.y & 0xffff0000u), acc[3]);
    acc[4] = fmaf(w, __uint_as_float(v.z << 16), acc[4]);
    acc[5] = fmaf(w, __uint_as_float(v.z & 0xffff0000u), acc[5]);
    acc[6] = fmaf(w, __uint_as_float(v.w << 16), acc[6]);
    acc[7] = fmaf(w, __uint_as_float(v.w & 0xffff0000u), acc[7]);
  }
};

template <typename T, int CPL>
__device__ __forceinline__ void store_acc(T* p, const float (&acc)[CPL]) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{acc[0], acc[1], acc[2], acc[3]};
  } else if constexpr (CPL == 4) {
    uint2 o;
    o.x = pack_bf16(acc[0], acc[1]);
    o.y = pack_bf16(acc[2], acc[3]);
    *reinterpret_cast<uint2*>(p) = o;
  } else {
    uint4 o;
    o.x = pack_bf16(acc[0], acc[1]);
    o.y = pack_bf16(acc[2], acc[3]);
    o.z = pack_bf16(acc[4], acc[5]);
    o.w = pack_bf16(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(p) = o;
  }
}

template <typename T, int L, int CPL, int NB>
__global__ __launch_bounds__(256, (CPL * NB * (int)sizeof(T) >= 64 ? 3 : 4)) void msda_fused_kernel(const T* __restrict__ value, const float* __restrict__ oa,
                                                         const float* __restrict__ r, LevelTable lv,
                                                         T* __restrict__ samp, int n_pairs, int Lq, int S) {
  constexpr int D = 32, P = 8, C = 256, LP = L * P;   // M = 8 heads
  constexpr int LPH = D / CPL;                        // lanes per head (8 or 4)
  constexpr int PPW = 64 / (8 * LPH);                 // (image,query) pairs per wavefront (1 or 2)
  // NB = samples per gather batch (P % NB == 0): 4*NB vector loads in flight per lane
  typedef RawVec<T, CPL> RV;
  // XCD-aware block remap (bijective): XCD x = blockIdx % 8 walks a contiguous block range
  const int nb = gridDim.x;
  const int q8 = nb >> 3, r8 = nb & 7;
  const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
  const int lblock = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + within;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int pair = (lblock * 4 + wave) * PPW + lane / (8 * LPH);
  const int m = (lane / LPH) & 7, sub = lane % LPH;
  const bool live = pair < n_pairs;
  if (!live) pair = n_pairs - 1;                       // keep the wavefront converged; store is masked
  const int n = pair / Lq;

  // ---- pass 1: max and sum of the head's L*P logits (softmax denominator), redundantly on the
  //      LPH lanes of a head; 16-byte loads that all lanes of a head share (one broadcast each)
  const float* oa_q = oa + (long)pair * L * 192;
  float mx = -INFINITY;
  {
    f32x4 lg[LP / 4];
#pragma unroll
    for (int i = 0; i < LP / 4; ++i) {
      const int fa = m * LP + 4 * i;
      lg[i] = *reinterpret_cast<const f32x4*>(oa_q + (fa >> 6) * 192 + 128 + (fa & 63));
      mx = fmaxf(fmaxf(fmaxf(mx, lg[i][0]), fmaxf(lg[i][1], lg[i][2])), lg[i][3]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP / 4; ++i)
      sum += __expf(lg[i][0] - mx) + __expf(lg[i][1] - mx) + __expf(lg[i][2] - mx) + __expf(lg[i][3] - mx);
    // fold 1/sum into the exponent: w = exp(x - mx - log(sum))
    mx += __logf(sum);
  }

  const T* vbase = value + (long)n * S * C + m * D + sub * CPL;
  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.f;

  // ---- pass 2: a REAL loop over batches of NB samples (not unrolled: a fully unrolled body lets
  //      the compiler pre-compute all 24 samples' corner weights / indices and spill); inside one
  //      iteration the 4*NB gathers are issued back to back before the first blend.
#pragma unroll 1
  for (int it = 0; it < LP / NB; ++it) {
    const int l = (it * NB) / P, b0 = (it * NB) % P;
    const int H = lv.H[l], W = lv.W[l];
    const float Wf = (float)W, Hf = (float)H;
    const float refx = r[((long)pair * L + l) * 2], refy = r[((long)pair * L + l) * 2 + 1];
    const float invW = 1.f / Wf, invH = 1.f / Hf;
    const T* lvl = vbase + (long)lv.start[l] * C;
    const int fa = m * LP + it * NB;
    f32x4 lg[NB / 4];
#pragma unroll
    for (int i = 0; i < NB / 4; ++i)
      lg[i] = *reinterpret_cast<const f32x4*>(oa_q + ((fa + 4 * i) >> 6) * 192 + 128 + ((fa + 4 * i) & 63));
    f32x4 o4[NB / 2];
#pragma unroll
    for (int i = 0; i < NB / 2; ++i) {
      const int fo = ((m * L + l) * P + b0 + 2 * i) * 2;
      o4[i] = *reinterpret_cast<const f32x4*>(oa_q + (fo >> 7) * 192 + (fo & 127));
    }
    float cw[NB][4];
    typename RV::type raw[NB][4];
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const float lx = refx + o4[s >> 1][2 * (s & 1)] * invW;       // projattn.py:186-191
      const float ly = refy + o4[s >> 1][2 * (s & 1) + 1] * invH;
      const float h_raw = ly * Hf - 0.5f;                            // cuh:295-296
      const float w_raw = lx * Wf - 0.5f;
      const bool inside = (h_raw > -1.f) && (w_raw > -1.f) && (h_raw < Hf) && (w_raw < Wf);   // cuh:298
      const float h_im = index_safe(h_raw, Hf), w_im = index_safe(w_raw, Wf);
      const float hl_f = floorf(h_im), wl_f = floorf(w_im);
      const int h_low = (int)hl_f, w_low = (int)wl_f;
      const float lh = h_im - hl_f, lw = w_im - wl_f, hh = 1.f - lh, hw = 1.f - lw;
      const float a = inside ? __expf(lg[s >> 2][s & 3] - mx) : 0.f;   // softmax weight (projattn.py:184)
      const bool hl_ok = h_low >= 0, hh_ok = h_low + 1 <= H - 1, wl_ok = w_low >= 0, wh_ok = w_low + 1 <= W - 1;
      cw[s][0] = (hl_ok && wl_ok) ? hh * hw * a : 0.f;               // cuh:66-88 zero padding
      cw[s][1] = (hl_ok && wh_ok) ? hh * lw * a : 0.f;
      cw[s][2] = (hh_ok && wl_ok) ? lh * hw * a : 0.f;
      cw[s][3] = (hh_ok && wh_ok) ? lh * lw * a : 0.f;
      const int hl_c = min(max(h_low, 0), H - 1), hh_c = min(max(h_low + 1, 0), H - 1);
      const int wl_c = min(max(w_low, 0), W - 1), wh_c = min(max(w_low + 1, 0), W - 1);
      raw[s][0] = RV::load(lvl + (hl_c * W + wl_c) * C);
      raw[s][1] = RV::load(lvl + (hl_c * W + wh_c) * C);
      raw[s][2] = RV::load(lvl + (hh_c * W + wl_c) * C);
      raw[s][3] = RV::load(lvl + (hh_c * W + wh_c) * C);
    }
    __builtin_amdgcn_sched_barrier(0);   // all 4*NB loads are issued before the first blend
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k) RV::fma(acc, raw[s][k], cw[s][k]);
  }
  if (live) store_acc<T, CPL>(samp + (long)pair * C + m * D + sub * CPL, acc);
}

// ------------------------------------------------------------------------------------------
// msda_gfused_f32_hp_kernel -- G-sampling in the reference's own arithmetic (fp32 storage, fp32 math; rounds 2 / 4).
// msda_fused_kernel needs the (pairs * L, 192) tensor `oa` of offsets / logits, i.e. a gather of 256-channel reference-point
// rows (`ain`, 236 MB per layer at cfg-2) and a (V * Lq * L) x 256 x 192 GEMM (177 MB out) per layer.  Bilinear sampling commutes
// with the Linear (see msda_gsamp_kernel), so here the Linear is applied once to the pyramid (G = feat @ Woa^T, fp32, columns
// in ops.gsamp_column_order) and every (pair, head) gathers its 24 logits + 48 offsets from G at the reference point and adds
// xw = (tgt + query_pos) @ Woa^T + b -- phase A below, 8 lanes per head, results parked in LDS --, then softmax, locations and
// sampling exactly as msda_fused_kernel<float> does them.
// Work decomposition (round 4; the round-2 kernel -- a wavefront took the 8 heads of one pair -- computed identical rows, 368 us
// against 238 at cfg-2, and is deleted): one (pair, head) on 8 lanes, a wavefront takes 8 NEIGHBOURING pairs of ONE head.  The 8 heads of a pair follow 8 different rays, so no two lanes groups of a gather instruction shared a 128-byte line and the
// lines in flight (16 loads x 8 heads per wavefront) turned the 32-KB L1 over before a neighbour could reuse them: 43 % L1 hits, 29.8 M
// L2 requests per launch against the bf16 kernel's 7.3 M (profiles/r04_rocprofv3_summary_fp32.txt).  Neighbouring pairs of one head sample
// the same or adjacent pixels: the lane groups of one instruction coalesce, and a workgroup's footprint is one head's patch.
#ifndef MVG_GFUSED_HP_OCC
#define MVG_GFUSED_HP_OCC 4
#endif
template <int L>
__global__ __launch_bounds__(256, MVG_GFUSED_HP_OCC) void msda_gfused_f32_hp_kernel(const float* __restrict__ value, const float* __restrict__ G,
                                                                 const float* __restrict__ xw, const float* __restrict__ r,
                                                                 LevelTable lv, float* __restrict__ samp,
                                                                 const uint8_t* __restrict__ pair_mask,
                                                                 const int* __restrict__ order, int n_pairs,
                                                                 int Lq, int S, int B, int map_ch) {
  constexpr int D = 32, P = 8, C = 256, LP = L * P, NB = 4, CPL = 4, SCP = 3 * LP + 8;
  typedef RawVec<float, CPL> RV;
  __shared__ __attribute__((aligned(16))) float scratch[4][8][SCP];
  // one head per workgroup, blockIdx & 7 = head = the XCD the hardware dispatches the block to: an XCD's L2 sees one head's lines;
  // blockIdx >> 3 walks the processing order in blocks of 32 pairs, a wavefront = 8 neighbouring pairs x 8 lanes
  int m, pb;
  if (map_ch == 0) {
    m = blockIdx.x & 7;
    pb = blockIdx.x >> 3;
  } else {                     // chunks of map_ch pair blocks per XCD, their 8 heads back to back (msda_gsamp_kernel's mapping)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    m = j & 7;
    const int t = j >> 3;
    pb = ((t / map_ch) * 8 + xcd) * map_ch + t % map_ch;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = lane >> 3, sub = lane & 7;
  const int slot = pb * 32 + wave * 8 + g;
  if (slot >= n_pairs) return;                         // whole 8-lane groups leave (the exchanges below stay inside a group)
  // Round 6: lane 0 of the group reads the slot's pair and its mask byte, lanes 0..L-1 the reference point of level `lane`; the
  // others get them through the swizzle crossbar.  The texture-address path takes a clock per 4 ACTIVE lanes of a load and this
  // kernel runs at its rate (profiles/r06_experiments.txt sections 2, 5): 8 + 8 + 8 L lane addresses per wavefront for 5 x 64.
#define MVG_GRP8(K, X) __builtin_amdgcn_ds_swizzle((X), 24 | ((K) << 5))      /* value of lane K of every group of 8 lanes */
  static_assert(L <= 8, "lane `sub` of a group carries the reference point of level `sub`");
  int pair = slot;
  if (order) {
    int p0 = 0;
    if (sub == 0) p0 = order[slot];
    pair = MVG_GRP8(0, p0);
  }
  float2 mine = float2{0.f, 0.f};
  if (sub < L) mine = *reinterpret_cast<const float2*>(r + ((long)pair * L + sub) * 2);
  if (pair_mask) {
    int mk = 0;
    if (sub == 0) mk = pair_mask[pair];
    if (!MVG_GRP8(0, mk)) {
      *reinterpret_cast<f32x4*>(samp + (long)pair * C + m * D + sub * CPL) = f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    }
  }
  const bool live = true;
  const int n = pair / Lq, q = pair - n * Lq, b = n % B;
  float* sc = &scratch[wave][g][0];

  // ---- phase A: this head's L*P logits and 2*L*P offsets = bilinear(G) + xw.  The head's 3 * L chunks of 8 columns are 6 * L pieces
  // of 16 bytes; lane `sub` takes pieces sub, sub + 8, ...: the 8 lanes of a gather instruction read 128 CONTIGUOUS bytes of one corner's
  // G row (two 64-byte accesses) where the chunk-per-lane form of msda_gfused_f32_kernel read 8 x 16 bytes at a 32-byte stride, twice
  // (four accesses each), plus a second round in which only lane 0 had a chunk.  Per element the same products in the same order.
  constexpr int NPC = 6 * L, NK = (NPC + 7) / 8;
  float2 rr[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int mx_ = __float_as_int(mine.x), my_ = __float_as_int(mine.y);
    switch (l) {          // (the swizzle pattern is an immediate)
      case 0: rr[l] = float2{__int_as_float(MVG_GRP8(0, mx_)), __int_as_float(MVG_GRP8(0, my_))}; break;
      case 1: rr[l] = float2{__int_as_float(MVG_GRP8(1, mx_)), __int_as_float(MVG_GRP8(1, my_))}; break;
      case 2: rr[l] = float2{__int_as_float(MVG_GRP8(2, mx_)), __int_as_float(MVG_GRP8(2, my_))}; break;
      case 3: rr[l] = float2{__int_as_float(MVG_GRP8(3, mx_)), __int_as_float(MVG_GRP8(3, my_))}; break;
      case 4: rr[l] = float2{__int_as_float(MVG_GRP8(4, mx_)), __int_as_float(MVG_GRP8(4, my_))}; break;
      case 5: rr[l] = float2{__int_as_float(MVG_GRP8(5, mx_)), __int_as_float(MVG_GRP8(5, my_))}; break;
      case 6: rr[l] = float2{__int_as_float(MVG_GRP8(6, mx_)), __int_as_float(MVG_GRP8(6, my_))}; break;
      default: rr[l] = float2{__int_as_float(MVG_GRP8(7, mx_)), __int_as_float(MVG_GRP8(7, my_))}; break;
    }
  }
#undef MVG_GRP8
  f32x4 ga[NK], gb[NK], gc[NK], gd[NK], gx4[NK];
  float w00[NK], w10[NK], w01[NK], w11[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int pj = min(sub + 8 * k, NPC - 1);          // piece: chunk pj >> 1, half pj & 1
    const int ci = pj >> 1;
    const int t = ci / 3, part = ci - 3 * t;           // group t of the head (= level of its samples), 16 offsets | 8 logits
    const int fg = m * L + t;
    const int l = fg >> 3;                             // level row of the reinterpreted view (projattn.py:180-184)
    const int col = 24 * (fg & 7) + 8 * part + 4 * (pj & 1);
    const int H = lv.H[l], W = lv.W[l];
    const float Wf = (float)W, Hf = (float)H;
    float refx = rr[0].x, refy = rr[0].y;
#pragma unroll
    for (int ll = 1; ll < L; ++ll) {
      refx = l == ll ? rr[ll].x : refx;
      refy = l == ll ? rr[ll].y : refy;
    }
    const float gx = fminf(fmaxf(refx * 2.f - 1.f, -1.1f), 1.1f);        // projattn.py:134
    const float gy = fminf(fmaxf(refy * 2.f - 1.f, -1.1f), 1.1f);
    const float ix = ((gx + 1.f) * Wf - 1.f) * 0.5f, iy = ((gy + 1.f) * Hf - 1.f) * 0.5f;    // grid_sample, align_corners=False
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = ix - x0f, ty = iy - y0f;
    const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
    w00[k] = (x0ok && y0ok) ? (1.f - tx) * (1.f - ty) : 0.f;
    w10[k] = (x1ok && y0ok) ? tx * (1.f - ty) : 0.f;
    w01[k] = (x0ok && y1ok) ? (1.f - tx) * ty : 0.f;
    w11[k] = (x1ok && y1ok) ? tx * ty : 0.f;
    const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x1, 0), W - 1);
    const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y1, 0), H - 1);
    const float* gp = G + ((long)n * S + lv.start[l]) * 192 + col;
    // the last round has pieces for NPC % 8 lanes of the group only: the others do not load (nor store below)
    if (8 * k + 8 <= NPC || sub + 8 * k < NPC) {
      ga[k] = *reinterpret_cast<const f32x4*>(gp + (long)(y0c * W + x0c) * 192);
      gb[k] = *reinterpret_cast<const f32x4*>(gp + (long)(y0c * W + x1c) * 192);
      gc[k] = *reinterpret_cast<const f32x4*>(gp + (long)(y1c * W + x0c) * 192);
      gd[k] = *reinterpret_cast<const f32x4*>(gp + (long)(y1c * W + x1c) * 192);
      gx4[k] = *reinterpret_cast<const f32x4*>(xw + ((long)b * Lq + q) * 192 + col);
    } else {
      ga[k] = gb[k] = gc[k] = gd[k] = gx4[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int pj = sub + 8 * k;
    const int ci = min(pj, NPC - 1) >> 1;
    const int t = ci / 3, part = ci - 3 * t;
    if (pj < NPC) {
      float* dst = sc + (part == 2 ? 8 * t : LP + 16 * t + 8 * part) + 4 * (pj & 1);
      *reinterpret_cast<f32x4*>(dst) = w00[k] * ga[k] + w10[k] * gb[k] + w01[k] * gc[k] + w11[k] * gd[k] + gx4[k];
    }
  }
  // head-private scratch rows inside one wavefront: LDS operations of a wavefront execute in order
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- pass 1: max and sum of the head's L*P logits (softmax denominator), redundantly on the 8 lanes of a head
  float mx = -INFINITY;
  {
    f32x4 lg[LP / 4];
#pragma unroll
    for (int i = 0; i < LP / 4; ++i) {
      lg[i] = *reinterpret_cast<const f32x4*>(sc + 4 * i);
      mx = fmaxf(fmaxf(fmaxf(mx, lg[i][0]), fmaxf(lg[i][1], lg[i][2])), lg[i][3]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP / 4; ++i)
      sum += __expf(lg[i][0] - mx) + __expf(lg[i][1] - mx) + __expf(lg[i][2] - mx) + __expf(lg[i][3] - mx);
    mx += __logf(sum);                                   // fold 1/sum into the exponent: w = exp(x - mx - log(sum))
  }

  const float* vbase = value + (long)n * S * C + m * D + sub * CPL;
  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
  // ---- pass 2 (round 3).  The first form computed every sample's coordinates, zero padding and softmax weight on all 8 lanes of
  // the head (PMC: 2 425 VALU instructions per wavefront, the VALU 57 % busy -- the kernel's bound).  Now lane `sub` computes ONE of
  // the 8 samples of a level (P = 8) and the 8 lanes exchange the results with ds_swizzle (crossbar only, no LDS memory): 4 corner
  // weights + 3 element offsets per sample; two half batches of 4 samples = 16 gathers in flight as before.  Same operations in
  // the same order for every sample and every accumulation: bit-identical to the first form.
  static_assert(P == 8 && NB == 4, "one sample per lane of the head's 8");
#define MVG_SW8(K, X) __builtin_amdgcn_ds_swizzle((X), 24 | ((K) << 5))      /* value of lane K of every group of 8 lanes */
#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const float Wf = (float)W, Hf = (float)H;
    float refx = rr[0].x, refy = rr[0].y;
#pragma unroll
    for (int ll = 1; ll < L; ++ll) {
      refx = l == ll ? rr[ll].x : refx;
      refy = l == ll ? rr[ll].y : refy;
    }
    const float* lvl = vbase + (long)lv.start[l] * C;
    int my_w[4], my_t, my_b, my_x;
    {
      const float lgs = sc[l * P + sub];
      const float2 of = *reinterpret_cast<const float2*>(sc + LP + (l * P + sub) * 2);
      const float lx = refx + of.x * lv.invW[l];                         // projattn.py:186-191
      const float ly = refy + of.y * lv.invH[l];
      const float h_raw = ly * Hf - 0.5f;                                // cuh:295-296
      const float w_raw = lx * Wf - 0.5f;
      const bool inside = (h_raw > -1.f) && (w_raw > -1.f) && (h_raw < Hf) && (w_raw < Wf);   // cuh:298
      const float h_im = index_safe(h_raw, Hf), w_im = index_safe(w_raw, Wf);
      const float hl_f = floorf(h_im), wl_f = floorf(w_im);
      const int h_low = (int)hl_f, w_low = (int)wl_f;
      const float lh = h_im - hl_f, lw = w_im - wl_f, hh = 1.f - lh, hw = 1.f - lw;
      const float a = inside ? __expf(lgs - mx) : 0.f;                   // softmax weight (projattn.py:184)
      const bool hl_ok = h_low >= 0, hh_ok = h_low + 1 <= H - 1, wl_ok = w_low >= 0, wh_ok = w_low + 1 <= W - 1;
      my_w[0] = __float_as_int((hl_ok && wl_ok) ? hh * hw * a : 0.f);   // cuh:66-88 zero padding
      my_w[1] = __float_as_int((hl_ok && wh_ok) ? hh * lw * a : 0.f);
      my_w[2] = __float_as_int((hh_ok && wl_ok) ? lh * hw * a : 0.f);
      my_w[3] = __float_as_int((hh_ok && wh_ok) ? lh * lw * a : 0.f);
      const int hl_c = min(max(h_low, 0), H - 1), hh_c = min(max(h_low + 1, 0), H - 1);
      const int wl_c = min(max(w_low, 0), W - 1), wh_c = min(max(w_low + 1, 0), W - 1);
      my_t = (hl_c * W + wl_c) * C;
      my_b = (hh_c * W + wl_c) * C;
      my_x = (wh_c - wl_c) * C;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float cw[NB][4];
      typename RV::type raw[NB][4];
#define MVG_GS8(S_, K)                                                                                \
      {                                                                                               \
        const int et = MVG_SW8(K, my_t), eb = MVG_SW8(K, my_b), ex = MVG_SW8(K, my_x);                \
        cw[S_][0] = __int_as_float(MVG_SW8(K, my_w[0]));                                              \
        cw[S_][1] = __int_as_float(MVG_SW8(K, my_w[1]));                                              \
        cw[S_][2] = __int_as_float(MVG_SW8(K, my_w[2]));                                              \
        cw[S_][3] = __int_as_float(MVG_SW8(K, my_w[3]));                                              \
        raw[S_][0] = RV::load(lvl + et);                                                              \
        raw[S_][1] = RV::load(lvl + (et + ex));                                                       \
        raw[S_][2] = RV::load(lvl + eb);                                                              \
        raw[S_][3] = RV::load(lvl + (eb + ex));                                                       \
      }
      if (half == 0) {
        MVG_GS8(0, 0) MVG_GS8(1, 1) MVG_GS8(2, 2) MVG_GS8(3, 3)
      } else {
        MVG_GS8(0, 4) MVG_GS8(1, 5) MVG_GS8(2, 6) MVG_GS8(3, 7)
      }
#undef MVG_GS8
      __builtin_amdgcn_sched_barrier(0);   // all 16 loads are issued before the first blend
#pragma unroll
      for (int s_ = 0; s_ < NB; ++s_)
#pragma unroll
        for (int k = 0; k < 4; ++k) RV::fma(acc, raw[s_][k], cw[s_][k]);
    }
  }
#undef MVG_SW8
  if (live) store_acc<float, CPL>(samp + (long)pair * C + m * D + sub * CPL, acc);
}

// ------------------------------------------------------------------------------------------
// msda_gsamp_kernel -- the bf16 sampling kernel of the decoder (the kernel bench.py's roofline reports).
//
// (1) G-sampling.  Bilinear sampling commutes with a Linear:
//        Linear(bilinear(feat, p) + x) = bilinear(feat @ W^T, p) + (x @ W^T + b).
//     Instead of gathering 256-channel reference-point features and running a (V*Lq*L x 256 x 192) GEMM per
//     layer (projattn.py:148-153,180-181), the offsets/logits Linear is applied ONCE to the pyramid
//     (G = feat @ Woa^T, a (V*S x 192) GEMM independent of the queries) and every (pair, head) gathers its own
//     24 logits + 48 offsets from G at the reference point (9 chunks of 8 columns, split over the 4 lanes of
//     the head, parked in LDS) and adds xw = (tgt+query_pos) @ Woa^T + b.  The 192 columns of G / xw are ordered
//     so that a head's chunks are contiguous in a pixel's row (see phase A).  No (rows x 192) fp32 tensor
//     (177 MB written + read per layer) and no per-(view, query, level) GEMM exist any more.
// (2) Head-plane value layout (written by wreg_gemm.hip):  vh[img][head][s][ch 32]: a pixel's 32 channels of one head
//     are 64 contiguous bytes (16 per lane of the head's quad).  The left / right corner pixels of a sample are
//     interleaved per channel with v_perm_b32 into (left[ch], right[ch]) words = the operand of v_dot2c_f32_bf16 with
//     the packed weights (w_left, w_right).  (A "pixel-pair" layout that stored this interleaving -- half the
//     requests, no perm, twice the bytes -- was faster while the pairs were processed in query order; in image-space
//     order the smaller footprint wins.)
// (3) Head-per-XCD mapping: block b works on head (b & 7); with gfx950's round-robin dispatch (block b -> XCD
//     b % 8) each XCD touches one head plane of vp (5 MB per view) instead of all eight, so L1 misses hit in its
//     4-MB L2 (fabric reads 10.3 M -> 4.5 M requests per launch).  A wavefront = 16 pairs x 1 head x 4 lanes,
//     each lane owning 8 channels.
// (4) VALU-lean inner loop (the previous form was VALU-bound: PMC 73 % VALU-busy, 3180 VALU instructions per
//     wavefront): each lane of a head's quad computes ONE of the 4 samples of a batch (coordinates, zero
//     padding, softmax weight, row offsets) and broadcasts 4 words with DPP quad_perm; the 16 gathers of the batch
//     are issued back to back (sched_barrier: hipcc otherwise sinks them to their uses); the blend is 64
//     v_dot2c_f32_bf16 per batch, fp32 accumulation, no bf16->fp32 unpacking.  (The bilinear x attention
//     weights are rounded to bf16; products and sums are fp32.)
// Probe build only (-DGSAMP_STAMPS, tools/probes/stamps_gsamp.py): wavefront 0 of every workgroup appends [blockIdx, start, end
// (s_memrealtime), HW_ID | XCC_ID << 32, lanes that sampled] to a device buffer.
#ifdef GSAMP_STAMPS
__device__ unsigned long long gsamp_stamps[65536 * 4];
__device__ unsigned int gsamp_stamp_count;
extern "C" int mvg_gsamp_read_stamps(unsigned long long* host, int max_records, int reset) {
  unsigned n = 0;
  hipError_t e = hipMemcpyFromSymbol(&n, HIP_SYMBOL(gsamp_stamp_count), sizeof(n));
  if (e != hipSuccess) return -(int)e;
  if (n > 65536) n = 65536;
  if ((int)n > max_records) n = max_records;
  if (host && n) e = hipMemcpyFromSymbol(host, HIP_SYMBOL(gsamp_stamps), sizeof(unsigned long long) * 4 * n);
  if (e != hipSuccess) return -(int)e;
  if (reset) {
    const unsigned z = 0;
    e = hipMemcpyToSymbol(HIP_SYMBOL(gsamp_stamp_count), &z, sizeof(z));
    if (e != hipSuccess) return -(int)e;
  }
  return (int)n;
}
#endif

template <int L, int NT, int PIPE = 0>   // NT threads per workgroup = NT/4 consecutive slots of the processing order, one head
__device__ __forceinline__ void msda_gsamp_body(const bf16_t* __restrict__ vp, const bf16_t* __restrict__ G,
                                                const float* __restrict__ xw, const float* __restrict__ r,
                                                const LevelTable& lv, bf16_t* __restrict__ samp,
                                                const uint8_t* __restrict__ pair_mask,
                                                const int* __restrict__ order, int n_pairs,
                                                int Lq, int S, int B, int map_ch) {
  constexpr int SCP = 3 * L * 8 + 8;   // scratch row: L*P logits, 2*L*P offsets, L reference points
  __shared__ __attribute__((aligned(16))) float scratch[NT / 64][16][SCP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane & 3, pl = lane >> 2;
  // workgroup -> (head, slot block).  map_ch == 0: head = blockIdx & 7 = the XCD the block is dispatched to (each L2
  // serves one head plane).  map_ch > 0 (default 4): XCD = blockIdx & 7 works on chunks of map_ch consecutive slot
  // blocks, the 8 heads of a slot block back to back -- with the pairs in image-space order an L2 then serves a
  // compact region of all 8 head planes, and the G rows of a pair are fetched into it once for its 8 heads.
  int m, pblk;
  if (map_ch == 0) {
    m = blockIdx.x & 7;
    pblk = blockIdx.x >> 3;
  } else {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    m = j & 7;
    const int t = j >> 3;
    pblk = ((t / map_ch) * 8 + xcd) * map_ch + t % map_ch;
  }
  // The 4 lanes of a quad share one (pair, head) and every LDS scratch row is private to its quad, so lanes may
  // leave early (no workgroup barrier below): slots past the end, and pairs the caller masks out (reference
  // points outside the image: the consumer multiplies their rows by 0, dq_decoder.py:585-586) -- zero-filled.
  const int slot = pblk * (NT / 4) + wave * 16 + pl;
#ifdef GSAMP_STAMPS
  const unsigned long long gs_t0 = __builtin_amdgcn_s_memrealtime();
  auto gs_flush = [&](bool sampled) {
    const unsigned long long act = __ballot(sampled);
    // (lane 0 may have left: the first lane still here writes)
    const int first = __ffsll((unsigned long long)__ballot(true)) - 1;
    if (wave == 0 && lane == first) {
      const unsigned s_ = atomicAdd(&gsamp_stamp_count, 1u);
      if (s_ < 65536) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        gsamp_stamps[s_ * 4] = blockIdx.x | ((unsigned long long)__popcll(act) << 32);
        gsamp_stamps[s_ * 4 + 1] = gs_t0;
        gsamp_stamps[s_ * 4 + 2] = __builtin_amdgcn_s_memrealtime();
        gsamp_stamps[s_ * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
      }
    }
  };
#endif
  if (slot >= n_pairs) return;
  // One lane of the quad reads the slot's pair, its mask byte (lane 0) and level `sub`'s reference point (lanes < L), the others get
  // them by quad broadcast: the texture-address path takes one clock per 4 ACTIVE lanes of a load, and this kernel runs at the rate
  // of that path (profiles/r06_experiments.txt sections 2, 5) -- 16 + 16 + 48 lane addresses per wavefront instead of 5 x 64.
  // The reference points are requested WITH the mask byte, not after it (one round trip less in front of phase A; wasted for the
  // masked pairs: 24 bytes each).
  int pair = slot;
  if (order) {
    int p0 = 0;
    if (sub == 0) p0 = order[slot];
    pair = (int)quad_bcast<0>((unsigned)p0);
  }
  float2 mine = float2{0.f, 0.f};
  if (sub < L) mine = *reinterpret_cast<const float2*>(r + ((long)pair * L + sub) * 2);
  if (pair_mask) {
    unsigned mk = 0;
    if (sub == 0) mk = pair_mask[pair];
    if (!quad_bcast<0>(mk)) {
      *reinterpret_cast<uint4*>(samp + (long)pair * 256 + m * 32 + sub * 8) = uint4{0u, 0u, 0u, 0u};
      return;
    }
  }
  float acc[8];
  gsamp_unit<L, PIPE>(vp, G, xw, r, lv, &scratch[wave][pl][0], pair, m, sub, Lq, S, B, acc, mine);
  store_acc<bf16_t, 8>(samp + (long)pair * 256 + m * 32 + sub * 8, acc);
#ifdef GSAMP_STAMPS
  gs_flush(true);
#endif
}

// (hipcc's own allocation is 101 VGPRs = 4 wavefronts per SIMD; a build pinned to 5 per SIMD -- 96 VGPRs, 5 dwords of scratch
// spill -- was 5 % slower: more loads in flight per CU only thrash the L1s.  An LDS window for the coarsest level was built twice
// and lost at 5 views and at 31: profiles/r03_experiments.txt, r05_experiments.txt.)
template <int L, int NT>
__global__ __launch_bounds__(NT) void msda_gsamp_kernel(const bf16_t* __restrict__ vp, const bf16_t* __restrict__ G,
                                                        const float* __restrict__ xw, const float* __restrict__ r,
                                                        LevelTable lv, bf16_t* __restrict__ samp,
                                                        const uint8_t* __restrict__ pair_mask, const int* __restrict__ order,
                                                        int n_pairs, int Lq, int S, int B, int map_ch) {
  msda_gsamp_body<L, NT>(vp, G, xw, r, lv, samp, pair_mask, order, n_pairs, Lq, S, B, map_ch);
}
// PIPE = 1: the gathers double-buffered in half batches (gsamp_dev.h) -- same results bit for bit
template <int L, int NT>
__global__ __launch_bounds__(NT) void msda_gsamp_pipe_kernel(const bf16_t* __restrict__ vp, const bf16_t* __restrict__ G,
                                                             const float* __restrict__ xw, const float* __restrict__ r,
                                                             LevelTable lv, bf16_t* __restrict__ samp,
                                                             const uint8_t* __restrict__ pair_mask, const int* __restrict__ order,
                                                             int n_pairs, int Lq, int S, int B, int map_ch) {
  msda_gsamp_body<L, NT, 1>(vp, G, xw, r, lv, samp, pair_mask, order, n_pairs, Lq, S, B, map_ch);
}

static int g_gsamp_pipe = 0;       // tuning knob "gsamp_pipe": gathers double-buffered in half batches (93 VGPRs) -- 0 = from 32 768 pairs per launch on, 1 = always, 2 = never
int g_auto_small = 1;              // tuning knob "auto_small": small launches pick their own workgroup / tile sizes (see mvg_msda_gsamp)
static int g_gsamp_map = 4;        // tuning knob "gsamp_map": 0 = head per XCD (159 us), n > 0 = chunks of n slot blocks per
                                   // XCD with their 8 heads back to back (1..4: 155 us, 8: 159, 16: 168, 64: 243)
static int g_gfused_chunk = 4;     // tuning knob "gfused_chunk": hp kernel, 0 = one head per XCD (270 us at cfg-2), n > 0 = chunks of n pair blocks per XCD with their 8 heads back to back (1..4: 252 us, 16: 263)
static int g_gsamp_lds_pad = 0;    // probe knob "gsamp_lds_pad": bytes of unused dynamic LDS per workgroup (caps the workgroups per CU)
static int g_gsamp_threads = 256;  // tuning knob "gsamp_threads": workgroup size of msda_gsamp_kernel (256 | 512 | 1024)

template <typename T, int CPL, int NB>
static int launch_msda_fused_cpl(const T* value, const float* oa, const float* r, const LevelTable& lv, T* samp,
                                 int n_pairs, int Lq, int S, hipStream_t st) {
  constexpr int PPW = CPL / 4;
  const int grid = (n_pairs + 4 * PPW - 1) / (4 * PPW);
  switch (lv.L) {
    case 1: hipLaunchKernelGGL((msda_fused_kernel<T, 1, CPL, NB>), dim3(grid), dim3(256), 0, st, value, oa, r, lv, samp, n_pairs, Lq, S); break;
    case 2: hipLaunchKernelGGL((msda_fused_kernel<T, 2, CPL, NB>), dim3(grid), dim3(256), 0, st, value, oa, r, lv, samp, n_pairs, Lq, S); break;
    case 3: hipLaunchKernelGGL((msda_fused_kernel<T, 3, CPL, NB>), dim3(grid), dim3(256), 0, st, value, oa, r, lv, samp, n_pairs, Lq, S); break;
    case 4: hipLaunchKernelGGL((msda_fused_kernel<T, 4, CPL, NB>), dim3(grid), dim3(256), 0, st, value, oa, r, lv, samp, n_pairs, Lq, S); break;
    default: return MVG_E_BADARG;
  }
  MVG_LAUNCH_CHECK();
  return 0;
}

static int launch_msda_fused(const float* value, const float* oa, const float* r, const LevelTable& lv, float* samp,
                             int n_pairs, int Lq, int S, hipStream_t st) {
  if (n_pairs <= 0) return 0;
  return launch_msda_fused_cpl<float, 4, 4>(value, oa, r, lv, samp, n_pairs, Lq, S, st);
}
static int launch_msda_fused(const bf16_t* value, const float* oa, const float* r, const LevelTable& lv, bf16_t* samp,
                             int n_pairs, int Lq, int S, hipStream_t st) {
  if (n_pairs <= 0) return 0;
  return launch_msda_fused_cpl<bf16_t, 8, 4>(value, oa, r, lv, samp, n_pairs, Lq, S, st);
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

int mvg_msda_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const double* sampling_loc, const double* attn_weight, double* out, int N, int S, int M, int D,
                         int L, int Lq, int P, void* stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !out) return MVG_E_BADARG;
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return MVG_E_BADARG;
  if (Lq == 0) return 0;
  const long total = (long)N * Lq * M * D;
  long grid = (total + 255) / 256;
  if (grid > (1L << 22)) grid = 1L << 22;
  hipLaunchKernelGGL(msda_fwd_f64_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, value, spatial_shapes,
                     level_start_index, sampling_loc, attn_weight, out, N, S, M, D, L, Lq, P);
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_msda_backward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                          const double* sampling_loc, const double* attn_weight, const double* grad_output,
                          double* grad_value, double* grad_sampling_loc, double* grad_attn_weight, int N, int S, int M,
                          int D, int L, int Lq, int P, void* stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_loc || !attn_weight || !grad_output ||
      !grad_value || !grad_sampling_loc || !grad_attn_weight)
    return MVG_E_BADARG;
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0) return MVG_E_BADARG;
  if (Lq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(grad_sampling_loc, 0, sizeof(double) * (size_t)N * Lq * M * L * P * 2, st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(grad_attn_weight, 0, sizeof(double) * (size_t)N * Lq * M * L * P, st);
  if (e != hipSuccess) return (int)e;
  const long total = (long)N * Lq * M * D;
  if (total > 0x7fffffffL * 256L) return MVG_E_BADARG;
  hipLaunchKernelGGL(msda_bwd_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, value, spatial_shapes,
                     level_start_index, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                     grad_attn_weight, N, S, M, D, L, Lq, P);
  MVG_LAUNCH_CHECK();
  return 0;
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

int mvg_msda_gsamp(const void* vp, const void* G, const float* xw, const float* ref_lvl, const int64_t* shapes_host,
                   const int64_t* starts_host, void* samp, const uint8_t* pair_mask, const int32_t* order, int N_img,
                   int Lq, int L, int S, int B, void* stream) {
  if (!vp || !G || !xw || !ref_lvl || !shapes_host || !starts_host || !samp || B <= 0) return MVG_E_BADARG;
  LevelTable lv;
  int e = mvg_fill_levels(&lv, shapes_host, starts_host, L);
  if (e) return e;
  const long pairs = (long)N_img * Lq;
  if (pairs > 0x7fffffffL / 4) return MVG_E_BADARG;
  if ((long)N_img * S * 384 >= 0xffffffffL || (long)N_img * 8 * S * 64 >= 0xffffffffL) return MVG_E_BADARG;   // 32-bit byte offsets
  if (pairs == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // Few pairs (a rank's shard of a query-sharded run: 128 of 1024 queries at 8 GPUs): the launch is one partial round of
  // workgroups and its time is a workgroup's latency -- smaller workgroups and single-block XCD chunks spread it over
  // more CUs (cfg-2 with 128 / 256 queries: 0.83 / 0.90 -> 0.80 / 0.88 ms per forward).  Results do not depend on either
  // (every (pair, head) is computed independently).
  // double-buffered gathers: from 32 768 pairs per launch on.  With the planes written just before the launch (just-in-time products:
  // Infinity-Cache hits) the half batches pay at full size -- cfg-2 -0.8 %, all pairs inside -1.6 %, cfg-5 -1.4 % per forward -- and cost
  // a rank's shard +1 % (128 queries); 2 samples per forward +-0.4 % (profiles/r05_experiments.txt section 13; before that schedule:
  // -1.1 % at 31 views, +1 % at 5-10 images)
  const bool pipe = g_gsamp_pipe == 1 || (g_gsamp_pipe == 0 && (long)N_img * Lq >= 32768);
  int nthreads = g_gsamp_threads, map = g_gsamp_map;
  if (g_auto_small && Lq <= 8192) {
    nthreads = 128;
    map = 1;
  }
#define MVG_GS(LL, NT)                                                                                            \
  {                                                                                                               \
    int npb = (int)((pairs + NT / 4 - 1) / (NT / 4));                                                             \
    if (map > 0) npb = (npb + 8 * map - 1) / (8 * map) * (8 * map);                                               \
    if (pipe)                                                                                                 \
      hipLaunchKernelGGL((msda_gsamp_pipe_kernel<LL, NT>), dim3(8 * npb), dim3(NT), g_gsamp_lds_pad, st, (const bf16_t*)vp,     \
                         (const bf16_t*)G, xw, ref_lvl, lv, (bf16_t*)samp, pair_mask, order, (int)pairs, Lq, S,   \
                         B, map);                                                                                 \
    else                                                                                                          \
      hipLaunchKernelGGL((msda_gsamp_kernel<LL, NT>), dim3(8 * npb), dim3(NT), g_gsamp_lds_pad, st, (const bf16_t*)vp,          \
                         (const bf16_t*)G, xw, ref_lvl, lv, (bf16_t*)samp, pair_mask, order, (int)pairs, Lq, S,   \
                         B, map);                                                                                 \
  }
#define MVG_GSN(LL)                                                                                               \
  if (nthreads == 1024) MVG_GS(LL, 1024) else if (nthreads == 512) MVG_GS(LL, 512)                                \
  else if (nthreads == 128) MVG_GS(LL, 128) else MVG_GS(LL, 256)
  switch (L) {
    case 1: MVG_GSN(1); break;
    case 2: MVG_GSN(2); break;
    case 3: MVG_GSN(3); break;
    case 4: MVG_GSN(4); break;
    default: return MVG_E_BADARG;
  }
#undef MVG_GSN
#undef MVG_GS
  MVG_LAUNCH_CHECK();
  return 0;
}

extern int g_chain_rm;
extern int g_chain_a_lds_pad;
extern int g_linear_tiles;
extern int g_f32_split;
extern int g_wreg_grid;
extern int g_f32h_rows;
extern int g_bin_multi;

int mvg_set_tuning(const char* key, int value) {
  if (!key) return MVG_E_BADARG;
  if (!strcmp(key, "linear_tiles") && value >= 0 && value <= 2) { g_linear_tiles = value; return 0; }
  if (!strcmp(key, "f32_split") && (value == 0 || value == 1)) { g_f32_split = value; return 0; }
  if (!strcmp(key, "f32h_rows") && (value == 0 || value == 32 || value == 64)) { g_f32h_rows = value; return 0; }
  if (!strcmp(key, "chain_rm") && (value == 64 || value == 128 || value == 256)) { g_chain_rm = value; return 0; }
  if (!strcmp(key, "wreg_grid") && value > 0) { g_wreg_grid = value; return 0; }
  if (!strcmp(key, "bin_multi") && (value == 0 || value == 1)) { g_bin_multi = value; return 0; }
  if (!strcmp(key, "auto_small") && (value == 0 || value == 1)) { g_auto_small = value; return 0; }
  if (!strcmp(key, "gsamp_map") && value >= 0 && value <= 4096) { g_gsamp_map = value; return 0; }
  if (!strcmp(key, "fwd_map") && (value >= 0 && value <= 2)) { g_fwd_map = value; return 0; }
  if (!strcmp(key, "gfused_chunk") && value >= 0 && value <= 4096) { g_gfused_chunk = value; return 0; }
  if (!strcmp(key, "gsamp_pipe") && value >= 0 && value <= 2) { g_gsamp_pipe = value; return 0; }
  if (!strcmp(key, "chain_a_lds_pad") && value >= 0 && value <= 120 * 1024) { g_chain_a_lds_pad = value; return 0; }
  if (!strcmp(key, "gsamp_lds_pad") && value >= 0 && value <= 120 * 1024) { g_gsamp_lds_pad = value; return 0; }
  if (!strcmp(key, "gsamp_threads") && (value == 128 || value == 256 || value == 512 || value == 1024)) { g_gsamp_threads = value; return 0; }
  return MVG_E_BADARG;
}

int mvg_msda_gfused_f32(const float* value, const float* G, const float* xw, const float* ref_lvl, const int64_t* shapes_host,
                        const int64_t* starts_host, float* samp, const uint8_t* pair_mask, const int32_t* order, int N_img,
                        int Lq, int L, int S, int B, void* stream) {
  if (!value || !G || !xw || !ref_lvl || !shapes_host || !starts_host || !samp || B <= 0) return MVG_E_BADARG;
  LevelTable lv;
  int e = mvg_fill_levels(&lv, shapes_host, starts_host, L);
  if (e) return e;
  const long pairs = (long)N_img * Lq;
  if (pairs > 0x7fffffffL / 4 || (long)N_img * S * 256 > 0x7fffffffL) return MVG_E_BADARG;     // int pixel offsets x C
  if (pairs == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  {
    int npb = (int)((pairs + 31) / 32);
    const int mc = g_gfused_chunk;
    if (mc > 0) npb = (npb + 8 * mc - 1) / (8 * mc) * (8 * mc);
    const int grid_hp = 8 * npb;
    switch (L) {
      case 1: hipLaunchKernelGGL((msda_gfused_f32_hp_kernel<1>), dim3(grid_hp), dim3(256), 0, st, value, G, xw, ref_lvl, lv, samp, pair_mask, order, (int)pairs, Lq, S, B, mc); break;
      case 2: hipLaunchKernelGGL((msda_gfused_f32_hp_kernel<2>), dim3(grid_hp), dim3(256), 0, st, value, G, xw, ref_lvl, lv, samp, pair_mask, order, (int)pairs, Lq, S, B, mc); break;
      case 3: hipLaunchKernelGGL((msda_gfused_f32_hp_kernel<3>), dim3(grid_hp), dim3(256), 0, st, value, G, xw, ref_lvl, lv, samp, pair_mask, order, (int)pairs, Lq, S, B, mc); break;
      case 4: hipLaunchKernelGGL((msda_gfused_f32_hp_kernel<4>), dim3(grid_hp), dim3(256), 0, st, value, G, xw, ref_lvl, lv, samp, pair_mask, order, (int)pairs, Lq, S, B, mc); break;
      default: return MVG_E_BADARG;
    }
    MVG_LAUNCH_CHECK();
    return 0;
  }
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
    return launch_msda_fused((const float*)value, oa, r, lv, (float*)samp, (int)pairs, Lq, S, (hipStream_t)stream);
  if (dtype == MVG_BF16)
    return launch_msda_fused((const bf16_t*)value, oa, r, lv, (bf16_t*)samp, (int)pairs, Lq, S, (hipStream_t)stream);
  return MVG_E_BADARG;
}

}  // extern "C"
