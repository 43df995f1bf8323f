// Device building blocks of the fp32 ("f32s") fused kernels: fp32 storage everywhere, products formed on the bf16 matrix
// pipe from 3-way split operands (csrc/gemm.hip, "split form"): x = h + m + l exactly (bf16 each, round-to-nearest at each
// step), a * w accumulated in fp32 as the six MFMAs  l*h + h*l + m*m + m*h + h*m + h*h  (what is dropped is below
// 2^-25 |a w|).  Here the split is done ONCE per value instead of once per tile that touches it:
//   * weights are split on the host when the operand cache is built (ops.split_swizzle_weight): three bf16 planes, each
//     in the MFMA-fragment order of csrc/chain_dev.h (stage_gemm), streamed from L2 straight into registers;
//   * activations live in LDS as three bf16 planes (row pitch 528 B: conflict-free ds_read_b128), written by the
//     epilogue that produced them (split4) and read by every wavefront of the next stage.
// A stage  acc[m][n] = sum_k act[m][k] W[n][k]  costs 6 MFMAs of 32 matrix-pipe cycles per (32 x 32 block, 16 k) against
// 3 fragment loads (1 KB each) + 3 ds_read_b128 per row block: the loop is matrix-pipe bound by construction.
// A wavefront works on ONE 32-column block of the weight (cb) and MT 32-row blocks of the tile, so each fragment is
// loaded by exactly one wavefront of the workgroup; the k-step order is rotated per column block only (never per tile):
// a row's result does not depend on the tile or the position it is computed in.
#pragma once
#include "common.h"

// Measurement hook (tools/ab_f32s.sh builds variants; the product is built with 0): bit 0 = no ring refills (every k-step reuses
// the first RING fragments), bit 1 = no LDS reads in the k loop, bit 2 = no MFMAs.  Results are wrong with any bit set.
#ifndef F32S_KO
#define F32S_KO 0
#endif

namespace f32s {

constexpr int PLP = 528;      // bytes per 256-column plane row in LDS (512 + 16 pad)
constexpr int PLP128 = 272;   // bytes per 128-column plane row (FFN hidden chunk)

// 4 fp32 -> (h, m, l) as three 8-byte packs of 4 bf16
__device__ __forceinline__ void split4(const f32x4& x, uint2 (&p)[3]) {
  f32x4 r = x;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    p[s].x = pack_bf16(r[0], r[1]);
    p[s].y = pack_bf16(r[2], r[3]);
    if (s < 2) {
      r[0] -= __uint_as_float(p[s].x << 16);
      r[1] -= __uint_as_float(p[s].x & 0xffff0000u);
      r[2] -= __uint_as_float(p[s].y << 16);
      r[3] -= __uint_as_float(p[s].y & 0xffff0000u);
    }
  }
}

// the fp32 value of 4 split elements (exact: h + m + l is the value that was split)
__device__ __forceinline__ f32x4 join4(const uint2& h, const uint2& m, const uint2& l) {
  f32x4 v;
  v[0] = (__uint_as_float(h.x << 16) + __uint_as_float(m.x << 16)) + __uint_as_float(l.x << 16);
  v[1] = (__uint_as_float(h.x & 0xffff0000u) + __uint_as_float(m.x & 0xffff0000u)) + __uint_as_float(l.x & 0xffff0000u);
  v[2] = (__uint_as_float(h.y << 16) + __uint_as_float(m.y << 16)) + __uint_as_float(l.y << 16);
  v[3] = (__uint_as_float(h.y & 0xffff0000u) + __uint_as_float(m.y & 0xffff0000u)) + __uint_as_float(l.y & 0xffff0000u);
  return v;
}

// write 4 consecutive fp32 columns of one row into the three planes (plane p at act + p * plane_bytes)
template <int PITCH>
__device__ __forceinline__ void store_split4(char* __restrict__ act, int plane_bytes, int row, int col, const f32x4& x) {
  uint2 p[3];
  split4(x, p);
#pragma unroll
  for (int s = 0; s < 3; ++s) *reinterpret_cast<uint2*>(act + s * plane_bytes + row * PITCH + col * 2) = p[s];
}

// fragment address of (plane-relative) weight Wf[nb][wn 4][ks KT][j 2][lane 64][8]: column block cb (0..7) of 256-row block nb
__device__ __forceinline__ const bf16_t* frag_ptr(const bf16_t* __restrict__ Wf, int nb, int cb, int kt_total, int lane) {
  return Wf + ((long)nb * 4 + (cb >> 1)) * kt_total * 1024 + (cb & 1) * 512 + lane * 8;
}

// First RING k-steps' fragments of a stage (3 planes each), requested by the caller ahead of the barrier / epilogue in front
// of the stage (stage with PRE = true starts on them).
template <int KSTEPS, int RING>
__device__ __forceinline__ void ring_prefetch(const bf16_t* __restrict__ wp, long wplane, f32x4 (&ring)[RING][3], int rot) {
#pragma unroll
  for (int p = 0; p < RING; ++p) {
    const int kq = (p + rot) & (KSTEPS - 1);
#pragma unroll
    for (int s = 0; s < 3; ++s) ring[p][s] = *reinterpret_cast<const f32x4*>(wp + s * wplane + kq * 1024);
  }
}

// Stage GEMM for one wavefront: rows [row0, row0 + 32 MT) of the activation planes x the 32 weight rows behind `wp`
// (frag_ptr of plane 0; plane s at wp + s * wplane elements; k-steps are 1024 elements apart).  acc[mt][4 g + t] =
// out[row0 + 32 mt + rl][8 g + 4 h + t] of the column block, lane = 32 h + rl.  ALT (default for MT = 1) alternates two accumulators
// (acc2) so that consecutive MFMAs never depend on each other; the caller adds them.  ALT = false with MT = 1 sums a row exactly as an
// MT = 2 stage does (one accumulator per row block, the six products in order): the 32-row and 64-row tiles of chain B agree bit for bit.
template <int MT, int KSTEPS, int PITCH, int RING = 4, bool PRE = false, bool ALT = (MT == 1)>
__device__ __forceinline__ void stage(const char* __restrict__ act, int plane_bytes, int row0, const bf16_t* __restrict__ wp,
                                      long wplane, f32x16 (&acc)[MT], f32x16* acc2, bool zero, int rot, int lane,
                                      f32x4 (*pre)[3] = nullptr, int prio_half = -1) {
  static_assert((KSTEPS & (KSTEPS - 1)) == 0, "KSTEPS must be a power of two");
  const int rl = lane & 31, h = lane >> 5;
  if (zero) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
    if (ALT) {
#pragma unroll
      for (int e = 0; e < 16; ++e) (*acc2)[e] = 0.f;
    }
  }
  f32x4 ring[RING][3];
#pragma unroll
  for (int p = 0; p < RING; ++p) {
    const int kq = (p + rot) & (KSTEPS - 1);
#pragma unroll
    for (int s = 0; s < 3; ++s) ring[p][s] = PRE ? pre[p][s] : *reinterpret_cast<const f32x4*>(wp + s * wplane + kq * 1024);
  }
  const char* arow = act + (row0 + rl) * PITCH + 16 * h;
  f32x4 a_nxt[MT][3];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < 3; ++s)
      a_nxt[mt][s] = *reinterpret_cast<const f32x4*>(arow + s * plane_bytes + mt * 32 * PITCH + (rot & (KSTEPS - 1)) * 32);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    // prio_half = 0 | 1 (the wavefront's half of an 8-wavefront workgroup: the two wavefronts of a SIMD differ in it): the two
    // take turns at the higher issue priority, k-step by k-step.  With equal priorities the older wavefront of a SIMD is served
    // first throughout, finishes its stage early and leaves the younger one to run the rest alone, uncovered (s_memtime: the
    // barrier behind a stage waited 4 k cycles for it).
    if (prio_half >= 0) {
      if ((ks & 1) == prio_half) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    bf16x8 a[MT][3], b[3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 3; ++s) a[mt][s] = __builtin_bit_cast(bf16x8, a_nxt[mt][s]);
    if (ks + 1 < KSTEPS && !(F32S_KO & 2)) {
      const int kn = (ks + 1 + rot) & (KSTEPS - 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 3; ++s)
          a_nxt[mt][s] = *reinterpret_cast<const f32x4*>(arow + s * plane_bytes + mt * 32 * PITCH + kn * 32);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) b[s] = __builtin_bit_cast(bf16x8, ring[ks % RING][s]);
    if (ks + RING < KSTEPS && !(F32S_KO & 1)) {
      const int kq = (ks + RING + rot) & (KSTEPS - 1);
#pragma unroll
      for (int s = 0; s < 3; ++s) ring[ks % RING][s] = *reinterpret_cast<const f32x4*>(wp + s * wplane + kq * 1024);
    }
    // smallest terms first; (plane of W, plane of A)
    constexpr int TB[6] = {2, 0, 1, 1, 0, 0}, TA[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (F32S_KO & 4) {
          acc[mt][t] += __builtin_bit_cast(f32x4, b[TB[t]])[0] * __builtin_bit_cast(f32x4, a[mt][TA[t]])[1];
          continue;
        }
        if (ALT && (t & 1))
          *acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[TB[t]], a[mt][TA[t]], *acc2, 0, 0, 0);
        else
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[TB[t]], a[mt][TA[t]], acc[mt], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);     // keep the ring refills a full RING of k-steps ahead of their MFMAs (chain_dev.h)
  }
  if (prio_half >= 0) __builtin_amdgcn_s_setprio(0);
}

// ---- two-part fp16 operands ("f32h"): x 2^s = h + l, a * w = l*h + h*l + h*h as three fp16 MFMAs (see csrc/f32s.hip, pyramid_f32h_kernel)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// power-of-two scale of a row whose largest |entry| is m: m 2^s lies in [2^13, 2^14) (rows of zeros / subnormals: capped)
__device__ __forceinline__ int row_scale(float m) {
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
  return min(13 - e, 100);
}

// 4 fp32, already scaled -> (h, l) as two 8-byte packs of 4 fp16
__device__ __forceinline__ void split4_h2(const f32x4& xs, uint2& ph, uint2& pl) {
  const half2_t h0 = {(_Float16)xs[0], (_Float16)xs[1]}, h1 = {(_Float16)xs[2], (_Float16)xs[3]};
  const half2_t l0 = {(_Float16)(xs[0] - (float)h0[0]), (_Float16)(xs[1] - (float)h0[1])};
  const half2_t l1 = {(_Float16)(xs[2] - (float)h1[0]), (_Float16)(xs[3] - (float)h1[1])};
  ph = uint2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
  pl = uint2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
}

// stage() on two-part fp16 planes (plane 1 = l at act + plane_bytes / wp + wplane): acc[mt][4 g + t] = 2^(s_row + s_w) out[row0 + 32 mt + rl][8 g + 4 h + t]
template <int MT, int KSTEPS, int PITCH, int RING = 4>
__device__ __forceinline__ void stage_h2(const char* __restrict__ act, int plane_bytes, int row0, const bf16_t* __restrict__ wp, long wplane,
                                         f32x16 (&acc)[MT], int rot, int lane) {
  static_assert((KSTEPS & (KSTEPS - 1)) == 0, "KSTEPS must be a power of two");
  const int rl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
  f32x4 ring[RING][2];
#pragma unroll
  for (int p = 0; p < RING; ++p) {
    const int kq = (p + rot) & (KSTEPS - 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) ring[p][s] = *reinterpret_cast<const f32x4*>(wp + s * wplane + kq * 1024);
  }
  const char* arow = act + (row0 + rl) * PITCH + 16 * h;
  f32x4 a_nxt[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < 2; ++s) a_nxt[mt][s] = *reinterpret_cast<const f32x4*>(arow + s * plane_bytes + mt * 32 * PITCH + (rot & (KSTEPS - 1)) * 32);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    half8 a[MT][2], b[2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 2; ++s) a[mt][s] = __builtin_bit_cast(half8, a_nxt[mt][s]);
    if (ks + 1 < KSTEPS) {
      const int kn = (ks + 1 + rot) & (KSTEPS - 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s) a_nxt[mt][s] = *reinterpret_cast<const f32x4*>(arow + s * plane_bytes + mt * 32 * PITCH + kn * 32);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) b[s] = __builtin_bit_cast(half8, ring[ks % RING][s]);
    if (ks + RING < KSTEPS) {
      const int kq = (ks + RING + rot) & (KSTEPS - 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) ring[ks % RING][s] = *reinterpret_cast<const f32x4*>(wp + s * wplane + kq * 1024);
    }
    constexpr int TB[3] = {0, 1, 0}, TA[3] = {1, 0, 0};        // (plane of W, plane of A): l*h, h*l, h*h
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[TB[t]], a[mt][TA[t]], acc[mt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// bias of the wavefront's column block: bv[g] = bias[8 g + 4 h .. + 3]
__device__ __forceinline__ void load_bias(const float* __restrict__ bias_cb, f32x4 (&bv)[4], int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(bias_cb + 8 * g + 4 * h);
}

// acc (+bias, relu, keep) -> the three planes, columns [col0, col0 + 32) of rows [row0, row0 + 32 MT)
template <int MT, int PITCH>
__device__ __forceinline__ void write_planes(char* __restrict__ act, int plane_bytes, int row0, int col0, const f32x16 (&acc)[MT],
                                             const f32x4 (&bv)[4], bool relu, const bool (&keep)[MT], int lane) {
  const int rl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 v;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float x = acc[mt][4 * g + t] + bv[g][t];
        v[t] = keep[mt] ? (relu ? fmaxf(x, 0.f) : x) : 0.f;
      }
      store_split4<PITCH>(act, plane_bytes, row0 + mt * 32 + rl, col0 + 8 * g + 4 * h, v);
    }
}

// write_planes in two halves: the arithmetic (bias, ReLU, keep mask, split) into registers BEFORE the barrier that frees the
// planes -- it then runs while the slower wavefront of the SIMD is still in its k loop -- and the LDS stores after it.
template <int MT>
__device__ __forceinline__ void split_planes(uint2 (&pk)[MT][4][3], const f32x16 (&acc)[MT], const f32x4 (&bv)[4], bool relu,
                                             const bool (&keep)[MT]) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 v;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float x = acc[mt][4 * g + t] + bv[g][t];
        v[t] = keep[mt] ? (relu ? fmaxf(x, 0.f) : x) : 0.f;
      }
      split4(v, pk[mt][g]);
#pragma unroll
      for (int s = 0; s < 3; ++s) asm volatile("" : "+v"(pk[mt][g][s].x), "+v"(pk[mt][g][s].y));     // computed HERE, not sunk to the stores
    }
}

template <int MT, int PITCH>
__device__ __forceinline__ void store_planes(char* __restrict__ act, int plane_bytes, int row0, int col0, const uint2 (&pk)[MT][4][3],
                                             int lane) {
  const int rl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        *reinterpret_cast<uint2*>(act + s * plane_bytes + (row0 + mt * 32 + rl) * PITCH + (col0 + 8 * g + 4 * h) * 2) = pk[mt][g][s];
}

}  // namespace f32s
