// Device building blocks of the fp32 fused kernels (csrc/f32s.hip): fp32 storage everywhere, products formed on the fp16 matrix
// pipe from TWO-part operands -- x 2^s = h + l (fp16 each), a * w accumulated in fp32 as the three MFMAs  l*h + h*l + h*h.
// The split is done ONCE per value instead of once per tile that touches it:
//   * weights are split on the host when the operand cache is built (ops.split_swizzle_weight_h2): two fp16 planes, each in the
//     MFMA-fragment order of csrc/chain_dev.h (stage_gemm), streamed from L2 straight into registers;
//   * activations live in LDS as two fp16 planes (row pitch 528 B: conflict-free ds_read_b128), written by the epilogue that
//     produced them (split4_h2) and read by every wavefront of the next stage.
// A wavefront works on ONE 32-column block of the weight (cb) and MT 32-row blocks of the tile, so each fragment is loaded by
// exactly one wavefront of the workgroup; the k-step order is rotated per column block only (never per tile): a row's result does
// not depend on the tile or the position it is computed in.
#pragma once
#include "common.h"

namespace f32s {

constexpr int PLP = 528;      // bytes per 256-column plane row in LDS (512 + 16 pad)

__device__ __forceinline__ const bf16_t* frag_ptr(const bf16_t* __restrict__ Wf, int nb, int cb, int kt_total, int lane) {
  return Wf + ((long)nb * 4 + (cb >> 1)) * kt_total * 1024 + (cb & 1) * 512 + lane * 8;
}

// ---- two-part fp16 operands ("f32h"): x 2^s = h + l, a * w = l*h + h*l + h*h as three fp16 MFMAs (see csrc/f32s.hip)
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

}  // namespace f32s
