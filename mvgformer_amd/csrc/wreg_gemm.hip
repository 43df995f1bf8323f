// Weight-stationary bf16 GEMMs of the ProjAttn front end (gfx950).
//
//   value projection : vp   = pairs_layout(feat @ Wv^T + bv)              (projattn.py:160-175)
//   offsets / logits : oa   = bilinear(feat, ref_lvl) @ Woa^T + xw        (projattn.py:134-181)
//        with xw[b,q,:] = (tgt+query_pos)[b,q] @ Woa^T + boa precomputed once per layer (the query
//        term of "(ref-point features + query) @ W" does not depend on the view or the level).
//
// Why weight-stationary: with K = 256 these GEMMs move ~1 KB per row and are bound by the ~6 TB/s a
// 256-CU chip pulls through its L1s; a tiled GEMM re-fetches its 64-128 KB weight tile for every row
// tile, which at 200 000+ rows is as much traffic as the activations themselves (measured:
// 107 us for the value projection = 510 MB at 4.8 TB/s, 200 MB of it weights).  Here a persistent
// workgroup loads its weight ONCE into registers (each of the 4 wavefronts keeps its 64 output
// columns x 256 k = 32 KB as 128 VGPRs, in MFMA-fragment order) and streams 64-row tiles through LDS:
// the weight traffic drops to 128 KB per workgroup and the k-loop is pure ds_read_b128 + MFMA.
// Two workgroups per CU overlap one tile's loads with the other's MFMAs.
//
// The A-tile loader is either a plain copy (value projection) or the reference-point bilinear gather
// (grid_sample, zeros padding, align_corners=False -- the same arithmetic as gather_ref_kernel), so
// the (rows x 256) "ref-point feature" tensor is never written to HBM.
#include "common.h"

namespace {

constexpr int ACT_PITCH = 528;    // bytes per bf16 activation row in LDS (256 bf16 + 16 pad)

struct WregParams {
  const bf16_t* A;        // plain mode: (M, 256) bf16 rows;  gather mode: feat (n_img, S, 256)
  const bf16_t* Wf;       // swizzled weight fragments [wn 4][ks 16][j 2][lane 64][8] (N padded to 256 with zeros)
  const float* bias;      // (256) f32 (padded); gather mode: unused (bias is inside xw)
  void* out;              // pairs mode: vp bf16;  gather mode: oa f32 (M, N)
  int M, N, S_img;
  // gather mode
  const float* ref_lvl;   // (pairs, L, 2)
  const float* xw;        // (B*Lq, N) f32
  LevelTable lv;
  int Lq, B;
  int rowmajor;           // plain mode: 0 = pixel-pair layout (N = 256), 1 = row-major bf16 (M, N)
};

__device__ __forceinline__ uint4 blend_bf16x8(const uint4& c00, const uint4& c10, const uint4& c01, const uint4& c11,
                                              float w00, float w10, float w01, float w11) {
  const unsigned a[4] = {c00.x, c00.y, c00.z, c00.w}, b[4] = {c10.x, c10.y, c10.z, c10.w};
  const unsigned c[4] = {c01.x, c01.y, c01.z, c01.w}, d[4] = {c11.x, c11.y, c11.z, c11.w};
  uint4 o;
  unsigned* op = reinterpret_cast<unsigned*>(&o);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float lo = w00 * __uint_as_float(a[t] << 16) + w10 * __uint_as_float(b[t] << 16) +
                     w01 * __uint_as_float(c[t] << 16) + w11 * __uint_as_float(d[t] << 16);
    const float hi = w00 * __uint_as_float(a[t] & 0xffff0000u) + w10 * __uint_as_float(b[t] & 0xffff0000u) +
                     w01 * __uint_as_float(c[t] & 0xffff0000u) + w11 * __uint_as_float(d[t] & 0xffff0000u);
    op[t] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
  }
  return o;
}

template <bool GATHER>
__global__ __launch_bounds__(256, 2) void wreg_gemm_kernel(WregParams p) {
  // rows per tile: 64 for the gather form; 32 for the plain form, whose next tile is prefetched into
  // registers (128 weight + 32 accumulator + 16 prefetch VGPRs fit the 256-register budget of 2 waves/SIMD)
  constexpr int RM = GATHER ? 64 : 32;
  constexpr int MT = RM / 32, NCH = RM * 32 / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;                       // RM x 256 bf16 A tile
  char* stage = smem + RM * ACT_PITCH;    // epilogue staging (pairs: RM x 528 B; gather: 32 x (N*4+16) B)
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, rl = lane & 31, h = lane >> 5;
  const bool wave_has_cols = wn * 64 < p.N;

  // ---- the weight slice of this wavefront -> registers, once
  f32x4 wreg[16][2];
  {
    const bf16_t* wp = p.Wf + (long)wn * 16 * 1024 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      wreg[ks][0] = *reinterpret_cast<const f32x4*>(wp + ks * 1024);
      wreg[ks][1] = *reinterpret_cast<const f32x4*>(wp + ks * 1024 + 512);
    }
  }

  const int ntiles = (p.M + RM - 1) / RM;
  uint4 xpre[GATHER ? 1 : NCH];            // plain mode: next A tile, in flight across the MFMAs / epilogue
  auto prefetch = [&](int tile) {
    if constexpr (!GATHER) {
      const int rr = min(tile, ntiles - 1) * RM;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = i * 256 + tid, row = c >> 5, v16 = c & 31;
        xpre[i] = *reinterpret_cast<const uint4*>(p.A + (long)min(rr + row, p.M - 1) * 256 + v16 * 8);
      }
    }
  };
  prefetch(blockIdx.x);
#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * RM;
    // ---------------- A tile -> LDS
    if constexpr (!GATHER) {
      // the tile was prefetched into registers during the previous iteration (or before the loop)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = i * 256 + tid, row = c >> 5, v16 = c & 31;
        *reinterpret_cast<uint4*>(act + row * ACT_PITCH + v16 * 16) = xpre[i];
      }
    } else {
      // thread -> (row = i*8 + tid/32, 16-byte chunk = tid%32); the 4 corner pointers / weights of a row are
      // recomputed per chunk (cheap next to the 4 gathers)
      const int L = p.lv.L;
#pragma unroll 2
      for (int i = 0; i < 8; ++i) {
        const int row = i * 8 + (tid >> 5), v16 = tid & 31;
        const int grow = min(r0 + row, p.M - 1);
        const int pair = grow / L, l = grow - pair * L;
        const int n = pair / p.Lq;
        const int H = p.lv.H[l], W = p.lv.W[l];
        const float Wf_ = (float)W, Hf_ = (float)H;
        const float refx = p.ref_lvl[(long)grow * 2], refy = p.ref_lvl[(long)grow * 2 + 1];
        const float gx = fminf(fmaxf(refx * 2.f - 1.f, -1.1f), 1.1f);                 // projattn.py:134
        const float gy = fminf(fmaxf(refy * 2.f - 1.f, -1.1f), 1.1f);
        const float ix = ((gx + 1.f) * Wf_ - 1.f) * 0.5f, iy = ((gy + 1.f) * Hf_ - 1.f) * 0.5f;
        const float x0f = floorf(ix), y0f = floorf(iy);
        const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        const float tx = ix - x0f, ty = iy - y0f;
        const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
        const float w00 = (x0ok && y0ok) ? (1.f - tx) * (1.f - ty) : 0.f, w10 = (x1ok && y0ok) ? tx * (1.f - ty) : 0.f;
        const float w01 = (x0ok && y1ok) ? (1.f - tx) * ty : 0.f, w11 = (x1ok && y1ok) ? tx * ty : 0.f;
        const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x1, 0), W - 1);
        const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y1, 0), H - 1);
        const bf16_t* fb = p.A + ((long)n * p.S_img + p.lv.start[l]) * 256 + v16 * 8;
        const uint4 c00 = *reinterpret_cast<const uint4*>(fb + (long)(y0c * W + x0c) * 256);
        const uint4 c10 = *reinterpret_cast<const uint4*>(fb + (long)(y0c * W + x1c) * 256);
        const uint4 c01 = *reinterpret_cast<const uint4*>(fb + (long)(y1c * W + x0c) * 256);
        const uint4 c11 = *reinterpret_cast<const uint4*>(fb + (long)(y1c * W + x1c) * 256);
        *reinterpret_cast<uint4*>(act + row * ACT_PITCH + v16 * 16) = blend_bf16x8(c00, c10, c01, c11, w00, w10, w01, w11);
      }
    }
    __syncthreads();
    prefetch(tile + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- MFMA: acc[mt][j] = A_tile(64 x 256) . W_slice(64 cols)^T, weights from registers
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mt][j][e] = 0.f;
    if (wave_has_cols) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        f32x4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          a[mt] = *reinterpret_cast<const f32x4*>(act + (mt * 32 + rl) * ACT_PITCH + ks * 32 + 16 * h);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[ks][j]),
                                                                __builtin_bit_cast(bf16x8, a[mt]), acc[mt][j], 0, 0, 0);
      }
    }

    // ---------------- epilogue
    if constexpr (!GATHER) {
      // + bias -> bf16 -> staging tile -> pixel-pair layout (see gemm.hip OUT_PAIRS)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nn = wn * 64 + j * 32 + 8 * g + 4 * h;
          const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nn) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(acc[mt][j][4 * g] + bv[0]) | ((unsigned)f32_to_bf16(acc[mt][j][4 * g + 1] + bv[1]) << 16);
            pk.y = (unsigned)f32_to_bf16(acc[mt][j][4 * g + 2] + bv[2]) | ((unsigned)f32_to_bf16(acc[mt][j][4 * g + 3] + bv[3]) << 16);
            *reinterpret_cast<uint2*>(stage + (mt * 32 + rl) * ACT_PITCH + nn * 2) = pk;
          }
        }
      __syncthreads();
      bf16_t* vp = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = i * 256 + tid, row = c >> 5, v16 = c & 31;
        const int grow = r0 + row;
        if (grow < p.M && p.rowmajor) {
          if (v16 * 8 < p.N)
            *reinterpret_cast<f32x4*>(vp + (long)grow * p.N + v16 * 8) =
                *reinterpret_cast<const f32x4*>(stage + row * ACT_PITCH + v16 * 16);
        } else if (grow < p.M) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * ACT_PITCH + v16 * 16);
          const int img = grow / p.S_img, s = grow - img * p.S_img;
          const int head = v16 >> 2, ch8 = v16 & 3;
          bf16_t* base = vp + (((long)img * 8 + head) * (p.S_img + 1) + s) * 64 + ch8 * 16;
          *reinterpret_cast<f32x4*>(base + 64) = v;      // pair 1+s, left corner
          *reinterpret_cast<f32x4*>(base + 8) = v;       // pair s,   right corner
        }
      }
      // the next iteration's loader writes `act` (all waves are past the MFMA loop: barrier above) and its
      // barrier orders the next staging writes after these staging reads
    } else {
      // fp32 out (M, N) + xw[(b, q), :]; two halves of 32 rows through a (N*4+16)-byte-pitch staging tile
      const int SP = p.N * 4 + 16;
      const int vpr = p.N / 4;                 // 16-byte vectors per output row
      float* oa = reinterpret_cast<float*>(p.out);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (wave_has_cols) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int nn = wn * 64 + j * 32 + 8 * g + 4 * h;
              *reinterpret_cast<f32x4*>(stage + rl * SP + nn * 4) =
                  f32x4{acc[mt][j][4 * g], acc[mt][j][4 * g + 1], acc[mt][j][4 * g + 2], acc[mt][j][4 * g + 3]};
            }
        }
        __syncthreads();
        for (int c = tid; c < 32 * vpr; c += 256) {
          const int row = c / vpr, vc = c - row * vpr;
          const int grow = r0 + mt * 32 + row;
          if (grow < p.M) {
            const int pair = grow / p.lv.L;
            const int n = pair / p.Lq, q = pair - n * p.Lq, b = n % p.B;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * SP + vc * 16) +
                            *reinterpret_cast<const f32x4*>(p.xw + ((long)b * p.Lq + q) * p.N + vc * 4);
            *reinterpret_cast<f32x4*>(oa + (long)grow * p.N + vc * 4) = v;
          }
        }
        __syncthreads();
      }
    }
  }
}

template <bool GATHER>
int launch_wreg(const WregParams& p, size_t lds, hipStream_t st) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wreg_gemm_kernel<GATHER>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  const int RMh = GATHER ? 64 : 32;
  const int ntiles = (p.M + RMh - 1) / RMh;
  const int grid = ntiles < 512 ? ntiles : 512;      // persistent: 2 workgroups per CU
  hipLaunchKernelGGL((wreg_gemm_kernel<GATHER>), dim3(grid), dim3(256), lds, st, p);
  MVG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int mvg_value_proj_pairs_ws(const void* feat, const void* Wf, const float* bias, void* vp, int n_img, int S,
                                       void* stream) {
  if (!feat || !Wf || !bias || !vp || n_img <= 0 || S <= 0) return MVG_E_BADARG;
  WregParams p = {};
  p.A = (const bf16_t*)feat; p.Wf = (const bf16_t*)Wf; p.bias = bias; p.out = vp;
  p.M = n_img * S; p.N = 256; p.S_img = S;
  return launch_wreg<false>(p, 2 * 32 * ACT_PITCH, (hipStream_t)stream);
}

extern "C" int mvg_oa_gather_gemm(const void* feat, const float* ref_lvl, const float* xw, const void* Wf,
                                  const int64_t* shapes_host, const int64_t* starts_host, float* oa, int V, int B,
                                  int Lq, int L, int S, int N, void* stream) {
  if (!feat || !ref_lvl || !xw || !Wf || !shapes_host || !starts_host || !oa) return MVG_E_BADARG;
  if (N <= 0 || N > 256 || N % 64 != 0) return MVG_E_BADARG;
  WregParams p = {};
  int e = mvg_fill_levels(&p.lv, shapes_host, starts_host, L);
  if (e) return e;
  p.A = (const bf16_t*)feat; p.Wf = (const bf16_t*)Wf; p.bias = nullptr; p.out = oa;
  const long rows = (long)V * B * Lq * L;
  if (rows > 0x7fffffffL) return MVG_E_BADARG;
  if (rows == 0) return 0;
  p.M = (int)rows; p.N = N; p.S_img = S; p.ref_lvl = ref_lvl; p.xw = xw; p.Lq = Lq; p.B = B;
  return launch_wreg<true>(p, 64 * ACT_PITCH + 32 * (N * 4 + 16), (hipStream_t)stream);
}

// G = feat @ W^T in row-major bf16 (n_img*S, N), N in {64,128,192,256}, no bias: the offsets/logits Linear applied to the
// pyramid itself (bilinear sampling commutes with the Linear, so the fused sampling kernel gathers G at the reference
// point instead of running a (rows x 256) GEMM per (view, query, level)).
extern "C" int mvg_feat_linear_ws(const void* feat, const void* Wf, void* G, int n_img, int S, int N, void* stream) {
  if (!feat || !Wf || !G || n_img <= 0 || S <= 0 || N <= 0 || N > 256 || N % 64 != 0) return MVG_E_BADARG;
  WregParams p = {};
  p.A = (const bf16_t*)feat; p.Wf = (const bf16_t*)Wf; p.bias = nullptr; p.out = G;
  p.M = n_img * S; p.N = N; p.S_img = S; p.rowmajor = 1;
  return launch_wreg<false>(p, 2 * 32 * ACT_PITCH, (hipStream_t)stream);
}
