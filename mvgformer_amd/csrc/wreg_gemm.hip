// Weight-stationary bf16 GEMMs over the feature pyramid (gfx950): out = feat (n_img*S, 256) @ W^T (+ bias)
//
//   value projection : vp = pixel-pair layout of feat @ Wv^T + bv            (projattn.py:160-175)
//   G projection     : G  = feat @ [Woff; Wattn]^T, row-major (n_img*S, 192) (projattn.py:180-181 applied to the
//                      pyramid itself: bilinear sampling commutes with the Linear, see msda_gsamp_kernel)
//
// Why weight-stationary: with K = 256 these GEMMs move ~1 KB per row and are bound by what a 256-CU chip pulls
// through its L1s (~6 TB/s); a tiled GEMM re-fetches its 64-128 KB weight tile for every row tile, which at
// 200 000 rows is as much traffic as the activations (measured: 107 us for the value projection = 510 MB, 200 MB
// of it weights).  Here 512 persistent workgroups load the weight ONCE into registers -- each of the 4
// wavefronts keeps its 64 output columns x 256 k = 32 KB as 128 VGPRs in MFMA-fragment order -- and stream
// 32-row tiles through LDS: the k-loop is pure ds_read_b128 + v_mfma_f32_32x32x16_bf16, the next tile is
// prefetched into registers across the MFMAs and the epilogue, two workgroups per CU overlap each other.
//
// Pixel-pair layout (consumed by msda_gsamp_kernel):  vp[img][head 8][1+s][ch 32][2] bf16, the 32-bit word of
// channel ch in line 1+s is (value(s)[ch], value(s+1)[ch]); line 0 is (0, value(0)), the right half of the last
// pixel's line is 0.  Tiles overlap by one row (tile t = rows [31t, 31t+32)) so that every tile owns the
// right-hand neighbour of its last output row.
#include "common.h"

namespace {

constexpr int RM = 32;            // rows per tile (one 32x32 MFMA row block)
constexpr int ACT_PITCH = 528;    // bytes per bf16 activation row in LDS (256 bf16 + 16 pad: conflict-free b128)
constexpr int NCH = RM * 32 / 256;

struct WregParams {
  const bf16_t* A;        // (M, 256) bf16 rows
  const bf16_t* Wf;       // swizzled weight fragments [wn 4][ks 16][j 2][lane 64][8] (N padded to 256 with zeros)
  const float* bias;      // (256) f32 or nullptr
  void* out;
  int M, N, S_img;
  int rowmajor;           // 0 = pixel-pair layout (N = 256), 1 = row-major bf16 (M, N)
};

__global__ __launch_bounds__(256, 2) void wreg_gemm_kernel(WregParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;                       // RM x 256 bf16 A tile
  char* stage = smem + RM * ACT_PITCH;    // RM x 256 bf16 output tile
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

  const int TS = p.rowmajor ? RM : RM - 1;      // tile stride in rows (pair layout: one row of overlap)
  const int ntiles = (p.M + TS - 1) / TS;
  uint4 xpre[NCH];                              // next A tile, in flight across the MFMAs / epilogue
  auto prefetch = [&](int tile) {
    const int rr = min(tile, ntiles - 1) * TS;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = i * 256 + tid, row = c >> 5, v16 = c & 31;
      xpre[i] = *reinterpret_cast<const uint4*>(p.A + (long)min(rr + row, p.M - 1) * 256 + v16 * 8);
    }
  };
  prefetch(blockIdx.x);
#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * TS;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = i * 256 + tid, row = c >> 5, v16 = c & 31;
      *reinterpret_cast<uint4*>(act + row * ACT_PITCH + v16 * 16) = xpre[i];
    }
    __syncthreads();
    prefetch(tile + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- MFMA: acc[j] = A_tile(32 x 256) . W_slice(64 cols)^T, weights from registers
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    if (wave_has_cols) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(act + rl * ACT_PITCH + ks * 32 + 16 * h);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[ks][j]),
                                                           __builtin_bit_cast(bf16x8, a), acc[j], 0, 0, 0);
      }
    }

    // ---------------- epilogue: + bias -> bf16 -> staging tile -> coalesced stores
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nn = wn * 64 + j * 32 + 8 * g + 4 * h;
        const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nn) : f32x4{0.f, 0.f, 0.f, 0.f};
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(acc[j][4 * g] + bv[0]) | ((unsigned)f32_to_bf16(acc[j][4 * g + 1] + bv[1]) << 16);
        pk.y = (unsigned)f32_to_bf16(acc[j][4 * g + 2] + bv[2]) | ((unsigned)f32_to_bf16(acc[j][4 * g + 3] + bv[3]) << 16);
        *reinterpret_cast<uint2*>(stage + rl * ACT_PITCH + nn * 2) = pk;
      }
    __syncthreads();
    bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = i * 256 + tid, row = c >> 5, v16 = c & 31;
      const int grow = r0 + row;
      if (grow >= p.M) continue;
      if (p.rowmajor) {
        if (v16 * 8 < p.N)
          *reinterpret_cast<f32x4*>(outp + (long)grow * p.N + v16 * 8) =
              *reinterpret_cast<const f32x4*>(stage + row * ACT_PITCH + v16 * 16);
      } else if (row < RM - 1) {
        // pair line 1+s of (img, head): this thread writes its 8 channels = 32 contiguous bytes
        const int img = grow / p.S_img, s = grow - img * p.S_img;
        const int head = v16 >> 2, ch8 = v16 & 3;
        const uint4 lf = *reinterpret_cast<const uint4*>(stage + row * ACT_PITCH + v16 * 16);
        uint4 rt = *reinterpret_cast<const uint4*>(stage + (row + 1) * ACT_PITCH + v16 * 16);
        if (s + 1 >= p.S_img) rt = uint4{0u, 0u, 0u, 0u};         // no right neighbour across an image boundary
        const unsigned l4[4] = {lf.x, lf.y, lf.z, lf.w}, r4[4] = {rt.x, rt.y, rt.z, rt.w};
        unsigned w8[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          w8[2 * t] = __builtin_amdgcn_perm(r4[t], l4[t], 0x05040100u);       // (left.ch 2t  , right.ch 2t  )
          w8[2 * t + 1] = __builtin_amdgcn_perm(r4[t], l4[t], 0x07060302u);   // (left.ch 2t+1, right.ch 2t+1)
        }
        bf16_t* line = outp + (((long)img * 8 + head) * (p.S_img + 1) + 1 + s) * 64 + ch8 * 16;
        *reinterpret_cast<uint4*>(line) = uint4{w8[0], w8[1], w8[2], w8[3]};
        *reinterpret_cast<uint4*>(line + 8) = uint4{w8[4], w8[5], w8[6], w8[7]};
        if (s == 0) {                                  // line 0 of the plane: (0, value(0)) -- the w_low = -1 column
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            w8[2 * t] = l4[t] << 16;
            w8[2 * t + 1] = l4[t] & 0xffff0000u;
          }
          *reinterpret_cast<uint4*>(line - 64) = uint4{w8[0], w8[1], w8[2], w8[3]};
          *reinterpret_cast<uint4*>(line - 64 + 8) = uint4{w8[4], w8[5], w8[6], w8[7]};
        }
      }
    }
    // next iteration: the act writes are ordered after this tile's MFMA reads by the barrier above, the next
    // staging writes after these staging reads by the barrier that follows the act writes
  }
}

int launch_wreg(const WregParams& p, hipStream_t st) {
  const size_t lds = 2 * RM * ACT_PITCH;
  const int TS = p.rowmajor ? RM : RM - 1;
  const int ntiles = (p.M + TS - 1) / TS;
  const int grid = ntiles < 512 ? ntiles : 512;      // persistent: 2 workgroups per CU
  hipLaunchKernelGGL(wreg_gemm_kernel, dim3(grid), dim3(256), lds, st, p);
  MVG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int mvg_value_proj_pairs_ws(const void* feat, const void* Wf, const float* bias, void* vp, int n_img, int S,
                                       void* stream) {
  if (!feat || !Wf || !bias || !vp || n_img <= 0 || S <= 0) return MVG_E_BADARG;
  WregParams p = {};
  p.A = (const bf16_t*)feat; p.Wf = (const bf16_t*)Wf; p.bias = bias; p.out = vp;
  p.M = n_img * S; p.N = 256; p.S_img = S; p.rowmajor = 0;
  return launch_wreg(p, (hipStream_t)stream);
}

extern "C" int mvg_feat_linear_ws(const void* feat, const void* Wf, void* G, int n_img, int S, int N, void* stream) {
  if (!feat || !Wf || !G || n_img <= 0 || S <= 0 || N <= 0 || N > 256 || N % 64 != 0) return MVG_E_BADARG;
  WregParams p = {};
  p.A = (const bf16_t*)feat; p.Wf = (const bf16_t*)Wf; p.bias = nullptr; p.out = G;
  p.M = n_img * S; p.N = N; p.S_img = S; p.rowmajor = 1;
  return launch_wreg(p, (hipStream_t)stream);
}
