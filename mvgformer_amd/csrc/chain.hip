// Fused per-token Linear chains of the decoder layer (bf16 MFMA, fp32 accumulate), gfx950.
//
// With K = 256 the dense projections of this model sit at ~128 FLOP/byte in bf16 -- below the
// MI355X ridge (2.5 PF / 8 TB/s = 312 FLOP/B): as separate GEMM launches they are bound by
// writing and re-reading (rows x 256) activations.  Here a workgroup keeps a tile of rows
// resident in LDS and walks it through a whole chain of Linears; only the weights (128 KB per
// 256x256 layer, L2-resident) are streamed.
//
// Weight operands of the chains are in the fragment order documented at stage_gemm (host helper:
// mvgformer_amd.ops.swizzle_weight).
//
//   chain A (per (image, query) row; dq_decoder.py:585-588 + 659-690):
//       attn = inside * (samp @ Wp^T + bp)            -> stored (needed for the view mean)
//       o    = W2 relu(W1 relu(W0 attn + b0) + b1) + b2   (dx, dy, confidence logit)
//   chain B (per joint token; dq_decoder.py:770-778, mvp_decoder.py:94-98, dq_decoder.py:889-893):
//       t1   = LN2(tgt + Wu mean_v(attn_v) + bu)
//       tgt' = LN3(t1 + W2f relu(W1f t1 + b1f) + b2f)
//       prob = mean_j sigmoid(Wc tgt' + bc), valid = prob[1] > thr
//
// Stage GEMM: 4 wavefronts, wave w owns output columns [64w, 64w+64) for all RM rows; the
// activation tile (RM x 256 bf16, 528-byte pitch -> conflict-free ds_read_b128) is the MFMA
// "B" operand, the weight fragment the "A" operand, so a lane ends with 4 consecutive output
// columns of one row and writes the next stage's input back to LDS with 8-byte stores.
#include "common.h"
#include <type_traits>

// Probe build only (tools/probes/stamps_chain.py, -DCHAIN_STAMPS): every workgroup appends one record
// [kernel id | blockIdx, start, end (s_memrealtime, 100 MHz), HW_ID | XCC_ID, up to 12 s_memtime phase stamps] (16 words) to a device buffer.
#ifdef CHAIN_STAMPS
__device__ unsigned long long chain_stamps[8192 * 16];
__device__ unsigned int chain_stamp_count;
#define CSTAMP_DECL unsigned long long cst_[16]; cst_[0] = 0
#define CSTAMP_REAL(i) cst_[i] = __builtin_amdgcn_s_memrealtime()
#define CSTAMP(i) cst_[i] = __builtin_amdgcn_s_memtime()
#define CSTAMP_FLUSH(kernel_id, flag)                                                                                     \
  do {                                                                                                                    \
    if (threadIdx.x == 0) {                                                                                               \
      const unsigned slot = atomicAdd(&chain_stamp_count, 1u);                                                            \
      if (slot < 8192) {                                                                                                  \
        unsigned hw, xcc;                                                                                                 \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                                  \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                                \
        cst_[0] = ((unsigned long long)(kernel_id) << 48) | ((unsigned long long)(flag) << 32) | blockIdx.x;              \
        cst_[3] = ((unsigned long long)xcc << 32) | hw;                                                                   \
        for (int i_ = 0; i_ < 16; ++i_) chain_stamps[slot * 16 + i_] = cst_[i_];                                          \
      }                                                                                                                   \
    }                                                                                                                     \
  } while (0)
extern "C" int mvg_chain_read_stamps(unsigned long long* host, int max_records, int reset) {
  unsigned n = 0;
  hipError_t e = hipMemcpyFromSymbol(&n, HIP_SYMBOL(chain_stamp_count), sizeof(n));
  if (e != hipSuccess) return -(int)e;
  if (n > 8192) n = 8192;
  if ((int)n > max_records) n = max_records;
  if (host && n) e = hipMemcpyFromSymbol(host, HIP_SYMBOL(chain_stamps), sizeof(unsigned long long) * 16 * n);
  if (e != hipSuccess) return -(int)e;
  if (reset) {
    const unsigned z = 0;
    e = hipMemcpyToSymbol(HIP_SYMBOL(chain_stamp_count), &z, sizeof(z));
    if (e != hipSuccess) return -(int)e;
  }
  return (int)n;
}
#else
#define CSTAMP_DECL
#define CSTAMP_REAL(i)
#define CSTAMP(i)
#define CSTAMP_FLUSH(kernel_id, flag)
#endif

#include "chain_dev.h"

namespace {

// ------------------------------------------------------------------------------------------
// chain A
template <int RM, int NT, int JN>   // NT = 256 (4 waves) | 512 (8 waves: JN = 2 two row blocks, JN = 1 column split)
__global__ __launch_bounds__(NT, 2) void chain_a_kernel(const bf16_t* __restrict__ samp, const uint8_t* __restrict__ inside,
                                                      const bf16_t* __restrict__ Wp, const float* __restrict__ bp,
                                                      const bf16_t* __restrict__ W0, const float* __restrict__ b0,
                                                      const bf16_t* __restrict__ W1, const float* __restrict__ b1,
                                                      const float* __restrict__ W2, const float* __restrict__ b2,
                                                      bf16_t* __restrict__ attn, float* __restrict__ o,
                                                      const int* __restrict__ order, const float* __restrict__ o_masked,
                                                      int R) {
  static_assert(JN == 2 || NT == 512, "column-split mapping needs 8 wavefronts");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;
  int* rid = reinterpret_cast<int*>(smem + RM * ACT_PITCH);  // global row of every tile row (-1: past the end)
  float* w2s = reinterpret_cast<float*>(smem + RM * ACT_PITCH + RM * sizeof(int));   // last pose layer (3 x 256 f32)
  int* keepf = reinterpret_cast<int*>(w2s + 768);                                    // in-image flag of every tile row
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * RM;
  CSTAMP_DECL;
  CSTAMP_REAL(1);
  CSTAMP(4);

  // Tile row i works on global row order[r0 + i] (the sampler's processing order: the rows whose reference point is outside
  // their image come last, behind the in-image rows of ALL images -- the computing tiles lead the launch; mvg_bin_pairs) or r0 + i.  A tile without a single in-image row has attn = 0
  // and o = the MLP of a zero row for all of its rows: o_masked holds that row's result (computed by this very
  // kernel on one masked row), so such tiles only write their outputs.
  for (int i = tid; i < 768; i += NT) w2s[i] = W2[i];      // read by every thread in the last stage: LDS, not 96 global loads each
  bool mine = false;
  if (tid < RM) {
    const int slot = r0 + tid;
    const int g = slot < R ? (order ? order[slot] : slot) : -1;
    rid[tid] = g;
    mine = g >= 0 && inside[g] != 0;
    keepf[tid] = mine ? 1 : 0;
  }
  const bool any_inside = __syncthreads_or(mine) != 0;
  if (!any_inside && o_masked) {
    const float m0 = o_masked[0], m1 = o_masked[1], m2 = o_masked[2];
#pragma unroll
    for (int c0 = 0; c0 < RM * 32; c0 += NT) {
      const int c = c0 + tid, row = c >> 5, v16 = c & 31;
      const int g = rid[row];
      if (g >= 0) *reinterpret_cast<f32x4*>(attn + (long)g * 256 + v16 * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < RM && rid[tid] >= 0) {
      float* og = o + (long)rid[tid] * 3;
      og[0] = m0;
      og[1] = m1;
      og[2] = m2;
    }
    CSTAMP_REAL(2);
    CSTAMP_FLUSH(1, 0);
    return;
  }

  f32x4 pf1[4][JN];
  chain_a_ring1<RM, NT, JN>(Wp, pf1, tid);          // stage 1's first weight fragments, in flight under the tile load
  // samp tile -> LDS (16-byte vectors, rows past R are zero); all loads in flight before the first write
  {
    f32x4 x[RM * 32 / NT];
#pragma unroll
    for (int i = 0; i < RM * 32 / NT; ++i) {
      const int c = i * NT + tid, row = c >> 5, v16 = c & 31;
      // clamped address + select instead of a branch: a guarded load makes hipcc wait vmcnt(0) per element
      x[i] = *reinterpret_cast<const f32x4*>(samp + (long)max(rid[row], 0) * 256 + v16 * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RM * 32 / NT; ++i) {
      const int c = i * NT + tid, row = c >> 5, v16 = c & 31;
      *reinterpret_cast<f32x4*>(act + row * ACT_PITCH + v16 * 16) = (rid[row] >= 0) ? x[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  CSTAMP(5);
#ifdef CHAIN_STAMPS
  chain_a_body<RM, NT, JN, true>(act, rid, w2s, inside, Wp, bp, W0, b0, W1, b1, b2, attn, o, pf1, keepf, cst_);
#else
  chain_a_body<RM, NT, JN, true>(act, rid, w2s, inside, Wp, bp, W0, b0, W1, b1, b2, attn, o, pf1, keepf);
#endif
  CSTAMP(12);
  CSTAMP_REAL(2);
  CSTAMP_FLUSH(1, 1);
}

// ------------------------------------------------------------------------------------------
// chain B: one workgroup = QPT person-queries x 15 joints (60 token rows in a 64-row tile).
//   mean over views -> feature_update_mlp -> +tgt -> LN2 -> FFN (4 hidden chunks of 256) -> +t1 -> LN3
//   -> class head (sigmoid, mean over the 15 joints, threshold).
constexpr int XP = 1040;         // bytes per fp32 row in LDS (256 fp32 + 16 pad)

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// sum over the 8 lanes of a lane group (lanes 8g .. 8g+7), result in all of them: 3 DPP adds (quad_perm xor 1,
// quad_perm xor 2, row_half_mirror) instead of the 6 ds_bpermute round trips of a 64-lane butterfly
__device__ __forceinline__ float sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
  return v;
}

// acc (+bias) -> fp32 LDS tile xb[row][n] (add = accumulate onto what is there)
template <int MT, int JN = 2>
__device__ __forceinline__ void acc_to_x(char* __restrict__ xb, const f32x16 (&acc)[MT][JN], const float* __restrict__ bias,
                                         bool add, int tid) {
  const WaveMap<JN> wm(tid, MT * 32);
  const int lane = tid & 63, rl = lane & 31, h = lane >> 5, row0 = wm.row0;
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = wm.wn * 64 + (wm.j0 + j) * 32 + 8 * g + 4 * h;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4* dst = reinterpret_cast<f32x4*>(xb + (row0 + mt * 32 + rl) * XP + n * 4);
        f32x4 v = {acc[mt][j][4 * g] + bv[0], acc[mt][j][4 * g + 1] + bv[1], acc[mt][j][4 * g + 2] + bv[2],
                   acc[mt][j][4 * g + 3] + bv[3]};
        if (add) v += *dst;
        *dst = v;
      }
    }
}

template <int BRING, int NT, int JN, int RMT = 64>   // RMT rows per tile: 64 (4 persons of 15 joints) | 32 (2 persons: small launches)
__global__ __launch_bounds__(NT) void chain_b_kernel(
    const bf16_t* __restrict__ attn, int V, const float* __restrict__ tgt, const bf16_t* __restrict__ Wu,
    const float* __restrict__ bu, const float* __restrict__ g2, const float* __restrict__ be2,
    const bf16_t* __restrict__ W1, const float* __restrict__ b1, const bf16_t* __restrict__ W2,
    const float* __restrict__ b2, const float* __restrict__ g3, const float* __restrict__ be3,
    const float* __restrict__ Wc, const float* __restrict__ bc, float threshold, const uint8_t* __restrict__ forced,
    float* __restrict__ tgt_out, float* __restrict__ prob, uint8_t* __restrict__ valid, int* __restrict__ any_valid,
    const float* __restrict__ qpos, const bf16_t* __restrict__ Wn, const float* __restrict__ bn,
    float* __restrict__ xw_next, int n_next, int rows, int J, int nq_total, int has_ffn, const uint8_t* __restrict__ inside) {
  constexpr int RM = RMT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;                       // RM x 256 bf16 : GEMM operand (mean, then t1)
  char* hbuf = smem + RM * ACT_PITCH;     // RM x 256 bf16 : FFN hidden chunk
  char* xb = hbuf + RM * ACT_PITCH;       // RM x 256 fp32 : pre-LN sums / t1 / tgt'
  float* pr = reinterpret_cast<float*>(xb + RM * XP);   // RM x 2 per-row class probabilities
  // LayerNorm scales / shifts, the class head's two rows and the biases of the fp32 epilogues (acc_to_x), staged once:
  // [g2 | be2 | g3 | be3 | Wc0 | Wc1 | bu | b2 | bn] (9 x 256 fp32).
  // As global loads inside the row phases they sat on the tile's critical path -- in LN3 behind the tgt' stores, which the
  // in-order vmcnt makes a load wait for (s_memtime: LN3 + class head 12 400 cycles against 3 800 for LN2).
  float* lnp = pr + RM * 2;
  constexpr int MT = (JN == 1) ? RM / 32 : (RM / 32) / (NT / 256), NW = NT / 64;   // JN = 1: every wave covers all row blocks
  static_assert(MT >= 1, "tile too small for this wave mapping");
  static_assert(JN == 2 || NT == 512, "column-split mapping needs 8 wavefronts");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qpt = RM / J;                                // queries per tile (4 for J = 15)
  const int rpt = qpt * J;                               // real rows per tile (60)
  const int q0 = blockIdx.x * qpt, r0 = q0 * J;
  const int nrow = min(rpt, rows - r0);
  CSTAMP_DECL;
  CSTAMP_REAL(1);
  CSTAMP(4);
  // All tiles walk the SAME weights; in lock-step the 32 tiles of an XCD would hit the same L2 channel at the same time
  // (stage_gemm).  The k-step order is rotated per wavefront (= per column group) and, per TILE, by half a weight: tiles
  // (blockIdx >> 3) even / odd -- neighbours on one XCD, whose L2 they share -- start 8 k-steps apart.  The stage GEMMs
  // accumulate the two halves of the k range separately (SPLIT) and add them, so both phases give bit-identical rows: a
  // row's result does not depend on the tile it sits in, and query-sharded, permuted and single-rank runs agree bit for
  // bit.  (A free per-tile rotation of one accumulator was as fast but made the last bf16 bit depend on the tile.)
  const int rot = (((JN == 1 ? wave : (wave & 3)) * 3) + ((blockIdx.x >> 3) & 1) * 8) & 15;

  // Row phases (LayerNorms, class head): 8 lanes per row, a wavefront works on 8 rows at once; lane (g = lane>>3,
  // part = lane&7) holds the 8 channel quads part, part+8, ..., part+56 of row  wave*8 + g (+ 8*NW per pass), so a
  // row statistic is a 32-value local sum + a 3-step DPP reduction, and no loop over rows serialises memory or
  // cross-lane latencies.  tgt is fetched here, long before its use.
  constexpr int RPASS = (RM + 8 * NW - 1) / (8 * NW);   // 1 with 8 wavefronts, 2 with 4; RM = 32: wavefronts 4..7 have no rows
  const int rgrp = lane >> 3, part = lane & 7;
  // in-image flags of this wavefront's (row, view) pairs for the view mean below: requested FIRST, so that the wait for them leaves
  // everything behind them in flight (vmcnt counts in order).  Lane j holds the flag of row slot j >> 3 (chunk (j >> 4), half
  // wavefront (j >> 3) & 1) and view v0 + (j & 7).
  constexpr int NCHUNK = RM * 32 / NT;
  static_assert(2 * NCHUNK <= 8 && NCHUNK % 2 == 0, "flag layout: 8 lanes per (row slot, view) flag, at most 8 row slots in the 64-bit ballot; the V > 5 path steps two chunks at a time");
  auto load_flag = [&](int v0) -> unsigned {
    const int slot = lane >> 3, view = v0 + (lane & 7);
    const int trow = (slot >> 1) * (NT / 32) + wave * 2 + (slot & 1);
    if (slot < 2 * NCHUNK && view < V) return inside[(long)view * rows + r0 + min(trow, nrow - 1)];
    return 0u;
  };
  unsigned flag = inside ? load_flag(0) : 1u;
  constexpr int NLN = 9 * 256, NLNP = (NLN + NT - 1) / NT;
  float lnv[NLNP];
#pragma unroll
  for (int i = 0; i < NLNP; ++i) {
    const int e = min(i * NT + tid, NLN - 1), seg = e >> 8, o = e & 255;
    const float* src = seg == 0 ? g2 : seg == 1 ? be2 : seg == 2 ? (has_ffn ? g3 : g2) : seg == 3 ? (has_ffn ? be3 : be2)
                       : seg < 6 ? Wc + (seg - 4) * 256 : seg == 6 ? bu : seg == 7 ? (has_ffn ? b2 : bu) : (Wn ? bn : bu);
    lnv[i] = src[o];                                            // (b_next is zero-padded to 256 entries, as W_next's fragments are)
  }
  f32x4 tg[RPASS][8];
#pragma unroll
  for (int ps = 0; ps < RPASS; ++ps)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      tg[ps][i] = *reinterpret_cast<const f32x4*>(tgt + (long)(r0 + min(ps * 8 * NW + wave * 8 + rgrp, nrow - 1)) * 256 +
                                                  (part + 8 * i) * 4);

  // ---- mean over views (dq_decoder.py:770) -> act (bf16): all NCHUNK 16-byte chunks of a thread x 8 views in flight at once.
  // The phase is a chip-wide read burst (every tile wants its 5 x 32 KB of attn + 64 KB of tgt at the same time: ~60 MB at
  // 6-7 TB/s, 20 k cycles): what shortens it is bytes.  Rows of attn whose reference point is outside their image are zero by
  // construction (chain A: dq_decoder.py:585-586) -- a third of them at cfg-2 -- and with `inside` given they are not read:
  // the lane reads a cached dummy line instead and selects +0, the value the row holds (bit-identical sums).
  {
    const float inv = 1.f / (float)V;
    float sacc[NCHUNK][8];
    long off[NCHUNK];
#pragma unroll
    for (int u = 0; u < NCHUNK; ++u) {
#pragma unroll
      for (int t = 0; t < 8; ++t) sacc[u][t] = 0.f;
      const int c = u * NT + tid;
      off[u] = (long)(r0 + min(c >> 5, nrow - 1)) * 256 + (c & 31) * 8;
    }
    const uint4* dummy = reinterpret_cast<const uint4*>(bu);
    // views in groups of KV, UC chunks of the thread in flight together (clamped / dummy address + select instead of a guard: a
    // guarded load makes hipcc wait vmcnt(0) per element).  Up to 5 views: one group, all chunks (20 loads); more: groups of 8, two chunks.
    auto group = [&](auto kv_tag, auto uc_tag, int v0, int u0, unsigned long long m) {
      constexpr int KV = decltype(kv_tag)::value, UC = decltype(uc_tag)::value;
      uint4 x[UC][KV];
#pragma unroll
      for (int u = 0; u < UC; ++u)
#pragma unroll
        for (int k = 0; k < KV; ++k) {
          const bool live = (v0 + k < V) && ((m >> (((u0 + u) * 2 + (lane >> 5)) * 8 + k)) & 1ull);
          const uint4* src = reinterpret_cast<const uint4*>(attn + (long)min(v0 + k, V - 1) * rows * 256 + off[u0 + u]);
          x[u][k] = *(live ? src : dummy);
        }
#pragma unroll
      for (int u = 0; u < UC; ++u)
#pragma unroll
        for (int k = 0; k < KV; ++k) {
          const bool live = (v0 + k < V) && ((m >> (((u0 + u) * 2 + (lane >> 5)) * 8 + k)) & 1ull);
          const unsigned w4[4] = {x[u][k].x, x[u][k].y, x[u][k].z, x[u][k].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const unsigned w = live ? w4[t] : 0u;
            sacc[u0 + u][2 * t] += __uint_as_float(w << 16);
            sacc[u0 + u][2 * t + 1] += __uint_as_float(w & 0xffff0000u);
          }
        }
    };
    if (V <= 5) {
      const unsigned long long m = inside ? __ballot(flag != 0) : ~0ull;
      group(std::integral_constant<int, 5>{}, std::integral_constant<int, NCHUNK>{}, 0, 0, m);
    } else {
      for (int v0 = 0; v0 < V; v0 += 8) {
        const unsigned long long m = inside ? __ballot(flag != 0) : ~0ull;
        if (inside && v0 + 8 < V) flag = load_flag(v0 + 8);        // the next group's flags, in flight under this group's rows
#pragma unroll
        for (int u0 = 0; u0 < NCHUNK; u0 += 2) group(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{}, v0, u0, m);
      }
    }
#pragma unroll
    for (int u = 0; u < NCHUNK; ++u) {
      const int c = u * NT + tid, row = c >> 5, v16 = c & 31;
      uint4 o;
      unsigned* op = reinterpret_cast<unsigned*>(&o);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float a = row < nrow ? sacc[u][2 * t] * inv : 0.f, b = row < nrow ? sacc[u][2 * t + 1] * inv : 0.f;
        op[t] = pack_bf16(a, b);
      }
      *reinterpret_cast<uint4*>(act + row * ACT_PITCH + v16 * 16) = o;
    }
  }
#pragma unroll
  for (int i = 0; i < NLNP; ++i)
    if (i * NT + tid < NLN) lnp[i * NT + tid] = lnv[i];
  __syncthreads();
  CSTAMP(5);

  // ---- u = feature_update_mlp(mean) ; x = u + bu ; t1 = LN2(tgt + x)   (dq_decoder.py:773-778)
  f32x16 acc[MT][JN], acc2[MT][JN];
  f32x4 pf[BRING][JN];
  stage_gemm<MT, 16, BRING, JN, false, true>(act, Wu, acc, tid, true, rot, 16 * 1024, nullptr, acc2);
  if (has_ffn) ring_prefetch<16, BRING, JN, MT>(W1, pf, tid, rot);       // first FFN stage, fetched under LN2
  merge_acc<MT, JN>(acc, acc2);
  acc_to_x<MT, JN>(xb, acc, lnp + 1536, false, tid);
  __syncthreads();
  CSTAMP(6);
#pragma unroll
  for (int ps = 0; ps < RPASS; ++ps) {
    if (ps * 8 * NW + wave * 8 >= RM) continue;        // wavefront without rows (RM = 32)
    const int row = ps * 8 * NW + wave * 8 + rgrp;
    f32x4 v[8];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = *reinterpret_cast<const f32x4*>(xb + row * XP + (part + 8 * i) * 16);
      if (row < nrow) v[i] += tg[ps][i];
      sm += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = sum8(sm) * (1.f / 256.f);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = v[i] - mean;
      sq += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
    const float rstd = 1.f / sqrtf(sum8(sq) * (1.f / 256.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = part + 8 * i;
      const f32x4 y = v[i] * rstd * *reinterpret_cast<const f32x4*>(lnp + c4 * 4) + *reinterpret_cast<const f32x4*>(lnp + 256 + c4 * 4);
      *reinterpret_cast<f32x4*>(xb + row * XP + c4 * 16) = y;                         // t1 (fp32, residual)
      uint2 pk;
      pk.x = pack_bf16(y[0], y[1]);
      pk.y = pack_bf16(y[2], y[3]);
      *reinterpret_cast<uint2*>(act + row * ACT_PITCH + c4 * 8) = pk;                 // t1 (bf16, GEMM operand)
    }
  }
  __syncthreads();

  // query_pos of the last row phase: requested behind the FFN's last weight fragment (vmcnt is in order: requested in front of the
  // FFN, its first stage waited for these 64 KB per tile -- update + LN2 + FFN 51.7 k cycles against 46.3 k in the last layer,
  // which has no next layer and loads no query_pos), consumed after the LN3 statistics
  f32x4 qp[RPASS][8];
  auto load_qp = [&]() {
#pragma unroll
    for (int ps = 0; ps < RPASS; ++ps)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        qp[ps][i] = (Wn && qpos)
                        ? *reinterpret_cast<const f32x4*>(qpos + (long)(r0 + min(ps * 8 * NW + wave * 8 + rgrp, nrow - 1)) * 256 +
                                                          (part + 8 * i) * 4)
                        : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  if (!has_ffn) load_qp();
  CSTAMP(7);
  const float bc0 = bc[0], bc1 = bc[1];

  if (has_ffn) {
    // ---- FFN (mvp_decoder.py:94-98): Y = sum_c relu(t1 W1_c^T + b1_c) W2[:, c]^T, hidden chunks of 256
    f32x16 accy[MT][JN], accy2[MT][JN];
    bool all[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) all[mt] = true;
    // the first fragments of every stage are fetched one stage ahead (before the epilogue and barrier of the
    // previous one), see ring_prefetch
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      stage_gemm<MT, 16, BRING, JN, true, true>(act, W1 + (long)c * 256 * 256, acc, tid, true, rot + 3 * c, 16 * 1024, pf, acc2);
      merge_acc<MT, JN>(acc, acc2);
      f32x4 bv1[JN][4];
      load_bias<JN>(b1 + c * 256, bv1, tid, MT * 32);
      ring_prefetch<16, BRING, JN, MT>(W2 + (long)c * 16 * 1024, pf, tid, rot + 3 * c + 1, 64 * 1024);
      __builtin_amdgcn_sched_barrier(0);
      write_act_pre<MT, JN>(hbuf, acc, bv1, true, all, tid);                                   // private buffer: no hazard with act
      __syncthreads();
      stage_gemm<MT, 16, BRING, JN, true, true>(hbuf, W2 + (long)c * 16 * 1024, accy, tid, c == 0, rot + 3 * c + 1, 64 * 1024, pf, accy2);
      if (c < 3) ring_prefetch<16, BRING, JN, MT>(W1 + (long)(c + 1) * 256 * 256, pf, tid, rot + 3 * (c + 1));
      __syncthreads();                                                               // hbuf free for the next chunk
      if (c == 0) CSTAMP(8);
      if (c == 1) CSTAMP(9);
      if (c == 2) CSTAMP(10);
      if (c == 3) CSTAMP(11);
    }
    load_qp();
    if (Wn) ring_prefetch<16, BRING, JN, MT>(Wn, pf, tid, rot + 7);
    merge_acc<MT, JN>(accy, accy2);
    acc_to_x<MT, JN>(xb, accy, lnp + 1792, true, tid);                                   // x = t1 + Y + b2
    __syncthreads();
  }
  CSTAMP(12);

  // ---- tgt' = LN3(x) (or t1 when the FFN is off) ; class head per row (dq_decoder.py:889-893)
  // With a next layer (Wn) the tgt' rows stay in registers and are stored AFTER the query-term GEMM: vmcnt counts loads and
  // stores in order, so every weight fragment that GEMM requests behind the 64-KB store burst of a tile waits for the burst
  // (s_memrealtime stamps: LN3 + class head 5.9 k cycles without the GEMM, 17 k with it).  Its first fragments are requested here.
  f32x4 ykeep[RPASS][8];
  if (Wn && !has_ffn) ring_prefetch<16, BRING, JN, MT>(Wn, pf, tid, rot + 7);
#pragma unroll
  for (int ps = 0; ps < RPASS; ++ps) {
    if (ps * 8 * NW + wave * 8 >= RM) continue;
    const int row = ps * 8 * NW + wave * 8 + rgrp;
    f32x4 y[8];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      y[i] = *reinterpret_cast<const f32x4*>(xb + row * XP + (part + 8 * i) * 16);
      sm += y[i][0] + y[i][1] + y[i][2] + y[i][3];
    }
    if (has_ffn) {
      const float mean = sum8(sm) * (1.f / 256.f);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        y[i] = y[i] - mean;
        sq += y[i][0] * y[i][0] + y[i][1] * y[i][1] + y[i][2] * y[i][2] + y[i][3] * y[i][3];
      }
      const float rstd = 1.f / sqrtf(sum8(sq) * (1.f / 256.f) + 1e-5f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c4 = part + 8 * i;
        y[i] = y[i] * rstd * *reinterpret_cast<const f32x4*>(lnp + 512 + c4 * 4) + *reinterpret_cast<const f32x4*>(lnp + 768 + c4 * 4);
      }
    }
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = part + 8 * i;
      ykeep[ps][i] = y[i];
      if (!Wn && row < nrow) *reinterpret_cast<f32x4*>(tgt_out + (long)(r0 + row) * 256 + c4 * 4) = y[i];
      if (Wn) {   // operand of the next layer's query-term GEMM: tgt' + query_pos (bf16), into the now free act tile
        const f32x4 x = y[i] + qp[ps][i];
        uint2 pk;
        pk.x = pack_bf16(x[0], x[1]);
        pk.y = pack_bf16(x[2], x[3]);
        *reinterpret_cast<uint2*>(act + row * ACT_PITCH + c4 * 8) = pk;
      }
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(lnp + 1024 + c4 * 4), w1 = *reinterpret_cast<const f32x4*>(lnp + 1280 + c4 * 4);
      a0 += y[i][0] * w0[0] + y[i][1] * w0[1] + y[i][2] * w0[2] + y[i][3] * w0[3];
      a1 += y[i][0] * w1[0] + y[i][1] * w1[1] + y[i][2] * w1[2] + y[i][3] * w1[3];
    }
    a0 = sum8(a0) + bc0;
    a1 = sum8(a1) + bc1;
    if (part == 0) {
      pr[2 * row] = 1.f / (1.f + expf(-a0));
      pr[2 * row + 1] = 1.f / (1.f + expf(-a1));
    }
  }
  __syncthreads();
  CSTAMP(13);
  if (Wn) {
    // ---- xw = (tgt' + query_pos) W_next^T + b_next: the query term of the NEXT layer's offsets/logits Linear
    //      (projattn.py:180-181), computed while the rows are still in LDS (saves a 15 360-row GEMM launch and the
    //      elementwise add per layer).  The barrier above ordered the act writes and the last xb reads.
    stage_gemm<MT, 16, BRING, JN, true, true>(act, Wn, acc, tid, true, rot + 7, 16 * 1024, pf, acc2);
#pragma unroll
    for (int ps = 0; ps < RPASS; ++ps) {
      if (ps * 8 * NW + wave * 8 >= RM) continue;
      const int row = ps * 8 * NW + wave * 8 + rgrp;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (row < nrow) *reinterpret_cast<f32x4*>(tgt_out + (long)(r0 + row) * 256 + (part + 8 * i) * 4) = ykeep[ps][i];
    }
    merge_acc<MT, JN>(acc, acc2);
    acc_to_x<MT, JN>(xb, acc, lnp + 2048, false, tid);
    __syncthreads();
    for (int row = wave; row < nrow; row += NW)
      if (lane * 4 < n_next)
        *reinterpret_cast<f32x4*>(xw_next + (long)(r0 + row) * n_next + lane * 4) =
            *reinterpret_cast<const f32x4*>(xb + row * XP + lane * 16);
  }
  CSTAMP(14);
  // per-query probabilities and validity (after the query-term GEMM: no load of the kernel waits behind these stores)
  if (tid < qpt && q0 + tid < nq_total) {
    float p0 = 0.f, p1 = 0.f;
    for (int j = 0; j < J; ++j) {
      p0 += pr[2 * (tid * J + j)];
      p1 += pr[2 * (tid * J + j) + 1];
    }
    p0 /= (float)J;
    p1 /= (float)J;
    const int qi = q0 + tid;
    prob[2 * (long)qi] = p0;
    prob[2 * (long)qi + 1] = p1;
    const bool ok = forced ? (forced[qi] != 0) : (p1 > threshold);                   // dq_decoder.py:605
    valid[qi] = ok ? 1 : 0;
    if (ok) atomicOr(any_valid, 1);
  }
  CSTAMP(15);
  CSTAMP_REAL(2);
  CSTAMP_FLUSH(2, 1);
}

}  // namespace

extern int g_auto_small;   // tuning knob "auto_small" (msda.hip): launches with few rows pick smaller tiles (bit-identical rows)
int g_chain_a_lds_pad = 0;   // probe knob "chain_a_lds_pad": unused dynamic LDS per chain-A workgroup (caps the workgroups per CU)
int g_chain_rm = 128;  // tuning knob "chain_rm": rows per workgroup of chain A (64 | 128 | 256); 128: 65 -> 55 us (half the weight
                       // bytes per row).  Geometries measured and deleted (rounds 1-4, Appendix A of DESIGN.md): 8 wavefronts for
                       // chain A (93 vs 79 us), 4 wavefronts / row-block split / fragment rings of 8 and 16 for chain B.

template <int RM, int NT, int JN>
static int launch_chain_a(const void* samp, const uint8_t* inside, const void* Wp, const float* bp, const void* W0,
                          const float* b0, const void* W1, const float* b1, const float* W2, const float* b2, void* attn,
                          float* o, const int* order, const float* o_masked, int rows, hipStream_t st) {
  const size_t lds = RM * ACT_PITCH + 2 * RM * sizeof(int) + 768 * sizeof(float) + (size_t)g_chain_a_lds_pad;
  // the attribute is per DEVICE: a process that drives several GPUs configures the > 64-KB LDS kernels on each of them
  static bool configured[MVG_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVG_MAX_DEVICES) return MVG_E_BADARG;
  if (!configured[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_a_kernel<RM, NT, JN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) return (int)e;
    configured[dev] = true;
  }
  hipLaunchKernelGGL((chain_a_kernel<RM, NT, JN>), dim3((rows + RM - 1) / RM), dim3(NT), lds, st, (const bf16_t*)samp, inside,
                     (const bf16_t*)Wp, bp, (const bf16_t*)W0, b0, (const bf16_t*)W1, b1, W2, b2, (bf16_t*)attn, o, order,
                     o_masked, rows);
  MVG_LAUNCH_CHECK();
  return 0;
}

extern "C" int mvg_chain_attn_pose(const void* samp, const uint8_t* inside, const void* Wp, const float* bp,
                                   const void* W0, const float* b0, const void* W1, const float* b1, const float* W2,
                                   const float* b2, void* attn, float* o, const int32_t* order, const float* o_masked,
                                   int rows, void* stream) {
  if (!samp || !inside || !Wp || !bp || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !attn || !o || rows < 0) return MVG_E_BADARG;
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // Every variant computes a row bit-identically (stage GEMMs: the k-step order depends on the column group only; last pose layer:
  // a reduction tree that does not depend on the threads per row, chain_a_body), so the tile size may follow the row count: with
  // at most 320 tiles of 128 rows -- a rank's shard of a query-sharded run, small scenes -- 64-row tiles put twice as many
  // workgroups on the chip (measured at cfg-2 with 128 / 256 / 512 queries: -2.6 / -0.6 / -1.4 % of the forward; the full 1024
  // queries are 1.4 % faster with 128-row tiles).
  if (g_auto_small && g_chain_rm == 128 && rows <= 320 * 128)
    return launch_chain_a<64, 256, 2>(samp, inside, Wp, bp, W0, b0, W1, b1, W2, b2, attn, o, order, o_masked, rows, st);
  if (g_chain_rm == 256) return launch_chain_a<256, 512, 1>(samp, inside, Wp, bp, W0, b0, W1, b1, W2, b2, attn, o, order, o_masked, rows, st);
  if (g_chain_rm == 128) return launch_chain_a<128, 256, 2>(samp, inside, Wp, bp, W0, b0, W1, b1, W2, b2, attn, o, order, o_masked, rows, st);
  return launch_chain_a<64, 256, 2>(samp, inside, Wp, bp, W0, b0, W1, b1, W2, b2, attn, o, order, o_masked, rows, st);
}

extern "C" int mvg_chain_update_ffn_class(const void* attn, int V, const float* tgt, const void* Wu, const float* bu,
                                          const float* g2, const float* be2, const void* W1, const float* b1,
                                          const void* W2, const float* b2, const float* g3, const float* be3,
                                          const float* Wc, const float* bc, float threshold, const uint8_t* forced_valid,
                                          float* tgt_out, float* prob, uint8_t* valid, int* any_valid,
                                          const float* query_pos, const void* W_next, const float* b_next,
                                          float* xw_next, int n_next, int B, int NQ, int J, int has_ffn,
                                          const uint8_t* attn_inside, void* stream) {
  if (!attn || !tgt || !Wu || !bu || !g2 || !be2 || !Wc || !bc || !tgt_out || !prob || !valid || !any_valid) return MVG_E_BADARG;
  if (has_ffn && (!W1 || !b1 || !W2 || !b2 || !g3 || !be3)) return MVG_E_BADARG;
  if (V <= 0 || J <= 0 || J > 64 || B < 0 || NQ < 0) return MVG_E_BADARG;
  if (W_next && (!b_next || !xw_next || n_next <= 0 || n_next > 256 || n_next % 4 != 0)) return MVG_E_BADARG;
  const int nq_total = B * NQ, rows = nq_total * J;
  if (rows == 0) return 0;
  // few rows (a rank's shard of a query-sharded run): 32-row tiles (2 persons) fill twice as many CUs
  const bool small = g_auto_small && J <= 16 && (nq_total + (64 / J) - 1) / (64 / J) <= 128;
  const int RMr = small ? 32 : 64;
  const int qpt = RMr / J;
  const size_t lds = 2 * RMr * ACT_PITCH + RMr * XP + RMr * 2 * sizeof(float) + 9 * 256 * sizeof(float);
  static bool configured[MVG_MAX_DEVICES] = {};   // per device, see launch_chain_a
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVG_MAX_DEVICES) return MVG_E_BADARG;
  const size_t lds64 = 2 * 64 * ACT_PITCH + 64 * XP + 64 * 2 * sizeof(float) + 9 * 256 * sizeof(float);
  if (!configured[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_b_kernel<4, 512, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_b_kernel<4, 512, 1, 32>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
    if (e != hipSuccess) return (int)e;
    configured[dev] = true;
  }
  const dim3 grid((nq_total + qpt - 1) / qpt);
#define MVG_CB(R, NTH, JNN, ...)                                                                                           \
  hipLaunchKernelGGL((chain_b_kernel<R, NTH, JNN, ##__VA_ARGS__>), grid, dim3(NTH), lds, (hipStream_t)stream, (const bf16_t*)attn, V, tgt,  \
                     (const bf16_t*)Wu, bu, g2, be2, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, g3, be3, Wc, bc,     \
                     threshold, forced_valid, tgt_out, prob, valid, any_valid, query_pos, (const bf16_t*)W_next, b_next, xw_next,  \
                     n_next, rows, J, nq_total, has_ffn, attn_inside)
  if (small) MVG_CB(4, 512, 1, 32);
  else MVG_CB(4, 512, 1);
#undef MVG_CB
  MVG_LAUNCH_CHECK();
  return 0;
}
