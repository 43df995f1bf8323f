// Device building blocks of the fused Linear chains (csrc/chain.hip).  See chain.hip for the design notes.
#pragma once
#include "common.h"

#ifndef CSTAMP
#define CSTAMP(i)
#endif

namespace {


constexpr int ACT_PITCH = 528;   // bytes per activation row in LDS (256 bf16 + 16 pad)


// Stage GEMM  acc[m][n] = sum_k act[m][k] * W[n][k]  for one 256-column block of W.
// Weights are NOT staged through LDS: they are pre-swizzled on the host into MFMA-fragment order
//   Wf[wn 4][ks K/16][j 2][lane 64][8 bf16],  lane (rl = lane&31, h = lane>>5) holds
//   W[n = wn*64 + j*32 + rl][k = ks*16 + h*8 .. +8],
// so every fragment load of a wavefront is one fully coalesced 1-KB global_load_dwordx4 from the
// L2-resident weight (128 KB per 256x256 layer), issued RING k-steps ahead of its MFMAs.  The
// waves of a workgroup never synchronise inside a stage (the activation tile is read-only).
// `rot` rotates the k-step order per workgroup: all workgroups walk the SAME 128-KB weight, and in
// lock-step they would all hit the same L2 channel at the same time (measured: 1.3 us average
// load latency, MFMA utilisation 13 %); with a per-workgroup rotation the requests spread over the
// whole weight at any instant.  (fp32 accumulation order changes with rot: results agree with the
// unfused kernels to fp32 rounding, not bit for bit.)
// Wave -> sub-tile mapping of a stage (JN = 32-column blocks per wavefront):
//   JN = 2: wave w owns columns [64 (w&3), +64) of row block (w>>2) -- 4 waves cover the 256 columns, 8 waves two
//           row blocks (both groups stream the SAME weight fragments: twice the L2->L1 traffic);
//   JN = 1: wave w owns columns [32 w, +32) of ALL MT row blocks -- 8 waves, every weight fragment is loaded by
//           exactly one wave (chain B: 641 -> 350 MB through the L1 miss path per launch).
template <int JN>
struct WaveMap {
  int wn, j0, row0;
  __device__ __forceinline__ WaveMap(int tid, int mt_rows) {
    if (JN == 2) {
      wn = (tid >> 6) & 3;
      j0 = 0;
      row0 = (tid >> 8) * mt_rows;
    } else {
      wn = (tid >> 7) & 3;
      j0 = (tid >> 6) & 1;
      row0 = 0;
    }
  }
};

// The first RING weight fragments of a stage.  Issued by the caller BEFORE the previous stage's epilogue and barrier
// (stage_gemm with PRE = true then starts on fragments that are already in flight): a stage that loads them itself
// exposes one full L2 round trip before its first MFMA, 10 times per chain-B tile.
template <int KSTEPS, int RING, int JN, int MT>
__device__ __forceinline__ void ring_prefetch(const bf16_t* __restrict__ Wf, f32x4 (&ring)[RING][JN], int tid, int rot,
                                              int wn_stride = KSTEPS * 1024) {
  const WaveMap<JN> wm(tid, MT * 32);
  const bf16_t* wp = Wf + (long)wm.wn * wn_stride + wm.j0 * 512 + (tid & 63) * 8;
#pragma unroll
  for (int p = 0; p < RING; ++p) {
    const int kq = (p + rot) & (KSTEPS - 1);
#pragma unroll
    for (int j = 0; j < JN; ++j) ring[p][j] = *reinterpret_cast<const f32x4*>(wp + kq * 1024 + j * 512);
  }
}

// SPLIT: the first KSTEPS/2 k-steps (in rotated order) accumulate into `acc`, the last KSTEPS/2 into `acc2`; the caller adds
// the two.  Because that final fp32 addition commutes, rot and rot + KSTEPS/2 give BIT-IDENTICAL results: a kernel can run
// its tiles in two phases that are half a weight apart (L2-channel decorrelation) without a row's result depending on
// the tile it happens to be in (chain B).
template <int MT, int KSTEPS, int RING = 4, int JN = 2, bool PRE = false, bool SPLIT = false>
__device__ __forceinline__ void stage_gemm(const char* __restrict__ act, const bf16_t* __restrict__ Wf,
                                           f32x16 (&acc)[MT][JN], int tid, bool zero, int rot,
                                           int wn_stride = KSTEPS * 1024, f32x4 (*pre)[JN] = nullptr,
                                           f32x16 (*acc2)[JN] = nullptr) {
  static_assert((KSTEPS & (KSTEPS - 1)) == 0, "KSTEPS must be a power of two");
  const WaveMap<JN> wm(tid, MT * 32);
  const int lane = tid & 63, rl = lane & 31, h = lane >> 5, row0 = wm.row0;
  if (zero) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          acc[mt][j][e] = 0.f;
          if (SPLIT) acc2[mt][j][e] = 0.f;
        }
  }
  const bf16_t* wp = Wf + (long)wm.wn * wn_stride + wm.j0 * 512 + lane * 8;   // wn_stride: elements between the wave slices
  f32x4 ring[RING][JN];
#pragma unroll
  for (int p = 0; p < RING; ++p) {
    const int kq = (p + rot) & (KSTEPS - 1);
#pragma unroll
    for (int j = 0; j < JN; ++j)
      ring[p][j] = PRE ? pre[p][j] : *reinterpret_cast<const f32x4*>(wp + kq * 1024 + j * 512);
  }
  // The activation fragments are read from LDS one k-step ahead of their MFMAs (register double buffer): issued in
  // the same k-step, every step exposed the LDS latency in front of its first MFMA (MFMA pipe ~60 % busy with the
  // two wavefronts of a SIMD alternating).
  const char* arow = act + (row0 + rl) * ACT_PITCH + 16 * h;
  f32x4 a_nxt[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    a_nxt[mt] = *reinterpret_cast<const f32x4*>(arow + mt * 32 * ACT_PITCH + (rot & (KSTEPS - 1)) * 32);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    f32x4 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = a_nxt[mt];
    if (ks + 1 < KSTEPS) {
      const int kn = (ks + 1 + rot) & (KSTEPS - 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = *reinterpret_cast<const f32x4*>(arow + mt * 32 * ACT_PITCH + kn * 32);
    }
    f32x4 b[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) b[j] = ring[ks % RING][j];
    if (ks + RING < KSTEPS) {
      const int kq = (ks + RING + rot) & (KSTEPS - 1);
#pragma unroll
      for (int j = 0; j < JN; ++j) ring[ks % RING][j] = *reinterpret_cast<const f32x4*>(wp + kq * 1024 + j * 512);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        if (SPLIT && ks >= KSTEPS / 2)
          acc2[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, a[mt]),
                                                               acc2[mt][j], 0, 0, 0);
        else
          acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, a[mt]),
                                                              acc[mt][j], 0, 0, 0);
      }
    // pin the k-step: without this hipcc sinks the ring refills down to their uses (issue -> vmcnt(0)
    // -> MFMA in the same step, i.e. no prefetch distance at all; measured MFMA utilisation 13 %)
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MT, int JN>
__device__ __forceinline__ void merge_acc(f32x16 (&acc)[MT][JN], const f32x16 (&acc2)[MT][JN]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[mt][j] += acc2[mt][j];
}

// acc (+bias, relu, row keep-mask) -> bf16 activation tile in LDS, IN PLACE: the caller puts a
// __syncthreads() before (every wave finished reading `act`) and after (next stage may read).
template <int MT, int JN = 2>
__device__ __forceinline__ void write_act(char* __restrict__ act, const f32x16 (&acc)[MT][JN],
                                          const float* __restrict__ bias, bool relu, const bool (&keep)[MT], int tid) {
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
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float x = acc[mt][j][4 * g + t] + bv[t];
          if (relu) x = fmaxf(x, 0.f);
          v[t] = keep[mt] ? x : 0.f;
        }
        uint2 pk;
        pk.x = pack_bf16(v[0], v[1]);
        pk.y = pack_bf16(v[2], v[3]);
        *reinterpret_cast<uint2*>(act + (row0 + mt * 32 + rl) * ACT_PITCH + n * 2) = pk;
      }
    }
}

// Same with the bias already in registers (load_bias): lets the caller order  bias loads -> next stage's ring_prefetch
// -> epilogue, so that the epilogue's wait on the bias (vmcnt counts in order) leaves the prefetch in flight.
template <int JN>
__device__ __forceinline__ void load_bias(const float* __restrict__ bias, f32x4 (&bv)[JN][4], int tid, int mt_rows) {
  const WaveMap<JN> wm(tid, mt_rows);
  const int h = (tid & 63) >> 5;
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bv[j][g] = *reinterpret_cast<const f32x4*>(bias + wm.wn * 64 + (wm.j0 + j) * 32 + 8 * g + 4 * h);
}

template <int MT, int JN>
__device__ __forceinline__ void write_act_pre(char* __restrict__ act, const f32x16 (&acc)[MT][JN], const f32x4 (&bv)[JN][4],
                                              bool relu, const bool (&keep)[MT], int tid) {
  const WaveMap<JN> wm(tid, MT * 32);
  const int lane = tid & 63, rl = lane & 31, h = lane >> 5, row0 = wm.row0;
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = wm.wn * 64 + (wm.j0 + j) * 32 + 8 * g + 4 * h;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float x = acc[mt][j][4 * g + t] + bv[j][g][t];
          v[t] = keep[mt] ? (relu ? fmaxf(x, 0.f) : x) : 0.f;
        }
        uint2 pk;
        pk.x = pack_bf16(v[0], v[1]);
        pk.y = pack_bf16(v[2], v[3]);
        *reinterpret_cast<uint2*>(act + (row0 + mt * 32 + rl) * ACT_PITCH + n * 2) = pk;
      }
    }
}

// Chain A on a tile that is already in LDS: act (RM x 256 bf16 sampled rows, ACT_PITCH), rid (global row of every
// tile row, -1 past the end), w2s (last pose layer, 3 x 256 f32).  attn = inside * (act Wp^T + bp) -> global (needed
// for the view mean), then the pose MLP -> o (dx, dy, confidence logit).  Called by all threads of the workgroup after a
// barrier that made the tile visible (chain_a_kernel: tile loaded from samp).
// PRE1: the first weight fragments of stage 1 were requested by the caller before it loaded the tile (chain_a_ring1), so that
// their round trip runs under the tile's (s_memtime: stage 1 took 11 500 cycles against 6 300 - 7 800 for stages 2 and 3,
// which have always had this prefetch).
template <int RM, int NT, int JN>
__device__ __forceinline__ void chain_a_ring1(const bf16_t* __restrict__ Wp, f32x4 (&pf1)[4][JN], int tid) {
  constexpr int MT = (JN == 1) ? RM / 32 : RM / 32 / (NT / 256);
  const int rot = ((JN == 1 ? (tid >> 6) : ((tid >> 6) & 3)) * 3) & 15;          // chain_a_body's rotation
  ring_prefetch<16, 4, JN, MT>(Wp, pf1, tid, rot);
}

template <int RM, int NT, int JN, bool PRE1 = false>
__device__ __forceinline__ void chain_a_body(char* __restrict__ act, const int* __restrict__ rid, const float* __restrict__ w2s,
                                             const uint8_t* __restrict__ inside, const bf16_t* __restrict__ Wp,
                                             const float* __restrict__ bp, const bf16_t* __restrict__ W0,
                                             const float* __restrict__ b0, const bf16_t* __restrict__ W1,
                                             const float* __restrict__ b1, const float* __restrict__ b2,
                                             bf16_t* __restrict__ attn, float* __restrict__ o,
                                             f32x4 (*pf1)[JN] = nullptr, const int* __restrict__ keep_lds = nullptr,
                                             unsigned long long* cst_ = nullptr) {   // cst_: probe build only (CSTAMP)
  constexpr int MT = (JN == 1) ? RM / 32 : RM / 32 / (NT / 256);                  // row tiles per wave
  const int tid = threadIdx.x, lane = tid & 63, rl = lane & 31;
  const int row0 = (JN == 1) ? 0 : (tid >> 8) * MT * 32;
  // k-step rotation per wavefront only, NOT per tile: which tile a row lands in depends on the processing order
  // (mvg_bin_pairs: the order inside a bin is whatever the LDS atomics produce), and a row's fp32 accumulation order
  // -- hence its result, bit for bit -- must not.  (A per-tile rotation was worth 1 us of 39.)
  const int rot = ((JN == 1 ? (tid >> 6) : ((tid >> 6) & 3)) * 3) & 15;
  f32x16 acc[MT][JN];
  bool keep[MT], all[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int g = rid[row0 + mt * 32 + rl];
    // dq_decoder.py:585-586.  keep_lds: the caller already fetched the rows' flags (one per tile row, next to rid) -- as a
    // global load here, dependent on rid, its round trip sat in front of stage 1's barrier
    keep[mt] = keep_lds ? keep_lds[row0 + mt * 32 + rl] != 0 : (g >= 0 && inside[max(g, 0)] != 0);
    all[mt] = true;
  }
  __syncthreads();
  // attn = inside * output_proj(samp).  Every stage's bias and the NEXT stage's first weight fragments are requested
  // before the barrier + epilogue that follow its k-loop (ring_prefetch).
  f32x4 pf[4][JN], bvr[JN][4];
  stage_gemm<MT, 16, 4, JN, PRE1>(act, Wp, acc, tid, true, rot, 16 * 1024, pf1);
  CSTAMP(6);
  load_bias<JN>(bp, bvr, tid, MT * 32);
  ring_prefetch<16, 4, JN, MT>(W0, pf, tid, rot + 5);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  write_act_pre<MT, JN>(act, acc, bvr, false, keep, tid);
  __syncthreads();
#pragma unroll
  for (int c0 = 0; c0 < RM * 32; c0 += NT) {
    const int c = c0 + tid, row = c >> 5, v16 = c & 31;
    const int g = rid[row];
    if (g >= 0)
      *reinterpret_cast<f32x4*>(attn + (long)g * 256 + v16 * 8) =
          *reinterpret_cast<const f32x4*>(act + row * ACT_PITCH + v16 * 16);
  }
  CSTAMP(7);
  // pose_embed MLP layers 0, 1 (ReLU)
  stage_gemm<MT, 16, 4, JN, true>(act, W0, acc, tid, true, rot + 5, 16 * 1024, pf);
  CSTAMP(8);
  load_bias<JN>(b0, bvr, tid, MT * 32);
  ring_prefetch<16, 4, JN, MT>(W1, pf, tid, rot + 10);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  write_act_pre<MT, JN>(act, acc, bvr, true, all, tid);
  __syncthreads();
  CSTAMP(9);
  stage_gemm<MT, 16, 4, JN, true>(act, W1, acc, tid, true, rot + 10, 16 * 1024, pf);
  __syncthreads();
  CSTAMP(10);
  write_act<MT, JN>(act, acc, b1, true, all, tid);
  __syncthreads();
  CSTAMP(11);
  // last layer (3 outputs): TPR = 2 | 4 | 8 threads per row (RM = 128 | 64 | 32 rows on 256 threads, ...).  The 256-term dot
  // products are summed in an order that does NOT depend on TPR: eight 32-column chunk sums c0..c7 (each accumulated in column
  // order), then the balanced tree ((c0+c1)+(c2+c3)) + ((c4+c5)+(c6+c7)) -- inside a lane while it owns several chunks, across
  // lanes (DPP) after that.  Every tile size therefore gives bit-identical rows, and the launcher may pick it by the row count.
  constexpr int TPR = NT / RM, CPT = 256 / TPR, NCHK = CPT / 32;
  static_assert(TPR == 2 || TPR == 4 || TPR == 8, "threads per row of the last pose layer");
  const int row = tid / TPR, part = tid % TPR;
  float s[3];
  {
    float cs[3][NCHK];
#pragma unroll
    for (int q = 0; q < NCHK; ++q) {
      float a3[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 32; c += 8) {
        const int col = part * CPT + q * 32 + c;
        const uint4 hv = *reinterpret_cast<const uint4*>(act + row * ACT_PITCH + col * 2);
        const unsigned w4[4] = {hv.x, hv.y, hv.z, hv.w};
        float hf[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          hf[2 * t] = __uint_as_float(w4[t] << 16);
          hf[2 * t + 1] = __uint_as_float(w4[t] & 0xffff0000u);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const f32x4 wa = *reinterpret_cast<const f32x4*>(w2s + k * 256 + col);
          const f32x4 wb = *reinterpret_cast<const f32x4*>(w2s + k * 256 + col + 4);
          a3[k] += hf[0] * wa[0] + hf[1] * wa[1] + hf[2] * wa[2] + hf[3] * wa[3] + hf[4] * wb[0] + hf[5] * wb[1] +
                   hf[6] * wb[2] + hf[7] * wb[3];
          // Keep the three accumulators scalar.  hipcc (ROCm 7.2) pairs them into v_pk_mul/v_pk_fma_f32 fed by
          // v_mov/v_pk_mov shuffles of the ds_read_b128 results, and that sequence returned wrong sums in lanes 48-63 of
          // a wavefront, run-to-run differently, whenever two workgroups shared a CU (found by tools/soak.py; the
          // packed pair was always the culprit: component 2, computed with scalar FMAs, never differed).
          asm volatile("" : "+v"(a3[k]));
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) cs[k][q] = a3[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v;
      if (NCHK == 4) v = (cs[k][0] + cs[k][1]) + (cs[k][2] + cs[k][3]);
      else if (NCHK == 2) v = cs[k][0] + cs[k][1];
      else v = cs[k][0];
      s[k] = v;
    }
  }
  // sum over the TPR (2 | 4 | 8) adjacent lanes of a row with DPP (quad_perm xor 1, xor 2, row_half_mirror)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = s[k];
    if (TPR >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    if (TPR >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    if (TPR >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    s[k] = v;
  }
  if (part == 0 && rid[row] >= 0) {
    float* og = o + (long)rid[row] * 3;
    og[0] = s[0] + b2[0];
    og[1] = s[1] + b2[1];
    og[2] = s[2] + b2[2];
  }
}

}  // namespace
