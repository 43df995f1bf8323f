// fp32 path of the decoder layer as fused kernels: fp32 storage at every boundary, every product formed on the fp16 matrix pipe
// from TWO-part operands ("f32h": x 2^s = h + l, three MFMAs per product, fp32 accumulation -- f32s_dev.h and the note in front of
// pyramid_ws_f32h_kernel).  The reference's arithmetic is fp32 (lib/models/ops/src/cuda/deform_cuda.cu:75, the Linears of
// lib/models/dq_decoder.py:763-848,659-717); rounds 1-3 ran it as 17 launches per layer with every intermediate in HBM.
//
//   mvg_pyramid_f32h                 value = feat Wv^T + bv  and  G = feat [Wo; Wa]^T  in ONE launch over the pyramid
//                                    (projattn.py:169 and the pyramid side of :180-181): weight-stationary persistent workgroups,
//                                    each owning one of the two products (pyramid_ws_f32h_kernel).
//   mvg_chain_attn_pose_f32h         chain A: attn = inside * (samp Wp^T + bp) -> stored; o = pose MLP(attn)   (dq_decoder.py:585-588,659-690)
//   mvg_chain_update_ffn_class_f32h  chain B: view mean -> update Linear -> +tgt -> LN2 -> FFN -> LN3 -> class head -> next layer's
//                                    query term (dq_decoder.py:770-778, mvp_decoder.py:94-98, dq_decoder.py:889-893)
//
// Geometry shared by the three: 512 threads = 8 wavefronts, wavefront w owns the 32-column block w of every 256-column weight
// block.  Weight operands: two fp16 planes (h, l), each in ops.swizzle_weight order, plane p at p * N * K elements, and the
// tensor's power-of-two scale.  The six-product bf16 forms of round 4's first half (pyramid_f32s / chain_*_f32s kernels, three
// bf16 parts per operand) computed the same rows at 1.4-1.6 x the time and are deleted (profiles/r04_experiments.txt holds their
// numbers); the range-safe alternative is the unfused path (MVG_F32_FUSED=0: mvg_linear's three-part split form per GEMM).
#include <algorithm>
#include <type_traits>

#include "common.h"

#include "f32s_dev.h"

// Measurement hook (variant builds of tools/ab_f32s.sh only; the product is built without it): s_memtime stamps of one thread
// per workgroup at phase boundaries, read back with mvg_f32s_read_stamps.
#ifndef F32S_PRIO
#define F32S_PRIO 1           // the two wavefronts of a SIMD alternate issue priority per k-step (f32s_dev.h: stage_h2)
#endif

// measurement builds (tools/probes/stamps_pyr_ws.py): s_memtime per wavefront at the phase boundaries of one steady-state tile
#ifdef PYRWS_STAMPS
__device__ unsigned long long pyrws_stamps[512 * 8 * 16];
#define PWSTAMP(i)                                                                                                        \
  do {                                                                                                                    \
    if (k == PYRWS_STAMPS && lane == 0) {                                                                                 \
      asm volatile("" ::: "memory");                                                                                      \
      pyrws_stamps[(blockIdx.x * 8 + w) * 16 + (i)] = __builtin_amdgcn_s_memtime();                                       \
      asm volatile("" ::: "memory");                                                                                      \
    }                                                                                                                     \
  } while (0)
extern "C" int mvg_pyrws_read_stamps(unsigned long long* host, int n_blocks) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pyrws_stamps), sizeof(unsigned long long) * 128 * n_blocks);
}
#else
#define PWSTAMP(i)
#endif

namespace {

using namespace f32s;

constexpr int RM = 64;                       // rows per tile
constexpr int NT = 512;

// ------------------------------------------------------------------------------------------------------------------
// pyramid products on TWO-part fp16 operands ("f32h", round 4): x 2^s = h + l with h = fp16(x 2^s), l = fp16(x 2^s - h) -- 22 bits
// of each operand -- and a * w as THREE fp16 MFMAs  l*h + h*l + h*h  with fp32 accumulation, half the matrix work of the six-product
// bf16 form.  fp16 has 5 exponent bits, so every operand carries a power-of-two scale (exact to apply and to remove): an activation
// row is scaled so that its largest entry lies in [2^13, 2^14) (s_row from the exponent of the row maximum, found by the wavefront that
// holds the row), a weight tensor once on the host (ops.split_swizzle_weight_h2).  What is dropped: l*l (2^-22 |a w|) and the parts'
// rounding (2^-23 of an entry, or 2^-25 of the row / tensor maximum for entries in fp16's subnormal range) -- measured against fp64 on
// operands with exact accumulation 4e-8 of sum|a||w| (six-product bf16 form: 3e-9), an order of magnitude under the 3e-7 the fp32
// accumulation itself leaves in either form.  A row's scale depends on that row only: results are position-independent as before.
// ------------------------------------------------------------------------------------------------------------------
// pyramid products, weight-stationary (round 5).  Round 4's tiled kernel (pyramid_f32h_kernel, deleted) streamed 448 KB of weight
// planes per 64-row tile through the CU's vector-memory path, found each row's maximum with a 64-lane butterfly and stored its outputs
// as 4-byte-per-lane stores (32 store instructions per stage and wavefront): ~100 vector-memory instructions per wavefront and tile
// next to 168 MFMAs -- 191 us per layer at cfg-2 with the matrix pipe 32 % busy.  Here a workgroup owns ONE of the two products for the whole launch, the way the bf16 kernels of wreg_gemm.hip do: wavefront w
// keeps the two planes of column block w (32 columns x 256 k x 2 planes = 128 registers, loaded in the k order it will walk them),
// 32-row tiles of the pyramid go through a double-buffered LDS image of their planes, the outputs through wavefront-private staging
// rows as 16-byte stores (8 x 128-byte lines per instruction).  Per wavefront and tile: 4 row loads, 4 stores, 48 MFMAs; the
// split of the NEXT tile's rows (row maximum over 16 lanes by four DPP steps, scale, two fp16 parts) and the stores of the PREVIOUS tile are
// issued between the MFMAs of the k loop.  One workgroup barrier per tile.  Workgroups of an XCD (block b -> XCD b & 7) are divided
// between the two products in proportion to their columns and walk the XCD's tiles in the same order: a tile comes from HBM once.
// Same products, same k order per column block (rot), same three-term order as the tiled kernel: bit-identical outputs
// (checked at six shapes before it was deleted; cfg-2 fp32 2.47 -> 2.36 ms, cfg-4 1.82 -> 1.71 ms on one box: profiles/r05_experiments.txt).
constexpr int WS_RM = 32;
constexpr int WS_PLANES = 2 * WS_RM * PLP;             // one slot: two fp16 planes of a 32-row tile (528-byte pitch)
constexpr int WS_STP = 144;                            // staging pitch: 32 fp32 + 16 B
constexpr int WS_LDS = 2 * WS_PLANES + 8 * WS_RM * WS_STP + 2 * WS_RM * (int)sizeof(int);

struct PyrWsParams {
  const float* feat;
  const bf16_t* W[2];     // weight planes of product 0 (value) / 1 (G)
  const float* bias[2];
  float* out[2];
  int sw[2], n[2], slots[2];
  long rows;
  unsigned char slot_job[32], slot_idx[32];
};

__global__ __launch_bounds__(NT, 1) void pyramid_ws_f32h_kernel(PyrWsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, rl = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int job = p.slot_job[slot];
  const int ncol = p.n[job], ld = ncol;
  const bool has_cols = 32 * w < ncol;
  char* stage = smem + 2 * WS_PLANES + w * WS_RM * WS_STP;
  int* rs = reinterpret_cast<int*>(smem + 2 * WS_PLANES + 8 * WS_RM * WS_STP);       // [slot][row]: s_row
  const int rot = (w * 3 + (job ? 7 : 0)) & 15;
  // the weight planes of this wavefront's column block, in the order its k loop walks them: wr[i] = fragment (i + rot) & 15
  f32x4 wr[16][2];
  {
    const bf16_t* wp = frag_ptr(p.W[job], 0, has_cols ? w : 0, 16, lane);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kq = (i + rot) & 15;
      wr[i][0] = *reinterpret_cast<const f32x4*>(wp + kq * 1024);
      wr[i][1] = *reinterpret_cast<const f32x4*>(wp + 65536 + kq * 1024);
    }
  }
  const float bias = (p.bias[job] && has_cols) ? p.bias[job][32 * w + rl] : 0.f;
  const int sw = p.sw[job];
  const long ntiles = (p.rows + WS_RM - 1) / WS_RM;
  const long first = p.slot_idx[slot] * 8 + xcd, G = (long)p.slots[job] * 8;
  float* outp = p.out[job];

  // rows of a tile: wavefront w takes rows 4 w .. 4 w + 3, 16 lanes per row; chunk i of a lane = columns 64 i + 4 (lane & 15) .. + 3 (one
  // load instruction = the i-th 256-byte quarter of four rows).  A lane holds 16 values of ONE row: the row maximum is a local
  // maximum + four DPP steps over the row's 16 lanes (the 64-lane butterfly of the tiled kernel, per row, was 6 cross-lane steps).
  // (two register sets -- rows in flight for two tiles instead of one -- measured the same: the row loads are never waited for)
  f32x4 xa[4];
  const int xrow = 4 * w + (lane >> 4), xcol = (lane & 15) * 4;
  auto load_rows = [&](long tile, f32x4 (&x)[4]) {
    const long r0 = min(tile, ntiles - 1) * WS_RM;
    const float* src = p.feat + min(r0 + xrow, p.rows - 1) * 256 + xcol;
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const f32x4*>(src + 64 * i);
  };
  int xsr = 0;
  auto split_scale = [&](const f32x4 (&x)[4], int ps) {   // s_row of this lane's row (the row maximum's exponent)
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) m = fmaxf(m, fmaxf(fmaxf(fabsf(x[i][0]), fabsf(x[i][1])), fmaxf(fabsf(x[i][2]), fabsf(x[i][3]))));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xf, 0xf, false)));    // lane ^ 1
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xf, 0xf, false)));    // lane ^ 2
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xf, 0xf, false)));   // half mirror
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xf, 0xf, false)));   // mirror
    xsr = row_scale(m);
    if ((lane & 15) == 0) rs[ps * WS_RM + xrow] = xsr;
  };
  auto split_row = [&](const f32x4 (&x)[4], int i, int ps) {     // chunk i -> two fp16 parts -> plane slot `ps`
    f32x4 xs;
#pragma unroll
    for (int t = 0; t < 4; ++t) xs[t] = __builtin_ldexpf(x[i][t], xsr);
    uint2 ph, pl;
    split4_h2(xs, ph, pl);
    char* dst = smem + ps * WS_PLANES + xrow * PLP + (64 * i + xcol) * 2;
    *reinterpret_cast<uint2*>(dst) = ph;
    *reinterpret_cast<uint2*>(dst + WS_RM * PLP) = pl;
  };
  // stores of a finished tile from the wavefront's staging rows: instruction i writes rows 8 i .. 8 i + 7, 128 bytes each
  f32x4 sv[4];
  auto st_read = [&](long tile, int i) {
    const int last_row = (int)(p.rows - 1 - tile * WS_RM);
    sv[i] = *reinterpret_cast<const f32x4*>(stage + min(8 * i + (lane >> 3), last_row) * WS_STP + (lane & 7) * 16);
  };
  auto st_store = [&](long tile, int i) {
    const int last_row = (int)(p.rows - 1 - tile * WS_RM);
    *reinterpret_cast<f32x4*>(outp + (tile * WS_RM + min(8 * i + (lane >> 3), last_row)) * ld + 32 * w + (lane & 7) * 4) = sv[i];
  };

  long tile = first;
  if (tile >= ntiles) return;
  load_rows(tile, xa);
  split_scale(xa, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) split_row(xa, i, 0);
  load_rows(tile + G, xa);
  int ps = 0, k = 0;
  long prev = -1;
  if (!has_cols) {
    // a wavefront without columns (the last two of a 192-column product) only moves and splits its share of the rows
#pragma unroll 1
    for (; tile < ntiles; tile += G, ps ^= 1) {
      __syncthreads();
      if (tile + G < ntiles) {
        split_scale(xa, ps ^ 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) split_row(xa, i, ps ^ 1);
        load_rows(tile + 2 * G, xa);
      }
    }
    return;
  }
  // One tile; WITH_PREV: the previous tile's stores ride on this k loop; HAS_NEXT: the next tile's rows (requested a tile ago) are
  // split under it and their registers re-used for the tile after.  Compile-time cases instead of branches: the k loop is ONE
  // basic block, pinned step by step.
  auto tile_body = [&](auto with_prev, auto has_next_t, const long t) {
    constexpr bool WITH_PREV = decltype(with_prev)::value, HAS_NEXT = decltype(has_next_t)::value;
    f32x4 (&x)[4] = xa;
    PWSTAMP(0);
    __syncthreads();            // planes + scales of this tile are complete; slot ps ^ 1 is no longer read
    PWSTAMP(1);
    const char* arow = smem + ps * WS_PLANES + rl * PLP + 16 * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 a[2][2];
    int4 srv[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kq = (i + rot) & 15;
      a[i][0] = *reinterpret_cast<const f32x4*>(arow + kq * 32);
      a[i][1] = *reinterpret_cast<const f32x4*>(arow + WS_RM * PLP + kq * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const half8 ah = __builtin_bit_cast(half8, a[i & 1][0]), al = __builtin_bit_cast(half8, a[i & 1][1]);
      const half8 bh = __builtin_bit_cast(half8, wr[i][0]), bl = __builtin_bit_cast(half8, wr[i][1]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);          // smallest terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      if (i + 2 < 16) {
        const int kq = (i + 2 + rot) & 15;
        a[i & 1][0] = *reinterpret_cast<const f32x4*>(arow + kq * 32);
        a[i & 1][1] = *reinterpret_cast<const f32x4*>(arow + WS_RM * PLP + kq * 32);
      }
      // fillers in the shadow of this step's MFMAs
      if (WITH_PREV) {
        if (i == 0) { st_read(prev, 0); st_read(prev, 1); }
        if (i == 1) { st_read(prev, 2); st_read(prev, 3); }
        if (i == 2) { st_store(prev, 0); st_store(prev, 1); }
        if (i == 3) { st_store(prev, 2); st_store(prev, 3); }
      }
      if (HAS_NEXT) {
        if (i == 5) split_scale(x, ps ^ 1);
        if (i >= 6 && i < 10) split_row(x, i - 6, ps ^ 1);
        if (i == 11) load_rows(t + 2 * G, x);
      }
      if (i == 13) {
#pragma unroll
        for (int g = 0; g < 4; ++g) srv[g] = *reinterpret_cast<const int4*>(rs + ps * WS_RM + 8 * g + 4 * h);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (i == 3) PWSTAMP(2);
      if (i == 4) PWSTAMP(3);
      if (i == 9) PWSTAMP(4);
      if (i == 11) PWSTAMP(5);
    }
    PWSTAMP(6);
    // epilogue: acc[4 g + t] = 2^(s_row + s_w) out[row 8 g + 4 h + t][column rl] -> staging, un-scaled, + bias
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int se[4] = {srv[g].x, srv[g].y, srv[g].z, srv[g].w};
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        *reinterpret_cast<float*>(stage + (8 * g + 4 * h + tt) * WS_STP + rl * 4) = __builtin_ldexpf(acc[4 * g + tt], -(se[tt] + sw)) + bias;
    }
    PWSTAMP(7);
    prev = t;
    ps ^= 1;
    ++k;
  };
  auto step = [&](auto wp, auto hn, const long t) { tile_body(wp, hn, t); };
  if (tile + G < ntiles) {
    step(std::false_type{}, std::true_type{}, tile);
    tile += G;
#pragma unroll 1
    for (; tile + G < ntiles; tile += G) step(std::true_type{}, std::true_type{}, tile);
    step(std::true_type{}, std::false_type{}, tile);
  } else {
    step(std::false_type{}, std::false_type{}, tile);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { st_read(prev, i); st_store(prev, i); }
}

// ------------------------------------------------------------------------------------------------------------------
// chain A on two-part fp16 operands ("f32h"; 32-row tiles, two workgroups per CU; round 4).  Every product as three fp16 MFMAs.  Between the stages the activations need a scale per ROW, and a row's 256 columns are spread over
// the 8 wavefronts: each wavefront leaves the maximum of its 32 columns in an LDS table in front of the barrier that frees the planes
// every lane then reads the 8 partials of its row -- no extra barrier.  attn
// (stored for chain B) and the inputs of the 3-output layer are taken from the fp32 registers, not from the 22-bit planes: attn is
// exactly the fp32 result of its stage, the last layer's partial sums go through an LDS table summed in wavefront order.
template <int MT>
__global__ __launch_bounds__(NT, MT == 1 ? 4 : 2) void chain_a_f32h_small_kernel(const float* __restrict__ samp, const uint8_t* __restrict__ inside,
                                                                   const bf16_t* __restrict__ Wp, const float* __restrict__ bp,
                                                                   const bf16_t* __restrict__ W0, const float* __restrict__ b0,
                                                                   const bf16_t* __restrict__ W1, const float* __restrict__ b1,
                                                                   const float* __restrict__ W2, const float* __restrict__ b2,
                                                                   int swp, int sw0, int sw1, float* __restrict__ attn, float* __restrict__ o,
                                                                   const int* __restrict__ order, const float* __restrict__ o_masked, int R) {
  constexpr int RMS = 32 * MT, SPLANE = RMS * PLP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;                                                   // 2 fp16 planes
  int* rid = reinterpret_cast<int*>(smem + 2 * SPLANE);
  int* keepf = rid + RMS;
  int* rs = keepf + RMS;                                              // s_row of the planes' rows
  float* pm = reinterpret_cast<float*>(rs + RMS);                     // [row][wavefront]: partial row maxima
  float* w2l = pm + RMS * 8;
  float* po = reinterpret_cast<float*>(act);                          // [row][wavefront][4]: partial outputs, over the planes once they are dead
  float* bias_l = w2l + 768;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), rl = lane & 31, h = lane >> 5;
  const int r0 = blockIdx.x * RMS;
  const int rot = (w * 3) & 15;
  for (int i = tid; i < 768; i += NT) {
    w2l[i] = W2[i];
    bias_l[i] = (i < 256 ? bp : i < 512 ? b0 : b1)[i & 255];
  }
  bool mine = false;
  if (tid < RMS) {
    const int slot = r0 + tid;
    const int g = slot < R ? (order ? order[slot] : slot) : -1;
    rid[tid] = g;
    mine = g >= 0 && inside[g] != 0;
    keepf[tid] = mine ? 1 : 0;
  }
  const bool any_inside = __syncthreads_or(mine) != 0;
  if (!any_inside && o_masked) {
#pragma unroll
    for (int i = 0; i < 4 * MT; ++i) {
      const int c = i * NT + tid, g = rid[c >> 6];
      if (g >= 0) *reinterpret_cast<f32x4*>(attn + (long)g * 256 + (c & 63) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < RMS && rid[tid] >= 0) {
      float* og = o + (long)rid[tid] * 3;
      og[0] = o_masked[0];
      og[1] = o_masked[1];
      og[2] = o_masked[2];
    }
    return;
  }
  {
    // chunk i of this thread = 4 columns of row 8 i + w: one wavefront holds a whole row -> its maximum by a butterfly
    f32x4 x[4 * MT];
#pragma unroll
    for (int i = 0; i < 4 * MT; ++i) {
      const int c = i * NT + tid;
      x[i] = *reinterpret_cast<const f32x4*>(samp + (long)max(rid[c >> 6], 0) * 256 + (c & 63) * 4);
    }
#pragma unroll
    for (int i = 0; i < 4 * MT; ++i) {
      const int row = 8 * i + w;
      if (rid[row] < 0) x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      float m = fmaxf(fmaxf(fabsf(x[i][0]), fabsf(x[i][1])), fmaxf(fabsf(x[i][2]), fabsf(x[i][3])));
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
      const int sr = row_scale(m);
      f32x4 xs;
#pragma unroll
      for (int t = 0; t < 4; ++t) xs[t] = __builtin_ldexpf(x[i][t], sr);
      uint2 ph, pl;
      split4_h2(xs, ph, pl);
      *reinterpret_cast<uint2*>(act + row * PLP + lane * 8) = ph;
      *reinterpret_cast<uint2*>(act + SPLANE + row * PLP + lane * 8) = pl;
      if (lane == 0) rs[row] = sr;
    }
  }
  bool keep[MT];
  int grow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    keep[mt] = keepf[32 * mt + rl] != 0;
    grow[mt] = rid[32 * mt + rl];
  }
  __syncthreads();
  f32x16 acc[MT];
  f32x4 bvr[4], y[MT][4];
  const bf16_t* wps[3] = {frag_ptr(Wp, 0, w, 16, lane), frag_ptr(W0, 0, w, 16, lane), frag_ptr(W1, 0, w, 16, lane)};
  const int sws[3] = {swp, sw0, sw1};
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    stage_h2<MT, 16, PLP, 2>(act, SPLANE, 0, wps[st], 65536, acc, (rot + 5 * st) & 15, lane);
    load_bias(bias_l + 256 * st + 32 * w, bvr, lane);
    float m[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int un = -(rs[32 * mt + rl] + sws[st]);
      m[mt] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float v = __builtin_ldexpf(acc[mt][4 * g + t], un) + bvr[g][t];
          y[mt][g][t] = st == 0 ? (keep[mt] ? v : 0.f) : fmaxf(v, 0.f);
          m[mt] = fmaxf(m[mt], fabsf(y[mt][g][t]));
        }
      if (st == 0 && grow[mt] >= 0) {
        // attn rows -> global, exactly the fp32 values (16 bytes per lane and column group)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(attn + (long)grow[mt] * 256 + 32 * w + 8 * g + 4 * h) = y[mt][g];
      }
    }
    if (st < 2) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        m[mt] = fmaxf(m[mt], __shfl_xor(m[mt], 32, 64));
        if (h == 0) pm[(32 * mt + rl) * 8 + w] = m[mt];
      }
      __syncthreads();                                   // every wavefront has read the planes and rs; the partial maxima are complete
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = 32 * mt + rl;
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(pm + row * 8), p1 = *reinterpret_cast<const f32x4*>(pm + row * 8 + 4);
        const float mr = fmaxf(fmaxf(fmaxf(p0[0], p0[1]), fmaxf(p0[2], p0[3])), fmaxf(fmaxf(p1[0], p1[1]), fmaxf(p1[2], p1[3])));
        const int sr = row_scale(mr);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 xs;
#pragma unroll
          for (int t = 0; t < 4; ++t) xs[t] = __builtin_ldexpf(y[mt][g][t], sr);
          uint2 ph, pl;
          split4_h2(xs, ph, pl);
          *reinterpret_cast<uint2*>(act + row * PLP + (32 * w + 8 * g + 4 * h) * 2) = ph;
          *reinterpret_cast<uint2*>(act + SPLANE + row * PLP + (32 * w + 8 * g + 4 * h) * 2) = pl;
        }
        if (w == 0 && h == 0) rs[row] = sr;
      }
      __syncthreads();
    }
  }
  // last layer (3 outputs, dq_decoder.py:97-111's third Linear): this lane's 16 columns of its rows, the two halves of the wavefront
  // combined, the 8 wavefronts' partial sums through LDS in wavefront order
  {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float a3[3];
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) {
        float s0 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 wk = *reinterpret_cast<const f32x4*>(w2l + k3 * 256 + 32 * w + 8 * g + 4 * h);
          s0 += (y[mt][g][0] * wk[0] + y[mt][g][1] * wk[1]) + (y[mt][g][2] * wk[2] + y[mt][g][3] * wk[3]);
        }
        a3[k3] = s0 + __shfl_xor(s0, 32, 64);
      }
      if (mt == 0) __syncthreads();                      // every wavefront is through the last stage's k loop: the planes are free
      if (h == 0) *reinterpret_cast<f32x4*>(po + ((32 * mt + rl) * 8 + w) * 4) = f32x4{a3[0], a3[1], a3[2], 0.f};
    }
    __syncthreads();
    if (tid < RMS && rid[tid] >= 0) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) {
        const f32x4 pp = *reinterpret_cast<const f32x4*>(po + (tid * 8 + ww) * 4);
        t0 += pp[0];
        t1 += pp[1];
        t2 += pp[2];
      }
      float* og = o + (long)rid[tid] * 3;
      og[0] = t0 + b2[0];
      og[1] = t1 + b2[1];
      og[2] = t2 + b2[2];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// chain B: one workgroup = 4 person-queries x 15 joints (60 token rows in a 64-row tile).
// Row statistics (LayerNorm, class head) live in the accumulator layout -- lane (rl, h) of wavefront w holds columns
// 32 w + 8 g + 4 h + t of rows rl and 32 + rl -- and are completed across the 8 wavefronts through a (64 x 8) LDS table
// summed in wavefront order by every reader (a fixed order: results do not depend on timing).
__device__ __forceinline__ float row_total(const float* __restrict__ part, int row) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(part + row * 8), b = *reinterpret_cast<const f32x4*>(part + row * 8 + 4);
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
}

// x[mt][g] (4 columns each) of rows 32 mt + rl -> LayerNorm over the 256 columns of the row, in place
template <int MT>
__device__ __forceinline__ void layernorm_rows(f32x4 (&x)[MT][4], const float* __restrict__ gamma_cb, const float* __restrict__ beta_cb,
                                               float* __restrict__ part, float* __restrict__ part2, int lane, int w) {
  const int rl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) v += (x[mt][g][0] + x[mt][g][1]) + (x[mt][g][2] + x[mt][g][3]);
    v += __shfl_xor(v, 32, 64);
    if (h == 0) part[(mt * 32 + rl) * 8 + w] = v;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float mean = row_total(part, mt * 32 + rl) * (1.f / 256.f);
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      x[mt][g] = x[mt][g] - mean;
      v += (x[mt][g][0] * x[mt][g][0] + x[mt][g][1] * x[mt][g][1]) + (x[mt][g][2] * x[mt][g][2] + x[mt][g][3] * x[mt][g][3]);
    }
    v += __shfl_xor(v, 32, 64);
    if (h == 0) part2[(mt * 32 + rl) * 8 + w] = v;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float rstd = 1.f / sqrtf(row_total(part2, mt * 32 + rl) * (1.f / 256.f) + 1e-5f);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma_cb + 8 * g + 4 * h), be = *reinterpret_cast<const f32x4*>(beta_cb + 8 * g + 4 * h);
      x[mt][g] = x[mt][g] * rstd * ga + be;
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// chain B on two-part fp16 operands ("f32h"; round 4): 32-row tiles (2 persons), 71 KB of LDS -- two workgroups per CU, which the
// six-product form's 32-row tile (101 KB) could not have.  Every plane write is preceded by the
// row-maximum exchange of chain_a_f32h_small_kernel (partials in an LDS table in front of a barrier that is there anyway).  The FFN's
// hidden activations get a scale per (row, 256-column chunk): each chunk's second GEMM starts from a zero accumulator and its result is
// un-scaled and added to the fp32 sum in chunk order.  t1 (the residual of the FFN) stays in registers: planes hold 22 bits.
__device__ __forceinline__ float absmax16(const f32x4 (&v)[4]) {
  float m = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[g][0]), fabsf(v[g][1])), fmaxf(fabsf(v[g][2]), fabsf(v[g][3]))));
  return fmaxf(m, __shfl_xor(m, 32, 64));
}
__device__ __forceinline__ int gather_scale(const float* __restrict__ pm, int row) {
  const f32x4 p0 = *reinterpret_cast<const f32x4*>(pm + row * 8), p1 = *reinterpret_cast<const f32x4*>(pm + row * 8 + 4);
  return row_scale(fmaxf(fmaxf(fmaxf(p0[0], p0[1]), fmaxf(p0[2], p0[3])), fmaxf(fmaxf(p1[0], p1[1]), fmaxf(p1[2], p1[3]))));
}
// this lane's 16 values of row `row` (columns col0 + 8 g .. + 3) -> the two planes, scaled by 2^sr
__device__ __forceinline__ void write_row_h2(char* __restrict__ planes, int plane_bytes, int row, int col0, const f32x4 (&v)[4], int sr) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 xs;
#pragma unroll
    for (int t = 0; t < 4; ++t) xs[t] = __builtin_ldexpf(v[g][t], sr);
    uint2 ph, pl;
    split4_h2(xs, ph, pl);
    *reinterpret_cast<uint2*>(planes + row * PLP + (col0 + 8 * g) * 2) = ph;
    *reinterpret_cast<uint2*>(planes + plane_bytes + row * PLP + (col0 + 8 * g) * 2) = pl;
  }
}

// MT = 1: 32-row tiles, two workgroups per CU (128 registers, fragment ring 2).  MT = 2: 64-row tiles, one workgroup per CU (135 KB, 256
// registers, ring 4) -- half the fragment loads per row, six MFMAs per wavefront and k-step.  Same chunks, same order per row: bit-identical.
template <int MT>
__global__ __launch_bounds__(NT, MT == 1 ? 4 : 2) void chain_b_f32h_kernel(
    const float* __restrict__ attn, int V, const float* __restrict__ tgt, const bf16_t* __restrict__ Wu, int su,
    const float* __restrict__ bu, const float* __restrict__ g2, const float* __restrict__ be2,
    const bf16_t* __restrict__ W1, int s1, const float* __restrict__ b1, const bf16_t* __restrict__ W2, int s2,
    const float* __restrict__ b2, const float* __restrict__ g3, const float* __restrict__ be3,
    const float* __restrict__ Wc, const float* __restrict__ bc, float threshold, const uint8_t* __restrict__ forced,
    float* __restrict__ tgt_out, float* __restrict__ prob, uint8_t* __restrict__ valid, int* __restrict__ any_valid,
    const float* __restrict__ qpos, const bf16_t* __restrict__ Wn, int sn, const float* __restrict__ bn,
    float* __restrict__ xw_next, int n_next, int rows, int J, int nq_total, int has_ffn) {
  constexpr int RMT = 32 * MT, APL = RMT * PLP, RB = MT == 1 ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;                                   // 2 planes x 32 rows x 256 columns: mean, then t1, then tgt' + query_pos
  char* hb = smem + 2 * APL;                          // 2 planes x 32 rows x 256 columns: FFN hidden chunk
  float* part = reinterpret_cast<float*>(hb + 2 * APL);
  float* part2 = part + RMT * 8;
  float* pm = part2 + RMT * 8;                        // partial row maxima [row][wavefront]
  float* pr = pm + RMT * 8;                           // per-row class probabilities (RMT x 2)
  int* rs = reinterpret_cast<int*>(pr + RMT * 2);     // s_row of act's rows
  int* rsh = rs + RMT;                                // s_row of the hidden chunk's rows
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), rl = lane & 31, h = lane >> 5;
  const int qpt = RMT / J, rpt = qpt * J;
  const int q0 = blockIdx.x * qpt, r0 = q0 * J;
  const int nrow = min(rpt, rows - r0);
  const int rot = (w * 3) & 15;
  const int colb = 32 * w, col0 = colb + 4 * h;       // the wavefront's column block; this lane's first column

  f32x4 tg[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      tg[mt][g] = *reinterpret_cast<const f32x4*>(tgt + (long)(r0 + min(32 * mt + rl, nrow - 1)) * 256 + col0 + 8 * g);

  // ---- mean over views (dq_decoder.py:770) -> planes; chunk i of a thread = 4 columns of row 8 i + w (a wavefront holds a whole row)
  {
    constexpr int NCH = RMT * 64 / NT;
    f32x4 sacc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) sacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int v = 0; v < V; v += 2) {
      f32x4 xv[2][NCH];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          xv[u][i] = *reinterpret_cast<const f32x4*>(attn + ((long)min(v + u, V - 1) * rows + r0 + min(8 * i + w, nrow - 1)) * 256 + lane * 4);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        sacc[i] += xv[0][i];
        if (v + 1 < V) sacc[i] += xv[1][i];
      }
    }
    const float Vf = (float)V;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = 8 * i + w;
      f32x4 m4 = {sacc[i][0] / Vf, sacc[i][1] / Vf, sacc[i][2] / Vf, sacc[i][3] / Vf};
      if (row >= nrow) m4 = f32x4{0.f, 0.f, 0.f, 0.f};
      float m = fmaxf(fmaxf(fabsf(m4[0]), fabsf(m4[1])), fmaxf(fabsf(m4[2]), fabsf(m4[3])));
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
      const int sr = row_scale(m);
      f32x4 xs;
#pragma unroll
      for (int t = 0; t < 4; ++t) xs[t] = __builtin_ldexpf(m4[t], sr);
      uint2 ph, pl;
      split4_h2(xs, ph, pl);
      *reinterpret_cast<uint2*>(act + row * PLP + lane * 8) = ph;
      *reinterpret_cast<uint2*>(act + APL + row * PLP + lane * 8) = pl;
      if (lane == 0) rs[row] = sr;
    }
  }
  __syncthreads();

  // ---- t1 = LN2(tgt + feature_update_mlp(mean))   (dq_decoder.py:773-778)
  f32x16 acc[MT];
  f32x4 bvr[4], t1[MT][4];
  stage_h2<MT, 16, PLP, RB>(act, APL, 0, frag_ptr(Wu, 0, w, 16, lane), 65536, acc, rot, lane);
  load_bias(bu + colb, bvr, lane);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int un = -(rs[32 * mt + rl] + su);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int t = 0; t < 4; ++t) t1[mt][g][t] = __builtin_ldexpf(acc[mt][4 * g + t], un) + bvr[g][t];
      if (32 * mt + rl < nrow) t1[mt][g] += tg[mt][g];
    }
  }
  layernorm_rows<MT>(t1, g2 + colb, be2 + colb, part, part2, lane, w);     // (its first barrier: every wavefront is done reading `act` and rs)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float m = absmax16(t1[mt]);
    if (h == 0) pm[(32 * mt + rl) * 8 + w] = m;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = 32 * mt + rl, sr = gather_scale(pm, row);
    write_row_h2(act, APL, row, col0, t1[mt], sr);
    if (w == 0 && h == 0) rs[row] = sr;
  }
  __syncthreads();

  f32x4 y[MT][4];
  if (has_ffn) {
    // ---- FFN (mvp_decoder.py:94-98): Y = sum over the 4 hidden chunks of relu(t1 W1^T + b1)[chunk] W2[:, chunk]^T
    f32x4 ysum[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) ysum[mt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wp2 = frag_ptr(W2, 0, w, 64, lane);
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      f32x4 hid[MT][4];
      stage_h2<MT, 16, PLP, RB>(act, APL, 0, frag_ptr(W1, c, w, 16, lane), 1024 * 256, acc, (w * 5) & 15, lane);
      load_bias(b1 + c * 256 + colb, bvr, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int un1 = -(rs[32 * mt + rl] + s1);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int t = 0; t < 4; ++t) hid[mt][g][t] = fmaxf(__builtin_ldexpf(acc[mt][4 * g + t], un1) + bvr[g][t], 0.f);
        const float m = absmax16(hid[mt]);
        if (h == 0) pm[(32 * mt + rl) * 8 + w] = m;
      }
      __syncthreads();                                              // the previous chunk's second GEMM has read hb and rsh; maxima complete
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = 32 * mt + rl, sr = gather_scale(pm, row);
        write_row_h2(hb, APL, row, col0, hid[mt], sr);
        if (w == 0 && h == 0) rsh[row] = sr;
      }
      __syncthreads();
      stage_h2<MT, 16, PLP, RB>(hb, APL, 0, wp2 + (long)c * 16 * 1024, 256 * 1024, acc, rot, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int un2 = -(rsh[32 * mt + rl] + s2);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int t = 0; t < 4; ++t) ysum[mt][g][t] += __builtin_ldexpf(acc[mt][4 * g + t], un2);
      }
    }
    load_bias(b2 + colb, bvr, lane);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) y[mt][g] = (ysum[mt][g] + bvr[g]) + t1[mt][g];
    layernorm_rows<MT>(y, g3 + colb, be3 + colb, part, part2, lane, w);
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) y[mt][g] = t1[mt][g];
  }

  // ---- tgt' -> global; class head (dq_decoder.py:889-893): per-row logits, completed across the wavefronts
  {
    float c0[MT], c1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (32 * mt + rl < nrow) *reinterpret_cast<f32x4*>(tgt_out + (long)(r0 + 32 * mt + rl) * 256 + col0 + 8 * g) = y[mt][g];
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wc + col0 + 8 * g);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(Wc + 256 + col0 + 8 * g);
        a0 += (y[mt][g][0] * w0[0] + y[mt][g][1] * w0[1]) + (y[mt][g][2] * w0[2] + y[mt][g][3] * w0[3]);
        a1 += (y[mt][g][0] * w1[0] + y[mt][g][1] * w1[1]) + (y[mt][g][2] * w1[2] + y[mt][g][3] * w1[3]);
      }
      c0[mt] = a0 + __shfl_xor(a0, 32, 64);
      c1[mt] = a1 + __shfl_xor(a1, 32, 64);
    }
    __syncthreads();                                                // part / part2 of the last LayerNorm have been read
    if (h == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        part[(32 * mt + rl) * 8 + w] = c0[mt];
        part2[(32 * mt + rl) * 8 + w] = c1[mt];
      }
    }
    // the maxima of tgt' + query_pos for the next layer's query term ride on the same barrier
    f32x4 z[MT][4];
    if (Wn) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          z[mt][g] = y[mt][g];
          if (qpos) z[mt][g] += *reinterpret_cast<const f32x4*>(qpos + (long)(r0 + min(32 * mt + rl, nrow - 1)) * 256 + col0 + 8 * g);
        }
        const float m = absmax16(z[mt]);
        if (h == 0) pm[(32 * mt + rl) * 8 + w] = m;
      }
    }
    __syncthreads();
    if (tid < RMT) {
      pr[2 * tid] = 1.f / (1.f + expf(-(row_total(part, tid) + bc[0])));
      pr[2 * tid + 1] = 1.f / (1.f + expf(-(row_total(part2, tid) + bc[1])));
    }
    if (Wn) {        // every wavefront passed the barriers above: `act` and rs are free
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = 32 * mt + rl, sr = gather_scale(pm, row);
        write_row_h2(act, APL, row, col0, z[mt], sr);
        if (w == 0 && h == 0) rs[row] = sr;
      }
    }
    __syncthreads();
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
  }
  if (Wn && colb < n_next) {
    // ---- xw = (tgt' + query_pos) W_next^T + b_next: the query term of the NEXT layer's offsets / logits Linear (projattn.py:180-181)
    stage_h2<MT, 16, PLP, RB>(act, APL, 0, frag_ptr(Wn, 0, w, 16, lane), 65536, acc, (rot + 7) & 15, lane);
    load_bias(bn + colb, bvr, lane);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int un = -(rs[32 * mt + rl] + sn);
      if (32 * mt + rl < nrow) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = __builtin_ldexpf(acc[mt][4 * g + t], un) + bvr[g][t];
          *reinterpret_cast<f32x4*>(xw_next + (long)(r0 + 32 * mt + rl) * n_next + col0 + 8 * g) = v;
        }
      }
    }
  }
}

template <typename K>
int configure_lds(K kernel, size_t lds, bool (&configured)[MVG_MAX_DEVICES]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVG_MAX_DEVICES) return MVG_E_BADARG;
  if (!configured[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    configured[dev] = true;
  }
  return 0;
}

int cu_count() {
  static int cus[MVG_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVG_MAX_DEVICES) return 256;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

}  // namespace

int g_f32h_rows = 0;      // tuning knob "f32h_rows": rows per tile of the two fp32 chains (0 = by the row count | 32 | 64); both sizes sum a row identically

extern "C" int mvg_chain_update_ffn_class_f32h(const float* attn, int V, const float* tgt, const void* Wu, int wu_scale, const float* bu,
                                              const float* g2, const float* be2, const void* W1, int w1_scale, const float* b1,
                                              const void* W2, int w2_scale, const float* b2, const float* g3, const float* be3,
                                              const float* Wc, const float* bc, float threshold, const uint8_t* forced_valid,
                                              float* tgt_out, float* prob, uint8_t* valid, int* any_valid, const float* query_pos,
                                              const void* W_next, int wn_scale, const float* b_next, float* xw_next, int n_next, int B,
                                              int NQ, int J, int has_ffn, void* stream) {
  if (!attn || !tgt || !Wu || !bu || !g2 || !be2 || !Wc || !bc || !tgt_out || !prob || !valid || !any_valid) return MVG_E_BADARG;
  if (has_ffn && (!W1 || !b1 || !W2 || !b2 || !g3 || !be3)) return MVG_E_BADARG;
  if (V <= 0 || J <= 0 || J > 32 || B < 0 || NQ < 0) return MVG_E_BADARG;
  if (W_next && (!b_next || !xw_next || n_next <= 0 || n_next > 256 || n_next % 32 != 0)) return MVG_E_BADARG;
  for (int sc : {wu_scale, w1_scale, w2_scale, wn_scale})
    if (sc < -100 || sc > 100) return MVG_E_BADARG;
  const int nq_total = B * NQ, rows = nq_total * J;
  if (rows == 0) return 0;
  // both tile sizes sum a row identically: 64-row tiles (one workgroup per CU, fragment ring 4) unless they would leave a quarter of
  // the CUs without a workgroup (then 32-row tiles, two workgroups per CU)
  const int tiles64 = (nq_total + 64 / J - 1) / (64 / J);
  const bool big = J <= 32 && (g_f32h_rows == 64 || (g_f32h_rows == 0 && tiles64 > (cu_count() * 3) / 4));
  static bool configured[MVG_MAX_DEVICES] = {}, configured2[MVG_MAX_DEVICES] = {};
#define MVG_CBH(MTV, CFG)                                                                                                              \
  {                                                                                                                                    \
    constexpr int R = 32 * MTV;                                                                                                        \
    const int qpt = R / J;                                                                                                             \
    const size_t lds = 4 * R * PLP + (3 * R * 8 + R * 2) * sizeof(float) + 2 * R * sizeof(int);                                        \
    if (int rc = configure_lds(&chain_b_f32h_kernel<MTV>, lds, CFG)) return rc;                                                        \
    hipLaunchKernelGGL(chain_b_f32h_kernel<MTV>, dim3((nq_total + qpt - 1) / qpt), dim3(NT), lds, (hipStream_t)stream, attn, V, tgt,   \
                       (const bf16_t*)Wu, wu_scale, bu, g2, be2, (const bf16_t*)W1, w1_scale, b1, (const bf16_t*)W2, w2_scale, b2, g3,  \
                       be3, Wc, bc, threshold, forced_valid, tgt_out, prob, valid, any_valid, query_pos, (const bf16_t*)W_next,        \
                       wn_scale, b_next, xw_next, n_next, rows, J, nq_total, has_ffn);                                                 \
  }
  if (big) MVG_CBH(2, configured2) else MVG_CBH(1, configured)
#undef MVG_CBH
  MVG_LAUNCH_CHECK();
  return 0;
}

extern "C" int mvg_chain_attn_pose_f32h(const float* samp, const uint8_t* inside, const void* Wp, int wp_scale, const float* bp,
                                       const void* W0, int w0_scale, const float* b0, const void* W1, int w1_scale, const float* b1,
                                       const float* W2, const float* b2, float* attn, float* o, const int32_t* order,
                                       const float* o_masked, int rows, void* stream) {
  if (!samp || !inside || !Wp || !bp || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !attn || !o || rows < 0) return MVG_E_BADARG;
  for (int sc : {wp_scale, w0_scale, w1_scale})
    if (sc < -100 || sc > 100) return MVG_E_BADARG;
  if (rows == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(samp) | reinterpret_cast<uintptr_t>(attn) | reinterpret_cast<uintptr_t>(Wp) | reinterpret_cast<uintptr_t>(W0) |
       reinterpret_cast<uintptr_t>(W1) | reinterpret_cast<uintptr_t>(W2)) % 16 != 0)
    return MVG_E_BADARG;
  static bool configured[MVG_MAX_DEVICES] = {}, configured2[MVG_MAX_DEVICES] = {};
  // both tile sizes sum a row identically: 64-row tiles (half the fragment loads per row) once they fill the CUs
  if (g_f32h_rows == 64 || (g_f32h_rows == 0 && (rows + 63) / 64 >= cu_count())) {
    const size_t lds = 2 * 64 * PLP + 3 * 64 * sizeof(int) + 64 * 8 * sizeof(float) + 2 * 768 * sizeof(float);
    if (int rc = configure_lds(&chain_a_f32h_small_kernel<2>, lds, configured2)) return rc;
    hipLaunchKernelGGL(chain_a_f32h_small_kernel<2>, dim3((rows + 63) / 64), dim3(NT), lds, (hipStream_t)stream, samp, inside, (const bf16_t*)Wp,
                       bp, (const bf16_t*)W0, b0, (const bf16_t*)W1, b1, W2, b2, wp_scale, w0_scale, w1_scale, attn, o, order, o_masked, rows);
  } else {
    const size_t lds = 2 * 32 * PLP + 3 * 32 * sizeof(int) + 32 * 8 * sizeof(float) + 2 * 768 * sizeof(float);
    if (int rc = configure_lds(&chain_a_f32h_small_kernel<1>, lds, configured)) return rc;
    hipLaunchKernelGGL(chain_a_f32h_small_kernel<1>, dim3((rows + 31) / 32), dim3(NT), lds, (hipStream_t)stream, samp, inside, (const bf16_t*)Wp,
                       bp, (const bf16_t*)W0, b0, (const bf16_t*)W1, b1, W2, b2, wp_scale, w0_scale, w1_scale, attn, o, order, o_masked, rows);
  }
  MVG_LAUNCH_CHECK();
  return 0;
}

extern "C" int mvg_pyramid_f32h(const float* feat, const void* Wv_planes, int wv_scale, const float* bv, const void* Wg_planes,
                                int wg_scale, float* value, float* G, int64_t rows, int n_g, void* stream) {
  if (!feat || !Wv_planes || !Wg_planes || !value || !G || rows < 0 || n_g <= 0 || n_g > 256 || n_g % 32 != 0) return MVG_E_BADARG;
  if (wv_scale < -100 || wv_scale > 100 || wg_scale < -100 || wg_scale > 100) return MVG_E_BADARG;
  if (rows == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(value) | reinterpret_cast<uintptr_t>(G) |
       reinterpret_cast<uintptr_t>(Wv_planes) | reinterpret_cast<uintptr_t>(Wg_planes)) % 16 != 0)
    return MVG_E_BADARG;
  static bool configured_ws[MVG_MAX_DEVICES] = {};
  if (int rc = configure_lds(&pyramid_ws_f32h_kernel, WS_LDS, configured_ws)) return rc;
  {
    PyrWsParams p = {};
    p.feat = feat; p.rows = rows;
    p.W[0] = (const bf16_t*)Wv_planes; p.W[1] = (const bf16_t*)Wg_planes;
    p.bias[0] = bv; p.bias[1] = nullptr;
    p.out[0] = value; p.out[1] = G;
    p.sw[0] = wv_scale; p.sw[1] = wg_scale;
    p.n[0] = 256; p.n[1] = n_g;
    // one workgroup per CU, 32 per XCD, divided in proportion to the products' columns; never more than the tiles an XCD has
    const long ntiles_ws = (rows + WS_RM - 1) / WS_RM;
    int per_xcd = cu_count() / 8;
    if (per_xcd > 32) per_xcd = 32;
    if (per_xcd < 2) per_xcd = 2;
    const long cap = (ntiles_ws + 7) / 8 * 2;
    if (per_xcd > cap) per_xcd = (int)cap;
    int s0 = (per_xcd * 256 + (256 + n_g) / 2) / (256 + n_g);
    if (s0 < 1) s0 = 1;
    if (s0 > per_xcd - 1) s0 = per_xcd - 1;
    p.slots[0] = s0; p.slots[1] = per_xcd - s0;
    int given[2] = {0, 0}, sidx = 0;
    while (sidx < per_xcd)                       // interleave the two products over the slots
      for (int j = 0; j < 2 && sidx < per_xcd; ++j)
        if (given[j] < p.slots[j]) {
          p.slot_job[sidx] = (unsigned char)j;
          p.slot_idx[sidx] = (unsigned char)given[j]++;
          ++sidx;
        }
    hipLaunchKernelGGL(pyramid_ws_f32h_kernel, dim3(8 * per_xcd), dim3(NT), WS_LDS, (hipStream_t)stream, p);
  }
  MVG_LAUNCH_CHECK();
  return 0;
}


