// Weight-stationary bf16 GEMMs over the feature pyramid (gfx950): out = feat (n_img*S, 256) @ W^T (+ bias)
//
//   value projection : value = feat @ Wv^T + bv, written as head planes  vh[img][head 8][s][ch 32]   (projattn.py:160-175)
//   G projection     : G  = feat @ [Woff; Wattn]^T, row-major (n_img*S, 192) (projattn.py:180-181 applied to the
//                      pyramid itself: bilinear sampling commutes with the Linear, see msda_gsamp_kernel)
//
// Why weight-stationary: with K = 256 these GEMMs move ~1 KB per row and are bound by what a 256-CU chip pulls
// through its L1s (~6 TB/s); a tiled GEMM re-fetches its 64-128 KB weight tile for every row tile, which at
// 200 000 rows is as much traffic as the activations (measured: 107 us for the value projection = 510 MB, 200 MB
// of it weights).  Here 512 persistent workgroups load the weight ONCE into registers -- each of the 4
// wavefronts keeps its 64 output columns x 256 k = 32 KB as 128 VGPRs in MFMA-fragment order -- and stream
// 32-row tiles through LDS: the k-loop is pure ds_read_b128 + v_mfma_f32_32x32x16_bf16; the loads of the tile
// after next and the stores of the previous one are issued together at the start of an iteration and waited for one
// MFMA + epilogue phase later; two workgroups per CU overlap each other.
//
// Head planes (consumed by msda_gsamp_kernel): a pixel's 32 channels of one head are 64 contiguous bytes, two
// horizontally adjacent pixels one 128-byte line when the left one has an even column -- the bilinear corners of a
// sample are 2 x (16 + 16) bytes per lane.  (An earlier "pixel-pair" layout stored (value(s), value(s+1)) interleaved:
// one line per corner pair and perm-free dot2 operands, but twice the bytes -- with the pairs processed in
// image-space order the smaller footprint wins: sampler 153 -> 145 us, this GEMM 59 -> 40 us.)
#include "common.h"

int g_wreg_grid = 512;   // tuning knob (mvg_set_tuning "wreg_grid"): persistent workgroups

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
  int rowmajor;           // 0 = head planes (N = 256), 1 = row-major bf16 (M, N)
};

// NCPT: 16-byte output chunks per thread and tile.  0 = head planes (N = 256: 4 chunks), 1..4 = row-major with
// N = 64 * NCPT columns.  Every thread issues the SAME number of stores for every tile (chunks of rows past the end
// of the matrix are redirected to re-write the last valid row's chunk with identical bytes), so the compiler can
// give the loads of a later tile an exact vmcnt instead of vmcnt(0).
// The persistent loop of one workgroup: row tiles first_tile, first_tile + tile_stride, ... of job `p`.
template <int NCPT>
__device__ __forceinline__ void wreg_body(const WregParams& p, char* smem, const int first_tile, const int tile_stride) {
  constexpr bool PLANES = NCPT == 0;
  char* act = smem;                                   // RM x 256 bf16 A tile
  char* stage0 = smem + RM * ACT_PITCH;               // 2 x (RM x 256 bf16) output tiles (double-buffered)
  float* bias_s = reinterpret_cast<float*>(smem + 3 * RM * ACT_PITCH);   // 256 f32
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6, rl = lane & 31, h = lane >> 5;
  const bool wave_has_cols = wn * 64 < p.N;
  bias_s[tid] = p.bias ? p.bias[tid] : 0.f;

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

  constexpr int TS = RM;                        // tile stride in rows
  const int ntiles = (p.M + TS - 1) / TS;
  bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
  // A chunk c = i*256 + tid of a tile: row = c >> 5, 16-byte column v16 = c & 31 (NCH chunks per thread)
  auto load_chunk = [&](int tile, int i) -> uint4 {
    const int c = i * 256 + tid, row = c >> 5, v16 = c & 31;
    const int rr = min(tile, ntiles - 1) * TS;
    return *reinterpret_cast<const uint4*>(p.A + (long)min(rr + row, p.M - 1) * 256 + v16 * 8);
  };
  // global stores of a finished tile from its staging buffer
  auto store_tile = [&](int tile, const char* stage) {
    const int r0 = tile * TS;
    const int last_row = p.M - 1 - r0;                 // rows past it do not exist (only in the final tile)
    if (PLANES) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = i * 256 + tid, v16 = c & 31;
        // the 64 bytes of (pixel, head) are written by 4 threads: thread q4 stores channels 8*q4 .. 8*q4+7
        const int row = min(c >> 5, last_row);
        const int grow = r0 + row;
        const int img = grow / p.S_img, sp = grow - img * p.S_img;
        const int head = v16 >> 2, q4 = v16 & 3;
        *reinterpret_cast<f32x4*>(outp + (((long)img * 8 + head) * p.S_img + sp) * 32 + q4 * 8) =
            *reinterpret_cast<const f32x4*>(stage + row * ACT_PITCH + v16 * 16);
      }
    } else {
      constexpr int CPR = 8 * (NCPT > 0 ? NCPT : 1);    // 16-byte chunks per output row
#pragma unroll
      for (int i = 0; i < NCPT; ++i) {
        const int c = i * 256 + tid;
        const int row = min(c / CPR, last_row), ch = c % CPR;
        *reinterpret_cast<f32x4*>(outp + (long)(r0 + row) * p.N + ch * 8) =
            *reinterpret_cast<const f32x4*>(stage + row * ACT_PITCH + ch * 16);
      }
    }
  };

  // Software pipeline, loads two tiles ahead.  Iteration t: [wait loads(t)] -> A tile to LDS -> barrier ->
  // stores(t-1) from the other staging buffer -> loads(t+2) into the registers just freed -> MFMA(t) -> epilogue(t)
  // -> barrier.  gfx9 has ONE in-order vmcnt for loads and stores: loads(t+1) are older than stores(t-1) and
  // loads(t+2), and because every iteration issues a fixed number of each, the wait at the top of t+1 is an exact
  // vmcnt(stores + loads), not vmcnt(0) -- a full iteration of latency hiding for every load.
  static_assert(NCH == 4, "prefetch registers are written out for 4 chunks per thread");
  const int G = tile_stride;
  uint4 xa0 = load_chunk(first_tile, 0), xa1 = load_chunk(first_tile, 1), xa2 = load_chunk(first_tile, 2),
        xa3 = load_chunk(first_tile, 3);
  uint4 xb0 = load_chunk(first_tile + G, 0), xb1 = load_chunk(first_tile + G, 1), xb2 = load_chunk(first_tile + G, 2),
        xb3 = load_chunk(first_tile + G, 3);
  int it = 0, prev_tile = -1;

#define WREG_ITERATION(X0, X1, X2, X3)                                                                           \
  {                                                                                                              \
    {                                                                                                            \
      const int row = tid >> 5, v16 = tid & 31;                                                                  \
      *reinterpret_cast<uint4*>(act + row * ACT_PITCH + v16 * 16) = X0;                                          \
      *reinterpret_cast<uint4*>(act + (row + 8) * ACT_PITCH + v16 * 16) = X1;                                    \
      *reinterpret_cast<uint4*>(act + (row + 16) * ACT_PITCH + v16 * 16) = X2;                                   \
      *reinterpret_cast<uint4*>(act + (row + 24) * ACT_PITCH + v16 * 16) = X3;                                   \
    }                                                                                                            \
    __syncthreads();                                                                                             \
    char* stage = stage0 + (it & 1) * RM * ACT_PITCH;                                                            \
    if (prev_tile >= 0) store_tile(prev_tile, stage0 + ((it & 1) ^ 1) * RM * ACT_PITCH);                         \
    X0 = load_chunk(tile + 2 * G, 0);                                                                            \
    X1 = load_chunk(tile + 2 * G, 1);                                                                            \
    X2 = load_chunk(tile + 2 * G, 2);                                                                            \
    X3 = load_chunk(tile + 2 * G, 3);                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    f32x16 acc[2];                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
      _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;                                            \
    if (wave_has_cols) {                                                                                         \
      _Pragma("unroll") for (int ks = 0; ks < 16; ++ks) {                                                        \
        const f32x4 a = *reinterpret_cast<const f32x4*>(act + rl * ACT_PITCH + ks * 32 + 16 * h);                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[ks][j]),              \
                                                           __builtin_bit_cast(bf16x8, a), acc[j], 0, 0, 0);      \
      }                                                                                                          \
    }                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                            \
        const int nn = wn * 64 + j * 32 + 8 * g + 4 * h;                                                         \
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + nn);                                           \
        uint2 pk;                                                                                                \
        pk.x = pack_bf16(acc[j][4 * g] + bv[0], acc[j][4 * g + 1] + bv[1]);     \
        pk.y = pack_bf16(acc[j][4 * g + 2] + bv[2], acc[j][4 * g + 3] + bv[3]); \
        *reinterpret_cast<uint2*>(stage + rl * ACT_PITCH + nn * 2) = pk;                                         \
      }                                                                                                          \
    prev_tile = tile;                                                                                            \
    ++it;                                                                                                        \
    __syncthreads(); /* act may be overwritten, this tile's staging buffer is complete */                        \
  }

#pragma unroll 1
  for (int tile = first_tile; tile < ntiles; tile += 2 * G) {
    WREG_ITERATION(xa0, xa1, xa2, xa3)
    tile += G;
    if (tile >= ntiles) break;
    WREG_ITERATION(xb0, xb1, xb2, xb3)
    tile -= G;
  }
#undef WREG_ITERATION
  if (prev_tile >= 0) store_tile(prev_tile, stage0 + ((it & 1) ^ 1) * RM * ACT_PITCH);
}

template <int NCPT>
__global__ __launch_bounds__(256, 2) void wreg_gemm_kernel(WregParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wreg_body<NCPT>(p, smem, blockIdx.x, gridDim.x);
}

// ---- several products of the SAME rows in one launch (round 5) -------------------------------------------------------------
// All layers' value planes and G are products of the one packed pyramid with different weights.  Launched one after the other,
// every product streams the 103-MB pyramid through the fabric again (8 x per forward at cfg-2).  Here the workgroups of an XCD
// (block b runs on XCD b & 7 -- observed dispatch order, used for speed only) are divided among the jobs in proportion to their
// column counts; every job sweeps the XCD's row tiles (tiles t = 8 u + xcd) in the same order and at the same pace, so a tile
// fetched by the first job's workgroup is served to the others by the XCD's L2: the pyramid crosses the fabric once per launch.
// A workgroup keeps ONE job's weight slice in registers for its whole life exactly like wreg_gemm_kernel -- same loop, same k
// order, bit-identical outputs.
constexpr int WREG_MAX_JOBS = 8;
constexpr int WREG_MAX_SLOTS = 64;       // workgroups per XCD: 32 CUs x 2
struct WregJob {
  const bf16_t* Wf;
  const float* bias;
  void* out;
  int N, rowmajor;
};
struct WregGroupParams {
  const bf16_t* A;
  int M, S_img, njobs, n_slots;
  WregJob job[WREG_MAX_JOBS];
  unsigned char slot_job[WREG_MAX_SLOTS];   // job of slot s = blockIdx >> 3
  unsigned char slot_idx[WREG_MAX_SLOTS];   // index of the slot among its job's slots
  unsigned char job_slots[WREG_MAX_JOBS];   // slots per XCD of job j
};

__global__ __launch_bounds__(256, 2) void wreg_group_kernel(WregGroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int j = gp.slot_job[slot];
  WregParams p;
  p.A = gp.A; p.M = gp.M; p.S_img = gp.S_img;
  p.Wf = gp.job[j].Wf; p.bias = gp.job[j].bias; p.out = gp.job[j].out; p.N = gp.job[j].N; p.rowmajor = gp.job[j].rowmajor;
  const int first = gp.slot_idx[slot] * 8 + xcd, stride = gp.job_slots[j] * 8;
  if (!p.rowmajor) wreg_body<0>(p, smem, first, stride);
  else wreg_body<3>(p, smem, first, stride);
}

int launch_wreg(const WregParams& p, hipStream_t st) {
  const size_t lds = 3 * RM * ACT_PITCH + 256 * sizeof(float);
  const int TS = RM;
  const int ntiles = (p.M + TS - 1) / TS;
  const int grid = ntiles < g_wreg_grid ? ntiles : g_wreg_grid;      // persistent: 2 workgroups per CU
#define WREG_LAUNCH(NC) hipLaunchKernelGGL((wreg_gemm_kernel<NC>), dim3(grid), dim3(256), lds, st, p)
  if (!p.rowmajor) WREG_LAUNCH(0);
  else switch (p.N / 64) {
    case 1: WREG_LAUNCH(1); break;
    case 2: WREG_LAUNCH(2); break;
    case 3: WREG_LAUNCH(3); break;
    default: WREG_LAUNCH(4); break;
  }
#undef WREG_LAUNCH
  MVG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int mvg_value_proj_planes_ws(const void* feat, const void* Wf, const float* bias, void* vp, int n_img, int S,
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

int g_wreg_gweight = 300;   // tuning knob "wreg_gweight": share of an XCD's workgroups a 192-column job gets, x100 (a 256-column job: 400)

extern "C" int mvg_pyramid_group_ws(const void* feat, int n_img, int S, int njobs, const void* const* Wf, const float* const* bias,
                                    void* const* out, const int* N, const int* planes, int slots_per_xcd, void* stream) {
  if (!feat || !Wf || !bias || !out || !N || !planes || n_img <= 0 || S <= 0 || njobs < 1 || njobs > WREG_MAX_JOBS) return MVG_E_BADARG;
  WregGroupParams gp = {};
  gp.A = (const bf16_t*)feat; gp.M = n_img * S; gp.S_img = S; gp.njobs = njobs;
  int weight[WREG_MAX_JOBS], total = 0;
  for (int j = 0; j < njobs; ++j) {
    if (!Wf[j] || !out[j]) return MVG_E_BADARG;
    if (planes[j] ? (N[j] != 256 || !bias[j]) : N[j] != 192) return MVG_E_BADARG;     // the two shapes the decoder has
    gp.job[j].Wf = (const bf16_t*)Wf[j]; gp.job[j].bias = bias[j]; gp.job[j].out = out[j];
    gp.job[j].N = N[j]; gp.job[j].rowmajor = planes[j] ? 0 : 1;
    weight[j] = planes[j] ? 400 : g_wreg_gweight;
    total += weight[j];
  }
  // slots per XCD in proportion to the jobs' work; never more than the 64 resident workgroups of an XCD
  const int ntiles = (gp.M + RM - 1) / RM;
  int budget = slots_per_xcd > 0 ? slots_per_xcd : g_wreg_grid / 8;
  if (budget > WREG_MAX_SLOTS) budget = WREG_MAX_SLOTS;
  if (budget > (ntiles + 7) / 8 * njobs) budget = (ntiles + 7) / 8 * njobs;
  if (budget < njobs) budget = njobs;
  int n_slots = 0;
  for (int j = 0; j < njobs; ++j) {
    int c = budget * weight[j] / total;
    if (c < 1) c = 1;
    gp.job_slots[j] = (unsigned char)c;
    n_slots += c;
  }
  if (n_slots > WREG_MAX_SLOTS) return MVG_E_BADARG;
  // interleave the jobs over the slots (slot s and s + 32 tend to share a CU)
  int given[WREG_MAX_JOBS] = {0}, s = 0;
  while (s < n_slots)
    for (int j = 0; j < njobs && s < n_slots; ++j)
      if (given[j] < gp.job_slots[j]) {
        gp.slot_job[s] = (unsigned char)j;
        gp.slot_idx[s] = (unsigned char)given[j]++;
        ++s;
      }
  gp.n_slots = n_slots;
  const size_t lds = 3 * RM * ACT_PITCH + 256 * sizeof(float);
  hipLaunchKernelGGL(wreg_group_kernel, dim3(8 * n_slots), dim3(256), lds, (hipStream_t)stream, gp);
  MVG_LAUNCH_CHECK();
  return 0;
}
