// Weight-stationary bf16 GEMMs over the feature pyramid (gfx950): out = feat (n_img*S, 256) @ W^T (+ bias)
//
//   value projection : value = feat @ Wv^T + bv, written as head planes  vh[img][head 8][s][ch 32]   (projattn.py:160-175)
//   G projection     : G  = feat @ [Woff; Wattn]^T, row-major (n_img*S, 192) (projattn.py:180-181 applied to the
//                      pyramid itself: bilinear sampling commutes with the Linear, see msda_gsamp_kernel)
//
// Why weight-stationary: with K = 256 these GEMMs move ~1 KB per row and are bound by what the chip streams (HBM writes,
// ~5.5 TB/s in the forward); a tiled GEMM re-fetches its 64-128 KB weight tile for every row tile, which at 200 000 rows is as
// much traffic as the activations (measured: 107 us for the value projection = 510 MB, 200 MB of it weights).  Here 512
// persistent workgroups load the weight ONCE into registers -- each of the 4 wavefronts keeps its 64 output columns x 256 k =
// 32 KB as 128 VGPRs in MFMA-fragment order -- and stream 32-row tiles through LDS (wreg2_body below: LDS-DMA ring for the
// rows, wavefront-private staging for the outputs, one barrier and one exact vmcnt wait per tile); two workgroups per CU.
//
// Head planes (consumed by msda_gsamp_kernel): a pixel's 32 channels of one head are 64 contiguous bytes, two
// horizontally adjacent pixels one 128-byte line when the left one has an even column -- the bilinear corners of a
// sample are 2 x (16 + 16) bytes per lane.  (An earlier "pixel-pair" layout stored (value(s), value(s+1)) interleaved:
// one line per corner pair and perm-free dot2 operands, but twice the bytes -- with the pairs processed in
// image-space order the smaller footprint wins: sampler 153 -> 145 us, this GEMM 59 -> 40 us.)
#include "common.h"
#include <type_traits>

int g_wreg_grid = 512;   // tuning knob (mvg_set_tuning "wreg_grid"): persistent workgroups

// measurement builds (tools/probes/stamps_wreg.py): s_memtime per wavefront at the phase boundaries of one steady-state tile
#ifdef WREG_STAMPS
__device__ unsigned long long wreg_stamps[1024 * 4 * 16];
#define W2STAMP_ALWAYS(i, val)                                                                                            \
  do {                                                                                                                    \
    if (lane == 0 && blockIdx.x < 1024) {                                                                                 \
      asm volatile("" ::: "memory");                                                                                      \
      wreg_stamps[(blockIdx.x * 4 + wn) * 16 + (i)] = (val);                                                              \
      asm volatile("" ::: "memory");                                                                                      \
    }                                                                                                                     \
  } while (0)
#define W2STAMP(i)                                                                                                        \
  do {                                                                                                                    \
    if (it == WREG_STAMPS && lane == 0 && blockIdx.x < 1024) {                                                            \
      asm volatile("" ::: "memory");                                                                                      \
      wreg_stamps[(blockIdx.x * 4 + wn) * 16 + (i)] = __builtin_amdgcn_s_memtime();                                       \
      asm volatile("" ::: "memory");                                                                                      \
    }                                                                                                                     \
  } while (0)
extern "C" int mvg_wreg_read_stamps(unsigned long long* host, int n_blocks) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(wreg_stamps), sizeof(unsigned long long) * 64 * n_blocks);
}
#else
#define W2STAMP(i)
#define W2STAMP_ALWAYS(i, val)
#endif

namespace {

constexpr int RM = 32;            // rows per tile (one 32x32 MFMA row block)

struct WregParams {
  const bf16_t* A;        // (M, 256) bf16 rows
  const bf16_t* Wf;       // swizzled weight fragments [wn 4][ks 16][j 2][lane 64][8] (N padded to 256 with zeros)
  const float* bias;      // (256) f32 or nullptr
  void* out;
  int M, N, S_img;
  int rowmajor;           // 0 = head planes (N = 256), 1 = row-major bf16 (M, N)
};

// ---- the persistent loop of one workgroup: row tiles first_tile, first_tile + tile_stride, ... of one product -------------------
// Rounds 1-4 staged the rows through registers (global_load -> ds_write), both products' outputs through a shared 33-KB staging
// buffer, with two workgroup barriers per tile; knock-out builds (tools/probes/ko_wreg.py, profiles/r05_experiments.txt) showed
// its phases ADDING UP (3 layers' products: 220 us = ~70 base + ~80 MFMA + ~53 stores + ~15 loads, power-throttled box) and the
// ISA showed why: vmcnt(0) at the loop head (every load and store of the previous iteration drained), an A fragment read,
// waited for, two MFMAs, and so on.  This body keeps the geometry (4 wavefronts x 64 output columns in 128 registers, 32-row
// tiles, 2 workgroups per CU), the k order and the accumulation -- outputs are bit-identical to the old body's -- and changes
// how the bytes move:
//   * A tiles arrive by LDS-DMA (global_load_lds_dwordx4, 4 x 1 KB per wavefront and tile, one M0 write per tile) into a ring of
//     3 unpadded 16-KB slots, two tiles ahead; the 16-byte chunks of a row are XOR-permuted by (row & 15) on the SOURCE side (the
//     DMA writes lane-linear), the MFMA operand reads apply the same permutation: conflict-free ds_read_b128 without a pad;
//   * a wavefront stages ITS OWN 64 columns x 32 rows (4.5 KB, private: no barrier between its epilogue and its stores) and
//     stores them during the next tile: head planes as 1-KB runs (16 rows x 64 B of one head), G rows as 8 x 128-B lines;
//   * one workgroup barrier per tile (hand-over of the A ring); the wait in front of it is an exact count -- in-order vmcnt:
//     behind the loads of tile t sit the 4 stores of tile t-2 and the 4 loads of tile t+1 -- so loads and stores stay in
//     flight across the barrier (hipcc does not see the asm loads: it neither counts nor drains them);
//   * the previous tile's stores, the DMA of the tile after next and the bias reads are issued between the MFMAs of the k loop
//     (pinned with sched_barrier): a wavefront's tile takes ~3.2 k cycles instead of ~4.4 k with the phases one after the other.
// What it buys (profiles/r05_experiments.txt): 8-10 % less kernel time alone (where 30 back-to-back launches throttle the
// clock to 1.1-1.9 GHz: fewer instructions, less LDS traffic), 4 % per forward at 4 samples per forward, nothing at one sample
// per forward -- there the launches run at 2.2 GHz and ~5.5 TB/s: HBM-write-bound either way.
constexpr int W2_NA = 3;                       // A ring slots
constexpr int W2_ABYTES = RM * 512;            // one slot: 32 rows x 256 bf16, unpadded
constexpr int W2_STP = 144;                    // staging pitch: 64 bf16 + 16 B (conflict-free b64 writes / b128 reads)
constexpr int W2_STW = RM * W2_STP;            // per wavefront
constexpr int W2_LDS = W2_NA * W2_ABYTES + 4 * W2_STW + 256 * (int)sizeof(float);

template <bool PLANES, bool SMALL_S>
__device__ __forceinline__ void wreg2_body(const WregParams& p, char* smem, const int first_tile, const int tile_stride) {
  const int tid = threadIdx.x, lane = tid & 63, rl = lane & 31, h = lane >> 5;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* stage = smem + W2_NA * W2_ABYTES + wn * W2_STW;
  float* bias_s = reinterpret_cast<float*>(smem + W2_NA * W2_ABYTES + 4 * W2_STW);
  const bool wave_has_cols = wn * 64 < p.N;
  bias_s[tid] = p.bias ? p.bias[tid] : 0.f;
  W2STAMP_ALWAYS(8, __builtin_amdgcn_s_memtime());
  W2STAMP_ALWAYS(13, __builtin_amdgcn_s_memrealtime());

  f32x4 wreg[16][2];
  {
    const bf16_t* wp = p.Wf + (long)wn * 16 * 1024 + lane * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      wreg[ks][0] = *reinterpret_cast<const f32x4*>(wp + ks * 1024);
      wreg[ks][1] = *reinterpret_cast<const f32x4*>(wp + ks * 1024 + 512);
    }
  }
  __syncthreads();                               // bias_s; the weight loads are drained here, before the counted waits start
  W2STAMP_ALWAYS(9, __builtin_amdgcn_s_memtime());
  const int ntiles = (p.M + RM - 1) / RM;
  const int G = tile_stride;
  const unsigned abuf_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // LDS-DMA of a tile into ring slot `slot`: piece q of wavefront wn fills tile positions (4 wn + q) * 64 .. + 63 = rows
  // 2 (4 wn + q), + 1; lane -> (row, slot chunk lane & 31) <- source chunk (lane & 31) ^ (row & 15).  Tiles past the end re-read
  // the last tile, rows past the end the last row (nobody uses them): every wavefront issues 4 loads per call, always.
  // ONE statement per tile: M0 (the LDS base) is written once and the four pieces go through the instruction's offset field,
  // which the hardware adds to the LDS address AND to the global address -- so lane pointer q is biased by -1024 q bytes.
  // (One M0 write per piece measured ~260 cycles per piece: the write waits for the previous piece to have read M0.)
  int srow[4], schunk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    srow[q] = 2 * (4 * wn + q) + (lane >> 5);
    schunk[q] = ((lane & 31) ^ (srow[q] & 15)) * 8 - 512 * q;          // bf16 elements
  }
  const bf16_t* dsrc[4];
  auto dma_addr = [&](int tile) {
    const long r0 = (long)min(tile, ntiles - 1) * RM;
#pragma unroll
    for (int q = 0; q < 4; ++q) dsrc[q] = p.A + min(r0 + srow[q], (long)p.M - 1) * 256 + schunk[q];
  };
  auto dma_issue = [&](int slot) {
    const unsigned dst = abuf_lds + slot * W2_ABYTES + wn * 4096;
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %3, off offset:3072"
                 :: "v"(dsrc[0]), "v"(dsrc[1]), "v"(dsrc[2]), "v"(dsrc[3]), "s"(dst) : "memory");
  };
  // The same four pieces one per k-step (round 6: s_memtime stamps inside the k loop showed the quarter that carried the four
  // stores and this block at 744 cycles for 256 of MFMA issue -- a vector-memory instruction takes 60-90 cycles to issue and the
  // MFMAs behind it in program order wait).  Piece 0 writes M0; pieces 1-3 rely on it: nothing the compiler emits between them
  // (MFMA, ds_read, VALU, global_store) touches M0 on gfx950.
  auto dma_piece = [&](int slot, auto qc) {
    constexpr int Q = decltype(qc)::value;
    if constexpr (Q == 0) {
      const unsigned dst = abuf_lds + slot * W2_ABYTES + wn * 4096;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(dsrc[0]), "s"(dst) : "memory");
    } else if constexpr (Q == 1) {
      asm volatile("global_load_lds_dwordx4 %0, off offset:1024" :: "v"(dsrc[1]) : "memory");
    } else if constexpr (Q == 2) {
      asm volatile("global_load_lds_dwordx4 %0, off offset:2048" :: "v"(dsrc[2]) : "memory");
    } else {
      asm volatile("global_load_lds_dwordx4 %0, off offset:3072" :: "v"(dsrc[3]) : "memory");
    }
  };
  // operand reads: chunk 2 ks + h of row rl sits at slot chunk (2 ks + h) ^ (rl & 15) = (((ks & 7) ^ (rl >> 1 & 7)) << 1 | (h ^ rl & 1))
  // + 16 (ks >> 3): 8 lane-dependent addresses, ks >= 8 through the instruction's offset field
  unsigned aoff[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) aoff[k] = rl * 512 + ((((k ^ ((rl >> 1) & 7)) << 1) | (h ^ (rl & 1))) << 4);

  // stores of a finished tile from the wavefront's staging rows (4 x 16 B per lane), in three steps that the k loop of the next
  // tile spreads over its MFMAs: staging rows -> registers, addresses, stores
  bf16_t* outp = reinterpret_cast<bf16_t*>(p.out);
  f32x4 sv[4];
  bf16_t* sdst[4];
  auto st_read = [&](int tile, int i) {
    const int last_row = p.M - 1 - tile * RM;           // rows past it do not exist (only in the final tile)
    if (PLANES) sv[i] = *reinterpret_cast<const f32x4*>(stage + min(16 * (i & 1) + (lane >> 2), last_row) * W2_STP + (i >> 1) * 64 + (lane & 3) * 16);
    else sv[i] = *reinterpret_cast<const f32x4*>(stage + min(8 * i + (lane >> 3), last_row) * W2_STP + (lane & 7) * 16);
  };
  // (one address per call: the ~25 vector instructions of an address are hidden by one k-step's MFMAs, the four together were not)
  auto st_addr1 = [&](int tile, int i) {
    const int r0 = tile * RM;
    const int last_row = p.M - 1 - r0;
    if (PLANES) {
      // image / pixel of the tile's first row (wave-uniform); a 32-row tile crosses at most one image boundary
      // when an image has >= 32 pixels (else: the division per lane)
      if (!SMALL_S) {
        const int img0 = r0 / p.S_img, sp0 = r0 - img0 * p.S_img;
        const int row = min(16 * (i & 1) + (lane >> 2), last_row);
        const bool over = sp0 + row >= p.S_img;
        const int sp = over ? sp0 + row - p.S_img : sp0 + row, img = over ? img0 + 1 : img0;
        sdst[i] = outp + (((long)img * 8 + 2 * wn + (i >> 1)) * p.S_img + sp) * 32 + (lane & 3) * 8;
      } else {
        const int grow = r0 + min(16 * (i & 1) + (lane >> 2), last_row);
        const int img = grow / p.S_img, sp = grow - img * p.S_img;
        sdst[i] = outp + (((long)img * 8 + 2 * wn + (i >> 1)) * p.S_img + sp) * 32 + (lane & 3) * 8;
      }
    } else {
      sdst[i] = outp + (long)(r0 + min(8 * i + (lane >> 3), last_row)) * p.N + wn * 64 + (lane & 7) * 8;
    }
  };
  auto st_addr = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) st_addr1(tile, i);
  };
  auto st_store = [&](int i) { *reinterpret_cast<f32x4*>(sdst[i]) = sv[i]; };

  dma_addr(first_tile); dma_issue(0);
  dma_addr(first_tile + G); dma_issue(1);
  int it = 0, slot = 0, prev_tile = -1;

  // One tile.  WITH_PREV: the previous tile's stores ride on this tile's k loop.  Order of the memory instructions of an
  // iteration: stores(t-1) x 4, then loads(t+2) x 4 -- so behind loads(t) sit exactly stores(t-2) and loads(t+1).
  auto tile_body = [&](auto with_prev, const int tile) {
    constexpr bool WITH_PREV = decltype(with_prev)::value;
    W2STAMP(0);
    if (it >= 2 && wave_has_cols) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    W2STAMP(1);
    __builtin_amdgcn_s_barrier();          // every wavefront's share of this tile has landed; slot (t-1) % 3 is no longer read
    asm volatile("" ::: "memory");
    W2STAMP(2);
    int nslot = slot + 2; if (nslot >= W2_NA) nslot -= W2_NA;
    if (!wave_has_cols) {                  // (the fourth wavefront of a 192-column job only moves its share of the tiles)
      dma_addr(tile + 2 * G);
      dma_issue(nslot);
    } else {
      const char* ab = smem + slot * W2_ABYTES;
      f32x16 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
      f32x4 a[4];
      f32x4 bv[8];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const f32x4*>(ab + aoff[ks & 7] + (ks >> 3) * 256);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[ks][j]),
                                                           __builtin_bit_cast(bf16x8, a[ks & 3]), acc[j], 0, 0, 0);
        if (ks + 4 < 16) a[ks & 3] = *reinterpret_cast<const f32x4*>(ab + aoff[(ks + 4) & 7] + ((ks + 4) >> 3) * 256);
        // fillers, issued in the shadow of this step's MFMAs
        if (WITH_PREV) {
          if (ks < 4) { st_read(prev_tile, ks); st_addr1(prev_tile, ks); }
          if (ks >= 4 && ks < 8) st_store(ks - 4);
        }
        if (ks == 3) W2STAMP(3);           // (probe build only: the k loop in quarters)
        if (ks == 7) W2STAMP(4);
        if (ks == 11) W2STAMP(7);
        if (ks == 7) dma_addr(tile + 2 * G);
        if (ks == 8) dma_piece(nslot, std::integral_constant<int, 0>{});
        if (ks == 9) dma_piece(nslot, std::integral_constant<int, 1>{});
        if (ks == 10) dma_piece(nslot, std::integral_constant<int, 2>{});
        if (ks == 11) dma_piece(nslot, std::integral_constant<int, 3>{});
        if (ks >= 12) {
          bv[2 * (ks - 12)] = *reinterpret_cast<const f32x4*>(bias_s + wn * 64 + ((ks - 12) >> 1) * 32 + 8 * (2 * ((ks - 12) & 1)) + 4 * h);
          bv[2 * (ks - 12) + 1] = *reinterpret_cast<const f32x4*>(bias_s + wn * 64 + ((ks - 12) >> 1) * 32 + 8 * (2 * ((ks - 12) & 1) + 1) + 4 * h);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      W2STAMP(5);
      // epilogue: bias (requested during the k loop), bf16, into the wavefront's staging rows
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nn = j * 32 + 8 * g + 4 * h;
          const f32x4 b4 = bv[4 * j + g];
          uint2 pk;
          pk.x = pack_bf16(acc[j][4 * g] + b4[0], acc[j][4 * g + 1] + b4[1]);
          pk.y = pack_bf16(acc[j][4 * g + 2] + b4[2], acc[j][4 * g + 3] + b4[3]);
          *reinterpret_cast<uint2*>(stage + rl * W2_STP + nn * 2) = pk;
        }
      prev_tile = tile;
      W2STAMP(6);
    }
    slot = slot + 1 == W2_NA ? 0 : slot + 1;
    ++it;
  };
  int tile = first_tile;
  if (tile < ntiles) {
    tile_body(std::false_type{}, tile);
    tile += G;
  }
#pragma unroll 1
  for (; tile < ntiles; tile += G) tile_body(std::true_type{}, tile);
  W2STAMP_ALWAYS(10, __builtin_amdgcn_s_memtime());
  W2STAMP_ALWAYS(12, (unsigned long long)it);
  if (wave_has_cols && prev_tile >= 0) {
    st_addr(prev_tile);
#pragma unroll
    for (int i = 0; i < 4; ++i) { st_read(prev_tile, i); st_store(i); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the ring's look-ahead loads must not outlive the workgroup's LDS
  W2STAMP_ALWAYS(11, __builtin_amdgcn_s_memtime());
  W2STAMP_ALWAYS(14, __builtin_amdgcn_s_memrealtime());
}

// ---- several products of the SAME rows in one launch (round 5) -------------------------------------------------------------
// All layers' value planes and G are products of the one packed pyramid with different weights.  Launched one after the other,
// every product streams the 103-MB pyramid through the fabric again (8 x per forward at cfg-2).  Here the workgroups of an XCD
// (block b runs on XCD b & 7 -- observed dispatch order, used for speed only) are divided among the jobs in proportion to their
// column counts; every job sweeps the XCD's row tiles (tiles t = 8 u + xcd) in the same order and at the same pace, so a tile
// fetched by the first job's workgroup is served to the others by the XCD's L2: the pyramid crosses the fabric once per launch.
// A workgroup keeps ONE job's weight slice in registers for its whole life (same loop and k order whatever the
// job mix: a product's bytes do not depend on what it is launched with).
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


template <bool SMALL_S>
__global__ __launch_bounds__(256, 2) void wreg2_group_kernel(WregGroupParams gp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int j = gp.slot_job[slot];
  WregParams p;
  p.A = gp.A; p.M = gp.M; p.S_img = gp.S_img;
  p.Wf = gp.job[j].Wf; p.bias = gp.job[j].bias; p.out = gp.job[j].out; p.N = gp.job[j].N; p.rowmajor = gp.job[j].rowmajor;
  const int first = gp.slot_idx[slot] * 8 + xcd, stride = gp.job_slots[j] * 8;
  if (!p.rowmajor) wreg2_body<true, SMALL_S>(p, smem, first, stride);
  else wreg2_body<false, false>(p, smem, first, stride);
}

// > 64 KB of dynamic LDS: the attribute is per DEVICE (see launch_chain_a)
int wreg2_configure() {
  static bool configured[MVG_MAX_DEVICES] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVG_MAX_DEVICES) return MVG_E_BADARG;
  if (!configured[dev]) {
    const void* fns[] = {reinterpret_cast<const void*>(&wreg2_group_kernel<false>),
                         reinterpret_cast<const void*>(&wreg2_group_kernel<true>)};
    hipError_t e = hipSuccess;
    for (const void* f : fns)
      if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS);
    if (e != hipSuccess) return (int)e;
    configured[dev] = true;
  }
  return 0;
}


}  // namespace



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
    // Equal shares: every workgroup gets the same number of tiles, whatever its job's column count.  A workgroup's time per tile is the
    // same for a 256- and a 192-column job (s_memtime stamps in the forward: 2 430 cycles), and a launch with one workgroup per CU
    // (the just-in-time launches) is bound by that tile loop, not by HBM: with the slots split by columns (18 : 13 of 32) the G
    // workgroups ran 56 tiles against 44 and finished 9 us late (forward -2.2 % with 16 : 16); the two-workgroups-per-CU launches are
    // HBM-bound and indifferent (2 / 4 samples per forward: -0.6 / -0.7 %).
    weight[j] = 1;
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
  if (int e = wreg2_configure()) return e;
  if (gp.S_img >= RM) hipLaunchKernelGGL(wreg2_group_kernel<false>, dim3(8 * n_slots), dim3(256), W2_LDS, (hipStream_t)stream, gp);
  else hipLaunchKernelGGL(wreg2_group_kernel<true>, dim3(8 * n_slots), dim3(256), W2_LDS, (hipStream_t)stream, gp);
  MVG_LAUNCH_CHECK();
  return 0;
}
