// Dense projections of the decoder layer on the gfx950 matrix cores.
//
//   out[M,N] = act((A[M,K] (+ A2[M,K])) @ W[N,K]^T + bias[N]) * rowmask[M]
//
// A is activations (row-major, K contiguous), W an nn.Linear weight (row-major (N,K), K
// contiguous) -- both operands are "K-major", so both are staged the same way.
//   fp32 compute : v_mfma_f32_32x32x2_f32   (exact fp32 fmaf chain; 157 TF peak)
//   fp32 "split" : every fp32 operand value x = h + m + l with h, m, l bf16 (round-to-nearest at each step: 8 + 8 + 8 significand
//                  bits, exact), the product as the six bf16 MFMAs  h*h + h*m + m*h + m*m + h*l + l*h  accumulated in fp32
//                  (the three dropped terms are below 2^-25 |a b|, under the rounding of one fp32 product): 6 x 32 cycles of
//                  v_mfma_f32_32x32x16_bf16 per 16 k instead of 8 x 64 cycles of 32x32x2_f32 -- gfx950 has no tf32 / xf32.
//                  Knob "f32_split" (mvg_set_tuning); the split happens when a tile is staged (5.5 VALU per value).
//   bf16 compute : v_mfma_f32_32x32x16_bf16 (fp32 accumulate; 2.5 PF peak)
// Tile 128x128 per 256-thread workgroup (4 wavefronts as 2x2, 64x64 per wavefront = 2x2 MFMA
// tiles, 64 accumulator registers).  K is consumed in slabs of 128 BYTES per row (32 fp32 or
// 64 bf16): with a 144-byte LDS row pitch every ds_read_b128 of a 16-lane group falls on 16
// distinct 16-byte slots (36 dwords * i mod 64 is a distinct multiple of 4 for i mod 16), i.e.
// bank-conflict-free, and the same byte offsets (32*g + 16*(lane>>5)) serve both data types:
// an fp32 lane gets k = 8g+4h..+3 (fed to 4 successive 32x32x2 MFMAs -- the k order inside a
// slab is permuted identically for A and W, which leaves the sum unchanged), a bf16 lane gets
// k = 16g+8h..+7 (one 32x32x16 MFMA).
// Global->register prefetch of slab t+1 overlaps the MFMAs of slab t.
#include "common.h"

int g_linear_tiles = 1;     // tuning knob "linear_tiles": 0 = always 128 x 128 tiles (rounds 1-2)
static const int g_linear_xcd = 1;   // the column tiles of one row tile run on the same XCD (shared L2); 0 measured slower (round 3)
int g_f32_split = 1;        // tuning knob "f32_split": 1 = fp32 GEMMs as six bf16 MFMAs on 3-way split operands (see above)

namespace {

constexpr int PITCH = 144;   // bytes per LDS row (128 data + 16 pad)
// split form: a row of the slab is [32 x h | 32 x m | 32 x l] bf16 = 3 x 64 bytes, no padding; the four 16-byte slots of a
// plane are permuted by bits 2-3 of the row (slot ^ ((row >> 2) & 3)).  Reads: the 16 rows of a ds_read_b128 lane group start
// at slot columns 12 r mod 16 = {0, 12, 8, 4} repeating every 4 rows, and the permutation moves each group of 4 rows to a
// different slot: 16 distinct slots.  Writes: a half wavefront's ds_write_b64 covers 4 rows x 64 contiguous bytes at 48 r
// dwords = {0, 48, 32, 16}: disjoint banks.  (A padded 208-byte pitch was read conflict-free but cost 31 % bank-conflict
// cycles on the 8-byte writes.)
constexpr int SPITCH = 192;
__device__ __forceinline__ int split_slot(int row, int slot) { return (slot ^ ((row >> 2) & 3)) * 16; }
// Tile shapes (round 3): 128 x 128 (default), 128 x 192 for N = 192 (the [offsets | logits] projections: two 128-column tiles
// left a quarter of the MFMAs on padding), 64 x 128 when 128-row tiles would not give every CU a workgroup (the FFN's second
// GEMM at 7 680 rows: 120 workgroups on 256 CUs).  Always 4 wavefronts as 2 x 2; a wavefront owns (BM/2) x (BN/2).

struct Chunk {
  f32x4 lo, hi;   // hi only used when converting an fp32 source to bf16 (8 values)
};

// load one 16-byte LDS chunk worth of K-values for (row, col16) of the current slab.  Rows past the end of the tile are read
// from the tile's last row (an unconditional load -- no exec-mask region per chunk): row i of A only reaches row i of the
// product and row j of W only column j, and the epilogue stores neither.
template <typename TSRC, bool BF16>
__device__ __forceinline__ Chunk load_chunk(const TSRC* __restrict__ base, long ld, int row, int nrows, int k0,
                                            int col16) {
  Chunk c;
  c.hi = f32x4{0.f, 0.f, 0.f, 0.f};
  const TSRC* p = base + (long)min(row, nrows - 1) * ld + k0;
  if constexpr (!BF16) {
    c.lo = *reinterpret_cast<const f32x4*>(p + col16 * 4);
  } else if constexpr (sizeof(TSRC) == 2) {
    c.lo = *reinterpret_cast<const f32x4*>(p + col16 * 8);   // 8 bf16 = 16 B, raw
  } else {
    c.lo = *reinterpret_cast<const f32x4*>(p + col16 * 8);
    c.hi = *reinterpret_cast<const f32x4*>(p + col16 * 8 + 4);
  }
  return c;
}

template <typename TSRC, bool BF16>
__device__ __forceinline__ void store_chunk(char* lds, int row, int col16, const Chunk& c) {
  f32x4 v = c.lo;
  if constexpr (BF16 && sizeof(TSRC) == 4) {
    unsigned w[4];
    w[0] = pack_bf16(c.lo[0], c.lo[1]);
    w[1] = pack_bf16(c.lo[2], c.lo[3]);
    w[2] = pack_bf16(c.hi[0], c.hi[1]);
    w[3] = pack_bf16(c.hi[2], c.hi[3]);
    v[0] = __uint_as_float(w[0]); v[1] = __uint_as_float(w[1]); v[2] = __uint_as_float(w[2]); v[3] = __uint_as_float(w[3]);
  }
  *reinterpret_cast<f32x4*>(lds + row * PITCH + col16 * 16) = v;
}

// split form: 4 fp32 values -> their h / m / l bf16 parts, 8 bytes each into the row's three planes
__device__ __forceinline__ void store_chunk_split(char* lds, int row, int col16, const f32x4& x) {
  uint2 p[3];
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
    *reinterpret_cast<uint2*>(lds + row * SPITCH + 64 * s + split_slot(row, col16 >> 1) + (col16 & 1) * 8) = p[s];
  }
}

// TA: storage type of A in global memory; BF16: compute type; TW = BF16 ? bf16 : float; TO: output storage
// IDX (round 3, fp32 path): tile row i works on global row order[m0 + i] of A / out (the sampler's processing order: rows whose
// reference point is outside the image come last, mvg_bin_pairs).  A tile without a single row of `inside` does no arithmetic: all
// of its output rows equal `masked_row` (N floats: what this very kernel computes for such a row -- zeros behind a row mask,
// act(bias) for a zero input row, a cached constant further down a chain), which it broadcasts and leaves.
// SPLIT (fp32 operands only): the split form described at the top of the file.
template <typename TA, bool BF16, typename TO, int BM = 128, int BN = 128, bool IDX = false, bool SPLIT = false>
__global__ __launch_bounds__(256, (BN > 128 ? 2 : (SPLIT ? 3 : 1))) void linear_kernel(const TA* __restrict__ A, const TA* __restrict__ A2, long lda,
                                                     const void* __restrict__ Wv,
                                                     const float* __restrict__ bias, TO* __restrict__ out, long ldc,
                                                     const uint8_t* __restrict__ rowmask, int relu, int M, int N,
                                                     int K, const int* __restrict__ order = nullptr,
                                                     const uint8_t* __restrict__ inside = nullptr,
                                                     const float* __restrict__ masked_row = nullptr, int xcd_map = 0,
                                                     int ldw = 0, int k_split = 0) {
  using TW = typename std::conditional<BF16, bf16_t, float>::type;
  const TW* __restrict__ W = reinterpret_cast<const TW*>(Wv);
  constexpr int KSLAB = BF16 ? 64 : 32;   // K elements per 128-byte slab
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NJ = WN / 32;      // per-wavefront tile and its 32 x 32 MFMA blocks
  constexpr int EPB = (int)sizeof(TO) * BN + 16;                           // epilogue staging pitch (bytes)
  static_assert(!SPLIT || (!BF16 && sizeof(TA) == 4), "the split form takes fp32 operands");
  constexpr int RP = SPLIT ? SPITCH : PITCH;                               // LDS row pitch of the staged slab
  constexpr int LDS_BYTES = (BM + BN) * RP > 64 * EPB ? (BM + BN) * RP : 64 * EPB;
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  char* ldsA = lds;
  char* ldsB = lds + BM * RP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroups go to the 8 XCDs round-robin by linear id, and every XCD has its own L2: with the plain (x = column tile) order
  // the column tiles of one row tile land on different XCDs and each fetches the same rows of A from HBM.  xcd_map: ids
  // i, i + 8, i + 16, ... (same XCD, dispatched back to back) are the column tiles of one row tile.
  int bx = blockIdx.x, by = blockIdx.y;
  if (xcd_map) {
    const int NT = gridDim.x, MT = gridDim.y, id = by * NT + bx;
    const int group = id / (8 * NT), within = id - group * 8 * NT;
    const int R = min(8, MT - group * 8);                   // row tiles of this group (the last group may be partial)
    bx = within / R;
    by = group * 8 + (within - bx * R);
  }
  const int m0 = by * BM, n0 = bx * BN;
  // split K (mvg_linear_splitk: the weight gradient dW = dY^T X, whose reduction runs over hundreds of thousands of rows and whose
  // output is one or four tiles): slice z = blockIdx.z works on k in [z k_split, (z + 1) k_split) and writes its own (M, N) partial
  const long wld = ldw > 0 ? ldw : K;
  const long koff = k_split > 0 ? (long)blockIdx.z * k_split : 0;
  if (k_split > 0) {
    out += (long)blockIdx.z * M * ldc;
    K = k_split;
  }
  const TA* Ab = A + (long)m0 * lda + koff;
  const TA* A2b = A2 ? A2 + (long)m0 * lda + koff : nullptr;      // optional addend (fp32 storage only): A + A2 formed on load
  const TW* Wb = W + (long)n0 * wld + koff;
  const int mrows = min(BM, M - m0), nrows = min(BN, N - n0);
  __shared__ int rid[IDX ? BM : 1];                         // IDX: global row of every tile row (-1: past the end)
  if constexpr (IDX) {
    bool mine = false;
    if (tid < BM) {
      const int slot = m0 + tid;
      const int g = slot < M ? order[slot] : -1;
      rid[tid] = g;
      mine = g >= 0 && inside[g] != 0;
    }
    if (__syncthreads_or(mine) == 0) {                      // nothing of this tile is inside an image: constant rows
      constexpr int VPT = BN / 4;                           // 16-byte vectors per output row of the tile
      for (int c = tid; c < BM * VPT; c += 256) {
        const int row = c / VPT, col = n0 + (c % VPT) * 4;
        const int g = rid[row];
        if (g >= 0 && col < N) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(masked_row + col);
          if constexpr (sizeof(TO) == 4) {
            *reinterpret_cast<f32x4*>(out + (long)g * ldc + col) = v;
          } else {
            uint2 pk;
            pk.x = pack_bf16(v[0], v[1]);
            pk.y = pack_bf16(v[2], v[3]);
            *reinterpret_cast<uint2*>(out + (long)g * ldc + col) = pk;
          }
        }
      }
      return;
    }
  }

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  constexpr int NCA = BM / 32, NCB = BN / 32;          // 16-byte chunks per thread and slab (a tile row = 8 chunks)
  // one slab in flight in registers (two, for the split form, measured slower: 222 registers + 64 accumulators leave one
  // wavefront per SIMD, or spill under a cap; the loop does not wait on memory -- profiles/r03_experiments.txt)
  Chunk ca[NCA], cb[NCB];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const int c = tid + 256 * i;
      if constexpr (IDX) {            // row pointer through the processing order; slots past the end read row order[..] = 0
        const int g = rid[c >> 3];
        ca[i] = load_chunk<TA, BF16>(A + (long)max(g, 0) * lda, lda, 0, 1, k0, c & 7);
      } else {
        ca[i] = load_chunk<TA, BF16>(Ab, lda, c >> 3, mrows, k0, c & 7);
      }
      if constexpr (sizeof(TA) == 4) {
        if (A2b) {
          const Chunk c2 = load_chunk<TA, BF16>(A2b, lda, c >> 3, mrows, k0, c & 7);
          ca[i].lo += c2.lo;
          ca[i].hi += c2.hi;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = tid + 256 * i;
      cb[i] = load_chunk<TW, BF16>(Wb, wld, c >> 3, nrows, k0, c & 7);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const int c = tid + 256 * i;
      if constexpr (SPLIT) store_chunk_split(ldsA, c >> 3, c & 7, ca[i].lo);
      else store_chunk<TA, BF16>(ldsA, c >> 3, c & 7, ca[i]);
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int c = tid + 256 * i;
      if constexpr (SPLIT) store_chunk_split(ldsB, c >> 3, c & 7, cb[i].lo);
      else store_chunk<TW, BF16>(ldsB, c >> 3, c & 7, cb[i]);
    }
  };

  const int nk = K / KSLAB;
  const int rl = lane & 31, h = lane >> 5;
  // the MFMAs of the slab staged in LDS
  auto compute = [&]() {
    if constexpr (SPLIT) {
      // 32 k per slab = 2 MFMA k-steps; lane (rl, h) holds k = 16 g + 8 h .. + 7 of its row in each of the three planes
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bf16x8 a[MI][3], b[NJ][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
          for (int i = 0; i < MI; ++i)
            a[i][s] = *reinterpret_cast<const bf16x8*>(ldsA + (wm * WM + i * 32 + rl) * SPITCH + 64 * s + split_slot(rl, 2 * g + h));
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            b[j][s] = *reinterpret_cast<const bf16x8*>(ldsB + (wn * WN + j * 32 + rl) * SPITCH + 64 * s + split_slot(rl, 2 * g + h));
        }
        // smallest terms first; (plane of W, plane of A)
        constexpr int TB[6] = {2, 0, 1, 1, 0, 0}, TA_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][TB[t]], a[i][TA_[t]], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a[MI], b[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const f32x4*>(ldsA + (wm * WM + i * 32 + rl) * PITCH + 32 * g + 16 * h);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const f32x4*>(ldsB + (wn * WN + j * 32 + rl) * PITCH + 32 * g + 16 * h);
        // D' = W_tile * A_tile^T: the MFMA "row" index (registers) runs over output COLUMNS n, the
        // lane index over output ROWS m, so a lane ends up with groups of 4 consecutive n.
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if constexpr (BF16) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]),
                                                                  __builtin_bit_cast(bf16x8, a[i]), acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
              for (int t = 0; t < 4; ++t)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][t], a[i][t], acc[i][j], 0, 0, 0);
            }
          }
      }
    }
  };

  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * KSLAB);
    compute();
    __syncthreads();
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }

  // ---- epilogue.  acc[i][j][e] = out[m0 + wm*WM + i*32 + rl][n0 + wn*WN + j*32 + (e&3) + 8*(e>>2) + 4*h].
  // MI phases of 64 rows x BN columns: bias / ReLU / row mask in registers, the tile is
  // transposed through LDS (16-byte writes of 4 consecutive columns), then stored with 16-byte
  // vectors, 2 * BN (bf16) or 4 * BN (fp32) contiguous bytes per row.
  constexpr int EP = EPB;                           // staging pitch in bytes
  constexpr int CPV = 16 / (int)sizeof(TO);         // columns per 16-byte vector
  constexpr int VPR = BN / CPV;                     // vectors per staged row
  static_assert((64 * VPR) % 256 == 0, "the staged tile is stored in whole passes of the workgroup");
  // the tile's bias values go through LDS (columns past N read as 0; they are not stored).  As a conditional global load per
  // element inside the loop below each of a lane's 32 was its own exec region with its own wait: 8 900 cycles per phase, a
  // quarter of a workgroup's lifetime (s_memtime, profiles/r03_experiments.txt).
  __shared__ __attribute__((aligned(16))) float sbias[BN];
  if (tid < BN) sbias[tid] = (bias && n0 + tid < N) ? bias[n0 + tid] : 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int grow = m0 + wm * WM + i * 32 + rl;
    if constexpr (IDX) grow = rid[wm * WM + i * 32 + rl];
    const bool keep = rowmask ? (grow >= 0 && grow < M && rowmask[grow] != 0) : true;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = wn * WN + j * 32 + 8 * g + 4 * h;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(sbias + nl);
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float x = acc[i][j][4 * g + t] + bq[t];
          if (relu) x = fmaxf(x, 0.f);
          v[t] = keep ? x : 0.f;
        }
        char* dst = lds + (wm * 32 + rl) * EP + nl * (int)sizeof(TO);
        if constexpr (sizeof(TO) == 4) {
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
          uint2 pk;
          pk.x = pack_bf16(v[0], v[1]);
          pk.y = pack_bf16(v[2], v[3]);
          *reinterpret_cast<uint2*>(dst) = pk;
        }
      }
    __syncthreads();
#pragma unroll
    for (int c0 = 0; c0 < 64 * VPR; c0 += 256) {
      const int c = c0 + tid;
      const int srow = c / VPR, vcol = c % VPR;            // staged row (0..63), vector column
      int row = m0 + (srow >> 5) * WM + i * 32 + (srow & 31);
      if constexpr (IDX) row = rid[(srow >> 5) * WM + i * 32 + (srow & 31)];
      const int col = n0 + vcol * CPV;
      if (row >= 0 && row < M && col < N) {
        *reinterpret_cast<f32x4*>(out + (long)row * ldc + col) =
            *reinterpret_cast<const f32x4*>(lds + srow * EP + vcol * 16);
      }
    }
    __syncthreads();
  }
}

// Weight gradient of a Linear under autograd:  dW[n][k] = sum_r dY[r][n] X[r][k]  (both operands row-major over the rows r that
// are the reduction: a "TN" product).  mvg_linear wants both operands K-major, i.e. dY^T and X^T -- as torch copies those two
// transposes were 37 % of a training step (profiles/r04_experiments.txt).  Here the transposition happens on the way into LDS:
// a thread loads an 8-row x 4-column patch of its operand's 32-row slab (8 x 16 B), and writes, per column and split part, the 8
// row values as ONE 16-byte LDS store into the split form's plane layout (row = output index, 32 k per slab) -- the compute
// phase is linear_kernel's.  blockIdx.z cuts the rows into slices; every slice writes its own (N, K) partial (summed by the caller
// in slice order: deterministic).
__device__ __forceinline__ void split8_store(char* lds, int row, int slot, const float (&v)[8]) {
  float r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = v[i];
#pragma unroll
  for (int sp = 0; sp < 3; ++sp) {
    uint4 p;
    p.x = pack_bf16(r[0], r[1]);
    p.y = pack_bf16(r[2], r[3]);
    p.z = pack_bf16(r[4], r[5]);
    p.w = pack_bf16(r[6], r[7]);
    if (sp < 2) {
      r[0] -= __uint_as_float(p.x << 16); r[1] -= __uint_as_float(p.x & 0xffff0000u);
      r[2] -= __uint_as_float(p.y << 16); r[3] -= __uint_as_float(p.y & 0xffff0000u);
      r[4] -= __uint_as_float(p.z << 16); r[5] -= __uint_as_float(p.z & 0xffff0000u);
      r[6] -= __uint_as_float(p.w << 16); r[7] -= __uint_as_float(p.w & 0xffff0000u);
    }
    *reinterpret_cast<uint4*>(lds + row * SPITCH + 64 * sp + split_slot(row, slot)) = p;
  }
}

// FUSED: the workgroups of the first column of tiles also publish the column sums of their dY slices (the bias gradient's partials);
// wgrad_reduce_kernel then adds the slices' partials of both in slice order.  (Adding them inside this launch -- the workgroup that
// draws a tile's last ticket sums the tile -- was built and measured: the last workgroup of each tile reads `splits` x 64 KB alone,
// 30 ms of tails per training step.)
template <bool FUSED>
__global__ __launch_bounds__(256, 2) void wgrad_f32s_kernel(const float* __restrict__ dY, long ldy, const float* __restrict__ X, long ldx,
                                                            float* __restrict__ partial, int rows, int N, int K, int rows_per_split,
                                                            float* __restrict__ partial_db) {
  constexpr int BM = 128, BN = 128;
  __shared__ __attribute__((aligned(16))) char lds[(BM + BN) * SPITCH];
  char* ldsA = lds;                 // dY^T tile: row = output feature n (of dY), 32 reduction rows per slab
  char* ldsB = lds + BM * SPITCH;   // X^T tile: row = input feature k
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, rl = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const long r_begin = (long)blockIdx.z * rows_per_split, r_end = min((long)rows, r_begin + rows_per_split);
  // loader role: threads 0..127 stage dY, 128..255 stage X; patch = 8 rows (rgrp) x 4 columns (cgrp) of the 32 x 128 slab
  const int opnd = tid >> 7, p = tid & 127, rgrp = p >> 5, cgrp = p & 31;
  const float* src = opnd ? X : dY;
  const long ld = opnd ? ldx : ldy;
  const int width = opnd ? K : N, c0 = (opnd ? n0 : m0) + 4 * cgrp;
  char* dst = opnd ? ldsB : ldsA;
  f32x4 x[8];
  auto gload = [&](long r0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long r = r0 + 8 * rgrp + i;
      // clamped address + select: rows past the slice and columns past the operand contribute zeros
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + min(r, (long)rows - 1) * ld + min(c0, width - 4));
      x[i] = (r < r_end && c0 < width) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};      // FUSED: this thread's share of the column sums of dY (8 rows x 4 columns per slab)
  auto lstore = [&]() {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v[8] = {x[0][c], x[1][c], x[2][c], x[3][c], x[4][c], x[5][c], x[6][c], x[7][c]};
      split8_store(dst, 4 * cgrp + c, rgrp, v);
      if constexpr (FUSED) bsum[c] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nslab = (int)((r_end - r_begin + 31) / 32);
  if (nslab > 0) {
    gload(r_begin);
    lstore();
  }
  __syncthreads();
  for (int t = 0; t < nslab; ++t) {
    if (t + 1 < nslab) gload(r_begin + (long)(t + 1) * 32);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i][sp] = *reinterpret_cast<const bf16x8*>(ldsA + (wm * 64 + i * 32 + rl) * SPITCH + 64 * sp + split_slot(rl, 2 * g + h));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[j][sp] = *reinterpret_cast<const bf16x8*>(ldsB + (wn * 64 + j * 32 + rl) * SPITCH + 64 * sp + split_slot(rl, 2 * g + h));
      }
      constexpr int TB[6] = {2, 0, 1, 1, 0, 0}, TA_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int tt = 0; tt < 6; ++tt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][TB[tt]], a[i][TA_[tt]], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (t + 1 < nslab) {
      lstore();
      __syncthreads();
    }
  }
  // acc[i][j][4 g + t] = dW[m0 + wm*64 + i*32 + rl][n0 + wn*64 + j*32 + 8 g + 4 h + t]
  float* outz = partial + (long)blockIdx.z * N * K;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + rl;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * h;
        if (m < N && n < K)
          *reinterpret_cast<f32x4*>(outz + (long)m * K + n) = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
      }
  }
  if constexpr (FUSED) {
    if (partial_db != nullptr && blockIdx.x == 0) {
      float* red = reinterpret_cast<float*>(lds);       // the operand tiles are dead (barrier at the end of the last slab)
      if (opnd == 0)
#pragma unroll
        for (int c = 0; c < 4; ++c) red[rgrp * BM + 4 * cgrp + c] = bsum[c];
      __syncthreads();
      if (tid < BM && m0 + tid < N)
        partial_db[(long)blockIdx.z * N + m0 + tid] = (red[tid] + red[BM + tid]) + (red[2 * BM + tid] + red[3 * BM + tid]);
    }
  }
}

// dW = sum over the slices of partial (Z, N*K) and db = sum of partial_db (Z, N), slice 0 first: one float4 (or one bias entry) per thread
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ partial_db,
                                                           float* __restrict__ dW, float* __restrict__ db, int Z, long nk4, int N) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < nk4) {
    const f32x4* src = reinterpret_cast<const f32x4*>(partial) + i;
    f32x4 sum = src[0];
    int z = 1;
    for (; z + 4 <= Z; z += 4) {
      const f32x4 a = src[(long)z * nk4], b = src[(long)(z + 1) * nk4], c = src[(long)(z + 2) * nk4], d = src[(long)(z + 3) * nk4];
      sum = (((sum + a) + b) + c) + d;
    }
    for (; z < Z; ++z) sum += src[(long)z * nk4];
    reinterpret_cast<f32x4*>(dW)[i] = sum;
  } else if (db != nullptr && i - nk4 < N) {
    const long n = i - nk4;
    float sum = partial_db[n];
    for (int z = 1; z < Z; ++z) sum += partial_db[(long)z * N + n];
    db[n] = sum;
  }
}

template <typename TA, bool BF16, typename TO, int BM, int BN>
int launch_linear_tile(const void* A, const void* A2, long lda, const void* W, const float* bias, void* out, long ldc,
                       const uint8_t* rowmask, int relu, int M, int N, int K, hipStream_t st) {
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
  if constexpr (!BF16 && sizeof(TA) == 4) {
    if (g_f32_split) {
      hipLaunchKernelGGL((linear_kernel<TA, false, TO, BM, BN, false, true>), grid, dim3(256), 0, st, (const TA*)A, (const TA*)A2,
                         lda, W, bias, (TO*)out, ldc, rowmask, relu, M, N, K, nullptr, nullptr, nullptr, g_linear_xcd);
      MVG_LAUNCH_CHECK();
      return 0;
    }
  }
  hipLaunchKernelGGL((linear_kernel<TA, BF16, TO, BM, BN>), grid, dim3(256), 0, st, (const TA*)A, (const TA*)A2, lda, W, bias,
                     (TO*)out, ldc, rowmask, relu, M, N, K, nullptr, nullptr, nullptr, g_linear_xcd);
  MVG_LAUNCH_CHECK();
  return 0;
}

// fp32 indexed form (mvg_linear_ordered)
template <int BM>
int launch_linear_idx(const float* A, long lda, const float* W, const float* bias, float* out, long ldc, const uint8_t* rowmask,
                      int relu, int M, int N, int K, const int* order, const uint8_t* inside, const float* masked_row,
                      hipStream_t st) {
  dim3 grid((N + 127) / 128, (M + BM - 1) / BM);
  if (g_f32_split) {
    hipLaunchKernelGGL((linear_kernel<float, false, float, BM, 128, true, true>), grid, dim3(256), 0, st, A, (const float*)nullptr,
                       lda, (const void*)W, bias, out, ldc, rowmask, relu, M, N, K, order, inside, masked_row, g_linear_xcd);
    MVG_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL((linear_kernel<float, false, float, BM, 128, true>), grid, dim3(256), 0, st, A, (const float*)nullptr, lda,
                     (const void*)W, bias, out, ldc, rowmask, relu, M, N, K, order, inside, masked_row, g_linear_xcd);
  MVG_LAUNCH_CHECK();
  return 0;
}

template <typename TA, bool BF16, typename TO>
int launch_linear(const void* A, const void* A2, long lda, const void* W, const float* bias, void* out, long ldc,
                  const uint8_t* rowmask, int relu, int M, int N, int K, hipStream_t st) {
  if (g_linear_tiles) {
    if (N % 192 == 0 && N % 128 != 0)        // N = 192, 576, ...: no padded column tile
      return launch_linear_tile<TA, BF16, TO, 128, 192>(A, A2, lda, W, bias, out, ldc, rowmask, relu, M, N, K, st);
    const long tiles128 = (long)((N + 127) / 128) * ((M + 127) / 128);
    if ((tiles128 < 384 && M > 64) || g_linear_tiles == 2)   // fewer than 1.5 workgroups per CU of an MI355X: halve the row tile
      return launch_linear_tile<TA, BF16, TO, 64, 128>(A, A2, lda, W, bias, out, ldc, rowmask, relu, M, N, K, st);
  }
  return launch_linear_tile<TA, BF16, TO, 128, 128>(A, A2, lda, W, bias, out, ldc, rowmask, relu, M, N, K, st);
}

}  // namespace

extern "C" int mvg_linear_sum(const void* A, const void* A2, int a_dtype, int lda, const void* W, int w_dtype, const float* bias,
                              void* out, int out_dtype, int ldc, const uint8_t* rowmask, int relu, int M, int N, int K,
                              void* stream);

extern "C" int mvg_linear_ordered(const float* A, int lda, const float* W, const float* bias, float* out, int ldc,
                                  const uint8_t* rowmask, int relu, int M, int N, int K, const int32_t* order,
                                  const uint8_t* inside, const float* masked_row, void* stream) {
  if (!A || !W || !out || !order || !inside || !masked_row || M < 0 || N <= 0 || K <= 0) return MVG_E_BADARG;
  if (M == 0) return 0;
  if (K % 32 != 0 || N % 8 != 0 || (lda % 4) != 0 || (ldc % 4) != 0) return MVG_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(out) |
       reinterpret_cast<uintptr_t>(masked_row)) % 16 != 0)
    return MVG_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const long tiles128 = (long)((N + 127) / 128) * ((M + 127) / 128);
  if (g_linear_tiles && tiles128 < 384 && M > 64)
    return launch_linear_idx<64>(A, lda, W, bias, out, ldc, rowmask, relu, M, N, K, order, inside, masked_row, st);
  return launch_linear_idx<128>(A, lda, W, bias, out, ldc, rowmask, relu, M, N, K, order, inside, masked_row, st);
}


extern "C" int mvg_linear_wgrad_bias_f32(const float* dY, int ldy, const float* X, int ldx, float* partial, float* partial_db, float* dW,
                                         float* db, int rows, int N, int K, int splits, void* stream) {
  if (!dY || !X || !partial || !dW || rows <= 0 || N <= 0 || K <= 0 || splits <= 0) return MVG_E_BADARG;
  if (db && !partial_db) return MVG_E_BADARG;
  if (N % 4 != 0 || K % 4 != 0 || ldy % 4 != 0 || ldx % 4 != 0 || ldy < N || ldx < K) return MVG_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(partial) | reinterpret_cast<uintptr_t>(dW)) % 16 != 0)
    return MVG_E_BADARG;
  const int rps = ((rows + splits - 1) / splits + 31) / 32 * 32;
  dim3 grid((K + 127) / 128, (N + 127) / 128, splits);
  hipLaunchKernelGGL(wgrad_f32s_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dY, (long)ldy, X, (long)ldx, partial, rows, N, K, rps,
                     db ? partial_db : (float*)nullptr);
  MVG_LAUNCH_CHECK();
  const long nk4 = (long)N * K / 4;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nk4 + (db ? N : 0) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial,
                     partial_db, dW, db, splits, nk4, N);
  MVG_LAUNCH_CHECK();
  return 0;
}


extern "C" int mvg_linear(const void* A, int a_dtype, int lda, const void* W, int w_dtype, const float* bias, void* out,
                          int out_dtype, int ldc, const uint8_t* rowmask, int relu, int M, int N, int K, void* stream) {
  return mvg_linear_sum(A, nullptr, a_dtype, lda, W, w_dtype, bias, out, out_dtype, ldc, rowmask, relu, M, N, K, stream);
}

extern "C" int mvg_linear_sum(const void* A, const void* A2, int a_dtype, int lda, const void* W, int w_dtype, const float* bias,
                              void* out, int out_dtype, int ldc, const uint8_t* rowmask, int relu, int M, int N, int K,
                              void* stream) {
  if (!A || !W || !out || M < 0 || N <= 0 || K <= 0) return MVG_E_BADARG;
  if (A2 && (a_dtype != MVG_F32 || (reinterpret_cast<uintptr_t>(A2) % 16) != 0)) return MVG_E_BADARG;
  if (M == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool bf = (w_dtype == MVG_BF16);
  if (!bf && w_dtype != MVG_F32) return MVG_E_BADARG;
  if (K % (bf ? 64 : 32) != 0 || N % 8 != 0) return MVG_E_BADARG;
  const int a_el = (a_dtype == MVG_BF16) ? 2 : 4, o_el = (out_dtype == MVG_BF16) ? 2 : 4;
  if (((long)lda * a_el) % 16 != 0 || (reinterpret_cast<uintptr_t>(A) % 16) != 0 || (reinterpret_cast<uintptr_t>(W) % 16) != 0)
    return MVG_E_BADARG;
  if ((((long)ldc * o_el) % 16) != 0 || (reinterpret_cast<uintptr_t>(out) % 16) != 0) return MVG_E_BADARG;
  if (!bf) {
    if (a_dtype != MVG_F32) return MVG_E_BADARG;   // fp32 MFMA path takes fp32 activations
    if (out_dtype == MVG_F32) return launch_linear<float, false, float>(A, A2, lda, W, bias, out, ldc, rowmask, relu, M, N, K, st);
    if (out_dtype == MVG_BF16) return launch_linear<float, false, bf16_t>(A, A2, lda, W, bias, out, ldc, rowmask, relu, M, N, K, st);
    return MVG_E_BADARG;
  }
  if (a_dtype == MVG_BF16) {
    if (out_dtype == MVG_F32) return launch_linear<bf16_t, true, float>(A, nullptr, lda, W, bias, out, ldc, rowmask, relu, M, N, K, st);
    if (out_dtype == MVG_BF16) return launch_linear<bf16_t, true, bf16_t>(A, nullptr, lda, W, bias, out, ldc, rowmask, relu, M, N, K, st);
  } else if (a_dtype == MVG_F32) {
    if (out_dtype == MVG_F32) return launch_linear<float, true, float>(A, A2, lda, W, bias, out, ldc, rowmask, relu, M, N, K, st);
    if (out_dtype == MVG_BF16) return launch_linear<float, true, bf16_t>(A, A2, lda, W, bias, out, ldc, rowmask, relu, M, N, K, st);
  }
  return MVG_E_BADARG;
}
