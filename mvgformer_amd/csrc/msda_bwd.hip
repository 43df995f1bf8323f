// Deterministic backward of the multi-scale deformable sampling op for gfx950 (D = 32 channels per head).
//
// Reference: deform_cuda.cu:94-164 + cuh:312-413 -- one thread per (query, head, channel) walks its L*P samples and
// atomicAdd's four bilinear contributions per sample into grad_value (fp32 global atomics; the D = 32 variant at least
// reduces grad_sampling_loc / grad_attn_weight in shared memory).  The drop-in of round 1 (msda_bwd_kernel, csrc/msda.hip)
// did the same with wavefront shuffles: 377 M fp32 global atomics per cfg-2 view-layer, 988 us, and a grad_value whose
// last bits depend on the order the atomics happen to land in.
//
// Here grad_value is a GATHER problem solved in LDS, and every sum is order-independent:
//   1. the samples are binned by DESTINATION: key = (image, level tile of T x T pixels, head) of the sample's upper-left
//      corner pixel (bin_count -> exclusive scan -> bin_scatter; integer atomics only, and the order inside a bin does not
//      matter, see 3.);
//   2. one workgroup per non-empty bin loads the (T+1) x (T+1) x 32 value patch of its tile into LDS -- the four corners of
//      every sample of the bin are inside it, so grad_sampling_loc / grad_attn_weight need no global gather at all --, and
//      keeps a (T+1) x (T+1) x 32 accumulator patch next to it;
//   3. per wavefront pass, 64 samples: each lane prepares one sample (location, weights, patch cells), then 8 lanes x 4
//      channels work on each sample: read grad_output[q, m, :] (128 contiguous bytes per sample), add the four corner
//      contributions to the LDS accumulator as 64-bit FIXED-POINT integers (ds_add_u64: integer addition is associative, so
//      the result does not depend on which wavefront gets there first), reduce the two location gradients and the weight
//      gradient over the channels in a fixed order and write them (one writer per sample);
//   4. the patch is added to a global 64-bit accumulator (integer atomics again: neighbouring tiles share their border
//      pixels), and a last pass converts it to the fp32 grad_value.
// Fixed point: a contribution grad_output * attn_weight * bilinear weight of image n is bounded by
// bound_n = max |grad_output[n]| * max |attn_weight[n]| (finite entries; the bilinear weights are <= 1).  Contributions are scaled
// by 2^30 / bound_n (clamped to 2^126 for bounds below 2^-96) and rounded to an int32 (exact: fp32 carries 24 bits), then summed as
// 64-bit integers -- no overflow before 2^33 contributions per pixel, each rounded to 2^-31 bound_n, 2^7 times finer than the fp32
// rounding (relative to a sum of the size of bound_n) of the atomics it replaces.  The scale is PER IMAGE and includes the
// attention weights' own maximum, so a generic caller (un-normalised weights > 1, one image of the batch with outlier
// gradients, gradients of 1e-35) neither saturates the int32 nor loses the other images' precision.  Dynamic range: what is
// below 2^-31 of its image's bound rounds to zero (fp32 atomics would keep such a term only while the running sum is as small).
// Everything is bit-reproducible run to run.
#include "common.h"

namespace {

constexpr int BW_T = 8;                   // tile edge (pixels); the LDS patch is (T+1)^2 pixels
constexpr int BW_P = BW_T + 1;
constexpr int BW_D = 32;
constexpr int BW_NT = 256;                // 8 samples x 32 channels per pass
constexpr int BW_SPLIT = 1;               // workgroups per bin (interleaved over its entries): 2 / 4 / 8 measured 635 / 770 / 761 us against 617 us
constexpr float BW_FIX = 1073741824.f;     // 2^30: a contribution (|.| <= max |grad_output|) is an exact int32

struct BwLevels {
  int H[MVG_MAX_LEVELS], W[MVG_MAX_LEVELS], start[MVG_MAX_LEVELS];
  int tiles_w[MVG_MAX_LEVELS], tile0[MVG_MAX_LEVELS + 1];      // tiles per row, first tile of a level
  int L, T_total;
};

// the reference's sample test and upper-left pixel (cuh:295-301); returns false for samples that contribute nothing
__device__ __forceinline__ bool sample_anchor(const float lx, const float ly, const int H, const int W, float& h_im, float& w_im,
                                              int& h_low, int& w_low) {
  h_im = ly * (float)H - 0.5f;
  w_im = lx * (float)W - 0.5f;
  if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) return false;   // also NaN
  h_low = (int)floorf(h_im);
  w_low = (int)floorf(w_im);
  return true;
}

// blockIdx.y = image, blockIdx.z = tensor (0: grad_output, 1: attn_weight); out[2 * image + tensor] = max |finite entry|
__global__ __launch_bounds__(256) void bw_absmax_kernel(const float* __restrict__ x0, long n0, const float* __restrict__ x1,
                                                        long n1, unsigned* __restrict__ out) {
  const long n = blockIdx.z ? n1 : n0;
  const float* __restrict__ x = (blockIdx.z ? x1 : x0) + (long)blockIdx.y * n;
  out += 2 * blockIdx.y + blockIdx.z;
  float m = 0.f;
  const long lead = min(n, (long)((4 - ((reinterpret_cast<uintptr_t>(x) >> 2) & 3)) & 3));      // floats up to 16-byte alignment
  const long n4 = (n - lead) >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + lead + 4 * i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = fabsf(v[k]);
      m = (a > m && a < INFINITY) ? a : m;      // NaN / Inf do not set the scale
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 8) {     // the unaligned head and the tail
    const long i = threadIdx.x < 4 ? (long)threadIdx.x : lead + 4 * n4 + (threadIdx.x - 4);
    const bool in = threadIdx.x < 4 ? i < lead : i < n;
    const float a = in ? fabsf(x[i]) : 0.f;
    m = (a > m && a < INFINITY) ? a : m;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  __shared__ float wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {                                   // one atomic per workgroup (4096 on one address took 40 us)
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (m > 0.f) atomicMax(out, __float_as_uint(m));        // non-negative floats order like their bit patterns
  }
}

// scale of image n's fixed-point contributions and its inverse (the same two expressions in every kernel that needs them)
__device__ inline float bw_fix_of(const unsigned* __restrict__ scale_bits, int n) {
  const float bound = __uint_as_float(scale_bits[2 * n]) * __uint_as_float(scale_bits[2 * n + 1]);
  return (bound > 0.f && bound < INFINITY) ? fminf(BW_FIX / bound, 0x1p126f) : 0.f;
}
__device__ inline double bw_inv_of(float fix) { return fix > 0.f ? 1.0 / (double)fix : 0.0; }

// Binning = a counting sort of the samples by (tile, head) per image, WITHOUT global atomics (2.9 M atomics on ~5 000
// counters serialise on their hot cache lines: 808 us per pass in the first version).  The samples of an image are cut into
// BW_PARTS contiguous parts; a workgroup histograms its part in LDS (bw_part_kernel<0>), a second kernel turns the (part, bin)
// counts into exclusive prefixes over the parts and per-bin totals, a one-workgroup scan of the totals gives the bin
// offsets, and bw_part_kernel<1> replays the part with LDS cursors.  The order inside a bin is whatever the LDS atomics
// produce: irrelevant, every sum downstream is order-independent.
constexpr int BW_PARTS = 128;
constexpr int BW_MAX_BPI = 12288;         // bins per image that fit the LDS histogram (48 KB)

// sample s of image n (s = (q * M + m) * LP + lp) -> its bin inside the image (tile * M + m), or -1
__device__ __forceinline__ int sample_bin(const float* __restrict__ loc, const BwLevels& lv, long n, long s, int Lq, int M, int P,
                                          int LP) {
  const int lp = (int)(s % LP);
  const int m = (int)((s / LP) % M);
  const int l = lp / P;
  const float2 xy = *reinterpret_cast<const float2*>(loc + (n * (long)Lq * M * LP + s) * 2);
  float h_im, w_im;
  int h_low, w_low;
  if (!sample_anchor(xy.x, xy.y, lv.H[l], lv.W[l], h_im, w_im, h_low, w_low)) return -1;
  return (lv.tile0[l] + (max(h_low, 0) / BW_T) * lv.tiles_w[l] + max(w_low, 0) / BW_T) * M + m;
}

// MODE 0: cnt[(n * PARTS + part) * bpi + bin] = samples of the part in the bin.
// MODE 1: cnt holds the part's first position of every bin (bw_prefix_kernel + bin offsets); writes list.
// MODE 1 also writes the zero location / weight gradients of the samples that fail the reference's bounds test: they are
// never binned, so nobody else writes them (this replaces two memsets of the whole gradient tensors).
template <int MODE>
__global__ __launch_bounds__(1024) void bw_part_kernel(const float* __restrict__ loc, BwLevels lv, int* __restrict__ cnt,
                                                       const int* __restrict__ offset, unsigned* __restrict__ list, int Lq, int M,
                                                       int P, int bpi, long per_img, long chunk, float* __restrict__ gloc,
                                                       float* __restrict__ gwgt) {
  extern __shared__ int hist[];
  const int n = blockIdx.x / BW_PARTS, part = blockIdx.x % BW_PARTS, tid = threadIdx.x;
  int* mine = cnt + ((long)n * BW_PARTS + part) * bpi;
  for (int i = tid; i < bpi; i += 1024) hist[i] = MODE == 0 ? 0 : mine[i] + offset[(long)n * bpi + i];
  __syncthreads();
  const long s0 = part * chunk, s1 = min(per_img, s0 + chunk);
  const int LP = lv.L * P;
  for (long s = s0 + tid; s < s1; s += 1024) {
    const int bin = sample_bin(loc, lv, n, s, Lq, M, P, LP);
    if (bin < 0) {
      if (MODE == 1) {
        const long sidx = (long)n * per_img + s;
        *reinterpret_cast<float2*>(gloc + sidx * 2) = make_float2(0.f, 0.f);
        gwgt[sidx] = 0.f;
      }
      continue;
    }
    const int pos = atomicAdd(&hist[bin], 1);
    if (MODE == 1) list[pos] = ((unsigned)(s / ((long)M * LP)) << 8) | (unsigned)(s % LP);      // (q, lp); m is the bin's
  }
  if (MODE == 0) {
    __syncthreads();
    for (int i = tid; i < bpi; i += 1024) mine[i] = hist[i];
  }
}

// per (image, bin): exclusive prefix of the counts over the parts (in place) and the bin total.  8 lanes per bin, 16 parts
// each (one thread per bin walked 128 dependent strided loads: 19 us for 5 040 bins), combined with a 3-step lane scan.
__global__ __launch_bounds__(256) void bw_prefix_kernel(int* __restrict__ cnt, int* __restrict__ total, int bpi) {
  static_assert(BW_PARTS % 8 == 0, "8 lanes per bin");
  constexpr int PPL = BW_PARTS / 8;
  const int n = blockIdx.y, sub = threadIdx.x >> 5, bin = blockIdx.x * 32 + (threadIdx.x & 31);
  // lanes of a wavefront: bins (lane & 31) for sub = 2 w and 2 w + 1; a bin's 8 sub-ranges live in 4 wavefronts -> combine in LDS
  __shared__ int part_sum[8][32];
  const bool ok = bin < bpi;
  int* c = cnt + (long)n * BW_PARTS * bpi + (ok ? bin : 0) + (long)sub * PPL * bpi;
  int v[PPL], run = 0;
#pragma unroll
  for (int p = 0; p < PPL; ++p) v[p] = ok ? c[(long)p * bpi] : 0;
#pragma unroll
  for (int p = 0; p < PPL; ++p) run += v[p];
  part_sum[sub][threadIdx.x & 31] = run;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int t = part_sum[k][threadIdx.x & 31];
    base += k < sub ? t : 0;
    tot += t;
  }
  if (!ok) return;
#pragma unroll
  for (int p = 0; p < PPL; ++p) {
    c[(long)p * bpi] = base;
    base += v[p];
  }
  if (sub == 0) total[(long)n * bpi + bin] = tot;
}

// exclusive scan of the bin totals over all images (one workgroup; a few 10 000 entries): offset[i], offset[nb] = all
__global__ __launch_bounds__(1024) void bw_scan_kernel(const int* __restrict__ total, int* __restrict__ offset, int nb) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024 * 4) {
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + tid * 4 + k;
      v[k] = i < nb ? total[i] : 0;
      s += v[k];
    }
    int incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if ((tid & 63) >= d) incl += up;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int t = wsum[w];
      wbase += (w < (tid >> 6)) ? t : 0;
      tot += t;
    }
    int ex = carry + wbase + incl - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + tid * 4 + k;
      if (i < nb) offset[i] = ex;
      ex += v[k];
    }
    __syncthreads();
    if (tid == 0) carry += tot;
    __syncthreads();
  }
  if (tid == 0) offset[nb] = carry;
}

// sum over the 8 lanes of a lane group (result in all of them), fixed order: quad_perm xor 1, xor 2, row_half_mirror
__device__ __forceinline__ float sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
  return v;
}

__global__ __launch_bounds__(BW_NT) void bw_reduce_kernel(
    const float* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ wgt, const float* __restrict__ gout,
    BwLevels lv, const int* __restrict__ offset, const unsigned* __restrict__ list, const unsigned* __restrict__ gmax_bits,
    unsigned long long* __restrict__ accum, float* __restrict__ gvalue, float* __restrict__ gloc, float* __restrict__ gwgt, int N,
    int S, int Lq, int M, int P) {
  __shared__ float vpatch[BW_P * BW_P][BW_D];                           // 10.1 KB
  __shared__ unsigned long long apatch[BW_P * BW_P][BW_D];              // 20.3 KB
  // BW_SPLIT workgroups share a bin (interleaved passes over its entries): bins differ a lot in size -- a coarse-level tile
  // collects 15x the samples of a fine-level one -- and one workgroup per bin left the kernel waiting for the largest ones.
  // Every workgroup adds its own patch to the global accumulator: integers, so still order-independent.
  // blockIdx = ((image, tile) * BW_SPLIT + split) * M + head: with M = 8, blockIdx % 8 -- the XCD the block is dispatched to --
  // is the HEAD, so an XCD's 4-MB L2 holds one head's slice of grad_output (2 MB at cfg-2) and of value; a mapping that put
  // all heads on every XCD ran 2.3x slower (its grad_output rows came from HBM / Infinity Cache)
  const int m_ = blockIdx.x % M, rest_ = blockIdx.x / M;
  const int split = rest_ % BW_SPLIT, bin = (rest_ / BW_SPLIT) * M + m_;
  static_assert(BW_SPLIT == 1, "the interior of a tile is written by exactly one workgroup");
  const int e0 = offset[bin], e1 = offset[bin + 1];
  const int m = bin % M;
  const int nt = bin / M;
  const int n = nt / lv.T_total, tile = nt - n * lv.T_total;
  int l = 0;
#pragma unroll
  for (int k = 1; k < MVG_MAX_LEVELS; ++k)
    if (k < lv.L && tile >= lv.tile0[k]) l = k;
  const int tl = tile - lv.tile0[l];
  const int H = lv.H[l], W = lv.W[l];
  const int y0 = (tl / lv.tiles_w[l]) * BW_T, x0 = (tl % lv.tiles_w[l]) * BW_T;     // patch origin (pixel)
  const int tid = threadIdx.x;
  const long row_stride = (long)M * BW_D;
  // Pixels with x % T != 0 and y % T != 0 receive contributions from THIS tile only (a neighbour's patch reaches just its first
  // row / column): they are written straight to grad_value, no global accumulator, no memset, no conversion pass for 49 of 64
  // pixels.  The first row and column of every tile are shared with up to three neighbours and go through `accum`.
  float* gbase = gvalue + ((long)n * S + lv.start[l]) * row_stride + (long)m * BW_D;
  if (e0 + split * (BW_NT / 64) * 64 >= e1) {                // an empty bin still owns its interior: zeros
    for (int i = tid; i < (BW_T - 1) * (BW_T - 1) * BW_D; i += BW_NT) {
      const int px = i >> 5, ch = i & 31;
      const int y = y0 + 1 + px / (BW_T - 1), x = x0 + 1 + px % (BW_T - 1);
      if (y < H && x < W) gbase[((long)y * W + x) * row_stride + ch] = 0.f;
    }
    return;
  }
  const float* vbase = value + ((long)n * S + lv.start[l]) * row_stride + (long)m * BW_D;
  for (int i = tid; i < BW_P * BW_P * BW_D; i += BW_NT) {
    const int px = i >> 5, ch = i & 31;
    const int y = y0 + px / BW_P, x = x0 + px % BW_P;
    vpatch[px][ch] = (y < H && x < W) ? vbase[((long)y * W + x) * row_stride + ch] : 0.f;
    apatch[px][ch] = 0ull;
  }
  __syncthreads();
  const float fix = bw_fix_of(gmax_bits, n);
  const int LP = lv.L * P;
  const float Hf = (float)H, Wf = (float)W;
  // Two phases per pass of 64 samples per wavefront (the first form recomputed every sample's coordinates, corner weights and
  // patch indices on all 32 channel lanes -- the kernel was bound by VALU instruction count):
  //   A. lane j of the wavefront prepares sample j of the pass: id -> location / weight (global loads, 64 in flight per
  //      wavefront), bounds, the four corner weights x attention weight, the four patch cells;
  //   B. 32 steps: lanes 0-31 / 32-63 (= the 32 channels) take samples 2 k / 2 k + 1, fetch the prepared scalars from their
  //      owner lane (ds_bpermute) and do the per-channel work: grad_output, four patch reads, four fixed-point LDS adds,
  //      the three partial sums and their reduction over the channels.
  const int wave = tid >> 6, lane = tid & 63;
  constexpr int NW = BW_NT / 64;
  for (int eb = e0 + (split * NW + wave) * 64; eb < e1; eb += BW_SPLIT * NW * 64) {
    // ---- phase A
    const int e = eb + lane;
    const bool live = e < e1;
    const unsigned id = list[min(e, e1 - 1)];
    const int q = (int)(id >> 8), lp = (int)(id & 255u);
    const long qm = ((long)n * Lq + q) * M + m;
    const long sidx = qm * LP + lp;
    const float2 xy = *reinterpret_cast<const float2*>(loc + sidx * 2);
    const float aw = wgt[sidx];
    const float h_im = xy.y * Hf - 0.5f, w_im = xy.x * Wf - 0.5f;
    const float hl_f = floorf(h_im), wl_f = floorf(w_im);
    const int h_low = (int)hl_f, w_low = (int)wl_f;
    const float lh = h_im - hl_f, lw = w_im - wl_f;
    // patch coordinates of the four corners; a corner outside the map (row / col -1 or H / W) contributes nothing (cuh:66-88)
    const int py = h_low - y0, pxx = w_low - x0;                         // -1 .. T-1
    const bool t_ok = h_low >= 0, b_ok = h_low + 1 <= H - 1, l_ok = w_low >= 0, r_ok = w_low + 1 <= W - 1;
    // packed: 4 clamped cell indices (7 bits each, < 81) and the 4 corner-valid bits
    const unsigned cells = (unsigned)(max(py, 0) * BW_P + max(pxx, 0)) | ((unsigned)(max(py, 0) * BW_P + pxx + 1) << 7) |
                           ((unsigned)((py + 1) * BW_P + max(pxx, 0)) << 14) | ((unsigned)((py + 1) * BW_P + pxx + 1) << 21) |
                           ((unsigned)(live && t_ok && l_ok) << 28) | ((unsigned)(live && t_ok && r_ok) << 29) |
                           ((unsigned)(live && b_ok && l_ok) << 30) | ((unsigned)(live && b_ok && r_ok) << 31);
    const int gq = (int)(qm);                                            // row of grad_output ((n * Lq + q) * M + m < 2^31)
    // ---- phase B: 8 samples per step -- lane (g = lane >> 3, j = lane & 7) takes channels 4 j .. 4 j + 3 of the sample prepared
    //      by lane 8 k + g.  (With 32 lanes x 1 channel per sample the per-sample scalars -- cell indices, corner flags and
    //      weights, addresses -- were recomputed on 32 lanes and the kernel was bound by VALU issue: 170 M wave-instructions
    //      per launch; 4 channels per lane amortise them 4x and shorten the channel reduction from 5 to 3 DPP steps.)
    constexpr int BW_U = 2;
    const int g8 = lane >> 3, j = lane & 7;
    const int steps = min(8, (e1 - eb + 7) >> 3);
    for (int k0 = 0; k0 < steps; k0 += BW_U) {
      unsigned cl[BW_U];
      int s_gq[BW_U];
      f32x4 go[BW_U];
#pragma unroll
      for (int u = 0; u < BW_U; ++u) {
        const int src = min(8 * (k0 + u) + g8, 63);
        cl[u] = (k0 + u < steps) ? (unsigned)__shfl((int)cells, src, 64) : 0u;
        s_gq[u] = __shfl(gq, src, 64);
        go[u] = *reinterpret_cast<const f32x4*>(gout + (long)s_gq[u] * BW_D + 4 * j);
      }
#pragma unroll
      for (int u = 0; u < BW_U; ++u) {
        const int src = min(8 * (k0 + u) + g8, 63);
        const float a_w = __shfl(aw, src, 64);
        const float s_lh = __shfl(lh, src, 64), s_lw = __shfl(lw, src, 64);
        const int s_lp = __shfl((int)id, src, 64) & 255;
        const float s_hh = 1.f - s_lh, s_hw = 1.f - s_lw;
        const float s_w1 = s_hh * s_hw, s_w2 = s_hh * s_lw, s_w3 = s_lh * s_hw, s_w4 = s_lh * s_lw;
        const bool any = (cl[u] >> 28) != 0u;                            // uniform over the 8 lanes of the sample
        const bool k1 = (cl[u] >> 28) & 1u, k2 = (cl[u] >> 29) & 1u, k3 = (cl[u] >> 30) & 1u, k4 = (cl[u] >> 31) & 1u;
        const int i1 = cl[u] & 127u, i2 = (cl[u] >> 7) & 127u, i3 = (cl[u] >> 14) & 127u, i4 = (cl[u] >> 21) & 127u;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 v1 = k1 ? *reinterpret_cast<const f32x4*>(&vpatch[i1][4 * j]) : z4;
        const f32x4 v2 = k2 ? *reinterpret_cast<const f32x4*>(&vpatch[i2][4 * j]) : z4;
        const f32x4 v3 = k3 ? *reinterpret_cast<const f32x4*>(&vpatch[i3][4 * j]) : z4;
        const f32x4 v4 = k4 ? *reinterpret_cast<const f32x4*>(&vpatch[i4][4 * j]) : z4;
        const f32x4 top = go[u] * a_w;
        // grad_value: fixed-point contributions, |top * w * fix| <= 2^30: exact in int32, accumulated as two's complement int64.
        // Branch-free: a corner outside the map adds 0 to a legal (clamped) cell -- rare, and cheaper than 16 exec-mask changes.
        const f32x4 ft = top * fix;
        const float f1 = k1 ? s_w1 : 0.f, f2 = k2 ? s_w2 : 0.f, f3 = k3 ? s_w3 : 0.f, f4 = k4 ? s_w4 : 0.f;
        // Every patch cell is one 256-byte row = all 64 LDS banks, and lane (sample g8, j) adds to banks 8 j + 2 ch (+1) of ITS
        // cell: with the same channel order on all 8 samples of the wavefront every instruction hit 16 banks 8 deep (PMC: 65 % of
        // the kernel's LDS cycles were bank conflicts).  Sample g8 therefore walks its 4 channels starting at channel g8 & 3:
        // 4 of the 8 samples are on disjoint banks at any step (2-way is the floor: 512 bytes per instruction on a 256-byte LDS).
        const int rot = g8 & 3;
        const f32x4 fa = (rot & 1) ? f32x4{ft[1], ft[2], ft[3], ft[0]} : ft;
        const f32x4 fr = (rot & 2) ? f32x4{fa[2], fa[3], fa[0], fa[1]} : fa;         // fr[t] = ft[(t + rot) & 3]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int cc = 4 * j + ((t + rot) & 3);
          atomicAdd(&apatch[i1][cc], (unsigned long long)(long long)__float2int_rn(fr[t] * f1));
          atomicAdd(&apatch[i2][cc], (unsigned long long)(long long)__float2int_rn(fr[t] * f2));
          atomicAdd(&apatch[i3][cc], (unsigned long long)(long long)__float2int_rn(fr[t] * f3));
          atomicAdd(&apatch[i4][cc], (unsigned long long)(long long)__float2int_rn(fr[t] * f4));
        }
        // d/d(w_im), d/d(h_im), d/d(attn) (cuh:128-167): this lane's 4 channels in ascending order, then the 8 lanes of the
        // sample with a fixed DPP tree
        float g_w = 0.f, g_h = 0.f, g_a = 0.f;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          g_w += (s_hh * (v2[ch] - v1[ch]) + s_lh * (v4[ch] - v3[ch])) * top[ch];
          g_h += (s_hw * (v3[ch] - v1[ch]) + s_lw * (v4[ch] - v2[ch])) * top[ch];
          g_a += go[u][ch] * (s_w1 * v1[ch] + s_w2 * v2[ch] + s_w3 * v3[ch] + s_w4 * v4[ch]);
        }
        g_w = sum8(g_w) * Wf;
        g_h = sum8(g_h) * Hf;
        g_a = sum8(g_a);
        if (j == 0 && any) {   // a live sample has at least one corner inside the map (it passed the reference's bounds test)
          const long s_sidx = (long)s_gq[u] * LP + s_lp;
          *reinterpret_cast<float2*>(gloc + s_sidx * 2) = make_float2(g_w, g_h);
          gwgt[s_sidx] = g_a;
        }
      }
    }
  }
  __syncthreads();
  unsigned long long* abase = accum + (((long)n * S + lv.start[l]) * M + m) * BW_D;
  const double inv = bw_inv_of(fix);                                     // bw_shared_kernel<1>'s arithmetic
  for (int i = tid; i < BW_P * BW_P * BW_D; i += BW_NT) {
    const int px = i >> 5, ch = i & 31;
    const int py = px / BW_P, pxx = px % BW_P;
    const int y = y0 + py, x = x0 + pxx;
    const unsigned long long v = apatch[px][ch];
    if (y >= H || x >= W) continue;
    if (py % BW_T != 0 && pxx % BW_T != 0) gbase[((long)y * W + x) * row_stride + ch] = (float)((double)(long long)v * inv);
    else if (v != 0ull) atomicAdd(abase + ((long)y * W + x) * M * BW_D + ch, v);
  }
}

// The shared pixels (x % T == 0 or y % T == 0) of every map, one workgroup per (map row, 32-pixel segment):
//   MODE 0: accum <- 0 (before the reduce kernel; also resets the image's scale words)   MODE 1: grad_value <- accum / scale
template <int MODE>
__global__ __launch_bounds__(256) void bw_shared_kernel(unsigned long long* __restrict__ accum, unsigned* __restrict__ gmax_bits,
                                                        float* __restrict__ gvalue, BwLevels lv, int S, int M, int rows_per_img) {
  const int n = blockIdx.x / rows_per_img;
  if (MODE == 0 && blockIdx.x == n * rows_per_img && blockIdx.y == 0 && threadIdx.x < 2) gmax_bits[2 * n + threadIdx.x] = 0u;
  int y = blockIdx.x - n * rows_per_img, l = 0;
  while (l + 1 < lv.L && y >= lv.H[l]) y -= lv.H[l++];
  const int W = lv.W[l], xa = blockIdx.y * 32;
  if (xa >= W) return;
  const int step = (y % BW_T == 0) ? 1 : BW_T, xb = min(W, xa + 32);
  const int per_px = M * BW_D;
  const double inv = MODE == 1 ? bw_inv_of(bw_fix_of(gmax_bits, n)) : 0.0;
  for (int x = xa; x < xb; x += step) {                       // xa is a multiple of 32, hence of T
    const long base = (((long)n * S + lv.start[l]) + (long)y * W + x) * per_px;
    for (int e = threadIdx.x; e < per_px; e += 256) {
      if (MODE == 0) accum[base + e] = 0ull;
      else gvalue[base + e] = (float)((double)(long long)accum[base + e] * inv);
    }
  }
}

int fill_bw_levels(BwLevels* lv, const int64_t* shapes_host, const int64_t* starts_host, int L) {
  if (L < 1 || L > MVG_MAX_LEVELS) return MVG_E_BADARG;
  lv->L = L;
  int t = 0;
  for (int l = 0; l < L; ++l) {
    lv->H[l] = (int)shapes_host[2 * l];
    lv->W[l] = (int)shapes_host[2 * l + 1];
    lv->start[l] = (int)starts_host[l];
    if (lv->H[l] <= 0 || lv->W[l] <= 0) return MVG_E_BADARG;
    lv->tiles_w[l] = (lv->W[l] + BW_T - 1) / BW_T;
    lv->tile0[l] = t;
    t += lv->tiles_w[l] * ((lv->H[l] + BW_T - 1) / BW_T);
  }
  lv->tile0[L] = t;
  lv->T_total = t;
  return 0;
}

struct BwLayout {
  size_t accum, cnt, total, offset, list, gmax, bytes;
};

BwLayout bw_layout(long N, long S, long M, long Lq, long LP, long bpi) {
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  BwLayout w;
  w.accum = 0;
  w.cnt = up((size_t)N * S * M * BW_D * 8);
  w.total = w.cnt + up((size_t)N * BW_PARTS * bpi * 4);
  w.offset = w.total + up((size_t)N * bpi * 4);
  w.list = w.offset + up((size_t)(N * bpi + 1) * 4);
  w.gmax = w.list + up((size_t)N * Lq * M * LP * 4);
  w.bytes = w.gmax + up((size_t)N * 8);
  return w;
}

}  // namespace

extern "C" {

size_t mvg_msda_backward_det_workspace(int N, int S, int M, int D, int L, int Lq, int P, const int64_t* shapes_host) {
  if (N <= 0 || S <= 0 || M <= 0 || D != BW_D || L < 1 || L > MVG_MAX_LEVELS || Lq <= 0 || P <= 0 || !shapes_host) return 0;
  if ((long)L * P > 256 || Lq >= (1 << 24)) return 0;
  BwLevels lv;
  int64_t zeros[MVG_MAX_LEVELS] = {0};
  if (fill_bw_levels(&lv, shapes_host, zeros, L)) return 0;
  const long bpi = (long)lv.T_total * M;
  if (bpi > BW_MAX_BPI || (long)N * bpi * BW_SPLIT > 0x3fffffffL || (long)N * Lq * M * L * P > 0x7fffffffL) return 0;
  return bw_layout(N, S, M, Lq, (long)L * P, bpi).bytes;
}

int mvg_msda_backward_det_f32(const float* value, const int64_t* shapes_host, const int64_t* starts_host,
                              const float* sampling_loc, const float* attn_weight, const float* grad_output,
                              float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, int N, int S, int M,
                              int D, int L, int Lq, int P, void* workspace, size_t workspace_bytes, void* stream) {
  if (!value || !shapes_host || !starts_host || !sampling_loc || !attn_weight || !grad_output || !grad_value ||
      !grad_sampling_loc || !grad_attn_weight || !workspace)
    return MVG_E_BADARG;
  const size_t need = mvg_msda_backward_det_workspace(N, S, M, D, L, Lq, P, shapes_host);
  if (need == 0 || workspace_bytes < need) return MVG_E_BADARG;
  BwLevels lv;
  int e = fill_bw_levels(&lv, shapes_host, starts_host, L);
  if (e) return e;
  hipStream_t st = (hipStream_t)stream;
  const long LP = (long)L * P, bpi = (long)lv.T_total * M, nbins = (long)N * bpi;
  const long per_img = (long)Lq * M * LP, nsamp = (long)N * per_img;
  const BwLayout w = bw_layout(N, S, M, Lq, LP, bpi);
  char* ws = reinterpret_cast<char*>(workspace);
  unsigned long long* accum = reinterpret_cast<unsigned long long*>(ws + w.accum);
  int* cnt = reinterpret_cast<int*>(ws + w.cnt);
  int* total = reinterpret_cast<int*>(ws + w.total);
  int* offset = reinterpret_cast<int*>(ws + w.offset);
  unsigned* list = reinterpret_cast<unsigned*>(ws + w.list);
  unsigned* gmax = reinterpret_cast<unsigned*>(ws + w.gmax);
  hipError_t he = hipSuccess;
  (void)nsamp;
  static bool configured[MVG_MAX_DEVICES] = {};       // > 64 KB of dynamic LDS is a per-device function attribute
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVG_MAX_DEVICES) return MVG_E_BADARG;
  if (!configured[dev]) {
    he = hipFuncSetAttribute(reinterpret_cast<const void*>(&bw_part_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             BW_MAX_BPI * 4);
    if (he == hipSuccess)
      he = hipFuncSetAttribute(reinterpret_cast<const void*>(&bw_part_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               BW_MAX_BPI * 4);
    if (he != hipSuccess) return (int)he;
    configured[dev] = true;
  }
  const long ngo = (long)N * Lq * M * BW_D;
  int rows_per_img = 0, max_w = 0;
  for (int l = 0; l < L; ++l) {
    rows_per_img += lv.H[l];
    max_w = lv.W[l] > max_w ? lv.W[l] : max_w;
  }
  const dim3 shared_grid((unsigned)(N * rows_per_img), (unsigned)((max_w + 31) / 32));
  hipLaunchKernelGGL((bw_shared_kernel<0>), shared_grid, dim3(256), 0, st, accum, gmax, grad_value, lv, S, M, rows_per_img);
  hipLaunchKernelGGL(bw_absmax_kernel, dim3(N >= 8 ? 64 : 256, N, 2), dim3(256), 0, st, grad_output, ngo / N, attn_weight,
                     per_img, gmax);
  const long chunk = (per_img + BW_PARTS - 1) / BW_PARTS;
  const size_t lds = (size_t)bpi * 4;
  hipLaunchKernelGGL((bw_part_kernel<0>), dim3(N * BW_PARTS), dim3(1024), lds, st, sampling_loc, lv, cnt, (const int*)nullptr,
                     (unsigned*)nullptr, Lq, M, P, (int)bpi, per_img, chunk, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(bw_prefix_kernel, dim3((unsigned)((bpi + 31) / 32), N), dim3(256), 0, st, cnt, total, (int)bpi);
  hipLaunchKernelGGL(bw_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)total, offset, (int)nbins);
  hipLaunchKernelGGL((bw_part_kernel<1>), dim3(N * BW_PARTS), dim3(1024), lds, st, sampling_loc, lv, cnt, (const int*)offset,
                     list, Lq, M, P, (int)bpi, per_img, chunk, grad_sampling_loc, grad_attn_weight);
  hipLaunchKernelGGL(bw_reduce_kernel, dim3((unsigned)(nbins * BW_SPLIT)), dim3(BW_NT), 0, st, value, sampling_loc, attn_weight,
                     grad_output, lv, (const int*)offset, (const unsigned*)list, (const unsigned*)gmax, accum, grad_value,
                     grad_sampling_loc, grad_attn_weight, N, S, Lq, M, P);
  hipLaunchKernelGGL((bw_shared_kernel<1>), shared_grid, dim3(256), 0, st, accum, gmax, grad_value, lv, S, M, rows_per_img);
  MVG_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
