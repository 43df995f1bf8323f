// Device building blocks of the G-sampling kernel (csrc/msda.hip: msda_gsamp_kernel).  See msda.hip for the design notes.
#pragma once
#include "common.h"

// Index-safe copy of an image coordinate: equal to x wherever the sample counts (-1 < x < n), -2 <= . <= n+1 otherwise
// (v_med3_f32 returns the minimum of the non-NaN operands when x is NaN: -2).  floor, the float->int conversion and
// the +1 of the lower-right corner are taken from it, so that they can neither overflow nor be undefined for NaN,
// +-Inf or 1e30 locations; the reference's bounds test (cuh:298) still looks at the original value and rejects those.
__device__ __forceinline__ float index_safe(float x, float n) { return __builtin_amdgcn_fmed3f(x, -2.f, n + 1.f); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return pack_bf16(lo, hi);
}
template <int S>   // value of lane S of every quad (v_mov_b32_dpp quad_perm:[S,S,S,S])
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, S * 0x55, 0xf, 0xf, true);
}

// quad_bcast with the source lane as an unrolled loop index (the DPP control must be an immediate)
__device__ __forceinline__ unsigned quad_bcast_rt(unsigned v, int lane) {
  switch (lane) {
    case 0: return quad_bcast<0>(v);
    case 1: return quad_bcast<1>(v);
    case 2: return quad_bcast<2>(v);
    default: return quad_bcast<3>(v);
  }
}

// acc += w.lo * d.lo + w.hi * d.hi with w read from lane S of the quad: the quad broadcast of the packed blend weights is a DPP
// operand of the v_dot2c itself (hipcc keeps a v_mov_b32_dpp per broadcast: the DPP combiner does not touch tied-accumulator VOP2s).
// The compiler does not see the DPP read: the 2 wait states a DPP source needs behind the VALU write of that register hold by
// construction (the weights are converted a batch ahead) and are checked on the built ISA by tools/check_dpp_hazard.py.
template <int S>
__device__ __forceinline__ void dot2c_quad(float& acc, unsigned w, unsigned d) {
  asm("v_dot2c_f32_bf16_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(d), "n"(S));
}

// clamp to [0, hi] in one instruction (hipcc cannot prove 0 <= hi and emits v_max + v_min)
__device__ __forceinline__ int clamp0_i32(int x, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
  return r;
}

// One lane's sample of gather batch `it` of a (pair, head): sample index it*4 + sub, level (it*4)/8.  Branch-free:
// packed (left, right) bf16 weights of the top / bottom pixel pair, the byte offsets of the top-left / bottom-left
// pixels inside the head plane and the byte distance to the right-hand pixel (0 at the image border, else 64).
// Round 6 (instruction diet, same results bit for bit): the reference's bounds test (cuh:298) and the zero padding of the
// four corners (cuh:66-88) are ONE unsigned compare per row / column of the 2 x 2 footprint, applied to the 1-D factors
// before the products -- a location outside the map lands on row / column -2, n or n + 1 through index_safe and fails both
// compares of its axis (h_raw == -1 exactly: row -1 fails, row 0 passes with factor lh == 0).  It was 4 float compares for
// the bounds test + 4 integer ones for the corners + 7 mask combinations + 5 selects on the products.  Pixel indices: one
// v_med3_i32 per clamp, rows through v_mad_u32_u24 on the pre-shifted row pitch.
template <int L>
__device__ __forceinline__ void gsamp_coords(int it, const float* __restrict__ sc, float mx, const LevelTable& lv,
                                             int sub, unsigned& wt, unsigned& wb, unsigned& ot, unsigned& ob,
                                             unsigned& dx) {
  constexpr int P = 8, NB = 4, LP = L * P;
  const int l = (it * NB) / P;
  const int H = lv.H[l], W = lv.W[l];
  const float Wf = lv.Wf[l], Hf = lv.Hf[l];
  const float2 rr = *reinterpret_cast<const float2*>(sc + 3 * LP + 2 * l);   // the pair's reference point at level l
  const float rx = rr.x, ry = rr.y;
  const float lgs = sc[it * NB + sub];
  const float2 of = *reinterpret_cast<const float2*>(sc + LP + (it * NB + sub) * 2);
  const float lx = rx + of.x * lv.invW[l], ly = ry + of.y * lv.invH[l];               // projattn.py:186-191
  const float h_raw = ly * Hf - 0.5f, w_raw = lx * Wf - 0.5f;                         // cuh:295-296
  const float h_im = __builtin_amdgcn_fmed3f(h_raw, -2.f, lv.Hp1[l]), w_im = __builtin_amdgcn_fmed3f(w_raw, -2.f, lv.Wp1[l]);
  const float hl_f = floorf(h_im), wl_f = floorf(w_im);
  const int h_low = (int)hl_f, w_low = (int)wl_f, h_hi = h_low + 1, w_hi = w_low + 1;
  const float lh = h_im - hl_f, lw = w_im - wl_f, hh = 1.f - lh, hw = 1.f - lw;
  const float e = __expf(lgs - mx);                                                   // <= 1: safe to evaluate always
  // cuh:298 + cuh:66-88 (see above): a row / column factor survives iff its pixel row / column exists
  const float hh_m = (unsigned)h_low < (unsigned)H ? hh : 0.f, lh_m = (unsigned)h_hi < (unsigned)H ? lh : 0.f;
  const float hw_m = (unsigned)w_low < (unsigned)W ? hw : 0.f, lw_m = (unsigned)w_hi < (unsigned)W ? lw : 0.f;
  wt = pack_bf16x2(hh_m * hw_m * e, hh_m * lw_m * e);
  wb = pack_bf16x2(lh_m * hw_m * e, lh_m * lw_m * e);
  // clamped pixel indices: a clamped corner always carries weight 0
  const int hl_c = clamp0_i32(h_low, H - 1), hh_c = clamp0_i32(h_hi, H - 1);
  const int wl_c = clamp0_i32(w_low, W - 1), wr_c = clamp0_i32(w_hi, W - 1);
  const unsigned col = ((unsigned)wl_c << 6) + (unsigned)lv.start[l] * 64u;
  const unsigned pitch = (unsigned)W * 64u;
  ot = __umul24((unsigned)hl_c, pitch) + col;                                         // < 2^32: checked by the host
  ob = __umul24((unsigned)hh_c, pitch) + col;
  dx = (unsigned)(wr_c - wl_c) * 64u;
}

// The footprint of a pair's reference point on the G map of level l (projattn.py:134,148-153: grid_sample, bilinear, zeros
// padding, align_corners = False): the 4 corner weights and the byte offsets of the 4 corners' G rows (192 bf16 columns) of image
// row0 / S.  l must be wave-uniform.
struct GFoot {
  float w00, w10, w01, w11;
  unsigned o00, o10, o01, o11;
};
template <int L>
__device__ __forceinline__ GFoot g_footprint(const float2 (&rr)[L], int l, const LevelTable& lv, int row0) {
  const int H = lv.H[l], W = lv.W[l];
  float refx = rr[0].x, refy = rr[0].y;
#pragma unroll
  for (int ll = 1; ll < L; ++ll) {
    refx = l == ll ? rr[ll].x : refx;
    refy = l == ll ? rr[ll].y : refy;
  }
  const float gx = fminf(fmaxf(refx * 2.f - 1.f, -1.1f), 1.1f);          // projattn.py:134
  const float gy = fminf(fmaxf(refy * 2.f - 1.f, -1.1f), 1.1f);
  const float ix = ((gx + 1.f) * lv.Wf[l] - 1.f) * 0.5f, iy = ((gy + 1.f) * lv.Hf[l] - 1.f) * 0.5f;
  const float x0f = floorf(ix), y0f = floorf(iy);
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const float tx = ix - x0f, ty = iy - y0f;
  // zeros padding on the 1-D factors (a product with a masked factor is the 0 the reference adds)
  const float ax0 = (unsigned)x0 < (unsigned)W ? 1.f - tx : 0.f, ax1 = (unsigned)x1 < (unsigned)W ? tx : 0.f;
  const float ay0 = (unsigned)y0 < (unsigned)H ? 1.f - ty : 0.f, ay1 = (unsigned)y1 < (unsigned)H ? ty : 0.f;
  GFoot f;
  f.w00 = ax0 * ay0;
  f.w10 = ax1 * ay0;
  f.w01 = ax0 * ay1;
  f.w11 = ax1 * ay1;
  const int x0c = clamp0_i32(x0, W - 1), x1c = clamp0_i32(x1, W - 1);
  const int y0c = clamp0_i32(y0, H - 1), y1c = clamp0_i32(y1, H - 1);
  const unsigned base = (unsigned)(row0 + lv.start[l]);
  const unsigned r0 = __umul24((unsigned)y0c, (unsigned)W) + base, r1 = __umul24((unsigned)y1c, (unsigned)W) + base;
  f.o00 = __umul24(r0 + (unsigned)x0c, 384u);                            // pixel index < 2^24: the host checks N_img*S*384 < 2^32
  f.o10 = __umul24(r0 + (unsigned)x1c, 384u);
  f.o01 = __umul24(r1 + (unsigned)x0c, 384u);
  f.o11 = __umul24(r1 + (unsigned)x1c, 384u);
  return f;
}

#ifdef GSAMP_EMUL_LDS_WINDOW
__shared__ __attribute__((aligned(16))) char gsamp_emul_window[32768];
#endif

// One (image-query pair, head) of the G-sampling kernel, computed by the 4 lanes of a quad (lane `sub` owns channels
// [8 sub, 8 sub + 8) of the head): phase A gathers the head's L*P logits + 2*L*P offsets = bilinear(G) + xw into the
// quad-private LDS row `sc` (3*L*P + 8 floats), pass 1 takes the softmax denominator, pass 2 samples the head plane.
// No workgroup barrier inside: the quads of a wavefront are independent, inactive quads may skip the call.
template <int L, int PIPE = 0>
__device__ __forceinline__ void gsamp_unit(const bf16_t* __restrict__ vp, const bf16_t* __restrict__ G,
                                           const float* __restrict__ xw, const float* __restrict__ r,
                                           const LevelTable& lv, float* __restrict__ sc, int pair, int m, int sub,
                                           int Lq, int S, int B, float (&acc)[8], float2 mine) {
  constexpr int P = 8, LP = L * P, NCHK = 3 * L, NB = 4;
  static_assert(L <= 4, "lane `sub` of the quad carries the reference point of level `sub`");
  const int n = pair / Lq, q = pair - n * Lq, b = n % B;

  // the pair's L reference points: lane l < L of the quad loaded level l's (`mine`, requested by the caller together with the
  // pair's mask byte), the other lanes get them by quad broadcast -- a load instruction costs the address path one clock per 4
  // ACTIVE lanes, and the quad's lanes would all fetch the same L x 8 bytes (profiles/r06_experiments.txt section 5)
  float2 rr[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    rr[l].x = __uint_as_float(quad_bcast_rt(__float_as_uint(mine.x), l));
    rr[l].y = __uint_as_float(quad_bcast_rt(__float_as_uint(mine.y), l));
  }
  if (sub < L) *reinterpret_cast<float2*>(sc + 3 * LP + 2 * sub) = mine;
  // ---- phase A: this head's L*P logits and 2*L*P offsets = bilinear(G) + xw, 8 columns per chunk.
  // Round 3: ALL its loads -- 4 G corners + 2 xw vectors for each of the lane's (NCHK + 3) / 4 chunks -- are requested before the
  // first one is used (lanes without a last chunk fetch chunk NCHK - 1 again and do not store it).  As a loop of "if (chunk <
  // NCHK) { load, blend, store }" every iteration was its own exec region and round trip: s_memtime showed phase A at 13 400 of
  // a wavefront's 35 000 cycles, as long as the six gather batches together.  Same arithmetic per chunk: bit-identical.
  // Round 6: the footprint of the pair on a level's G map (4 weights, 4 row offsets) is computed once per LEVEL THE HEAD TOUCHES
  // instead of once per chunk: head m owns the flat groups m*L .. m*L + L - 1 of the reinterpreted (level, column) view, i.e. at
  // most two level rows, and m is uniform in the workgroup -- so are the level indices (scalar loads of H / W / start instead of
  // vector loads from the kernel arguments in front of the first gather) and the branch for the second level (taken by heads 2 and
  // 5 of 8 at L = 3).  Same arithmetic per corner: bit-identical.
  constexpr int NK = (NCHK + 3) / 4;
  uint4 c00[NK], c10[NK], c01[NK], c11[NK];
  f32x4 xa[NK], xb[NK];
  float w00[NK], w10[NK], w01[NK], w11[NK];
  const int fg0 = m * L, l_lo = fg0 >> 3, l_hi = (fg0 + L - 1) >> 3;
  const GFoot fa = g_footprint<L>(rr, l_lo, lv, n * S);
  GFoot fb = fa;
  if (l_hi != l_lo) fb = g_footprint<L>(rr, l_hi, lv, n * S);
  const char* g_bytes = reinterpret_cast<const char*>(G);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int ci = min(sub + 4 * k, NCHK - 1);       // (lanes past the last chunk: a valid address they do not use)
    // G / xw columns are grouped per (16 offsets | 8 logits): group g of a level row = columns [24g, 24g+24) =
    // offsets 16g..16g+15 then logits 8g..8g+7 of that row, so the 3 chunks of a group -- and the L groups of a
    // head, flat groups m*L .. m*L+L-1 -- are contiguous bytes of a pixel's G row (ops.gsamp_column_order)
    const int t = ci / 3, part = ci - 3 * t;
    const int fg = fg0 + t;
    const bool second = (fg >> 3) != l_lo;                               // level row of the reinterpreted view
    const int col = 24 * (fg & 7) + 8 * part;
    w00[k] = second ? fb.w00 : fa.w00;
    w10[k] = second ? fb.w10 : fa.w10;
    w01[k] = second ? fb.w01 : fa.w01;
    w11[k] = second ? fb.w11 : fa.w11;
    // uniform base + 32-bit byte offsets (the host checks that G is smaller than 4 GB)
    const unsigned cb = (unsigned)col * 2u;
    const float* xq = xw + ((long)b * Lq + q) * 192 + col;
    // the last round has chunks for NCHK % 4 lanes of the quad only: the others do not load (their blend below runs on whatever the
    // registers hold and is not stored) -- 16 instead of 64 lane addresses for each of the round's 6 loads at L = 3
    if (4 * k + 4 <= NCHK || sub + 4 * k < NCHK) {
      c00[k] = *reinterpret_cast<const uint4*>(g_bytes + ((second ? fb.o00 : fa.o00) + cb));
      c10[k] = *reinterpret_cast<const uint4*>(g_bytes + ((second ? fb.o10 : fa.o10) + cb));
      c01[k] = *reinterpret_cast<const uint4*>(g_bytes + ((second ? fb.o01 : fa.o01) + cb));
      c11[k] = *reinterpret_cast<const uint4*>(g_bytes + ((second ? fb.o11 : fa.o11) + cb));
      xa[k] = *reinterpret_cast<const f32x4*>(xq);
      xb[k] = *reinterpret_cast<const f32x4*>(xq + 4);
    } else {
      c00[k] = c10[k] = c01[k] = c11[k] = uint4{0u, 0u, 0u, 0u};
      xa[k] = xb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int ci = sub + 4 * k;
    const int cc = min(ci, NCHK - 1);
    const int t = cc / 3, part = cc - 3 * t;
    const bool is_logit = part == 2;
    const unsigned a4[4] = {c00[k].x, c00[k].y, c00[k].z, c00[k].w}, b4[4] = {c10[k].x, c10[k].y, c10[k].z, c10[k].w};
    const unsigned c4[4] = {c01[k].x, c01[k].y, c01[k].z, c01[k].w}, d4[4] = {c11[k].x, c11[k].y, c11[k].z, c11[k].w};
    float v[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // explicit fma chain (the product of corner 10 is the plain multiply, as hipcc contracted it through round 5): "a*b + c*d + ..." leaves the choice of which product stays a plain multiply to
      // hipcc's contraction, and that choice moved with unrelated edits (17 of 614 400 output units off by one bf16 ulp)
      v[2 * u] = fmaf(w11[k], __uint_as_float(d4[u] << 16), fmaf(w01[k], __uint_as_float(c4[u] << 16),
                      fmaf(w00[k], __uint_as_float(a4[u] << 16), w10[k] * __uint_as_float(b4[u] << 16))));
      v[2 * u + 1] = fmaf(w11[k], __uint_as_float(d4[u] & 0xffff0000u), fmaf(w01[k], __uint_as_float(c4[u] & 0xffff0000u),
                          fmaf(w00[k], __uint_as_float(a4[u] & 0xffff0000u), w10[k] * __uint_as_float(b4[u] & 0xffff0000u))));
    }
    if (ci < NCHK) {
      float* dst = sc + (is_logit ? 8 * t : LP + 16 * t + 8 * part);
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0] + xa[k][0], v[1] + xa[k][1], v[2] + xa[k][2], v[3] + xa[k][3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4] + xb[k][0], v[5] + xb[k][1], v[6] + xb[k][2], v[7] + xb[k][3]};
    }
  }
  // quad-private scratch: LDS operations of one wavefront execute in order, only the compiler must not reorder
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- pass 1: softmax denominator of the head's logits; the 4 lanes of the quad split the LP logits and combine
  //      with DPP (same value in all four: the butterfly adds commute)
  float mx = -INFINITY;
  {
    constexpr int MYC = (LP / 4 + 3) / 4;            // 16-byte chunks per lane
    f32x4 lg[MYC];
#pragma unroll
    for (int i = 0; i < MYC; ++i) {
      const int c = sub + 4 * i;
      lg[i] = (c < LP / 4) ? *reinterpret_cast<const f32x4*>(sc + 4 * c) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      mx = fmaxf(fmaxf(fmaxf(mx, lg[i][0]), fmaxf(lg[i][1], lg[i][2])), lg[i][3]);
    }
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0xB1, 0xf, 0xf, false)));
    mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mx), 0x4E, 0xf, 0xf, false)));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MYC; ++i)     // exp(-inf) = 0 for the padding chunks
      sum += __expf(lg[i][0] - mx) + __expf(lg[i][1] - mx) + __expf(lg[i][2] - mx) + __expf(lg[i][3] - mx);
    sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0xB1, 0xf, 0xf, false));
    sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0x4E, 0xf, 0xf, false));
    mx += __logf(sum);
  }

#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;

  if constexpr (PIPE == 0) {
    // byte offset of this lane's 32-byte column slice inside vp (uniform base + 32-bit offsets: < 4 GB)
    const unsigned lane_off = (unsigned)((((long)n * 8 + m) * S) * 64 + sub * 16);
    const char* vp_bytes = reinterpret_cast<const char*>(vp);
    // Explicit software pipeline over the LP / NB batches (a real loop: unrolled, hipcc computes all 24 samples
    // first and spills):   gathers(it) issued  ->  coordinates(it + 1) computed under their latency  ->  blend(it)
    unsigned cw_t, cw_b, co_t, co_b, co_x;            // this lane's sample of the batch: packed weights / pixel offsets
    gsamp_coords<L>(0, sc, mx, lv, sub, cw_t, cw_b, co_t, co_b, co_x);
#pragma unroll 1
    for (int it = 0; it < LP / NB; ++it) {
      // ---- quad broadcast of the pixel offsets + 16 gathers in flight: top-left, top-right, bottom-left, bottom-right
      uint4 raw[NB][4];
#define MVG_QS(SS)                                                                                      \
      {                                                                                                 \
        const unsigned ot = quad_bcast<SS>(co_t) + lane_off, ob = quad_bcast<SS>(co_b) + lane_off;      \
        const unsigned dxs = quad_bcast<SS>(co_x);                                                      \
        raw[SS][0] = *reinterpret_cast<const uint4*>(vp_bytes + ot);                                    \
        raw[SS][1] = *reinterpret_cast<const uint4*>(vp_bytes + (ot + dxs));                            \
        raw[SS][2] = *reinterpret_cast<const uint4*>(vp_bytes + ob);                                    \
        raw[SS][3] = *reinterpret_cast<const uint4*>(vp_bytes + (ob + dxs));                            \
      }
      MVG_QS(0) MVG_QS(1) MVG_QS(2) MVG_QS(3)
#undef MVG_QS
      const unsigned pw_t = cw_t, pw_b = cw_b;        // this batch's weights, broadcast at blend time (fewer live VGPRs)
      __builtin_amdgcn_sched_barrier(0);
      // next batch's coordinates while the gathers are in flight (the last iteration recomputes batch 0: branch-free)
      gsamp_coords<L>(it + 1 < LP / NB ? it + 1 : 0, sc, mx, lv, sub, cw_t, cw_b, co_t, co_b, co_x);
      __builtin_amdgcn_sched_barrier(0);
#define MVG_BLEND0(SS)                                                                                  \
      _Pragma("unroll") for (int row = 0; row < 2; ++row) {                                             \
        /* left / right pixel (8 channels each) -> per channel the word (left[ch], right[ch]) = the v_dot2c operand for */ \
        /* the packed weights (w_left, w_right): 2 v_perm + 2 v_dot2c per pair of channels */           \
        const unsigned wv = row ? pw_b : pw_t;                                                          \
        const unsigned l4[4] = {raw[SS][2 * row].x, raw[SS][2 * row].y, raw[SS][2 * row].z, raw[SS][2 * row].w};             \
        const unsigned r4[4] = {raw[SS][2 * row + 1].x, raw[SS][2 * row + 1].y, raw[SS][2 * row + 1].z, raw[SS][2 * row + 1].w}; \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                 \
          dot2c_quad<SS>(acc[2 * t], wv, __builtin_amdgcn_perm(r4[t], l4[t], 0x05040100u));     /* (left[2t],   right[2t]) */   \
          dot2c_quad<SS>(acc[2 * t + 1], wv, __builtin_amdgcn_perm(r4[t], l4[t], 0x07060302u)); /* (left[2t+1], right[2t+1]) */ \
        }                                                                                               \
      }
      MVG_BLEND0(0) MVG_BLEND0(1) MVG_BLEND0(2) MVG_BLEND0(3)
#undef MVG_BLEND0
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // PIPE == 1 (round 3): the same arithmetic in the same order, the gathers double-buffered in half batches of 2 samples
    // (8 loads): while half A is blended, half B's 8 loads -- issued before -- are in flight, and the next half A is issued
    // before half B is blended.  A wavefront never sits in a blend with nothing outstanding (the single-buffered form
    // drained all 16 gathers, blended ~150 VALU instructions, then started the next round trip: 7 serialised round trips
    // per (pair, head)); vmcnt is in order, so "half A arrived" is s_waitcnt vmcnt(8).  Same 64 gather VGPRs.
    const unsigned lane_off = (unsigned)((((long)n * 8 + m) * S) * 64 + sub * 16);
    const char* vp_bytes = reinterpret_cast<const char*>(vp);
#ifdef GSAMP_ABLATE_LAST_LEVEL   // timing probe only (results garbage): the last level is not gathered
    constexpr int NIT = LP / NB - 2;
#else
    constexpr int NIT = LP / NB;
#endif
    unsigned cw_t, cw_b, co_t, co_b, co_x;
    gsamp_coords<L>(0, sc, mx, lv, sub, cw_t, cw_b, co_t, co_b, co_x);
    uint4 ra[2][4], rb[2][4];
#ifdef GSAMP_EMUL_LDS_WINDOW   // timing probe only (results garbage): the value gathers as ds_read_b128 from a 64-KB LDS window
#define MVG_ISSUE(BUF, J, SS)                                                                           \
    {                                                                                                   \
      const unsigned ot = quad_bcast<SS>(co_t) + lane_off, ob = quad_bcast<SS>(co_b) + lane_off;        \
      const unsigned dxs = quad_bcast<SS>(co_x);                                                        \
      BUF[J][0] = *reinterpret_cast<const uint4*>(gsamp_emul_window + (ot & 0x7ff0u));                  \
      BUF[J][1] = *reinterpret_cast<const uint4*>(gsamp_emul_window + ((ot + dxs) & 0x7ff0u));          \
      BUF[J][2] = *reinterpret_cast<const uint4*>(gsamp_emul_window + (ob & 0x7ff0u));                  \
      BUF[J][3] = *reinterpret_cast<const uint4*>(gsamp_emul_window + ((ob + dxs) & 0x7ff0u));          \
    }
#else
#define MVG_ISSUE(BUF, J, SS)                                                                           \
    {                                                                                                   \
      const unsigned ot = quad_bcast<SS>(co_t) + lane_off, ob = quad_bcast<SS>(co_b) + lane_off;        \
      const unsigned dxs = quad_bcast<SS>(co_x);                                                        \
      BUF[J][0] = *reinterpret_cast<const uint4*>(vp_bytes + ot);                                       \
      BUF[J][1] = *reinterpret_cast<const uint4*>(vp_bytes + (ot + dxs));                               \
      BUF[J][2] = *reinterpret_cast<const uint4*>(vp_bytes + ob);                                       \
      BUF[J][3] = *reinterpret_cast<const uint4*>(vp_bytes + (ob + dxs));                               \
    }
#endif
#define MVG_BLEND(BUF, J, SS)                                                                           \
    _Pragma("unroll") for (int row = 0; row < 2; ++row) {                                               \
      const unsigned wv = row ? pw_b : pw_t;                                                            \
      const unsigned l4[4] = {BUF[J][2 * row].x, BUF[J][2 * row].y, BUF[J][2 * row].z, BUF[J][2 * row].w};                 \
      const unsigned r4[4] = {BUF[J][2 * row + 1].x, BUF[J][2 * row + 1].y, BUF[J][2 * row + 1].z, BUF[J][2 * row + 1].w}; \
      _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                   \
        dot2c_quad<SS>(acc[2 * t], wv, __builtin_amdgcn_perm(r4[t], l4[t], 0x05040100u));               \
        dot2c_quad<SS>(acc[2 * t + 1], wv, __builtin_amdgcn_perm(r4[t], l4[t], 0x07060302u));           \
      }                                                                                                 \
    }
    MVG_ISSUE(ra, 0, 0) MVG_ISSUE(ra, 1, 1)
    unsigned pw_t, pw_b;
#pragma unroll 1
    for (int it = 0; it < NIT - 1; ++it) {
      MVG_ISSUE(rb, 0, 2) MVG_ISSUE(rb, 1, 3)
      pw_t = cw_t;
      pw_b = cw_b;
      __builtin_amdgcn_sched_barrier(0);
      gsamp_coords<L>(it + 1, sc, mx, lv, sub, cw_t, cw_b, co_t, co_b, co_x);
      __builtin_amdgcn_sched_barrier(0);
      MVG_BLEND(ra, 0, 0) MVG_BLEND(ra, 1, 1)
      __builtin_amdgcn_sched_barrier(0);
      MVG_ISSUE(ra, 0, 0) MVG_ISSUE(ra, 1, 1)          // first half of batch it + 1
      __builtin_amdgcn_sched_barrier(0);
      MVG_BLEND(rb, 0, 2) MVG_BLEND(rb, 1, 3)
      __builtin_amdgcn_sched_barrier(0);
    }
    MVG_ISSUE(rb, 0, 2) MVG_ISSUE(rb, 1, 3)            // last batch: nothing left to issue behind it
    pw_t = cw_t;
    pw_b = cw_b;
    __builtin_amdgcn_sched_barrier(0);
    MVG_BLEND(ra, 0, 0) MVG_BLEND(ra, 1, 1)
    __builtin_amdgcn_sched_barrier(0);
    MVG_BLEND(rb, 0, 2) MVG_BLEND(rb, 1, 3)
#undef MVG_ISSUE
#undef MVG_BLEND
  }
}
