// Fused sampler + chain A for gfx950: msda_gsamp_kernel (csrc/msda.hip) and chain_a_kernel (csrc/chain.hip) as ONE
// kernel, with the sampled tile never leaving the CU.
//
// Why: as two launches the sampler (bound by the L1 / latency of its gathers, matrix pipe idle: PMC MFMA 0) and chain A
// (MFMA, bound by the L2 -> L1 weight stream) run one after the other with a 39-MB `samp` round trip in between; the
// sampler alone is 43 % of the forward.  Running them concurrently as two kernels on two streams does not work on this
// chip -- the sampler's workgroups hold every SIMD's wave slots and the chain starves (profiles/r02_query_groups_rejected.txt)
// -- so the overlap has to happen inside one kernel with one register budget:
//
//   * a workgroup = 64 consecutive slots of the processing order (mvg_bin_pairs: image-space order, out-of-image pairs
//     last) x ALL 8 heads, 8 wavefronts, 2 workgroups per CU (78 KB LDS each, <= 128 VGPRs: the sampler's 4 waves / SIMD);
//   * gather phase: the 32 units (16 pairs x 1 head) of the tile are spread over the 8 wavefronts (4 each, gsamp_unit of
//     gsamp_dev.h -- exactly the arithmetic of msda_gsamp_kernel) and land as bf16 rows in the LDS tile;
//   * chain phase: chain_a_body<64, 512, 1> of chain_dev.h on that tile (output projection x in-image mask -> attn, pose
//     MLP -> o), 8 column-split wavefronts, every weight fragment loaded by exactly one of them.
//   The two workgroups of a CU drift apart, so one's MFMA stages run under the other's gather latency, and the weight
//   stream of a stage hides behind gathers instead of being waited for.
//   * tiles without a single in-image row (the tail of every image's order) only write attn = 0 and o = o_masked.
// Every (pair, head) and every row is computed exactly as in the two-kernel form and independently of where it sits in
// the launch: outputs are bit-identical to msda_gsamp + chain_attn_pose<64, 512, 1>, run to run, and for any order.
#include "chain_dev.h"
#include "gsamp_dev.h"

int g_sampchain_mode = 0;      // diagnostic knob "sampchain_mode": 0 = full kernel, 1 = gather phase only, 2 = chain phase only

namespace {

constexpr int SC_RM = 64, SC_NT = 512;

template <int L>
struct SampChainSmem {
  static constexpr int SCP = 3 * L * 8 + 8;                         // floats per quad-private scratch row (gsamp_unit)
  static constexpr int ACT = 0;
  static constexpr int RID = SC_RM * ACT_PITCH;
  static constexpr int W2S = RID + SC_RM * (int)sizeof(int);
  static constexpr int SCR = W2S + 768 * (int)sizeof(float);
  static constexpr int BYTES = SCR + (SC_NT / 64) * 16 * SCP * (int)sizeof(float);
};

template <int L>
__global__ __launch_bounds__(SC_NT, 2) void samp_chain_kernel(
    const bf16_t* __restrict__ vp, const bf16_t* __restrict__ G, const float* __restrict__ xw, const float* __restrict__ r,
    LevelTable lv, const uint8_t* __restrict__ inside, const int* __restrict__ order, const bf16_t* __restrict__ Wp,
    const float* __restrict__ bp, const bf16_t* __restrict__ W0, const float* __restrict__ b0,
    const bf16_t* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, bf16_t* __restrict__ attn, float* __restrict__ o, const float* __restrict__ o_masked,
    int R, int Lq, int S, int B, int ntiles, int map_ch, int mode) {
  typedef SampChainSmem<L> SM;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem + SM::ACT;
  int* rid = reinterpret_cast<int*>(smem + SM::RID);
  float* w2s = reinterpret_cast<float*>(smem + SM::W2S);
  float* scratch = reinterpret_cast<float*>(smem + SM::SCR);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, sub = lane & 3, pl = lane >> 2;
  // XCD = blockIdx & 7 works on chunks of map_ch consecutive tiles: its L2 serves a compact region of the head planes / G
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tile = ((j / map_ch) * 8 + xcd) * map_ch + j % map_ch;
  if (tile >= ntiles) return;
  const int r0 = tile * SC_RM;

  for (int i = tid; i < 768; i += SC_NT) w2s[i] = W2[i];
  bool mine = false;
  if (tid < SC_RM) {
    const int slot = r0 + tid;
    const int g = slot < R ? (order ? order[slot] : slot) : -1;
    rid[tid] = g;
    mine = g >= 0 && inside[g] != 0;
  }
  const bool any_inside = __syncthreads_or(mine) != 0;
  if (!any_inside && o_masked) {
    const float m0 = o_masked[0], m1 = o_masked[1], m2 = o_masked[2];
#pragma unroll
    for (int c0 = 0; c0 < SC_RM * 32; c0 += SC_NT) {
      const int c = c0 + tid, row = c >> 5, v16 = c & 31;
      const int g = rid[row];
      if (g >= 0) *reinterpret_cast<f32x4*>(attn + (long)g * 256 + v16 * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < SC_RM && rid[tid] >= 0) {
      float* og = o + (long)rid[tid] * 3;
      og[0] = m0;
      og[1] = m1;
      og[2] = m2;
    }
    return;
  }

  // ---- gather phase: unit u = (pair group u & 3: tile rows 16 (u & 3) .. +15, head u >> 2); wavefront w takes units
  //      w, w + 8, w + 16, w + 24.  A quad (4 lanes) = one (pair, head); rows whose pair is outside the image (their attn
  //      is multiplied by 0, dq_decoder.py:585-586) and rows past the end are zero-filled without sampling.
  float* sc = scratch + ((wave * 16 + pl) * SM::SCP);
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const int u = wave + 8 * k, row = (u & 3) * 16 + pl, m = u >> 2;
    const int g = rid[row];
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    if (g >= 0 && inside[g] != 0 && mode != 2) gsamp_unit<L>(vp, G, xw, r, lv, sc, g, m, sub, Lq, S, B, acc);
    uint4 pk;
    pk.x = pack_bf16(acc[0], acc[1]);
    pk.y = pack_bf16(acc[2], acc[3]);
    pk.z = pack_bf16(acc[4], acc[5]);
    pk.w = pack_bf16(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(act + row * ACT_PITCH + (m * 32 + sub * 8) * 2) = pk;
  }
  if (mode == 1) {     // diagnostic: gather phase only (the tile goes out as attn)
    __syncthreads();
#pragma unroll
    for (int c0 = 0; c0 < SC_RM * 32; c0 += SC_NT) {
      const int c = c0 + tid, row = c >> 5, v16 = c & 31;
      if (rid[row] >= 0)
        *reinterpret_cast<f32x4*>(attn + (long)rid[row] * 256 + v16 * 8) = *reinterpret_cast<const f32x4*>(act + row * ACT_PITCH + v16 * 16);
    }
    return;
  }
  // ---- chain phase (starts with the barrier that makes the tile visible)
  chain_a_body<SC_RM, SC_NT, 1>(act, rid, w2s, inside, Wp, bp, W0, b0, W1, b1, b2, attn, o);
}

template <int L>
int launch_samp_chain(const void* vp, const void* G, const float* xw, const float* r, const LevelTable& lv,
                      const uint8_t* inside, const int* order, const void* Wp, const float* bp, const void* W0,
                      const float* b0, const void* W1, const float* b1, const float* W2, const float* b2, void* attn,
                      float* o, const float* o_masked, int rows, int Lq, int S, int B, int map_ch, hipStream_t st) {
  typedef SampChainSmem<L> SM;
  static bool configured[MVG_MAX_DEVICES] = {};     // the attribute is per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVG_MAX_DEVICES) return MVG_E_BADARG;
  if (!configured[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&samp_chain_kernel<L>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SM::BYTES);
    if (e != hipSuccess) return (int)e;
    configured[dev] = true;
  }
  const int ntiles = (rows + SC_RM - 1) / SC_RM;
  const int grid = (ntiles + 8 * map_ch - 1) / (8 * map_ch) * (8 * map_ch);
  hipLaunchKernelGGL((samp_chain_kernel<L>), dim3(grid), dim3(SC_NT), SM::BYTES, st, (const bf16_t*)vp, (const bf16_t*)G,
                     xw, r, lv, inside, order, (const bf16_t*)Wp, bp, (const bf16_t*)W0, b0, (const bf16_t*)W1, b1, W2, b2,
                     (bf16_t*)attn, o, o_masked, rows, Lq, S, B, ntiles, map_ch, g_sampchain_mode);
  MVG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int g_sampchain_map = 4;       // tuning knob "sampchain_map": consecutive 64-row tiles per XCD chunk

extern "C" int mvg_msda_gsamp_chain(const void* vh, const void* G, const float* xw, const float* ref_lvl,
                                    const int64_t* shapes_host, const int64_t* starts_host, const uint8_t* inside,
                                    const int32_t* order, const void* Wp, const float* bp, const void* W0, const float* b0,
                                    const void* W1, const float* b1, const float* W2, const float* b2, void* attn, float* o,
                                    const float* o_masked, int N_img, int Lq, int L, int S, int B, void* stream) {
  if (!vh || !G || !xw || !ref_lvl || !shapes_host || !starts_host || !inside || !Wp || !bp || !W0 || !b0 || !W1 || !b1 ||
      !W2 || !b2 || !attn || !o || B <= 0)
    return MVG_E_BADARG;
  LevelTable lv;
  int e = mvg_fill_levels(&lv, shapes_host, starts_host, L);
  if (e) return e;
  const long pairs = (long)N_img * Lq;
  if (pairs > 0x7fffffffL / 4) return MVG_E_BADARG;
  if ((long)N_img * S * 384 >= 0xffffffffL || (long)N_img * 8 * S * 64 >= 0xffffffffL) return MVG_E_BADARG;   // 32-bit byte offsets
  if (pairs == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int map = g_sampchain_map > 0 ? g_sampchain_map : 1;
  switch (L) {
    case 1: return launch_samp_chain<1>(vh, G, xw, ref_lvl, lv, inside, order, Wp, bp, W0, b0, W1, b1, W2, b2, attn, o, o_masked, (int)pairs, Lq, S, B, map, st);
    case 2: return launch_samp_chain<2>(vh, G, xw, ref_lvl, lv, inside, order, Wp, bp, W0, b0, W1, b1, W2, b2, attn, o, o_masked, (int)pairs, Lq, S, B, map, st);
    case 3: return launch_samp_chain<3>(vh, G, xw, ref_lvl, lv, inside, order, Wp, bp, W0, b0, W1, b1, W2, b2, attn, o, o_masked, (int)pairs, Lq, S, B, map, st);
    case 4: return launch_samp_chain<4>(vh, G, xw, ref_lvl, lv, inside, order, Wp, bp, W0, b0, W1, b1, W2, b2, attn, o, o_masked, (int)pairs, Lq, S, B, map, st);
    default: return MVG_E_BADARG;
  }
}
