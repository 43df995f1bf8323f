// Weight-stationary chain A, 4-wavefront form: ONE 256-thread workgroup per CU, one wavefront per SIMD with the whole 512-entry
// register file: wavefront w holds columns [64 w, 64 w + 64) of ALL THREE 256x256 weights (384 VGPRs/AGPRs) and runs, per step,
// stage 1 on tile t, stage 2 on tile t - 1, stage 3 on tile t - 2 (32-row tiles in LDS, double-buffered; one barrier per step).
// Measures clk per step (MFMA floor 96 x 32 = 3072).  (measurement only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "chain_dev.h"

// MFMA with the weight fragment read straight from an AGPR quad ("a") or a VGPR quad ("v"); accumulator in VGPRs.
__device__ __forceinline__ void mfma_wa(f32x16& acc, const f32x4& w, const f32x4& a) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(a));
}
__device__ __forceinline__ void mfma_wv(f32x16& acc, const f32x4& w, const f32x4& a) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ void mfma_wa0(f32x16& acc, const f32x4& w, const f32x4& a) {     // first k-step: C = 0
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(w), "v"(a));
}
__device__ __forceinline__ void mfma_wv0(f32x16& acc, const f32x4& w, const f32x4& a) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(w), "v"(a));
}

constexpr int TP = ACT_PITCH;
constexpr int TILE = 32 * TP;

__device__ __forceinline__ void epi(char* __restrict__ dst, const f32x16& acc, int col0, int rl, int h, bool relu) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float x = acc[4 * g + t] + 0.125f;
      v[t] = relu ? fmaxf(x, 0.f) : x;
    }
    uint2 pk;
    pk.x = pack_bf16(v[0], v[1]);
    pk.y = pack_bf16(v[2], v[3]);
    *reinterpret_cast<uint2*>(dst + rl * TP + (col0 + 8 * g + 4 * h) * 2) = pk;
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void systolic4_kernel(const bf16_t* __restrict__ samp, const bf16_t* __restrict__ W,
                                                           bf16_t* __restrict__ attn, int steps, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* X = smem;
  char* Y1 = smem + 2 * TILE;
  char* Y2 = smem + 4 * TILE;
  char* Y3 = smem + 6 * TILE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 31, h = lane >> 5;
  f32x4 w[3][16][2];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const bf16_t* wp = W + (long)s * 65536 + (long)wave * 16 * 1024 + lane * 8;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) w[s][i][j] = *reinterpret_cast<const f32x4*>(wp + i * 1024 + j * 512);
  }
  const long tile0 = (long)blockIdx.x * steps;
  {
    const int row = tid >> 3, v8 = tid & 7;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      *reinterpret_cast<f32x4*>(X + row * TP + (v8 + 8 * u) * 16) = *reinterpret_cast<const f32x4*>(samp + (tile0 * 32 + row) * 256 + (v8 + 8 * u) * 8);
  }
  long long t_start = 0;
  f32x16 acc[3][2];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[2][0][e] = acc[2][1][e] = 0.f;
  for (int t = 0; t < steps + 2; ++t) {
    __syncthreads();
    if (t == 2 && tid == 0) t_start = __builtin_readcyclecounter();
    f32x4 nx[4];
    const bool has_next = t + 1 < steps;
    const int row = tid >> 3, v8 = tid & 7;
    if (has_next) {
#pragma unroll
      for (int u = 0; u < 4; ++u) nx[u] = *reinterpret_cast<const f32x4*>(samp + ((tile0 + t + 1) * 32 + row) * 256 + (v8 + 8 * u) * 8);
    }
    // Rolling schedule: while GEMM s issues its MFMAs, the epilogue of the PREVIOUS GEMM (a different accumulator) runs in their
    // shadow, one 4-value group per other k-step: phase A = stage 1 (tile t) || epilogue of stage 3 (tile t - 3, from the last
    // step); phase B = stage 2 (tile t - 1) || epilogue of stage 1; phase C = stage 3 (tile t - 2) || epilogue of stage 2.
    const char* s1 = X + (t & 1) * TILE + rl * TP + 16 * h;
    const char* s2 = Y1 + ((t - 1) & 1) * TILE + rl * TP + 16 * h;
    const char* s3 = Y2 + ((t - 2) & 1) * TILE + rl * TP + 16 * h;
    char* d1 = Y1 + (t & 1) * TILE;
    char* d2 = Y2 + ((t - 1) & 1) * TILE;
    char* d3 = Y3 + (t & 1) * TILE;
#define PHASE(S, SRC, PREV, DST, RELU)                                                                                        \
  {                                                                                                                           \
    f32x4 ar[2];                                                                                                              \
    ar[0] = *reinterpret_cast<const f32x4*>(SRC);                                                                             \
    ar[1] = *reinterpret_cast<const f32x4*>(SRC + 32);                                                                        \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                                          \
      const f32x4 ac = ar[i & 1];                                                                                             \
      if (i + 2 < 16) ar[i & 1] = *reinterpret_cast<const f32x4*>(SRC + (i + 2) * 32);                                        \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                         \
        /* 256 AGPRs hold stage 1 and all but the last k-step of stage 2 (a full 256 left hipcc two fragments short) */     \
        if (S == 2 || (S == 1 && i == 15)) { if (i == 0) mfma_wv0(acc[S][j], w[S][i][j], ac); else mfma_wv(acc[S][j], w[S][i][j], ac); } \
        else { if (i == 0) mfma_wa0(acc[S][j], w[S][i][j], ac); else mfma_wa(acc[S][j], w[S][i][j], ac); }                    \
      }                                                                                                                       \
      if (MODE != 1 && (i & 1) == 0) {                                                                                        \
        if (i == 0) asm volatile("s_nop 15");      /* MFMA results of the previous phase -> VALU reads below */                \
        const int j = i >> 3, g = (i & 7) >> 1;                                                                               \
        float v[4];                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                       \
          const float x = acc[PREV][j][4 * g + q] + 0.125f;                                                                   \
          v[q] = RELU ? fmaxf(x, 0.f) : x;                                                                                    \
        }                                                                                                                     \
        uint2 pk;                                                                                                             \
        pk.x = pack_bf16(v[0], v[1]);                                                                                         \
        pk.y = pack_bf16(v[2], v[3]);                                                                                         \
        *reinterpret_cast<uint2*>(DST + rl * TP + (wave * 64 + j * 32 + 8 * g + 4 * h) * 2) = pk;                             \
      }                                                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                                      \
    }                                                                                                                         \
  }
    PHASE(0, s1, 2, d3, true)
    PHASE(1, s2, 0, d1, false)
    PHASE(2, s3, 1, d2, true)
#undef PHASE
    if (MODE == 1) {      // keep every accumulator alive (no MFMA may be eliminated)
      float keep = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
        for (int j = 0; j < 2; ++j) keep += acc[s_][j][0] + acc[s_][j][15];
      if (keep == 12345.f) Y3[tid] = 1;
    }
    if (t >= 1 && t - 1 < steps) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<f32x4*>(attn + ((tile0 + t - 1) * 32 + row) * 256 + (v8 + 8 * u) * 8) =
            *reinterpret_cast<const f32x4*>(Y1 + ((t - 1) & 1) * TILE + row * TP + (v8 + 8 * u) * 16);
    }
    if (has_next) {
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(X + ((t + 1) & 1) * TILE + row * TP + (v8 + 8 * u) * 16) = nx[u];
    }
  }
  if (tid == 0 && stamps) stamps[blockIdx.x * 2] = __builtin_readcyclecounter() - t_start;
  if (Y3[tid] == 77 && tid == 999) attn[0] = 1;
}

template <int MODE>
void run(const char* name, const bf16_t* samp, const bf16_t* W, bf16_t* attn, long long* stamps, int grid, int steps) {
  const int smem = 8 * TILE;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&systolic4_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  systolic4_kernel<MODE><<<grid, 256, smem>>>(samp, W, attn, steps, stamps);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) systolic4_kernel<MODE><<<grid, 256, smem>>>(samp, W, attn, steps, stamps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(grid * 2);
  hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < grid; ++i) s += double(h[2 * i]);
  printf("%-34s steps %3d: %7.1f us per launch (%d rows); %7.0f clk per step in the steady state\n", name, steps, ms * 1e2, grid * steps * 32,
         s / grid / steps);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  bf16_t *samp, *W, *attn;
  long long* stamps;
  const long rows = (long)cus * 64 * 32;
  hipMalloc(&samp, rows * 512);
  hipMemset(samp, 0, rows * 512);
  hipMalloc(&attn, rows * 512);
  hipMalloc(&W, 3 * 131072);
  hipMemset(W, 0, 3 * 131072);
  hipMalloc(&stamps, cus * 2 * sizeof(long long));
  for (int steps : {10, 20, 40}) {
    run<0>("systolic chain (4 waves), full", samp, W, attn, stamps, cus, steps);
    run<1>("  without epilogues", samp, W, attn, stamps, cus, steps);
  }
  return 0;
}
