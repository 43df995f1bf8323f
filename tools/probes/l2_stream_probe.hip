// How fast can all 256 CUs pull the SAME L2-resident weights, the way the fused linear chains do?  (measurement only)
//
// One 512-thread workgroup per CU (8 wavefronts, like chain_b_kernel); a "stage" = 128 KB of weights = 8 column groups x 16
// k-steps x 1-KB fragments (one global_load_dwordx4 per lane); wavefront w owns column group w and walks its 16 fragments
// through a register ring of DEPTH loads in flight.  No MFMA, no LDS: this is the weight stream alone.  Modes differ in where
// fragment (stage, colgroup, kstep) lives and in which order / phase a wavefront and a workgroup walk them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Mode {
  int layout;     // 0: [stage][colgroup][kstep]   1: [stage][kstep][colgroup]   2: [stage][kstep ^ swz][colgroup] 4-KB blocks interleaved
  int wave_rot;   // k-step rotation per wavefront: (wave * wave_rot) & 15
  int wg_phases;  // workgroups of an XCD start (blockIdx >> 3) % wg_phases * (16 / wg_phases) k-steps apart
  int wg_stage;   // workgroups start (blockIdx >> 3) % wg_stage stages apart (not possible in a real chain: upper bound only)
};

template <int DEPTH, int LAYOUT>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ w, int stage_mask, int reps, Mode m, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x >> 3;
  const int rot = (wave * m.wave_rot + (m.wg_phases > 1 ? (j & (m.wg_phases - 1)) * (16 / m.wg_phases) : 0)) & 15;
  const int s0 = m.wg_stage > 1 ? (j & (m.wg_stage - 1)) : 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 ring[DEPTH];
  const int total = (stage_mask + 1) * reps * 16;
  const char* base = w + lane * 16 + (LAYOUT == 0 ? wave * 16384 : wave * 1024);
  auto addr = [&](int i) -> const f32x4* {
    const int st = ((i >> 4) + s0) & stage_mask, k = (i + rot) & 15;
    return reinterpret_cast<const f32x4*>(base + st * 131072 + k * (LAYOUT == 0 ? 1024 : 8192));
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) ring[d] = *addr(d);
  for (int i = 0; i < total; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc += ring[d];                                    // waits for the oldest load only (in-order vmcnt)
      ring[d] = *addr((i + d + DEPTH) & (total - 1));    // (wraps around at the end: total is a power of two)
      __builtin_amdgcn_sched_barrier(0);                 // keep the ring a ring: hipcc otherwise drains it and refills in batches
    }
  }
  if (acc.x + acc.y == 1234.5f) sink[blockIdx.x * 512 + threadIdx.x] = acc.z;
}

template <int DEPTH>
float run(const char* w, int stages, int reps, Mode m, float* sink, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto launch = [&](int r) {
    if (m.layout == 0) stream_kernel<DEPTH, 0><<<grid, 512>>>(w, stages - 1, r, m, sink);
    else stream_kernel<DEPTH, 1><<<grid, 512>>>(w, stages - 1, r, m, sink);
  };
  launch(2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch(reps);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / (stages * reps);      // us per 128-KB stage
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  char* w = nullptr;
  hipMalloc(&w, 64 << 20);
  hipMemset(w, 0, 64 << 20);
  float* sink = nullptr;
  hipMalloc(&sink, 16 << 20);
  printf("# %d CUs; us per 128-KB stage pulled by EVERY workgroup (1 per CU, 8 wavefronts); GB/s per CU\n", cus);
  printf("# layout wave_rot wg_phases wg_stage | depth 2      4      8      16\n");
  const Mode modes[] = {{0, 0, 1, 1}, {0, 3, 1, 1}, {0, 3, 2, 1}, {0, 3, 4, 1}, {0, 3, 16, 1}, {0, 1, 16, 1}, {0, 5, 16, 1},
                        {1, 0, 1, 1}, {1, 0, 2, 1}, {1, 0, 16, 1}, {1, 2, 1, 1}, {1, 2, 16, 1}, {1, 1, 16, 1},
                        {0, 3, 1, 8}, {1, 0, 1, 8}, {0, 3, 16, 8}};
  for (int stages : {8, 64}) {
    printf("# stages = %d (%d KB of weights)\n", stages, stages * 128);
    for (const Mode& m : modes) {
      const int reps = 4096 / stages;
      float t2 = run<2>(w, stages, reps, m, sink, cus), t4 = run<4>(w, stages, reps, m, sink, cus);
      float t8 = run<8>(w, stages, reps, m, sink, cus), t16 = run<16>(w, stages, reps, m, sink, cus);
      printf("  %d      %2d      %2d       %2d     | %5.2f (%3.0f)  %5.2f (%3.0f)  %5.2f (%3.0f)  %5.2f (%3.0f)\n", m.layout, m.wave_rot,
             m.wg_phases, m.wg_stage, t2, 131.072 / t2, t4, 131.072 / t4, t8, 131.072 / t8, t16, 131.072 / t16);
    }
  }
  // the same with 2 workgroups per CU
  printf("# 2 workgroups per CU (grid = 2 x CUs), stages = 8\n");
  for (const Mode& m : modes) {
    float t4 = run<4>(w, 8, 512, m, sink, 2 * cus), t8 = run<8>(w, 8, 512, m, sink, 2 * cus);
    printf("  %d      %2d      %2d       %2d     | d4 %5.2f  d8 %5.2f   (per CU: %3.0f / %3.0f GB/s)\n", m.layout, m.wave_rot, m.wg_phases,
           m.wg_stage, t4, t8, 2 * 131.072 / t4, 2 * 131.072 / t8);
  }
  return 0;
}
