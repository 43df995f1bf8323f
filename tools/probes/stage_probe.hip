// Where do the 2.4 us of a 256 x 256 stage GEMM of the fused chains go?  (measurement only; uses the library's own
// chain_dev.h building blocks: build with -I mvgformer_amd/csrc and the library's flags)
//
// One 512-thread workgroup per CU walks a 64-row LDS tile through STAGES Linear(256 -> 256) + ReLU stages exactly like
// chain_b_kernel's FFN stages (8 column-split wavefronts, JN = 1, MT = 2, ring of 4).  Modes switch parts of a stage off:
//   0 full   1 no epilogue / barriers (k-loops back to back)   2 k-loop without the weight loads   3 k-loop without LDS reads
//   4 epilogue + barriers only
// and lane 0 of every wavefront of workgroup 0 / 100 records s_memtime at the phase boundaries of the last stages.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "chain_dev.h"

template <int MODE>
__global__ __launch_bounds__(512, 1) void stage_kernel(const bf16_t* __restrict__ W, const float* __restrict__ bias, int stages,
                                                       long long* __restrict__ stamps, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 64 * ACT_PITCH / 4; i += 512) reinterpret_cast<unsigned*>(act)[i] = 0x3c003c00u + (i & 0xff);
  __syncthreads();
  const int rot = (wave * 3 + ((blockIdx.x >> 3) & 1) * 8) & 15;
  f32x16 acc[2][1];
  f32x4 pf[4][1], bvr[1][4];
  bool all[2] = {true, true};
  ring_prefetch<16, 4, 1, 2>(W, pf, tid, rot);
  const bool rec = (blockIdx.x == 0 || blockIdx.x == 100) && lane == 0;
  long long* st = stamps + ((blockIdx.x == 0 ? 0 : 1) * 8 + wave) * 5 * 64;
  for (int s = 0; s < stages; ++s) {
    const bf16_t* Ws = W + (long)(s & 7) * 65536;
    const bf16_t* Wn = W + (long)((s + 1) & 7) * 65536;
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (rec) t0 = __builtin_readcyclecounter();
    if (MODE != 4) {
      if (MODE == 2) {
        // weights: always the prefetched fragments (no global loads inside the loop)
        stage_gemm<2, 4, 4, 1, true>(act, Ws, acc, tid, true, 0, 16 * 1024, pf);
        stage_gemm<2, 4, 4, 1, true>(act + 128, Ws, acc, tid, false, 0, 16 * 1024, pf);
        stage_gemm<2, 4, 4, 1, true>(act + 256, Ws, acc, tid, false, 0, 16 * 1024, pf);
        stage_gemm<2, 4, 4, 1, true>(act + 384, Ws, acc, tid, false, 0, 16 * 1024, pf);
      } else {
        stage_gemm<2, 16, 4, 1, true>(act, Ws, acc, tid, true, rot, 16 * 1024, pf);
      }
    }
    if (rec) t1 = __builtin_readcyclecounter();
    load_bias<1>(bias, bvr, tid, 64);
    ring_prefetch<16, 4, 1, 2>(Wn, pf, tid, rot);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE != 1 && MODE != 2 && MODE != 3) __syncthreads();
    if (rec) t2 = __builtin_readcyclecounter();
    if (MODE != 1 && MODE != 2 && MODE != 3) write_act_pre<2, 1>(act, acc, bvr, true, all, tid);
    else if (acc[0][0][0] == 123.f) write_act_pre<2, 1>(act, acc, bvr, true, all, tid);
    if (rec) t3 = __builtin_readcyclecounter();
    if (MODE != 1 && MODE != 2 && MODE != 3) __syncthreads();
    if (rec) t4 = __builtin_readcyclecounter();
    if (rec && s >= stages - 64) {
      const int q = s - (stages - 64);
      st[0 * 64 + q] = t0; st[1 * 64 + q] = t1; st[2 * 64 + q] = t2; st[3 * 64 + q] = t3; st[4 * 64 + q] = t4;
    }
  }
  if (act[tid] == 77 && acc[1][0][3] == 5.f) sink[tid] = 1.f;
}

template <int MODE>
void run(const char* name, const bf16_t* W, const float* bias, long long* stamps, float* sink, int grid) {
  const int stages = 512, smem = 64 * ACT_PITCH;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  stage_kernel<MODE><<<grid, 512, smem>>>(W, bias, 16, stamps, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  stage_kernel<MODE><<<grid, 512, smem>>>(W, bias, stages, stamps, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(2 * 8 * 5 * 64);
  hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  printf("%-44s %6.3f us / stage", name, ms * 1e3 / stages);
  for (int blk = 0; blk < 2; ++blk) {
    double ph[4] = {0, 0, 0, 0};
    for (int w = 0; w < 8; ++w)
      for (int q = 0; q < 64; ++q)
        for (int p = 0; p < 4; ++p) ph[p] += double(h[((blk * 8 + w) * 5 + p + 1) * 64 + q] - h[((blk * 8 + w) * 5 + p) * 64 + q]);
    printf("  | wg %3d ticks: k-loop %6.1f  barrier %6.1f  epilogue %6.1f  barrier %6.1f", blk ? 100 : 0, ph[0] / 512, ph[1] / 512,
           ph[2] / 512, ph[3] / 512);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  bf16_t* W = nullptr;
  hipMalloc(&W, 8 * 131072);
  hipMemset(W, 0, 8 * 131072);
  float* bias = nullptr;
  hipMalloc(&bias, 4096);
  hipMemset(bias, 0, 4096);
  long long* stamps = nullptr;
  hipMalloc(&stamps, 2 * 8 * 5 * 64 * sizeof(long long));
  float* sink = nullptr;
  hipMalloc(&sink, 4096);
  printf("# %d CUs; s_memtime ticks are 100 MHz (10 ns) on gfx9\n", cus);
  for (int grid : {cus, 1}) {
    printf("# grid = %d\n", grid);
    run<0>("full stage", W, bias, stamps, sink, grid);
    run<1>("k-loops only (no epilogue, no barriers)", W, bias, stamps, sink, grid);
    run<2>("k-loops without weight loads", W, bias, stamps, sink, grid);
    run<4>("epilogue + barriers only", W, bias, stamps, sink, grid);
  }
  return 0;
}
