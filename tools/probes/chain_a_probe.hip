// Phase timeline of chain A (csrc/chain.hip: tile load, 3 stage GEMMs with their epilogues, attn store, last pose layer) with
// s_memtime stamps, 1 or 2 workgroups per CU.  (measurement only; the body below follows chain_a_kernel / chain_a_body of the
// library step by step, with the library's own building blocks from chain_dev.h)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "chain_dev.h"

constexpr int RM = 128, NT = 256, JN = 2, MT = 4, NSTAMP = 12;

__global__ __launch_bounds__(NT, 2) void chain_a_probe(const bf16_t* __restrict__ samp, const bf16_t* __restrict__ Wp,
                                                       const float* __restrict__ bp, const bf16_t* __restrict__ W0,
                                                       const bf16_t* __restrict__ W1, bf16_t* __restrict__ attn,
                                                       float* __restrict__ o, long long* __restrict__ stamps, int skip) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* act = smem;
  const int tid = threadIdx.x, lane = tid & 63, rl = lane & 31;
  const int r0 = blockIdx.x * RM;
  const bool rec = lane == 0 && (blockIdx.x % 37) == 0 && blockIdx.x / 37 < 8;
  long long* st = stamps + ((blockIdx.x / 37) * 4 + (tid >> 6)) * NSTAMP;
  int ns = 0;
#define STAMP() do { if (rec) st[ns] = __builtin_readcyclecounter(); ++ns; } while (0)
  STAMP();                                                              // 0 start
  {
    f32x4 x[RM * 32 / NT];
#pragma unroll
    for (int i = 0; i < RM * 32 / NT; ++i) {
      const int c = i * NT + tid, row = c >> 5, v16 = c & 31;
      x[i] = *reinterpret_cast<const f32x4*>(samp + (long)(r0 + row) * 256 + v16 * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < RM * 32 / NT; ++i) {
      const int c = i * NT + tid, row = c >> 5, v16 = c & 31;
      *reinterpret_cast<f32x4*>(act + row * ACT_PITCH + v16 * 16) = x[i];
    }
  }
  const int rot = (((tid >> 6) & 3) * 3) & 15;
  f32x16 acc[MT][JN];
  bool keep[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) keep[mt] = true;
  __syncthreads();
  STAMP();                                                              // 1 tile in LDS
  f32x4 pf[4][JN], bvr[JN][4];
  if (!(skip & 1)) stage_gemm<MT, 16, 4, JN>(act, Wp, acc, tid, true, rot);
  STAMP();                                                              // 2 k-loop 1
  load_bias<JN>(bp, bvr, tid, MT * 32);
  ring_prefetch<16, 4, JN, MT>(W0, pf, tid, rot + 5);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  STAMP();                                                              // 3 barrier
  if (!(skip & 2)) write_act_pre<MT, JN>(act, acc, bvr, false, keep, tid);
  __syncthreads();
  STAMP();                                                              // 4 epilogue 1 + barrier
  if (!(skip & 4)) {
#pragma unroll
    for (int c0 = 0; c0 < RM * 32; c0 += NT) {
      const int c = c0 + tid, row = c >> 5, v16 = c & 31;
      *reinterpret_cast<f32x4*>(attn + (long)(r0 + row) * 256 + v16 * 8) = *reinterpret_cast<const f32x4*>(act + row * ACT_PITCH + v16 * 16);
    }
  }
  STAMP();                                                              // 5 attn store issued
  if (!(skip & 1)) stage_gemm<MT, 16, 4, JN, true>(act, W0, acc, tid, true, rot + 5, 16 * 1024, pf);
  STAMP();                                                              // 6 k-loop 2
  load_bias<JN>(bp, bvr, tid, MT * 32);
  ring_prefetch<16, 4, JN, MT>(W1, pf, tid, rot + 10);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  if (!(skip & 2)) write_act_pre<MT, JN>(act, acc, bvr, true, keep, tid);
  __syncthreads();
  STAMP();                                                              // 7 epilogue 2 + barriers
  if (!(skip & 1)) stage_gemm<MT, 16, 4, JN, true>(act, W1, acc, tid, true, rot + 10, 16 * 1024, pf);
  STAMP();                                                              // 8 k-loop 3
  __syncthreads();
  if (!(skip & 2)) write_act<MT, JN>(act, acc, bp, true, keep, tid);
  __syncthreads();
  STAMP();                                                              // 9 epilogue 3 + barriers
  // last layer stand-in: 2 threads per row read half a row each (the library's VALU dot, without the weights)
  {
    const int row = tid >> 1, part = tid & 1;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 128; c += 8) {
      const uint4 hv = *reinterpret_cast<const uint4*>(act + row * ACT_PITCH + (part * 128 + c) * 2);
      s += __uint_as_float(hv.x << 16) + __uint_as_float(hv.y << 16) + __uint_as_float(hv.z << 16) + __uint_as_float(hv.w << 16);
    }
    if (!(skip & 8)) o[(long)(r0 + row) * 3 + part] = s;
  }
  STAMP();                                                              // 10 last layer
  if (acc[0][0][0] == 1234.5f && rl == 77) o[0] = acc[1][1][3];
}

int main() {
  const int smem = RM * ACT_PITCH + RM * 4 + 768 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_a_probe), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int max_tiles = 1024;
  bf16_t *samp, *W, *attn;
  float *bias, *o;
  long long* stamps;
  hipMalloc(&samp, (size_t)max_tiles * RM * 512);
  hipMemset(samp, 0, (size_t)max_tiles * RM * 512);
  hipMalloc(&attn, (size_t)max_tiles * RM * 512);
  hipMalloc(&W, 3 * 131072);
  hipMemset(W, 0, 3 * 131072);
  hipMalloc(&bias, 4096);
  hipMemset(bias, 0, 4096);
  hipMalloc(&o, (size_t)max_tiles * RM * 12);
  hipMalloc(&stamps, 8 * 4 * NSTAMP * sizeof(long long));
  const char* names[] = {"tile load", "k-loop 1", "barrier", "epilogue 1 + barrier", "attn store", "k-loop 2", "epilogue 2 + barriers",
                         "k-loop 3", "epilogue 3 + barriers", "last layer"};
  for (int tiles : {256, 512}) {
    for (int skip : {0, 1, 2, 4, 3}) {
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipMemset(stamps, 0, 8 * 4 * NSTAMP * sizeof(long long));
      chain_a_probe<<<tiles, NT, smem>>>(samp, W, bias, W + 65536, W + 131072, attn, o, stamps, skip);
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int r = 0; r < 20; ++r) chain_a_probe<<<tiles, NT, smem>>>(samp, W, bias, W + 65536, W + 131072, attn, o, stamps, skip);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, a, b);
      std::vector<long long> h(8 * 4 * NSTAMP);
      hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      printf("tiles %4d  skip %d (1 = k-loops, 2 = epilogues, 4 = attn store): %6.1f us per launch; clk per phase (mean of 8 workgroups x 4 waves):\n",
             tiles, skip, ms * 1e3 / 20);
      double tot = 0;
      for (int p = 0; p < 10; ++p) {
        double s = 0;
        int n = 0;
        for (int w = 0; w < 32; ++w)
          if (h[w * NSTAMP] != 0) { s += double(h[w * NSTAMP + p + 1] - h[w * NSTAMP + p]); ++n; }
        printf("   %-24s %8.0f\n", names[p], n ? s / n : 0.0);
        tot += n ? s / n : 0.0;
      }
      printf("   %-24s %8.0f  (= %.1f us at 2.1 GHz)\n", "total", tot, tot / 2100.0);
    }
  }
  return 0;
}
