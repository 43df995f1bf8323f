// What can one CU's vector L1 deliver to gathers of the sampler's shape?  (tools/probes: measurement only, not part of the library)
//
// Every lane loads 16 B (global_load_dwordx4); the lanes of a GROUP (4, 8 or 64 lanes) read consecutive 16-B chunks of one
// randomly placed, 64-B-aligned span inside a footprint that is either L1-resident (16 KB per workgroup) or far larger than
// the L2 (so most lines come through the miss path).  Mode 4 is msda_gsamp_kernel's pattern: a quad = one 64-byte
// (pixel, head) row.  Prints bytes / clk / CU (at the measured shader clock) for each mode.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int GROUP, int BATCH>
__global__ __launch_bounds__(256) void gather_kernel(const char* __restrict__ base, long footprint_per_wg, int shared_fp,
                                                     int iters, float* __restrict__ sink) {
  const int tid = threadIdx.x, lane = tid & 63, grp = tid / GROUP, sub = tid % GROUP;
  const char* mine = base + (shared_fp ? 0 : (long)blockIdx.x * footprint_per_wg);
  const unsigned spans = (unsigned)(footprint_per_wg / (GROUP * 16));
  unsigned state = (blockIdx.x * 977u + grp * 131u + 7u) * 2654435761u;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x4 v[BATCH];
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      state = state * 1664525u + 1013904223u;
      const unsigned span = (state >> 8) % spans;
      v[b] = *reinterpret_cast<const f32x4*>(mine + (long)span * (GROUP * 16) + sub * 16);
    }
#pragma unroll
    for (int b = 0; b < BATCH; ++b) acc += v[b];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x * 256 + tid] = acc.x + lane;
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// the same with 8 bytes per lane (global_load_dwordx2): GROUP lanes x 8 B per span
template <int GROUP, int BATCH>
__global__ __launch_bounds__(256) void gather8_kernel(const char* __restrict__ base, long footprint_per_wg, int shared_fp,
                                                      int iters, float* __restrict__ sink) {
  const int tid = threadIdx.x, lane = tid & 63, grp = tid / GROUP, sub = tid % GROUP;
  const char* mine = base + (shared_fp ? 0 : (long)blockIdx.x * footprint_per_wg);
  const unsigned spans = (unsigned)(footprint_per_wg / (GROUP * 8));
  unsigned state = (blockIdx.x * 977u + grp * 131u + 7u) * 2654435761u;
  f32x2_t acc = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x2_t v[BATCH];
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      state = state * 1664525u + 1013904223u;
      const unsigned span = (state >> 8) % spans;
      v[b] = *reinterpret_cast<const f32x2_t*>(mine + (long)span * (GROUP * 8) + sub * 8);
    }
#pragma unroll
    for (int b = 0; b < BATCH; ++b) acc += v[b];
  }
  if (acc.x + acc.y == 12345.678f) sink[blockIdx.x * 256 + tid] = acc.x + lane;
}

template <int GROUP, int BATCH>
double run8(const char* buf, long fp, int shared_fp, int grid, int iters, float* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  gather8_kernel<GROUP, BATCH><<<grid, 256>>>(buf, fp, shared_fp, 8, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  gather8_kernel<GROUP, BATCH><<<grid, 256>>>(buf, fp, shared_fp, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return (double)grid * 256 * 8.0 * BATCH * iters / (ms * 1e-3);
}

template <int GROUP>
double run(const char* buf, long fp, int shared_fp, int grid, int iters, float* sink, double* clk_mhz) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  gather_kernel<GROUP, 16><<<grid, 256>>>(buf, fp, shared_fp, 8, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  gather_kernel<GROUP, 16><<<grid, 256>>>(buf, fp, shared_fp, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  (void)clk_mhz;
  const double bytes = (double)grid * 256 * 16.0 * 16 * iters;
  return bytes / (ms * 1e-3);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;   // Hz
  const long big = 2L << 30;
  char* buf = nullptr;
  hipMalloc(&buf, big);
  hipMemset(buf, 0, big);
  float* sink = nullptr;
  hipMalloc(&sink, 64 << 20);
  printf("# %s, %d CUs, %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
  printf("# waves/CU  footprint            group  TB/s   B/clk/CU\n");
  for (int wg_per_cu : {2, 4, 5, 8}) {
    const int grid = cus * wg_per_cu;
    struct { const char* name; long fp; int shared; int iters; } fps[] = {
        {"16 KB / CU-set (L1 hits)", 16 << 10, 1, 2000},
        {"2 MB shared (L2 hits)", 2 << 20, 1, 400},
        {"64 KB per WG (L2/MALL)", 64 << 10, 0, 400},
        {"2 GB shared (HBM)", big, 1, 100}};
    for (auto& f : fps) {
      double r4 = run<4>(buf, f.fp, f.shared, grid, f.iters, sink, nullptr);
      double r8 = run<8>(buf, f.fp, f.shared, grid, f.iters, sink, nullptr);
      double r64 = run<64>(buf, f.fp, f.shared, grid, f.iters, sink, nullptr);
      printf("%6d     %-26s  4   %7.2f  %6.1f\n", wg_per_cu * 4, f.name, r4 / 1e12, r4 / clk / cus);
      printf("%6d     %-26s  8   %7.2f  %6.1f\n", wg_per_cu * 4, f.name, r8 / 1e12, r8 / clk / cus);
      printf("%6d     %-26s  64  %7.2f  %6.1f\n", wg_per_cu * 4, f.name, r64 / 1e12, r64 / clk / cus);
      double q8 = run8<8, 16>(buf, f.fp, f.shared, grid, f.iters, sink), q8b = run8<8, 32>(buf, f.fp, f.shared, grid, f.iters, sink);
      double q16 = run8<16, 32>(buf, f.fp, f.shared, grid, f.iters, sink);
      printf("%6d     %-26s  8 lanes x 8 B (64-B span), 16 / 32 loads in flight: %6.1f / %6.1f B/clk/CU;  16 lanes x 8 B (128-B span), 32 in flight: %6.1f\n",
             wg_per_cu * 4, f.name, q8 / clk / cus, q8b / clk / cus, q16 / clk / cus);
    }
  }
  return 0;
}
