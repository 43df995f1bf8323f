// Shared device helpers for libmvgformer_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvg_decoder.h"

#define MVG_WAVE 64
#define MVG_MAX_DEVICES 64   // per-device one-time kernel configuration flags

typedef unsigned short bf16_t;  // raw bf16 bits

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even, NaN preserved
// fp32 -> bf16, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (one instruction for two values; the integer
// emulation -- NaN test, rounding add, shift -- was ~5 VALU instructions and a branch per value)
typedef __bf16 mvg_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  const mvg_bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16(f, 0.f) & 0xffffu); }

// ---- 4-channel vector load / store in either storage type (always fp32 in registers)
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4<bf16_t> {
  static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
    u16x4 r = *reinterpret_cast<const u16x4*>(p);
    f32x4 v;
    v[0] = bf16_to_f32(r[0]); v[1] = bf16_to_f32(r[1]); v[2] = bf16_to_f32(r[2]); v[3] = bf16_to_f32(r[3]);
    return v;
  }
  static __device__ __forceinline__ void store(bf16_t* p, f32x4 v) {
    *reinterpret_cast<uint2*>(p) = uint2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
  }
};

template <typename T> __device__ __forceinline__ float load1(const T* p);
template <> __device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load1<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void store1(T* p, float v);
template <> __device__ __forceinline__ void store1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store1<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }
// 4 consecutive elements, pointer aligned to 4 elements
template <typename T> __device__ __forceinline__ void store_vec4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store_vec4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store_vec4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(p) = uint2{pack_bf16(a, b), pack_bf16(c, d)};
}

// Levels are passed by value to kernels (L <= MVG_MAX_LEVELS)
#define MVG_MAX_LEVELS 8
struct LevelTable {
  int H[MVG_MAX_LEVELS];
  int W[MVG_MAX_LEVELS];
  int start[MVG_MAX_LEVELS];
  float invH[MVG_MAX_LEVELS];   // 1.0f / H, 1.0f / W (IEEE fp32 division on the host: what the kernels used to compute)
  float invW[MVG_MAX_LEVELS];
  float Hf[MVG_MAX_LEVELS];     // (float)H, (float)W and the upper bounds H + 1, W + 1 of index_safe: per-sample conversions / adds
  float Wf[MVG_MAX_LEVELS];     // of the samplers' inner loops otherwise (wave-uniform: scalar loads from the kernel arguments)
  float Hp1[MVG_MAX_LEVELS];
  float Wp1[MVG_MAX_LEVELS];
  int L;
};

static inline int mvg_fill_levels(LevelTable* t, const int64_t* shapes_host, const int64_t* starts_host, int L) {
  if (L < 1 || L > MVG_MAX_LEVELS) return MVG_E_BADARG;
  t->L = L;
  for (int l = 0; l < L; ++l) {
    t->H[l] = (int)shapes_host[2 * l];
    t->W[l] = (int)shapes_host[2 * l + 1];
    t->start[l] = (int)starts_host[l];
    t->invH[l] = 1.0f / (float)t->H[l];
    t->invW[l] = 1.0f / (float)t->W[l];
    t->Hf[l] = (float)t->H[l];
    t->Wf[l] = (float)t->W[l];
    t->Hp1[l] = (float)t->H[l] + 1.f;
    t->Wp1[l] = (float)t->W[l] + 1.f;
  }
  return 0;
}

#define MVG_LAUNCH_CHECK()                 \
  do {                                     \
    hipError_t _e = hipGetLastError();     \
    if (_e != hipSuccess) return (int)_e;  \
  } while (0)

static inline int mvg_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
