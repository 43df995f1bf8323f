// Geometry + small per-token kernels of the MVGFormer decoder layer for gfx950:
// pyramid packing, camera projection, reference-point feature gather, view mean,
// residual+LayerNorm, class head, last pose-MLP layer, and the fused
// refine-2D -> undistort -> DLT -> smallest-singular-vector -> scatter kernel.
#include "common.h"

// ------------------------------------------------------------------------------------------
// NCHW level -> channels-last pyramid rows.  64 (pixels) x 64 (channels) tile transposed through LDS (65-float
// pitch: conflict-free both ways): reads are 256-byte runs of one channel plane, writes 128-byte (bf16) /
// 256-byte (fp32) runs of one pixel row, 4 channels per lane.
template <typename T>
__device__ __forceinline__ void pack_tile(const float* __restrict__ src, T* __restrict__ feat, int C, int HW, int S, int start,
                                          int bx, float (&tile)[64][65]) {
  const int n = blockIdx.z, tid = threadIdx.x;
  const int hw0 = bx * 64, c0 = blockIdx.y * 64;
  const int tx = tid & 63, ty = tid >> 6;
  const float* sp = src + (long)n * C * HW;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {    // clamped address + select: every load of the thread is in flight at once
    const int c = min(c0 + ty + 4 * i, C - 1), hw = min(hw0 + tx, HW - 1);
    v[i] = sp[(long)c * HW + hw];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[ty + 4 * i][tx] = v[i];
  __syncthreads();
  T* fp = feat + ((long)n * S + start) * C;
  const int cq = tid & 15, p = tid >> 4, c = c0 + 4 * cq;
  const bool vec = (C & 3) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int hw = hw0 + p + 16 * i;
    if (hw >= HW || c >= C) continue;
    const float a0 = tile[4 * cq][p + 16 * i], a1 = tile[4 * cq + 1][p + 16 * i], a2 = tile[4 * cq + 2][p + 16 * i],
                a3 = tile[4 * cq + 3][p + 16 * i];
    T* dst = fp + (long)hw * C + c;
    if (vec) {
      store_vec4<T>(dst, a0, a1, a2, a3);
    } else {
      const float a[4] = {a0, a1, a2, a3};
      for (int k = 0; k < 4 && c + k < C; ++k) store1<T>(dst + k, a[k]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_level_kernel(const float* __restrict__ src, T* __restrict__ feat, int C,
                                                         int HW, int S, int start) {
  __shared__ float tile[64][65];
  pack_tile<T>(src, feat, C, HW, S, start, blockIdx.x, tile);
}

// all levels of the pyramid in ONE launch (round 3): blockIdx.x walks the 64-pixel tiles of level 0, then level 1, ...
struct PackLevels {
  const float* src[MVG_MAX_LEVELS];
  int hw[MVG_MAX_LEVELS], start[MVG_MAX_LEVELS], tile0[MVG_MAX_LEVELS + 1];
  int L;
};
template <typename T>
__global__ __launch_bounds__(256) void pack_pyramid_kernel(PackLevels pl, T* __restrict__ feat, int C, int S) {
  __shared__ float tile[64][65];
  int l = 0;
#pragma unroll
  for (int k = 1; k < MVG_MAX_LEVELS; ++k)
    if (k < pl.L && (int)blockIdx.x >= pl.tile0[k]) l = k;
  pack_tile<T>(pl.src[l], feat, C, pl.hw[l], S, pl.start[l], (int)blockIdx.x - pl.tile0[l], tile);
}

// ------------------------------------------------------------------------------------------
// A.1 projection: pinhole + radial/tangential distortion, in-image test, clamp, crop affine.
// one (image n, token q) projection of the point (x0, x1, x2) [mm]; idx = n * Lq + q
__device__ __forceinline__ void project_point(const float x0, const float x1, const float x2, const float* __restrict__ cam,
                                              const LevelTable& lv, float* __restrict__ r, float* __restrict__ ref_lvl,
                                              uint8_t* __restrict__ inside, const long idx) {
  const float d0 = x0 - cam[9], d1 = x1 - cam[10], d2 = x2 - cam[11];   // cameras.py:188
  const float xc0 = cam[0] * d0 + cam[1] * d1 + cam[2] * d2;
  const float xc1 = cam[3] * d0 + cam[4] * d1 + cam[5] * d2;
  const float xc2 = cam[6] * d0 + cam[7] * d1 + cam[8] * d2;
  const float zz = xc2 + 1e-5f;                                                    // cameras.py:190
  float y0 = xc0 / zz, y1 = xc1 / zz;
  const float r2 = y0 * y0 + y1 * y1;
  const float radial = 1.f + (cam[16] * r2 + cam[17] * (r2 * r2) + cam[18] * (r2 * r2 * r2));   // :195-198
  const float tang = cam[19] * y1 + cam[20] * y0;                                  // :200
  const float corr = radial + 2.f * tang;
  y0 = y0 * corr + cam[20] * r2;                                                   // :201-204
  y1 = y1 * corr + cam[19] * r2;
  float u = cam[12] * y0 + cam[14];                                                // :206
  float v = cam[13] * y1 + cam[15];
  const bool in_img = (u >= 0.f) && (v >= 0.f) && (u < cam[33]) && (v < cam[34]);  // dq_decoder.py:374-379
  u = fminf(fmaxf(u, -1.f), cam[35]);                                              // :382-383
  v = fminf(fmaxf(v, -1.f), cam[35]);
  const float nx = cam[21] * u + cam[22] * v + cam[23];                            // transforms.py:135-141
  const float ny = cam[24] * u + cam[25] * v + cam[26];
  const float rx = nx / cam[36], ry = ny / cam[37];                                // dq_decoder.py:390-392
  r[idx * 2] = rx;
  r[idx * 2 + 1] = ry;
  for (int l = 0; l < lv.L; ++l) {                                                 // dq_decoder.py:570-573
    const float Wf = (float)lv.W[l], Hf = (float)lv.H[l];
    ref_lvl[(idx * lv.L + l) * 2] = (rx * Wf) / (Wf - 1.f);
    ref_lvl[(idx * lv.L + l) * 2 + 1] = (ry * Hf) / (Hf - 1.f);
  }
  inside[idx] = in_img ? 1 : 0;
}

__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ X, const float* __restrict__ cams,
                                                      LevelTable lv, float* __restrict__ r,
                                                      float* __restrict__ ref_lvl, uint8_t* __restrict__ inside,
                                                      int B, int Lq, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % Lq);
  const int n = (int)(idx / Lq);
  const int b = n % B;
  const float* xp = X + ((long)b * Lq + q) * 3;
  project_point(xp[0], xp[1], xp[2], cams + (long)n * MVG_CAM_STRIDE, lv, r, ref_lvl, inside, idx);
}

// ------------------------------------------------------------------------------------------
// A.2 + A.3(1): bilinear ref-point features (grid_sample, zeros padding, align_corners=False)
// of every level + query.  One wavefront per (image, query); lane owns 4 channels per 256.
template <typename T>
__global__ __launch_bounds__(256) void gather_ref_kernel(const T* __restrict__ feat, const float* __restrict__ ref_lvl,
                                                         const float* __restrict__ x, LevelTable lv,
                                                         T* __restrict__ ain, int B, int Lq, int S, int C,
                                                         int n_pairs) {
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n_pairs) return;
  const int lane = threadIdx.x & 63;
  const int q = pair % Lq, n = pair / Lq, b = n % B;
  const float* xq = x + ((long)b * Lq + q) * C;
  const T* fbase = feat + (long)n * S * C;
  for (int l = 0; l < lv.L; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const float Wf = (float)W, Hf = (float)H;
    const float refx = ref_lvl[((long)pair * lv.L + l) * 2], refy = ref_lvl[((long)pair * lv.L + l) * 2 + 1];
    const float gx = fminf(fmaxf(refx * 2.f - 1.f, -1.1f), 1.1f);                 // projattn.py:134
    const float gy = fminf(fmaxf(refy * 2.f - 1.f, -1.1f), 1.1f);
    const float ix = ((gx + 1.f) * Wf - 1.f) * 0.5f;                              // grid_sample unnormalize
    const float iy = ((gy + 1.f) * Hf - 1.f) * 0.5f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = ix - x0f, ty = iy - y0f;
    const bool x0ok = x0 >= 0 && x0 < W, x1ok = x1 >= 0 && x1 < W, y0ok = y0 >= 0 && y0 < H, y1ok = y1 >= 0 && y1 < H;
    const float w00 = (x0ok && y0ok) ? (1.f - tx) * (1.f - ty) : 0.f;
    const float w10 = (x1ok && y0ok) ? tx * (1.f - ty) : 0.f;
    const float w01 = (x0ok && y1ok) ? (1.f - tx) * ty : 0.f;
    const float w11 = (x1ok && y1ok) ? tx * ty : 0.f;
    const int x0c = min(max(x0, 0), W - 1), x1c = min(max(x1, 0), W - 1);
    const int y0c = min(max(y0, 0), H - 1), y1c = min(max(y1, 0), H - 1);
    const T* p00 = fbase + ((long)lv.start[l] + (long)y0c * W + x0c) * C;
    const T* p10 = fbase + ((long)lv.start[l] + (long)y0c * W + x1c) * C;
    const T* p01 = fbase + ((long)lv.start[l] + (long)y1c * W + x0c) * C;
    const T* p11 = fbase + ((long)lv.start[l] + (long)y1c * W + x1c) * C;
    T* op = ain + ((long)pair * lv.L + l) * C;
    for (int c = lane * 4; c < C; c += 256) {
      const f32x4 v = w00 * Vec4<T>::load(p00 + c) + w10 * Vec4<T>::load(p10 + c) + w01 * Vec4<T>::load(p01 + c) +
                      w11 * Vec4<T>::load(p11 + c);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xq + c);
      Vec4<T>::store(op + c, v + xv);
    }
  }
}

// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mean_views_kernel(const T* __restrict__ attn, T* __restrict__ out, int V,
                                                         long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 acc = Vec4<T>::load(attn + i * 4);
  for (int v = 1; v < V; ++v) acc += Vec4<T>::load(attn + ((long)v * n4 + i) * 4);
  Vec4<T>::store(out + i * 4, acc / (float)V);
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// y = LN(res + h): one wavefront per row, C = 256 -> 4 channels per lane (any C % 4 == 0, C <= 1024).
template <typename T>
__global__ __launch_bounds__(256) void add_ln_kernel(const float* __restrict__ res, const T* __restrict__ h,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, int rows, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  f32x4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < C) {
      v[i] = *reinterpret_cast<const f32x4*>(res + (long)row * C + c) + Vec4<T>::load(h + (long)row * C + c);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (lane * 4 + 256 * i < C) {
      const f32x4 d = v[i] - mean;
      ss += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)C + 1e-5f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < C) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), bb = *reinterpret_cast<const f32x4*>(beta + c);
      *reinterpret_cast<f32x4*>(y + (long)row * C + c) = (v[i] - mean) * rstd * g + bb;
    }
  }
}

// ------------------------------------------------------------------------------------------
// A.5 class head: prob[b,i,:] = mean_j sigmoid(Wc tgt[b, i*J+j] + bc); one wavefront per query.
__global__ __launch_bounds__(256) void class_head_kernel(const float* __restrict__ tgt, const float* __restrict__ Wc,
                                                         const float* __restrict__ bc, float threshold,
                                                         const uint8_t* __restrict__ forced, float* __restrict__ prob,
                                                         uint8_t* __restrict__ valid, int* __restrict__ any_valid,
                                                         int nq_total, int J, int C) {
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= nq_total) return;
  const int lane = threadIdx.x & 63;
  float p0 = 0.f, p1 = 0.f;
  if (C == 256) {
    // the query's J token rows are requested together (16 at a time), then reduced: as one row per iteration this was a chain
    // of J dependent round trips -- 23 us for 15 joints.  Same sums in the same order per row and per query: bit-identical.
    constexpr int JB = 16;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wc + lane * 4), w1 = *reinterpret_cast<const f32x4*>(Wc + C + lane * 4);
    const float b0 = bc[0], b1 = bc[1];
    for (int j0 = 0; j0 < J; j0 += JB) {
      f32x4 tv[JB];
#pragma unroll
      for (int jj = 0; jj < JB; ++jj)
        tv[jj] = *reinterpret_cast<const f32x4*>(tgt + ((long)qi * J + min(j0 + jj, J - 1)) * C + lane * 4);
      float a0[JB], a1[JB];
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        a0[jj] = 0.f + (tv[jj][0] * w0[0] + tv[jj][1] * w0[1] + tv[jj][2] * w0[2] + tv[jj][3] * w0[3]);
        a1[jj] = 0.f + (tv[jj][0] * w1[0] + tv[jj][1] * w1[1] + tv[jj][2] * w1[2] + tv[jj][3] * w1[3]);
      }
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        a0[jj] = wave_sum(a0[jj]) + b0;
        a1[jj] = wave_sum(a1[jj]) + b1;
      }
#pragma unroll
      for (int jj = 0; jj < JB; ++jj)
        if (j0 + jj < J) {
          p0 += 1.f / (1.f + expf(-a0[jj]));
          p1 += 1.f / (1.f + expf(-a1[jj]));
        }
    }
  } else {
    for (int j = 0; j < J; ++j) {
      const float* t = tgt + ((long)qi * J + j) * C;
      float a0 = 0.f, a1 = 0.f;
      for (int c = lane * 4; c < C; c += 256) {
        const f32x4 tv = *reinterpret_cast<const f32x4*>(t + c);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(Wc + c), w1 = *reinterpret_cast<const f32x4*>(Wc + C + c);
        a0 += tv[0] * w0[0] + tv[1] * w0[1] + tv[2] * w0[2] + tv[3] * w0[3];
        a1 += tv[0] * w1[0] + tv[1] * w1[1] + tv[2] * w1[2] + tv[3] * w1[3];
      }
      a0 = wave_sum(a0) + bc[0];
      a1 = wave_sum(a1) + bc[1];
      p0 += 1.f / (1.f + expf(-a0));
      p1 += 1.f / (1.f + expf(-a1));
    }
  }
  if (lane == 0) {
    p0 /= (float)J;
    p1 /= (float)J;
    prob[2 * (long)qi] = p0;
    prob[2 * (long)qi + 1] = p1;
    const bool ok = forced ? (forced[qi] != 0) : (p1 > threshold);   // dq_decoder.py:605
    valid[qi] = ok ? 1 : 0;
    if (ok) atomicOr(any_valid, 1);
  }
}

// last pose_embed layer: 3 dot products per row, one wavefront per row.
template <typename T>
__global__ __launch_bounds__(256) void rowdot3_kernel(const T* __restrict__ h, const float* __restrict__ W3,
                                                      const float* __restrict__ b3, float* __restrict__ o, int rows,
                                                      int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float a[3] = {0.f, 0.f, 0.f};
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 hv = Vec4<T>::load(h + (long)row * C + c);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(W3 + k * C + c);
      a[k] += hv[0] * w[0] + hv[1] * w[1] + hv[2] * w[2] + hv[3] * w[3];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) a[k] = wave_sum(a[k]);
  if (lane < 3) o[(long)row * 3 + lane] = (lane == 0 ? a[0] : lane == 1 ? a[1] : a[2]) + b3[lane];
}

// ------------------------------------------------------------------------------------------
// A.6-A.8.  Eight lanes per (batch, query, joint).
//  - refined 2D = (r + (dx,dy)/img) * img, view confidence = softmax over views of the logit
//  - un-crop (inverse affine), K^-1, 5 fixed-point undistortion iterations, K
//  - DLT rows conf*(x*P2 - P0), conf*(y*P2 - P1) with P = K [R | -R T] in fp32 exactly as the
//    reference builds them; the 4x4 Gram matrix of the 2V rows is accumulated in fp64 and its
//    smallest eigenvector (= smallest right singular vector of A, torch.linalg.svd's Vh[3]) is
//    found with cyclic Jacobi rotations in fp64.  Squaring the condition number in fp64
//    (eps 1.1e-16) leaves far more headroom than an fp32 SVD of A (eps 6e-8) has.
__device__ __forceinline__ void jacobi_angle3(const double app, const double aqq, const double apq, double& c, double& s);
__device__ __forceinline__ void jacobi_angle(const double (&a)[4][4], const int p, const int q, double& c, double& s) {
  jacobi_angle3(a[p][p], a[q][q], a[p][q], c, s);
}
__device__ __forceinline__ void jacobi_angle3(const double app, const double aqq, const double apq, double& c, double& s) {
  // The rotation ANGLE only steers convergence, so it is computed in fp32 (one fast division, one sqrt); what
  // must hold to fp64 precision is c^2 + s^2 = 1 (the similarity transform stays orthogonal): c = rsqrt(1+t^2)
  // starts from the fp32 rsqrt and takes two Newton steps in fp64.  (IEEE fp64 div/sqrt sequences were ~80 % of
  // this kernel's time.)
  // Branch-free: "nothing to rotate" (a[p][q] = 0 or negligible against the diagonal gap: theta infinite or 0/0) is a select
  // at the end.  As two early returns each angle sat in its own exec-mask region and the two angles of a round -- independent
  // chains of ~30 dependent instructions -- ran one after the other instead of interleaved.
  const float theta = (float)(aqq - app) / (2.f * (float)apq);
  const bool rotate = !(fabs(apq) < 1e-300) && (fabsf(theta) <= 3.0e38f);
  const float tf = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
  const double t = (double)tf;
  const double w = t * t + 1.0;
  double cc = (double)rsqrtf((float)w);
  cc = cc * (1.5 - 0.5 * w * cc * cc);
  cc = cc * (1.5 - 0.5 * w * cc * cc);
  c = rotate ? cc : 1.0;
  s = rotate ? t * cc : 0.0;
}

// The two halves of a plane rotation with their roundings pinned (one product rounded, then one fma), so that the one-lane
// Jacobi below and the lane-parallel one of triangulate_kernel give the same bits:  x' = c x - s y,  y' = s x + c y.
__device__ __forceinline__ double rot_lo(double c, double x, double s, double y) { return __builtin_fma(c, x, -(s * y)); }
__device__ __forceinline__ double rot_hi(double c, double x, double s, double y) { return __builtin_fma(s, x, c * y); }

__device__ __forceinline__ void jacobi_apply(double (&a)[4][4], double (&v)[4][4], const int p, const int q, const double c,
                                             const double s) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // columns p,q of A
    const double akp = a[k][p], akq = a[k][q];
    a[k][p] = rot_lo(c, akp, s, akq);
    a[k][q] = rot_hi(c, akp, s, akq);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // rows p,q of A
    const double apk = a[p][k], aqk = a[q][k];
    a[p][k] = rot_lo(c, apk, s, aqk);
    a[q][k] = rot_hi(c, apk, s, aqk);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double vkp = v[k][p], vkq = v[k][q];
    v[k][p] = rot_lo(c, vkp, s, vkq);
    v[k][q] = rot_hi(c, vkp, s, vkq);
  }
}

// Two rotations on disjoint index pairs: (p2, q2) does not see what (p1, q1) changes in the entries its angle is
// computed from, so both angles come from the same matrix -- two independent latency chains instead of one after
// the other -- and applying them in sequence IS the cyclic order (p1,q1), (p2,q2).
__device__ __forceinline__ void jacobi_rotate2(double (&a)[4][4], double (&v)[4][4], const int p1, const int q1,
                                               const int p2, const int q2) {
  double c1, s1, c2, s2;
  jacobi_angle(a, p1, q1, c1, s1);
  jacobi_angle(a, p2, q2, c2, s2);
  jacobi_apply(a, v, p1, q1, c1, s1);
  jacobi_apply(a, v, p2, q2, c2, s2);
}

// ---- the null direction of the DLT's Gram matrix by inverse iteration (round 6; multiview.py:208-221 takes the right singular
// vector of the smallest singular value of the row matrix A = the eigenvector of the smallest eigenvalue of G = A^T A).
// G - sigma I = L D L^T without pivoting (positive definite as long as sigma < lambda_4), then x <- (G - sigma I)^-1 x: the component
// along the smallest eigenvalue grows by (lambda_3 - sigma) / (lambda_4 - sigma) per solve, and a solve is 12 fma + 4 multiplies
// where a Jacobi sweep is ~350 dependent fp64 instructions (the cyclic Jacobi was 60 % of the triangulation kernel: ~2 100 dependent
// instructions on ONE wavefront per workgroup, profiles/r05_experiments.txt).
//   * sigma = 0 first: rays that meet have lambda_4 / lambda_3 = 1e-10 ... 1e-3 and settle in one or two rounds of 4 solves;
//   * rays that do not meet (a view whose projection was clamped at the image border contributes a wrong ray: a third of the
//     (view, query) pairs at cfg-2's synthetic poses) have ratios up to ~1: after every round that has not settled the shift moves
//     to sigma = rho - |G x - rho x| (Rayleigh quotient minus residual: an eigenvalue lies within the residual of rho, so sigma stays
//     below lambda_4 once x leans towards it -- and the new factors are only taken if all four pivots are positive, i.e. sigma
//     IS below lambda_4).  Convergence becomes superlinear: <= 3 rounds for > 99 % of such matrices (numpy model of this routine,
//     profiles/r06_experiments.txt section 6); what has not settled after MAXR = 12 rounds goes to the Jacobi.
// Accuracy: Cholesky-type factors inherit the scaling D0 G D0 of the matrix (the homogeneous column of P is 1e3 x the others), so the
// vector is exact to cond(scaled G) x eps -- 1e-6 mm against the fp64 SVD of the row matrix in the model, where LAPACK's eigh on G
// itself is off by up to millimetres.
// Per lane and deterministic: a round ends with the test "last two iterates parallel to 1e-11"; a lane that passed keeps its vector
// whatever its wavefront neighbours still do (a problem's result does not depend on which problems share its wavefront: query-
// sharded and single-rank runs agree bit for bit).  Returns false -- the caller falls back to the Jacobi -- for a matrix whose
// first three pivots are not safely positive (fewer than 2 useful views) or that has not settled.
struct Ldl4 {
  double i0, i1, i2, i3, l10, l20, l30, l21, l31, l32;
};
template <bool FIRST>
__device__ __forceinline__ bool ldl4_factor(const double (&G)[4][4], double sigma, Ldl4& f) {
  const double g10 = G[0][1], g20 = G[0][2], g30 = G[0][3], g21 = G[1][2], g31 = G[1][3], g32 = G[2][3];
  const double d0 = G[0][0] - sigma;
  const double i0 = 1.0 / d0;
  const double l10 = g10 * i0, l20 = g20 * i0, l30 = g30 * i0;
  const double d1 = G[1][1] - sigma - l10 * g10;
  const double i1 = 1.0 / d1;
  const double t21 = g21 - l20 * g10, t31 = g31 - l30 * g10;
  const double l21 = t21 * i1, l31 = t31 * i1;
  const double d2 = G[2][2] - sigma - l20 * g20 - l21 * t21;
  const double i2 = 1.0 / d2;
  const double t32 = g32 - l30 * g20 - l31 * t21;
  const double l32 = t32 * i2;
  double d3 = G[3][3] - sigma - l30 * g30 - l31 * t31 - l32 * t32;
  bool ok;
  if (FIRST) {
    // rays that meet exactly: d3 is rounding noise of either sign -- any tiny positive pivot makes the solve return the null direction
    d3 = fmax(d3, 1e-18 * G[3][3]);
    ok = d0 > 0.0 && d1 > 1e-13 * G[1][1] && d2 > 1e-13 * G[2][2];          // (false for NaN)
  } else {
    ok = d0 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0;                       // sigma < lambda_4
  }
  f.i0 = i0; f.i1 = i1; f.i2 = i2; f.i3 = 1.0 / d3;
  f.l10 = l10; f.l20 = l20; f.l30 = l30; f.l21 = l21; f.l31 = l31; f.l32 = l32;
  return ok;
}

// wanted == false: the caller discards this problem's result (a query that did not pass the filter: its output is zero,
// dq_decoder.py:887-967 triangulates the queries that passed only) -- the lane does not iterate and does not hold its wavefront.
__device__ __forceinline__ bool null_vector_invit(const double (&G)[4][4], double (&ev)[4], const bool wanted = true) {
  constexpr int MAXR = 12;
  Ldl4 f;
  const bool pivots_ok = ldl4_factor<true>(G, 0.0, f) || !wanted;
  double sigma = 0.0;
  double x0 = 1.0, x1 = 1.0, x2 = 1.0, x3 = 1.0;
  bool done = !wanted;
  for (int round = 0; round < MAXR; ++round) {
    double p0 = x0, p1 = x1, p2 = x2, p3 = x3, w0 = x0, w1 = x1, w2 = x2, w3 = x3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      p0 = w0; p1 = w1; p2 = w2; p3 = w3;
      const double y1 = w1 - f.l10 * w0;                            // L y = w
      const double y2 = w2 - f.l20 * w0 - f.l21 * y1;
      const double y3 = w3 - f.l30 * w0 - f.l31 * y1 - f.l32 * y2;
      const double z0 = w0 * f.i0, z1 = y1 * f.i1, z2 = y2 * f.i2;  // D z = y
      w3 = y3 * f.i3;                                               // L^T w = z
      w2 = z2 - f.l32 * w3;
      w1 = z1 - f.l21 * w2 - f.l31 * w3;
      w0 = z0 - f.l10 * w1 - f.l20 * w2 - f.l30 * w3;
    }
    // parallel?  every 2 x 2 minor of (w | p) against the product of the two largest components
    const double mw = fmax(fmax(fabs(w0), fabs(w1)), fmax(fabs(w2), fabs(w3)));
    const double mp = fmax(fmax(fabs(p0), fabs(p1)), fmax(fabs(p2), fabs(p3)));
    const double c01 = fabs(w0 * p1 - w1 * p0), c02 = fabs(w0 * p2 - w2 * p0), c03 = fabs(w0 * p3 - w3 * p0);
    const double c12 = fabs(w1 * p2 - w2 * p1), c13 = fabs(w1 * p3 - w3 * p1), c23 = fabs(w2 * p3 - w3 * p2);
    const double cmax = fmax(fmax(fmax(c01, c02), fmax(c03, c12)), fmax(c13, c23));
    const bool parallel = cmax <= 1e-11 * mw * mp;                  // (false for NaN / Inf)
    // rescale by a power of two (exact): no overflow however many rounds follow
    const int ex = ilogb(mw);
    const double s0 = scalbn(w0, -ex), s1 = scalbn(w1, -ex), s2 = scalbn(w2, -ex), s3 = scalbn(w3, -ex);
    if (!done) {
      x0 = s0; x1 = s1; x2 = s2; x3 = s3;
      done = parallel;
    }
    if (__all(done || !pivots_ok)) break;
    if (!done) {
      // not settled: lambda_4 / lambda_3 is not small.  Shift to just below the Rayleigh quotient for the next round.
      const double n2 = x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
      const double a0 = G[0][0] * x0 + G[0][1] * x1 + G[0][2] * x2 + G[0][3] * x3;
      const double a1 = G[0][1] * x0 + G[1][1] * x1 + G[1][2] * x2 + G[1][3] * x3;
      const double a2 = G[0][2] * x0 + G[1][2] * x1 + G[2][2] * x2 + G[2][3] * x3;
      const double a3 = G[0][3] * x0 + G[1][3] * x1 + G[2][3] * x2 + G[3][3] * x3;
      const double in2 = 1.0 / n2;
      const double rho = (x0 * a0 + x1 * a1 + x2 * a2 + x3 * a3) * in2;
      const double r0 = a0 - rho * x0, r1 = a1 - rho * x1, r2 = a2 - rho * x2, r3 = a3 - rho * x3;
      const double cand = rho - sqrt((r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3) * in2);
      if (cand > sigma) {
        Ldl4 f2;
        if (ldl4_factor<false>(G, cand, f2)) {
          f = f2;
          sigma = cand;
        }
      }
    }
  }
  ev[0] = x0; ev[1] = x1; ev[2] = x2; ev[3] = x3;
  return done && pivots_ok;
}

// Two phases per workgroup of 512 threads = 64 (batch, query, joint) problems:
//   1. 8 lanes per problem: lane `sub` handles views sub, sub+8, ... (view-softmax, un-crop, undistortion, its two
//      DLT rows and their contribution to the 4x4 Gram matrix) -- one load round trip for the whole problem;
//      the partial Gram matrices (upper triangle, fp64) go to LDS.
//   2. one lane per problem (the first wavefront): sums the 8 partials and takes the null direction of the Gram matrix by inverse
//      iteration (null_vector_invit; the fp64 Jacobi of rounds 1-5 is the fallback for matrices it declines).
// (All 8 lanes of a problem running the Jacobi redundantly made the kernel fp64-VALU-bound: 14 of its 22 us at
// cfg-2; one thread per problem for BOTH phases took 27 us: 240 wavefronts, each a chain of dependent loads.)
constexpr int TRI_PROBS = 64;           // problems per workgroup
constexpr int TRI_PAD = 9;              // doubles per (entry, problem) row: 8 partials + 1 pad (bank spread)

__device__ __forceinline__ float max8(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false)));
  return v;
}
__device__ __forceinline__ float add8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
  return v;
}

// ---- lane-parallel Jacobi (round 3): the 8 lanes that built a problem's DLT rows also diagonalise its Gram matrix.  Lane
// j < 4 holds COLUMN j of the symmetric 4 x 4 matrix, lane 4 + j column j of the accumulated rotations V.  A round of the
// cyclic order rotates the two disjoint pairs (0, M) and the other two, M = 1, 2, 3: the partner's column is one quad_perm
// exchange away (lane ^ M, in the A quad and in the V quad alike), a column update is then local to the two lanes of a
// pair, a row update local to every lane.  The operations and their order per matrix element are those of jacobi_rotate2
// (both angles from the matrix as the round finds it; rotation 1: columns, rows, V; rotation 2: columns, rows, V; roundings
// pinned by rot_lo / rot_hi), so the result is the one-lane Jacobi's bit for bit (tests: knob tri_lanes) -- on 8 wavefronts
// instead of one, and a wavefront stops when its own 8 problems have converged.
// MEASURED SLOWER and not the default: cfg-2 forward 1.279 -> 1.309 ms.  The one-lane form is 19 200 of the kernel's 32 900
// cycles (s_memtime) at 8.5 cycles per dependent fp64 instruction, but only ~150 instructions per round for 64 problems; this
// form needs ~200 per round and wavefront (two 32-bit DPP moves per exchanged double, selects between the p / q forms,
// the pair-1 / pair-2 / A-only steps each issued for the whole wavefront) on 8 wavefronts = two per SIMD: 1 600 cycles of
// fp64 issue per round and SIMD against 1 280 of latency.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void row_rot(double (&x)[4], const int p, const int q, const double c, const double s) {
  const double xp = x[p], xq = x[q];
  x[p] = rot_lo(c, xp, s, xq);
  x[q] = rot_hi(c, xp, s, xq);
}
template <int M>
__device__ __forceinline__ void jacobi_round_lanes(double (&col)[4], const int jj, const bool isA) {
  constexpr int XORC = M == 1 ? 0xB1 : (M == 2 ? 0x4E : 0x1B);          // quad_perm of lane ^ M
  constexpr int P1 = 0, Q1 = M, P2 = (M == 1 ? 2 : 1), Q2 = (M == 3 ? 2 : 3);
  double pc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) pc[k] = dpp_f64<XORC>(col[k]);
  const bool is_p = jj < (jj ^ M);                                       // the lower index of its pair
  const bool pair1 = jj == 0 || jj == M;
  // rotation angle of the lane's own pair, from the columns as the round finds them (A lanes; V lanes take the A lanes')
  double cp[4], cq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    cp[k] = is_p ? col[k] : pc[k];
    cq[k] = is_p ? pc[k] : col[k];
  }
  double c, s;
  jacobi_angle3(pair1 ? cp[P1] : cp[P2], pair1 ? cq[Q1] : cq[Q2], pair1 ? cq[P1] : cq[P2], c, s);
  double c1 = dpp_f64<P1 * 0x55>(c), s1 = dpp_f64<P1 * 0x55>(s), c2 = dpp_f64<P2 * 0x55>(c), s2 = dpp_f64<P2 * 0x55>(s);
  {
    const double c1v = dpp_f64<0x114>(c1), s1v = dpp_f64<0x114>(s1), c2v = dpp_f64<0x114>(c2), s2v = dpp_f64<0x114>(s2);   // row_shr:4
    if (!isA) { c1 = c1v; s1 = s1v; c2 = c2v; s2 = s2v; }
  }
  // rotation 1: columns P1, Q1 (of A and of V) ...
  if (pair1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) col[k] = is_p ? rot_lo(c1, col[k], s1, pc[k]) : rot_hi(c1, pc[k], s1, col[k]);
  }
  // ... rows P1, Q1 of A: of the lane's own column and of its copy of the partner's (the pair-2 lanes rotate with it next)
  if (isA) {
    row_rot(col, P1, Q1, c1, s1);
    row_rot(pc, P1, Q1, c1, s1);
  }
  // rotation 2: columns P2, Q2, then rows P2, Q2 of A
  if (!pair1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) col[k] = is_p ? rot_lo(c2, col[k], s2, pc[k]) : rot_hi(c2, pc[k], s2, col[k]);
  }
  if (isA) row_rot(col, P2, Q2, c2, s2);
}


// Probe build only (-DTRI_STAMPS, tools/probes/stamps_tri.py): thread 0 of every workgroup appends [blockIdx, start, end (s_memrealtime),
// HW_ID | XCC_ID << 32, s_memtime at: start, phase 1 done, barrier passed, Jacobi + divide done, next projection done].
#ifdef TRI_STAMPS
__device__ unsigned long long tri_stamps[4096 * 12];
__device__ unsigned int tri_stamp_count;
#define TSTAMP_DECL unsigned long long tst_[12]; tst_[0] = 0
#define TSTAMP_REAL(i) tst_[i] = __builtin_amdgcn_s_memrealtime()
#define TSTAMP(i) tst_[i] = __builtin_amdgcn_s_memtime()
#define TSTAMP_FLUSH()                                                                                                    \
  do {                                                                                                                    \
    if (threadIdx.x == 0) {                                                                                               \
      const unsigned slot_ = atomicAdd(&tri_stamp_count, 1u);                                                             \
      if (slot_ < 4096) {                                                                                                 \
        unsigned hw_, xcc_;                                                                                               \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                                 \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                               \
        tst_[0] = blockIdx.x;                                                                                             \
        tst_[3] = ((unsigned long long)xcc_ << 32) | hw_;                                                                 \
        for (int i_ = 0; i_ < 12; ++i_) tri_stamps[slot_ * 12 + i_] = tst_[i_];                                           \
      }                                                                                                                   \
    }                                                                                                                     \
  } while (0)
extern "C" int mvg_tri_read_stamps(unsigned long long* host, int max_records, int reset) {
  unsigned n = 0;
  hipError_t e = hipMemcpyFromSymbol(&n, HIP_SYMBOL(tri_stamp_count), sizeof(n));
  if (e != hipSuccess) return -(int)e;
  if (n > 4096) n = 4096;
  if ((int)n > max_records) n = max_records;
  if (host && n) e = hipMemcpyFromSymbol(host, HIP_SYMBOL(tri_stamps), sizeof(unsigned long long) * 12 * n);
  if (e != hipSuccess) return -(int)e;
  if (reset) {
    const unsigned z = 0;
    e = hipMemcpyToSymbol(HIP_SYMBOL(tri_stamp_count), &z, sizeof(z));
    if (e != hipSuccess) return -(int)e;
  }
  return (int)n;
}
#else
#define TSTAMP_DECL
#define TSTAMP_REAL(i)
#define TSTAMP(i)
#define TSTAMP_FLUSH()
#endif

template <bool LANES>
__global__ __launch_bounds__(512) void triangulate_kernel(const float* __restrict__ r, const float* __restrict__ o,
                                                          const float* __restrict__ cams,
                                                          const uint8_t* __restrict__ valid,
                                                          const int* __restrict__ any_valid,
                                                          float* __restrict__ new_ref, float* __restrict__ ref2d,
                                                          float* __restrict__ proj2d, int V, int B, int NQ, int J,
                                                          LevelTable lv, float* __restrict__ r_next,
                                                          float* __restrict__ ref_lvl_next,
                                                          uint8_t* __restrict__ inside_next) {
  __shared__ double gram[10][TRI_PROBS][TRI_PAD];
  __shared__ float xnew[TRI_PROBS][3];
  TSTAMP_DECL;
  TSTAMP_REAL(1);
  TSTAMP(4);
  const int tid = threadIdx.x, pl = tid >> 3, sub = tid & 7;
  const int Lq = NQ * J;
  const long nprob = (long)B * Lq;
  {
    long idx = (long)blockIdx.x * TRI_PROBS + pl;
    const bool live = idx < nprob;
    if (!live) idx = nprob - 1;                     // keep the 8-lane groups converged for the DPP reductions
    const int q = (int)(idx % Lq), b = (int)(idx / Lq);
    const int i = q / J;
    bool ok = valid[b * NQ + i] != 0;
    if (!ok && any_valid[0] == 0 && b == 0 && i == 0) ok = true;   // dq_decoder.py:620-623

    // Everything this lane needs is REQUESTED first -- validity, and per view of the lane (views sub, sub + 8, ...; the first
    // one held in registers) the three pose outputs, the reference point and the camera record -- so that the kernel pays one
    // memory round trip, not a chain of them (logits -> softmax -> per-view loads): it runs on 240 wavefronts' latency.
    const long pair0 = ((long)min(sub, V - 1) * B + b) * Lq + q;
    const float* cam0 = cams + ((long)min(sub, V - 1) * B + b) * MVG_CAM_STRIDE;
    const float o0x = o[pair0 * 3], o0y = o[pair0 * 3 + 1], o0l = o[pair0 * 3 + 2];
    const float2 r0 = *reinterpret_cast<const float2*>(r + pair0 * 2);
    float camr[40];
#pragma unroll
    for (int k = 0; k < 40; k += 4) *reinterpret_cast<f32x4*>(&camr[k]) = *reinterpret_cast<const f32x4*>(cam0 + k);

    // softmax over views of the confidence logit (dq_decoder.py:706-707)
    float mx = sub < V ? o0l : -INFINITY;
    for (int v = sub + 8; v < V; v += 8) mx = fmaxf(mx, o[(((long)v * B + b) * Lq + q) * 3 + 2]);
    mx = max8(mx);
    float den = sub < V ? expf(o0l - mx) : 0.f;
    for (int v = sub + 8; v < V; v += 8) den += expf(o[(((long)v * B + b) * Lq + q) * 3 + 2] - mx);
    den = add8(den);

    double G[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) G[e] = 0.0;

    for (int v = sub; v < V; v += 8) {
      const long pair = ((long)v * B + b) * Lq + q;
      const bool first = v == sub;
      float cam[40];
      if (first) {
#pragma unroll
        for (int k = 0; k < 40; ++k) cam[k] = camr[k];
      } else {
        const float* cp = cams + ((long)v * B + b) * MVG_CAM_STRIDE;
#pragma unroll
        for (int k = 0; k < 40; k += 4) *reinterpret_cast<f32x4*>(&cam[k]) = *reinterpret_cast<const f32x4*>(cp + k);
      }
      const float imgw = cam[36], imgh = cam[37];
      const float rx = first ? r0.x : r[pair * 2], ry = first ? r0.y : r[pair * 2 + 1];
      const float dx = first ? o0x : o[pair * 3], dy = first ? o0y : o[pair * 3 + 1];
      const float conf = expf((first ? o0l : o[pair * 3 + 2]) - mx) / den;
      const float px = rx * imgw, py = ry * imgh;                                   // dq_decoder.py:699
      const float kx = (rx + dx / imgw) * imgw, ky = (ry + dy / imgh) * imgh;       // :679-685,696
      if (live) {
        const long oidx = (((long)b * V + v) * Lq + q) * 2;
        *reinterpret_cast<float2*>(ref2d + oidx) = ok ? make_float2(kx, ky) : make_float2(0.f, 0.f);
        *reinterpret_cast<float2*>(proj2d + oidx) = ok ? make_float2(px, py) : make_float2(0.f, 0.f);
      }
      // un-crop (dq_decoder.py:414-420)
      const float uo = cam[27] * kx + cam[28] * ky + cam[29];
      const float vo = cam[30] * kx + cam[31] * ky + cam[32];
      // undistort (dq_decoder.py:119-204)
      const float fx = cam[12], fy = cam[13], cx = cam[14], cy = cam[15];
      const float k1 = cam[16], k2 = cam[17], k3 = cam[18], p1 = cam[19], p2 = cam[20];
      const float x0 = uo * (1.f / fx) + (-cx / fx), y0 = vo * (1.f / fy) + (-cy / fy);
      float x = x0, y = y0;
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const float r2 = x * x + y * y;
        const float icd = 1.f / (1.f + ((k3 * r2 + k2) * r2 + k1) * r2);
        const float dX = 2.f * p1 * x * y + p2 * (r2 + 2.f * x * x);
        const float dY = p1 * (r2 + 2.f * y * y) + 2.f * p2 * x * y;
        x = (x0 - dX) * icd;
        y = (y0 - dY) * icd;
      }
      const float udx = fx * x + cx, udy = fy * y + cy;
      // P = K [R | -R T]  (dq_decoder.py:223-246)
      float RT[3][4];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        RT[a][0] = cam[3 * a]; RT[a][1] = cam[3 * a + 1]; RT[a][2] = cam[3 * a + 2];
        RT[a][3] = -(cam[3 * a] * cam[9] + cam[3 * a + 1] * cam[10] + cam[3 * a + 2] * cam[11]);
      }
      float a1[4], a2[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float P0 = fx * RT[0][c] + cx * RT[2][c];
        const float P1 = fy * RT[1][c] + cy * RT[2][c];
        const float P2 = RT[2][c];
        a1[c] = (P2 * udx - P0) * conf;                                             // multiview.py:196-202
        a2[c] = (P2 * udy - P1) * conf;
      }
      int e = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = a; c < 4; ++c, ++e) G[e] += (double)a1[a] * (double)a1[c] + (double)a2[a] * (double)a2[c];
    }
#pragma unroll
    for (int e = 0; e < 10; ++e) gram[e][pl][sub] = G[e];
  }
  TSTAMP(5);
  __syncthreads();
  TSTAMP(6);
  if constexpr (LANES) {
    // ---- phase 2, lane-parallel: see jacobi_round_lanes
    const long idx = min((long)blockIdx.x * TRI_PROBS + pl, nprob - 1);
    const bool live = (long)blockIdx.x * TRI_PROBS + pl < nprob;
    const int q = (int)(idx % Lq), b = (int)(idx / Lq);
    const int i = q / J;
    bool ok = valid[b * NQ + i] != 0;
    if (!ok && any_valid[0] == 0 && b == 0 && i == 0) ok = true;
    const bool isA = sub < 4;
    const int jj = sub & 3;
    double col[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lo = min(k, jj), hi = max(k, jj);
      const int e = lo * 4 - (lo * (lo - 1)) / 2 + (hi - lo);           // entry (lo, hi) of the upper triangle, as phase 1 numbered it
      double g = 0.0;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) g += gram[e][pl][s8];                // Gram matrix over all views
      col[k] = isA ? g : (k == jj ? 1.0 : 0.0);
    }
    int act = 1;
    for (int sweep = 0; sweep < 12; ++sweep) {
      if (act) {
        const double dj = jj == 0 ? col[0] : (jj == 1 ? col[1] : (jj == 2 ? col[2] : col[3]));
        const double g01 = dpp_f64<0x55>(col[0]), g02 = dpp_f64<0xAA>(col[0]), g03 = dpp_f64<0xFF>(col[0]);
        const double g12 = dpp_f64<0xAA>(col[1]), g13 = dpp_f64<0xFF>(col[1]), g23 = dpp_f64<0xFF>(col[2]);
        const double d0 = dpp_f64<0x00>(dj), d1 = dpp_f64<0x55>(dj), d2 = dpp_f64<0xAA>(dj), d3 = dpp_f64<0xFF>(dj);
        const double off = fabs(g01) + fabs(g02) + fabs(g03) + fabs(g12) + fabs(g13) + fabs(g23);
        const double lg = fmax(fmax(fabs(d0), fabs(d1)), fmax(fabs(d2), fabs(d3)));
        int go = (off <= 2e-16 * lg) ? 0 : 1;    // off-diagonals at the fp64 rounding floor of the matrix: converged
        const int gov = __builtin_amdgcn_update_dpp(0, go, 0x114, 0xf, 0xf, false);
        act = isA ? go : gov;
      }
      if (act) {
        jacobi_round_lanes<1>(col, jj, isA);
        jacobi_round_lanes<2>(col, jj, isA);
        jacobi_round_lanes<3>(col, jj, isA);
      }
      if (!__any(act)) break;
    }
    // eigenvector of the smallest eigenvalue (the first one on ties): column cstar of V, i.e. lane 4 + cstar
    int cstar = 0;
    {
      const double dj = jj == 0 ? col[0] : (jj == 1 ? col[1] : (jj == 2 ? col[2] : col[3]));
      const double d0 = dpp_f64<0x00>(dj), d1 = dpp_f64<0x55>(dj), d2 = dpp_f64<0xAA>(dj), d3 = dpp_f64<0xFF>(dj);
      double best = d0;
      if (d1 < best) { best = d1; cstar = 1; }
      if (d2 < best) { best = d2; cstar = 2; }
      if (d3 < best) { best = d3; cstar = 3; }
      const int cv = __builtin_amdgcn_update_dpp(0, cstar, 0x114, 0xf, 0xf, false);
      if (!isA) cstar = cv;
    }
    if (!isA && jj == cstar && live) {
      const float X0 = ok ? (float)(col[0] / col[3]) : 0.f;                             // multiview.py:220-221
      const float X1 = ok ? (float)(col[1] / col[3]) : 0.f;
      const float X2 = ok ? (float)(col[2] / col[3]) : 0.f;
      float* nr = new_ref + ((long)b * Lq + q) * 3;
      nr[0] = X0;
      nr[1] = X1;
      nr[2] = X2;
      xnew[pl][0] = X0;
      xnew[pl][1] = X1;
      xnew[pl][2] = X2;
    }
  } else
  // ---- phase 2: one lane per problem
  if (tid < TRI_PROBS && (long)blockIdx.x * TRI_PROBS + tid < nprob) {
  const long idx = (long)blockIdx.x * TRI_PROBS + tid;
  const int q = (int)(idx % Lq), b = (int)(idx / Lq);
  const int i = q / J;
  bool ok = valid[b * NQ + i] != 0;
  if (!ok && any_valid[0] == 0 && b == 0 && i == 0) ok = true;
  double G[4][4];
  {
    int e = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = a; c < 4; ++c, ++e) {
        double g = 0.0;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) g += gram[e][tid][s8];     // Gram matrix over all views
        G[a][c] = g;
        G[c][a] = g;
      }
  }

  double ev[4];
  const bool need_jacobi = !null_vector_invit(G, ev, ok);
#ifdef TRI_STAMPS
  tst_[9] = __popcll(__ballot(need_jacobi));      // lanes of the solving wavefront that fall back to the Jacobi
  {   // ... of them with unsafe pivots (the rest did not settle)
    Ldl4 f_;
    tst_[11] = __popcll(__ballot(need_jacobi && !ldl4_factor<true>(G, 0.0, f_)));
  }
  TSTAMP(10);
#endif
  if (need_jacobi) {
    double Vm[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) Vm[a][c] = (a == c) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
      const double off = fabs(G[0][1]) + fabs(G[0][2]) + fabs(G[0][3]) + fabs(G[1][2]) + fabs(G[1][3]) + fabs(G[2][3]);
      const double lg = fmax(fmax(fabs(G[0][0]), fabs(G[1][1])), fmax(fabs(G[2][2]), fabs(G[3][3])));
      if (off <= 2e-16 * lg) break;     // off-diagonals at the fp64 rounding floor of the matrix: converged
      jacobi_rotate2(G, Vm, 0, 1, 2, 3);
      jacobi_rotate2(G, Vm, 0, 2, 1, 3);
      jacobi_rotate2(G, Vm, 0, 3, 1, 2);
    }
    double best = G[0][0];
    ev[0] = Vm[0][0]; ev[1] = Vm[1][0]; ev[2] = Vm[2][0]; ev[3] = Vm[3][0];
#pragma unroll
    for (int c = 1; c < 4; ++c) {
      if (G[c][c] < best) {
        best = G[c][c];
        ev[0] = Vm[0][c]; ev[1] = Vm[1][c]; ev[2] = Vm[2][c]; ev[3] = Vm[3][c];
      }
    }
  }
  const float X0 = ok ? (float)(ev[0] / ev[3]) : 0.f;                               // multiview.py:220-221
  const float X1 = ok ? (float)(ev[1] / ev[3]) : 0.f;
  const float X2 = ok ? (float)(ev[2] / ev[3]) : 0.f;
  float* nr = new_ref + ((long)b * Lq + q) * 3;
  nr[0] = X0;
  nr[1] = X1;
  nr[2] = X2;
  xnew[tid][0] = X0;
  xnew[tid][1] = X1;
  xnew[tid][2] = X2;
  }
  TSTAMP(7);
  if (!r_next) {
    TSTAMP_REAL(2);
    TSTAMP_FLUSH();
    return;                             // uniform: the caller does not want the next layer's projections
  }
  // ---- phase 3: the NEXT layer's projection of the new points (project_kernel's arithmetic on new_ref, which is what the
  //      next layer receives as reference_points: zeros for queries that did not pass, dq_decoder.py:1013-1029), by the
  //      8 lanes of each problem: lane `sub` takes views sub, sub + 8, ...
  __syncthreads();
  {
    const long idx = (long)blockIdx.x * TRI_PROBS + pl;
    if (idx < nprob) {
      const int q = (int)(idx % Lq), b = (int)(idx / Lq);
      const float x0 = xnew[pl][0], x1 = xnew[pl][1], x2 = xnew[pl][2];
      for (int v = sub; v < V; v += 8) {
        const long n = (long)v * B + b;
        project_point(x0, x1, x2, cams + n * MVG_CAM_STRIDE, lv, r_next, ref_lvl_next, inside_next, n * Lq + q);
      }
    }
  }
  TSTAMP(8);
  TSTAMP_REAL(2);
  TSTAMP_FLUSH();
}

// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// bin_pairs_kernel -- processing order of the (image, query) pairs for the sampling kernel.
// The sampling kernel's time is set by L1 misses of its gathers; queries arrive in person-major order, which is
// random in the image.  One workgroup per image counting-sorts its Lq pairs by the Morton code of the level-0
// cell block (2^shift x 2^shift cells) their reference point falls into; pairs outside the image (masked by
// the consumer, dq_decoder.py:585-586) get the last key.  Consecutive slots of `order` are then neighbours in the
// feature maps, so the 64 pairs of a sampling workgroup share their pair lines.  The order inside a bin is
// whatever the LDS atomics produce: `order` only decides WHERE a pair is computed, never its result.
constexpr int BIN_BITS = 6, BIN_KEYS = 1 << (2 * BIN_BITS);   // 64 x 64 cell blocks + 1 key for "outside"

__device__ __forceinline__ unsigned spread1(unsigned v) {        // 6 bits -> every second bit
  v &= 0x3fu;
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

template <int KPT>   // keys per thread: Lq <= 1024 * KPT
__global__ __launch_bounds__(1024) void bin_pairs_kernel(const float* __restrict__ ref_lvl,
                                                         const uint8_t* __restrict__ inside, int* __restrict__ order,
                                                         int Lq, int L, int W0, int H0, int shift) {
  __shared__ int hist[BIN_KEYS + 1];
  __shared__ int wave_tot[16];
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i <= BIN_KEYS; i += 1024) hist[i] = 0;
  // all loads of the thread's pairs are issued before the first key is used (one memory round trip)
  float rx[KPT], ry[KPT];
  uint8_t in[KPT];
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const long pair = (long)n * Lq + min(tid + 1024 * k, Lq - 1);
    const float2 rr = *reinterpret_cast<const float2*>(ref_lvl + pair * L * 2);     // level-0 reference point
    rx[k] = rr.x;
    ry[k] = rr.y;
    in[k] = inside ? inside[pair] : (uint8_t)1;
  }
  __syncthreads();
  int key[KPT];
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const int cx = min(max((int)(fminf(fmaxf(rx[k], 0.f), 1.f) * (float)W0), 0), W0 - 1) >> shift;
    const int cy = min(max((int)(fminf(fmaxf(ry[k], 0.f), 1.f) * (float)H0), 0), H0 - 1) >> shift;
    key[k] = in[k] ? (int)(spread1((unsigned)cx) | (spread1((unsigned)cy) << 1)) : BIN_KEYS;
    const bool live = tid + 1024 * k < Lq;
    // the "outside" key is shared by a large part of the pairs: one LDS atomic per wavefront, not per lane
    const unsigned long long outm = __ballot(live && key[k] == BIN_KEYS);
    if (live && key[k] != BIN_KEYS) atomicAdd(&hist[key[k]], 1);
    if (outm && (tid & 63) == 0) atomicAdd(&hist[BIN_KEYS], __popcll(outm));
  }
  __syncthreads();
  // exclusive scan of the 4096 keys: 4 per thread, wave shuffles, then the 16 wave totals
  const int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
  const int mine = c0 + c1 + c2 + c3;
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if ((tid & 63) >= d) incl += up;
  }
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int t = wave_tot[w];
    base += (w < (tid >> 6)) ? t : 0;
    total += t;
  }
  __syncthreads();
  const int ex = base + incl - mine;
  hist[4 * tid] = ex;
  hist[4 * tid + 1] = ex + c0;
  hist[4 * tid + 2] = ex + c0 + c1;
  hist[4 * tid + 3] = ex + c0 + c1 + c2;
  if (tid == 0) hist[BIN_KEYS] = total;          // the "outside" pairs go last
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const bool live = tid + 1024 * k < Lq;
    const bool out = live && key[k] == BIN_KEYS;
    const unsigned long long outm = __ballot(out);
    int pos = 0;
    if (live && !out) pos = atomicAdd(&hist[key[k]], 1);
    int obase = 0;
    if (outm && (tid & 63) == 0) obase = atomicAdd(&hist[BIN_KEYS], __popcll(outm));
    obase = __shfl(obase, 0, 64);
    if (out) pos = obase + __popcll(outm & ((1ull << (tid & 63)) - 1ull));
    if (live) order[(long)n * Lq + pos] = n * Lq + tid + 1024 * k;
  }
}

// ---- the same counting sort spread over BIN_PARTS workgroups per image (large Lq): the single-workgroup kernel is
// bound by the LDS-atomic throughput of ONE CU (2 x Lq returning atomics on its LDS unit: 16.6 us at Lq = 15 360).
//   bin_count_kernel  : part p counts the keys of its slice of the pairs in LDS and writes the 4097 counters and its
//                       16-bit keys to the workspace;
//   bin_scatter_kernel: every part re-derives the global offsets from all parts' counters (key-major, then part:
//                       offset(k, p) = sum_{k' < k} total(k') + sum_{p' < p} count(p', k)) and scatters its slice.
// No workgroup waits for another one: the dependency is the kernel boundary.
int g_bin_multi = 1;                            // tuning knob "bin_multi": 0 = always the single-workgroup binning kernel
constexpr int BIN_PARTS = 8;                    // 4 / 16 parts per image measured: +0.7 / +2.8 % per forward
constexpr int BIN_MULTI_MIN = 8192;             // pairs per image from which the multi-workgroup variant is used
constexpr int BIN_CNT_STRIDE = BIN_KEYS + 4;        // counters per (image, part), padded to 16 bytes

template <int KPT>   // keys per thread: slice length <= 1024 * KPT
__global__ __launch_bounds__(1024) void bin_count_kernel(const float* __restrict__ ref_lvl,
                                                         const uint8_t* __restrict__ inside, unsigned* __restrict__ cnt,
                                                         unsigned short* __restrict__ keys_out, int Lq, int Lp, int L,
                                                         int W0, int H0, int shift) {
  __shared__ int hist[BIN_KEYS + 4];
  const int n = blockIdx.x / BIN_PARTS, part = blockIdx.x % BIN_PARTS, tid = threadIdx.x;
  const int q0 = part * Lp, nq = max(0, min(Lp, Lq - q0));
  for (int i = tid; i < BIN_KEYS + 4; i += 1024) hist[i] = 0;
  float rx[KPT], ry[KPT];
  uint8_t in[KPT];
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const long pair = (long)n * Lq + min(q0 + tid + 1024 * k, Lq - 1);
    const float2 rr = *reinterpret_cast<const float2*>(ref_lvl + pair * L * 2);
    rx[k] = rr.x;
    ry[k] = rr.y;
    in[k] = inside ? inside[pair] : (uint8_t)1;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const int cx = min(max((int)(fminf(fmaxf(rx[k], 0.f), 1.f) * (float)W0), 0), W0 - 1) >> shift;
    const int cy = min(max((int)(fminf(fmaxf(ry[k], 0.f), 1.f) * (float)H0), 0), H0 - 1) >> shift;
    const int key = in[k] ? (int)(spread1((unsigned)cx) | (spread1((unsigned)cy) << 1)) : BIN_KEYS;
    const bool live = tid + 1024 * k < nq;
    const unsigned long long outm = __ballot(live && key == BIN_KEYS);
    if (live && key != BIN_KEYS) atomicAdd(&hist[key], 1);
    if (outm && (tid & 63) == 0) atomicAdd(&hist[BIN_KEYS], __popcll(outm));
    if (live) keys_out[(long)n * Lq + q0 + tid + 1024 * k] = (unsigned short)key;
  }
  __syncthreads();
  unsigned* dst = cnt + (long)blockIdx.x * BIN_CNT_STRIDE;
  *reinterpret_cast<uint4*>(dst + 4 * tid) = *reinterpret_cast<const uint4*>(&hist[4 * tid]);
  if (tid == 0) *reinterpret_cast<uint4*>(dst + BIN_KEYS) = *reinterpret_cast<const uint4*>(&hist[BIN_KEYS]);
}

// Layout of `order` (this variant): the in-image pairs of ALL images first (image by image, Morton order inside an image), then
// the pairs outside their image (image by image).  With every image's outside pairs behind its own in-image pairs, a consumer that
// walks the order in workgroups of 64 slots met a run of do-nothing workgroups per image: they took their share of the dispatch
// slots and the chip ran 30-40 % under-occupied for ~10 us five times per sampler launch (s_memrealtime stamps per workgroup,
// tools/probes/stamps_gsamp.py: sampler 127 -> 116 us); chain A's computing tiles now lead its launch by themselves.  The offsets
// of the other images come from their parts' "outside" counters (N_img x BIN_PARTS words, one per thread).
template <int KPT>
__global__ __launch_bounds__(1024) void bin_scatter_kernel(const unsigned* __restrict__ cnt,
                                                           const unsigned short* __restrict__ keys_in,
                                                           int* __restrict__ order, int Lq, int Lp) {
  __shared__ int hist[BIN_KEYS + 4];
  __shared__ int wave_tot[16];
  __shared__ int out_sums[2];           // outside pairs of the images before this one | of all images
  const int n = blockIdx.x / BIN_PARTS, part = blockIdx.x % BIN_PARTS, tid = threadIdx.x;
  const int n_img = gridDim.x / BIN_PARTS;
  // (the first wavefront fetches the other images' "outside" counters here and adds them up behind the loads below: no extra barrier,
  // no round trip in front of the key loads)
  constexpr int OVN = 16;               // x 64 lanes = 1024 (image, part) counters; more images: the loop below continues
  int ov[OVN];
  if (tid < 64) {
#pragma unroll
    for (int j = 0; j < OVN; ++j) {
      const int i = tid + 64 * j;
      ov[j] = i < n_img * BIN_PARTS ? (int)cnt[(long)i * BIN_CNT_STRIDE + BIN_KEYS] : 0;
    }
  }
  const int q0 = part * Lp, nq = max(0, min(Lp, Lq - q0));
  int key[KPT];
#pragma unroll
  for (int k = 0; k < KPT; ++k) key[k] = keys_in[(long)n * Lq + min(q0 + tid + 1024 * k, Lq - 1)];
  // totals over the parts and this part's offset inside every key, 4 keys per thread
  const unsigned* c0 = cnt + (long)n * BIN_PARTS * BIN_CNT_STRIDE;
  int tot[4] = {0, 0, 0, 0}, before[4] = {0, 0, 0, 0};
  int out_before = 0;
#pragma unroll
  for (int p = 0; p < BIN_PARTS; ++p) {
    const uint4 c = *reinterpret_cast<const uint4*>(c0 + (long)p * BIN_CNT_STRIDE + 4 * tid);
    const int cc[4] = {(int)c.x, (int)c.y, (int)c.z, (int)c.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tot[i] += cc[i];
      before[i] += (p < part) ? cc[i] : 0;
    }
    if (tid == 0) out_before += (p < part) ? (int)c0[(long)p * BIN_CNT_STRIDE + BIN_KEYS] : 0;
  }
  const int mine = tot[0] + tot[1] + tot[2] + tot[3];
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if ((tid & 63) >= d) incl += up;
  }
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  if (tid < 64) {
    int ob = 0, oa = 0;
#pragma unroll
    for (int j = 0; j < OVN; ++j) {
      oa += ov[j];
      ob += ((tid + 64 * j) / BIN_PARTS < n) ? ov[j] : 0;
    }
    for (int i = tid + 64 * OVN; i < n_img * BIN_PARTS; i += 64) {      // (more than 128 images)
      const int v = (int)cnt[(long)i * BIN_CNT_STRIDE + BIN_KEYS];
      oa += v;
      ob += (i / BIN_PARTS < n) ? v : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ob += __shfl_xor(ob, off, 64);
      oa += __shfl_xor(oa, off, 64);
    }
    if (tid == 0) {
      out_sums[0] = ob;
      out_sums[1] = oa;
    }
  }
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int t = wave_tot[w];
    base += (w < (tid >> 6)) ? t : 0;
    total += t;
  }
  const int ex = base + incl - mine;
  hist[4 * tid] = ex + before[0];
  hist[4 * tid + 1] = ex + tot[0] + before[1];
  hist[4 * tid + 2] = ex + tot[0] + tot[1] + before[2];
  hist[4 * tid + 3] = ex + tot[0] + tot[1] + tot[2] + before[3];
  if (tid == 0) hist[BIN_KEYS] = total + out_before;          // the "outside" pairs go last
  __syncthreads();
  // image-local position -> position in the launch-wide order: in-image pairs behind those of the images before, outside pairs
  // behind ALL in-image pairs and the outside pairs of the images before
  const int out_prev = out_sums[0], out_all = out_sums[1];
  const long in_base = (long)n * Lq - out_prev;                              // in-image pairs of images < n
  const long out_base = (long)n_img * Lq - out_all + out_prev - total;       // (+ local position, which starts at `total`)
#pragma unroll
  for (int k = 0; k < KPT; ++k) {
    const bool live = tid + 1024 * k < nq;
    const bool out = live && key[k] == BIN_KEYS;
    const unsigned long long outm = __ballot(out);
    int pos = 0;
    if (live && !out) pos = atomicAdd(&hist[key[k]], 1);
    int obase = 0;
    if (outm && (tid & 63) == 0) obase = atomicAdd(&hist[BIN_KEYS], __popcll(outm));
    obase = __shfl(obase, 0, 64);
    if (out) pos = obase + __popcll(outm & ((1ull << (tid & 63)) - 1ull));
    if (live) order[(out ? out_base : in_base) + pos] = n * Lq + q0 + tid + 1024 * k;
  }
}

// ---- batched eigen-decomposition of symmetric 4x4 matrices (fp64, cyclic Jacobi as in triangulate_kernel): the
// differentiable DLT of the training path (geometry_torch.dlt) takes the eigenvector of the smallest eigenvalue of the
// Gram matrix A^T A and needs all four pairs for its backward (rocSOLVER's batched SVD of the (2V, 4) row matrices was
// 70 % of a training step).  One lane per matrix; evecs holds the eigenvectors as COLUMNS, evals in no particular order.
__global__ __launch_bounds__(256) void sym4_eigh_kernel(const double* __restrict__ Gin, double* __restrict__ evals,
                                                        double* __restrict__ evecs, long n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  double G[4][4], Vm[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      G[a][c] = 0.5 * (Gin[idx * 16 + a * 4 + c] + Gin[idx * 16 + c * 4 + a]);
      Vm[a][c] = (a == c) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(G[0][1]) + fabs(G[0][2]) + fabs(G[0][3]) + fabs(G[1][2]) + fabs(G[1][3]) + fabs(G[2][3]);
    const double lg = fmax(fmax(fabs(G[0][0]), fabs(G[1][1])), fmax(fabs(G[2][2]), fabs(G[3][3])));
    if (off <= 2e-16 * lg) break;
    jacobi_rotate2(G, Vm, 0, 1, 2, 3);
    jacobi_rotate2(G, Vm, 0, 2, 1, 3);
    jacobi_rotate2(G, Vm, 0, 3, 1, 2);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    evals[idx * 4 + a] = G[a][a];
#pragma unroll
    for (int c = 0; c < 4; ++c) evecs[idx * 16 + a * 4 + c] = Vm[a][c];
  }
}

// ------------------------------------------------------------------------------------------
// Training path: un-crop + undistortion of the refined 2D points (dq_decoder.py:414-420, 119-204) as ONE
// launch that also returns the 2 x 2 Jacobian d(ud) / d(ref2d) of every point -- forward-mode through the inverse crop affine and the
// 5 fixed-point iterations -- so that the backward is one small product per point instead of torch autograd through ~80
// elementwise kernels forward and ~160 backward per decoder layer.
__global__ __launch_bounds__(256) void uncrop_undistort_jac_kernel(const float* __restrict__ ref2d, const float* __restrict__ cams,
                                                                   float* __restrict__ ud, float* __restrict__ jac, int V, int B,
                                                                   int Lq) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * V * Lq;
  if (i >= total) return;
  const int v = (int)((i / Lq) % V), b = (int)(i / ((long)Lq * V));
  const float* cam = cams + ((long)v * B + b) * MVG_CAM_STRIDE;
  const float2 kp = *reinterpret_cast<const float2*>(ref2d + i * 2);
  const float a00 = cam[27], a01 = cam[28], a10 = cam[30], a11 = cam[31];
  const float uo = a00 * kp.x + a01 * kp.y + cam[29];
  const float vo = a10 * kp.x + a11 * kp.y + cam[32];
  const float fx = cam[12], fy = cam[13], cx = cam[14], cy = cam[15];
  const float k1 = cam[16], k2 = cam[17], k3 = cam[18], p1 = cam[19], p2 = cam[20];
  const float x0 = uo * (1.f / fx) + (-cx / fx), y0 = vo * (1.f / fy) + (-cy / fy);
  float x = x0, y = y0;
  // tangents of (x, y) with respect to (x0, y0)
  float xa = 1.f, xb = 0.f, ya = 0.f, yb = 1.f;
#pragma unroll
  for (int it = 0; it < 5; ++it) {
    const float r2 = x * x + y * y;
    const float q = 1.f + ((k3 * r2 + k2) * r2 + k1) * r2;
    const float icd = 1.f / q;
    const float dq = (3.f * k3 * r2 + 2.f * k2) * r2 + k1;          // dq / dr2
    const float dX = 2.f * p1 * x * y + p2 * (r2 + 2.f * x * x);
    const float dY = p1 * (r2 + 2.f * y * y) + 2.f * p2 * x * y;
    const float dXx = 2.f * p1 * y + 6.f * p2 * x, dXy = 2.f * p1 * x + 2.f * p2 * y;
    const float dYx = 2.f * p1 * x + 2.f * p2 * y, dYy = 6.f * p1 * y + 2.f * p2 * x;
    const float nx = x0 - dX, ny = y0 - dY;
    const float c = -dq * icd * icd;                                   // d(icd) = c * d(r2)
    // direction a (d / dx0), direction b (d / dy0)
    const float r2a = 2.f * (x * xa + y * ya), r2b = 2.f * (x * xb + y * yb);
    const float nxa = (1.f - (dXx * xa + dXy * ya)) * icd + nx * c * r2a;
    const float nxb = (0.f - (dXx * xb + dXy * yb)) * icd + nx * c * r2b;
    const float nya = (0.f - (dYx * xa + dYy * ya)) * icd + ny * c * r2a;
    const float nyb = (1.f - (dYx * xb + dYy * yb)) * icd + ny * c * r2b;
    x = nx * icd;
    y = ny * icd;
    xa = nxa; xb = nxb; ya = nya; yb = nyb;
  }
  *reinterpret_cast<float2*>(ud + i * 2) = make_float2(fx * x + cx, fy * y + cy);
  // d(ud) / d(uo, vo) = diag(fx, fy) T diag(1 / fx, 1 / fy); times the inverse crop affine's 2 x 2 part
  const float t00 = xa, t01 = xb * fx / fy, t10 = ya * fy / fx, t11 = yb;
  *reinterpret_cast<f32x4*>(jac + i * 4) = f32x4{t00 * a00 + t01 * a10, t00 * a01 + t01 * a11, t10 * a00 + t11 * a10, t10 * a01 + t11 * a11};
}

// ------------------------------------------------------------------------------------------
// Training path: the differentiable triangulation (multiview.py:170-228 under autograd; geometry_torch.dlt as torch ops) as one launch
// forward and one backward over the DENSE (B, Lq) token grid -- tokens of unmatched queries are skipped (zeros out, zero gradients), so
// the caller needs no nonzero / index / index_put round trip (a host sync and ~30 sort launches in their backward).  Per token: the
// 2V x 4 row matrix A (rows conf * (P[2] * u - P[0|1])), its Gram matrix and the Jacobi eigen-solve, all in fp64,
// X = v0[:3] / v0[3] (v0: eigenvector of the smallest eigenvalue).  The backward recomputes the decomposition and applies
//   dL/dG = sym(m v0^T),  m = sum_{i != 0} v_i (v_i^T g) / (l0 - l_i);   dA = A (dG + dG^T);   A's rows -> (u, conf).
constexpr int DLT_MAX_VIEWS = 32;

struct DltEig {
  double w[4], Vm[4][4];
  int k;
};

__device__ __forceinline__ void dlt_row(const float* __restrict__ P, const float u, const int c, double (&row)[4]) {
  // P: one view's (3, 4) projection matrix; row = P[2] * u - P[c].  In fp64: the torch form rounds the rows to fp32, which costs its
  // confidence gradients (sums that cancel to 1e-3 of their terms) three digits against the SVD's autograd in fp64
#pragma unroll
  for (int e = 0; e < 4; ++e) row[e] = __builtin_fma((double)P[8 + e], (double)u, -(double)P[4 * c + e]);
}

__device__ __forceinline__ void dlt_decompose(const float* __restrict__ ud, const float* __restrict__ conf, const float* __restrict__ Pm,
                                              const int b, const long t, const int V, const long Lq, DltEig& E) {
  double G[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      G[a][c] = 0.0;
      E.Vm[a][c] = (a == c) ? 1.0 : 0.0;
    }
  for (int v = 0; v < V; ++v) {
    const long pv = ((long)b * V + v) * Lq + t;
    const float2 u = *reinterpret_cast<const float2*>(ud + pv * 2);
    const float cf = conf[pv];
    const float* P = Pm + ((long)b * V + v) * 12;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double row[4];
      dlt_row(P, c ? u.y : u.x, c, row);
      double rd[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) rd[e] = row[e] * (double)cf;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = a; e < 4; ++e) G[a][e] = __builtin_fma(rd[a], rd[e], G[a][e]);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < a; ++e) G[a][e] = G[e][a];
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(G[0][1]) + fabs(G[0][2]) + fabs(G[0][3]) + fabs(G[1][2]) + fabs(G[1][3]) + fabs(G[2][3]);
    const double lg = fmax(fmax(fabs(G[0][0]), fabs(G[1][1])), fmax(fabs(G[2][2]), fabs(G[3][3])));
    if (off <= 2e-16 * lg) break;
    jacobi_rotate2(G, E.Vm, 0, 1, 2, 3);
    jacobi_rotate2(G, E.Vm, 0, 2, 1, 3);
    jacobi_rotate2(G, E.Vm, 0, 3, 1, 2);
  }
  E.k = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    E.w[a] = G[a][a];
    if (a > 0 && E.w[a] < E.w[E.k]) E.k = a;
  }
}

__device__ __forceinline__ double pick4(const double (&x)[4], const int k) {
  return k == 0 ? x[0] : k == 1 ? x[1] : k == 2 ? x[2] : x[3];
}

__global__ __launch_bounds__(64) void dlt_fwd_kernel(const float* __restrict__ ud, const float* __restrict__ conf,
                                                     const float* __restrict__ Pm, const uint8_t* __restrict__ valid,
                                                     float* __restrict__ X, int V, int B, long Lq, int J) {
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  if (i >= (long)B * Lq) return;
  const int b = (int)(i / Lq);
  const long t = i - (long)b * Lq;
  float* out = X + i * 3;
  if (!valid[(long)b * (Lq / J) + t / J]) {
    out[0] = out[1] = out[2] = 0.f;
    return;
  }
  DltEig E;
  dlt_decompose(ud, conf, Pm, b, t, V, Lq, E);
  float xh[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) xh[a] = (float)pick4(E.Vm[a], E.k);
#pragma unroll
  for (int a = 0; a < 3; ++a) out[a] = xh[a] / xh[3];
}

__global__ __launch_bounds__(64) void dlt_bwd_kernel(const float* __restrict__ ud, const float* __restrict__ conf,
                                                     const float* __restrict__ Pm, const uint8_t* __restrict__ valid,
                                                     const float* __restrict__ gX, float* __restrict__ g_ud, float* __restrict__ g_conf,
                                                     int V, int B, long Lq, int J) {
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  if (i >= (long)B * Lq) return;
  const int b = (int)(i / Lq);
  const long t = i - (long)b * Lq;
  if (!valid[(long)b * (Lq / J) + t / J]) {
    for (int v = 0; v < V; ++v) {
      const long pv = ((long)b * V + v) * Lq + t;
      *reinterpret_cast<float2*>(g_ud + pv * 2) = make_float2(0.f, 0.f);
      g_conf[pv] = 0.f;
    }
    return;
  }
  DltEig E;
  dlt_decompose(ud, conf, Pm, b, t, V, Lq, E);
  double v0[4], g[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) v0[a] = pick4(E.Vm[a], E.k);
  {
    // X = xh[:3] / xh[3] in fp32 (xh = v0 rounded to fp32): its gradient with respect to xh
    float xh[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) xh[a] = (float)v0[a];
    const double iw = 1.0 / (double)xh[3];
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      g[a] = (double)gX[i * 3 + a] * iw;
      s += (double)gX[i * 3 + a] * (double)xh[a];
    }
    g[3] = -s * iw * iw;
  }
  const double l0 = pick4(E.w, E.k);
  const double scale = fmax(fmax(fmax(fabs(E.w[0]), fabs(E.w[1])), fmax(fabs(E.w[2]), fabs(E.w[3]))), 1e-300);
  double m[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double den = l0 - E.w[e];
    const bool safe = fabs(den) > 1e-14 * scale && e != E.k;
    const double dot = E.Vm[0][e] * g[0] + E.Vm[1][e] * g[1] + E.Vm[2][e] * g[2] + E.Vm[3][e] * g[3];
    const double coef = safe ? dot / den : 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) m[a] = __builtin_fma(E.Vm[a][e], coef, m[a]);
  }
  for (int v = 0; v < V; ++v) {
    const long pv = ((long)b * V + v) * Lq + t;
    const float2 u = *reinterpret_cast<const float2*>(ud + pv * 2);
    const float cf = conf[pv];
    const float* P = Pm + ((long)b * V + v) * 12;
    double gu[2], gc = 0.0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double row[4];
      dlt_row(P, c ? u.y : u.x, c, row);
      // a = conf * row;  da = (a . m) v0 + (a . v0) m
      double am = 0.0, av = 0.0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double a = row[e] * (double)cf;
        am = __builtin_fma(a, m[e], am);
        av = __builtin_fma(a, v0[e], av);
      }
      double d_row = 0.0, d_p2 = 0.0;         // da . row (-> conf), da . P[2] (-> u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double da = am * v0[e] + av * m[e];
        d_row = __builtin_fma(da, row[e], d_row);
        d_p2 = __builtin_fma(da, (double)P[8 + e], d_p2);
      }
      gc += d_row;
      gu[c] = (double)cf * d_p2;
    }
    *reinterpret_cast<float2*>(g_ud + pv * 2) = make_float2((float)gu[0], (float)gu[1]);
    g_conf[pv] = (float)gc;
  }
}

extern "C" {


int mvg_pack_pyramid(const float* const* src_nchw_host, void* feat, int dtype, int N_img, int C, const int64_t* shapes_host,
                     const int64_t* starts_host, int L, int S, void* stream) {
  if (!src_nchw_host || !feat || !shapes_host || !starts_host || N_img <= 0 || C <= 0 || L < 1 || L > MVG_MAX_LEVELS)
    return MVG_E_BADARG;
  PackLevels pl;
  pl.L = L;
  int t = 0;
  for (int l = 0; l < L; ++l) {
    const long hw = (long)shapes_host[2 * l] * shapes_host[2 * l + 1];
    if (!src_nchw_host[l] || hw <= 0 || starts_host[l] < 0 || starts_host[l] + hw > S) return MVG_E_BADARG;
    pl.src[l] = src_nchw_host[l];
    pl.hw[l] = (int)hw;
    pl.start[l] = (int)starts_host[l];
    pl.tile0[l] = t;
    t += (int)((hw + 63) / 64);
  }
  pl.tile0[L] = t;
  dim3 grid(t, (C + 63) / 64, N_img);
  if (dtype == MVG_F32)
    hipLaunchKernelGGL((pack_pyramid_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, pl, (float*)feat, C, S);
  else if (dtype == MVG_BF16)
    hipLaunchKernelGGL((pack_pyramid_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, pl, (bf16_t*)feat, C, S);
  else
    return MVG_E_BADARG;
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_project(const float* X, const float* cams, const int64_t* shapes_host, int L, float* r, float* ref_lvl,
                uint8_t* inside, int V, int B, int Lq, void* stream) {
  if (!X || !cams || !shapes_host || !r || !ref_lvl || !inside || V <= 0 || B <= 0 || Lq < 0) return MVG_E_BADARG;
  LevelTable lv;
  int64_t zeros[MVG_MAX_LEVELS] = {0};
  int e = mvg_fill_levels(&lv, shapes_host, zeros, L);
  if (e) return e;
  const long total = (long)V * B * Lq;
  if (total == 0) return 0;
  hipLaunchKernelGGL(project_kernel, dim3(mvg_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, X, cams, lv, r,
                     ref_lvl, inside, B, Lq, total);
  MVG_LAUNCH_CHECK();
  return 0;
}

size_t mvg_bin_pairs_workspace(int N_img, int Lq) {
  if (N_img <= 0 || Lq < BIN_MULTI_MIN) return 0;
  return (size_t)N_img * BIN_PARTS * BIN_CNT_STRIDE * sizeof(unsigned) + (size_t)N_img * Lq * sizeof(unsigned short);
}

int mvg_bin_pairs(const float* ref_lvl, const uint8_t* inside, const int64_t* shapes_host, int L, int32_t* order,
                  int N_img, int Lq, void* workspace, size_t workspace_bytes, void* stream) {
  if (!ref_lvl || !shapes_host || !order || L <= 0 || N_img < 0 || Lq < 0) return MVG_E_BADARG;
  if ((long)N_img * Lq > 0x7fffffffL) return MVG_E_BADARG;
  if (N_img == 0 || Lq == 0) return 0;
  const int H0 = (int)shapes_host[0], W0 = (int)shapes_host[1];
  if (H0 <= 0 || W0 <= 0) return MVG_E_BADARG;
  int shift = 2;                                  // 4 x 4 level-0 cells per bin, coarser for maps wider than 256 cells
  while (((W0 - 1) >> shift) >= (1 << BIN_BITS) || ((H0 - 1) >> shift) >= (1 << BIN_BITS)) ++shift;
  if (Lq > 64 * 1024) return MVG_E_BADARG;        // more than 65 536 tokens per image: run the sampler unordered
  hipStream_t st = (hipStream_t)stream;
  if (g_bin_multi && Lq >= BIN_MULTI_MIN && workspace && workspace_bytes >= mvg_bin_pairs_workspace(N_img, Lq)) {
    // many pairs per image: BIN_PARTS workgroups per image, two kernels (see bin_count_kernel)
    unsigned* cnt = reinterpret_cast<unsigned*>(workspace);
    unsigned short* keys = reinterpret_cast<unsigned short*>(cnt + (size_t)N_img * BIN_PARTS * BIN_CNT_STRIDE);
    const int Lp = ((Lq + BIN_PARTS - 1) / BIN_PARTS + 63) / 64 * 64;       // slice length, whole wavefronts
#define MVG_BIN2(K)                                                                                              \
  {                                                                                                              \
    hipLaunchKernelGGL((bin_count_kernel<K>), dim3(N_img * BIN_PARTS), dim3(1024), 0, st, ref_lvl, inside, cnt,  \
                       keys, Lq, Lp, L, W0, H0, shift);                                                          \
    hipLaunchKernelGGL((bin_scatter_kernel<K>), dim3(N_img * BIN_PARTS), dim3(1024), 0, st, cnt, keys,           \
                       (int*)order, Lq, Lp);                                                                     \
  }
    if (Lp <= 1024) MVG_BIN2(1)
    else if (Lp <= 2 * 1024) MVG_BIN2(2)
    else if (Lp <= 4 * 1024) MVG_BIN2(4)
    else MVG_BIN2(8)
#undef MVG_BIN2
    MVG_LAUNCH_CHECK();
    return 0;
  }
#define MVG_BIN(K)                                                                                               \
  hipLaunchKernelGGL((bin_pairs_kernel<K>), dim3(N_img), dim3(1024), 0, st, ref_lvl, inside,                     \
                     (int*)order, Lq, L, W0, H0, shift)
  if (Lq <= 2 * 1024) MVG_BIN(2);                // few queries per image (a rank's shard of a query-sharded run)
  else if (Lq <= 4 * 1024) MVG_BIN(4);
  else if (Lq <= 8 * 1024) MVG_BIN(8);
  else if (Lq <= 16 * 1024) MVG_BIN(16);
  else if (Lq <= 32 * 1024) MVG_BIN(32);
  else MVG_BIN(64);
#undef MVG_BIN
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_gather_ref(const void* feat, int dtype, const float* ref_lvl, const float* x, const int64_t* shapes_host,
                   const int64_t* starts_host, void* ain, int V, int B, int Lq, int L, int S, int C, void* stream) {
  if (!feat || !ref_lvl || !x || !shapes_host || !starts_host || !ain || C % 4 != 0) return MVG_E_BADARG;
  LevelTable lv;
  int e = mvg_fill_levels(&lv, shapes_host, starts_host, L);
  if (e) return e;
  const long pairs = (long)V * B * Lq;
  if (pairs == 0) return 0;
  const int grid = mvg_ceil_div(pairs, 4);
  if (dtype == MVG_F32)
    hipLaunchKernelGGL((gather_ref_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)feat, ref_lvl, x,
                       lv, (float*)ain, B, Lq, S, C, (int)pairs);
  else if (dtype == MVG_BF16)
    hipLaunchKernelGGL((gather_ref_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feat, ref_lvl,
                       x, lv, (bf16_t*)ain, B, Lq, S, C, (int)pairs);
  else
    return MVG_E_BADARG;
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_mean_views(const void* attn, int dtype, void* out, int V, int rows, int C, void* stream) {
  if (!attn || !out || V <= 0 || C % 4 != 0) return MVG_E_BADARG;
  const long n4 = (long)rows * C / 4;
  if (n4 == 0) return 0;
  const int grid = mvg_ceil_div(n4, 256);
  if (dtype == MVG_F32)
    hipLaunchKernelGGL((mean_views_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)attn, (float*)out, V, n4);
  else if (dtype == MVG_BF16)
    hipLaunchKernelGGL((mean_views_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)attn, (bf16_t*)out, V, n4);
  else
    return MVG_E_BADARG;
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_add_layernorm(const float* res, const void* h, int h_dtype, const float* gamma, const float* beta, float* y,
                      int rows, int C, void* stream) {
  if (!res || !h || !gamma || !beta || !y || C % 4 != 0 || C > 1024) return MVG_E_BADARG;
  if (rows == 0) return 0;
  const int grid = mvg_ceil_div(rows, 4);
  if (h_dtype == MVG_F32)
    hipLaunchKernelGGL((add_ln_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, res, (const float*)h, gamma, beta, y, rows, C);
  else if (h_dtype == MVG_BF16)
    hipLaunchKernelGGL((add_ln_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, res, (const bf16_t*)h, gamma, beta, y, rows, C);
  else
    return MVG_E_BADARG;
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_class_head(const float* tgt, const float* Wc, const float* bc, float threshold, const uint8_t* forced_valid,
                   float* prob, uint8_t* valid, int* any_valid, int B, int NQ, int J, int C, void* stream) {
  if (!tgt || !Wc || !bc || !prob || !valid || !any_valid || C % 4 != 0) return MVG_E_BADARG;
  const int nq_total = B * NQ;
  if (nq_total == 0) return 0;
  hipLaunchKernelGGL(class_head_kernel, dim3(mvg_ceil_div(nq_total, 4)), dim3(256), 0, (hipStream_t)stream, tgt, Wc, bc,
                     threshold, forced_valid, prob, valid, any_valid, nq_total, J, C);
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_rowdot3(const void* h, int h_dtype, const float* W3, const float* b3, float* o, int rows, int C, void* stream) {
  if (!h || !W3 || !b3 || !o || C % 4 != 0) return MVG_E_BADARG;
  if (rows == 0) return 0;
  const int grid = mvg_ceil_div(rows, 4);
  if (h_dtype == MVG_F32)
    hipLaunchKernelGGL((rowdot3_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)h, W3, b3, o, rows, C);
  else if (h_dtype == MVG_BF16)
    hipLaunchKernelGGL((rowdot3_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, W3, b3, o, rows, C);
  else
    return MVG_E_BADARG;
  MVG_LAUNCH_CHECK();
  return 0;
}

static int launch_triangulate(const float* r, const float* o, const float* cams, const uint8_t* valid, const int* any_valid,
                              float* new_ref, float* ref2d, float* proj2d, int V, int B, int NQ, int J, const LevelTable& lv,
                              float* r_next, float* ref_lvl_next, uint8_t* inside_next, void* stream) {
  if (!r || !o || !cams || !valid || !any_valid || !new_ref || !ref2d || !proj2d || V <= 0) return MVG_E_BADARG;
  const long nprob = (long)B * NQ * J;            // 64 problems per 512-thread workgroup (8 lanes each in phase 1)
  if (nprob == 0) return 0;
  // (a lane-parallel Jacobi -- triangulate_kernel<true>, bit-identical -- measured 2.3 % slower in the forward: not instantiated)
  hipLaunchKernelGGL(triangulate_kernel<false>, dim3(mvg_ceil_div(nprob, TRI_PROBS)), dim3(512), 0, (hipStream_t)stream, r, o,
                     cams, valid, any_valid, new_ref, ref2d, proj2d, V, B, NQ, J, lv, r_next, ref_lvl_next, inside_next);
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_triangulate(const float* r, const float* o, const float* cams, const uint8_t* valid, const int* any_valid,
                    float* new_ref, float* ref2d, float* proj2d, int V, int B, int NQ, int J, void* stream) {
  LevelTable lv = {};
  return launch_triangulate(r, o, cams, valid, any_valid, new_ref, ref2d, proj2d, V, B, NQ, J, lv, nullptr, nullptr, nullptr,
                            stream);
}

int mvg_triangulate_project(const float* r, const float* o, const float* cams, const uint8_t* valid, const int* any_valid,
                            float* new_ref, float* ref2d, float* proj2d, int V, int B, int NQ, int J,
                            const int64_t* shapes_host, int L, float* r_next, float* ref_lvl_next, uint8_t* inside_next,
                            void* stream) {
  if (!shapes_host || !r_next || !ref_lvl_next || !inside_next) return MVG_E_BADARG;
  LevelTable lv;
  int64_t zeros[MVG_MAX_LEVELS] = {0};
  int e = mvg_fill_levels(&lv, shapes_host, zeros, L);
  if (e) return e;
  return launch_triangulate(r, o, cams, valid, any_valid, new_ref, ref2d, proj2d, V, B, NQ, J, lv, r_next, ref_lvl_next,
                            inside_next, stream);
}

int mvg_uncrop_undistort_jac(const float* ref2d, const float* cams, float* ud, float* jac, int V, int B, int Lq, void* stream) {
  if (!ref2d || !cams || !ud || !jac || V <= 0 || B <= 0 || Lq < 0) return MVG_E_BADARG;
  const long total = (long)B * V * Lq;
  if (total == 0) return 0;
  hipLaunchKernelGGL(uncrop_undistort_jac_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ref2d, cams, ud,
                     jac, V, B, Lq);
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_dlt_forward(const float* ud, const float* conf, const float* Pm, const uint8_t* valid, float* X, int V, int B, int NQ, int J,
                    void* stream) {
  if (!ud || !conf || !Pm || !valid || !X || V <= 0 || V > DLT_MAX_VIEWS || B <= 0 || NQ < 0 || J <= 0) return MVG_E_BADARG;
  const long total = (long)B * NQ * J;
  if (total == 0) return 0;
  hipLaunchKernelGGL(dlt_fwd_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream, ud, conf, Pm, valid, X, V, B,
                     (long)NQ * J, J);
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_dlt_backward(const float* ud, const float* conf, const float* Pm, const uint8_t* valid, const float* gX, float* g_ud,
                     float* g_conf, int V, int B, int NQ, int J, void* stream) {
  if (!ud || !conf || !Pm || !valid || !gX || !g_ud || !g_conf || V <= 0 || V > DLT_MAX_VIEWS || B <= 0 || NQ < 0 || J <= 0)
    return MVG_E_BADARG;
  const long total = (long)B * NQ * J;
  if (total == 0) return 0;
  hipLaunchKernelGGL(dlt_bwd_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)stream, ud, conf, Pm, valid, gX, g_ud,
                     g_conf, V, B, (long)NQ * J, J);
  MVG_LAUNCH_CHECK();
  return 0;
}

int mvg_sym4_eigh(const double* G, double* evals, double* evecs, long n, void* stream) {
  if (!G || !evals || !evecs || n < 0) return MVG_E_BADARG;
  if (n == 0) return 0;
  hipLaunchKernelGGL(sym4_eigh_kernel, dim3((unsigned)mvg_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, G, evals, evecs, n);
  MVG_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
