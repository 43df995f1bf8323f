/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Plain-C restatement of the reference's sampling forward and backward.
 *
 * Follows lib/models/ops/src/cuda/deform_im2col_cuda.cuh:248-309 (one output element per
 * (b, q, m, c), loop over levels and points) and the bilinear helper :44-94 (zero padding per
 * corner).  Used by tests/ and the cpu_baseline leg of bench.py; never by the product path.
 * Pinned against the golden vectors of the reference's own CPU twin (tests/golden/msda.npz).
 */
#include <math.h>
#include <stdint.h>

static float bilinear(const float* lvl, int H, int W, int M, int D, float h, float w, int m, int c) {
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low, hh = 1.f - lh, hw = 1.f - lw;
  const long ws = (long)M * D, hs = (long)W * ws;
  const long base = (long)m * D + c;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = lvl[h_low * hs + w_low * ws + base];
  if (h_low >= 0 && w_high <= W - 1) v2 = lvl[h_low * hs + w_high * ws + base];
  if (h_high <= H - 1 && w_low >= 0) v3 = lvl[h_high * hs + w_low * ws + base];
  if (h_high <= H - 1 && w_high <= W - 1) v4 = lvl[h_high * hs + w_high * ws + base];
  return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

/* value (N,S,M,D); shapes (L,2) (H,W); starts (L); loc (N,Lq,M,L,P,2); wgt (N,Lq,M,L,P); out (N,Lq,M*D) */
void msda_forward_ref(const float* value, const int64_t* shapes, const int64_t* starts, const float* loc,
                      const float* wgt, float* out, int N, int S, int M, int D, int L, int Lq, int P) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int q = 0; q < Lq; ++q)
      for (int m = 0; m < M; ++m) {
        const long qm = ((long)n * Lq + q) * M + m;
        for (int c = 0; c < D; ++c) {
          float col = 0.f;
          for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const float* lvl = value + ((long)n * S + starts[l]) * M * D;
            for (int p = 0; p < P; ++p) {
              const long s = qm * L * P + (long)l * P + p;
              const float h_im = loc[2 * s + 1] * H - 0.5f, w_im = loc[2 * s] * W - 0.5f;
              if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                col += bilinear(lvl, H, W, M, D, h_im, w_im, m, c) * wgt[s];
            }
          }
          out[qm * D + c] = col;
        }
      }
}

/* Backward of the same op -- lib/models/ops/src/cuda/deform_im2col_cuda.cuh:98-169 (deform_col2im_bilinear: the four
 * grad_value contributions, grad_h_weight / grad_w_weight, *grad_attn_weight) called per (b, q, m, c) and (l, p) by
 * cuh:312-413.  Accumulation in DOUBLE (the reference accumulates in fp32 with atomics; double makes this the truth the
 * fp32 kernels are compared with).  grad_value (N,S,M,D), grad_loc (N,Lq,M,L,P,2), grad_wgt (N,Lq,M,L,P): all double,
 * zeroed here.  Serial over (n, q) -- the grad_value scatter is not parallel-safe; ~2 s at a full cfg-2 view-layer. */
void msda_backward_ref(const float* value, const int64_t* shapes, const int64_t* starts, const float* loc,
                       const float* wgt, const float* gout, double* gvalue, double* gloc, double* gwgt,
                       int N, int S, int M, int D, int L, int Lq, int P) {
  const long nv = (long)N * S * M * D, ns = (long)N * Lq * M * L * P;
  for (long i = 0; i < nv; ++i) gvalue[i] = 0.0;
  for (long i = 0; i < ns; ++i) { gloc[2 * i] = 0.0; gloc[2 * i + 1] = 0.0; gwgt[i] = 0.0; }
  const long ws = (long)M * D;
  for (int n = 0; n < N; ++n)
    for (int q = 0; q < Lq; ++q)
      for (int m = 0; m < M; ++m) {
        const long qm = ((long)n * Lq + q) * M + m;
        for (int l = 0; l < L; ++l) {
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
          const long lbase = ((long)n * S + starts[l]) * ws + (long)m * D;
          const long hs = (long)W * ws;
          for (int p = 0; p < P; ++p) {
            const long s = qm * L * P + (long)l * P + p;
            const float h = loc[2 * s + 1] * H - 0.5f, w = loc[2 * s] * W - 0.5f;      /* cuh:385-386 */
            if (!(h > -1 && w > -1 && h < H && w < W)) continue;                      /* cuh:390 */
            const int h_low = (int)floorf(h), w_low = (int)floorf(w), h_high = h_low + 1, w_high = w_low + 1;
            const double lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;     /* cuh:113-117 */
            const double aw = wgt[s];
            double gw_acc = 0, gh_acc = 0, ga_acc = 0;
            for (int c = 0; c < D; ++c) {
              const double go = gout[qm * D + c], top = go * aw;                       /* cuh:127 */
              double v1 = 0, v2 = 0, v3 = 0, v4 = 0;
              if (h_low >= 0 && w_low >= 0) { const long o = lbase + h_low * hs + w_low * ws + c; v1 = value[o]; gvalue[o] += hh * hw * top; }
              if (h_low >= 0 && w_high <= W - 1) { const long o = lbase + h_low * hs + w_high * ws + c; v2 = value[o]; gvalue[o] += hh * lw * top; }
              if (h_high <= H - 1 && w_low >= 0) { const long o = lbase + h_high * hs + w_low * ws + c; v3 = value[o]; gvalue[o] += lh * hw * top; }
              if (h_high <= H - 1 && w_high <= W - 1) { const long o = lbase + h_high * hs + w_high * ws + c; v4 = value[o]; gvalue[o] += lh * lw * top; }
              gw_acc += (hh * (v2 - v1) + lh * (v4 - v3)) * top;                        /* cuh:130-160 grad_w_weight */
              gh_acc += (hw * (v3 - v1) + lw * (v4 - v2)) * top;                        /* grad_h_weight */
              ga_acc += go * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);   /* cuh:165 */
            }
            gloc[2 * s] = gw_acc * W;                                                   /* cuh:166 */
            gloc[2 * s + 1] = gh_acc * H;                                               /* cuh:167 */
            gwgt[s] = ga_acc;
          }
        }
      }
}
