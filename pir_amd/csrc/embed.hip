// Frequency (positional) encoding of every PIR network input, forward and the three derivative forms autograd needs:
//   E(x)       = [x | sin(2^k x) | cos(2^k x)]_{k < L}                      src/models/embedder.py:20-41
//   B(x, g)    = g_x + sum_k 2^k (g_sin,k cos(2^k x) - g_cos,k sin(2^k x))   its vector-Jacobian product
//   T(x, u)    = [u | 2^k cos(2^k x) u | -2^k sin(2^k x) u]                  d B / d g applied to u (Jacobian-vector)
//   S(x, g, u) = -u sum_k 4^k (g_sin,k sin(2^k x) + g_cos,k cos(2^k x))      d B / d x applied to u
// T and S are what the backward of B needs when the material smoothness loss differentiates the networks w.r.t. the
// surface point a second time (render_surface.py:40-48, 76-78).  The reference runs 3 L + 1 eager launches per
// encoding and about twice that in the backward; each of these is one launch.  HBM-bound, a few hundred bytes/point.
#include "pir_common.h"

namespace pir {

constexpr int kEmbBlock = 256;

// one thread per (point, frequency-or-identity block, input dimension): ONE sincosf serves the sin and the cos column
// (the kernel is bound by the vector ALU, not by HBM: ocml's full-range sinf / cosf cost ~100 instructions each);
// stores of a wave cover two runs of contiguous columns per row, x[p, d] is re-read through L1/L2 (12 B per point)
template <bool kTangent>
__global__ void __launch_bounds__(kEmbBlock)
embed_out_kernel(long n_work, int D, int L, const float* __restrict__ x, const float* __restrict__ u,
                 float* __restrict__ out) {
  const int W = D * (1 + 2 * L), B = D * (1 + L);
  for (long i = (long)blockIdx.x * kEmbBlock + threadIdx.x; i < n_work; i += (long)gridDim.x * kEmbBlock) {
    const long p = i / B;
    const int j = (int)(i - p * B);
    const int blk = j / D, d = j - blk * D;
    const float xv = x[p * D + d];
    float* __restrict__ row = out + p * W;
    if (blk == 0) {
      row[d] = kTangent ? u[p * D + d] : xv;
    } else {
      const int k = blk - 1;
      const float f = (float)(1u << k);
      float sn, cs;
      sincosf(xv * f, &sn, &cs);                 // xv * f is exact: a power-of-two scale
      if (kTangent) {
        const float uv = u[p * D + d];
        row[(1 + 2 * k) * D + d] = (f * cs) * uv;
        row[(2 + 2 * k) * D + d] = -(f * sn) * uv;
      } else {
        row[(1 + 2 * k) * D + d] = sn;
        row[(2 + 2 * k) * D + d] = cs;
      }
    }
  }
}

// one thread per (point, input dimension): gathers the 2 L + 1 slots of its column
template <bool kSecond>
__global__ void __launch_bounds__(kEmbBlock)
embed_in_kernel(long n_in, int D, int L, const float* __restrict__ x, const float* __restrict__ g,
                const float* __restrict__ u, float* __restrict__ gx) {
  const int W = D * (1 + 2 * L);
  for (long i = (long)blockIdx.x * kEmbBlock + threadIdx.x; i < n_in; i += (long)gridDim.x * kEmbBlock) {
    const long p = i / D;
    const int d = (int)(i - p * D);
    const float xv = x[i];
    const float* __restrict__ gp = g + p * W + d;
    float acc = kSecond ? 0.0f : gp[0];
    for (int k = 0; k < L; ++k) {
      const float f = (float)(1u << k);
      float s, c;
      sincosf(xv * f, &s, &c);
      const float gs = gp[(1 + 2 * k) * D], gc = gp[(2 + 2 * k) * D];
      if (kSecond) acc -= (f * f) * (gs * s + gc * c);
      else acc += f * (gs * c - gc * s);
    }
    gx[i] = kSecond ? acc * u[i] : acc;
  }
}

static inline unsigned emb_grid(long n) {
  const long blocks = (n + kEmbBlock - 1) / kEmbBlock;
  const long cap = 16L * pir_num_cus();
  return (unsigned)(blocks < cap ? blocks : cap);
}

static inline int emb_args_ok(long P, int D, int L) { return P >= 0 && D >= 1 && D <= 16 && L >= 0 && L <= 16; }

}  // namespace pir

using namespace pir;

extern "C" {

int pir_embed_forward(long P, int D, int L, const float* x, float* out, void* stream) {
  if (!emb_args_ok(P, D, L)) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!x || !out) return PIR_ERR_ARG;
  const long n = P * D * (1 + L);
  hipLaunchKernelGGL(embed_out_kernel<false>, dim3(emb_grid(n)), dim3(kEmbBlock), 0, (hipStream_t)stream, n, D, L, x,
                     (const float*)nullptr, out);
  return pir_check_launch();
}

int pir_embed_backward(long P, int D, int L, const float* x, const float* g_out, float* g_x, void* stream) {
  if (!emb_args_ok(P, D, L)) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!x || !g_out || !g_x) return PIR_ERR_ARG;
  const long n = P * D;
  hipLaunchKernelGGL(embed_in_kernel<false>, dim3(emb_grid(n)), dim3(kEmbBlock), 0, (hipStream_t)stream, n, D, L, x,
                     g_out, (const float*)nullptr, g_x);
  return pir_check_launch();
}

int pir_embed_tangent(long P, int D, int L, const float* x, const float* u, float* out, void* stream) {
  if (!emb_args_ok(P, D, L)) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!x || !u || !out) return PIR_ERR_ARG;
  const long n = P * D * (1 + L);
  hipLaunchKernelGGL(embed_out_kernel<true>, dim3(emb_grid(n)), dim3(kEmbBlock), 0, (hipStream_t)stream, n, D, L, x, u,
                     out);
  return pir_check_launch();
}

int pir_embed_second(long P, int D, int L, const float* x, const float* g_out, const float* u, float* g_x,
                     void* stream) {
  if (!emb_args_ok(P, D, L)) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!x || !g_out || !u || !g_x) return PIR_ERR_ARG;
  const long n = P * D;
  hipLaunchKernelGGL(embed_in_kernel<true>, dim3(emb_grid(n)), dim3(kEmbBlock), 0, (hipStream_t)stream, n, D, L, x,
                     g_out, u, g_x);
  return pir_check_launch();
}

}  // extern "C"
