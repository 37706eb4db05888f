// Light-visibility march around the SDF kernels (reference: light_visibility, src/renderer/raytracer.py:669-729).
//
//   pir_visibility_points           x[b,s] = light_pos + rays_d[b] * mid[b,s]          (march sample positions)
//   pir_visibility_forward          vis[b] = 1 - sum_s alpha[b,s] * prod_{j<s}(1 - alpha[b,j] + 1e-7)
//   pir_visibility_backward         d vis -> d sdf, d normal (per sample), d rays_d, d length (per point)
//   pir_visibility_points_backward  d x   -> d light_pos (per point partial), d rays_d, d length
//
// The SDF value and normal of every sample come from pir_sdf_get_all in between.  All four kernels are
// HBM-bound streams over the [bs, n] sample arrays; the composite kernels use one wavefront per surface point
// (n = n_samples - 10 <= 512 samples, a contiguous run of E samples per lane) with wave-level product / sum
// scans, so the cumulative product never leaves registers.
#include <hip/hip_runtime.h>
#include "pir_common.h"

namespace pir {

constexpr int kDiscard = 10;      // raytracer.py:682: the last 10 samples of linspace(0, 1, n_samples) are dropped
constexpr int kVisMaxE = 8;       // samples per lane -> n <= 512

__device__ __forceinline__ float vmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float vadd(float a, float b) { return __fadd_rn(a, b); }

// torch.linspace(0, 1, n)[i] (symmetric evaluation, as in trace.hip)
__device__ __forceinline__ float lin01(int i, int n) {
  const float step = 1.0f / (float)(n - 1);
  return i < n / 2 ? vmul(step, (float)i) : 1.0f - vmul(step, (float)(n - 1 - i));
}

struct March {   // per-sample march geometry of one point: z = t L, dist, mid
  float dist, mid;
};
__device__ __forceinline__ March march_sample(int s, int n, int n_samples, float L) {
  const float z = vmul(lin01(s, n_samples), L);
  March m;
  m.dist = s < n - 1 ? vadd(vmul(lin01(s + 1, n_samples), L), -z) : 1.0f / (float)n;
  m.mid = vadd(z, vmul(m.dist, 0.5f));
  return m;
}
// d mid / d L and d dist / d L
__device__ __forceinline__ void march_sample_dL(int s, int n, int n_samples, float& dmid, float& ddist) {
  const float t0 = lin01(s, n_samples);
  if (s < n - 1) {
    ddist = lin01(s + 1, n_samples) - t0;
    dmid = t0 + 0.5f * ddist;
  } else {
    ddist = 0.f;
    dmid = t0;
  }
}

__global__ void __launch_bounds__(256)
vis_points_kernel(const float* __restrict__ light_pos, const float* __restrict__ rays_d,
                  const float* __restrict__ length, long bs, int n, int n_samples, float* __restrict__ x) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bs * n) return;
  const long b = i / n;
  const int s = (int)(i - b * n);
  const March m = march_sample(s, n, n_samples, length[b]);
#pragma unroll
  for (int c = 0; c < 3; ++c) x[3 * i + c] = vadd(light_pos[c], vmul(rays_d[3 * b + c], m.mid));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per point: sums over the samples of d x
__global__ void __launch_bounds__(256)
vis_points_backward_kernel(const float* __restrict__ g_x, const float* __restrict__ rays_d,
                           const float* __restrict__ length, long bs, int n, int n_samples,
                           float* __restrict__ g_lp_point, float* __restrict__ g_rays_d, float* __restrict__ g_length) {
  const long b = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= bs) return;
  const int lane = threadIdx.x & 63;
  const float L = length[b];
  const float d0 = rays_d[3 * b], d1 = rays_d[3 * b + 1], d2 = rays_d[3 * b + 2];
  float lp[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f}, gL = 0.f;
  for (int s = lane; s < n; s += 64) {
    const float* g = g_x + 3 * (b * n + s);
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    const March m = march_sample(s, n, n_samples, L);
    float dmid, ddist;
    march_sample_dL(s, n, n_samples, dmid, ddist);
    lp[0] += g0; lp[1] += g1; lp[2] += g2;
    gd[0] += m.mid * g0; gd[1] += m.mid * g1; gd[2] += m.mid * g2;
    gL += (d0 * g0 + d1 * g1 + d2 * g2) * dmid;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { lp[c] = wave_sum(lp[c]); gd[c] = wave_sum(gd[c]); }
  gL = wave_sum(gL);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { g_lp_point[3 * b + c] = lp[c]; g_rays_d[3 * b + c] = gd[c]; }
    g_length[b] = gL;
  }
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

struct SampleFwd {
  float alpha, u;          // clipped alpha, 1 - alpha + 1e-7
  float prev, next, a, b;  // cdf values, numerator, denominator of the unclipped alpha
  float cosv, dist;
};
__device__ __forceinline__ SampleFwd sample_forward(float sdf, float n0, float n1, float n2, float d0, float d1,
                                                    float d2, float dist, float k) {
  SampleFwd f;
  f.dist = dist;
  f.cosv = d0 * n0 + d1 * n1 + d2 * n2;
  const float half = fminf(f.cosv, 0.0f) * dist * 0.5f;      // -relu(-cos) * dists * 0.5
  f.prev = sigmoidf((sdf - half) * k);
  f.next = sigmoidf((sdf + half) * k);
  f.a = f.prev - f.next + 1e-5f;
  f.b = f.prev + 1e-5f;
  const float raw = f.a / f.b;
  f.alpha = fminf(fmaxf(raw, 0.0f), 1.0f);
  f.u = 1.0f - f.alpha + 1e-7f;
  return f;
}

// exclusive multiplicative scan over the wave (lane 0 gets 1)
__device__ __forceinline__ float wave_exclusive_prod(float v, int lane) {
  float inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(inc, o);
    if (lane >= o) inc *= t;
  }
  const float ex = __shfl_up(inc, 1);
  return lane == 0 ? 1.0f : ex;
}
// exclusive suffix sum over the wave (lane 63 gets 0)
__device__ __forceinline__ float wave_exclusive_suffix_sum(float v, int lane) {
  float inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(inc, o);
    if (lane + o < 64) inc += t;
  }
  const float ex = __shfl_down(inc, 1);
  return lane == 63 ? 0.0f : ex;
}

template <int E, bool BACKWARD>
__global__ void __launch_bounds__(256)
vis_composite_kernel(const float* __restrict__ sdf, const float* __restrict__ normal,
                     const float* __restrict__ rays_d, const float* __restrict__ length,
                     const float* __restrict__ inv_s, long bs, int n, int n_samples, float* __restrict__ vis,
                     const float* __restrict__ g_vis, float* __restrict__ g_sdf, float* __restrict__ g_normal,
                     float* __restrict__ g_rays_d, float* __restrict__ g_length) {
  const long b = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= bs) return;
  const int lane = threadIdx.x & 63;
  const float k = inv_s[0];
  const float L = length[b];
  const float d0 = rays_d[3 * b], d1 = rays_d[3 * b + 1], d2 = rays_d[3 * b + 2];
  SampleFwd f[E];
  float nrm[E][3];
  float local = 1.0f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int s = lane * E + e;
    if (s < n) {
      const long i = b * n + s;
      nrm[e][0] = normal[3 * i]; nrm[e][1] = normal[3 * i + 1]; nrm[e][2] = normal[3 * i + 2];
      f[e] = sample_forward(sdf[i], nrm[e][0], nrm[e][1], nrm[e][2], d0, d1, d2,
                            march_sample(s, n, n_samples, L).dist, k);
    } else {
      f[e] = SampleFwd{0.f, 1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f};
      nrm[e][0] = nrm[e][1] = nrm[e][2] = 0.f;
    }
    local *= f[e].u;
  }
  float T[E];   // transmittance in front of each sample
  float run = wave_exclusive_prod(local, lane);
  float wsum = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    T[e] = run;
    wsum += f[e].alpha * run;
    run *= f[e].u;
  }
  if (!BACKWARD) {
    const float tot = wave_sum(wsum);
    if (lane == 0) vis[b] = 1.0f - tot;
    return;
  }
  // d vis / d alpha_s = -T_s + (sum_{r>s} alpha_r T_r) / u_s
  float after = wave_exclusive_suffix_sum(wsum, lane);
  const float gv = g_vis[b];
  float gd[3] = {0.f, 0.f, 0.f}, gL = 0.f;
#pragma unroll
  for (int e = E - 1; e >= 0; --e) {
    const int s = lane * E + e;
    const float G = gv * (-T[e] + after / f[e].u);
    after += f[e].alpha * T[e];
    if (s < n) {
      const float raw = f[e].a / f[e].b;
      const float Gr = (raw >= 0.0f && raw <= 1.0f) ? G : 0.0f;           // clip passes gradient inside [0, 1]
      const float dprev = f[e].prev * (1.0f - f[e].prev) * k;
      const float dnext = f[e].next * (1.0f - f[e].next) * k;
      const float r_prev = Gr * ((f[e].b - f[e].a) / (f[e].b * f[e].b)) * dprev;   // via prev_cdf
      const float r_next = -Gr / f[e].b * dnext;                                    // via next_cdf
      const float gs = r_prev + r_next;
      const float g_half = r_next - r_prev;
      const float neg = f[e].cosv < 0.0f ? 1.0f : 0.0f;                   // -relu(-cos) passes gradient for cos < 0
      const float g_cos = g_half * f[e].dist * 0.5f * neg;
      const float g_dist = g_half * fminf(f[e].cosv, 0.0f) * 0.5f;
      const long i = b * n + s;
      g_sdf[i] = gs;
      g_normal[3 * i] = g_cos * d0; g_normal[3 * i + 1] = g_cos * d1; g_normal[3 * i + 2] = g_cos * d2;
      gd[0] += g_cos * nrm[e][0]; gd[1] += g_cos * nrm[e][1]; gd[2] += g_cos * nrm[e][2];
      float dmid, ddist;
      march_sample_dL(s, n, n_samples, dmid, ddist);
      gL += g_dist * ddist;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) gd[c] = wave_sum(gd[c]);
  gL = wave_sum(gL);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) g_rays_d[3 * b + c] = gd[c];
    g_length[b] = gL;
  }
}

template <bool BACKWARD>
static int launch_composite(const float* sdf, const float* normal, const float* rays_d, const float* length,
                            const float* inv_s, long bs, int n_samples, float* vis, const float* g_vis, float* g_sdf,
                            float* g_normal, float* g_rays_d, float* g_length, void* stream) {
  const int n = n_samples - kDiscard;
  if (bs < 0 || n < 2 || n > 64 * kVisMaxE) return PIR_ERR_ARG;
  if (bs == 0) return PIR_OK;
  if (!sdf || !normal || !rays_d || !length || !inv_s) return PIR_ERR_ARG;
  const dim3 block(256), grid((unsigned)((bs + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
#define PIR_VIS_LAUNCH(E)                                                                                       \
  hipLaunchKernelGGL((vis_composite_kernel<E, BACKWARD>), grid, block, 0, st, sdf, normal, rays_d, length, inv_s, \
                     bs, n, n_samples, vis, g_vis, g_sdf, g_normal, g_rays_d, g_length)
  if (n <= 128) PIR_VIS_LAUNCH(2);
  else if (n <= 256) PIR_VIS_LAUNCH(4);
  else PIR_VIS_LAUNCH(8);
#undef PIR_VIS_LAUNCH
  return pir_check_launch();
}

}  // namespace pir

using namespace pir;

extern "C" {

int pir_visibility_points(const float* light_pos, const float* rays_d, const float* length, long bs, int n_samples,
                          float* x, void* stream) {
  const int n = n_samples - kDiscard;
  if (bs < 0 || n < 2) return PIR_ERR_ARG;
  if (bs == 0) return PIR_OK;
  if (!light_pos || !rays_d || !length || !x) return PIR_ERR_ARG;
  const long total = bs * n;
  hipLaunchKernelGGL(vis_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     light_pos, rays_d, length, bs, n, n_samples, x);
  return pir_check_launch();
}

int pir_visibility_points_backward(const float* g_x, const float* rays_d, const float* length, long bs, int n_samples,
                                   float* g_light_pos_point, float* g_rays_d, float* g_length, void* stream) {
  const int n = n_samples - kDiscard;
  if (bs < 0 || n < 2) return PIR_ERR_ARG;
  if (bs == 0) return PIR_OK;
  if (!g_x || !rays_d || !length || !g_light_pos_point || !g_rays_d || !g_length) return PIR_ERR_ARG;
  hipLaunchKernelGGL(vis_points_backward_kernel, dim3((unsigned)((bs + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     g_x, rays_d, length, bs, n, n_samples, g_light_pos_point, g_rays_d, g_length);
  return pir_check_launch();
}

int pir_visibility_forward(const float* sdf, const float* normal, const float* rays_d, const float* length,
                           const float* inv_s, long bs, int n_samples, float* vis, void* stream) {
  if (bs > 0 && !vis) return PIR_ERR_ARG;
  return launch_composite<false>(sdf, normal, rays_d, length, inv_s, bs, n_samples, vis, nullptr, nullptr, nullptr,
                                 nullptr, nullptr, stream);
}

int pir_visibility_backward(const float* g_vis, const float* sdf, const float* normal, const float* rays_d,
                            const float* length, const float* inv_s, long bs, int n_samples, float* g_sdf,
                            float* g_normal, float* g_rays_d, float* g_length, void* stream) {
  if (bs > 0 && (!g_vis || !g_sdf || !g_normal || !g_rays_d || !g_length)) return PIR_ERR_ARG;
  return launch_composite<true>(sdf, normal, rays_d, length, inv_s, bs, n_samples, nullptr, g_vis, g_sdf, g_normal,
                                g_rays_d, g_length, stream);
}

}  // extern "C"
