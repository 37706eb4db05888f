// Point-light GGX shader (Mitsuba roughplastic-aligned) forward + analytic backward, one fused HBM-bound
// kernel each.  Reference: GGXColocatedRenderer.forward, src/renderer/renderer_ggx.py:39-127.
// [P,3] tensors stay in the reference's AoS layout in HBM; a block moves them as flat, fully coalesced
// float4 streams through LDS and each thread picks its point's 3 floats from there (stride 3: conflict-free).
#include <hip/hip_runtime.h>
#include "pir_common.h"

namespace pir {

constexpr int kGgxBlock = 256;
constexpr float kEta = 1.48958738f;
constexpr float kF0 = 0.03867f;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kLo = 0.00001f, kHi = 0.99999f;

typedef float f4 __attribute__((ext_vector_type(4)));

// global [P,3] -> LDS [<=768] for the block's points [base, base+256)
__device__ __forceinline__ void stage_in3(const float* __restrict__ g, float* s, long base, long P) {
  const long f0 = 3 * base;                    // 3*256*k: 16-byte aligned
  const long nf = 3 * (P - base < kGgxBlock ? P - base : kGgxBlock);
  const int t = threadIdx.x;
  if (t < 192) {
    if (4 * t + 3 < nf) {
      *reinterpret_cast<f4*>(s + 4 * t) = *reinterpret_cast<const f4*>(g + f0 + 4 * t);
    } else {
      for (int k = 0; k < 4; ++k) if (4 * t + k < nf) s[4 * t + k] = g[f0 + 4 * t + k];
    }
  }
}
__device__ __forceinline__ void stage_out3(float* __restrict__ g, const float* s, long base, long P) {
  const long f0 = 3 * base;
  const long nf = 3 * (P - base < kGgxBlock ? P - base : kGgxBlock);
  const int t = threadIdx.x;
  if (t < 192) {
    if (4 * t + 3 < nf) {
      *reinterpret_cast<f4*>(g + f0 + 4 * t) = *reinterpret_cast<const f4*>(s + 4 * t);
    } else {
      for (int k = 0; k < 4; ++k) if (4 * t + k < nf) g[f0 + 4 * t + k] = s[4 * t + k];
    }
  }
}

struct GgxTables { const float* rough; int n_rough; const float* diff; int n_diff; };

struct GgxPoint {
  float I[3];                 // light intensity after falloff (all equal for a scalar light)
  float co, ci, ch, a;        // clamped cosines / alpha
  bool pass_o, pass_i, pass_h, pass_a;
  float H[3], hnorm;          // half vector, unclamped |v + l|
  float D, G1i, G1o, Kd, Ks;  // Kd, Ks: diffuse / specular factors multiplying I * albedo
  float a2, root, inv4ci;
};

__device__ __forceinline__ float smith_g1(float c, float a) {
  const float rt = a * (sqrtf(1.0f - c * c) / (c + 1e-10f));
  return 2.0f / (1.0f + hypotf(rt, 1.0f));
}

// floor(fl(fl(x ** 0.25) * scale)) with x ** 0.25 as torch computes it for fp32 tensors on the CPU the fixtures come
// from (a <= 1 ulp vectorised pow, in practice the correctly rounded value).  powf is within 2 ulp of that, which can
// only move the result when w * scale sits within ~2.4e-5 of an integer; those points (about 2 in 1000) take the
// exact path: x ** 0.25 correctly rounded through double (two correctly rounded square roots, one rounding to float).
// The fast path only has to be within 1e-3 / scale of the true value: sqrtf(sqrtf(x)) is (<= 1 ulp).
__device__ __forceinline__ long quarter_bin(float x, float scale) {
#ifdef PIR_GGX_POWF      // A/B: the round-2 form (bare powf: 0.2 % of the bins one off)
  return (long)floorf(powf(x, 0.25f) * scale);
#else
#ifdef PIR_GGX_POWF_FAST     // A/B: powf on the fast path (round 3 until the last day: ~100 instructions per call)
  float t = powf(x, 0.25f) * scale;
#else
  float t = sqrtf(sqrtf(x)) * scale;     // two correctly rounded roots: within 1 ulp of x ** 0.25, a tenth of powf's cost
#endif
  float fl = floorf(t);
  if (t - fl < 1e-3f || fl + 1.0f - t < 1e-3f) {
    t = (float)sqrt(sqrt((double)x)) * scale;
    fl = floorf(t);
  }
  return (long)fl;
#endif
}

// Table bins of the rough-transmittance lookups (renderer_ggx.py:88-112): tx = floor(cos^0.25 * 100),
// ty = floor((alpha / 4)^0.25 * 50), index = ty * 100 + tx clamped to the table; the 50-entry table uses ty clamped.
__device__ __forceinline__ void ggx_table_indices(float a, float ci, float co, const GgxTables& tb, long& ii, long& io,
                                                  long& id) {
  const long ty = quarter_bin(a / 4.0f, 50.0f);
  ii = ty * 100 + quarter_bin(ci, 100.0f);
  io = ty * 100 + quarter_bin(co, 100.0f);
  ii = ii < 0 ? 0 : (ii > tb.n_rough - 1 ? tb.n_rough - 1 : ii);
  io = io < 0 ? 0 : (io > tb.n_rough - 1 ? tb.n_rough - 1 : io);
  id = ty < 0 ? 0 : (ty > tb.n_diff - 1 ? tb.n_diff - 1 : ty);
}

__device__ __forceinline__ void ggx_eval(GgxPoint& p, const float* light, int light_rgb, int edr, float dist,
                                         const float (&n)[3], const float (&l)[3], const float (&v)[3], float alpha,
                                         const GgxTables& tb) {
  const float fall = edr ? 1.0f / (dist * dist + 1e-10f) : 1.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) p.I[c] = (light_rgb ? light[c] : light[0]) * fall;
  const float co_raw = v[0] * n[0] + v[1] * n[1] + v[2] * n[2];
  const float ci_raw = l[0] * n[0] + l[1] * n[1] + l[2] * n[2];
  p.pass_o = co_raw >= kLo && co_raw <= kHi;
  p.pass_i = ci_raw >= kLo && ci_raw <= kHi;
  p.co = fminf(fmaxf(co_raw, kLo), kHi);
  p.ci = fminf(fmaxf(ci_raw, kLo), kHi);
  float hs[3] = {v[0] + l[0], v[1] + l[1], v[2] + l[2]};
  p.hnorm = sqrtf(hs[0] * hs[0] + hs[1] * hs[1] + hs[2] * hs[2]);
  const float hd = fmaxf(p.hnorm, 1e-10f);
#pragma unroll
  for (int c = 0; c < 3; ++c) p.H[c] = hs[c] / hd;
  const float ch_raw = p.H[0] * n[0] + p.H[1] * n[1] + p.H[2] * n[2];
  p.pass_h = ch_raw >= kLo && ch_raw <= kHi;
  p.ch = fminf(fmaxf(ch_raw, kLo), kHi);
  p.pass_a = alpha >= 0.0001f;
  p.a = fmaxf(alpha, 0.0001f);
  const float ch2 = p.ch * p.ch;
  p.a2 = p.a * p.a;
  p.root = ch2 + (1.0f - ch2) / (p.a2 + 1e-10f);
  p.D = 1.0f / (kPi * p.a2 * p.root * p.root + 1e-10f);
  p.G1i = smith_g1(p.ci, p.a);
  p.G1o = smith_g1(p.co, p.a);
  p.inv4ci = 1.0f / (4.0f * p.ci + 1e-10f);
  p.Ks = kF0 * p.D * (p.G1i * p.G1o) * p.inv4ci;
  // tabulated rough transmittance (floor-indexed, no gradient)
  long ii, io, id;
  ggx_table_indices(p.a, p.ci, p.co, tb, ii, io, id);
  const float T12 = fminf(fmaxf(tb.rough[ii], 0.0f), 1.0f);
  const float T21 = fminf(fmaxf(tb.rough[io], 0.0f), 1.0f);
  const float Fdr = fminf(fmaxf(1.0f - tb.diff[id], 0.0f), 1.0f);
  // Kd / co: the reference multiplies in this order: (kd/(1-Fdr+eps)/pi) * co * T12 * T21 * invEta2
  p.Kd = (1.0f / (1.0f - Fdr + 1e-10f) / kPi) * p.co * T12 * T21 * (1.0f / (kEta * kEta));
}

__global__ void __launch_bounds__(kGgxBlock)
ggx_forward_kernel(long P, const float* __restrict__ light, int light_rgb, int edr, const float* __restrict__ dist,
                   const float* __restrict__ normal, const float* __restrict__ ldir, const float* __restrict__ vdir,
                   const float* __restrict__ kd, const float* __restrict__ ks, const float* __restrict__ alpha,
                   GgxTables tb, float* __restrict__ out_d, float* __restrict__ out_s, float* __restrict__ out_rgb) {
  __shared__ float s[6][3 * kGgxBlock];
  for (long base = (long)blockIdx.x * kGgxBlock; base < P; base += (long)gridDim.x * kGgxBlock) {
    stage_in3(normal, s[0], base, P);
    stage_in3(ldir, s[1], base, P);
    stage_in3(vdir, s[2], base, P);
    stage_in3(kd, s[3], base, P);
    stage_in3(ks, s[4], base, P);
    if (light_rgb) stage_in3(light, s[5], base, P);
    __syncthreads();
    const int t = threadIdx.x;
    const long i = base + t;
    float od[3] = {0, 0, 0}, os[3] = {0, 0, 0};
    if (i < P) {
      const float n[3] = {s[0][3 * t], s[0][3 * t + 1], s[0][3 * t + 2]};
      const float l[3] = {s[1][3 * t], s[1][3 * t + 1], s[1][3 * t + 2]};
      const float v[3] = {s[2][3 * t], s[2][3 * t + 1], s[2][3 * t + 2]};
      GgxPoint p;
      ggx_eval(p, light_rgb ? &s[5][3 * t] : light, light_rgb, edr, dist[i], n, l, v, alpha[i], tb);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        od[c] = p.I[c] * s[3][3 * t + c] * p.Kd;
        os[c] = p.I[c] * s[4][3 * t + c] * p.Ks;
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) { s[0][3 * t + c] = od[c]; s[1][3 * t + c] = os[c]; s[2][3 * t + c] = od[c] + os[c]; }
    __syncthreads();
    stage_out3(out_d, s[0], base, P);
    stage_out3(out_s, s[1], base, P);
    stage_out3(out_rgb, s[2], base, P);
    __syncthreads();
  }
}

// Forward without LDS: a thread owns FOUR consecutive points, i.e. 12 consecutive floats of every [P,3] array = three
// 16-byte loads per array straight into registers (17 independent loads in flight per thread, no barrier, no
// load -> barrier -> compute -> barrier -> store phases).  The last P % 4 points are done by scalar accesses.
__device__ __forceinline__ void ggx_point_out(const GgxPoint& p, const float* kd, const float* ks, float* od, float* os,
                                              float* orgb) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    od[c] = p.I[c] * kd[c] * p.Kd;
    os[c] = p.I[c] * ks[c] * p.Ks;
    orgb[c] = od[c] + os[c];
  }
}

__global__ void __launch_bounds__(kGgxBlock)
ggx_forward_v4_kernel(long P, const float* __restrict__ light, int light_rgb, int edr, const float* __restrict__ dist,
                      const float* __restrict__ normal, const float* __restrict__ ldir, const float* __restrict__ vdir,
                      const float* __restrict__ kd, const float* __restrict__ ks, const float* __restrict__ alpha,
                      GgxTables tb, float* __restrict__ out_d, float* __restrict__ out_s, float* __restrict__ out_rgb) {
  const long nquad = P >> 2;
  const float light0 = light_rgb ? 0.f : light[0];
  for (long q = (long)blockIdx.x * kGgxBlock + threadIdx.x; q < nquad; q += (long)gridDim.x * kGgxBlock) {
    union V12 { f4 v[3]; float f[12]; };
    V12 n, l, v, a_kd, a_ks, lt, od, os, orgb;
    union V4 { f4 v; float f[4]; } d4, al4;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      n.v[k] = reinterpret_cast<const f4*>(normal)[3 * q + k];
      l.v[k] = reinterpret_cast<const f4*>(ldir)[3 * q + k];
      v.v[k] = reinterpret_cast<const f4*>(vdir)[3 * q + k];
      a_kd.v[k] = reinterpret_cast<const f4*>(kd)[3 * q + k];
      a_ks.v[k] = reinterpret_cast<const f4*>(ks)[3 * q + k];
      if (light_rgb) lt.v[k] = reinterpret_cast<const f4*>(light)[3 * q + k];
    }
    d4.v = reinterpret_cast<const f4*>(dist)[q];
    al4.v = reinterpret_cast<const f4*>(alpha)[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float nn[3] = {n.f[3 * j], n.f[3 * j + 1], n.f[3 * j + 2]};
      const float ll[3] = {l.f[3 * j], l.f[3 * j + 1], l.f[3 * j + 2]};
      const float vv[3] = {v.f[3 * j], v.f[3 * j + 1], v.f[3 * j + 2]};
      // by value: selecting between the addresses of two register arrays would put both into scratch memory
      const float lrgb[3] = {light_rgb ? lt.f[3 * j] : light0, light_rgb ? lt.f[3 * j + 1] : light0,
                             light_rgb ? lt.f[3 * j + 2] : light0};
      GgxPoint p;
      ggx_eval(p, lrgb, 1, edr, d4.f[j], nn, ll, vv, al4.f[j], tb);
      ggx_point_out(p, &a_kd.f[3 * j], &a_ks.f[3 * j], &od.f[3 * j], &os.f[3 * j], &orgb.f[3 * j]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      reinterpret_cast<f4*>(out_d)[3 * q + k] = od.v[k];
      reinterpret_cast<f4*>(out_s)[3 * q + k] = os.v[k];
      reinterpret_cast<f4*>(out_rgb)[3 * q + k] = orgb.v[k];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (P & 3)) {
    const long i = 4 * nquad + threadIdx.x;
    const float nn[3] = {normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
    const float ll[3] = {ldir[3 * i], ldir[3 * i + 1], ldir[3 * i + 2]};
    const float vv[3] = {vdir[3 * i], vdir[3 * i + 1], vdir[3 * i + 2]};
    const float lrgb[3] = {light_rgb ? light[3 * i] : light0, light_rgb ? light[3 * i + 1] : light0,
                           light_rgb ? light[3 * i + 2] : light0};
    GgxPoint p;
    ggx_eval(p, lrgb, 1, edr, dist[i], nn, ll, vv, alpha[i], tb);
    ggx_point_out(p, kd + 3 * i, ks + 3 * i, out_d + 3 * i, out_s + 3 * i, out_rgb + 3 * i);
  }
}

// d G1 / d c and d G1 / d a
__device__ __forceinline__ void smith_g1_grad(float c, float a, float& dc, float& da) {
  const float sn = sqrtf(1.0f - c * c);
  const float ce = c + 1e-10f;
  const float tn = sn / ce;
  const float rt = a * tn;
  const float hy = hypotf(rt, 1.0f);
  const float dg_dhy = -2.0f / ((1.0f + hy) * (1.0f + hy));
  const float dg_drt = dg_dhy * (rt / hy);
  const float dtn_dc = (-c / sn) / ce - sn / (ce * ce);
  dc = dg_drt * a * dtn_dc;
  da = dg_drt * tn;
}

__global__ void __launch_bounds__(kGgxBlock)
ggx_backward_kernel(long P, const float* __restrict__ light, int light_rgb, int edr, const float* __restrict__ dist,
                    const float* __restrict__ normal, const float* __restrict__ ldir, const float* __restrict__ vdir,
                    const float* __restrict__ kd, const float* __restrict__ ks, const float* __restrict__ alpha,
                    GgxTables tb, const float* __restrict__ g_diff, const float* __restrict__ g_spec,
                    float* __restrict__ g_light /* [P] or [P,3] per-point contributions */,
                    float* __restrict__ g_dist, float* __restrict__ g_normal, float* __restrict__ g_ldir,
                    float* __restrict__ g_vdir, float* __restrict__ g_kd, float* __restrict__ g_ks,
                    float* __restrict__ g_alpha) {
  __shared__ float s[8][3 * kGgxBlock];
  for (long base = (long)blockIdx.x * kGgxBlock; base < P; base += (long)gridDim.x * kGgxBlock) {
    stage_in3(normal, s[0], base, P);
    stage_in3(ldir, s[1], base, P);
    stage_in3(vdir, s[2], base, P);
    stage_in3(kd, s[3], base, P);
    stage_in3(ks, s[4], base, P);
    stage_in3(g_diff, s[5], base, P);
    stage_in3(g_spec, s[6], base, P);
    if (light_rgb) stage_in3(light, s[7], base, P);
    __syncthreads();
    const int t = threadIdx.x;
    const long i = base + t;
    float gn[3] = {0, 0, 0}, gl[3] = {0, 0, 0}, gv[3] = {0, 0, 0}, gkd[3] = {0, 0, 0}, gks[3] = {0, 0, 0}, gI[3] = {0, 0, 0};
    float gdist = 0.f, galpha = 0.f;
    if (i < P) {
      const float n[3] = {s[0][3 * t], s[0][3 * t + 1], s[0][3 * t + 2]};
      const float l[3] = {s[1][3 * t], s[1][3 * t + 1], s[1][3 * t + 2]};
      const float v[3] = {s[2][3 * t], s[2][3 * t + 1], s[2][3 * t + 2]};
      const float d = dist[i];
      GgxPoint p;
      ggx_eval(p, light_rgb ? &s[7][3 * t] : light, light_rgb, edr, d, n, l, v, alpha[i], tb);
      float SdI = 0.f, SsI = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float gd = s[5][3 * t + c], gs = s[6][3 * t + c];
        const float kdc = s[3][3 * t + c], ksc = s[4][3 * t + c];
        gkd[c] = gd * p.I[c] * p.Kd;
        gks[c] = gs * p.I[c] * p.Ks;
        gI[c] = gd * kdc * p.Kd + gs * ksc * p.Ks;
        SdI += gd * kdc * p.I[c];
        SsI += gs * ksc * p.I[c];
      }
      // specular factor Ks = F D G1i G1o inv4ci ; diffuse factor Kd = const * co
      float dG1i_dc, dG1i_da, dG1o_dc, dG1o_da;
      smith_g1_grad(p.ci, p.a, dG1i_dc, dG1i_da);
      smith_g1_grad(p.co, p.a, dG1o_dc, dG1o_da);
      const float FD4 = kF0 * p.D * p.inv4ci;
      float g_co = SdI * (p.Kd / p.co) + SsI * FD4 * p.G1i * dG1o_dc;
      float g_ci = SsI * (FD4 * dG1i_dc * p.G1o - kF0 * p.D * p.G1i * p.G1o * 4.0f * p.inv4ci * p.inv4ci);
      const float g_D = SsI * kF0 * p.G1i * p.G1o * p.inv4ci;
      const float a2e = p.a2 + 1e-10f;
      const float ch2 = p.ch * p.ch;
      const float g_den = -g_D * p.D * p.D;
      const float g_root = g_den * 2.0f * kPi * p.a2 * p.root;
      const float g_a2 = g_den * kPi * p.root * p.root - g_root * (1.0f - ch2) / (a2e * a2e);
      float g_ch = g_root * (1.0f - 1.0f / a2e) * 2.0f * p.ch;
      const float g_a = g_a2 * 2.0f * p.a + SsI * FD4 * (dG1i_da * p.G1o + p.G1i * dG1o_da);
      galpha = p.pass_a ? g_a : 0.f;
      if (!p.pass_o) g_co = 0.f;
      if (!p.pass_i) g_ci = 0.f;
      if (!p.pass_h) g_ch = 0.f;
      // half vector
      float gH[3] = {g_ch * n[0], g_ch * n[1], g_ch * n[2]};
      float gHs[3];
      if (p.hnorm >= 1e-10f) {
        const float dot = p.H[0] * gH[0] + p.H[1] * gH[1] + p.H[2] * gH[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) gHs[c] = (gH[c] - p.H[c] * dot) / p.hnorm;
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) gHs[c] = gH[c] / 1e-10f;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gn[c] = g_co * v[c] + g_ci * l[c] + g_ch * p.H[c];
        gv[c] = g_co * n[c] + gHs[c];
        gl[c] = g_ci * n[c] + gHs[c];
      }
      if (edr) {
        const float den = d * d + 1e-10f;
        float gsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float lc = light_rgb ? s[7][3 * t + c] : light[0];
          gsum += gI[c] * lc;
          gI[c] = gI[c] / den;   // -> gradient of the light value itself
        }
        gdist = -gsum * 2.0f * d / (den * den);
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s[0][3 * t + c] = gn[c]; s[1][3 * t + c] = gl[c]; s[2][3 * t + c] = gv[c];
      s[3][3 * t + c] = gkd[c]; s[4][3 * t + c] = gks[c]; s[5][3 * t + c] = gI[c];
    }
    if (i < P) {
      g_dist[i] = gdist;
      g_alpha[i] = galpha;
      if (!light_rgb) g_light[i] = gI[0] + gI[1] + gI[2];
    }
    __syncthreads();
    stage_out3(g_normal, s[0], base, P);
    stage_out3(g_ldir, s[1], base, P);
    stage_out3(g_vdir, s[2], base, P);
    stage_out3(g_kd, s[3], base, P);
    stage_out3(g_ks, s[4], base, P);
    if (light_rgb) stage_out3(g_light, s[5], base, P);
    __syncthreads();
  }
}

template <typename... Ts>
static bool ggx_aligned16(const Ts*... ptrs) {
  return (((reinterpret_cast<uintptr_t>(ptrs)) | ...) & 15) == 0;
}

static unsigned ggx_grid(long P) {
  long blocks = (P + kGgxBlock - 1) / kGgxBlock;
  long cap = 8L * pir_num_cus();
  return (unsigned)(blocks < cap ? blocks : cap);
}

}  // namespace pir

using namespace pir;

// Debug / parity output: the table bins every point lands in (same device function as the shading kernels).
__global__ void ggx_indices_kernel(long P, const float* __restrict__ normal, const float* __restrict__ ldir,
                                   const float* __restrict__ vdir, const float* __restrict__ alpha, GgxTables tb,
                                   int* __restrict__ idx_i, int* __restrict__ idx_o, int* __restrict__ idx_d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float n[3] = {normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]};
  const float co_raw = vdir[3 * i] * n[0] + vdir[3 * i + 1] * n[1] + vdir[3 * i + 2] * n[2];
  const float ci_raw = ldir[3 * i] * n[0] + ldir[3 * i + 1] * n[1] + ldir[3 * i + 2] * n[2];
  const float co = fminf(fmaxf(co_raw, kLo), kHi), ci = fminf(fmaxf(ci_raw, kLo), kHi);
  const float a = fmaxf(alpha[i], 0.0001f);
  long ii, io, id;
  ggx_table_indices(a, ci, co, tb, ii, io, id);
  idx_i[i] = (int)ii; idx_o[i] = (int)io; idx_d[i] = (int)id;
}

extern "C" {

int pir_ggx_table_indices(long P, const float* normal, const float* lightdir, const float* viewdir, const float* alpha,
                          int n_rough, int n_diff, int* idx_wi, int* idx_wo, int* idx_diff, void* stream) {
  if (P < 0 || n_rough <= 0 || n_diff <= 0) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!normal || !lightdir || !viewdir || !alpha || !idx_wi || !idx_wo || !idx_diff) return PIR_ERR_ARG;
  GgxTables tb{nullptr, n_rough, nullptr, n_diff};
  hipLaunchKernelGGL(ggx_indices_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, normal,
                     lightdir, viewdir, alpha, tb, idx_wi, idx_wo, idx_diff);
  return pir_check_launch();
}

int pir_ggx_forward(long P, const float* light, int light_is_rgb, int direct_radiance, const float* distance,
                    const float* normal, const float* lightdir, const float* viewdir, const float* kd, const float* ks,
                    const float* alpha, const float* table_rough, int n_rough, const float* table_diff, int n_diff,
                    float* out_diffuse, float* out_specular, float* out_rgb, void* stream) {
  if (P < 0) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!light || !distance || !normal || !lightdir || !viewdir || !kd || !ks || !alpha || !table_rough || !table_diff ||
      !out_diffuse || !out_specular || !out_rgb || n_rough <= 0 || n_diff <= 0)
    return PIR_ERR_ARG;
  GgxTables tb{table_rough, n_rough, table_diff, n_diff};
  PirProfScope prof(PIR_K_GGX_FWD, (double)P, stream);
#ifndef PIR_GGX_LDS       // A/B: -DPIR_GGX_LDS = the round-1/2 kernel (AoS staged through LDS, one point per thread)
  if (ggx_aligned16(normal, lightdir, viewdir, kd, ks, out_diffuse, out_specular, out_rgb, distance, alpha) &&
      (!light_is_rgb || ggx_aligned16(light))) {
    const long quads = (P + 3) / 4, blocks = (quads + kGgxBlock - 1) / kGgxBlock, cap = 8L * pir_num_cus();
    hipLaunchKernelGGL(ggx_forward_v4_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(kGgxBlock), 0,
                       (hipStream_t)stream, P, light, light_is_rgb, direct_radiance, distance, normal, lightdir, viewdir,
                       kd, ks, alpha, tb, out_diffuse, out_specular, out_rgb);
    return pir_check_launch();
  }
#endif
  hipLaunchKernelGGL(ggx_forward_kernel, dim3(ggx_grid(P)), dim3(kGgxBlock), 0, (hipStream_t)stream, P, light,
                     light_is_rgb, direct_radiance, distance, normal, lightdir, viewdir, kd, ks, alpha, tb, out_diffuse,
                     out_specular, out_rgb);
  return pir_check_launch();
}

int pir_ggx_backward(long P, const float* light, int light_is_rgb, int direct_radiance, const float* distance,
                     const float* normal, const float* lightdir, const float* viewdir, const float* kd, const float* ks,
                     const float* alpha, const float* table_rough, int n_rough, const float* table_diff, int n_diff,
                     const float* grad_diffuse, const float* grad_specular, float* g_light, float* g_distance,
                     float* g_normal, float* g_lightdir, float* g_viewdir, float* g_kd, float* g_ks, float* g_alpha,
                     void* stream) {
  if (P < 0) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!light || !distance || !normal || !lightdir || !viewdir || !kd || !ks || !alpha || !table_rough || !table_diff ||
      !grad_diffuse || !grad_specular || !g_light || !g_distance || !g_normal || !g_lightdir || !g_viewdir || !g_kd ||
      !g_ks || !g_alpha || n_rough <= 0 || n_diff <= 0)
    return PIR_ERR_ARG;
  GgxTables tb{table_rough, n_rough, table_diff, n_diff};
  PirProfScope prof(PIR_K_GGX_BWD, (double)P, stream);
  hipLaunchKernelGGL(ggx_backward_kernel, dim3(ggx_grid(P)), dim3(kGgxBlock), 0, (hipStream_t)stream, P, light,
                     light_is_rgb, direct_radiance, distance, normal, lightdir, viewdir, kd, ks, alpha, tb,
                     grad_diffuse, grad_specular, g_light, g_distance, g_normal, g_lightdir, g_viewdir,
                     g_kd, g_ks, g_alpha);
  return pir_check_launch();
}

}  // extern "C"
