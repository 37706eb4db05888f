// Sphere tracer over the SDF MLP: RayTracer.forward of the reference (src/renderer/raytracer.py:56-250)
// as a chain of launches on one stream with no host synchronisation:
//   init -> 17 x sphere-trace step (evaluate the compacted active list, step, re-compact by wave ballot)
//        -> dense sampler (128 samples per unfinished ray, first sign change) -> bisection steps -> final eval.
// Every comparison follows the reference literally (strict >, <, the order of mask update and iteration cap);
// ray state arithmetic uses un-contracted mul/add so it rounds like the eager PyTorch ops it replaces.
#include "sdf_sweeps.cuh"
#include "pir_common.h"
#include "host_logic.h"
#include <mutex>

namespace pir {

struct TraceWs {
  int* list[2];     // ping-pong active lists
  int* slist;       // rays handed to the dense sampler
  int* blist;       // rays handed to bisection
  float* d_lo; float* d_hi; float* f_lo; float* f_hi;   // bisection brackets, indexed by ray
  unsigned char* bwork;                                    // per-ray bisection work flag
  int* min_idx;     // per ray: index of the first negative sample of the dense sampler (raytracer.py:192-197);
                    // -1 = sampled, no negative sample; -2 = the ray never reached the sampler
  int* counters;    // [0..17] list sizes per sphere-trace eval (17 = sampler list), [32] bisect count, [33] status,
                    // [34] rays left after the sampler's head pass (listed in list[0])
  int* flags;       // [(max_bisect + 1)][nchunks] "any ray still bisecting" per reference chunk
};
constexpr int kCntBisect = 32, kCntStatus = 33, kCntSampler2 = 34, kNumCounters = 64;
constexpr int kHead = 8;                      // samples per ray of the sampler's head pass
constexpr int kHeadRays = 64 / kHead;         // rays per 64-point tile in that pass

// Profiling only (pir_profile_enable): SDF evaluations executed by the three tracer kernel families since the last
// read, accumulated on the device (index 0 sphere-trace steps, 1 sampler incl. its head pass, 2 bisection).
__device__ unsigned long long g_trace_evals[4];

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }

// Evaluate the SDF at the 64 points of this workgroup (lane (w, j, h) supplies the point of entry 32w + j);
// on return every thread may read the value of entry e via read_sdf(lds, pk, e >> 5, e & 31).
__device__ __forceinline__ void eval_tile(Lds& lds, const SdfPacked& pk, float x0, float x1, float x2) {
  forward_hidden<false>(lds, pk, x0, x1, x2, nullptr);
  sdf_head(lds, pk);
  __syncthreads();
}

// Short work lists (a 64 x 64 training patch is 4 096 rays = 64 tiles of 64 points for 256 CUs): when one 32-entry tile
// per workgroup covers the list, a workgroup takes 32 entries in operand set 1 only -- twice as many CUs busy, half the
// matrix work per pass.  Entry e of the tile is supplied by lane (wave 1, j = e) and read back as (set 1, e).
#ifndef PIR_TRACE_HALF
#define PIR_TRACE_HALF 1       // 0: A/B build without the short-list mode
#endif
constexpr int kHalfPts = 32;
__device__ __forceinline__ bool half_tiles(int count) { return PIR_TRACE_HALF && count <= (int)gridDim.x * kHalfPts; }
__device__ __forceinline__ void eval_half_tile(Lds& lds, const SdfPacked& pk, float x0, float x1, float x2) {
  forward_hidden<false, true>(lds, pk, x0, x1, x2, nullptr);
  sdf_head(lds, pk);
  __syncthreads();
}

// Wave-level compaction: lanes with `keep` append `value` to out[] (count in *counter).  Call from wave 0 with
// all 64 lanes active.
__device__ __forceinline__ void wave_append(bool keep, int value, int* __restrict__ out, int* __restrict__ counter) {
  const unsigned long long m = __ballot(keep);
  if (m == 0) return;
  const int lane = lane_id();
  const int n = __popcll(m);
  int base = 0;
  if (lane == 0) base = atomicAdd(counter, n);
  base = __shfl(base, 0);
  if (keep) out[base + __popcll(m & ((1ull << lane) - 1ull))] = value;
}

__global__ void trace_init_kernel(long N, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
                                  const float* __restrict__ min_dis, float* __restrict__ pts, float* __restrict__ dis,
                                  unsigned char* __restrict__ conv, TraceWs ws) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) ws.counters[0] = (int)N;
  if (i >= N) return;
  const float a = min_dis[i];
  dis[i] = a;
#pragma unroll
  for (int c = 0; c < 3; ++c) pts[3 * i + c] = add_rn(ray_o[3 * i + c], mul_rn(ray_d[3 * i + c], a));
  conv[i] = 0;
  ws.list[0][i] = (int)i;
  ws.min_idx[i] = -2;
}

// One sphere-tracing evaluation (reference loop body, raytracer.py:142-156): eval `it` of 0..max_iters.
__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
trace_step_kernel(SdfPacked pk, int it, int max_iters, float thr, const float* __restrict__ ray_d,
                  const float* __restrict__ max_dis, const unsigned char* __restrict__ work_mask,
                  float* __restrict__ pts, float* __restrict__ sdf, float* __restrict__ dis,
                  unsigned char* __restrict__ conv, TraceWs ws, int prof) {
  __shared__ Lds lds;
  const int count = ws.counters[it];
  if (prof && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_trace_evals[0], (unsigned long long)count);
  const int* __restrict__ lin = ws.list[it & 1];
  int* __restrict__ lout = it == max_iters ? ws.slist : ws.list[(it + 1) & 1];
  int* cout = ws.counters + it + 1;
  const int lane = lane_id(), w = wave_id(), j = lane & 31;
  const bool half = half_tiles(count);
  const int tile_pts = half ? kHalfPts : kWgPts;
  for (int base = blockIdx.x * tile_pts; base < count; base += gridDim.x * tile_pts) {
    const int e = base + (half ? 0 : 32 * (w & 1)) + j;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (e < count) {
      const int r = lin[e];
#ifdef PIR_AB_X_SOA      // A/B (timing only): the ray-state gather from three planes of the same buffer
      const int NN = ws.counters[0];
      x0 = pts[r]; x1 = pts[NN + r]; x2 = pts[2 * NN + r];
#else
      x0 = pts[3 * r]; x1 = pts[3 * r + 1]; x2 = pts[3 * r + 2];
#endif
    }
    if (half) eval_half_tile(lds, pk, x0, x1, x2); else eval_tile(lds, pk, x0, x1, x2);
    if (threadIdx.x < kWgPts) {
      const int e2 = base + threadIdx.x;
      bool keep = false;
      int r = 0;
      if (e2 < count && (int)threadIdx.x < tile_pts) {
        r = lin[e2];
        const float v = read_sdf(lds, pk, half ? 1 : threadIdx.x >> 5, threadIdx.x & 31);
        sdf[r] = v;
        const bool work = work_mask[r] != 0;
        const float a = dis[r], amax = max_dis[r];
        const bool unf = work && (fabsf(v) > thr) && (a < amax);
        if (!unf) {
          conv[r] = (work && (fabsf(v) <= thr) && (a < amax)) ? 1 : 0;
        } else {
          keep = true;
          if (it < max_iters) {
            dis[r] = add_rn(a, v);
#ifdef PIR_AB_X_SOA
            const int NN = ws.counters[0];
#pragma unroll
            for (int c = 0; c < 3; ++c) pts[c * NN + r] = add_rn(pts[c * NN + r], mul_rn(ray_d[c * NN + r], v));
#else
#pragma unroll
            for (int c = 0; c < 3; ++c) pts[3 * r + c] = add_rn(pts[3 * r + c], mul_rn(ray_d[3 * r + c], v));
#endif
          }
        }
      }
      wave_append(keep, r, lout, cout);
    }
    __syncthreads();
  }
}

// torch.linspace(0, 1, n) in fp32: start + step*i in the lower half, end - step*(n-1-i) in the upper half.
__device__ __forceinline__ float linspace01(int i, int n) {
  const float step = 1.0f / (float)(n - 1);
  return i < n / 2 ? mul_rn(step, (float)i) : 1.0f - mul_rn(step, (float)(n - 1 - i));
}

// Dense sampler (raytracer.py:77-100, :172-211): one workgroup per unfinished ray, 64 samples per pass,
// stops after the pass that holds the first negative sample (later samples cannot change the result).
__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
sampler_kernel(SdfPacked pk, const int* __restrict__ rays, int count_idx, int first_sample, int n_steps, int chunk,
               int nchunks, const float* __restrict__ ray_o,
               const float* __restrict__ ray_d, const float* __restrict__ min_dis, const float* __restrict__ max_dis,
               float* __restrict__ pts, float* __restrict__ sdf, float* __restrict__ dis,
               unsigned char* __restrict__ conv, TraceWs ws, int prof) {
  __shared__ Lds lds;
  __shared__ float s_prev_last;   // value of the last sample of the previous pass
  __shared__ int s_found;
  const int count = ws.counters[count_idx];
  const int lane = lane_id(), w = wave_id(), j = lane & 31;
  for (int q = blockIdx.x; q < count; q += gridDim.x) {
    const int r = rays[q];
    const float cur = sdf[r], a = dis[r];
    const bool outside = cur > 0.0f;
    const float lo = outside ? a : min_dis[r];
    const float hi = outside ? max_dis[r] : a;
    const float span = hi - lo;
    const float o0 = ray_o[3 * r], o1 = ray_o[3 * r + 1], o2 = ray_o[3 * r + 2];
    const float d0 = ray_d[3 * r], d1 = ray_d[3 * r + 1], d2 = ray_d[3 * r + 2];
    // first_sample > 0: the head pass already evaluated samples 0..first_sample-1 (no negative among them) and left
    // the value of the last one in f_lo[r]
    if (threadIdx.x == 0) { s_found = -1; s_prev_last = first_sample > 0 ? ws.f_lo[r] : 0.f; }
    __syncthreads();
    for (int pass = 0; first_sample + pass * kWgPts < n_steps; ++pass) {
      const int i = first_sample + pass * kWgPts + 32 * (w & 1) + j;
      float x0 = 0.f, x1 = 0.f, x2 = 0.f;
      if (i < n_steps) {
        const float z = add_rn(lo, mul_rn(linspace01(i, n_steps), span));
        x0 = add_rn(o0, mul_rn(d0, z)); x1 = add_rn(o1, mul_rn(d1, z)); x2 = add_rn(o2, mul_rn(d2, z));
      }
      eval_tile(lds, pk, x0, x1, x2);
      if (prof && threadIdx.x == 0) {
        const int left = n_steps - (first_sample + pass * kWgPts);
        atomicAdd(&g_trace_evals[1], (unsigned long long)(left < kWgPts ? left : kWgPts));
      }
      if (threadIdx.x < kWgPts) {
        const int i2 = first_sample + pass * kWgPts + threadIdx.x;
        const float v = read_sdf(lds, pk, threadIdx.x >> 5, threadIdx.x & 31);
        const unsigned long long neg = __ballot(i2 < n_steps && v < 0.0f);
        if (neg != 0) {
          const int first = __ffsll((long long)neg) - 1;
          const float v_first = __shfl(v, first);
          const float v_before = first > 0 ? __shfl(v, first - 1) : s_prev_last;
          if (threadIdx.x == 0) {
            const int idx = first_sample + pass * kWgPts + first;
            s_found = idx;
            if (idx >= 1) {
              ws.d_lo[r] = add_rn(lo, mul_rn(linspace01(idx - 1, n_steps), span));
              ws.d_hi[r] = add_rn(lo, mul_rn(linspace01(idx, n_steps), span));
              ws.f_lo[r] = v_before;
              ws.f_hi[r] = v_first;
              const bool wk = v_before > 0.0f && v_first < 0.0f;
              ws.bwork[r] = wk ? 1 : 0;
              if (wk) ws.flags[min(r / chunk, nchunks - 1)] = 1;
              const int pos = atomicAdd(ws.counters + kCntBisect, 1);
              ws.blist[pos] = r;
            }
          }
        } else if (threadIdx.x == kWgPts - 1) {
          s_prev_last = v;
        }
      }
      __syncthreads();
      if (s_found >= 0) break;
    }
    if (threadIdx.x == 0) ws.min_idx[r] = s_found;
    if (threadIdx.x == 0 && s_found < 1) {
      // no sign change (or a negative first sample): the reference returns zeros and "not convergent"
      pts[3 * r] = 0.f; pts[3 * r + 1] = 0.f; pts[3 * r + 2] = 0.f;
      sdf[r] = 0.f; dis[r] = 0.f; conv[r] = 0;
    }
    __syncthreads();
  }
}

// Head pass of the dense sampler: the first kHead samples of 8 rays per 64-point tile.  On the standard view 60 % of
// the rays that reach the sampler have their first negative sample among these (they stopped sphere tracing just short
// of the surface), so they finish after 8 evaluations instead of a 64-sample pass of their own; the others continue
// in sampler_kernel from sample kHead (list[0], counter kCntSampler2).  Same samples, same arithmetic, same
// first-sign-change rule as the one-pass form (raytracer.py:172-211).
__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
sampler_head_kernel(SdfPacked pk, int max_iters, int n_steps, int chunk, int nchunks, const float* __restrict__ ray_o,
                    const float* __restrict__ ray_d, const float* __restrict__ min_dis,
                    const float* __restrict__ max_dis, float* __restrict__ pts, float* __restrict__ sdf,
                    float* __restrict__ dis, unsigned char* __restrict__ conv, TraceWs ws, int prof) {
  __shared__ Lds lds;
  const int count = ws.counters[max_iters + 1];
  if (prof && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_trace_evals[1], (unsigned long long)count * kHead);
  const int lane = lane_id(), w = wave_id(), j = lane & 31;
  for (int base = blockIdx.x * kHeadRays; base < count; base += gridDim.x * kHeadRays) {
    {
      const int e = 32 * (w & 1) + j;                 // tile entry: ray base + e / kHead, sample e % kHead
      const int q = base + e / kHead, i = e % kHead;
      float x0 = 0.f, x1 = 0.f, x2 = 0.f;
      if (q < count) {
        const int r = ws.slist[q];
        const float cur = sdf[r], a = dis[r];
        const bool outside = cur > 0.0f;
        const float lo = outside ? a : min_dis[r];
        const float hi = outside ? max_dis[r] : a;
        const float z = add_rn(lo, mul_rn(linspace01(i, n_steps), hi - lo));
        x0 = add_rn(ray_o[3 * r], mul_rn(ray_d[3 * r], z));
        x1 = add_rn(ray_o[3 * r + 1], mul_rn(ray_d[3 * r + 1], z));
        x2 = add_rn(ray_o[3 * r + 2], mul_rn(ray_d[3 * r + 2], z));
      }
      eval_tile(lds, pk, x0, x1, x2);
    }
    if (threadIdx.x < kWgPts) {                       // wave 0: lane = tile entry
      const int e = threadIdx.x, q = base + e / kHead, i = e % kHead, lead = e - i;
      const bool valid = q < count;
      const float v = read_sdf(lds, pk, e >> 5, e & 31);
      const unsigned long long neg = __ballot(valid && v < 0.0f);
      const unsigned m = (unsigned)(neg >> lead) & ((1u << kHead) - 1u);
      const int first = m ? __ffs((int)m) - 1 : -1;
      const float v_first = __shfl(v, lead + (first > 0 ? first : 0));
      const float v_before = __shfl(v, lead + (first > 1 ? first - 1 : 0));
      const float v_last = __shfl(v, lead + kHead - 1);
      bool cont = false;
      int r = 0;
      if (valid && i == 0) {
        r = ws.slist[q];
        if (first >= 0) ws.min_idx[r] = first;
        if (first < 0) {                              // no negative sample yet: continue from sample kHead
          cont = true;
          ws.f_lo[r] = v_last;
        } else if (first >= 1) {
          const float cur = sdf[r], a = dis[r];
          const bool outside = cur > 0.0f;
          const float lo = outside ? a : min_dis[r];
          const float span = (outside ? max_dis[r] : a) - lo;
          ws.d_lo[r] = add_rn(lo, mul_rn(linspace01(first - 1, n_steps), span));
          ws.d_hi[r] = add_rn(lo, mul_rn(linspace01(first, n_steps), span));
          ws.f_lo[r] = v_before;
          ws.f_hi[r] = v_first;
          const bool wk = v_before > 0.0f && v_first < 0.0f;
          ws.bwork[r] = wk ? 1 : 0;
          if (wk) ws.flags[min(r / chunk, nchunks - 1)] = 1;
          const int pos = atomicAdd(ws.counters + kCntBisect, 1);
          ws.blist[pos] = r;
        } else {                                      // negative first sample: the reference returns zeros
          pts[3 * r] = 0.f; pts[3 * r + 1] = 0.f; pts[3 * r + 2] = 0.f;
          sdf[r] = 0.f; dis[r] = 0.f; conv[r] = 0;
        }
      }
      wave_append(cont, r, ws.list[0], ws.counters + kCntSampler2);
    }
    __syncthreads();
  }
}

// One bisection round (raytracer.py:234-247) or, with `final`, the closing evaluation (:248-250).
__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
bisect_kernel(SdfPacked pk, int it, int final_pass, int max_bisect, float thr, int chunk, int nchunks,
              const float* __restrict__ ray_o, const float* __restrict__ ray_d, float* __restrict__ pts,
              float* __restrict__ sdf, float* __restrict__ dis, unsigned char* __restrict__ conv, TraceWs ws,
              int prof) {
  __shared__ Lds lds;
  const int count = ws.counters[kCntBisect];
  const int* __restrict__ flag_in = ws.flags + it * nchunks;
  int* __restrict__ flag_out = ws.flags + (it + 1) * nchunks;
  const int lane = lane_id(), w = wave_id(), j = lane & 31;
  const bool half = half_tiles(count);
  const int tile_pts = half ? kHalfPts : kWgPts;
  for (int base = blockIdx.x * tile_pts; base < count; base += gridDim.x * tile_pts) {
    const int e = base + (half ? 0 : 32 * (w & 1)) + j;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    bool active = false;
    if (e < count) {
      const int r = ws.blist[e];
      active = final_pass || flag_in[min(r / chunk, nchunks - 1)] != 0;
      const float dm = (ws.d_lo[r] + ws.d_hi[r]) / 2.0f;
      x0 = add_rn(ray_o[3 * r], mul_rn(ray_d[3 * r], dm));
      x1 = add_rn(ray_o[3 * r + 1], mul_rn(ray_d[3 * r + 1], dm));
      x2 = add_rn(ray_o[3 * r + 2], mul_rn(ray_d[3 * r + 2], dm));
    }
    if (!__syncthreads_or(active)) continue;   // no ray of this tile is in a chunk that still iterates
    if (prof && threadIdx.x == 0)
      atomicAdd(&g_trace_evals[2], (unsigned long long)(count - base < tile_pts ? count - base : tile_pts));
    if (half) eval_half_tile(lds, pk, x0, x1, x2); else eval_tile(lds, pk, x0, x1, x2);
    if ((int)threadIdx.x < tile_pts) {
      const int e2 = base + threadIdx.x;
      if (e2 < count) {
        const int r = ws.blist[e2];
        const float v = read_sdf(lds, pk, half ? 1 : threadIdx.x >> 5, threadIdx.x & 31);
        const int ch = min(r / chunk, nchunks - 1);
        float lo = ws.d_lo[r], hi = ws.d_hi[r];
        const float dm = (lo + hi) / 2.0f;
        if (final_pass) {
          pts[3 * r] = add_rn(ray_o[3 * r], mul_rn(ray_d[3 * r], dm));
          pts[3 * r + 1] = add_rn(ray_o[3 * r + 1], mul_rn(ray_d[3 * r + 1], dm));
          pts[3 * r + 2] = add_rn(ray_o[3 * r + 2], mul_rn(ray_d[3 * r + 2], dm));
          sdf[r] = v; dis[r] = dm; conv[r] = 1;
          if (it == max_bisect && flag_in[ch] != 0) ws.counters[kCntStatus] = 1;  // iteration cap hit
        } else if (flag_in[ch] != 0) {
          if (v > 0.0f) { lo = dm; ws.d_lo[r] = dm; ws.f_lo[r] = v; }
          else { hi = dm; ws.d_hi[r] = dm; ws.f_hi[r] = v; }
          const bool wk = ws.bwork[r] != 0 && (hi - lo) > 2.0f * thr;
          ws.bwork[r] = wk ? 1 : 0;
          if (wk) flag_out[ch] = 1;
        }
      }
    }
    __syncthreads();
  }
}

// Camera.get_rays + intersect_sphere fused (raytracer.py:305-326, :253-273): uv -> rays, sphere interval.
__global__ void camera_rays_kernel(long N, int W, const float* __restrict__ uv /* null: pixel centres */,
                                   const float* __restrict__ cam /* Kinv3x3 | C2Wrot3x3 | origin3 */, float radius,
                                   float* __restrict__ ray_o, float* __restrict__ ray_d, float* __restrict__ ray_n,
                                   unsigned char* __restrict__ hit, float* __restrict__ dmin, float* __restrict__ dmax) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float u, v;
  if (uv) { u = uv[2 * i]; v = uv[2 * i + 1]; }
  else { u = (float)(i % W) + 0.5f; v = (float)(i / W) + 0.5f; }
  float p[3], d[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) p[c] = add_rn(add_rn(mul_rn(u, cam[3 * c]), mul_rn(v, cam[3 * c + 1])), cam[3 * c + 2]);
#pragma unroll
  for (int c = 0; c < 3; ++c)
    d[c] = add_rn(add_rn(mul_rn(p[0], cam[9 + 3 * c]), mul_rn(p[1], cam[9 + 3 * c + 1])), mul_rn(p[2], cam[9 + 3 * c + 2]));
  const float nrm = sqrtf(add_rn(add_rn(mul_rn(d[0], d[0]), mul_rn(d[1], d[1])), mul_rn(d[2], d[2])));
  float o[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { d[c] = d[c] / nrm; o[c] = cam[18 + c]; ray_d[3 * i + c] = d[c]; ray_o[3 * i + c] = o[c]; }
  ray_n[i] = nrm;
  const float dd = add_rn(add_rn(mul_rn(d[0], d[0]), mul_rn(d[1], d[1])), mul_rn(d[2], d[2]));
  const float od = add_rn(add_rn(mul_rn(d[0], o[0]), mul_rn(d[1], o[1])), mul_rn(d[2], o[2]));
  const float t1 = -od / dd;
  float q[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) q[c] = add_rn(o[c], mul_rn(t1, d[c]));
  const float tmp = radius * radius - add_rn(add_rn(mul_rn(q[0], q[0]), mul_rn(q[1], q[1])), mul_rn(q[2], q[2]));
  hit[i] = tmp > 0.0f ? 1 : 0;
  const float t2 = sqrtf(fmaxf(tmp, 0.0f)) / sqrtf(dd);
  dmin[i] = fmaxf(t1 - t2, 0.0f);
  dmax[i] = t1 + t2;
}

static size_t ws_layout(long N, int nchunks, int max_bisect, char* base, TraceWs* ws) {
  size_t off[pir_host::kTrArrays];
  const size_t total = pir_host::trace_ws_offsets(N, nchunks, max_bisect, off);
  if (ws && base) {
    using namespace pir_host;
    ws->counters = (int*)(base + off[kTrCounters]); ws->flags = ws->counters + kNumCounters;
    ws->list[0] = (int*)(base + off[kTrList0]); ws->list[1] = (int*)(base + off[kTrList1]);
    ws->slist = (int*)(base + off[kTrSList]); ws->blist = (int*)(base + off[kTrBList]);
    ws->d_lo = (float*)(base + off[kTrDLo]); ws->d_hi = (float*)(base + off[kTrDHi]);
    ws->f_lo = (float*)(base + off[kTrFLo]); ws->f_hi = (float*)(base + off[kTrFHi]);
    ws->bwork = (unsigned char*)(base + off[kTrBWork]); ws->min_idx = (int*)(base + off[kTrMinIdx]);
  }
  return total;
}
static_assert(kNumCounters == pir_host::kTraceCounters, "host_logic.h and trace.hip disagree on the header size");

}  // namespace pir

using namespace pir;

extern "C" {

size_t pir_trace_workspace_bytes(long n_rays, long chunk, int max_bisect) {
  if (n_rays <= 0) return 256;
  return ws_layout(n_rays, pir_host::trace_chunks(n_rays, chunk), max_bisect, nullptr, nullptr);
}

// Enqueues the whole tracer on `stream` over a workspace laid out by ws_layout at `workspace` (arguments checked by
// the callers).
static int trace_enqueue(const void* packed, long n_rays, const float* ray_o, const float* ray_d, const float* min_dis,
                         const float* max_dis, const unsigned char* work_mask, float sdf_threshold,
                         int sphere_tracing_iters, int n_steps, long chunk, int nchunks, int max_bisect,
                         unsigned char* convergent, float* points, float* sdf, float* distance, char* workspace,
                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TraceWs ws;
  ws_layout(n_rays, nchunks, max_bisect, workspace, &ws);
  SdfPacked pk;
  pk.mats = (const f32x4*)packed;
  pk.vecs = (const float*)packed + (size_t)kPackedF4 * 4;
  if (hipMemsetAsync(ws.counters, 0, sizeof(int) * kNumCounters + sizeof(int) * (size_t)(max_bisect + 2) * nchunks,
                     stream) != hipSuccess)
    return PIR_ERR_LAUNCH;
  const int N = (int)n_rays;
  const int prof_on = pir_prof_enabled();
  hipLaunchKernelGGL(trace_init_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, n_rays, ray_o, ray_d, min_dis,
                     points, distance, convergent, ws);
  // enough workgroups for one 32-entry tile each (the kernels switch to 64-entry tiles when the list is longer)
  const long tiles = PIR_TRACE_HALF ? (n_rays + kHalfPts - 1) / kHalfPts : (n_rays + kWgPts - 1) / kWgPts;
  const long resident = 2L * pir_num_cus();
  const unsigned grid = (unsigned)(tiles < resident ? tiles : resident);
  for (int it = 0; it <= sphere_tracing_iters; ++it) {
    PirProfScope prof(PIR_K_TRACE_STEP, 0.0, stream_);
    hipLaunchKernelGGL(trace_step_kernel, dim3(grid), dim3(kThreads), 0, stream, pk, it, sphere_tracing_iters,
                       sdf_threshold, ray_d, max_dis, work_mask, points, sdf, distance, convergent, ws, prof_on);
  }
  const unsigned sgrid = (unsigned)(n_rays < resident ? n_rays : resident);
#ifdef PIR_SAMPLER_ONEPASS     // A/B: the round-1 form, one workgroup per ray from sample 0
  const bool head = false;
#else
  const bool head = n_steps > kHead;
#endif
  if (head) {
    const long htiles = (n_rays + kHeadRays - 1) / kHeadRays;
    PirProfScope prof(PIR_K_SAMPLER, 0.0, stream_);
    hipLaunchKernelGGL(sampler_head_kernel, dim3((unsigned)(htiles < resident ? htiles : resident)), dim3(kThreads), 0,
                       stream, pk, sphere_tracing_iters, n_steps, (int)chunk, nchunks, ray_o, ray_d, min_dis, max_dis,
                       points, sdf, distance, convergent, ws, prof_on);
  }
  {
  PirProfScope prof(PIR_K_SAMPLER, 0.0, stream_);
  hipLaunchKernelGGL(sampler_kernel, dim3(sgrid), dim3(kThreads), 0, stream, pk,
                     (const int*)(head ? ws.list[0] : ws.slist), head ? kCntSampler2 : sphere_tracing_iters + 1,
                     head ? kHead : 0, n_steps, (int)chunk, nchunks, ray_o, ray_d, min_dis, max_dis, points, sdf,
                     distance, convergent, ws, prof_on);
  }
  for (int it = 0; it <= max_bisect; ++it) {
    PirProfScope prof(PIR_K_BISECT, 0.0, stream_);
    hipLaunchKernelGGL(bisect_kernel, dim3(grid), dim3(kThreads), 0, stream, pk, it, it == max_bisect ? 1 : 0,
                       max_bisect, sdf_threshold, (int)chunk, nchunks, ray_o, ray_d, points, sdf, distance, convergent,
                       ws, prof_on);
  }
  return pir_check_launch();
}

using pir_host::trace_check_args;

int pir_trace(const void* packed, long n_rays, const float* ray_o, const float* ray_d, const float* min_dis,
              const float* max_dis, const unsigned char* work_mask, float sdf_threshold, int sphere_tracing_iters,
              int n_steps, long chunk, int max_bisect, unsigned char* convergent, float* points, float* sdf,
              float* distance, void* workspace, size_t workspace_bytes, void* stream_) {
  const int rc = trace_check_args(packed, n_rays, ray_o, ray_d, min_dis, max_dis, work_mask, sphere_tracing_iters,
                                  n_steps, max_bisect, convergent, points, sdf, distance);
  if (rc != PIR_OK || n_rays == 0) return rc;
  if (!workspace) return PIR_ERR_ARG;
  if (chunk <= 0) chunk = n_rays;
  const int nchunks = (int)((n_rays + chunk - 1) / chunk);
  if (workspace_bytes < ws_layout(n_rays, nchunks, max_bisect, nullptr, nullptr)) return PIR_ERR_SCRATCH;
  return trace_enqueue(packed, n_rays, ray_o, ray_d, min_dis, max_dis, work_mask, sdf_threshold, sphere_tracing_iters,
                       n_steps, chunk, nchunks, max_bisect, convergent, points, sdf, distance, (char*)workspace, stream_);
}

// ---- allocation-free variant ----------------------------------------------------------------------------------
// The tracer workspace (active lists, bisection brackets, counters) of pir_trace_static lives in this file-scope
// device array instead of caller memory.  Every GPU the library runs on holds its own copy at its own address; the
// host-side name below is only a handle for hipGetSymbolAddress, never a kernel argument.
constexpr long kStaticMaxRays = PIR_TRACE_STATIC_MAX_RAYS;
constexpr int kStaticMaxChunks = 64;
constexpr size_t kStaticHeader = (sizeof(int) * (kNumCounters + (64 + 2) * kStaticMaxChunks) + 255) & ~(size_t)255;
constexpr size_t kStaticBytes = kStaticHeader + 9 * sizeof(int) * kStaticMaxRays + kStaticMaxRays;   // 2 441 984
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

static std::mutex g_static_mu;
static char* g_static_ptr[64] = {nullptr};   // per device: address of that device's copy of `scratch`

// Device address of the current device's `scratch`, looked up once per device.
// The kernels run on the device that owns `stream`; that must be the current device (NULL stream: it is by definition),
// otherwise the array of another GPU would be handed to them.
static char* static_workspace(hipStream_t stream, int* err) {
  int dev = -1;
  *err = PIR_ERR_LAUNCH;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (stream) {
    hipDevice_t sdev = -1;
    if (hipStreamGetDevice(stream, &sdev) != hipSuccess) return nullptr;
    if ((int)sdev != dev) { *err = PIR_ERR_ARG; return nullptr; }
  }
  std::lock_guard<std::mutex> lk(g_static_mu);
  if (!g_static_ptr[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return nullptr;
    g_static_ptr[dev] = (char*)p;
  }
  return g_static_ptr[dev];
}

int pir_trace_static(const void* packed, long n_rays, const float* ray_o, const float* ray_d, const float* min_dis,
                     const float* max_dis, const unsigned char* work_mask, float sdf_threshold,
                     int sphere_tracing_iters, int n_steps, long chunk, int max_bisect, unsigned char* convergent,
                     float* points, float* sdf, float* distance, int* status, void* stream_) {
  const int rc = trace_check_args(packed, n_rays, ray_o, ray_d, min_dis, max_dis, work_mask, sphere_tracing_iters,
                                  n_steps, max_bisect, convergent, points, sdf, distance);
  if (rc != PIR_OK) return rc;
  if (n_rays > kStaticMaxRays) return PIR_ERR_TOO_LARGE;
  if (n_rays == 0) {
    if (status && hipMemsetAsync(status, 0, sizeof(int), (hipStream_t)stream_) != hipSuccess) return PIR_ERR_LAUNCH;
    return PIR_OK;
  }
  if (chunk <= 0) chunk = n_rays;
  const int nchunks = (int)((n_rays + chunk - 1) / chunk);
  if (nchunks > kStaticMaxChunks) return PIR_ERR_TOO_LARGE;
  if (ws_layout(n_rays, nchunks, max_bisect, nullptr, nullptr) > kStaticBytes) return PIR_ERR_TOO_LARGE;
  int ws_err = PIR_OK;
  char* ws = static_workspace((hipStream_t)stream_, &ws_err);
  if (!ws) return ws_err;
  const int rc2 = trace_enqueue(packed, n_rays, ray_o, ray_d, min_dis, max_dis, work_mask, sdf_threshold,
                                sphere_tracing_iters, n_steps, chunk, nchunks, max_bisect, convergent, points, sdf,
                                distance, ws, stream_);
  if (rc2 != PIR_OK) return rc2;
  if (status && hipMemcpyAsync(status, (const int*)ws + kCntStatus, sizeof(int), hipMemcpyDeviceToDevice,
                               (hipStream_t)stream_) != hipSuccess)
    return PIR_ERR_LAUNCH;
  return PIR_OK;
}

int pir_trace_sampler_index(const void* workspace, long n_rays, long chunk, int max_bisect, int* min_idx,
                            void* stream_) {
  if (n_rays < 0 || max_bisect < 1 || max_bisect > 64) return PIR_ERR_ARG;
  if (n_rays == 0) return PIR_OK;
  if (!workspace || !min_idx) return PIR_ERR_ARG;
  if (chunk <= 0) chunk = n_rays;
  TraceWs ws;
  ws_layout(n_rays, (int)((n_rays + chunk - 1) / chunk), max_bisect, (char*)workspace, &ws);
  return hipMemcpyAsync(min_idx, ws.min_idx, sizeof(int) * (size_t)n_rays, hipMemcpyDeviceToDevice,
                        (hipStream_t)stream_) == hipSuccess ? PIR_OK : PIR_ERR_LAUNCH;
}

// internal (prof.hip): evaluations executed by the tracer kernels of the current device since the last call;
// synchronises the device.
int pir_trace_profile_read(double* evals3) {
  unsigned long long h[4] = {0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return PIR_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace_evals), sizeof(h)) != hipSuccess) return PIR_ERR_LAUNCH;
  for (int i = 0; i < 3; ++i) evals3[i] = (double)h[i];
  const unsigned long long zero[4] = {0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_evals), zero, sizeof(zero)) == hipSuccess ? PIR_OK : PIR_ERR_LAUNCH;
}

int pir_trace_status(const void* workspace, int* host_status, void* stream_) {
  if (!workspace || !host_status) return PIR_ERR_ARG;
  const int* counters = (const int*)workspace;
  if (hipMemcpyAsync(host_status, counters + kCntStatus, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream_) !=
      hipSuccess)
    return PIR_ERR_LAUNCH;
  return hipStreamSynchronize((hipStream_t)stream_) == hipSuccess ? PIR_OK : PIR_ERR_LAUNCH;
}

int pir_camera_rays(long n, int width, const float* uv, const float* cam21, float radius, float* ray_o, float* ray_d,
                    float* ray_d_norm, unsigned char* hit, float* min_dis, float* max_dis, void* stream_) {
  if (n < 0 || width <= 0) return PIR_ERR_ARG;
  if (n == 0) return PIR_OK;
  if (!cam21 || !ray_o || !ray_d || !ray_d_norm || !hit || !min_dis || !max_dis) return PIR_ERR_ARG;
  hipLaunchKernelGGL(camera_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, n,
                     width, uv, cam21, radius, ray_o, ray_d, ray_d_norm, hit, min_dis, max_dis);
  return pir_check_launch();
}

}  // extern "C"
