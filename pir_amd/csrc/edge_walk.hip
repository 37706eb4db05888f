// IRON's surface walk to the silhouette (locate_edge_points, src/renderer/raytracer.py:497-560) as ONE launch:
// every workgroup owns 64 walk-start points and iterates them itself -- get_all (value + normal through the same
// sweeps as pir_sdf_get_all), the test |n_hat . view_hat| <= dot_threshold, the step along n_hat - view_hat / dot
// re-normalised and pulled back to the zero level set by sdf * n_hat -- until none of its points is still walking or
// max_step steps were taken.  The reference (and round 2 of this repo) ran one get_all launch and about thirty
// elementwise launches per step, with a host synchronisation each (the loop's `.sum() == 0` test).  Points evolve
// independently, so evaluating the finished points of a tile along with the walking ones changes nothing; the
// per-point arithmetic uses un-contracted mul/add in the order of the eager expressions it replaces.
#include "sdf_sweeps.cuh"
#include "pir_common.h"

namespace pir {

__device__ __forceinline__ float ew_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float ew_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float ew_norm3(float a, float b, float c) {
  return sqrtf(ew_add(ew_add(ew_mul(a, a), ew_mul(b, b)), ew_mul(c, c)));
}

// HALF: 32 start points per workgroup in operand set 1 only (set 0's products are not issued): the walk is a chain of up
// to 17 dependent forward + reverse passes, i.e. pure latency for the few dozen edge points of a training patch, and a
// pass over one operand set is half the matrix work.  Chosen by the host when 32-point tiles fit the resident slots.
template <bool HALF>
__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
edge_walk_kernel(SdfPacked pk, long n, float* __restrict__ points, const float* __restrict__ cam_o, int max_step,
                 float step_size, float dot_threshold, unsigned char* __restrict__ found,
                 f32x4* __restrict__ scratch_all) {
  __shared__ Lds lds;
  constexpr int kPts = HALF ? 32 : kWgPts;
  __shared__ float s_x[kWgPts][3];
  __shared__ int s_state[kWgPts];     // 0 walking, 1 found, 2 padding
  const int lane = lane_id(), w = wave_id(), j = lane & 31;
  f32x4* scratch = scratch_all + (size_t)blockIdx.x * kScratchWgF4;
  const long ntiles = (n + kPts - 1) / kPts;
  const float o0 = cam_o[0], o1 = cam_o[1], o2 = cam_o[2];
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long base = tile * kPts;
    if (threadIdx.x < kPts) {
      const long q = base + threadIdx.x;
      const bool valid = q < n;
#pragma unroll
      for (int c = 0; c < 3; ++c) s_x[threadIdx.x][c] = valid ? points[3 * q + c] : 0.f;
      s_state[threadIdx.x] = valid ? 0 : 2;
    }
    __syncthreads();
    for (int it = 0;; ++it) {
      const int e = HALF ? j : 32 * (w & 1) + j;          // HALF: point j lives in set 1 (encoded by wave 1)
      forward_hidden<true, HALF>(lds, pk, s_x[e][0], s_x[e][1], s_x[e][2], scratch);
      sdf_head(lds, pk);
      __syncthreads();
      float val = 0.f;
      if (threadIdx.x < kPts) val = read_sdf(lds, pk, HALF ? 1 : threadIdx.x >> 5, threadIdx.x & 31);
      normal_reverse<HALF>(lds, pk, scratch);
      __syncthreads();
      bool walking = false;
      if (threadIdx.x < kPts && s_state[threadIdx.x] == 0) {
        const int t = threadIdx.x, set = HALF ? 1 : t >> 5;
        const float x0 = s_x[t][0], x1 = s_x[t][1], x2 = s_x[t][2];
        float v0 = o0 - x0, v1 = o1 - x1, v2 = o2 - x2;
        const float vn = ew_norm3(v0, v1, v2) + 1e-10f;
        v0 /= vn; v1 /= vn; v2 /= vn;
        float n0 = read_normal(lds, set, t & 31, 0), n1 = read_normal(lds, set, t & 31, 1),
              n2 = read_normal(lds, set, t & 31, 2);
        const float nn = ew_norm3(n0, n1, n2) + 1e-10f;
        n0 /= nn; n1 /= nn; n2 /= nn;
        const float dot = ew_add(ew_add(ew_mul(n0, v0), ew_mul(n1, v1)), ew_mul(n2, v2));
        walking = fabsf(dot) > dot_threshold;
        if (!walking) {
          s_state[t] = 1;
        } else if (it < max_step) {
          float d0 = n0 - v0 / dot, d1 = n1 - v1 / dot, d2 = n2 - v2 / dot;
          const float dn = ew_norm3(d0, d1, d2) + 1e-10f;
          d0 = d0 / dn - ew_mul(val, n0);
          d1 = d1 / dn - ew_mul(val, n1);
          d2 = d2 / dn - ew_mul(val, n2);
          s_x[t][0] = ew_add(x0, ew_mul(step_size, d0));
          s_x[t][1] = ew_add(x1, ew_mul(step_size, d1));
          s_x[t][2] = ew_add(x2, ew_mul(step_size, d2));
        }
      }
      // the reference leaves the loop after the test of step max_step, or when nothing walks any more
      const int any = __syncthreads_or(walking ? 1 : 0);
      if (it >= max_step || !any) break;
    }
    if (threadIdx.x < kPts) {
      const long q = base + threadIdx.x;
      if (q < n) {
#pragma unroll
        for (int c = 0; c < 3; ++c) points[3 * q + c] = s_x[threadIdx.x][c];
        found[q] = s_state[threadIdx.x] == 1 ? 1 : 0;
      }
    }
    __syncthreads();
  }
}

}  // namespace pir

using namespace pir;

extern "C" int pir_edge_walk(const void* packed, long n, float* points, const float* cam_o, int max_step,
                             float step_size, float dot_threshold, unsigned char* found, void* scratch,
                             size_t scratch_bytes, void* stream) {
  if (n < 0 || max_step < 0 || max_step > 4096) return PIR_ERR_ARG;
  if (n == 0) return PIR_OK;
  if (!packed || !points || !cam_o || !found || !scratch) return PIR_ERR_ARG;
  long nwg = (long)(scratch_bytes / kScratchWgBytes);
  if (nwg < 1) return PIR_ERR_SCRATCH;
  const long resident = 2L * pir_num_cus();
  if (nwg > resident) nwg = resident;
  const bool half = (n + 31) / 32 <= (nwg < pir_num_cus() ? nwg : pir_num_cus());   // one 32-point tile per CU covers the list
  const long ntiles = half ? (n + 31) / 32 : (n + kWgPts - 1) / kWgPts;
  if (nwg > ntiles) nwg = ntiles;
  SdfPacked pk;
  pk.mats = (const f32x4*)packed;
  pk.vecs = (const float*)packed + (size_t)kPackedF4 * 4;
  if (half)
    hipLaunchKernelGGL(edge_walk_kernel<true>, dim3((unsigned)nwg), dim3(kThreads), 0, (hipStream_t)stream, pk, n, points,
                       cam_o, max_step, step_size, dot_threshold, found, (f32x4*)scratch);
  else
    hipLaunchKernelGGL(edge_walk_kernel<false>, dim3((unsigned)nwg), dim3(kThreads), 0, (hipStream_t)stream, pk, n, points,
                       cam_o, max_step, step_size, dot_threshold, found, (f32x4*)scratch);
  return pir_check_launch();
}
