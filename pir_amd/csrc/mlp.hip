// Fused ReLU MLPs of the material side of the path: RenderingNetwork x3, DINONetwork and
// BlendNetwork-shaped nets of the reference (src/network.py:146-267, :382-467) -- any MLP whose hidden layers are
// 256 wide with ReLU, an input of arbitrary width that may be concatenated again at one skip layer
// ([h | x] / sqrt2), and a linear output layer.  Same machinery as the SDF engine (sdf_core.cuh): a workgroup
// of 4 waves pushes two operand sets of 32 points through all layers with the activations resident in LDS and the
// packed weights streamed from L2 on v_mfma_f32_32x32x2_f32; the input rows stream through the activation region
// in slices of 256 columns.
#include "sdf_core.cuh"
#include "sdf_sweeps.cuh"
#include "pir_common.h"
#include <string.h>

namespace pir {

// ---- packing: one product = [nkb][8 tiles][64 lanes] f32x4, k-step 4 kb + c of lane (i, h) is column 8 kb + 4 h + c
// of the product's K range, row 32 T + i of its output range.
__global__ void mlp_pack_product_kernel(const float* __restrict__ W, int ld, int row0, int nrows, int col0, int ncols,
                                        int transposed, float scale, float* __restrict__ dst, int nkb) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)nkb * 8 * 64 * 4) return;
  const int c = idx & 3, lane = (idx >> 2) & 63, T = (idx >> 8) & 7, kb = (int)(idx >> 11);
  const int r = 32 * T + (lane & 31), k = 8 * kb + 4 * (lane >> 5) + c;
  float v = 0.f;
  if (r < nrows && k < ncols)
    v = (transposed ? W[(size_t)(row0 + k) * ld + col0 + r] : W[(size_t)(row0 + r) * ld + col0 + k]) * scale;
  dst[idx] = v;
}

struct MlpPlan {
  int kin_kb;        // k-blocks of the (zero-padded) input: rows of X are 8 * kin_kb floats
  int n_hidden;      // 256-wide ReLU layers, 1..8
  int skip;          // hidden layer that also reads the input ([h | x] / sqrt2, scale folded into the pack), or -1
  int n_out;         // outputs of the linear last layer: <= 4 -> VALU heads, else MFMA tiles in passes of 256
  int prod_in;       // f32x4 offsets of the packed products: lin0 over x (kin_kb k-blocks)
  int prod_h[8];     //   hidden layer l over the previous activations (32 k-blocks), l = 1..n_hidden-1
  int prod_skip_x;   //   the skip layer over x (kin_kb k-blocks)
  int prod_out;      //   output layer, ceil(n_out / 256) passes of 32 k-blocks (n_out > 4 only)
  int bias_off;      // float offsets into the vector region: hidden biases [n_hidden][256]
  int head_off;      //   output rows [n_out][256] (n_out <= 4 only)
  int out_bias_off;  //   output bias, padded to 4 or to 256 * passes
};

// x-slice of both operand sets -> activation region, then acc += W[:, slice] x slice, for every slice of the input.
// Leaves with all waves past their last read of the final slice NOT guaranteed: callers synchronise before they
// overwrite the activation region.
__device__ __forceinline__ void input_gemm(Lds& lds, const f32x4* __restrict__ mats, int prod_off, int kin_kb,
                                           const float* __restrict__ X, long base, long P,
                                           f32x16 (&acc)[kSets][kTPW]) {
  const int lane = lane_id(), w = wave_id(), j = lane & 31, h = lane >> 5;
  const size_t ldx = (size_t)8 * kin_kb;
  for (int c0 = 0; c0 < kin_kb; c0 += kActKB) {
    const int n = kin_kb - c0 < kActKB ? kin_kb - c0 : kActKB;
    __syncthreads();   // every read of the previous content of the activation region is done
    for (int kb = w; kb < n; kb += kWaves) {
#pragma unroll
      for (int s = 0; s < kSets; ++s) {
        const long Pt = base + 32 * s + j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (Pt < P) v = *reinterpret_cast<const f32x4*>(X + (size_t)Pt * ldx + 8 * (c0 + kb) + 4 * h);
        lds.act[s][kb][lane] = v;
      }
    }
    __syncthreads();
    gemm_phase<kTPW>(mats, Prod{prod_off + c0 * 8 * kTileKB, n, 8}, kTPW * w, lds, 0, acc);
  }
}

// a = relu(acc + bias) -> LDS activation region of both sets; optionally also to a [row][256] array.
template <bool STORE>
__device__ __forceinline__ void epilogue_relu(Lds& lds, f32x16 (&acc)[kSets][kTPW], const float* __restrict__ bias,
                                              float* __restrict__ H, long row0, long nrows) {
  const int lane = lane_id(), h = lane >> 5, w = wave_id(), j = lane & 31;
#pragma unroll
  for (int t = 0; t < kTPW; ++t) {
    const int T = kTPW * w + t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = load_feat4(bias, T, g, h);
#pragma unroll
      for (int s = 0; s < kSets; ++s) {
        f32x4 a;
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = fmaxf(acc[s][t][4 * g + c] + b[c], 0.f);
        lds.act[s][4 * T + g][lane] = a;
        if (STORE) {
          const long r = row0 + 32 * s + j;
          if (r < nrows) *reinterpret_cast<f32x4*>(H + (size_t)r * 256 + 32 * T + 8 * g + 4 * h) = a;
        }
      }
    }
  }
}

// All hidden layers of one 64-point tile; leaves the last activations in LDS (all waves synchronised).
// STORE: activations of layer l also go to Hs + l * h_stride ([row][256], rows relative to row0 = base - chunk_base).
template <bool STORE>
__device__ __forceinline__ void mlp_hidden(Lds& lds, const f32x4* __restrict__ mats, const float* __restrict__ vecs,
                                           const MlpPlan& pl, const float* __restrict__ X, long base, long P,
                                           float* __restrict__ Hs, size_t h_stride, long row0, long nrows) {
  const int w = wave_id();
  f32x16 acc[kSets][kTPW];
  zero_acc(acc);
  input_gemm(lds, mats, pl.prod_in, pl.kin_kb, X, base, P, acc);
  __syncthreads();
  epilogue_relu<STORE>(lds, acc, vecs + pl.bias_off, Hs, row0, nrows);
  __syncthreads();
#pragma nounroll
  for (int l = 1; l < pl.n_hidden; ++l) {
    zero_acc(acc);
    gemm_phase<kTPW>(mats, Prod{pl.prod_h[l], kActKB, 8}, kTPW * w, lds, 0, acc);
    if (l == pl.skip) input_gemm(lds, mats, pl.prod_skip_x, pl.kin_kb, X, base, P, acc);
    __syncthreads();
    epilogue_relu<STORE>(lds, acc, vecs + pl.bias_off + 256 * l, Hs + (size_t)l * h_stride, row0, nrows);
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
mlp_eval_kernel(const f32x4* __restrict__ mats, const float* __restrict__ vecs, MlpPlan pl,
                const float* __restrict__ X, long P, float* __restrict__ out) {
  __shared__ Lds lds;
  const int lane = lane_id(), w = wave_id(), j = lane & 31, h = lane >> 5;
  const long ntiles = (P + kWgPts - 1) / kWgPts;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long base = tile * kWgPts;
    mlp_hidden<false>(lds, mats, vecs, pl, X, base, P, nullptr, 0, 0, 0);
    if (pl.n_out <= 4) {
      for (int o = 0; o < pl.n_out; ++o) {
        float part[kSets];
        head_partial(lds, vecs + pl.head_off + 256 * o, part);
        if (lane < 32) { lds.red[w][0][lane][o] = part[0]; lds.red[w][1][lane][o] = part[1]; }
      }
      __syncthreads();
      if (threadIdx.x < kWgPts) {
        const long Q = base + threadIdx.x;
        if (Q < P) {
          for (int o = 0; o < pl.n_out; ++o) {
            float s = vecs[pl.out_bias_off + o];
#pragma unroll
            for (int ww = 0; ww < kWaves; ++ww) s += lds.red[ww][threadIdx.x >> 5][threadIdx.x & 31][o];
            out[Q * pl.n_out + o] = s;
          }
        }
      }
    } else {
      const int passes = (pl.n_out + 255) / 256;
      for (int p = 0; p < passes; ++p) {
        f32x16 acc[kSets][kTPW];
        zero_acc(acc);
        gemm_phase<kTPW>(mats, Prod{pl.prod_out + p * kActKB * 8 * kTileKB, kActKB, 8}, kTPW * w, lds, 0, acc);
        const float* bias = vecs + pl.out_bias_off + 256 * p;
#pragma unroll
        for (int s = 0; s < kSets; ++s) {
          const long Q = base + 32 * s + j;
#pragma unroll
          for (int t = 0; t < kTPW; ++t) {
            const int T = kTPW * w + t;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int col = 256 * p + 32 * T + 8 * g + 4 * h;   // n_out is a multiple of 4
              if (Q < P && col < pl.n_out) {
                const f32x4 b = load_feat4(bias, T, g, h);
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = acc[s][t][4 * g + c] + b[c];
                *reinterpret_cast<f32x4*>(out + (size_t)Q * pl.n_out + col) = o;
              }
            }
          }
        }
      }
    }
    // the next tile's first slice load synchronises before it overwrites the activation region; the reduction
    // slots are only rewritten after that barrier too
  }
}

static bool plan_ok(const MlpPlan& pl) {
  if (pl.kin_kb < 1 || pl.n_hidden < 1 || pl.n_hidden > 8 || pl.n_out < 1) return false;
  if (pl.skip != -1 && (pl.skip < 1 || pl.skip >= pl.n_hidden)) return false;
  if (pl.n_out > 4 && (pl.n_out & 3)) return false;
  return true;
}

}  // namespace pir

using namespace pir;

extern "C" {

size_t pir_mlp_product_floats(int n_kblocks) { return (size_t)n_kblocks * 8 * kTileKB * 4; }

int pir_mlp_pack_product(const float* W, int ld, int row0, int nrows, int col0, int ncols, int transposed, float scale,
                         float* dst, int n_kblocks, void* stream) {
  if (!W || !dst || ld < 1 || n_kblocks < 1 || nrows < 0 || ncols < 0 || nrows > 256 || ncols > 8 * n_kblocks)
    return PIR_ERR_ARG;
  const long n = (long)n_kblocks * 8 * 64 * 4;
  hipLaunchKernelGGL(mlp_pack_product_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ld,
                     row0, nrows, col0, ncols, transposed, scale, dst, n_kblocks);
  return pir_check_launch();
}

int pir_mlp_forward(const void* packed, size_t mats_floats, const PirMlpPlan* plan, const float* X, long P, float* out,
                    void* stream) {
  static_assert(sizeof(PirMlpPlan) == sizeof(MlpPlan), "PirMlpPlan and the kernel's MlpPlan must agree");
  if (!packed || !plan || P < 0 || (mats_floats & 3)) return PIR_ERR_ARG;
  MlpPlan pl;
  memcpy(&pl, plan, sizeof(pl));
  if (!plan_ok(pl)) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!X || !out) return PIR_ERR_ARG;
  const f32x4* mats = (const f32x4*)packed;
  const float* vecs = (const float*)packed + mats_floats;
  const long ntiles = (P + kWgPts - 1) / kWgPts;
  const long resident = 2L * pir_num_cus();
  PirProfScope prof(PIR_K_MLP_FWD, (double)P, stream);
  hipLaunchKernelGGL(mlp_eval_kernel, dim3((unsigned)(ntiles < resident ? ntiles : resident)), dim3(kThreads), 0,
                     (hipStream_t)stream, mats, vecs, pl, X, P, out);
  return pir_check_launch();
}

}  // extern "C"
