// Fused ReLU MLPs of the material side of the path: RenderingNetwork x3, DINONetwork and
// BlendNetwork-shaped nets of the reference (src/network.py:146-267, :382-467) -- any MLP whose hidden layers are
// 256 wide with ReLU, an input of arbitrary width that may be concatenated again at one skip layer
// ([h | x] / sqrt2), and a linear output layer.  Same machinery as the SDF engine (sdf_core.cuh): a workgroup
// of 4 waves pushes two operand sets of 32 points through all layers with the activations resident in LDS and the
// packed weights streamed from L2 on v_mfma_f32_32x32x2_f32; the input rows stream through the activation region
// in slices of 256 columns.
#include "sdf_core.cuh"
#include "sdf_sweeps.cuh"
#include "wgrad.h"
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
  // transposed products of the backward (0 when the network was packed for inference only)
  int has_bwd;
  int rprod_h[8];    //   W_l^T over the 256 outputs of hidden layer l (32 k-blocks), l = 1..n_hidden-1
  int rprod_in;      //   W_0^T: ceil(kin_kb / 32) passes of 256 input columns, 32 k-blocks each
  int rprod_skip_x;  //   the x part of the skip layer, transposed: same shape as rprod_in
  int rprod_out;     //   W_out^T over the n_out outputs (n_out / 8 k-blocks; n_out > 4 only, n_out % 8 == 0)
};

// Short calls (at most one 32-point tile per CU: the hit points of a 64 x 64 patch): HALF puts 32 points per workgroup
// into operand set 1 only and does not issue set 0's products -- twice the CUs busy, half the matrix work per tile.
// Same arithmetic per point, so the results do not depend on the mode.
#ifndef PIR_MLP_HALF
#define PIR_MLP_HALF 1         // 0: A/B build with 64-point tiles only
#endif
template <bool HALF> constexpr int kS0 = HALF ? 1 : 0;           // first operand set in use
template <bool HALF> constexpr int kPts = HALF ? 32 : kWgPts;    // points per tile
template <bool HALF> __device__ __forceinline__ int tile_pt(int s, int j) { return HALF ? j : 32 * s + j; }

// x-slice of both operand sets -> activation region, then acc += W[:, slice] x slice, for every slice of the input.
// Leaves with all waves past their last read of the final slice NOT guaranteed: callers synchronise before they
// overwrite the activation region.
template <bool HALF>
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
      for (int s = kS0<HALF>; s < kSets; ++s) {
        const long Pt = base + tile_pt<HALF>(s, j);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (Pt < P) v = *reinterpret_cast<const f32x4*>(X + (size_t)Pt * ldx + 8 * (c0 + kb) + 4 * h);
        lds.act[s][kb][lane] = v;
      }
    }
    __syncthreads();
    gemm_phase<kTPW, HALF>(mats, Prod{prod_off + c0 * 8 * kTileKB, n, 8}, kTPW * w, lds, 0, acc);
  }
}

// a = relu(acc + bias) -> LDS activation region of both sets; optionally also to a [row][256] array.
template <bool STORE, bool HALF>
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
      for (int s = kS0<HALF>; s < kSets; ++s) {
        f32x4 a;
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = fmaxf(acc[s][t][4 * g + c] + b[c], 0.f);
        lds.act[s][4 * T + g][lane] = a;
        if (STORE) {
          const long r = row0 + tile_pt<HALF>(s, j);
          if (r < nrows) *reinterpret_cast<f32x4*>(H + (size_t)r * 256 + 32 * T + 8 * g + 4 * h) = a;
        }
      }
    }
  }
}

// All hidden layers of one tile; leaves the last activations in LDS (all waves synchronised).
// STORE: activations of layer l also go to Hs + l * h_stride ([row][256], rows relative to row0 = base - chunk_base).
template <bool STORE, bool HALF>
__device__ __forceinline__ void mlp_hidden(Lds& lds, const f32x4* __restrict__ mats, const float* __restrict__ vecs,
                                           const MlpPlan& pl, const float* __restrict__ X, long base, long P,
                                           float* __restrict__ Hs, size_t h_stride, long row0, long nrows) {
  const int w = wave_id();
  f32x16 acc[kSets][kTPW];
  zero_acc(acc);
  input_gemm<HALF>(lds, mats, pl.prod_in, pl.kin_kb, X, base, P, acc);
  __syncthreads();
  epilogue_relu<STORE, HALF>(lds, acc, vecs + pl.bias_off, Hs, row0, nrows);
  __syncthreads();
#pragma nounroll
  for (int l = 1; l < pl.n_hidden; ++l) {
    zero_acc(acc);
    gemm_phase<kTPW, HALF>(mats, Prod{pl.prod_h[l], kActKB, 8}, kTPW * w, lds, 0, acc);
    if (l == pl.skip) input_gemm<HALF>(lds, mats, pl.prod_skip_x, pl.kin_kb, X, base, P, acc);
    __syncthreads();
    epilogue_relu<STORE, HALF>(lds, acc, vecs + pl.bias_off + 256 * l, Hs + (size_t)l * h_stride, row0, nrows);
    __syncthreads();
  }
}

template <bool HALF>
__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
mlp_eval_kernel(const f32x4* __restrict__ mats, const float* __restrict__ vecs, MlpPlan pl,
                const float* __restrict__ X, long P, float* __restrict__ out) {
  __shared__ Lds lds;
  const int lane = lane_id(), w = wave_id(), j = lane & 31, h = lane >> 5;
  const long ntiles = (P + kPts<HALF> - 1) / kPts<HALF>;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long base = tile * kPts<HALF>;
    mlp_hidden<false, HALF>(lds, mats, vecs, pl, X, base, P, nullptr, 0, 0, 0);
    if (pl.n_out <= 4) {
      for (int o = 0; o < pl.n_out; ++o) {
        float part[kSets];
        head_partial(lds, vecs + pl.head_off + 256 * o, part);
        if (lane < 32) { lds.red[w][0][lane][o] = part[0]; lds.red[w][1][lane][o] = part[1]; }
      }
      __syncthreads();
      if (threadIdx.x < kPts<HALF>) {
        const long Q = base + threadIdx.x;
        const int set = HALF ? 1 : threadIdx.x >> 5;
        if (Q < P) {
          for (int o = 0; o < pl.n_out; ++o) {
            float s = vecs[pl.out_bias_off + o];
#pragma unroll
            for (int ww = 0; ww < kWaves; ++ww) s += lds.red[ww][set][threadIdx.x & 31][o];
            out[Q * pl.n_out + o] = s;
          }
        }
      }
    } else {
      const int passes = (pl.n_out + 255) / 256;
      for (int p = 0; p < passes; ++p) {
        f32x16 acc[kSets][kTPW];
        zero_acc(acc);
        gemm_phase<kTPW, HALF>(mats, Prod{pl.prod_out + p * kActKB * 8 * kTileKB, kActKB, 8}, kTPW * w, lds, 0, acc);
        const float* bias = vecs + pl.out_bias_off + 256 * p;
#pragma unroll
        for (int s = kS0<HALF>; s < kSets; ++s) {
          const long Q = base + tile_pt<HALF>(s, j);
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

// ---- backward ------------------------------------------------------------------------------------------------
// Per tile: recompute the hidden activations (stored to H_l for the weight-gradient GEMM), then
//   dz_{L-1} = (W_out^T g_out) (.) [a_{L-1} > 0]            (n_out <= 4: a rank-n_out update on the VALU)
//   dz_{l-1} = (W_l^T dz_l) (.) [a_{l-1} > 0]               (transposed products on MFMA)
//   g_x      = W_0^T dz_0 (+ W_skip,x^T dz_skip)
// dz_l goes to DZ_l ([row][256]); wgrad.hip then forms dW_l = DZ_l^T H_{l-1} (H_{-1} = x) and db_l.
struct MlpBwdScratch { float* H; float* DZ; size_t stride; };   // layer l at + l * stride, [chunk rows][256]

// acc (input-gradient tiles of one pass of 256 columns) -> g_x[P][8 kin_kb], optionally accumulating
template <bool HALF>
__device__ __forceinline__ void store_gx_pass(const f32x16 (&acc)[kSets][kTPW], float* __restrict__ g_x, int kin_kb,
                                              int pass, long base, long P, bool accumulate) {
  const int lane = lane_id(), w = wave_id(), j = lane & 31, h = lane >> 5;
  const size_t ldx = (size_t)8 * kin_kb;
#pragma unroll
  for (int s = kS0<HALF>; s < kSets; ++s) {
    const long Q = base + tile_pt<HALF>(s, j);
#pragma unroll
    for (int t = 0; t < kTPW; ++t) {
      const int T = kTPW * w + t;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = 256 * pass + 32 * T + 8 * g + 4 * h;
        if (Q < P && col < (int)ldx) {
          f32x4* dst = reinterpret_cast<f32x4*>(g_x + (size_t)Q * ldx + col);
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = acc[s][t][4 * g + c];
          if (accumulate) { const f32x4 old = *dst; o += old; }
          *dst = o;
        }
      }
    }
  }
}

template <bool HALF>
__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
mlp_bwd_sweep_kernel(const f32x4* __restrict__ mats, const float* __restrict__ vecs, MlpPlan pl,
                     const float* __restrict__ X, const float* __restrict__ g_out, long chunk_base, int chunk_n,
                     long P, MlpBwdScratch sc, float* __restrict__ g_x) {
  __shared__ Lds lds;
  const int lane = lane_id(), w = wave_id(), j = lane & 31, h = lane >> 5;
  const int ntiles = (chunk_n + kPts<HALF> - 1) / kPts<HALF>;
  const int L = pl.n_hidden;
  const int xpasses = (pl.kin_kb + kActKB - 1) / kActKB;
  const long chunk_end = chunk_base + chunk_n;
  const long rpad = (long)ntiles * kPts<HALF>;   // rows of the chunk's scratch arrays that this launch defines (zeros / finite
                                             // values beyond chunk_n: the weight-gradient GEMM reads whole 32-row steps)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long base = chunk_base + (long)tile * kPts<HALF>;
    const long row0 = (long)tile * kPts<HALF>;
    mlp_hidden<true, HALF>(lds, mats, vecs, pl, X, base, chunk_end, sc.H, sc.stride, row0, rpad);
    f32x16 acc[kSets][kTPW];
    if (pl.n_out > 4) {
      // ---- wide output layer (DINO field): dz_{L-1} = (W_out^T g_out) (.) [a_{L-1} > 0] on MFMA, g_out rows streamed
      // through the activation region like an input (the mask comes from the stored H_{L-1})
      zero_acc(acc);
      input_gemm<HALF>(lds, mats, pl.rprod_out, pl.n_out / 8, g_out, base, chunk_end, acc);
      __syncthreads();
      const float* Hl = sc.H + (size_t)(L - 1) * sc.stride;
      float* DZl = sc.DZ + (size_t)(L - 1) * sc.stride;
#pragma unroll
      for (int t = 0; t < kTPW; ++t) {
        const int T = kTPW * w + t;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int s = kS0<HALF>; s < kSets; ++s) {
            const size_t off = (size_t)(row0 + tile_pt<HALF>(s, j)) * 256 + 32 * T + 8 * g + 4 * h;
            const f32x4 a = *reinterpret_cast<const f32x4*>(Hl + off);
            f32x4 dz;
#pragma unroll
            for (int c = 0; c < 4; ++c) dz[c] = a[c] > 0.f ? acc[s][t][4 * g + c] : 0.f;
            lds.act[s][4 * T + g][lane] = dz;
            *reinterpret_cast<f32x4*>(DZl + off) = dz;
          }
        }
      }
    } else {
    // ---- output layer: dz_{L-1} from g_out, masked by the activations still in LDS
    float go[kSets][4];
#pragma unroll
    for (int s = kS0<HALF>; s < kSets; ++s) {
      const long Q = base + tile_pt<HALF>(s, j);
#pragma unroll
      for (int o = 0; o < 4; ++o) go[s][o] = (o < pl.n_out && Q < chunk_end) ? g_out[Q * pl.n_out + o] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < kTPW; ++t) {
      const int T = kTPW * w + t;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 wv[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          wv[o] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (o < pl.n_out) wv[o] = load_feat4(vecs + pl.head_off + 256 * o, T, g, h);
        }
#pragma unroll
        for (int s = kS0<HALF>; s < kSets; ++s) {
          const f32x4 a = lds.act[s][4 * T + g][lane];
          f32x4 dz;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float d = go[s][0] * wv[0][c] + go[s][1] * wv[1][c] + go[s][2] * wv[2][c] + go[s][3] * wv[3][c];
            dz[c] = a[c] > 0.f ? d : 0.f;
          }
          lds.act[s][4 * T + g][lane] = dz;
          const long r = row0 + tile_pt<HALF>(s, j);
          *reinterpret_cast<f32x4*>(sc.DZ + (size_t)(L - 1) * sc.stride + (size_t)r * 256 + 32 * T + 8 * g + 4 * h) = dz;
        }
      }
    }
    }
    __syncthreads();
    // ---- hidden layers, last to first
#pragma nounroll
    for (int l = L - 1; l >= 1; --l) {
      if (l == pl.skip && g_x) {   // the x share of the skip layer's input gradient
        for (int p = 0; p < xpasses; ++p) {
          zero_acc(acc);
          gemm_phase<kTPW, HALF>(mats, Prod{pl.rprod_skip_x + p * kActKB * 8 * kTileKB, kActKB, 8}, kTPW * w, lds, 0, acc);
          store_gx_pass<HALF>(acc, g_x, pl.kin_kb, p, base, chunk_end, false);
        }
      }
      zero_acc(acc);
      gemm_phase<kTPW, HALF>(mats, Prod{pl.rprod_h[l], kActKB, 8}, kTPW * w, lds, 0, acc);
      __syncthreads();   // every read of dz_l is done
      const float* Hl = sc.H + (size_t)(l - 1) * sc.stride;
      float* DZl = sc.DZ + (size_t)(l - 1) * sc.stride;
#pragma unroll
      for (int t = 0; t < kTPW; ++t) {
        const int T = kTPW * w + t;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int s = kS0<HALF>; s < kSets; ++s) {
            const long r = row0 + tile_pt<HALF>(s, j);
            const f32x4 a = *reinterpret_cast<const f32x4*>(Hl + (size_t)r * 256 + 32 * T + 8 * g + 4 * h);
            f32x4 dz;
#pragma unroll
            for (int c = 0; c < 4; ++c) dz[c] = a[c] > 0.f ? acc[s][t][4 * g + c] : 0.f;
            lds.act[s][4 * T + g][lane] = dz;
            *reinterpret_cast<f32x4*>(DZl + (size_t)r * 256 + 32 * T + 8 * g + 4 * h) = dz;
          }
        }
      }
      __syncthreads();
    }
    // ---- input gradient of lin0
    if (g_x) {
      for (int p = 0; p < xpasses; ++p) {
        zero_acc(acc);
        gemm_phase<kTPW, HALF>(mats, Prod{pl.rprod_in + p * kActKB * 8 * kTileKB, kActKB, 8}, kTPW * w, lds, 0, acc);
        store_gx_pass<HALF>(acc, g_x, pl.kin_kb, p, base, chunk_end, pl.skip >= 1);
      }
    }
    // the next tile's first slice load synchronises before the activation region is overwritten
  }
}

// ---- second-order backward (the smoothness loss differentiates g_x = d out / d x once more) -----------------------
// With the ReLU masks m_l locally constant, <gbar_x, g_x> = <W_out t_{L-1}, g_out> for the forward-mode tangent
//   t_0 = m_0 (.) W_0 gbar_x,   t_l = m_l (.) (W_l t_{l-1} [+ W_skip,x gbar_x]),
// so d/dW_l = dz_l t_{l-1}^T (t_{-1} = gbar_x), d/dW_out = g_out t_{L-1}^T, d/dg_out = W_out t_{L-1}, and nothing
// reaches x or the biases.  This kernel is the forward kernel on gbar_x without biases, masked by the stored H_l.
template <bool HALF>
__device__ __forceinline__ void epilogue_mask(Lds& lds, f32x16 (&acc)[kSets][kTPW], const float* __restrict__ Hl,
                                              float* __restrict__ Tl, long row0) {
  const int lane = lane_id(), h = lane >> 5, w = wave_id(), j = lane & 31;
#pragma unroll
  for (int t = 0; t < kTPW; ++t) {
    const int T = kTPW * w + t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int s = kS0<HALF>; s < kSets; ++s) {
        const size_t off = (size_t)(row0 + tile_pt<HALF>(s, j)) * 256 + 32 * T + 8 * g + 4 * h;
        const f32x4 a = *reinterpret_cast<const f32x4*>(Hl + off);
        f32x4 tv;
#pragma unroll
        for (int c = 0; c < 4; ++c) tv[c] = a[c] > 0.f ? acc[s][t][4 * g + c] : 0.f;
        lds.act[s][4 * T + g][lane] = tv;
        *reinterpret_cast<f32x4*>(Tl + off) = tv;
      }
    }
  }
}

template <bool HALF>
__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
mlp_tangent_kernel(const f32x4* __restrict__ mats, const float* __restrict__ vecs, MlpPlan pl,
                   const float* __restrict__ GX, long chunk_base, int chunk_n, const float* __restrict__ H,
                   float* __restrict__ Tn, size_t stride, float* __restrict__ ghat_out) {
  __shared__ Lds lds;
  const int lane = lane_id(), w = wave_id();
  const int ntiles = (chunk_n + kPts<HALF> - 1) / kPts<HALF>;
  const long chunk_end = chunk_base + chunk_n;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long base = chunk_base + (long)tile * kPts<HALF>;
    const long row0 = (long)tile * kPts<HALF>;
    f32x16 acc[kSets][kTPW];
    zero_acc(acc);
    input_gemm<HALF>(lds, mats, pl.prod_in, pl.kin_kb, GX, base, chunk_end, acc);
    __syncthreads();
    epilogue_mask<HALF>(lds, acc, H, Tn, row0);
    __syncthreads();
#pragma nounroll
    for (int l = 1; l < pl.n_hidden; ++l) {
      zero_acc(acc);
      gemm_phase<kTPW, HALF>(mats, Prod{pl.prod_h[l], kActKB, 8}, kTPW * w, lds, 0, acc);
      if (l == pl.skip) input_gemm<HALF>(lds, mats, pl.prod_skip_x, pl.kin_kb, GX, base, chunk_end, acc);
      __syncthreads();
      epilogue_mask<HALF>(lds, acc, H + (size_t)l * stride, Tn + (size_t)l * stride, row0);
      __syncthreads();
    }
    for (int o = 0; o < pl.n_out; ++o) {
      float part[kSets];
      head_partial(lds, vecs + pl.head_off + 256 * o, part);
      if (lane < 32) { lds.red[w][0][lane][o] = part[0]; lds.red[w][1][lane][o] = part[1]; }
    }
    __syncthreads();
    if (threadIdx.x < kPts<HALF>) {
      const long Q = base + threadIdx.x;
      const int set = HALF ? 1 : threadIdx.x >> 5;
      if (Q < chunk_end) {
        for (int o = 0; o < pl.n_out; ++o) {
          float sacc = 0.f;
#pragma unroll
          for (int ww = 0; ww < kWaves; ++ww) sacc += lds.red[ww][set][threadIdx.x & 31][o];
          ghat_out[Q * pl.n_out + o] = sacc;
        }
      }
    }
  }
}

// dW_out[o][f] += sum_p g_out[p][o] a_{L-1}[p][f],  db_out[o] += sum_p g_out[p][o]   (n_out <= 4)
constexpr int kHeadBlocks = 512;
__global__ void __launch_bounds__(256)
mlp_head_grad_kernel(const float* __restrict__ H, const float* __restrict__ g_out /* chunk-local */, int n_out, int chunk_n,
                     float* __restrict__ partial /* [blocks][4][257] */) {
  const int col = threadIdx.x;
  const int r0 = (int)((long)blockIdx.x * chunk_n / gridDim.x), r1 = (int)((long)(blockIdx.x + 1) * chunk_n / gridDim.x);
  float s[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8   // rows are independent loads: keep several in flight (a short chunk gives a block ~64 rows)
  for (int r = r0; r < r1; ++r) {
    const float a = H[(size_t)r * 256 + col];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float g = o < n_out ? g_out[(size_t)r * n_out + o] : 0.f;
      s[o] += g * a;
      sg[o] += g;
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    partial[((size_t)blockIdx.x * 4 + o) * 257 + col] = s[o];
    if (col == 0) partial[((size_t)blockIdx.x * 4 + o) * 257 + 256] = sg[o];
  }
}
__global__ void __launch_bounds__(256)
mlp_head_reduce_kernel(const float* __restrict__ partial, int nblocks /* multiple of 8 */, int n_out,
                       float* __restrict__ gw, float* __restrict__ gb) {
  const int o = blockIdx.x, col = threadIdx.x;
  if (o >= n_out) return;
  // eight interleaved partial sums in a fixed order: the kernel runs on n_out <= 4 workgroups, so it is bound by the
  // latency of its dependent load-add steps unless several loads are in flight (and the host keeps nblocks small for
  // short chunks: 512 partials cost 57 us whatever the chunk)
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sg = 0.f;
  for (int b = 0; b < nblocks; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += partial[((size_t)(b + u) * 4 + o) * 257 + col];
  }
  if (col == 0) {
    float g8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < nblocks; b += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) g8[u] += partial[((size_t)(b + u) * 4 + o) * 257 + 256];
    }
    sg = ((g8[0] + g8[1]) + (g8[2] + g8[3])) + ((g8[4] + g8[5]) + (g8[6] + g8[7]));
  }
  const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  gw[o * 256 + col] += s;
  if (col == 0 && gb) gb[o] += sg;
}

// partial-sum blocks of the head gradient for a chunk of n rows: about 64 rows per block, a multiple of 8, at most
// kHeadBlocks (blocks without rows write zeros)
static int head_blocks(int n) {
  static_assert(kHeadBlocks % 8 == 0, "the reduction is unrolled by 8");
  const int b = ((n + 63) / 64 + 7) / 8 * 8;
  return b < 8 ? 8 : (b > kHeadBlocks ? kHeadBlocks : b);
}

static bool mlp_half(long n) { return PIR_MLP_HALF && (n + 31) / 32 <= (long)pir_num_cus(); }

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
  const bool half = mlp_half(P);
  const long ntiles = half ? (P + 31) / 32 : (P + kWgPts - 1) / kWgPts;
  const long resident = 2L * pir_num_cus();
  const dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
  PirProfScope prof(PIR_K_MLP_FWD, (double)P, stream);
  if (half)
    hipLaunchKernelGGL(mlp_eval_kernel<true>, grid, dim3(kThreads), 0, (hipStream_t)stream, mats, vecs, pl, X, P, out);
  else
    hipLaunchKernelGGL(mlp_eval_kernel<false>, grid, dim3(kThreads), 0, (hipStream_t)stream, mats, vecs, pl, X, P, out);
  return pir_check_launch();
}

size_t pir_mlp_backward_scratch_bytes(int n_hidden, long chunk_points) {
  const size_t rows = (size_t)((chunk_points + 63) / 64) * 64;
  return wgrad_fixed_bytes() + (size_t)kHeadBlocks * 4 * 257 * 4 + 512 + rows * (size_t)n_hidden * 2 * 1024;
}

int pir_mlp_backward(const void* packed, size_t mats_floats, const PirMlpPlan* plan, int k0, const float* X, long P,
                     const float* g_out, float* g_x, float* const* gw, float* const* gb, void* scratch,
                     size_t scratch_bytes, void* stream_) {
  if (!packed || !plan || P < 0 || (mats_floats & 3) || !gw || !gb) return PIR_ERR_ARG;
  MlpPlan pl;
  memcpy(&pl, plan, sizeof(pl));
  if (!plan_ok(pl) || !pl.has_bwd || (pl.n_out > 4 && (pl.n_out & 7)) || k0 < 1 || k0 > 8 * pl.kin_kb) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!X || !g_out || !scratch) return PIR_ERR_ARG;
  const int L = pl.n_hidden;
  GradOut out;
  for (int l = 0; l < 9; ++l) { out.w[l] = l <= L ? gw[l] : nullptr; out.b[l] = l <= L ? gb[l] : nullptr; }
  for (int l = 0; l <= L; ++l) if (!out.w[l] || !out.b[l]) return PIR_ERR_ARG;
  const size_t head_bytes = ((size_t)kHeadBlocks * 4 * 257 * 4 + 255) & ~(size_t)255;
  const size_t row_bytes = (size_t)L * 2 * 1024;
  if (scratch_bytes < wgrad_fixed_bytes() + head_bytes + 256 + 64 * row_bytes) return PIR_ERR_SCRATCH;
  long cap = (long)((scratch_bytes - wgrad_fixed_bytes() - head_bytes - 256) / row_bytes) / 64 * 64;
  if (cap > 262144) cap = 262144;
  const WgFixed fx = wgrad_carve(scratch);
  float* head_partial = (float*)fx.rest;
  MlpBwdScratch sc;
  sc.stride = (size_t)cap * 256;
  sc.H = (float*)(fx.rest + head_bytes);
  sc.DZ = sc.H + (size_t)L * sc.stride;
  const f32x4* mats = (const f32x4*)packed;
  const float* vecs = (const float*)packed + mats_floats;
  const int kpad = 8 * pl.kin_kb;
  const float is2 = 0.70710678118654752440f;
  hipStream_t stream = (hipStream_t)stream_;
  for (long base = 0; base < P; base += cap) {
    const int n = (int)(P - base < cap ? P - base : cap);
    const int n_rows = (n + 31) / 32 * 32;
    const bool half = mlp_half(n);
    const int ntiles = half ? (n + 31) / 32 : (n + kWgPts - 1) / kWgPts;
    const long resident = 2L * pir_num_cus();
    const dim3 grid((unsigned)(ntiles < resident ? ntiles : resident));
    {
      PirProfScope prof(PIR_K_MLP_BWD, (double)n, stream_);
      if (half)
        hipLaunchKernelGGL(mlp_bwd_sweep_kernel<true>, grid, dim3(kThreads), 0, stream, mats, vecs, pl, X, g_out, base, n, P,
                           sc, g_x);
      else
        hipLaunchKernelGGL(mlp_bwd_sweep_kernel<false>, grid, dim3(kThreads), 0, stream, mats, vecs, pl, X, g_out, base, n, P,
                           sc, g_x);
    }
    int rc = pir_check_launch();
    if (rc != PIR_OK) return rc;
    // weight-gradient jobs: dW_l = DZ_l^T [H_{l-1} | x], db_l = column sums of DZ_l
    WgJobs jobs;
    jobs.n = 0;
    auto flush = [&]() -> int {
      if (jobs.n == 0) return PIR_OK;
      const int r = wgrad_run(jobs, n_rows, n, fx, nullptr, out, PIR_K_MLP_BWD, stream_);
      jobs.n = 0;
      return r;
    };
    auto add = [&](int l, const float* Y, int ldy, int y_rows, int ncols_y, int dest_cols, int col_off, float scale,
                   bool with_bias) -> int {
      for (int ot = 0; ot < 2; ++ot)
        for (int i0 = 0; i0 < ncols_y; i0 += 128) {
          if (jobs.n == kMaxJobs) { const int r = flush(); if (r != PIR_OK) return r; }
          WgJob& j = jobs.j[jobs.n++];
          j.X1 = sc.DZ + (size_t)l * sc.stride; j.X2 = nullptr; j.Y1 = Y; j.Y2 = nullptr;
          j.ldx = 256; j.ldy = ldy; j.x1_rows = -1; j.y1_rows = y_rows;
          j.o0 = 128 * ot; j.i0 = i0;
          j.layer = l; j.row_off = 0; j.rows = 256; j.cols = dest_cols; j.col_off = col_off; j.i_limit = ncols_y;
          j.scale = scale; j.bias = with_bias && i0 == 0;
        }
      return PIR_OK;
    };
    const float* Xc = X + (size_t)base * kpad;
    rc = add(0, Xc, kpad, n, k0, k0, 0, 1.0f, true);
    for (int l = 1; l < L && rc == PIR_OK; ++l) {
      const bool sk = l == pl.skip;
      rc = add(l, sc.H + (size_t)(l - 1) * sc.stride, 256, -1, 256, sk ? 256 + k0 : 256, 0, sk ? is2 : 1.0f, true);
      if (rc == PIR_OK && sk) rc = add(l, Xc, kpad, n, k0, 256 + k0, 256, is2, false);
    }
    if (rc == PIR_OK && pl.n_out > 4) {
      // wide output layer: dW_out = g_out^T H_{L-1} ([n_out, 256]) and db_out through the same GEMM, g_out as the X operand
      for (int o0 = 0; o0 < pl.n_out && rc == PIR_OK; o0 += 128)
        for (int i0 = 0; i0 < 256; i0 += 128) {
          if (jobs.n == kMaxJobs) { rc = flush(); if (rc != PIR_OK) break; }
          WgJob& j = jobs.j[jobs.n++];
          j.X1 = g_out + (size_t)base * pl.n_out; j.X2 = nullptr; j.Y1 = sc.H + (size_t)(L - 1) * sc.stride; j.Y2 = nullptr;
          j.ldx = pl.n_out; j.ldy = 256; j.x1_rows = n; j.y1_rows = -1;
          j.o0 = o0; j.i0 = i0;
          j.layer = L; j.row_off = 0; j.rows = pl.n_out; j.cols = 256; j.col_off = 0; j.i_limit = 256;
          j.scale = 1.0f; j.bias = i0 == 0;
        }
    }
    if (rc == PIR_OK) rc = flush();
    if (rc != PIR_OK) return rc;
    if (pl.n_out <= 4) {
      const int hb = head_blocks(n);
      hipLaunchKernelGGL(mlp_head_grad_kernel, dim3(hb), dim3(256), 0, stream, sc.H + (size_t)(L - 1) * sc.stride,
                         g_out + (size_t)base * pl.n_out, pl.n_out, n, head_partial);
      hipLaunchKernelGGL(mlp_head_reduce_kernel, dim3(pl.n_out), dim3(256), 0, stream, head_partial, hb, pl.n_out,
                         out.w[L], out.b[L]);
      rc = pir_check_launch();
      if (rc != PIR_OK) return rc;
    }
  }
  return PIR_OK;
}

int pir_mlp_backward_tangent(const void* packed, size_t mats_floats, const PirMlpPlan* plan, int k0, const float* X,
                             long P, const float* g_out, const float* gbar_x, float* ghat_out, float* const* gw,
                             void* scratch, size_t scratch_bytes, void* stream_) {
  if (!packed || !plan || P < 0 || (mats_floats & 3) || !gw) return PIR_ERR_ARG;
  MlpPlan pl;
  memcpy(&pl, plan, sizeof(pl));
  if (!plan_ok(pl) || !pl.has_bwd || pl.n_out > 4 || k0 < 1 || k0 > 8 * pl.kin_kb) return PIR_ERR_ARG;
  if (P == 0) return PIR_OK;
  if (!X || !g_out || !gbar_x || !ghat_out || !scratch) return PIR_ERR_ARG;
  const int L = pl.n_hidden;
  GradOut out;
  for (int l = 0; l < 9; ++l) { out.w[l] = l <= L ? gw[l] : nullptr; out.b[l] = nullptr; }
  for (int l = 0; l <= L; ++l) if (!out.w[l]) return PIR_ERR_ARG;
  const size_t head_bytes = ((size_t)kHeadBlocks * 4 * 257 * 4 + 255) & ~(size_t)255;
  const size_t row_bytes = (size_t)L * 3 * 1024;
  if (scratch_bytes < wgrad_fixed_bytes() + head_bytes + 256 + 64 * row_bytes) return PIR_ERR_SCRATCH;
  long cap = (long)((scratch_bytes - wgrad_fixed_bytes() - head_bytes - 256) / row_bytes) / 64 * 64;
  if (cap > 262144) cap = 262144;
  const WgFixed fx = wgrad_carve(scratch);
  float* head_partial = (float*)fx.rest;
  MlpBwdScratch sc;
  sc.stride = (size_t)cap * 256;
  sc.H = (float*)(fx.rest + head_bytes);
  sc.DZ = sc.H + (size_t)L * sc.stride;
  float* Tn = sc.DZ + (size_t)L * sc.stride;
  const f32x4* mats = (const f32x4*)packed;
  const float* vecs = (const float*)packed + mats_floats;
  const int kpad = 8 * pl.kin_kb;
  const float is2 = 0.70710678118654752440f;
  hipStream_t stream = (hipStream_t)stream_;
  for (long base = 0; base < P; base += cap) {
    const int n = (int)(P - base < cap ? P - base : cap);
    const int n_rows = (n + 31) / 32 * 32;
    const bool half = mlp_half(n);
    const int ntiles = half ? (n + 31) / 32 : (n + kWgPts - 1) / kWgPts;
    const long resident = 2L * pir_num_cus();
    const unsigned grid = (unsigned)(ntiles < resident ? ntiles : resident);
    {
      PirProfScope prof(PIR_K_MLP_BWD, (double)n, stream_);
      if (half) {
        hipLaunchKernelGGL(mlp_bwd_sweep_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, mats, vecs, pl, X, g_out, base,
                           n, P, sc, (float*)nullptr);
        hipLaunchKernelGGL(mlp_tangent_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, mats, vecs, pl, gbar_x, base, n,
                           (const float*)sc.H, Tn, sc.stride, ghat_out);
      } else {
        hipLaunchKernelGGL(mlp_bwd_sweep_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, mats, vecs, pl, X, g_out, base,
                           n, P, sc, (float*)nullptr);
        hipLaunchKernelGGL(mlp_tangent_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, mats, vecs, pl, gbar_x, base, n,
                           (const float*)sc.H, Tn, sc.stride, ghat_out);
      }
    }
    int rc = pir_check_launch();
    if (rc != PIR_OK) return rc;
    WgJobs jobs;
    jobs.n = 0;
    auto flush = [&]() -> int {
      if (jobs.n == 0) return PIR_OK;
      const int r = wgrad_run(jobs, n_rows, n, fx, nullptr, out, PIR_K_MLP_BWD, stream_);
      jobs.n = 0;
      return r;
    };
    auto add = [&](int l, const float* Y, int ldy, int y_rows, int ncols_y, int dest_cols, int col_off, float scale) -> int {
      for (int ot = 0; ot < 2; ++ot)
        for (int i0 = 0; i0 < ncols_y; i0 += 128) {
          if (jobs.n == kMaxJobs) { const int r = flush(); if (r != PIR_OK) return r; }
          WgJob& j = jobs.j[jobs.n++];
          j.X1 = sc.DZ + (size_t)l * sc.stride; j.X2 = nullptr; j.Y1 = Y; j.Y2 = nullptr;
          j.ldx = 256; j.ldy = ldy; j.x1_rows = -1; j.y1_rows = y_rows;
          j.o0 = 128 * ot; j.i0 = i0;
          j.layer = l; j.row_off = 0; j.rows = 256; j.cols = dest_cols; j.col_off = col_off; j.i_limit = ncols_y;
          j.scale = scale; j.bias = 0;
        }
      return PIR_OK;
    };
    const float* Gc = gbar_x + (size_t)base * kpad;
    rc = add(0, Gc, kpad, n, k0, k0, 0, 1.0f);
    for (int l = 1; l < L && rc == PIR_OK; ++l) {
      const bool sk = l == pl.skip;
      rc = add(l, Tn + (size_t)(l - 1) * sc.stride, 256, -1, 256, sk ? 256 + k0 : 256, 0, sk ? is2 : 1.0f);
      if (rc == PIR_OK && sk) rc = add(l, Gc, kpad, n, k0, 256 + k0, 256, is2);
    }
    if (rc == PIR_OK) rc = flush();
    if (rc != PIR_OK) return rc;
    const int hb = head_blocks(n);
    hipLaunchKernelGGL(mlp_head_grad_kernel, dim3(hb), dim3(256), 0, stream, Tn + (size_t)(L - 1) * sc.stride,
                       g_out + (size_t)base * pl.n_out, pl.n_out, n, head_partial);
    hipLaunchKernelGGL(mlp_head_reduce_kernel, dim3(pl.n_out), dim3(256), 0, stream, head_partial, hb, pl.n_out,
                       out.w[L], (float*)nullptr);
    rc = pir_check_launch();
    if (rc != PIR_OK) return rc;
  }
  return PIR_OK;
}

}  // extern "C"
