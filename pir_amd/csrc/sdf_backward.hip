// Backward of SDFNetwork.get_all(x, is_training=True) (reference: src/network.py:125-142 differentiated by
// autograd, i.e. a double backward through the softplus MLP), written out analytically.
//
// For upstream gradients (g_sdf, g_feat, g_normal) the parameter gradient is
//     dW_l = Zb_l H_l^T + Q_l Pb_l^T ,   db_l = sum_pts Zb_l
// with, per point,
//     H_l   input of layer l (a_{l-1}; PE for l = 0; [a_3 | PE]/sqrt2 for l = 4)
//     Pb_l  tangent of H_l along g_normal  (forward-mode derivative: the adjoint of the d sdf/dx chain)
//     Q_l   d sdf / d z_l                   (the chain that produced the normal)
//     Zb_l  d L / d z_l                     (ordinary backprop + the softplus'' term 100 (1-s) Q_l (z_l-tangent))
// Two fused sweeps per 32-point tile compute them with the two operand sets of the wave pair sharing every
// weight fragment:  FT (set 0 = activations, set 1 = tangents, layers 0..7) and GZ (set 0 = Q, set 1 = Zb,
// layers 8..0).  The four operand families go to HBM scratch (fragment order, sdf_backward.h) for the
// split-K weight-gradient GEMM in wgrad.hip.  d L / d x comes out of the PE Jacobian at the end of GZ.
#include "sdf_sweeps.cuh"
#include "sdf_backward.h"
#include "pir_common.h"

#ifndef PIR_PE_KSPLIT_BWD
#define PIR_PE_KSPLIT_BWD PIR_PE_KSPLIT
#endif

namespace pir {

// float4 (features 32 T + 8 g + 4 h + 0..3 of the point in scratch row `row`) of a [rows][256] family; h = lane / 32,
// row % 32 = lane % 32 for every caller, so in fragment order the 64 lanes of one instruction cover 1 KB contiguously
__device__ __forceinline__ size_t scratch_off(int row, int T, int g, int h) {
#if PIR_BWD_FRAG
  return (((size_t)(row >> 5) * 32 + 4 * T + g) * 64 + 32 * h + (row & 31)) * 4;
#else
  return (size_t)row * 256 + 32 * T + 8 * g + 4 * h;
#endif
}
// PIR_ABL_* : ablation builds for timing only (tools/build_variant.sh); their results are wrong
__device__ __forceinline__ void store_nat(float* __restrict__ arr, int row, int T, int g, int h, f32x4 v) {
#ifdef PIR_ABL_NO_A_STORE
  if (v[0] != 12345.678f) return;
#endif
#ifdef PIR_BWD_NT_A
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(arr + scratch_off(row, T, g, h)));
#else
  *reinterpret_cast<f32x4*>(arr + scratch_off(row, T, g, h)) = v;
#endif
}
// Q / Zb are written once here and read once by the weight-gradient GEMM of a later launch (PIR_BWD_NT_QZ, sdf_backward.h)
__device__ __forceinline__ void store_nat_stream(float* __restrict__ arr, int row, int T, int g, int h, f32x4 v) {
#ifdef PIR_ABL_NO_QZ_STORE
  if (v[0] != 12345.678f) return;
#endif
#if PIR_BWD_NT_QZ
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(arr + scratch_off(row, T, g, h)));
#else
  store_nat(arr, row, T, g, h, v);
#endif
}
__device__ __forceinline__ f32x4 load_nat(const float* __restrict__ arr, int row, int T, int g, int h) {
#ifdef PIR_ABL_NO_A_LOAD
  if (row != -12345) return f32x4{0.001f * T, 0.002f * g, 0.003f * h, 0.004f};
#endif
#if PIR_BWD_NT_LD
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(arr + scratch_off(row, T, g, h)));
#else
  return *reinterpret_cast<const f32x4*>(arr + scratch_off(row, T, g, h));
#endif
}

// value of PE column `col` (reference order) of point j in operand set `set`
__device__ __forceinline__ float pe_nat(const Lds& lds, int set, int j, int col) {
  int slot, hh;
  if (col < 2) { slot = 0; hh = col; }
  else if (col == 2) { slot = 1; hh = 0; }
  else { const int k = (col - 3) / 6, rem = (col - 3) % 6; hh = rem / 3; slot = 2 + 3 * k + rem % 3; }
  return lds.act[set][kActKB + (slot >> 2)][32 * hh + j][slot & 3];
}

// tangent of the positional encoding along nb: slot values (nb0, nb1), (nb2, 0), (f cos(f x_c) nb_c, -f sin(f x_c) nb_c)
__device__ __forceinline__ void write_pe_tangent(Lds& lds, int set, float x0, float x1, float x2, float n0, float n1,
                                                 float n2) {
  const int lane = lane_id();
  const int h = lane >> 5;
  float v[20];
  v[0] = h ? n1 : n0;
  v[1] = h ? 0.f : n2;
  const float xs[3] = {x0, x1, x2};
  const float ns[3] = {n0, n1, n2};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float sn, cs;
      const float f = (float)(1 << k);
      sincos_pe(xs[c] * f, sn, cs);
      v[2 + 3 * k + c] = (h ? -f * sn : f * cs) * ns[c];
    }
  }
#pragma unroll
  for (int q = 0; q < kPeKB; ++q) {
    f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    lds.act[set][kActKB + q][lane] = o;
  }
}

// copy the 20 PE slots of this lane to a [row][64] array in reference column order
__device__ __forceinline__ void store_pe_cols(const Lds& lds, int set, float* __restrict__ arr, int row) {
  const int lane = lane_id(), h = lane >> 5;
#pragma unroll
  for (int slot = 0; slot < 20; ++slot) {
    const int col = pe_column(slot, h);
    if (col >= 0) arr[(size_t)row * 64 + col] = lds.act[set][kActKB + (slot >> 2)][lane][slot & 3];
  }
}

// d L / d x from the PE tile owned by wave W (slots 16W..16W+15): J_e^T e-bar plus the second derivative of the
// encoding contracted with v (set 0) and the normal gradient.  W is a template parameter so every slot index,
// frequency and LDS offset is a compile-time constant.
template <int W>
__device__ __forceinline__ void pe_xbar(const Lds& lds, const f32x16 (&accpe)[kSets][1], float nb0, float nb1,
                                        float nb2, int h, float (&xb)[3]) {
  const int lane = lane_id();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int slot = 16 * W + r;
    if (slot >= 20) continue;
    const float v = accpe[0][0][r], eb = accpe[1][0][r];
    if (slot == 0) { xb[0] += h ? 0.f : eb; xb[1] += h ? eb : 0.f; }
    else if (slot == 1) { xb[2] += h ? 0.f : eb; }
    else {
      const int k = (slot - 2) / 3, c = (slot - 2) % 3;
      const float f = (float)(1 << k);
      const float own = lds.act[0][kActKB + (slot >> 2)][lane][slot & 3];
      const float other = lds.act[0][kActKB + (slot >> 2)][lane ^ 32][slot & 3];
      const float t1 = (h ? -f : f) * other * eb;           // J_e^T e-bar
      const float t2 = -(f * f) * own * v;                  // second derivative of sin/cos times v
      const float nbc = c == 0 ? nb0 : (c == 1 ? nb1 : nb2);
      xb[c] += t1 + nbc * t2;
    }
  }
}

__global__ void __launch_bounds__(kThreads, kWavesPerSimd)
bwd_sweep_kernel(SdfPacked pk, const float* __restrict__ x, const float* __restrict__ g_sdf,
                 const float* __restrict__ g_feat, const float* __restrict__ g_nrm, long chunk_base, int chunk_n,
                 BwdScratch sc, float* __restrict__ g_x) {
  __shared__ Lds lds;
  const int lane = lane_id(), w = wave_id(), j = lane & 31;
  const int ntiles = (chunk_n + kTilePts - 1) / kTilePts;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // Launder the lane half so that hipcc does not hoist the hundreds of per-(layer, tile, group) scratch
    // addresses out of this loop (they would all spill).
    int h = lane >> 5;
    asm volatile("" : "+v"(h));
    const int row = tile * kTilePts + j;           // chunk-local scratch row of this lane's point
    const bool valid = row < chunk_n;
    const long P = chunk_base + row;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f, nb0 = 0.f, nb1 = 0.f, nb2 = 0.f, gs = 0.f;
    if (valid) {
      x0 = x[3 * P]; x1 = x[3 * P + 1]; x2 = x[3 * P + 2];
      if (g_nrm) { nb0 = g_nrm[3 * P]; nb1 = g_nrm[3 * P + 1]; nb2 = g_nrm[3 * P + 2]; }
      if (g_sdf) gs = g_sdf[P];
    }
    if (w == 0) write_pe(lds, 0, x0, x1, x2);
    else if (w == 1) write_pe_tangent(lds, 1, x0, x1, x2, nb0, nb1, nb2);
    if (w < 2) store_pe_cols(lds, w, w == 0 ? sc.E : sc.Ed, row);
    __syncthreads();

    f32x16 acc[kSets][kTPW];
    f32x4 pre[kTPW];
    load_first_block<kTPW>(pk.mats, kFwd[0], kTPW * w, pre);
    // ------------------------------------------------------------------ FT sweep: activations + tangents
#pragma nounroll
    for (int ph = 0; ph < kFwdPhases; ++ph) {
      if (fwd_phase_first(ph)) zero_acc(acc);
      // (plain two-tile phases here: the padding-tile skip of fwd_gemm_phase costs this kernel 15 spilled registers)
      gemm_phase_pre<kTPW, false, true>(pk.mats, kFwdPh[ph], kTPW * w, lds, fwd_phase_kb0(ph), acc, pre, true,
                                        ph < kFwdPhases - 1 ? kFwdPh[ph + 1] : kRev[7]);
      if (!fwd_phase_last(ph)) continue;
      const int l = fwd_phase_layer(ph);
      __syncthreads();
      PIR_PRIO_HI();
      const float* bias = pk.vecs + kVecBias + 256 * l;
#pragma unroll
      for (int t = 0; t < kTPW; ++t) {
        const int T = kTPW * w + t;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = load_feat4(bias, T, g, h);
          f32x4 a, ad;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            a[c] = softplus100(acc[0][t][4 * g + c] + b[c]);
            ad[c] = acc[1][t][4 * g + c] * dsoftplus_from_act(a[c]);
          }
          lds.act[0][4 * T + g][lane] = a;
          lds.act[1][4 * T + g][lane] = ad;
          if (l == 3 && T >= 6) {   // rows 217..255 of the lin4 input are the encoding (src/network.py:90-91)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const int f = 32 * T + 8 * g + 4 * h + c;
              if (f >= kSkipKeep) { a[c] = pe_nat(lds, 0, j, f - kSkipKeep); ad[c] = pe_nat(lds, 1, j, f - kSkipKeep); }
            }
          }
          store_nat(sc.A(l), row, T, g, h, a);
          store_nat(sc.Ad(l), row, T, g, h, ad);
        }
#ifndef PIR_BWD_FT_NOSB
        __builtin_amdgcn_sched_barrier(0);   // keep one tile's loads/stores in flight at a time (register pressure)
#endif
      }
      PIR_PRIO_LO();
      __syncthreads();
    }

    // ------------------------------------------------------------------ GZ sweep: Q (d sdf/d z) and Zb (d L/d z)
    // layer 8: u = W8^T e0 = w8 row;  y = W8^T [g_sdf; g_feat]
#pragma unroll
    for (int t = 0; t < kTPW; ++t) {
      const int T = kTPW * w + t;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 wv = load_feat4(pk.vecs + kVecW8, T, g, h);
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[0][t][4 * g + c] = wv[c]; acc[1][t][4 * g + c] = gs * wv[c]; }
        if (g_feat) {
          f32x4 gf = {0.f, 0.f, 0.f, 0.f};
          if (valid) gf = *reinterpret_cast<const f32x4*>(g_feat + (size_t)P * 256 + 32 * T + 8 * g + 4 * h);
          lds.act[1][4 * T + g][lane] = gf;
        }
      }
    }
    if (g_feat) {
      __syncthreads();
      gemm_phase<kTPW, true>(pk.mats, kR8, kTPW * w, lds, 0, acc);
      __syncthreads();
    }
    // gradient w.r.t. the encoding slots = lin4's share (skip connection) + lin0's share.  The first is parked in
    // global memory between the two products instead of occupying 32 registers through layers 3..0.
#if PIR_PE_KSPLIT_BWD
    float* pe_park = sc.pe_tmp + ((size_t)blockIdx.x * 4 + w) * (32 * 64) + lane;
#else
    float* pe_park = sc.pe_tmp + ((size_t)blockIdx.x * 2 + (w & 1)) * (32 * 64) + lane;
#endif
#pragma nounroll
    for (int l = 8; l >= 1; --l) {
      // acc = (u_l, y_l): gradients w.r.t. the input of layer l  ->  (q_{l-1}, zb_{l-1})
      const int m = l - 1;
      PIR_PRIO_HI();
#pragma unroll
      for (int t = 0; t < kTPW; ++t) {
        const int T = kTPW * w + t;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 a = load_nat(sc.A(m), row, T, g, h);
          const f32x4 ad = load_nat(sc.Ad(m), row, T, g, h);
          f32x4 q, zb;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float s = dsoftplus_from_act(a[c]);
            const float u = acc[0][t][4 * g + c];
            q[c] = u * s;
            zb[c] = acc[1][t][4 * g + c] * s + 100.f * (1.f - s) * u * ad[c];
            if (m == 3 && 32 * T + 8 * g + 4 * h + c >= kSkipKeep) { q[c] = 0.f; zb[c] = 0.f; }
          }
          lds.act[0][4 * T + g][lane] = q;
          lds.act[1][4 * T + g][lane] = zb;
          store_nat_stream(sc.Q(m), row, T, g, h, q);
          store_nat_stream(sc.Zb(m), row, T, g, h, zb);
        }
#ifndef PIR_BWD_GZ_NOSB
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      PIR_PRIO_LO();
      __syncthreads();
      if (m >= 1) {
#if PIR_BWD_TOUCH
        // pull the A / Ad fragments the NEXT epilogue reads (8 KB each per wave, contiguous in fragment order) towards
        // the L2 while this layer's products run: one dword per 128-byte line
        const size_t touch = ((size_t)(tile * kTilePts + 8 * w) * 64) * 4 + lane * 32;
        const float t0 = sc.A(m - 1)[touch], t1 = sc.Ad(m - 1)[touch];
        __builtin_amdgcn_sched_barrier(0);
#endif
        zero_acc(acc);
        gemm_phase_pre<kTPW, false, true>(pk.mats, kRev[m], kTPW * w, lds, 0, acc, pre, m >= 2, kRev[m >= 2 ? m - 1 : m]);
#if PIR_BWD_TOUCH
        asm volatile("" ::"v"(t0), "v"(t1));
#endif
        if (m == 4 && (PIR_PE_KSPLIT_BWD || w < 2)) {
          f32x16 accpe[kSets][1];
          zero_acc(accpe);
#if PIR_PE_KSPLIT_BWD
          gemm_phase<1>(pk.mats, pe_khalf(kR4P, w >> 1), w & 1, lds, (kActKB / 2) * (w >> 1), accpe);
#else
          gemm_phase<1>(pk.mats, kR4P, w, lds, 0, accpe);
#endif
#pragma unroll
          for (int s2 = 0; s2 < kSets; ++s2)
#pragma unroll
            for (int r = 0; r < 16; ++r) pe_park[(16 * s2 + r) * 64] = accpe[s2][0][r];
        }
      }
      __syncthreads();
    }
    f32x16 accpe[kSets][1];
    zero_acc(accpe);
    if (PIR_PE_KSPLIT_BWD || w < 2) {
#if PIR_PE_KSPLIT_BWD
      gemm_phase<1>(pk.mats, pe_khalf(kR0P, w >> 1), w & 1, lds, (kActKB / 2) * (w >> 1), accpe);
#else
      gemm_phase<1>(pk.mats, kR0P, w, lds, 0, accpe);
#endif
#pragma unroll
      for (int s2 = 0; s2 < kSets; ++s2)
#pragma unroll
        for (int r = 0; r < 16; ++r) accpe[s2][0][r] += pe_park[(16 * s2 + r) * 64];
    }

    // ------------------------------------------------------------------ d L / d x through the encoding
    if (g_x) {
      float xb[3] = {0.f, 0.f, 0.f};
#if PIR_PE_KSPLIT_BWD
      if ((w & 1) == 0) pe_xbar<0>(lds, accpe, nb0, nb1, nb2, h, xb);     // linear in accpe: partial sums per K half
      else pe_xbar<1>(lds, accpe, nb0, nb1, nb2, h, xb);
#else
      if (w == 0) pe_xbar<0>(lds, accpe, nb0, nb1, nb2, h, xb);
      else if (w == 1) pe_xbar<1>(lds, accpe, nb0, nb1, nb2, h, xb);
#endif
#pragma unroll
      for (int c = 0; c < 3; ++c) xb[c] += __shfl_xor(xb[c], 32);
      if (lane < 32 && (PIR_PE_KSPLIT_BWD || w < 2)) {
#pragma unroll
        for (int c = 0; c < 3; ++c) lds.red[w][0][lane][1 + c] = xb[c];
      }
      __syncthreads();
      if (threadIdx.x < kTilePts && tile * kTilePts + (int)threadIdx.x < chunk_n) {
        const long Q = chunk_base + tile * kTilePts + threadIdx.x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v = lds.red[0][0][threadIdx.x][1 + c] + lds.red[1][0][threadIdx.x][1 + c];
          if (PIR_PE_KSPLIT_BWD) v += lds.red[2][0][threadIdx.x][1 + c] + lds.red[3][0][threadIdx.x][1 + c];
          g_x[3 * Q + c] = v;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace pir

using namespace pir;

extern "C" int pir_internal_bwd_sweep(const void* packed, const float* x, const float* g_sdf, const float* g_feat,
                                      const float* g_nrm, long chunk_base, int chunk_n, const BwdScratch* sc,
                                      float* g_x, void* stream) {
  SdfPacked pk;
  pk.mats = (const f32x4*)packed;
  pk.vecs = (const float*)packed + (size_t)kPackedF4 * 4;
  const int ntiles = (chunk_n + kTilePts - 1) / kTilePts;
  const int resident = 2 * pir_num_cus();
  const unsigned grid = (unsigned)(ntiles < resident ? ntiles : resident);
  PirProfScope prof(PIR_K_BWD_SWEEP, (double)chunk_n, stream);
  hipLaunchKernelGGL(bwd_sweep_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, pk, x, g_sdf, g_feat,
                     g_nrm, chunk_base, chunk_n, *sc, g_x);
  return pir_check_launch();
}
