// Forward / reverse sweeps of the SDF MLP built on sdf_core.cuh.
//   forward_hidden : PE -> lin0..lin7 (+softplus); leaves a7 in LDS, optionally stores a0..a7 to scratch
//   sdf_head       : sdf = w8[0,:] . a7 + b8[0]          (VALU dot + cross-half / cross-wave reduce)
//   feature_head   : lin8 rows 1..256 -> global [M,256]
//   normal_reverse : d sdf / d x by the reverse sweep through the transposed products (needs scratch)
// Reference semantics: SDFNetwork.forward / get_all, src/network.py:81-97, :125-142.
#pragma once
#include "sdf_core.cuh"

namespace pir {

__device__ constexpr Prod kFwd[9] = {kF0, kF1, kF2, kF3, kF4, kF5, kF6, kF7, kF8};
// GEMM phases of the hidden layers: with kSkipTwoPhase lin4 is two phases (activation part, encoding part)
constexpr int kFwdPhases = kSkipTwoPhase ? 9 : 8;
__device__ constexpr Prod kFwdPh[9] = {kF0, kF1, kF2, kF3, kSkipTwoPhase ? kF4a : kF4, kSkipTwoPhase ? kF4b : kF5,
                                       kSkipTwoPhase ? kF5 : kF6, kSkipTwoPhase ? kF6 : kF7, kF7};
__device__ __forceinline__ int fwd_phase_layer(int ph) { return kSkipTwoPhase && ph > 4 ? ph - 1 : ph; }
__device__ __forceinline__ bool fwd_phase_first(int ph) { return !(kSkipTwoPhase && ph == 5); }   // starts a layer
__device__ __forceinline__ bool fwd_phase_last(int ph) { return !(kSkipTwoPhase && ph == 4); }    // ends a layer
__device__ __forceinline__ int fwd_phase_kb0(int ph) { return ph == 0 || (kSkipTwoPhase && ph == 5) ? kActKB : 0; }
__device__ constexpr Prod kRev[9] = {kR0P, kR1, kR2, kR3, kR4, kR5, kR6, kR7, kR8};

// One forward GEMM phase of this wave's tiles.  lin3 has 217 outputs, so tile 7 (rows 224..255: the second tile of
// the last wave) is all padding: its MFMAs are not issued and its accumulators stay 0, which is what the zero weight
// rows produced (round 1 executed them).  The matrix pipe is shared with the co-resident workgroup, so the saving
// shows although the other waves of this workgroup still run two tiles.
template <bool ONLY1 = false>
__device__ __forceinline__ void fwd_gemm_phase(const SdfPacked& pk, int ph, const Lds& lds, f32x16 (&acc)[kSets][kTPW],
                                               f32x4 (&pre)[kTPW], bool prefetch_next, Prod next) {
  const int w = wave_id();
  if (kSkipTwoPhase && kTPW == 2 && ph == 3 && w == kWaves - 1) {
    f32x16 a1[kSets][1];
    f32x4 p1[1] = {pre[0]};
#pragma unroll
    for (int s = 0; s < kSets; ++s) a1[s][0] = acc[s][0];
    gemm_phase_pre<1, ONLY1, true>(pk.mats, kFwdPh[ph], kTPW * w, lds, fwd_phase_kb0(ph), a1, p1, prefetch_next, next);
#pragma unroll
    for (int s = 0; s < kSets; ++s) acc[s][0] = a1[s][0];
    pre[0] = p1[0];
    if (prefetch_next) pre[kTPW - 1] = pk.mats[next.off + (size_t)(kTPW * w + kTPW - 1) * kTileKB + lane_id()];
  } else {
    gemm_phase_pre<kTPW, ONLY1, true>(pk.mats, kFwdPh[ph], kTPW * w, lds, fwd_phase_kb0(ph), acc, pre, prefetch_next, next);
  }
}

constexpr int kScratchLayerF4 = kSets * 8 * 4 * 64;       // f32x4 per layer image: [set][T][g][lane]
#ifdef PIR_STAMPS
constexpr int kScratchWgF4 = 8 * kScratchLayerF4 + kWaves * 4;   // a0..a7 + 8 stamp counters per wave
#else
constexpr int kScratchWgF4 = 8 * kScratchLayerF4;         // a0..a7
#endif
constexpr size_t kScratchWgBytes = (size_t)kScratchWgF4 * sizeof(f32x4);

// ONLY1: only operand set 1 carries points (a 32-point tile for short work lists): the products of set 0 are not
// issued -- half the matrix work of a pass; its accumulators stay 0 and its epilogues run on bias values.
template <bool STORE, bool ONLY1 = false>
__device__ __forceinline__ void forward_hidden(Lds& lds, const SdfPacked& pk, float x0, float x1, float x2,
                                               f32x4* __restrict__ scratch) {
  const int w = wave_id();
  if (w < kSets) write_pe(lds, w, x0, x1, x2);   // wave w encodes the points of set w
  __syncthreads();
  f32x16 acc[kSets][kTPW];
  f32x4 pre[kTPW];
  load_first_block<kTPW>(pk.mats, kFwd[0], kTPW * w, pre);
#pragma nounroll
  for (int ph = 0; ph < kFwdPhases; ++ph) {
    if (fwd_phase_first(ph)) zero_acc(acc);
    fwd_gemm_phase<ONLY1>(pk, ph, lds, acc, pre, ph < kFwdPhases - 1, kFwdPh[ph < kFwdPhases - 1 ? ph + 1 : ph]);
    PIR_STAMP(1);
    if (!fwd_phase_last(ph)) continue;
    const int l = fwd_phase_layer(ph);
    __syncthreads();   // every read of the previous activations is done
    PIR_STAMP(2);
    PIR_PRIO_HI();
    epilogue_softplus<STORE>(lds, acc, pk.vecs + kVecBias + 256 * l, scratch + (size_t)l * kScratchLayerF4);
    PIR_PRIO_LO();
    PIR_STAMP(3);
    __syncthreads();
    PIR_STAMP(4);
  }
}

// Partial dot of this wave's 4 tiles of the LDS activations with a 256-vector; result complete over
// the two lane halves, still partial over the two waves.
__device__ __forceinline__ void head_partial(const Lds& lds, const float* __restrict__ vec, float (&out)[kSets]) {
  const int lane = lane_id(), h = lane >> 5, w = wave_id();
  out[0] = out[1] = 0.f;
#pragma unroll
  for (int t = 0; t < kTPW; ++t) {
    const int T = kTPW * w + t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 wv = load_feat4(vec, T, g, h);
#pragma unroll
      for (int s = 0; s < kSets; ++s) {
        f32x4 a = lds.act[s][4 * T + g][lane];
        out[s] += wv[0] * a[0] + wv[1] * a[1] + wv[2] * a[2] + wv[3] * a[3];
      }
    }
  }
#pragma unroll
  for (int s = 0; s < kSets; ++s) out[s] += __shfl_xor(out[s], 32);
}

// sdf of (set, j) -> lds.red[*][set][j][0] (sum the two wave slots and add b8_0 when reading).
__device__ __forceinline__ void sdf_head(Lds& lds, const SdfPacked& pk) {
  float part[kSets];
  head_partial(lds, pk.vecs + kVecW8, part);
  const int lane = lane_id(), w = wave_id();
  if (lane < 32) {
    lds.red[w][0][lane][0] = part[0];
    lds.red[w][1][lane][0] = part[1];
  }
}
__device__ __forceinline__ float read_sdf(const Lds& lds, const SdfPacked& pk, int set, int j) {
  float s = pk.vecs[kVecB80];
#pragma unroll
  for (int w = 0; w < kWaves; ++w) s += lds.red[w][set][j][0];
  return s;
}

// lin8 rows 1..256 (+bias) -> feat[(base + 32*set + j) * 256 + f]   (ONLY1: set 1 only, point base + j)
template <bool ONLY1 = false>
__device__ __forceinline__ void feature_head(const Lds& lds, const SdfPacked& pk, float* __restrict__ feat,
                                             long base, long M) {
  const int lane = lane_id(), h = lane >> 5, j = lane & 31, w = wave_id();
  f32x16 acc[kSets][kTPW];
  zero_acc(acc);
  gemm_phase<kTPW, ONLY1>(pk.mats, kF8, kTPW * w, lds, 0, acc);
  const float* bias = pk.vecs + kVecBias + 256 * 8;
#pragma unroll
  for (int s = ONLY1 ? 1 : 0; s < kSets; ++s) {
    const long P = base + (ONLY1 ? 0 : 32 * s) + j;
    if (P < M) {
#pragma unroll
      for (int t = 0; t < kTPW; ++t) {
        const int T = kTPW * w + t;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 b = load_feat4(bias, T, g, h);
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = acc[s][t][4 * g + c] + b[c];
          *reinterpret_cast<f32x4*>(feat + P * 256 + 32 * T + 8 * g + 4 * h) = o;
        }
      }
    }
  }
}

// acc (this wave's 4 tiles, both sets) *= sigma(100 z_l) recovered from the stored activations a_l,
// result -> LDS natural region.
__device__ __forceinline__ void epilogue_mul_dact(Lds& lds, f32x16 (&acc)[kSets][kTPW],
                                                  const f32x4* __restrict__ scratch_layer) {
  const int lane = lane_id(), w = wave_id();
#pragma unroll
  for (int t = 0; t < kTPW; ++t) {
    const int T = kTPW * w + t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int s = 0; s < kSets; ++s) {
        f32x4 a = spill_load(scratch_layer + ((s * 8 + T) * 4 + g) * 64 + lane);
        f32x4 q;
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = acc[s][t][4 * g + c] * dsoftplus_from_act(a[c]);
        lds.act[s][4 * T + g][lane] = q;
      }
    }
  }
}

// J_e^T v for the PE tile owned by wave W (slots 16W .. 16W+15); v in accpe registers.
template <int W>
__device__ __forceinline__ void pe_jacobian_t(const Lds& lds, const f32x16 (&accpe)[kSets][1], float (&n)[kSets][3]) {
  const int lane = lane_id(), h = lane >> 5;
#pragma unroll
  for (int s = 0; s < kSets; ++s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      constexpr int dummy = 0;
      (void)dummy;
      const int slot = 16 * W + r;
      if (slot >= 20) continue;
      const float v = accpe[s][0][r];
      if (slot == 0) {
        n[s][0] += h ? 0.f : v;
        n[s][1] += h ? v : 0.f;
      } else if (slot == 1) {
        n[s][2] += h ? 0.f : v;
      } else {
        const int k = (slot - 2) / 3, c = (slot - 2) % 3;
        const float other = lds.act[s][kActKB + (slot >> 2)][lane ^ 32][slot & 3];
        const float f = (float)(1 << k);
        // half 0 holds the sin slot: d sin = f cos (cos is the other half's value); half 1: d cos = -f sin
        n[s][c] += (h ? -f : f) * other * v;
      }
    }
  }
}

// Reverse sweep for d sdf / d x.  Precondition: a7 in the LDS natural region, a0..a7 in scratch,
// PE region intact.  Leaves the normal of (set, j) in lds.red[*][set][j][1..3] (sum the two wave slots).
template <bool ONLY1 = false>
__device__ __forceinline__ void normal_reverse(Lds& lds, const SdfPacked& pk, const f32x4* __restrict__ scratch) {
  const int lane = lane_id(), h = lane >> 5, w = wave_id();
  // q7 = w8 row (.) sigma(100 z7)
#pragma unroll
  for (int t = 0; t < kTPW; ++t) {
    const int T = kTPW * w + t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 wv = load_feat4(pk.vecs + kVecW8, T, g, h);
#pragma unroll
      for (int s = 0; s < kSets; ++s) {
        f32x4 a = lds.act[s][4 * T + g][lane];
        f32x4 q;
#pragma unroll
        for (int c = 0; c < 4; ++c) q[c] = wv[c] * dsoftplus_from_act(a[c]);
        lds.act[s][4 * T + g][lane] = q;
      }
    }
  }
  __syncthreads();
  f32x16 acc[kSets][kTPW];
  f32x16 accpe[kSets][1];
  zero_acc(accpe);
  f32x4 pre[kTPW];
  load_first_block<kTPW>(pk.mats, kRev[7], kTPW * w, pre);
#pragma nounroll
  for (int l = 7; l >= 1; --l) {
    zero_acc(acc);
    gemm_phase_pre<kTPW, ONLY1, true>(pk.mats, kRev[l], kTPW * w, lds, 0, acc, pre, l >= 2, kRev[l >= 2 ? l - 1 : l]);
#if PIR_PE_KSPLIT
    if (l == 4) gemm_phase<1, ONLY1>(pk.mats, pe_khalf(kR4P, w >> 1), w & 1, lds, (kActKB / 2) * (w >> 1), accpe);
#else
    if (l == 4 && w < 2) gemm_phase<1, ONLY1>(pk.mats, kR4P, w, lds, 0, accpe);
#endif
    PIR_STAMP(7);
    __syncthreads();
    PIR_STAMP(2);
    PIR_PRIO_HI();
    epilogue_mul_dact(lds, acc, scratch + (size_t)(l - 1) * kScratchLayerF4);
    PIR_PRIO_LO();
    PIR_STAMP(6);
    __syncthreads();
    PIR_STAMP(4);
  }
  float n[kSets][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#if PIR_PE_KSPLIT
  // the two PE tiles over the two halves of K: wave w = (K half w / 2, tile w % 2); the encoding Jacobian is linear
  // in the accumulator, so every wave transforms its partial sum and read_normal adds the four
  gemm_phase<1, ONLY1>(pk.mats, pe_khalf(kR0P, w >> 1), w & 1, lds, (kActKB / 2) * (w >> 1), accpe);
  if ((w & 1) == 0) pe_jacobian_t<0>(lds, accpe, n); else pe_jacobian_t<1>(lds, accpe, n);
#else
  if (w < 2) gemm_phase<1, ONLY1>(pk.mats, kR0P, w, lds, 0, accpe);   // the two PE tiles belong to waves 0 and 1
  if (w == 0) pe_jacobian_t<0>(lds, accpe, n); else if (w == 1) pe_jacobian_t<1>(lds, accpe, n);
#endif
#pragma unroll
  for (int s = 0; s < kSets; ++s)
#pragma unroll
    for (int c = 0; c < 3; ++c) n[s][c] += __shfl_xor(n[s][c], 32);
  if (lane < 32 && (PIR_PE_KSPLIT || w < 2)) {
#pragma unroll
    for (int s = 0; s < kSets; ++s)
#pragma unroll
      for (int c = 0; c < 3; ++c) lds.red[w][s][lane][1 + c] = n[s][c];
  }
}
__device__ __forceinline__ float read_normal(const Lds& lds, int set, int j, int c) {
#if PIR_PE_KSPLIT
  return (lds.red[0][set][j][1 + c] + lds.red[1][set][j][1 + c]) + (lds.red[2][set][j][1 + c] + lds.red[3][set][j][1 + c]);
#else
  return lds.red[0][set][j][1 + c] + lds.red[1][set][j][1 + c];
#endif
}

}  // namespace pir
