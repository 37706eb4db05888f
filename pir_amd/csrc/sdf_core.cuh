// SDF-MLP core for gfx950: a workgroup of FOUR wavefronts (256 threads, two workgroups per CU = two waves per
// SIMD) pushes two "operand sets" of 32 points each through the 8x256 softplus MLP of PIR's SDFNetwork
// (reference: src/network.py:8-142) on the exact-f32 MFMA v_mfma_f32_32x32x2_f32.
//
// Orientation: activations are the B operand (k = feature, column = point), weights the A
// operand (row = output neuron), so a layer's 32x32 accumulator tile has the point on the
// lane and the neuron in the register -- exactly the B-operand shape of the next layer.
// Activations live in LDS as [k-step][lane] images (one ds_read_b128 = 4 k-steps), weights
// stream from L2 in a pre-packed fragment order (one global_load_dwordx4 = 4 k-steps).
// Each wave owns 2 of the 8 output tiles for both sets: 64 accumulator registers.
// Measured alternatives (round 2): with the weight stream compiled out (every k-block re-reads
// one resident 1 KB block) the forward gains 1.6 % and get_all 1.9 %, the ceiling for any LDS weight-staging scheme;
// point loads from SoA planes instead of the reference's AoS [M,3], or no point loads at all, change nothing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pir {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef PIR_WAVES
#define PIR_WAVES 4
#endif
constexpr int kWaves = PIR_WAVES;   // waves per workgroup (2 or 4); two workgroups per CU -> kWaves/2 waves per SIMD
constexpr int kThreads = 64 * kWaves;
constexpr int kWavesPerSimd = kWaves / 2;
constexpr int kTilePts = 32;       // points per operand set
constexpr int kSets = 2;           // operand sets per workgroup
constexpr int kWgPts = 64;         // points per workgroup when both sets carry points
constexpr int kTPW = 8 / kWaves;   // output tiles per wave
constexpr int kActKB = 32;         // k-blocks (4 k-steps = 8 features) of the 256-feature region
constexpr int kPeKB = 5;           // k-blocks of the positional-encoding region (20 slots x 2 halves)
constexpr int kSetKB = kActKB + kPeKB;
constexpr int kHidden = 256;
constexpr int kSkipKeep = 217;     // outputs of lin3 (256 - 39)
constexpr int kPeDim = 39;
constexpr int kFeat = 256;         // feature outputs of lin8 (d_out - 1)
// lin4's input is [a3 (217) | PE (39)] / sqrt2 (src/network.py:90-91).  lin3 is padded to 256 outputs, so round 1 ran
// lin4 over the whole activation image plus the PE region: 128 + 20 k-steps, 16 of them (k-blocks 28..31) all zero.
// Now lin4 is two back-to-back GEMM phases into the same accumulators -- k-blocks 0..27 of the activation image
// (features 0..223), then the 5 k-blocks of the PE region: the same products in the same order (bit-identical
// results), 33 k-blocks instead of 37 (forward 30.15 -> 29.71 ms per 4 M points).  -DPIR_SKIP_TWOPHASE=0 = round 1.
// Measured and not kept: carrying the 39 encoding rows of lin4^T in its dead rows 217..255 and parking them from
// the epilogue instead of the separate two-tile product of waves 0 and 1 -- that product is only 0.7 % of the
// reverse sweep (the co-resident workgroup fills the matrix pipe meanwhile) and the parking stores cost more:
// get_all 62.2 -> 64.9 ms per 4 M points, bwd_sweep 35.8 -> 36.4 ms per 1 M.
#ifndef PIR_SKIP_TWOPHASE
#define PIR_SKIP_TWOPHASE 1
#endif
constexpr bool kSkipTwoPhase = PIR_SKIP_TWOPHASE != 0;
constexpr int kSkipNatKB = kSkipTwoPhase ? 28 : kActKB;   // k-blocks of lin4's input taken from the activation image
constexpr int kF4KB = kSkipNatKB + kPeKB;

// ---- packed weight stream ---------------------------------------------------------------
// A product is stored as f32x4 [kb][tile][lane]; element c of the f32x4 is k-step 4*kb+c.
// Offsets (in f32x4 units) into the one packed buffer, fixed by the architecture.
constexpr int kTileKB = 64;  // f32x4 per (kb, tile)
struct Prod { int off; int nkb; int ntiles; };
constexpr int prod_size(int nkb, int ntiles) { return nkb * ntiles * kTileKB; }

constexpr Prod kF0 = {0, kPeKB, 8};
constexpr Prod kF1 = {kF0.off + prod_size(kF0.nkb, 8), kActKB, 8};
constexpr Prod kF2 = {kF1.off + prod_size(kActKB, 8), kActKB, 8};
constexpr Prod kF3 = {kF2.off + prod_size(kActKB, 8), kActKB, 8};
constexpr Prod kF4 = {kF3.off + prod_size(kActKB, 8), kF4KB, 8};
constexpr Prod kF4a = {kF4.off, kSkipNatKB, 8};                          // ... its activation part
constexpr Prod kF4b = {kF4.off + prod_size(kSkipNatKB, 8), kPeKB, 8};    // ... its encoding part
constexpr Prod kF5 = {kF4.off + prod_size(kF4KB, 8), kActKB, 8};
constexpr Prod kF6 = {kF5.off + prod_size(kActKB, 8), kActKB, 8};
constexpr Prod kF7 = {kF6.off + prod_size(kActKB, 8), kActKB, 8};
constexpr Prod kF8 = {kF7.off + prod_size(kActKB, 8), kActKB, 8};       // feature rows 1..256 of lin8
// transposed products (A = W^T): row = input feature of the layer, k = output neuron
constexpr Prod kR8 = {kF8.off + prod_size(kActKB, 8), kActKB, 8};       // lin8 feature rows, transposed
constexpr Prod kR7 = {kR8.off + prod_size(kActKB, 8), kActKB, 8};
constexpr Prod kR6 = {kR7.off + prod_size(kActKB, 8), kActKB, 8};
constexpr Prod kR5 = {kR6.off + prod_size(kActKB, 8), kActKB, 8};
constexpr Prod kR4 = {kR5.off + prod_size(kActKB, 8), kActKB, 8};
constexpr Prod kR4P = {kR4.off + prod_size(kActKB, 8), kActKB, 2};      // PE rows of lin4^T
// lin3^T: k = the 217 outputs of lin3 -> k-blocks 0..27 (round 1 also ran the four all-zero k-blocks 28..31)
constexpr int kR3KB = kSkipTwoPhase ? kSkipNatKB : kActKB;
constexpr Prod kR3 = {kR4P.off + prod_size(kActKB, 2), kR3KB, 8};
constexpr Prod kR2 = {kR3.off + prod_size(kR3KB, 8), kActKB, 8};
constexpr Prod kR1 = {kR2.off + prod_size(kActKB, 8), kActKB, 8};
constexpr Prod kR0P = {kR1.off + prod_size(kActKB, 8), kActKB, 2};      // PE rows of lin0^T
constexpr int kPackedF4 = kR0P.off + prod_size(kActKB, 2);              // total f32x4 count

// The two transposed products onto the encoding rows (kR4P, kR0P) have two tiles for four waves.  1 (default): every
// wave takes one tile over one half of K and the partial results are added where they are consumed; 0: waves 0 and 1
// run the whole K while waves 2 and 3 wait at the next barrier (rounds 1-2).
#ifndef PIR_PE_KSPLIT
#define PIR_PE_KSPLIT 1
#endif
__host__ __device__ constexpr Prod pe_khalf(Prod p, int half) {
  return Prod{p.off + half * (p.nkb / 2) * p.ntiles * kTileKB, p.nkb / 2, p.ntiles};
}

// vectors appended after the matrices (floats): bias[9][256], w8row[256]
constexpr int kVecBias = 0;
constexpr int kVecW8 = 9 * 256;
constexpr int kVecB80 = 10 * 256;     // bias of the sdf output (lin8 row 0)
constexpr int kVecFloats = 10 * 256 + 4;

struct SdfPacked {
  const f32x4* mats;   // kPackedF4 entries
  const float* vecs;   // kVecFloats entries: bias of lin0..lin7, b8[1..256], w8 row 0, b8[0]
};


// ---- feature <-> (k-step, half) maps ---------------------------------------------------------
// C/D layout of the 32x32 MFMA: lane = 32*h + point, register r holds tile row (r&3) + 8*(r>>2) + 4*h.
__host__ __device__ constexpr int nat_feature(int s, int h) {
  return 32 * (s >> 4) + ((s & 15) & 3) + 8 * ((s & 15) >> 2) + 4 * h;
}
// PE slots: slot 0 = (x0, x1), slot 1 = (x2, pad), slot 2+3k+c = (sin(2^k x_c), cos(2^k x_c)).
// Returns the column of the 39-wide encoding (reference order, src/models/embedder.py:20-41) or -1.
__host__ __device__ constexpr int pe_column(int slot, int h) {
  if (slot == 0) return h;
  if (slot == 1) return h == 0 ? 2 : -1;
  if (slot >= 20) return -1;
  int k = (slot - 2) / 3, c = (slot - 2) % 3;
  return 3 + 6 * k + 3 * h + c;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Epilogues (VALU / memory phases) run at wave priority 1 so that they win issue arbitration against the co-resident
// wave's GEMM phase and get back to the matrix pipe sooner: get_all +0.8 %, bwd_sweep +1.7 % (-DPIR_EPI_PRIO=0 is the
// A/B switch; priority 3 measures the same as 1).
#ifndef PIR_EPI_PRIO
#define PIR_EPI_PRIO 1
#endif
#if PIR_EPI_PRIO > 0
#define PIR_PRIO_HI() __builtin_amdgcn_s_setprio(PIR_EPI_PRIO)
#define PIR_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define PIR_PRIO_HI()
#define PIR_PRIO_LO()
#endif

// ---- LDS image ------------------------------------------------------------------------------
struct Lds {
  f32x4 act[kSets][kSetKB][64];   // 2 x 37 KB
  float red[kWaves][kSets][kTilePts][4];  // cross-wave reductions (sdf: all waves; normal / dx: waves 0,1)
};
static_assert(sizeof(Lds) <= 80 * 1024, "two workgroups per CU must fit in 160 KB");

// ---- in-kernel stamps (diagnosis build -DPIR_STAMPS only; tools/dev/stamps.py reads them) ------------------------
// Lane 0 of every wave adds the shader cycles since its previous stamp to one of 8 LDS counters: 1 GEMM phase,
// 2 barrier after it, 3 epilogue, 4 barrier after the epilogue, 5 everything else of a tile, 6 reverse epilogue, 7 reverse GEMM phase (1 and 3 = forward).
#ifdef PIR_STAMPS
__shared__ long long g_stamp[kWaves][8];
__device__ __forceinline__ void pir_stamp(int i) {
  if (lane_id() == 0) {
    const long long t = (long long)__builtin_amdgcn_s_memtime();
    long long* s = g_stamp[wave_id()];
    s[i] += t - s[0];
    s[0] = t;
  }
}
#define PIR_STAMP(i) pir_stamp(i)
#else
#define PIR_STAMP(i)
#endif

// ---- the GEMM phase ---------------------------------------------------------------------------
// acc[set][t] += W[tile 4*wave+t (or given first tile)] x act[set], over nkb k-blocks starting at LDS
// k-block lds_kb0.  NT tiles for this wave starting at global tile `tile0`.
template <int NT, bool ONLY1 = false>
__device__ __forceinline__ void mfma_block(const f32x4 (&w)[NT], const f32x4 (&b)[kSets], f32x16 (&acc)[kSets][NT]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!ONLY1) acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t][c], b[0][c], acc[0][t], 0, 0, 0);
      acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t][c], b[1][c], acc[1][t], 0, 0, 0);
    }
  }
}

// Software pipeline: the operands of k-block kb+1 (weights from L2, activations from LDS) are issued
// before the 8*NT MFMAs of k-block kb; sched_barrier pins that order (hipcc otherwise sinks the loads
// below the MFMAs and waits vmcnt(0) on them at once).
// PRE: the first weight block is already in `pre` (loaded by the previous phase, so its L2 latency hid under
// that phase's epilogue and barriers); on return `pre` holds the first block of product `next` if
// `prefetch_next` (a wave-uniform flag).
// ONLY1: multiply operand set 1 only (set 0's accumulators are left untouched).
template <int NT>
__device__ __forceinline__ void load_first_block(const f32x4* __restrict__ mats, Prod p, int tile0, f32x4 (&pre)[NT]) {
  const f32x4* w = mats + p.off + (size_t)tile0 * kTileKB + lane_id();
#pragma unroll
  for (int t = 0; t < NT; ++t) pre[t] = w[t * kTileKB];
}

template <int NT, bool ONLY1 = false, bool PRE = false>
__device__ __forceinline__ void gemm_phase_pre(const f32x4* __restrict__ mats, Prod p, int tile0, const Lds& lds,
                                               int lds_kb0, f32x16 (&acc)[kSets][NT], f32x4 (&pre)[NT],
                                               bool prefetch_next, Prod next) {
  const int lane = lane_id();
  const f32x4* w = mats + p.off + (size_t)tile0 * kTileKB + lane;
  int kb_stride = p.ntiles * kTileKB;
  const int last = p.nkb - 1;
#ifdef PIR_ABLATE_WLOAD   // diagnosis build: every k-block re-reads the same 1 KB (no L2 weight stream)
  w = mats + lane;
  kb_stride = 0;
#endif
  f32x4 wa[NT], wb[NT], ba[kSets], bb[kSets];
#pragma unroll
  for (int t = 0; t < NT; ++t) wa[t] = PRE ? pre[t] : w[t * kTileKB];
  ba[0] = lds.act[0][lds_kb0][lane];
  ba[1] = lds.act[1][lds_kb0][lane];
  int kb = 0;
#pragma nounroll
  for (; kb < last; kb += 2) {
    {
      const f32x4* wq = w + (size_t)(kb + 1) * kb_stride;
#pragma unroll
      for (int t = 0; t < NT; ++t) wb[t] = wq[t * kTileKB];
      bb[0] = lds.act[0][lds_kb0 + kb + 1][lane];
      bb[1] = lds.act[1][lds_kb0 + kb + 1][lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_block<NT, ONLY1>(wa, ba, acc);
    __builtin_amdgcn_sched_barrier(0);
    {
      const int k2 = kb + 2 <= last ? kb + 2 : last;
      const f32x4* wq = w + (size_t)k2 * kb_stride;
#pragma unroll
      for (int t = 0; t < NT; ++t) wa[t] = wq[t * kTileKB];
      ba[0] = lds.act[0][lds_kb0 + k2][lane];
      ba[1] = lds.act[1][lds_kb0 + k2][lane];
    }
    if (prefetch_next && kb + 2 >= last) load_first_block<NT>(mats, next, tile0, pre);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block<NT, ONLY1>(wb, bb, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kb == last) mfma_block<NT, ONLY1>(wa, ba, acc);
}

template <int NT, bool ONLY1 = false>
__device__ __forceinline__ void gemm_phase(const f32x4* __restrict__ mats, Prod p, int tile0, const Lds& lds,
                                           int lds_kb0, f32x16 (&acc)[kSets][NT]) {
  f32x4 pre[NT];
  gemm_phase_pre<NT, ONLY1, false>(mats, p, tile0, lds, lds_kb0, acc, pre, false, p);
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[kSets][NT]) {
#pragma unroll
  for (int s = 0; s < kSets; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;
}

// bias for tile T, register group g (registers 4g..4g+3) on lane half h: 4 consecutive features.
__device__ __forceinline__ f32x4 load_feat4(const float* __restrict__ v, int T, int g, int h) {
  return *reinterpret_cast<const f32x4*>(v + 32 * T + 8 * g + 4 * h);
}

// softplus(beta=100, threshold=20) and its derivative sigma(100 z); reference: src/network.py:79.
//   softplus(z) = max(z,0) + ln2/100 * log2(1 + 2^(-|z| * 100 log2 e))
// on the bare v_exp_f32 / v_log_f32 (the log argument is in [1,2]: none of the denormal fix-ups that
// __expf/__logf expand to are needed).  For 100 z > 20 the log term is exactly 0 in fp32, which is PyTorch's
// threshold branch.  Absolute error of the activation ~1e-9.
__device__ __forceinline__ float softplus100(float z) {
#ifdef PIR_ABLATE_ACT
  return z;
#endif
  const float t = __builtin_amdgcn_exp2f(fabsf(z) * -144.26950408889634f);
  return fmaf(0.0069314718055994531f, __builtin_amdgcn_logf(1.0f + t), fmaxf(z, 0.f));
}
// sigma(100 z) recovered from a = softplus100(z): 1 - exp(-100 a)  (exact identity; 1 in the linear branch)
__device__ __forceinline__ float dsoftplus_from_act(float a) {
#ifdef PIR_ABLATE_ACT
  return a;
#endif
  return 1.f - __builtin_amdgcn_exp2f(a * -144.26950408889634f);
}

// sin/cos with a 3-term Cody-Waite reduction by pi/2 and the Cephes single-precision kernels
// (|error| ~1e-7 for |a| < ~1e4; PE arguments are at most 32 |x|).  Avoids the Payne-Hanek slow
// path that sincosf() inlines 18 times into every kernel.
__device__ __forceinline__ void sincos_pe(float a, float& sn, float& cs) {
  const float n = rintf(a * 0.636619772367581343f);
  float r = fmaf(n, -1.57079625129699707031e+00f, a);
  r = fmaf(n, -7.54978941586159635335e-08f, r);
  r = fmaf(n, -5.39030252995776476554e-15f, r);
  const float r2 = r * r;
  float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
  ps = fmaf(ps * r2, r, r);
  float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
  pc = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
  const int q = (int)n;
  const float s1 = (q & 1) ? pc : ps;
  const float c1 = (q & 1) ? ps : pc;
  sn = (q & 2) ? -s1 : s1;
  cs = ((q + 1) & 2) ? -c1 : c1;
}

// ---- positional encoding into the PE region of one set ------------------------------------------
// Lane (point j, half h) of wave `w` fills set w.  x = the point of (set w, j).
__device__ __forceinline__ void write_pe(Lds& lds, int set, float x0, float x1, float x2) {
  const int lane = lane_id();
  const int h = lane >> 5;
  float v[20];
  v[0] = h ? x1 : x0;
  v[1] = h ? 0.f : x2;
  const float xs[3] = {x0, x1, x2};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float sn, cs;
      sincos_pe(xs[c] * (float)(1 << k), sn, cs);
      v[2 + 3 * k + c] = h ? cs : sn;
    }
  }
#pragma unroll
  for (int q = 0; q < kPeKB; ++q) {
    f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    lds.act[set][kActKB + q][lane] = o;
  }
}

// Activation spill of get_all (written by the forward sweep, read back once by the reverse sweep of the same tile).
// PIR_SPILL_NT_ST / PIR_SPILL_NT_LD: non-temporal accesses, so that the stream does not displace the packed weights
// (2 x 2.1 MB, re-read by every tile) from the 4 MB L2 of the XCD.
__device__ __forceinline__ void spill_store(f32x4* __restrict__ p, f32x4 v) {
#ifdef PIR_SPILL_NT_ST
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ f32x4 spill_load(const f32x4* __restrict__ p) {
#ifdef PIR_SPILL_NT_LD
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// Hidden-layer epilogue: a = softplus(acc + bias) -> LDS natural region of both sets.
// Optionally also stores the activations to a private scratch image (float4 per lane, coalesced).
template <bool STORE>
__device__ __forceinline__ void epilogue_softplus(Lds& lds, f32x16 (&acc)[kSets][kTPW], const float* __restrict__ bias,
                                                  f32x4* __restrict__ scratch /* [set][T][g][lane] */) {
  const int lane = lane_id(), h = lane >> 5, w = wave_id();
#pragma unroll
  for (int t = 0; t < kTPW; ++t) {
    const int T = kTPW * w + t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 b = load_feat4(bias, T, g, h);
#pragma unroll
      for (int s = 0; s < kSets; ++s) {
        f32x4 a;
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = softplus100(acc[s][t][4 * g + c] + b[c]);
        lds.act[s][4 * T + g][lane] = a;
        if (STORE) spill_store(scratch + ((s * 8 + T) * 4 + g) * 64 + lane, a);
      }
    }
  }
}

}  // namespace pir
