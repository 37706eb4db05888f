// Weight-gradient GEMM of the SDF backward:  dW_l += Zb_l H_l^T + Q_l Pb_l^T  (see sdf_backward.hip), a
// [out x points] x [points x in] product with the reduction over the points of a chunk.  Split-K over
// workgroups into fp32 slabs, then a deterministic slab reduction that accumulates into the caller's
// gradient buffers (no float atomics: bitwise reproducible gradients).
#include <hip/hip_runtime.h>
#include "sdf_core.cuh"
#include "sdf_backward.h"
#include "wgrad.h"
#include "pir_common.h"
#include "host_logic.h"

namespace pir {

constexpr int kWgThreads = 256;
constexpr int kWgTile = 128;     // C tile edge
constexpr int kWgKPts = 32;      // points per K step staged in LDS

struct TileRegs { f32x4 v[4]; };

// 32 rows x 128 floats of a [rows][ld] array -> registers (thread t holds float4 #(q*256 + t), q < 4)
__device__ __forceinline__ void load_tile(TileRegs& t, const float* __restrict__ src, int ld, int col0, int row0,
                                          int rows_valid) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * kWgThreads + threadIdx.x;
    const int r = idx >> 5, c4 = idx & 31;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col0 + 4 * c4 < ld && (rows_valid < 0 || row0 + r < rows_valid))
      v = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + r) * ld + col0 + 4 * c4);
    t.v[q] = v;
  }
}
__device__ __forceinline__ void store_tile(float (*dst)[kWgTile], const TileRegs& t) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * kWgThreads + threadIdx.x;
    *reinterpret_cast<f32x4*>(&dst[idx >> 5][4 * (idx & 31)]) = t.v[q];
  }
}

#ifdef PIR_WG_LDSDMA
// ---- A/B variant (round 2, slower: 19.3 vs 18.2 ms per 1 M points): operand staging by LDS-DMA (global_load_lds_dwordx4): no register hop, two LDS stages, one barrier per step.
// A wave instruction moves 64 lanes x 16 B = two 512-byte tile rows; the LDS side is wave-uniform base + 16 * lane,
// which is exactly the row-major [32][128] tile, the global side is per lane.  Lanes outside the operand (columns
// beyond a 64-wide array, rows beyond a caller-owned array) read a zero quad instead.
__device__ __align__(16) const float kZeroQuad[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma_tile(float (*dst)[kWgTile], const float* __restrict__ src, int ld, int col0, int row0,
                                         int rows_valid) {
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * kWgThreads + threadIdx.x;
    const int r = idx >> 5, c4 = idx & 31;
    const bool ok = col0 + 4 * c4 < ld && (rows_valid < 0 || row0 + r < rows_valid);
    const float* g = ok ? src + (size_t)(row0 + r) * ld + col0 + 4 * c4 : kZeroQuad;
    float* l = &dst[0][0] + (q * kWgThreads + 64 * wv) * 4;      // wave-uniform
    __builtin_amdgcn_global_load_lds((const void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  }
}

// One workgroup = one (job, K slice): C tile 128x128 over the slice's points, both products of the job.
// The operands of step s+1 stream into the other LDS stage while the MFMAs of step s run.
__global__ void __launch_bounds__(kWgThreads, 2)
wgrad_kernel(WgJobs jobs, int n_rows /* multiple of 32 */, int ksplit, float* __restrict__ slabs,
             float* __restrict__ bias_slabs) {
  __shared__ float Xs[2][kWgKPts][kWgTile];
  __shared__ float Ys[2][kWgKPts][kWgTile];
  const int job_id = blockIdx.x / ksplit, ks = blockIdx.x % ksplit;
  const WgJob jb = jobs.j[job_id];
  const int total_it = n_rows / kWgKPts;
  const int it0 = (int)((long)ks * total_it / ksplit), it1 = (int)((long)(ks + 1) * total_it / ksplit);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1, lo = lane & 31, kh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum = 0.f;
  const int ncols_y = jb.ldy - jb.i0 < kWgTile ? jb.ldy - jb.i0 : kWgTile;
  const bool wave_has_cols = 64 * wc < ncols_y;
  const int nprod = jb.X2 ? 2 : 1;
  const int nsteps = (it1 - it0) * nprod;
  auto fetch = [&](int step) {
    const int prod = step % nprod, row0 = (it0 + step / nprod) * kWgKPts, buf = step & 1;
    dma_tile(Xs[buf], prod ? jb.X2 : jb.X1, prod ? 256 : jb.ldx, jb.o0, row0, prod ? -1 : jb.x1_rows);
    dma_tile(Ys[buf], prod ? jb.Y2 : jb.Y1, jb.ldy, jb.i0, row0, prod ? -1 : jb.y1_rows);
  };
  if (nsteps > 0) {
    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int step = 0; step < nsteps; ++step) {
    const int buf = step & 1;
#ifndef PIR_ABL_WG_NOFETCH
    if (step + 1 < nsteps) fetch(step + 1);      // lands in the other stage while this one is multiplied
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (jb.bias && step % nprod == 0) {
      const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
#pragma unroll
      for (int r = 0; r < 16; ++r) bsum += Xs[buf][16 * half + r][col];
    }
    if (wave_has_cols) {
#pragma unroll
      for (int s = 0; s < kWgKPts / 2; ++s) {
        const float a0 = Xs[buf][2 * s + kh][64 * wr + lo], a1 = Xs[buf][2 * s + kh][64 * wr + 32 + lo];
        const float b0 = Ys[buf][2 * s + kh][64 * wc + lo], b1 = Ys[buf][2 * s + kh][64 * wc + 32 + lo];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // this wave's DMA for the next step has landed; after the barrier everybody's has, and nobody still reads `buf`
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef PIR_ABL_WG_NOBAR
    __syncthreads();
#endif
  }
  float* slab = slabs + ((size_t)ks * jobs.n + job_id) * kWgTile * kWgTile;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 64 * wr + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int col = 64 * wc + 32 * b + lo;
        slab[row * kWgTile + col] = acc[a][b][r];
      }
  if (jb.bias) {
    float* red = &Xs[0][0][0];
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < 128)
      bias_slabs[((size_t)ks * jobs.n + job_id) * kWgTile + threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + 128];
  }
}
#else   // default: global -> registers -> LDS with the register prefetch one step ahead
// One workgroup = one (job, K slice): C tile 128x128 over the slice's points, both products of the job.
// The operands of step s+1 are fetched into registers while the MFMAs of step s run from LDS.
__global__ void __launch_bounds__(kWgThreads)
wgrad_kernel(WgJobs jobs, int n_rows /* multiple of 32 */, int ksplit, float* __restrict__ slabs,
             float* __restrict__ bias_slabs) {
  __shared__ float Xs[kWgKPts][kWgTile];
  __shared__ float Ys[kWgKPts][kWgTile];
  const int job_id = blockIdx.x / ksplit, ks = blockIdx.x % ksplit;
  const WgJob jb = jobs.j[job_id];
  const int total_it = n_rows / kWgKPts;
  const int it0 = (int)((long)ks * total_it / ksplit), it1 = (int)((long)(ks + 1) * total_it / ksplit);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1, lo = lane & 31, kh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum = 0.f;
  const int ncols_y = jb.ldy - jb.i0 < kWgTile ? jb.ldy - jb.i0 : kWgTile;
  const bool wave_has_cols = 64 * wc < ncols_y;
  const int nprod = jb.X2 ? 2 : 1;
  const int nsteps = (it1 - it0) * nprod;
  TileRegs tx, ty;
  auto fetch = [&](int step) {
    const int prod = step % nprod, row0 = (it0 + step / nprod) * kWgKPts;
    load_tile(tx, prod ? jb.X2 : jb.X1, prod ? 256 : jb.ldx, jb.o0, row0, prod ? -1 : jb.x1_rows);
    load_tile(ty, prod ? jb.Y2 : jb.Y1, jb.ldy, jb.i0, row0, prod ? -1 : jb.y1_rows);
  };
  if (nsteps > 0) fetch(0);
  for (int step = 0; step < nsteps; ++step) {
#ifdef PIR_ABL_WG_NOSTAGE   // ablation: operands stay in LDS, no barriers, no refetch
    if (step == 0) { store_tile(Xs, tx); store_tile(Ys, ty); __syncthreads(); }
#else
    __syncthreads();
    store_tile(Xs, tx);
    store_tile(Ys, ty);
    __syncthreads();
#ifndef PIR_ABL_WG_NOFETCH
    if (step + 1 < nsteps) fetch(step + 1);
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (jb.bias && step % nprod == 0) {
      const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
#pragma unroll
      for (int r = 0; r < 16; ++r) bsum += Xs[16 * half + r][col];
    }
    if (wave_has_cols) {
#pragma unroll
      for (int s = 0; s < kWgKPts / 2; ++s) {
        const float a0 = Xs[2 * s + kh][64 * wr + lo], a1 = Xs[2 * s + kh][64 * wr + 32 + lo];
        const float b0 = Ys[2 * s + kh][64 * wc + lo], b1 = Ys[2 * s + kh][64 * wc + 32 + lo];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float* slab = slabs + ((size_t)ks * jobs.n + job_id) * kWgTile * kWgTile;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 64 * wr + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int col = 64 * wc + 32 * b + lo;
        slab[row * kWgTile + col] = acc[a][b][r];
      }
  if (jb.bias) {
    __syncthreads();
    float* red = &Xs[0][0];
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < 128)
      bias_slabs[((size_t)ks * jobs.n + job_id) * kWgTile + threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + 128];
  }
}

#endif

// row 0 of dW8 and db8[0]:  sum_pts (g_sdf a7 + a7-tangent),  sum_pts g_sdf   -> partial[block][257]
constexpr int kRow0Blocks = 1024;
#if PIR_BWD_FRAG
// A7 / Ad7 in fragment order: wave w owns k-blocks 8w .. 8w+7 (columns 64w .. 64w+63), lane (h, j) reads one float4 =
// columns 8 kb + 4 h + 0..3 of point j per k-block and tile (1 KB per wave instruction); the 32 points of a tile are
// summed across lanes at the end.  n_rows = the padded row count (multiple of 32); rows >= chunk_n carry zero tangents.
__global__ void __launch_bounds__(256)
row0_kernel(const float* __restrict__ A7, const float* __restrict__ Ad7, const float* __restrict__ g_sdf /* chunk-local or null */,
            int chunk_n, int n_rows, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const int ntiles = n_rows / 32;
  float s[8][4];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) s[k][c] = 0.f;
  float sg = 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row = tile * 32 + j;
    const float g = (g_sdf && row < chunk_n) ? g_sdf[row] : 0.f;
    sg += g;
    const f32x4* __restrict__ a = reinterpret_cast<const f32x4*>(A7) + ((size_t)tile * 32 + 8 * w) * 64 + lane;
    const f32x4* __restrict__ d = reinterpret_cast<const f32x4*>(Ad7) + ((size_t)tile * 32 + 8 * w) * 64 + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f32x4 av = a[k * 64], dv = d[k * 64];
#pragma unroll
      for (int c = 0; c < 4; ++c) s[k][c] += g * av[c] + dv[c];
    }
  }
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) {
    sg += __shfl_xor(sg, m);
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) s[k][c] += __shfl_xor(s[k][c], m);
  }
  if (j == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) partial[blockIdx.x * 257 + 64 * w + 8 * k + 4 * h + c] = s[k][c];
    if (w == 0 && h == 0) partial[blockIdx.x * 257 + 256] = sg;
  }
}
#else
__global__ void __launch_bounds__(256)
row0_kernel(const float* __restrict__ A7, const float* __restrict__ Ad7, const float* __restrict__ g_sdf /* chunk-local or null */,
            int chunk_n, int /*n_rows*/, float* __restrict__ partial) {
  const int col = threadIdx.x;
  const int r0 = (int)((long)blockIdx.x * chunk_n / gridDim.x), r1 = (int)((long)(blockIdx.x + 1) * chunk_n / gridDim.x);
  float s = 0.f, sg = 0.f;
  int r = r0;
  for (; r + 4 <= r1; r += 4) {   // 8 independent loads in flight per thread
    float a[4], d[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = A7[(size_t)(r + u) * 256 + col];
      d[u] = Ad7[(size_t)(r + u) * 256 + col];
      g[u] = g_sdf ? g_sdf[r + u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s += g[u] * a[u] + d[u]; sg += g[u]; }
  }
  for (; r < r1; ++r) {
    const float g = g_sdf ? g_sdf[r] : 0.f;
    s += g * A7[(size_t)r * 256 + col] + Ad7[(size_t)r * 256 + col];
    sg += g;
  }
  partial[blockIdx.x * 257 + col] = s;
  if (col == 0) partial[blockIdx.x * 257 + 256] = sg;
}
#endif

__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(WgJobs jobs, int ksplit, const float* __restrict__ slabs, const float* __restrict__ bias_slabs,
                    const float* __restrict__ row0_partial, GradOut out) {
  constexpr int kParts = 16;   // blocks per C tile (1024 elements each)
  const int job_id = blockIdx.x / kParts, part = blockIdx.x % kParts;
  if (job_id == jobs.n) {   // the row-0 blocks (SDF only; launched only when row0_partial is given): 257 columns
    for (int col = part * 256 + threadIdx.x; col < 257; col += kParts * 256) {
      float s = 0.f;
      for (int b = 0; b < kRow0Blocks; ++b) s += row0_partial[b * 257 + col];
      if (col < 256) out.w[8][col] += s; else out.b[8][0] += s;
    }
    return;
  }
  const WgJob jb = jobs.j[job_id];
  const size_t stride = (size_t)jobs.n * kWgTile * kWgTile;
  const float* base = slabs + (size_t)job_id * kWgTile * kWgTile;
  for (int e = part * (kWgTile * kWgTile / kParts) + threadIdx.x; e < (part + 1) * (kWgTile * kWgTile / kParts); e += 256) {
    const int o = jb.o0 + e / kWgTile, i = jb.i0 + e % kWgTile;
    if (o >= jb.rows || i >= jb.i_limit) continue;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += base[k * stride + e];
    out.w[jb.layer][(size_t)(jb.row_off + o) * jb.cols + jb.col_off + i] += jb.scale * s;
  }
  if (jb.bias && part == 0 && threadIdx.x < kWgTile) {
    const int o = jb.o0 + threadIdx.x;
    if (o < jb.rows) {
      float s = 0.f;
      for (int k = 0; k < ksplit; ++k) s += bias_slabs[((size_t)k * jobs.n + job_id) * kWgTile + threadIdx.x];
      out.b[jb.layer][jb.row_off + o] += s;
    }
  }
}

// ---- 256 x 256 C tiles (a whole 256 x 256 layer per workgroup) ------------------------------------------------
// The 128 x 128 kernel above spends 21 % of its time staging operands: a 128 x 128 fp32 tile step moves 32 KB for
// 1 MFLOP (32 FLOP/B).  One workgroup of 8 waves per 256 x 256 tile moves 64 KB for 4 MFLOP per step: half the loads,
// LDS writes and barriers per MFMA.  Each wave owns a 64 (outputs) x 128 (inputs) block = 8 accumulator tiles.
// Used for the SDF layers whose operands are all 256 wide (layers 1..7 and the feature rows of lin8).
constexpr int kW2Threads = 512;
constexpr int kW2Tile = 256;

struct TileRegs2 { f32x4 v[4]; };
// LDS image of a 32-point step: Xs[point][256 + 4].  Rows 1 KB apart would make the 32 lanes that un-permute a
// fragment-order tile (one point each, same column) write to the same banks; the 16-byte pad shifts every row by four
// banks, so 16 consecutive points cover all 64.  Row-major sources and the MFMA operand reads (32 consecutive columns
// of one row per half wave) are conflict-free either way, and every row offset stays a compile-time immediate.
constexpr int kW2Ld = kW2Tile + 4;

// 32 rows x 256 floats of a [rows][256] array (thread t holds float4 #(q*512 + t), q < 4).
// frag: the array is in fragment order (sdf_backward.h): float4 #idx of the tile image = (k-block idx / 64, lane idx % 64)
__device__ __forceinline__ void load_tile2(TileRegs2& t, const float* __restrict__ src, int ld, int row0,
                                           int rows_valid, int frag) {
  if (frag) {
    const f32x4* __restrict__ img = reinterpret_cast<const f32x4*>(src + (size_t)row0 * 256);   // row0 % 32 == 0
#pragma unroll
#ifdef PIR_WG_NT_LD
    for (int q = 0; q < 4; ++q) t.v[q] = __builtin_nontemporal_load(&img[q * kW2Threads + threadIdx.x]);   // read once
#else
    for (int q = 0; q < 4; ++q) t.v[q] = img[q * kW2Threads + threadIdx.x];
#endif
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * kW2Threads + threadIdx.x;
    const int r = idx >> 6, c4 = idx & 63;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (4 * c4 < ld && (rows_valid < 0 || row0 + r < rows_valid))
      v = *reinterpret_cast<const f32x4*>(src + (size_t)(row0 + r) * ld + 4 * c4);
    t.v[q] = v;
  }
}
__device__ __forceinline__ void store_tile2(float (*dst)[kW2Ld], const TileRegs2& t, int frag) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = q * kW2Threads + threadIdx.x;
    int r, c4;
    if (frag) { const int kb = idx >> 6, lane = idx & 63; r = lane & 31; c4 = 2 * kb + (lane >> 5); }
    else { r = idx >> 6; c4 = idx & 63; }
    *reinterpret_cast<f32x4*>(&dst[r][4 * c4]) = t.v[q];
  }
}

// K slices per job: job j owns workgroups / slabs first[j] .. first[j+1]-1 (a job with a narrow operand does less
// work per step and gets proportionally fewer slices, so that all workgroups finish together).
struct Wg2Split { int first[12]; };

__global__ void __launch_bounds__(kW2Threads, 2)
wgrad256_kernel(WgJobs jobs, Wg2Split sp, int n_rows /* multiple of 32 */, float* __restrict__ slabs,
                float* __restrict__ bias_slabs) {
  __shared__ float Xs[kWgKPts][kW2Ld];
  __shared__ float Ys[kWgKPts][kW2Ld];
  int job_id = 0;
  while (job_id + 1 < jobs.n && (int)blockIdx.x >= sp.first[job_id + 1]) ++job_id;
  const int ks = blockIdx.x - sp.first[job_id], ksplit = sp.first[job_id + 1] - sp.first[job_id];
  const WgJob jb = jobs.j[job_id];
  const int total_it = n_rows / kWgKPts;
  const int it0 = (int)((long)ks * total_it / ksplit), it1 = (int)((long)(ks + 1) * total_it / ksplit);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wr = wv >> 1, wc = wv & 1, lo = lane & 31, kh = lane >> 5;
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum = 0.f;
  const int nprod = jb.X2 ? 2 : 1;
  const int nsteps = (it1 - it0) * nprod;
  TileRegs2 tx, ty;
  auto fetch_into = [&](int step, TileRegs2& fx_, TileRegs2& fy_) {
    const int prod = step % nprod, row0 = (it0 + step / nprod) * kWgKPts;
    load_tile2(fx_, prod ? jb.X2 : jb.X1, 256, row0, prod ? -1 : jb.x1_rows, jb.x_frag);
    load_tile2(fy_, prod ? jb.Y2 : jb.Y1, jb.ldy, row0, prod ? -1 : jb.y1_rows, jb.y_frag);
  };
  auto bias_rows = [&](int step) {
    if (jb.bias && step % nprod == 0) {
      const int col = threadIdx.x & 255, half = threadIdx.x >> 8;
#pragma unroll
      for (int r = 0; r < 16; ++r) bsum += Xs[16 * half + r][col];
    }
  };
  if (jb.ldy >= kW2Tile) {
    if (nsteps > 0) fetch_into(0, tx, ty);
    for (int step = 0; step < nsteps; ++step) {
      __syncthreads();
      store_tile2(Xs, tx, jb.x_frag);
      store_tile2(Ys, ty, jb.y_frag);
      __syncthreads();
      if (step + 1 < nsteps) fetch_into(step + 1, tx, ty);
      __builtin_amdgcn_sched_barrier(0);
      bias_rows(step);
#pragma unroll
      for (int s2 = 0; s2 < kWgKPts / 2; ++s2) {
        const int kr = 2 * s2 + kh;
        const float a0 = Xs[kr][64 * wr + lo], a1 = Xs[kr][64 * wr + 32 + lo];
        float b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = Ys[kr][128 * wc + 32 * t + lo];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[t], acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[t], acc[1][t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // The 64-wide encoding operand of layer 0: every wave takes 32 output rows x 64 columns, a quarter of the MFMAs
    // of a full-width step (1.7 us), which is less than the latency of the operand fetch.  Two register stages keep
    // the fetch two steps ahead (the accumulators of this path leave the registers free).
    TileRegs2 ux, uy;
    auto narrow_step = [&](int step, TileRegs2& fx_, TileRegs2& fy_) {
      __syncthreads();
      store_tile2(Xs, fx_, jb.x_frag);
      store_tile2(Ys, fy_, jb.y_frag);
      __syncthreads();
      if (step + 2 < nsteps) fetch_into(step + 2, fx_, fy_);
      __builtin_amdgcn_sched_barrier(0);
      bias_rows(step);
#pragma unroll
      for (int s2 = 0; s2 < kWgKPts / 2; ++s2) {
        const int kr = 2 * s2 + kh;
        const float a0 = Xs[kr][32 * wv + lo];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float bt = Ys[kr][32 * t + lo];
          acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bt, acc[0][t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (nsteps > 0) fetch_into(0, tx, ty);
    if (nsteps > 1) fetch_into(1, ux, uy);
    for (int step = 0; step < nsteps; step += 2) {
      narrow_step(step, tx, ty);
      if (step + 1 < nsteps) narrow_step(step + 1, ux, uy);
    }
  }
  float* slab = slabs + (size_t)blockIdx.x * kW2Tile * kW2Tile;
  if (jb.ldy >= kW2Tile) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 64 * wr + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const int col = 128 * wc + 32 * t + lo;
          slab[row * kW2Tile + col] = acc[a][t][r];
        }
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * kh;
        slab[row * kW2Tile + 32 * t + lo] = acc[0][t][r];
      }
  }
  if (jb.bias) {
    __syncthreads();
    float* red = &Xs[0][0];
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < 256)
      bias_slabs[(size_t)blockIdx.x * kW2Tile + threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + 256];
  }
}

__global__ void __launch_bounds__(256)
wgrad256_reduce_kernel(WgJobs jobs, Wg2Split sp, const float* __restrict__ slabs, const float* __restrict__ bias_slabs,
                       GradOut out) {
  constexpr int kParts = 64;   // blocks per C tile (1024 elements each)
  const int job_id = blockIdx.x / kParts, part = blockIdx.x % kParts;
  const WgJob jb = jobs.j[job_id];
  const int ksplit = sp.first[job_id + 1] - sp.first[job_id];
  const size_t stride = (size_t)kW2Tile * kW2Tile;
  const float* base = slabs + (size_t)sp.first[job_id] * kW2Tile * kW2Tile;
  for (int e = part * 1024 + threadIdx.x; e < (part + 1) * 1024; e += 256) {
    const int o = e / kW2Tile, i = e % kW2Tile;
    if (o >= jb.rows || i >= jb.i_limit) continue;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += base[k * stride + e];
    out.w[jb.layer][(size_t)(jb.row_off + o) * jb.cols + jb.col_off + i] += jb.scale * s;
  }
  if (jb.bias && part == 0) {
    const int o = threadIdx.x;
    if (o < jb.rows) {
      float s = 0.f;
      for (int k = 0; k < ksplit; ++k) s += bias_slabs[(size_t)(sp.first[job_id] + k) * kW2Tile + o];
      out.b[jb.layer][jb.row_off + o] += s;
    }
  }
}

// Jobs: whole-layer 256 x 256 products (o0 = i0 = 0, ldx = 256, ldy = 256 or 64).
int wgrad256_run(const WgJobs& jobs, int n_rows, int n_points, const WgFixed& fx, const GradOut& out, int prof_id,
                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (jobs.n < 1 || jobs.n > 10 || n_rows < kWgKPts || (n_rows % kWgKPts)) return PIR_ERR_ARG;
  // One resident workgroup per CU (64 KB LDS, 512 threads at 256 registers).  Work units per 32 points: 20 for a
  // job with two products over full-width operands, 10 for the 64-wide layer-0 operand (a quarter of the MFMAs, but
  // its steps stay bound by the operand fetch even with the fetch two steps ahead; measured optimum of 3..6), half
  // for a single product (the feature rows of lin8); every job gets K slices in proportion, as many as fill one
  // round of the CUs, so that all workgroups finish together.
  int cost[10];
  for (int j = 0; j < jobs.n; ++j) {
#ifndef PIR_W2_NARROW_COST
#define PIR_W2_NARROW_COST 5
#endif
    cost[j] = (jobs.j[j].ldy < kW2Tile ? PIR_W2_NARROW_COST : 10) * (jobs.j[j].X2 ? 2 : 1);
  }
  const int steps = n_rows / kWgKPts;
  const long slab_cap = (long)kKSplitMax * kMaxJobs * kWgTile * kWgTile / ((long)kW2Tile * kW2Tile);   // slabs that fit
  Wg2Split sp;
  const int n_wg = pir_host::wgrad256_split(jobs.n, cost, steps, pir_num_cus(), slab_cap, sp.first);
  {
    PirProfScope prof(prof_id, (double)n_points, stream_);
    hipLaunchKernelGGL(wgrad256_kernel, dim3(n_wg), dim3(kW2Threads), 0, stream, jobs, sp, n_rows, fx.slabs,
                       fx.bias_slabs);
  }
  hipLaunchKernelGGL(wgrad256_reduce_kernel, dim3(jobs.n * 64), dim3(256), 0, stream, jobs, sp, fx.slabs, fx.bias_slabs,
                     out);
  return pir_check_launch();
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr size_t kSlabBytes = (size_t)kKSplitMax * kMaxJobs * kWgTile * kWgTile * 4;
constexpr size_t kBiasSlabBytes = (size_t)kKSplitMax * kMaxJobs * kWgTile * 4;
constexpr size_t kRow0Bytes = (size_t)kRow0Blocks * 257 * 4;
constexpr size_t kFixedBytes = ((kSlabBytes + 255) & ~(size_t)255) + ((kBiasSlabBytes + 255) & ~(size_t)255) +
                               ((kRow0Bytes + 255) & ~(size_t)255);

size_t wgrad_fixed_bytes() { return kFixedBytes; }

WgFixed wgrad_carve(void* scratch) {
  char* p = (char*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
  WgFixed fx;
  fx.slabs = (float*)p; p += al256(kSlabBytes);
  fx.bias_slabs = (float*)p; p += al256(kBiasSlabBytes);
  fx.row0 = (float*)p; p += al256(kRow0Bytes);
  fx.rest = p;
  return fx;
}

int wgrad_run(const WgJobs& jobs, int n_rows, int n_points, const WgFixed& fx, const float* row0_partial,
              const GradOut& out, int prof_id, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (jobs.n < 1 || jobs.n > kMaxJobs || n_rows < kWgKPts || (n_rows % kWgKPts)) return PIR_ERR_ARG;
  // K slices of >= 8 staged steps each, their number chosen so that the workgroups fill whole rounds of the
  // 2-per-CU resident slots (30 jobs x 64 slices left the last round a quarter empty: -5.5 %)
  const int ksplit = pir_host::wgrad128_ksplit(jobs.n, n_rows, kWgKPts, 2L * pir_num_cus(), kKSplitMax);
  {
    PirProfScope prof(prof_id, (double)n_points, stream_);
    hipLaunchKernelGGL(wgrad_kernel, dim3(jobs.n * ksplit), dim3(kWgThreads), 0, stream, jobs, n_rows, ksplit,
                       fx.slabs, fx.bias_slabs);
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((jobs.n + (row0_partial ? 1 : 0)) * 16), dim3(256), 0, stream, jobs,
                     ksplit, fx.slabs, fx.bias_slabs, row0_partial, out);
  return pir_check_launch();
}

}  // namespace pir

using namespace pir;

extern "C" {

size_t pir_sdf_backward_scratch_bytes(long chunk_points) {
  const size_t rows = (size_t)((chunk_points + 31) / 32) * 32;
  return kFixedBytes + rows * kBwdRowBytes + 256;
}

int pir_sdf_get_all_backward(const void* packed, const float* x, long M, const float* g_sdf, const float* g_feat,
                             const float* g_normal, float* g_x, float* const* gw9, float* const* gb9, void* scratch,
                             size_t scratch_bytes, void* stream_) {
  if (M < 0 || !packed || !gw9 || !gb9) return PIR_ERR_ARG;
  if (M == 0) return PIR_OK;
  if (!x || !scratch) return PIR_ERR_ARG;
  const long cap = pir_host::bwd_chunk_cap(scratch_bytes, kFixedBytes, kBwdRowBytes, 262144);
  if (cap <= 0) return PIR_ERR_SCRATCH;
  hipStream_t stream = (hipStream_t)stream_;
  const WgFixed fx = wgrad_carve(scratch);
  float* slabs = fx.slabs;
  float* row0 = fx.row0;
  char* p = fx.rest;
  BwdScratch sc;
  float* f = (float*)p;
  sc.layer_stride = (size_t)cap * 256;
  sc.A_ = f; f += 8 * sc.layer_stride;
  sc.Ad_ = f; f += 8 * sc.layer_stride;
  sc.Zb_ = f; f += 8 * sc.layer_stride;
  sc.Q_ = f; f += 8 * sc.layer_stride;
  sc.E = f; f += cap * 64;
  sc.Ed = f;
  sc.pe_tmp = slabs;   // the slabs are idle while the sweep runs (stream order): 32 KB per resident workgroup
  GradOut out;
  for (int l = 0; l < 9; ++l) { out.w[l] = gw9[l]; out.b[l] = gb9[l]; if (!out.w[l] || !out.b[l]) return PIR_ERR_ARG; }
  static const int rows[9] = {256, 256, 256, kSkipKeep, 256, 256, 256, 256, 257};
  static const int cols[9] = {kPeDim, 256, 256, 256, 256, 256, 256, 256, 256};
  for (long base = 0; base < M; base += cap) {
    const int n = (int)(M - base < cap ? M - base : cap);
    const int n_rows = (n + 31) / 32 * 32;
    int rc = pir_internal_bwd_sweep(packed, x, g_sdf, g_feat, g_normal, base, n, &sc, g_x, stream_);
    if (rc != PIR_OK) return rc;
    WgJobs jobs, big;     // whole-layer 256 x 256 tiles (`big`); the 128 x 128 job list stays for the A/B build
    jobs.n = 0;
    big.n = 0;
#ifdef PIR_WG_TILE128      // A/B: the round-1 tiling for every layer
    const int first_big = 9;
#else
    const int first_big = 0;
#endif
    for (int l = 0; l < 8; ++l) {
      if (l >= first_big) {
        WgJob& j = big.j[big.n++];
        j.X1 = sc.Zb(l); j.X2 = sc.Q(l);
        j.Y1 = l == 0 ? sc.E : sc.A(l - 1);
        j.Y2 = l == 0 ? sc.Ed : sc.Ad(l - 1);
        j.ldx = 256; j.ldy = l == 0 ? 64 : 256; j.x1_rows = -1; j.y1_rows = -1; j.o0 = 0; j.i0 = 0;
        j.layer = l; j.row_off = 0; j.rows = rows[l]; j.cols = cols[l]; j.col_off = 0; j.i_limit = cols[l];
        j.scale = l == 4 ? 0.70710678118654752440f : 1.0f;
        j.bias = 1;
        j.x_frag = PIR_BWD_FRAG;                  // Zb / Q
        j.y_frag = l == 0 ? 0 : PIR_BWD_FRAG;     // A / Ad (layer 0: the row-major encoding arrays)
        continue;
      }
      const int ni = l == 0 ? 1 : 2;
      for (int ot = 0; ot < 2; ++ot)
        for (int itl = 0; itl < ni; ++itl) {
          WgJob& j = jobs.j[jobs.n++];
          j.X1 = sc.Zb(l); j.X2 = sc.Q(l);
          j.Y1 = l == 0 ? sc.E : sc.A(l - 1);
          j.Y2 = l == 0 ? sc.Ed : sc.Ad(l - 1);
          j.ldx = 256; j.ldy = l == 0 ? 64 : 256;
          j.x1_rows = -1; j.y1_rows = -1;
          j.o0 = 128 * ot; j.i0 = 128 * itl;
          j.layer = l; j.row_off = 0; j.rows = rows[l]; j.cols = cols[l]; j.col_off = 0; j.i_limit = cols[l];
          j.scale = l == 4 ? 0.70710678118654752440f : 1.0f;
          j.bias = itl == 0;
        }
    }
    if (g_feat) {
      if (first_big <= 8) {
        WgJob& j = big.j[big.n++];
        j.X1 = g_feat + (size_t)base * 256; j.X2 = nullptr; j.Y1 = sc.A(7); j.Y2 = nullptr;
        j.ldx = 256; j.ldy = 256; j.x1_rows = n; j.y1_rows = -1; j.o0 = 0; j.i0 = 0;
        j.layer = 8; j.row_off = 1; j.rows = 256; j.cols = 256; j.col_off = 0; j.i_limit = 256; j.scale = 1.0f; j.bias = 1;
        j.x_frag = 0;                             // the caller's g_feat, row-major
        j.y_frag = PIR_BWD_FRAG;
      } else {
      for (int ot = 0; ot < 2; ++ot)
        for (int itl = 0; itl < 2; ++itl) {
          WgJob& j = jobs.j[jobs.n++];
          j.X1 = g_feat + (size_t)base * 256; j.X2 = nullptr; j.Y1 = sc.A(7); j.Y2 = nullptr;
          j.ldx = 256; j.ldy = 256; j.x1_rows = n; j.y1_rows = -1; j.o0 = 128 * ot; j.i0 = 128 * itl;
          j.layer = 8; j.row_off = 1; j.rows = 256; j.cols = 256; j.col_off = 0; j.i_limit = 256; j.scale = 1.0f; j.bias = itl == 0;
        }
      }
    }
    if (big.n > 0) {
      rc = wgrad256_run(big, n_rows, n, fx, out, PIR_K_WGRAD, stream_);
      if (rc != PIR_OK) return rc;
    }
    hipLaunchKernelGGL(row0_kernel, dim3(kRow0Blocks), dim3(256), 0, stream, sc.A(7), sc.Ad(7),
                       g_sdf ? g_sdf + base : nullptr, n, n_rows, row0);
    if (jobs.n > 0) {
      rc = wgrad_run(jobs, n_rows, n, fx, row0, out, PIR_K_WGRAD, stream_);
    } else {      // row 0 of dW8 / db8[0] alone: the reduction kernel's extra block
      WgJobs none;
      none.n = 0;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(16), dim3(256), 0, stream, none, 1, fx.slabs, fx.bias_slabs, row0, out);
      rc = pir_check_launch();
    }
    if (rc != PIR_OK) return rc;
    rc = pir_check_launch();
    if (rc != PIR_OK) return rc;
  }
  return PIR_OK;
}

}  // extern "C"
