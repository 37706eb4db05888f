// Shared between sdf_backward.hip (the sweeps that fill the scratch) and wgrad.hip (the GEMM that consumes it).
#pragma once

namespace pir {

// Layout of the four [chunk_rows][256] operand families in HBM scratch.  1 (default): MFMA FRAGMENT order -- per
// 32-point tile an image [k-block 0..31][lane 0..63] of float4, lane (h, j) holding features 8 kb + 4 h + 0..3 of point
// j, i.e. element (row, f) at ((row / 32 * 32 + f / 8) * 64 + (f % 8) / 4 * 32 + row % 32) * 4 + f % 4.  The sweep then
// stores / re-loads one contiguous 1 KB per wave instruction instead of 32 rows x 32 B, and the weight-gradient GEMM
// un-permutes on its way into LDS (it stages through registers anyway).  0: natural row-major [row][256] (round 1/2;
// required by the 128 x 128-tile A/B build of the GEMM).
#ifndef PIR_BWD_FRAG
#ifdef PIR_WG_TILE128
#define PIR_BWD_FRAG 0
#else
#define PIR_BWD_FRAG 1
#endif
#endif

// Cache policy of the sweep's scratch traffic (A/B builds pass -D...=0 / =1).  The 2 x 2.1 MB of packed weights every
// tile re-reads share the 4 MB L2 of an XCD with 55 KB per point of scratch traffic streaming through it:
//   PIR_BWD_NT_QZ (on)   Q / Zb (written here, read by a later launch) are stored non-temporally
//   PIR_BWD_NT_LD (on)   the A / Ad fragments the GZ epilogues read back (once) are loaded non-temporally
//   PIR_BWD_TOUCH (off)  one dword per 128-byte line of the NEXT layer's A / Ad fragments loaded before each GZ product
// Measured on 2 M points (profiles/r03_bwd_cache_policy_ab.log): 68.8 -> 67.0 ms for the first two.  The touch-ahead
// is worth 1.1 % on its own but nothing on top of them (65.25 vs 65.05 ms) and re-fetches half of the lines it touches
// (PMC: +9 KB / point of HBM reads); storing A / Ad non-temporally as well costs 1.3 % (they are re-read within the
// tile); neither policy helps get_all's spill ring.
#ifndef PIR_BWD_NT_QZ
#define PIR_BWD_NT_QZ 1
#endif
#ifndef PIR_BWD_NT_LD
#define PIR_BWD_NT_LD 1
#endif
#ifndef PIR_BWD_TOUCH
#define PIR_BWD_TOUCH 0
#endif

// HBM scratch of one chunk of points, every array [chunk_rows][256] fp32 (layout above)
// (E / Ed: [chunk_rows][64], the 39 encoding columns in reference order, always row-major).
struct BwdScratch {
  float* A_;      // a_l   = softplus(z_l), l = 0..7   (row 217.. of layer 3: the encoding, i.e. the lin4 input)
  float* Ad_;     // tangent of a_l along g_normal     (same convention for layer 3)
  float* Zb_;     // d L / d z_l
  float* Q_;      // d sdf / d z_l
  float* E;       // positional encoding
  float* Ed;      // its tangent
  float* pe_tmp;  // per-workgroup parking space of the sweep (4 waves x 32 registers x 64 lanes floats each):
                  // the lin4 share of the encoding gradient waits here while layers 3..0 run
  size_t layer_stride;   // floats between consecutive layers of one family (= chunk rows * 256)
  __host__ __device__ float* A(int l) const { return A_ + l * layer_stride; }
  __host__ __device__ float* Ad(int l) const { return Ad_ + l * layer_stride; }
  __host__ __device__ float* Zb(int l) const { return Zb_ + l * layer_stride; }
  __host__ __device__ float* Q(int l) const { return Q_ + l * layer_stride; }
};

constexpr int kBwdRowBytes = 32 * 1024 + 2 * 256;

}  // namespace pir

extern "C" int pir_internal_bwd_sweep(const void* packed, const float* x, const float* g_sdf, const float* g_feat,
                                      const float* g_nrm, long chunk_base, int chunk_n, const pir::BwdScratch* sc,
                                      float* g_x, void* stream);
