// Shared between sdf_backward.hip (the sweeps that fill the scratch) and wgrad.hip (the GEMM that consumes it).
#pragma once

namespace pir {

// HBM scratch of one chunk of points, every array [chunk_rows][256] fp32 in natural feature order
// (E / Ed: [chunk_rows][64], the 39 encoding columns in reference order).
struct BwdScratch {
  float* A_;      // a_l   = softplus(z_l), l = 0..7   (row 217.. of layer 3: the encoding, i.e. the lin4 input)
  float* Ad_;     // tangent of a_l along g_normal     (same convention for layer 3)
  float* Zb_;     // d L / d z_l
  float* Q_;      // d sdf / d z_l
  float* E;       // positional encoding
  float* Ed;      // its tangent
  float* pe_tmp;  // per-workgroup parking space of the sweep (2 waves x 32 registers x 64 lanes floats each):
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
