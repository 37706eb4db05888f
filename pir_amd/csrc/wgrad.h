// Job list of the split-K weight-gradient GEMM (wgrad.hip), shared with the material-MLP backward (mlp.hip).
#pragma once
#include <stddef.h>

namespace pir {

constexpr int kMaxJobs = 40;
constexpr int kKSplitMax = 64;

struct WgJob {
  const float* X1; const float* Y1;   // product 1 (Zb, H): C += X1^T Y1 over the rows (points) of a chunk
  const float* X2; const float* Y2;   // product 2 (Q, Pb) or null
  int ldx;                            // leading dimension of X (256; n_out for a caller-owned wide g_out)
  int ldy;                            // leading dimension of Y (256; 64 for the SDF encoding arrays; the padded
                                      // input width of a material network)
  int x1_rows;                        // rows of X1 that exist (caller-owned g_feat is not padded); <0: all
  int y1_rows;                        // same for Y1 (the caller-owned input rows of a material network)
  int o0, i0;                         // C tile origin: rows of X (outputs), columns of Y (inputs)
  int layer, row_off, rows, cols;     // destination: gW[layer][(row_off + o) * cols + col_off + i], o < rows
  int col_off, i_limit;               //   ... for the Y columns i < i_limit
  float scale;
  int bias;                           // 1: this job also produces column sums of X1 (i0 == 0 jobs)
  int x_frag = 0, y_frag = 0;         // 256 x 256-tile kernel only: X1/X2 resp. Y1/Y2 are in fragment order
                                      // (sdf_backward.h) instead of row-major; such arrays are 256 wide and padded
};
struct WgJobs { WgJob j[kMaxJobs]; int n; };
struct GradOut { float* w[9]; float* b[9]; };

// fixed part of a backward scratch buffer: split-K slabs, bias slabs, row-0 partials (each 256-byte aligned)
struct WgFixed { float* slabs; float* bias_slabs; float* row0; char* rest; };
size_t wgrad_fixed_bytes();
WgFixed wgrad_carve(void* scratch);

// Enqueue the split-K GEMM of every job over n_rows (multiple of 32) rows and the deterministic slab reduction
// into `out`.  row0_partial != null adds the SDF's row-0 block to the reduction.
int wgrad_run(const WgJobs& jobs, int n_rows, int n_points, const WgFixed& fx, const float* row0_partial,
              const GradOut& out, int prof_id, void* stream);

}  // namespace pir
