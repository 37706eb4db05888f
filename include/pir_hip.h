/* C ABI of libpir_hip.so: the MI355X (gfx950) kernels behind PIR's per-ray surface-rendering hot path.
 *
 * The reference (ZqlwMatt/PIR) has no FFI layer: its boundary is the Python call signatures of
 * src/network.py and src/renderer/*.py.  These entry points are what pir_amd's Python mirrors of those
 * signatures bind through ctypes; each comment names the reference interface the entry point stands behind.
 *
 * Conventions: all pointers are DEVICE pointers to contiguous fp32 (masks: uint8) unless noted; `stream`
 * is a hipStream_t passed as void*; every call only enqueues work (no host sync, no allocation); the return
 * value is PIR_OK or a PIR_ERR_* code.  Sizes of 0 are accepted and do nothing.
 */
#ifndef PIR_HIP_H
#define PIR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PIR_OK = 0, PIR_ERR_ARG = 1, PIR_ERR_LAUNCH = 2, PIR_ERR_SCRATCH = 3, PIR_ERR_TOO_LARGE = 4 };

/* ---- SDF network (src/network.py:8-142, SDFNetwork with the configs/model_config.json architecture:
 *      d_in 3, multires 6, 8 x 256 softplus(beta=100), skip at 4, d_out 257) ------------------------- */

/* Bytes of the packed-weight buffer pir_sdf_pack fills. */
size_t pir_sdf_packed_bytes(void);
/* Bytes of scratch one resident workgroup of pir_sdf_get_all needs (activations a0..a7 of 64 points). */
size_t pir_sdf_scratch_bytes_per_workgroup(void);

/* Re-lay the effective (weight-normed) matrices W_l = g v/|v| (row-major [out_l, in_l]) and biases of
 * lin0..lin8 into MFMA fragment order.  w9/b9 are HOST arrays of 9 device pointers.
 * Replaces the per-call weight_norm recompute + nn.Linear of SDFNetwork.forward (src/network.py:87-93). */
int pir_sdf_pack(const float* const* w9, const float* const* b9, void* packed, void* stream);

/* sdf[M] (and feat[M,256] if non-null) for x[M,3].
 * SDFNetwork.forward / .sdf / .sdf_hidden_appearance (src/network.py:81-106). */
int pir_sdf_forward(const void* packed, const float* x, long M, float* sdf, float* feat, void* stream);

/* sdf[M], normal[M,3] = d sdf / d x, and feat[M,256] if non-null.
 * SDFNetwork.get_all(x, is_training=False) / .gradient (src/network.py:108-142).
 * scratch: at least one, ideally 2*CUs, multiples of pir_sdf_scratch_bytes_per_workgroup(). */
int pir_sdf_get_all(const void* packed, const float* x, long M, float* sdf, float* feat, float* normal,
                    void* scratch, size_t scratch_bytes, void* stream);

/* Scratch bytes for pir_sdf_get_all_backward when it processes chunk_points points per pass (fixed split-K
 * slabs + 33 KB per point of stored operands). */
size_t pir_sdf_backward_scratch_bytes(long chunk_points);

/* Backward of SDFNetwork.get_all(x, is_training=True) (src/network.py:125-142 under autograd, incl. the double
 * backward through the normal): given dL/dsdf [M], dL/dfeat [M,256], dL/dnormal [M,3] (each may be NULL = zero)
 * accumulates (+=) dL/dW_l into gw9[l] (row-major [out_l,in_l], effective weights) and dL/db_l into gb9[l], and
 * writes dL/dx into g_x [M,3] if non-NULL.  gw9/gb9 are HOST arrays of 9 device pointers, zero-initialised by the
 * caller.  Processes the points in chunks sized by scratch_bytes; deterministic (no float atomics). */
int pir_sdf_get_all_backward(const void* packed, const float* x, long M, const float* g_sdf, const float* g_feat,
                             const float* g_normal, float* g_x, float* const* gw9, float* const* gb9, void* scratch,
                             size_t scratch_bytes, void* stream);

/* ---- sphere tracer (src/renderer/raytracer.py:28-250, RayTracer) ----------------------------------- */

/* Workspace bytes for pir_trace on n_rays rays. chunk = the reference's max_num_rays split (bisection rounds
 * are shared by the rays of one chunk, raytracer.py:234); chunk <= 0 means one chunk. */
size_t pir_trace_workspace_bytes(long n_rays, long chunk, int max_bisect);

/* RayTracer.forward(sdf, ray_o, ray_d, min_dis, max_dis, work_mask) (raytracer.py:56-124) with sdf = the packed
 * network: sphere tracing (<= sphere_tracing_iters steps), dense sampler (n_steps samples) and bisection.
 * ray_o/ray_d [n,3], min_dis/max_dis [n], work_mask uint8[n] -> convergent uint8[n], points [n,3], sdf [n],
 * distance [n].  max_bisect bounds the bisection rounds that are enqueued (the reference loops until every
 * bracket is below 2*threshold; 8 rounds for unit-sphere rays); pir_trace_status reports if the bound was hit. */
int pir_trace(const void* packed, long n_rays, const float* ray_o, const float* ray_d, const float* min_dis,
              const float* max_dis, const unsigned char* work_mask, float sdf_threshold, int sphere_tracing_iters,
              int n_steps, long chunk, int max_bisect, unsigned char* convergent, float* points, float* sdf,
              float* distance, void* workspace, size_t workspace_bytes, void* stream);

/* Parity output: for every ray of the LAST pir_trace on this workspace (same n_rays / chunk / max_bisect), the index
 * of the first negative sample the dense sampler found = `min_idx` of RayTracer.ray_sampler (raytracer.py:192-197)
 * where its min_val < 0; -1 = sampled without a negative sample; -2 = the ray never reached the sampler.
 * min_idx: DEVICE int[n_rays], written in stream order. */
int pir_trace_sampler_index(const void* workspace, long n_rays, long chunk, int max_bisect, int* min_idx,
                            void* stream);

/* Synchronising read-back of the tracer status word (*host_status: 0 ok, 1 bisection bound hit). */
int pir_trace_status(const void* workspace, int* host_status, void* stream);

/* pir_trace for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same inputs, outputs and results as pir_trace,
 * but the tracer workspace is a 2.4 MB array inside the library's device image (one copy per GPU, held until the
 * process ends) and the call performs no allocation of any kind.  Limits: n_rays <= PIR_TRACE_STATIC_MAX_RAYS
 * (a 256 x 256 view: the largest training patch, render_surface.py:556-559) and at most 64 chunks, else
 * PIR_ERR_TOO_LARGE.  status: optional DEVICE int that receives the tracer status word (0 ok, 1 bisection bound
 * hit) in stream order, since there is no caller workspace for pir_trace_status to read.  All calls on one GPU share
 * that GPU's array: calls that overlap on one GPU (different streams or threads) race; calls on one stream are safe. */
#define PIR_TRACE_STATIC_MAX_RAYS 65536L
int pir_trace_static(const void* packed, long n_rays, const float* ray_o, const float* ray_d, const float* min_dis,
                     const float* max_dis, const unsigned char* work_mask, float sdf_threshold,
                     int sphere_tracing_iters, int n_steps, long chunk, int max_bisect, unsigned char* convergent,
                     float* points, float* sdf, float* distance, int* status, void* stream);

/* Camera.get_rays + intersect_sphere (raytracer.py:305-326, :253-273).  uv [n,2] or NULL for the pixel
 * centres of a width-wide image; cam21 = K_inv[:3,:3] (9) | C2W[:3,:3] (9) | C2W[:3,3] (3). */
int pir_camera_rays(long n, int width, const float* uv, const float* cam21, float radius, float* ray_o, float* ray_d,
                    float* ray_d_norm, unsigned char* hit, float* min_dis, float* max_dis, void* stream);

/* ---- light-visibility march (src/renderer/raytracer.py:669-729, light_visibility) -------------------- */
/* A march has n = n_samples - 10 samples per surface point (the reference drops the last 10 of
 * linspace(0, 1, n_samples), :682), n <= 512.  rays_d [bs,3] = normalize(points - light_pos), length [bs] =
 * |points - light_pos| (:676-678).  The SDF value and normal of the samples come from pir_sdf_get_all. */

/* Sample positions x [bs*n, 3] = light_pos + rays_d * mid, mid = z + dists/2, z = linspace * length (:682-692). */
int pir_visibility_points(const float* light_pos, const float* rays_d, const float* length, long bs, int n_samples,
                          float* x, void* stream);
/* Backward of the above: d x [bs*n,3] -> per-point partial d light_pos [bs,3] (sum over bs for the total),
 * d rays_d [bs,3], d length [bs]. */
int pir_visibility_points_backward(const float* g_x, const float* rays_d, const float* length, long bs, int n_samples,
                                   float* g_light_pos_point, float* g_rays_d, float* g_length, void* stream);
/* NeuS-style opacity compositing (:700-727): from sdf [bs*n] and normal [bs*n,3] of the samples,
 * alpha = clip((sigma((sdf - h) inv_s) - sigma((sdf + h) inv_s) + 1e-5) / (sigma((sdf - h) inv_s) + 1e-5), 0, 1) with
 * h = -relu(-rays_d.normal) dists / 2, vis [bs] = 1 - sum_s alpha_s prod_{j<s} (1 - alpha_j + 1e-7).
 * inv_s: device scalar. */
int pir_visibility_forward(const float* sdf, const float* normal, const float* rays_d, const float* length,
                           const float* inv_s, long bs, int n_samples, float* vis, void* stream);
/* Analytic backward of pir_visibility_forward: d vis [bs] -> d sdf [bs*n], d normal [bs*n,3], d rays_d [bs,3],
 * d length [bs] (what autograd derives for :700-727; samples whose sigmoids saturate get exact zeros). */
int pir_visibility_backward(const float* g_vis, const float* sdf, const float* normal, const float* rays_d,
                            const float* length, const float* inv_s, long bs, int n_samples, float* g_sdf,
                            float* g_normal, float* g_rays_d, float* g_length, void* stream);

/* ---- GGX point-light shader (src/renderer/renderer_ggx.py:39-127, GGXColocatedRenderer.forward) ------ */

/* Parity output: the table bins of GGXColocatedRenderer.forward (renderer_ggx.py:88-112) for every point --
 * idx_wi / idx_wo = clamp(ty * 100 + floor(cos^0.25 * 100)) into the n_rough-entry table for the light / view
 * cosine, idx_diff = clamp(ty) into the n_diff-entry table, ty = floor((alpha / 4)^0.25 * 50) -- computed by the same
 * device function as pir_ggx_forward / _backward.  Integer work: must equal the reference's indices exactly. */
int pir_ggx_table_indices(long P, const float* normal, const float* lightdir, const float* viewdir, const float* alpha,
                          int n_rough, int n_diff, int* idx_wi, int* idx_wo, int* idx_diff, void* stream);

/* light: device scalar, or [P,3] radiance when light_is_rgb; direct_radiance = EDirectRadiance. */
int pir_ggx_forward(long P, const float* light, int light_is_rgb, int direct_radiance, const float* distance,
                    const float* normal, const float* lightdir, const float* viewdir, const float* kd, const float* ks,
                    const float* alpha, const float* table_rough, int n_rough, const float* table_diff, int n_diff,
                    float* out_diffuse, float* out_specular, float* out_rgb, void* stream);

/* Analytic backward. grad_diffuse / grad_specular: total gradients reaching diffuse_rgb and specular_rgb
 * (the gradient of `rgb` is added to both by the caller).  g_light: per-point contributions, [P] (scalar
 * light; sum them) or [P,3]. */
int pir_ggx_backward(long P, const float* light, int light_is_rgb, int direct_radiance, const float* distance,
                     const float* normal, const float* lightdir, const float* viewdir, const float* kd, const float* ks,
                     const float* alpha, const float* table_rough, int n_rough, const float* table_diff, int n_diff,
                     const float* grad_diffuse, const float* grad_specular, float* g_light, float* g_distance,
                     float* g_normal, float* g_lightdir, float* g_viewdir, float* g_kd, float* g_ks, float* g_alpha,
                     void* stream);

/* ---- shading inputs (src/renderer/raytracer.py:18-25 reparam_points, :830-833 lightdir / distance;
 *      render_surface.py:121 normal normalisation, :134-135 collocated distance) ---------------------------- */

/* n_hat [P,3] = normal / (|normal| + 1e-10); distance [P] = |points - origin|; lightdir [P,3] (NULL to skip) =
 * (points - origin) / max(|points - origin|, 1e-10).  origin: light_pos (3 floats) or, with origin_per_point, ray_o
 * [P,3] (the collocated light).  The re-parameterised point of reparam_points equals `points` in value. */
int pir_shade_inputs_forward(long P, const float* points, const float* normal, const float* origin,
                             int origin_per_point, float* nhat, float* lightdir, float* distance, void* stream);
/* Backward of the above together with reparam_points: g_points [P,3] (NULL = zero) is the gradient reaching the
 * re-parameterised point from its other consumers (the material MLPs).  Writes g_normal [P,3] (from n_hat only: the
 * reference detaches the normal inside reparam_points), g_sdf [P] = -((g_points + g_p(lightdir, distance)) . dir) /
 * clamp(normal . dir, 1e-4) with dir = -ray_d (NULL: no re-parameterisation, is_training=False), and g_origin [P,3]
 * per-point contributions to d L / d light_pos (NULL to skip; sum them). */
int pir_shade_inputs_backward(long P, const float* points, const float* normal, const float* ray_d, const float* origin,
                              int origin_per_point, const float* g_nhat, const float* g_lightdir,
                              const float* g_distance, const float* g_points, float* g_normal, float* g_sdf,
                              float* g_origin, void* stream);

/* ---- IRON edge sampling: the surface walk of locate_edge_points (src/renderer/raytracer.py:497-560) ------------- */

/* Walks points[n,3] (in place) along the surface towards the silhouette seen from cam_o[3] (device pointer): per step
 * get_all at the point, stop when |n_hat . view_hat| <= dot_threshold (found[i] = 1), else move by
 * step_size * (normalize(n_hat - view_hat / dot) - sdf n_hat); at most max_step steps (points still walking after
 * the test of step max_step keep found[i] = 0, as in the reference loop :521-541).  One launch, no host sync.
 * scratch: as for pir_sdf_get_all. */
int pir_edge_walk(const void* packed, long n, float* points, const float* cam_o, int max_step, float step_size,
                  float dot_threshold, unsigned char* found, void* scratch, size_t scratch_bytes, void* stream);

/* ---- frequency encoding (src/models/embedder.py:20-41, get_embedder(multires, input_dims): every network input
 *      [x | sin(2^k x) | cos(2^k x)]_{k < L}, blocks of D columns; row width D (1 + 2 L)) -------------------------- */

/* out[P, D(1+2L)] = E(x) for x[P, D].  Embedder.embed / the embed_fn returned by get_embedder (embedder.py:38-41). */
int pir_embed_forward(long P, int D, int L, const float* x, float* out, void* stream);
/* g_x[P, D] = the vector-Jacobian product of E at x with g_out[P, D(1+2L)] (autograd's backward of the encoding). */
int pir_embed_backward(long P, int D, int L, const float* x, const float* g_out, float* g_x, void* stream);
/* The two halves of the backward OF that backward, taken when the material smoothness loss differentiates the
 * networks w.r.t. the surface point with create_graph=True (render_surface.py:40-48) and loss.backward() then
 * differentiates once more: out[P, D(1+2L)] = J_E(x) u (gradient w.r.t. g_out, for u[P, D]) ... */
int pir_embed_tangent(long P, int D, int L, const float* x, const float* u, float* out, void* stream);
/* ... and g_x[P, D] = u * d/dx <g_out, dE/dx> (gradient w.r.t. x). */
int pir_embed_second(long P, int D, int L, const float* x, const float* g_out, const float* u, float* g_x,
                     void* stream);

/* ---- material / feature-field MLPs (src/network.py:146-267 RenderingNetwork, :382-467 DINONetwork, :295-367
 *      BlendNetwork): ReLU MLPs with 256-wide hidden layers, the input optionally concatenated again at one skip
 *      layer ([h | x] / sqrt 2, :237-238), linear output layer ------------------------------------------------ */

/* Floats one packed product of n_kblocks k-blocks (8 input columns each) x 256 outputs occupies. */
size_t pir_mlp_product_floats(int n_kblocks);

/* Re-lay the block W[row0 : row0+nrows, col0 : col0+ncols] (row-major, leading dimension ld; nrows <= 256,
 * ncols <= 8 n_kblocks, the rest is zero-filled) times `scale` into MFMA fragment order at dst.  transposed != 0
 * packs the transpose of the block W[row0 : row0+ncols, col0 : col0+nrows] instead (the backward's W^T products).
 * Replaces the per-call weight_norm + nn.Linear of the reference modules' forward (src/network.py:234-243). */
int pir_mlp_pack_product(const float* W, int ld, int row0, int nrows, int col0, int ncols, int transposed, float scale,
                         float* dst, int n_kblocks, void* stream);

/* Layer plan of one packed network.  The packed buffer is [mats_floats floats of products | vectors]; prod_*
 * are offsets in units of 4 floats into the products, *_off float offsets into the vectors. */
typedef struct PirMlpPlan {
  int kin_kb;        /* k-blocks of the zero-padded input: rows of X are 8 * kin_kb floats */
  int n_hidden;      /* 256-wide ReLU layers, 1..8 */
  int skip;          /* hidden layer (1..n_hidden-1) whose input is [h | x] / sqrt 2 (scale folded into the pack), or -1 */
  int n_out;         /* outputs of the linear last layer (<= 4, or a multiple of 4) */
  int prod_in;       /* lin0 over x: kin_kb k-blocks */
  int prod_h[8];     /* hidden layer l over the previous activations: 32 k-blocks, l = 1..n_hidden-1 */
  int prod_skip_x;   /* the skip layer over x: kin_kb k-blocks */
  int prod_out;      /* output layer, ceil(n_out / 256) passes of 32 k-blocks (n_out > 4 only) */
  int bias_off;      /* hidden biases [n_hidden][256] */
  int head_off;      /* output-layer rows [n_out][256] (n_out <= 4 only) */
  int out_bias_off;  /* output bias, padded to 4 floats or to 256 per pass */
  /* transposed products for pir_mlp_backward (has_bwd = 0: packed for inference only) */
  int has_bwd;
  int rprod_h[8];    /* W_l^T, l = 1..n_hidden-1: 32 k-blocks each */
  int rprod_in;      /* W_0^T: ceil(kin_kb / 32) passes of 256 input columns x 32 k-blocks */
  int rprod_skip_x;  /* the x part of the skip layer, transposed: same shape */
  int rprod_out;     /* W_out^T over the outputs: n_out / 8 k-blocks (n_out > 4, a multiple of 8, only) */
} PirMlpPlan;

/* out [P, n_out] = last layer(relu(...relu(lin0 x)...)) for x [P, 8 kin_kb] (zero-padded input rows).
 * RenderingNetwork.forward / DINONetwork.forward / BlendNetwork.forward up to, not including, the output
 * transform (output_scale, output_bias, sigmoid; src/network.py:245-247), which stays a [P, n_out] PyTorch op. */
int pir_mlp_forward(const void* packed, size_t mats_floats, const PirMlpPlan* plan, const float* X, long P, float* out,
                    void* stream);

/* Scratch bytes for pir_mlp_backward when it processes chunk_points points per pass (split-K slabs + the stored
 * activations and pre-activation gradients, 2 KB per point and hidden layer). */
size_t pir_mlp_backward_scratch_bytes(int n_hidden, long chunk_points);

/* Backward of pir_mlp_forward (what autograd derives for src/network.py:232-243; n_out <= 4 or a multiple of 8):
 * given g_out [P, n_out]
 * accumulates (+=) dL/dW_l into gw[l] (row-major, the true shapes [256, k0], [256, 256], [256, 256 + k0] for the
 * skip layer, [n_out, 256]) and dL/db_l into gb[l], l = 0..n_hidden, and writes dL/dx into g_x [P, 8 kin_kb] if
 * non-NULL.  gw / gb: HOST arrays of n_hidden + 1 device pointers, zero-initialised by the caller.  k0 = true input
 * width.  Recomputes the activations (nothing is kept from the forward); deterministic (no float atomics). */
int pir_mlp_backward(const void* packed, size_t mats_floats, const PirMlpPlan* plan, int k0, const float* X, long P,
                     const float* g_out, float* g_x, float* const* gw, float* const* gb, void* scratch,
                     size_t scratch_bytes, void* stream);

/* Second-order backward: what autograd derives when g_x of pir_mlp_backward is itself differentiated (the
 * smoothness loss on d material / d points, render_surface.py:40-48,76-78, :669-672).  Given the upstream gradient
 * gbar_x [P, 8 kin_kb] of g_x: accumulates (+=) d<gbar_x, g_x>/dW_l into gw[l] (l = 0..n_hidden, shapes as above) and
 * writes d<gbar_x, g_x>/dg_out into ghat_out [P, n_out].  Nothing reaches x or the biases (the ReLU masks are
 * locally constant).  Needs 3 KB of scratch per point and hidden layer. */
int pir_mlp_backward_tangent(const void* packed, size_t mats_floats, const PirMlpPlan* plan, int k0, const float* X,
                             long P, const float* g_out, const float* gbar_x, float* ghat_out, float* const* gw,
                             void* scratch, size_t scratch_bytes, void* stream);

/* ---- measurement aid (no counterpart in the reference) ------------------------------------------------ */

/* Kernel families timed by the optional HIP-event profiler. */
enum { PIR_K_SDF_EVAL = 0, PIR_K_SDF_GETALL = 1, PIR_K_TRACE_STEP = 2, PIR_K_SAMPLER = 3, PIR_K_BISECT = 4,
       PIR_K_BWD_SWEEP = 5, PIR_K_WGRAD = 6, PIR_K_GGX_FWD = 7, PIR_K_GGX_BWD = 8,
       PIR_K_SDF_EVAL_FEAT = 9, PIR_K_SDF_GETALL_FEAT = 10,   /* the variants that also write the 256 features */
       PIR_K_MLP_FWD = 11, PIR_K_MLP_BWD = 12,                /* material / feature-field MLP kernels */
       PIR_K_COUNT = 13 };
/* When enabled every kernel launch of the families above is bracketed by hipEvents on its stream. */
int pir_profile_enable(int on);
/* Synchronises and returns, per family id < n_ids, the number of launches, their summed duration (ms) and the
 * summed units they processed: points, or -- for the three tracer families, whose list sizes only exist on the
 * device -- the SDF evaluations their kernels executed on the CURRENT device (counted by the kernels while the
 * profiler is enabled, one atomic per tile). */
int pir_profile_read(int n_ids, long* launches, double* total_ms, double* units);

#ifdef __cplusplus
}
#endif
#endif
