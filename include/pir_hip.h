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

enum { PIR_OK = 0, PIR_ERR_ARG = 1, PIR_ERR_LAUNCH = 2, PIR_ERR_SCRATCH = 3 };

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

#ifdef __cplusplus
}
#endif
#endif
