"""ctypes binding of libpir_hip.so (the C ABI declared in include/pir_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `python -m pir_amd.build`.  There is no CPU or
PyTorch fallback behind these calls: if the shared object is missing the import of any kernel raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIR_HIP_LIB", os.path.join(_HERE, "libpir_hip.so"))  # override: development A/B builds
_lib = None

PIR_OK = 0
_ERR = {1: "invalid argument", 2: "kernel launch failed", 3: "scratch buffer too small", 4: "size beyond the supported maximum"}


class PirHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PirHipError(
                f"{LIB_PATH} not found: build the HIP extension first (python -m pir_amd.build). "
                "pir_amd has no fallback path for its kernels.")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    vp, sz, lg, i32, f32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_long, ctypes.c_int, ctypes.c_float
    L.pir_sdf_packed_bytes.restype = sz
    L.pir_sdf_packed_bytes.argtypes = []
    L.pir_sdf_scratch_bytes_per_workgroup.restype = sz
    L.pir_sdf_scratch_bytes_per_workgroup.argtypes = []
    L.pir_sdf_pack.restype = i32
    L.pir_sdf_pack.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp]
    L.pir_sdf_forward.restype = i32
    L.pir_sdf_forward.argtypes = [vp, vp, lg, vp, vp, vp]
    L.pir_sdf_get_all.restype = i32
    L.pir_sdf_get_all.argtypes = [vp, vp, lg, vp, vp, vp, vp, sz, vp]
    L.pir_sdf_backward_scratch_bytes.restype = sz
    L.pir_sdf_backward_scratch_bytes.argtypes = [lg]
    L.pir_sdf_get_all_backward.restype = i32
    L.pir_sdf_get_all_backward.argtypes = [vp, vp, lg, vp, vp, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, sz, vp]
    L.pir_trace_workspace_bytes.restype = sz
    L.pir_trace_workspace_bytes.argtypes = [lg, lg, i32]
    L.pir_trace.restype = i32
    L.pir_trace.argtypes = [vp, lg, vp, vp, vp, vp, vp, f32, i32, i32, lg, i32, vp, vp, vp, vp, vp, sz, vp]
    L.pir_trace_static.restype = i32
    L.pir_trace_static.argtypes = [vp, lg, vp, vp, vp, vp, vp, f32, i32, i32, lg, i32, vp, vp, vp, vp, vp, vp]
    L.pir_trace_status.restype = i32
    L.pir_trace_status.argtypes = [vp, ctypes.POINTER(i32), vp]
    L.pir_camera_rays.restype = i32
    L.pir_camera_rays.argtypes = [lg, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]
    L.pir_ggx_forward.restype = i32
    L.pir_ggx_forward.argtypes = [lg, vp, i32, i32] + [vp] * 8 + [i32, vp, i32, vp, vp, vp, vp]
    L.pir_ggx_backward.restype = i32
    L.pir_ggx_backward.argtypes = [lg, vp, i32, i32] + [vp] * 8 + [i32, vp, i32] + [vp] * 10 + [vp]
    L.pir_visibility_points.restype = i32
    L.pir_visibility_points.argtypes = [vp, vp, vp, lg, i32, vp, vp]
    L.pir_visibility_points_backward.restype = i32
    L.pir_visibility_points_backward.argtypes = [vp, vp, vp, lg, i32, vp, vp, vp, vp]
    L.pir_visibility_forward.restype = i32
    L.pir_visibility_forward.argtypes = [vp, vp, vp, vp, vp, lg, i32, vp, vp]
    L.pir_visibility_backward.restype = i32
    L.pir_visibility_backward.argtypes = [vp, vp, vp, vp, vp, vp, lg, i32, vp, vp, vp, vp, vp]
    L.pir_shade_inputs_forward.restype = i32
    L.pir_shade_inputs_forward.argtypes = [lg, vp, vp, vp, i32, vp, vp, vp, vp]
    L.pir_shade_inputs_backward.restype = i32
    L.pir_shade_inputs_backward.argtypes = [lg, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    for name, n_ptr in (("pir_embed_forward", 2), ("pir_embed_backward", 3), ("pir_embed_tangent", 3), ("pir_embed_second", 4)):
        fn = getattr(L, name)
        fn.restype = i32
        fn.argtypes = [lg, i32, i32] + [vp] * n_ptr + [vp]
    L.pir_trace_sampler_index.restype = i32
    L.pir_trace_sampler_index.argtypes = [vp, lg, lg, i32, vp, vp]
    L.pir_ggx_table_indices.restype = i32
    L.pir_ggx_table_indices.argtypes = [lg, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.pir_edge_walk.restype = i32
    L.pir_edge_walk.argtypes = [vp, lg, vp, vp, i32, f32, f32, vp, vp, sz, vp]
    L.pir_mlp_product_floats.restype = sz
    L.pir_mlp_product_floats.argtypes = [i32]
    L.pir_mlp_pack_product.restype = i32
    L.pir_mlp_pack_product.argtypes = [vp, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp]
    L.pir_mlp_forward.restype = i32
    L.pir_mlp_forward.argtypes = [vp, sz, vp, vp, lg, vp, vp]
    L.pir_mlp_backward_scratch_bytes.restype = sz
    L.pir_mlp_backward_scratch_bytes.argtypes = [i32, lg]
    L.pir_mlp_backward.restype = i32
    L.pir_mlp_backward.argtypes = [vp, sz, vp, i32, vp, lg, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), vp, sz, vp]
    L.pir_mlp_backward_tangent.restype = i32
    L.pir_mlp_backward_tangent.argtypes = [vp, sz, vp, i32, vp, lg, vp, vp, vp, ctypes.POINTER(vp), vp, sz, vp]
    L.pir_profile_enable.restype = i32
    L.pir_profile_enable.argtypes = [i32]
    L.pir_profile_read.restype = i32
    L.pir_profile_read.argtypes = [i32, ctypes.POINTER(lg), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    for name, args in _OPTIONAL.items():
        if hasattr(L, name):
            fn = getattr(L, name)
            fn.restype = i32
            fn.argtypes = args


_OPTIONAL = {}


def check(code, what):
    if code != PIR_OK:
        raise PirHipError(f"{what}: {_ERR.get(code, 'error')} (code {code})")


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "pir_amd kernels take contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def stream_key(device):
    """(device index, current stream handle): scratch / workspace caches are keyed by it, so that work enqueued on
    different streams of one GPU never shares a scratch buffer (kernels on one stream are ordered, across streams
    they are not)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return idx, int(torch.cuda.current_stream(idx).cuda_stream)
