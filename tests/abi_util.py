"""Error conventions of the C ABI: a size of 0 is accepted and does nothing (PIR_OK without touching a
pointer), a negative size or a missing required pointer is PIR_ERR_ARG -- returned BEFORE anything is launched, so a
caller's mistake is a status code, never a GPU fault.  None of the calls below reaches a kernel: the same list runs on
the GPU box (tests/test_gpu_abi.py) and, with a host address standing in for the device buffers, without a GPU."""


def check_error_conventions(L, p, s):
    """L: the loaded library; p: address of a 16 KB buffer nobody dereferences here; s: a stream handle (or None)."""
    OK, ARG = 0, 1
    N = None
    # (function, args with size 0 -> OK, args with a negative size -> ARG, args with a null required pointer -> ARG)
    cases = [
        (L.pir_sdf_forward, (p, N, 0, N, N, s), (p, p, -1, p, N, s), (p, N, 8, p, N, s)),
        (L.pir_sdf_get_all, (p, N, 0, N, N, N, N, 0, s), (p, p, -1, p, N, p, p, 4096, s), (p, p, 8, p, N, N, p, 4096, s)),
        (L.pir_camera_rays, (0, 4, N, N, 1.0, N, N, N, N, N, N, s), (-1, 4, N, p, 1.0, p, p, p, p, p, p, s),
         (8, 4, N, N, 1.0, p, p, p, p, p, p, s)),
        (L.pir_ggx_forward, (0, N, 0, 1) + (N,) * 8 + (5000, N, 50, N, N, N, s), (-1, p, 0, 1) + (p,) * 8 + (5000, p, 50, p, p, p, s),
         (8, p, 0, 1) + (p,) * 7 + (N,) + (5000, p, 50, p, p, p, s)),
        (L.pir_ggx_table_indices, (0, N, N, N, N, 5000, 50, N, N, N, s), (-1, p, p, p, p, 5000, 50, p, p, p, s),
         (8, p, p, p, p, 5000, 50, p, N, p, s)),
        (L.pir_visibility_points, (N, N, N, 0, 128, N, s), (p, p, p, -1, 128, p, s), (p, p, p, 8, 128, N, s)),
        (L.pir_visibility_forward, (N, N, N, N, N, 0, 128, N, s), (p, p, p, p, p, -1, 128, p, s), (p, p, p, p, p, 8, 128, N, s)),
        (L.pir_visibility_backward, (N,) * 6 + (0, 128, N, N, N, N, s), (p,) * 6 + (-1, 128, p, p, p, p, s),
         (p,) * 6 + (8, 128, p, N, p, p, s)),
        (L.pir_shade_inputs_forward, (0, N, N, N, 0, N, N, N, s), (-1, p, p, p, 0, p, p, p, s), (8, p, N, p, 0, p, p, p, s)),
        (L.pir_embed_forward, (0, 3, 10, N, N, s), (-1, 3, 10, p, p, s), (8, 3, 10, p, N, s)),
        (L.pir_embed_backward, (0, 3, 10, N, N, N, s), (-1, 3, 10, p, p, p, s), (8, 3, 10, p, N, p, s)),
        (L.pir_edge_walk, (p, 0, N, N, 16, 1e-3, 0.1, N, N, 0, s), (p, -1, p, p, 16, 1e-3, 0.1, p, p, 4096, s),
         (p, 8, p, N, 16, 1e-3, 0.1, p, p, 4096, s)),
        (L.pir_trace_sampler_index, (N, 0, 0, 18, N, s), (p, -1, 0, 18, p, s), (p, 8, 0, 18, N, s)),
    ]
    for fn, zero, negative, null in cases:
        assert fn(*zero) == OK, (fn.__name__, "size 0")
        assert fn(*negative) == ARG, (fn.__name__, "negative size")
        assert fn(*null) == ARG, (fn.__name__, "null pointer")
    # out-of-range parameters that are not sizes
    assert L.pir_embed_forward(8, 3, 17, p, p, s) == ARG and L.pir_embed_forward(8, 0, 4, p, p, s) == ARG
    assert L.pir_visibility_points(p, p, p, 8, 11, p, s) == ARG          # n_samples - 10 < 2
    assert L.pir_edge_walk(p, 8, p, p, 5000, 1e-3, 0.1, p, p, 4096, s) == ARG     # max_step > 4096
    assert L.pir_sdf_get_all(p, p, 8, p, N, p, p, 16, s) == 3            # scratch smaller than one workgroup's: PIR_ERR_SCRATCH
