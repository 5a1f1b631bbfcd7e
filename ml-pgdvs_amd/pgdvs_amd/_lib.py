"""ctypes binding of libpgdvs_hip.so (C ABI declared in include/pgdvs_hip.h).

There is no CPU fallback: if the shared library is missing or an op is handed a
non-GPU tensor the call raises.  Build with ``python __graft_entry__.py`` (or
``make -C ml-pgdvs_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import pathlib

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = _PKG.parent / "lib" / "libpgdvs_hip.so"

CAM_BLOCK = 80

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> (restype, argtypes); mirrors include/pgdvs_hip.h one-to-one
SIGNATURES = {
    "pgdvs_last_error": (C.c_char_p, []),
    "pgdvs_abi_version": (_i, []),
    "pgdvs_build_arch": (C.c_char_p, []),
    "pgdvs_prof_enable": (None, [_i]),
    "pgdvs_prof_report": (_i, [C.c_char_p, _i]),
    "pgdvs_prof_overhead_ms": (C.c_double, []),
    "pgdvs_cam_prep": (_i, [_vp, _i, _vp, _vp]),
    "pgdvs_get_rays": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "pgdvs_dyn_warp": (_i, [_i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pgdvs_compact_workspace_bytes": (_i64, [_i64]),
    "pgdvs_compact_u8": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "pgdvs_gather_rows": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "pgdvs_knn_workspace_bytes": (_i64, [_i64]),
    "pgdvs_knn_mean_dist": (_i, [_vp, _vp, _i64, _i, _vp, _i, _vp, _i64, _vp]),
    "pgdvs_outlier_workspace_bytes": (_i64, [_i64]),
    "pgdvs_outlier_flags": (_i, [_vp, _vp, _i64, _f, _i, _vp, _vp, _vp, _i64, _vp]),
    "pgdvs_scatter_keep": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "pgdvs_project_flow_dense": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pgdvs_project_points": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "pgdvs_backwarp_l1": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pgdvs_softsplat_workspace_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "pgdvs_softsplat_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "pgdvs_softsplat_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pgdvs_dyn_splat_workspace_bytes": (_i64, [_i, _i]),
    "pgdvs_dyn_splat_composite": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "pgdvs_dyn_splat_composite_rng": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "pgdvs_splat_noise_field": (_i, [_i, _i, _vp, _vp, _vp]),
    "pgdvs_points_raster_workspace_bytes": (_i64, [_i64, _i, _i, _f]),
    "pgdvs_points_raster": (_i, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i64, _vp]),
    "pgdvs_points_raster_bounded": (_i, [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp,
                                         _vp, _i64, _vp]),
    "pgdvs_static_aggregate_workspace_bytes": (_i64, [_i, _i, _i, _i64]),
    "pgdvs_static_aggregate": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _vp, _i64, _vp]),
    "pgdvs_static_aggregate_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "pgdvs_gnt_gather": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _vp,
                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pgdvs_gnt_embed_weight_floats": (_i64, [_i]),
    "pgdvs_gnt_embed": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "pgdvs_gnt_posfc": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _i, _vp, _vp]),
    "pgdvs_gnt_head": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pgdvs_gnt_view_weight_floats": (_i64, []),
    "pgdvs_gnt_view_layer": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "pgdvs_gnt_ray_layer": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "pgdvs_mesh_render_workspace_bytes": (_i64, [_i, _i]),
    "pgdvs_mesh_render": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "pgdvs_track_points": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "pgdvs_knn_cross_workspace_bytes": (_i64, [_i64, _i64]),
    "pgdvs_knn_cross_mean_dist": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _i, _vp, _vp, _i64, _vp]),
    "pgdvs_threshold_flags": (_i, [_vp, _vp, _i64, _vp, _f, _vp, _vp, _vp, _vp]),
    "pgdvs_concat_rows": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "pgdvs_combine": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
}

_lib = None


class PgdvsHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libpgdvs_hip.so; raises (never falls back) when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise PgdvsHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` from the repo root "
            "(needs hipcc). There is no CPU fallback for the product path."
        )
    # PyTorch-ROCm ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be the ONE runtime of the
    # process: loaded first, the dynamic loader binds this library's libamdhip64 references to it.  Loaded
    # second (after /opt/rocm's copy came in through this library), torch's streams and allocations live in a
    # different runtime instance than our launches ("no ROCm-capable device is detected").
    import torch  # noqa: F401

    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().pgdvs_last_error().decode("utf-8", "replace")
        raise PgdvsHipError(f"{what} failed (rc={rc}): {msg}")
