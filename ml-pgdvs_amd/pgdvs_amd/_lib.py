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
    "pgdvs_option_set": (_i, [C.c_char_p, C.c_double]),
    "pgdvs_option_get": (C.c_double, [C.c_char_p]),
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



class ViewGeoDesc(C.Structure):
    """``pgdvs_view_geo_desc`` (include/pgdvs_hip.h), field for field."""
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32),
        ("flat_cam_tgt", _vp), ("flat_cam_src", _vp), ("time_src", _vp), ("time_tgt", _vp),
        ("rgb1", _vp), ("rgb2", _vp), ("depth1", _vp), ("depth2", _vp), ("dyn_mask1", _vp), ("flow12", _vp), ("flow_occ", _vp),
        ("use_flow_consistency", C.c_int32), ("remove_outlier", C.c_int32), ("outlier_knn", C.c_int32),
        ("outlier_std_thres", _f), ("alpha", _f),
        ("noise", _vp), ("rng_state", _vp),
        ("st_pcl_rgb", _vp), ("st_pcl_xyz", _vp), ("st_rows", _i64), ("st_count_dev", _vp),
        ("agg_S", C.c_int32), ("agg_rgbs", _vp), ("agg_depths", _vp), ("agg_masks", _vp), ("agg_K3s_host", _vp),
        ("agg_c2ws_host", _vp), ("agg_cloud_out", _vp), ("agg_xyz_out", _vp), ("agg_capacity", _i64), ("agg_count_out", _vp),
        ("row_bound", _i64), ("radius", _f), ("K", C.c_int32),
        ("static_rgb", _vp), ("static_mask", _vp), ("raster_status", _vp), ("render_dyn_rgb", _vp), ("render_dyn_mask", _vp),
        ("combined", _vp), ("combined_static", _vp), ("combined_dyn", _vp),
        ("side_stream", _vp), ("agg_params_cached", C.c_int32),
    ]


SIGNATURES.update({
    "pgdvs_eval_psnr_workspace_bytes": (_i64, []),
    "pgdvs_eval_psnr_sums": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "pgdvs_view_geo_desc_size": (_i64, []),
    "pgdvs_view_geo_workspace_bytes": (_i64, [C.POINTER(ViewGeoDesc)]),
    "pgdvs_view_geo_forward": (_i, [C.POINTER(ViewGeoDesc), _vp, _i64, _vp]),
    "pgdvs_view_geo_counters": (_i, [C.POINTER(ViewGeoDesc), _vp, _i64, _vp, _vp]),
    "pgdvs_view_geo_host_stats": (None, [C.POINTER(_i64), C.POINTER(C.c_double)]),
})

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
    if lib.pgdvs_view_geo_desc_size() != C.sizeof(ViewGeoDesc):
        raise PgdvsHipError(f"pgdvs_view_geo_desc: the library's struct has {lib.pgdvs_view_geo_desc_size()} bytes, this binding's "
                            f"{C.sizeof(ViewGeoDesc)} -- rebuild the extension (stale {LIB_PATH.name})")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().pgdvs_last_error().decode("utf-8", "replace")
        raise PgdvsHipError(f"{what} failed (rc={rc}): {msg}")
