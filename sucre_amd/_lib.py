"""ctypes binding of libsucre_hip.so (include/sucre_hip.h).

The HIP library is the product: if it is missing or does not export the ABI this module raises -- there is no
CPU or PyTorch fallback for the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

ABI_VERSION = 1
# SUCRE_HIP_LIB selects another build of the same ABI (experiment builds, tools/exp/build_variants.sh); default = the product
LIB_PATH = Path(os.environ.get('SUCRE_HIP_LIB', Path(__file__).resolve().parent / 'libsucre_hip.so'))

FIT_CLOSED_FORM = 1
FIT_OBS_U16MM = 2
FIT_EXT_COLOUR = 4
FIT_EXT_BOTH = 16
FIT_KEEP_J = 8
OBS_F32, OBS_U16MM, OBS_F32_PLAIN, OBS_F32_Z26 = 0, 1, 2, 3
EXT_POINTS, EXT_COLOUR, EXT_POINTS_COLOUR = 1, 2, 3
OBS_FORMATS = {'f32': OBS_F32, 'u16mm': OBS_U16MM, 'f32plain': OBS_F32_PLAIN, 'f32z26': OBS_F32_Z26}
WS_VIEW_COUNT, WS_VIEW_KEEP, WS_N_OBS, WS_PARAMS, WS_SUMS, WS_N_OBS_TOTAL, WS_STORE_FORMAT = range(7)
STORE_F32, STORE_U16MM, STORE_Z24, STORE_Z26 = 0, 1, 2, 3   # what the compaction chose (SUCRE_WS_STORE_FORMAT)


class SucreView(C.Structure):
    """sucre_view_t"""
    _fields_ = [('depth', C.c_void_p), ('rgb', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32),
                ('K', C.c_float * 9), ('Kinv', C.c_float * 9), ('R', C.c_float * 9), ('t', C.c_float * 3),
                ('Rinv', C.c_float * 9), ('tinv', C.c_float * 3)]


assert C.sizeof(SucreView) == 192


class GroupImage(C.Structure):
    """sucre_group_image_t"""
    _fields_ = [('ws', C.c_void_p), ('H', C.c_int32), ('W', C.c_int32), ('n_views', C.c_int32), ('reserved', C.c_int32)]

# name -> (restype, argtypes); mirrors include/sucre_hip.h one to one (tests/test_abi.py checks both ways)
_i, _vp, _d, _u64 = C.c_int, C.c_void_p, C.c_double, C.c_uint64
SIGNATURES = {
    'sucre_version': (_i, []),
    'sucre_last_error': (C.c_char_p, []),
    'sucre_workspace_bytes': (C.c_size_t, [_i, _i, _i]),
    'sucre_ws_offset': (C.c_int64, [_i, _i, _i, _i]),
    'sucre_match_views': (_i, [_vp, _i, _i, _i, C.POINTER(SucreView), _vp, _i, _i, _vp]),
    'sucre_match_map': (_i, [_i, _i, _i, C.POINTER(SucreView), _vp, _i, _vp, _vp]),
    'sucre_pack_view': (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    'sucre_pack_views': (_i, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _vp]),
    'sucre_project_points': (_i, [C.POINTER(SucreView), _vp, C.c_int64, _vp, _vp]),
    'sucre_import_view': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    'sucre_finalize_matches': (_i, [_vp, _i, _i, _i, _d, _vp]),
    'sucre_finalize_matches_fmt': (_i, [_vp, _i, _i, _i, _d, _i, _vp]),
    'sucre_fit_init': (_i, [_vp, _i, _i, _i, _vp, _vp, C.POINTER(C.c_float), _vp, _vp]),
    'sucre_fit_run': (_i, [_vp, _i, _i, _i, _i, _i, _d, _d, _d, _d, C.c_uint, _vp, _vp]),
    'sucre_fit_grad': (_i, [_vp, _i, _i, _i, _i, _d, _d, _d, _d, C.c_uint, _vp]),
    'sucre_fit_step': (_i, [_vp, _i, _i, _i, _i, _d, _d, _d, _d, _vp, _vp]),
    'sucre_set_n_obs_total': (_i, [_vp, _i, _i, _i, _u64, _vp]),
    'sucre_group_bytes': (C.c_size_t, [_i]),
    'sucre_group_sums_offset': (C.c_int64, []),
    'sucre_group_init': (_i, [_vp, _i, C.POINTER(GroupImage), C.POINTER(C.c_float), _vp]),
    'sucre_group_iter': (_i, [_vp, _i, _i, _d, _d, _d, _d, C.c_uint, _u64, _vp, _vp]),
    'sucre_group_finish': (_i, [_vp, _i, _i, _d, _d, _d, _d, _u64, _vp, _vp]),
    'sucre_batch_bytes': (C.c_size_t, [_i]),
    'sucre_fit_run_batch': (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), _i, _i, C.POINTER(_i), _i, _i, _d, _d, _d, _d, C.c_uint, _vp]),
    'sucre_update_J': (_i, [_vp, _i, _i, _i, _vp]),
    'sucre_update_J_fmt': (_i, [_vp, _i, _i, _i, _i, _vp]),
    'sucre_export_J': (_i, [_vp, _i, _i, _i, _vp, _vp]),
    'sucre_select_scratch_bytes': (C.c_size_t, []),
    'sucre_select_ranks': (_i, [_vp, _i, _i, _i, C.POINTER(C.c_uint64), _vp, _vp, _vp]),
    'sucre_count_valid': (_i, [_vp, _i, _i, _vp, _vp]),
    'sucre_plot_stretch': (_i, [_vp, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp]),
    'sucre_check_store': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    'sucre_export_view': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'sucre_light_workspace_bytes': (C.c_size_t, [_i, _i, _i]),
    'sucre_light_params_offset': (C.c_int64, [_i, _i, _i]),
    'sucre_match_views_light': (_i, [_vp, _vp, _i, _i, _i, C.POINTER(SucreView), _vp, _i, _i, _vp]),
    'sucre_match_views_fcolour': (_i, [_vp, _vp, _i, _i, _i, C.POINTER(SucreView), _vp, _i, _i, _vp]),
    'sucre_light_workspace_bytes_ext': (C.c_size_t, [_i, _i, _i, _i]),
    'sucre_match_views_light_fcolour': (_i, [_vp, _vp, _i, _i, _i, C.POINTER(SucreView), _vp, _i, _i, _vp]),
    'sucre_finalize_matches_ext': (_i, [_vp, _vp, _i, _i, _i, _d, _i, _vp]),
    'sucre_import_view_ext': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _vp]),
    'sucre_export_view_ext': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    'sucre_finalize_matches_light': (_i, [_vp, _vp, _i, _i, _i, _d, _vp]),
    'sucre_fit_init_light': (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, C.POINTER(C.c_float), _vp, _vp]),
    'sucre_update_J_light': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'sucre_update_J_ext': (_i, [_vp, _vp, _i, _i, _i, C.c_uint, _vp]),
    'sucre_fit_run_light': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _d, _d, _d, _d, C.c_uint, _vp, _vp]),
}

_lib = None


class SucreError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Loads the HIP library; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SucreError(f'{LIB_PATH} is missing: build the gfx950 HIP library first '
                         f'(make -C {LIB_PATH.parent / "csrc"}); there is no fallback path')
    # PyTorch-ROCm ships its own HIP runtime; it must be the process's one BEFORE this library is loaded (the library then
    # binds to the copy already mapped).  Loaded first, the library pulls in /opt/rocm's copy and torch its own afterwards:
    # two runtimes in one process, and every launch on a torch stream fails with "no ROCm-capable device is detected"
    # (seen with build() and smoke() in one interpreter).
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.sucre_version() != ABI_VERSION:
        raise SucreError(f'libsucre_hip.so has ABI {lib.sucre_version()}, expected {ABI_VERSION}: rebuild it')
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise SucreError(f'libsucre_hip error {rc}: {load().sucre_last_error().decode()}')
