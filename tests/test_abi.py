"""The C-ABI library loads on a machine without a GPU and exports exactly what include/sucre_hip.h declares;
the ctypes binding mirrors the header one to one; host-side argument validation works without launching."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

from sucre_amd import _lib

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / 'include' / 'sucre_hip.h').read_text()


def declared_functions():
    body = re.sub(r'/\*.*?\*/', '', HEADER, flags=re.S)
    return sorted(set(re.findall(r'\b(sucre_[a-z_A-Z0-9]+)\s*\(', body)))


def test_header_declares_what_binding_binds():
    assert declared_functions() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    assert _lib.LIB_PATH.exists(), 'build first: python -c "import __graft_entry__ as g; g.build()"'
    out = subprocess.run(['nm', '-D', '--defined-only', str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if ' T ' in line}
    for name in declared_functions():
        assert name in exported, name
    lib = _lib.load()
    assert lib.sucre_version() == _lib.ABI_VERSION
    m = re.search(r'#define SUCRE_ABI_VERSION (\d+)', HEADER)
    assert int(m.group(1)) == _lib.ABI_VERSION


def test_view_struct_matches_header():
    assert C.sizeof(_lib.SucreView) == 192  # 2 pointers + 2 int32 + 42 float32
    assert _lib.SucreView.K.offset == 24 and _lib.SucreView.tinv.offset == 24 + 4 * 39


def test_geometry_queries_and_errors():
    lib = _lib.load()
    n = lib.sucre_workspace_bytes(1080, 1920, 65)
    assert n > 8160 * 65 * 1792 and n % 256 == 0
    assert lib.sucre_workspace_bytes(1080, 1920, 66) > n
    assert lib.sucre_workspace_bytes(0, 1920, 65) == 0 and b'invalid geometry' in lib.sucre_last_error()
    assert lib.sucre_workspace_bytes(1080, 1920, 0) == 0
    assert lib.sucre_workspace_bytes(40000, 1920, 1) == 0  # pixel indices must fit int16 (loader.py:71-74)
    offs = [lib.sucre_ws_offset(1080, 1920, 65, r) for r in range(6)]
    assert all(o > 0 and o % 256 == 0 for o in offs) and len(set(offs)) == 6
    assert lib.sucre_ws_offset(1080, 1920, 65, 99) < 0 and b'unknown workspace region' in lib.sucre_last_error()


def test_argument_validation_happens_before_any_launch():
    """NULL / misaligned / out-of-range arguments are rejected on the host (no GPU needed to see that)."""
    lib = _lib.load()
    assert lib.sucre_finalize_matches(None, 48, 64, 3, 1e-6, None) == -1 and b'NULL' in lib.sucre_last_error()
    assert lib.sucre_finalize_matches(C.c_void_p(4), 48, 64, 3, 1e-6, None) == -1 and b'aligned' in lib.sucre_last_error()
    ws = C.c_void_p(256)
    assert lib.sucre_finalize_matches(ws, 48, 64, 3, float('nan'), None) == -1
    assert lib.sucre_fit_run(ws, 48, 64, 3, 0, -1, 0.05, 0.9, 0.999, 1e-8, 0, None, None) == -2
    assert lib.sucre_fit_run(ws, 48, 64, 3, 0, 1, 0.05, 1.5, 0.999, 1e-8, 0, None, None) == -1
    assert lib.sucre_fit_run(ws, 48, 64, 3, 0, 1, 0.05, 0.9, 0.999, 1e-8, 64, None, None) == -1 and b'flags' in lib.sucre_last_error()
    assert lib.sucre_fit_grad(ws, 48, 64, 3, 0, 0.05, 0.9, 0.999, 1e-8, 0, None) == -2
    assert lib.sucre_export_view(ws, 48, 64, 3, 3, ws, None, None) == -2
    assert lib.sucre_export_J(ws, 48, 64, 3, None, None) == -1
    assert lib.sucre_set_n_obs_total(ws, 48, 64, 3, 0, None) == -2
    tgt = _lib.SucreView()
    assert lib.sucre_match_views(ws, 48, 64, 3, C.byref(tgt), ws, 0, 3, None) == -1  # NULL depth
    tgt.depth, tgt.H, tgt.W = 256, 48, 64
    assert lib.sucre_match_views(ws, 48, 64, 3, C.byref(tgt), ws, 2, 2, None) == -2
    assert lib.sucre_match_views(ws, 48, 64, 3, C.byref(tgt), ws, 0, 4, None) == -2
    tgt.H = 47
    assert lib.sucre_match_views(ws, 48, 64, 3, C.byref(tgt), ws, 0, 3, None) == -1 and b'laid out' in lib.sucre_last_error()
    # round 3 entry points
    cam = _lib.SucreView()
    assert lib.sucre_project_points(None, ws, 4, ws, None) == -1 and b'view is NULL' in lib.sucre_last_error()
    assert lib.sucre_project_points(C.byref(cam), ws, 4, ws, None) == -1 and b'sensor size' in lib.sucre_last_error()
    cam.H, cam.W = 48, 64
    assert lib.sucre_project_points(C.byref(cam), ws, -1, ws, None) == -2
    assert lib.sucre_project_points(C.byref(cam), None, 4, ws, None) == -1
    assert lib.sucre_project_points(C.byref(cam), None, 0, None, None) == 0          # nothing to do, nothing launched
    assert lib.sucre_pack_view(ws, ws, 0, 64, ws, None) == -1 and b'image size' in lib.sucre_last_error()
    assert lib.sucre_pack_view(ws, None, 48, 64, ws, None) == -1 and b'NULL' in lib.sucre_last_error()
    assert lib.sucre_pack_view(ws, ws, 48, 64, C.c_void_p(260), None) == -1 and b'8-byte aligned' in lib.sucre_last_error()
    three = (C.c_void_p * 3)(256, 512, 768)
    assert lib.sucre_pack_views(three, three, three, 0, 48, 64, None) == 0            # nothing to do
    assert lib.sucre_pack_views(three, three, three, -1, 48, 64, None) == -2
    assert lib.sucre_pack_views(three, None, three, 3, 48, 64, None) == -1 and b'NULL' in lib.sucre_last_error()
    assert lib.sucre_pack_views(three, three, (C.c_void_p * 3)(256, 0, 768), 3, 48, 64, None) == -1 and b'view 1' in lib.sucre_last_error()
    assert lib.sucre_pack_views(three, three, (C.c_void_p * 3)(256, 512, 772), 3, 48, 64, None) == -1 and b'view 2' in lib.sucre_last_error()
    # round 5: independent images in one launch per iteration
    assert lib.sucre_batch_bytes(0) == 0 and lib.sucre_batch_bytes(3) % 256 == 0 and lib.sucre_batch_bytes(100) >= 1600
    two = (C.c_void_p * 2)(256, 512)
    same = (C.c_void_p * 2)(256, 256)
    nv = (C.c_int * 2)(3, 5)
    assert lib.sucre_fit_run_batch(None, 2, two, None, 48, 64, nv, 0, 1, 0.05, 0.9, 0.999, 1e-8, 0, None) == -1 and b'batch buffer' in lib.sucre_last_error()
    assert lib.sucre_fit_run_batch(ws, 0, two, None, 48, 64, nv, 0, 1, 0.05, 0.9, 0.999, 1e-8, 0, None) == -1
    assert lib.sucre_fit_run_batch(ws, 2, two, None, 48, 64, None, 0, 1, 0.05, 0.9, 0.999, 1e-8, 0, None) == -1
    assert lib.sucre_fit_run_batch(ws, 2, same, None, 48, 64, nv, 0, 1, 0.05, 0.9, 0.999, 1e-8, 0, None) == -1 and b'share a workspace' in lib.sucre_last_error()
    assert lib.sucre_fit_run_batch(ws, 2, two, None, 48, 64, (C.c_int * 2)(3, 0), 0, 1, 0.05, 0.9, 0.999, 1e-8, 0, None) == -1 and b'invalid geometry' in lib.sucre_last_error()
    assert lib.sucre_fit_run_batch(ws, 2, two, None, 48, 64, nv, 0, -1, 0.05, 0.9, 0.999, 1e-8, 0, None) == -2
    assert lib.sucre_fit_run_batch(ws, 2, two, None, 48, 64, nv, 0, 1, 0.05, 0.9, 0.999, 1e-8, 64, None) == -1 and b'flags' in lib.sucre_last_error()
    assert lib.sucre_fit_run_batch(ws, 2, (C.c_void_p * 2)(256, 260), None, 48, 64, nv, 0, 1, 0.05, 0.9, 0.999, 1e-8, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', tmp_path / 'libsucre_hip.so')
    with pytest.raises(_lib.SucreError, match='no fallback'):
        _lib.load()
