"""ctypes front-end of ``libsucre_oracle.so`` (oracle/sucre_oracle.c).  TEST INFRASTRUCTURE ONLY.

The oracle works on the reference's own data shapes: depth maps ``(H,W)`` float32, colour ``(H,W,3)`` uint8,
per-view observation lists ``(u int16[n], v int16[n], cP float32[3,n], I float32[3,n])`` (loader.py:33-53) and
``J (H,W,3)`` float32.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / 'libsucre_oracle.so'
_lib = None


class CamStruct(C.Structure):
    _fields_ = [('H', C.c_int32), ('W', C.c_int32), ('K', C.c_float * 9), ('Kinv', C.c_float * 9),
                ('R', C.c_float * 9), ('t', C.c_float * 3), ('Rinv', C.c_float * 9), ('tinv', C.c_float * 3)]


def build(force: bool = False) -> Path:
    """Compiles the oracle with gcc (oracle/Makefile)."""
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < (_HERE / 'sucre_oracle.c').stat().st_mtime:
        subprocess.run(['make', '-C', str(_HERE), '-B' if force else '-s'], check=True, capture_output=True)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.oracle_match_view.restype = C.c_int64
        _lib.oracle_fit.restype = C.c_int
        _lib.oracle_update_J.restype = C.c_int
        _lib.oracle_set_num_threads(_usable_cpus())
    return _lib


def _usable_cpus() -> int:
    """OpenMP's default is one thread per CPU of the machine; inside a container with a CPU quota (16 of 256 on the GPU
    boxes) that many threads only get the process throttled.  Scheduler affinity capped by the cgroup quota."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path, split in (('/sys/fs/cgroup/cpu.max', True), ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', False)):
        try:
            if split:
                quota, period = Path(path).read_text().split()[:2]
                if quota != 'max':
                    n = min(n, max(1, int(quota) // int(period)))
            else:
                quota = int(Path(path).read_text())
                period = int(Path('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read_text())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(int(n))


def make_cam(H: int, W: int, K, Kinv, R, t, Rinv, tinv) -> CamStruct:
    """All matrices are float32 arrays computed by the caller exactly as the reference computes them
    (K.inverse(), R.T, -R.T @ t with torch) so the oracle only does the per-pixel arithmetic."""
    cam = CamStruct()
    cam.H, cam.W = int(H), int(W)
    for name, val, n in (('K', K, 9), ('Kinv', Kinv, 9), ('R', R, 9), ('t', t, 3), ('Rinv', Rinv, 9), ('tinv', tinv, 3)):
        arr = np.ascontiguousarray(np.asarray(val, dtype=np.float32)).ravel()
        assert arr.size == n, name
        setattr(cam, name, (C.c_float * n)(*arr.tolist()))
    return cam


def _p(a: np.ndarray, ct):
    return a.ctypes.data_as(C.POINTER(ct))


@dataclass
class ViewMatches:
    u1: np.ndarray
    v1: np.ndarray
    u2: np.ndarray
    v2: np.ndarray
    d: np.ndarray

    def __len__(self) -> int:
        return int(self.u1.shape[0])


def match_view(depth1: np.ndarray, cam1: CamStruct, depth2: np.ndarray, cam2: CamStruct) -> ViewMatches:
    """sfm.Image.match_two_way + d = depth2[v2,u2] (sfm.py:121-125,137)."""
    d1 = np.ascontiguousarray(depth1, dtype=np.float32)
    d2 = np.ascontiguousarray(depth2, dtype=np.float32)
    n_max = d1.size
    u1, v1, u2, v2 = (np.empty(n_max, np.int16) for _ in range(4))
    d = np.empty(n_max, np.float32)
    n = lib().oracle_match_view(_p(d1, C.c_float), C.byref(cam1), _p(d2, C.c_float), C.byref(cam2),
                                _p(u1, C.c_int16), _p(v1, C.c_int16), _p(u2, C.c_int16), _p(v2, C.c_int16),
                                _p(d, C.c_float))
    if n < 0:
        raise MemoryError('oracle_match_view')
    return ViewMatches(u1[:n].copy(), v1[:n].copy(), u2[:n].copy(), v2[:n].copy(), d[:n].copy())


def unproject(cam: CamStruct, u: np.ndarray, v: np.ndarray, d: np.ndarray) -> np.ndarray:
    n = int(u.shape[0])
    out = np.empty((3, n), np.float32)
    u = np.ascontiguousarray(u, np.int16); v = np.ascontiguousarray(v, np.int16)
    d = np.ascontiguousarray(d, np.float32)
    lib().oracle_unproject(C.byref(cam), _p(u, C.c_int16), _p(v, C.c_int16), _p(d, C.c_float), C.c_int64(n),
                           _p(out, C.c_float))
    return out


def gather_rgb(rgb_u8: np.ndarray, u: np.ndarray, v: np.ndarray) -> np.ndarray:
    n = int(u.shape[0])
    out = np.empty((3, n), np.float32)
    rgb = np.ascontiguousarray(rgb_u8, np.uint8)
    u = np.ascontiguousarray(u, np.int16); v = np.ascontiguousarray(v, np.int16)
    lib().oracle_gather_rgb(_p(rgb, C.c_uint8), C.c_int(rgb.shape[1]), _p(u, C.c_int16), _p(v, C.c_int16),
                            C.c_int64(n), _p(out, C.c_float))
    return out


def init_J(rgb_u8: np.ndarray, depth: np.ndarray) -> np.ndarray:
    H, W = depth.shape
    rgb = np.ascontiguousarray(rgb_u8, np.uint8)
    d = np.ascontiguousarray(depth, np.float32)
    J = np.empty((H, W, 3), np.float32)
    lib().oracle_init_J(_p(rgb, C.c_uint8), _p(d, C.c_float), C.c_int(H), C.c_int(W), _p(J, C.c_float))
    return J


class _Samples:
    """Marshals a list of (u, v, cP, I) samples into the pointer tables the oracle takes."""

    def __init__(self, samples):
        self.keep = []
        n = len(samples)
        self.counts = np.array([int(s[0].shape[0]) for s in samples], np.int64)
        self.us = (C.POINTER(C.c_int16) * n)()
        self.vs = (C.POINTER(C.c_int16) * n)()
        self.cPs = (C.POINTER(C.c_float) * n)()
        self.Is = (C.POINTER(C.c_float) * n)()
        for i, (u, v, cP, I) in enumerate(samples):
            u = np.ascontiguousarray(u, np.int16); v = np.ascontiguousarray(v, np.int16)
            cP = np.ascontiguousarray(cP, np.float32); I = np.ascontiguousarray(I, np.float32)
            self.keep += [u, v, cP, I]
            self.us[i], self.vs[i] = _p(u, C.c_int16), _p(v, C.c_int16)
            self.cPs[i], self.Is[i] = _p(cP, C.c_float), _p(I, C.c_float)
        self.n = n


def fit(H: int, W: int, samples, J0: np.ndarray | None, params0=None, num_iter: int = 200, lr: float = 0.05,
        use_closed_form: bool = False):
    """sucre.adam (sucre.py:124-157).  Returns (J (H,W,3), params (9,), trace (num_iter,10))."""
    s = _Samples(samples)
    J = np.ascontiguousarray(J0, np.float32).copy() if J0 is not None else np.zeros((H, W, 3), np.float32)
    params = np.full(9, 0.1, np.float32) if params0 is None else np.ascontiguousarray(params0, np.float32).copy()
    trace = np.zeros((num_iter, 10), np.float64)
    rc = lib().oracle_fit(C.c_int(H), C.c_int(W), C.c_int(s.n), _p(s.counts, C.c_int64), s.us, s.vs, s.cPs, s.Is,
                          _p(J, C.c_float), _p(params, C.c_float), C.c_int(num_iter), C.c_double(lr),
                          C.c_int(int(use_closed_form)), _p(trace, C.c_double))
    if rc != 0:
        raise MemoryError('oracle_fit')
    return J, params, trace


def quantize_ranges_u16mm(samples):
    """The engine's compact observation format (include/sucre_hip.h SUCRE_OBS_U16MM, BASELINE config 5) restated:
    the range z = ||cP|| (sucre.py:53) is kept as uint16 millimetres, rint(1000 z) clamped to [1, 65535], and read
    back as float32(mm) * 0.001f.  Returns samples whose camera points are (0, 0, z') so that every fit function of
    this module sees exactly that range (sqrt(fl(z'^2)) == z' in binary floating point; asserted)."""
    out = []
    for u, v, cP, I in samples:
        cP = np.ascontiguousarray(cP, np.float32)
        z = np.sqrt(cP[0] * cP[0] + cP[1] * cP[1] + cP[2] * cP[2])          # norm3 of sucre_oracle.c, float32
        mm = np.clip(np.rint(z * np.float32(1000.0)), np.float32(1.0), np.float32(65535.0)).astype(np.uint16)
        zq = mm.astype(np.float32) * np.float32(0.001)
        assert np.array_equal(np.sqrt(zq * zq), zq)
        cq = np.zeros_like(cP)
        cq[2] = zq
        out.append((u, v, cq, I))
    return out


def update_J(H: int, W: int, samples, params) -> np.ndarray:
    """SUCRe.update_J (sucre.py:66-77)."""
    s = _Samples(samples)
    J = np.zeros((H, W, 3), np.float32)
    params = np.ascontiguousarray(params, np.float32)
    rc = lib().oracle_update_J(C.c_int(H), C.c_int(W), C.c_int(s.n), _p(s.counts, C.c_int64), s.us, s.vs, s.cPs,
                               s.Is, _p(params, C.c_float), _p(J, C.c_float))
    if rc != 0:
        raise MemoryError('oracle_update_J')
    return J


class SharedWaterImage:
    """One image of a shared-water (lock-step) fit on the CPU oracle: holds J and its Adam moments.
    ``use_closed_form``: J is re-solved by update_J from the shared parameters at the top of every pass
    (sucre.py:141) and takes no Adam step; ``final_update_J`` is the update_J of sucre.py:156."""

    def __init__(self, H: int, W: int, samples, J0: np.ndarray | None, lr: float = 0.05, use_closed_form: bool = False):
        self.H, self.W, self.lr = H, W, lr
        self.closed = bool(use_closed_form)
        self.s = _Samples(samples)
        self.J = np.ascontiguousarray(J0, np.float32).copy() if J0 is not None else np.zeros((H, W, 3), np.float32)
        self.mJ = np.zeros_like(self.J)
        self.vJ = np.zeros_like(self.J)
        self.n_obs = int(self.s.counts.sum())

    def grad(self, params: np.ndarray, step: int, n_obs_total: int) -> np.ndarray:
        sums = np.zeros(10, np.float64)
        params = np.ascontiguousarray(params, np.float32)
        lib().oracle_shared_grad_mode.restype = C.c_int
        rc = lib().oracle_shared_grad_mode(C.c_int(self.H), C.c_int(self.W), C.c_int(self.s.n), _p(self.s.counts, C.c_int64),
                                           self.s.us, self.s.vs, self.s.cPs, self.s.Is, _p(self.J, C.c_float),
                                           _p(self.mJ, C.c_float), _p(self.vJ, C.c_float), _p(params, C.c_float),
                                           C.c_int(step), C.c_double(self.lr), C.c_int64(n_obs_total),
                                           C.c_int(int(self.closed)), _p(sums, C.c_double))
        if rc != 0:
            raise MemoryError('oracle_shared_grad_mode')
        return sums

    def final_update_J(self, params: np.ndarray) -> None:
        params = np.ascontiguousarray(params, np.float32)
        rc = lib().oracle_update_J(C.c_int(self.H), C.c_int(self.W), C.c_int(self.s.n), _p(self.s.counts, C.c_int64),
                                   self.s.us, self.s.vs, self.s.cPs, self.s.Is, _p(params, C.c_float), _p(self.J, C.c_float))
        if rc != 0:
            raise MemoryError('oracle_update_J')


def shared_step(pstate: np.ndarray, sums: np.ndarray, step: int, n_obs_total: int, lr: float = 0.05) -> None:
    """In-place Adam step of the 27-float water state (params, exp_avg, exp_avg_sq)."""
    assert pstate.dtype == np.float32 and pstate.size == 27 and pstate.flags.c_contiguous
    sums = np.ascontiguousarray(sums, np.float64)
    lib().oracle_shared_step.restype = None
    lib().oracle_shared_step(_p(pstate, C.c_float), _p(sums, C.c_double), C.c_int(step), C.c_double(lr),
                             C.c_int64(n_obs_total))


def se3_exp(xi) -> tuple[np.ndarray, np.ndarray]:
    """se3.exp (se3.py:22-27)."""
    xi = np.ascontiguousarray(xi, np.float32)
    R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    lib().oracle_se3_exp(_p(xi, C.c_float), _p(R, C.c_float), _p(t, C.c_float))
    return R.reshape(3, 3), t.reshape(3, 1)


def fit_light(H: int, W: int, samples, J0, params0=None, num_iter: int = 200, lr: float = 0.05,
              use_closed_form: bool = False):
    """sucre.adam with light_model=True (sucre.py:54-61, 124-157).  params: B, beta, gamma, cam2light[6], sigma[4].
    Returns (J, params (19,), trace (num_iter, 20))."""
    s = _Samples(samples)
    J = np.ascontiguousarray(J0, np.float32).copy() if J0 is not None else np.zeros((H, W, 3), np.float32)
    if params0 is None:
        params = np.concatenate([np.full(9, 0.1), np.zeros(6), [1, 0, 0, 1]]).astype(np.float32)
    else:
        params = np.ascontiguousarray(params0, np.float32).copy()
    trace = np.zeros((num_iter, 20), np.float64)
    lib().oracle_fit_light.restype = C.c_int
    rc = lib().oracle_fit_light(C.c_int(H), C.c_int(W), C.c_int(s.n), _p(s.counts, C.c_int64), s.us, s.vs, s.cPs, s.Is,
                                _p(J, C.c_float), _p(params, C.c_float), C.c_int(num_iter), C.c_double(lr),
                                C.c_int(int(use_closed_form)), _p(trace, C.c_double))
    if rc != 0:
        raise MemoryError('oracle_fit_light')
    return J, params, trace
