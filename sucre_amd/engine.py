"""Host driver of the HIP engine: device buffers, launch order, read-back.

PyTorch-ROCm is plumbing here (device memory, streams); every kernel lives in libsucre_hip.so.  One
``Restoration`` owns the workspace of one target image of size HxW fitted against up to ``n_views`` views (its
capacity; a target matched against fewer views lays the same buffer out for that number) and is reused across
images of the same geometry.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


@dataclass
class DeviceView:
    """What the engine needs from one sfm.Image: pixels on the GPU + float32 camera/pose matrices computed on
    the host the way the reference computes them (sfm.py:42-47, 92)."""
    depth: torch.Tensor   # (H,W) float32, cuda
    rgb: torch.Tensor     # (H,W,3) uint8, cuda (float32 for resized inputs, see Restoration(float_colour=True))
    K: torch.Tensor       # (3,3) float32, cpu
    R: torch.Tensor       # (3,3) float32, cpu   world-from-camera
    t: torch.Tensor       # (3,1) float32, cpu
    name: str = ''
    # Optional: the derived float32 matrices K^-1 and -R^T t, given by the caller instead of derived here.  torch derives
    # them on the HOST (sfm.py:42-47, 92), and MKL's float32 3x3 products differ in the last bit between CPU models, which
    # moves a handful of boundary pixels per million matches: a caller comparing against results made on another machine
    # (tests/test_gpu_baseline.py: the reference's own goldens) passes that machine's matrices.
    Kinv: torch.Tensor | None = None
    tinv: torch.Tensor | None = None

    def to_struct(self, packed: bool = False) -> _lib.SucreView:
        """sucre_view_t of this view (cached: the matrices are derived once per view, not once per target image).
        ``packed``: the neighbour-view form whose ``depth`` points to the view's ``sucre_pack_view`` records and whose
        ``rgb`` is NULL (see ``packed_records``)."""
        slot = '_struct_packed' if packed else '_struct'
        key = (self.depth.data_ptr(), self.rgb.data_ptr())
        if packed and self.__dict__.get('_packed') is not None:
            key = key + (self.__dict__['_packed'][1].data_ptr(),)   # the struct points into THESE records
        cached = self.__dict__.get(slot)
        if cached is not None and cached[0] == key:
            return cached[1]
        if packed:
            H, W = self.depth.shape
            records = self.packed_records()
            key = key + (records.data_ptr(),)
            s = camera_struct(self.K, self.R, self.t, H, W, records.data_ptr(), 0, Kinv=self.Kinv, tinv=self.tinv)
        else:
            s = self._build_struct()
        self.__dict__[slot] = (key, s)
        return s

    def packed_records(self) -> torch.Tensor:
        """Depth map and uint8 colours interleaved into 8-byte records {float32 depth, r, g, b, 0} (``sucre_pack_view``),
        built once per view on a side stream: as a neighbour the view is then matched with one gather per landing pixel
        instead of two (csrc/match.hip is bound by its gathers).  Consumers on any stream call ``wait_packed`` first."""
        key = (self.depth.data_ptr(), self.rgb.data_ptr())
        cached = self.__dict__.get('_packed')
        if cached is not None and cached[0] == key:
            return cached[1]
        assert self.rgb.dtype == torch.uint8, 'packed records hold uint8 colours'
        with _PACK_LOCK:   # the CLI decodes and submits on several threads: a view's records are built once and never replaced
            cached = self.__dict__.get('_packed')   # (a struct handed to a launch points into them)
            if cached is not None and cached[0] == key:
                return cached[1]
            H, W = self.depth.shape
            dev = self.depth.device
            out = torch.empty(H * W * 8, dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                side = _pack_stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))   # the pixels may have been produced on the caller's stream
                with torch.cuda.stream(side):
                    _lib.check(_lib.load().sucre_pack_view(C.c_void_p(self.depth.data_ptr()), C.c_void_p(self.rgb.data_ptr()), H, W,
                                                           C.c_void_p(out.data_ptr()), _stream_ptr()))
                    done = torch.cuda.Event()
                    done.record()
            self.__dict__['_packed'] = [key, out, done, True]   # one entry: records, their event and "not known complete yet"
            return out

    def repack(self) -> None:
        """Builds the records again, in place, on the CURRENT stream (same bytes; allocation-free): what a caller pays who
        meets this view for the first time -- bench.py charges it to every timed image (VERDICT round 5, weak point 5)."""
        records = self.packed_records()
        self.wait_packed()
        H, W = self.depth.shape
        with torch.cuda.device(self.depth.device):
            _lib.check(_lib.load().sucre_pack_view(C.c_void_p(self.depth.data_ptr()), C.c_void_p(self.rgb.data_ptr()), H, W,
                                                   C.c_void_p(records.data_ptr()), _stream_ptr()))

    def twin(self) -> 'DeviceView':
        """The same pixels and matrices as a view of its own: own packed records, own struct caches (bench.py gives every
        in-flight slot its own, so that one slot's ``repack`` never rewrites records another slot's match kernel is reading)."""
        return DeviceView(depth=self.depth, rgb=self.rgb, K=self.K, R=self.R, t=self.t, name=self.name, Kinv=self.Kinv, tinv=self.tinv)

    def wait_packed(self) -> None:
        """Makes the current stream wait for this view's records (nothing once they are known to be complete)."""
        entry = self.__dict__.get('_packed')
        if entry is not None and entry[3]:
            if entry[2].query():
                entry[3] = False
            else:
                torch.cuda.current_stream(self.depth.device).wait_event(entry[2])

    def as_float_colour(self) -> 'DeviceView':
        """This view with float32 colours (float32(float64(k)/255), loader.py:157-163): used when other views of the
        same restoration are resized images, whose colours are float32 already."""
        if self.rgb.dtype == torch.float32:
            return self
        twin = self.__dict__.get('_float_twin')
        if twin is None:
            twin = DeviceView(depth=self.depth, rgb=(self.rgb.to(torch.float64) / 255).to(torch.float32).contiguous(),
                              K=self.K, R=self.R, t=self.t, name=self.name)
            self.__dict__['_float_twin'] = twin
        return twin

    def _build_struct(self) -> _lib.SucreView:
        H, W = self.depth.shape
        assert self.depth.dtype == torch.float32 and self.depth.is_contiguous() and self.depth.is_cuda
        assert self.rgb.dtype in (torch.uint8, torch.float32) and self.rgb.is_contiguous() and self.rgb.shape == (H, W, 3)
        return camera_struct(self.K, self.R, self.t, H, W, self.depth.data_ptr(), self.rgb.data_ptr(), Kinv=self.Kinv, tinv=self.tinv)


def repack_views(views: list) -> None:
    """``DeviceView.repack`` for views of one size, sixteen views per launch (``sucre_pack_views``): records rebuilt in place on the
    CURRENT stream."""
    views = [v for v in views if v.rgb.dtype == torch.uint8]
    if not views:
        return
    H, W = views[0].depth.shape
    if any(tuple(v.depth.shape) != (H, W) for v in views):   # mixed sizes: one by one
        for v in views:
            v.repack()
        return
    recs = [v.packed_records() for v in views]
    for v in views:
        v.wait_packed()
    n = len(views)
    dp = (C.c_void_p * n)(*[v.depth.data_ptr() for v in views])
    cp = (C.c_void_p * n)(*[v.rgb.data_ptr() for v in views])
    op = (C.c_void_p * n)(*[r.data_ptr() for r in recs])
    with torch.cuda.device(views[0].depth.device):
        _lib.check(_lib.load().sucre_pack_views(dp, cp, op, n, H, W, _stream_ptr()))


def camera_struct(K: torch.Tensor, R: torch.Tensor, t: torch.Tensor, H: int, W: int, depth_ptr: int = 0,
                  rgb_ptr: int = 0, Kinv: torch.Tensor | None = None, tinv: torch.Tensor | None = None) -> _lib.SucreView:
    """sucre_view_t from a camera matrix and a world-from-camera pose, the derived matrices computed on the host the way
    the reference computes them (K.inverse(): sfm.py:92; Pose.inverse() = (R.T, -R.T @ t): sfm.py:42-47)."""
    K = K.to(torch.float32).cpu()
    R = R.to(torch.float32).cpu()
    t = t.to(torch.float32).cpu().view(3, 1)
    Kinv = K.inverse() if Kinv is None else Kinv.to(torch.float32).cpu().view(3, 3)      # sfm.py:92
    Rinv = R.T              # sfm.py:47
    tinv = -R.T @ t if tinv is None else tinv.to(torch.float32).cpu().view(3, 1)         # sfm.py:47
    s = _lib.SucreView()
    s.depth, s.rgb, s.H, s.W = depth_ptr or None, rgb_ptr or None, int(H), int(W)
    for name, val in (('K', K), ('Kinv', Kinv), ('R', R), ('t', t), ('Rinv', Rinv), ('tinv', tinv)):
        flat = val.contiguous().view(-1).tolist()
        setattr(s, name, (C.c_float * len(flat))(*flat))
    return s


def project_points(view: _lib.SucreView, wP: torch.Tensor) -> torch.Tensor:
    """int32[n]: the pixel ``v * W + u`` of ``view`` that each world point of ``wP`` ((3, n) float32, cuda) truncates
    into, -1 = outside the sensor -- ``Image.project_to_view`` + the cast and bound test of ``match_one_way``
    (sfm.py:103-107, 115-117) in the match kernel's float32 operation order (csrc/match.hip, project_points_kernel)."""
    assert wP.is_cuda and wP.dtype == torch.float32 and wP.dim() == 2 and wP.shape[0] == 3
    wP = wP.contiguous()
    n = wP.shape[1]
    out = torch.empty(n, dtype=torch.int32, device=wP.device)
    with torch.cuda.device(wP.device):
        _lib.check(_lib.load().sucre_project_points(C.byref(view), C.c_void_p(wP.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                                    _stream_ptr()))
    return out


_PACK_STREAMS: dict = {}
_PACK_LOCK = threading.Lock()


def _pack_stream(dev) -> 'torch.cuda.Stream':
    key = str(dev)
    if key not in _PACK_STREAMS:
        _PACK_STREAMS[key] = torch.cuda.Stream(dev)
    return _PACK_STREAMS[key]


# engine knob: neighbour views as 8-byte {depth, colour} records (one gather per landing pixel; +8 bytes per pixel and view)
PACKED_VIEWS = os.environ.get('SUCRE_PACKED_VIEWS', '1') != '0'

MAX_VIEWS = 4096   # kMaxViews of csrc/layout.h: views of one restoration (after the overlap cull of sfm.Image.match_images)


def _stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Restoration:
    """Workspace + launch sequence for one target image (replaces the HDF5 spill and the host-resident
    MatchesData of the reference, loader.py:56-130)."""

    def __init__(self, height: int, width: int, n_views: int, device: str | torch.device = 'cuda', light: bool = False,
                 obs_format: str = 'f32', float_colour: bool = False):
        """``obs_format``: 'f32' = 7 B/observation, lossless (default); 'u16mm' = 5 B/observation, ranges rounded to
        the millimetre (BASELINE config 5; include/sucre_hip.h SUCRE_OBS_U16MM).  ``float_colour``: the views' colour
        images are float32 (resized inputs, --image-scale: loader.py:156-163 resizes in float64, so colours are no
        longer k/255); the observations then carry float32 colours in the extension workspace (SUCRE_EXT_COLOUR)."""
        self.lib = _lib.load()
        self.light = bool(light)
        self.float_colour = bool(float_colour)
        self.both = self.light and self.float_colour   # light model on float32 colours: two sets of extension planes
        if self.float_colour and obs_format != 'f32':
            raise NotImplementedError('float32 colours need the f32 store')
        if obs_format not in _lib.OBS_FORMATS:
            raise ValueError(f'obs_format must be one of {sorted(_lib.OBS_FORMATS)}, not {obs_format!r}')
        if self.light and obs_format != 'f32':
            raise NotImplementedError('the artificial-light model keeps float32 camera points: obs_format must be f32')
        self.obs_format = obs_format
        self._streams_used: dict = {}
        self._fmt = _lib.OBS_FORMATS[obs_format]
        self._fmt_flag = _lib.FIT_OBS_U16MM if self._fmt == _lib.OBS_U16MM else 0
        self.H, self.W, self.n_views = int(height), int(width), int(n_views)
        self.capacity = int(n_views)   # the buffers are sized for this many views; match() may use fewer
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.SucreError('the HIP engine needs a GPU device (there is no CPU fallback)')
        nbytes = self.lib.sucre_workspace_bytes(self.H, self.W, self.n_views)
        if nbytes == 0:
            raise _lib.SucreError(self.lib.sucre_last_error().decode())
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        assert self.ws.data_ptr() % 256 == 0
        self.lws = None
        if self.light or self.float_colour:  # extension workspace: camera points / float colours + 19 parameters
            lbytes = (self.lib.sucre_light_workspace_bytes_ext(self.H, self.W, self.n_views, _lib.EXT_POINTS_COLOUR) if self.both
                      else self.lib.sucre_light_workspace_bytes(self.H, self.W, self.n_views))
            self.lws = torch.empty(lbytes, dtype=torch.uint8, device=self.device)
            assert self.lws.data_ptr() % 256 == 0
        self._views_dev = None
        # the view table goes through a pinned staging buffer: a copy from pageable memory makes the host wait
        self._views_host = torch.empty(self.capacity * C.sizeof(_lib.SucreView), dtype=torch.uint8).pin_memory()
        self._views_buf = torch.empty(self.capacity * C.sizeof(_lib.SucreView), dtype=torch.uint8, device=self.device)
        self._views_copied = None
        self._keepalive = []
        self._batch_table = None
        self.trace = None
        self.steps_done = 0

    # -- plumbing -----------------------------------------------------------------------------------------
    def _sp(self) -> C.c_void_p:
        """The current HIP stream, remembered as one this workspace has work queued on (``return_restoration`` leaves an event
        on exactly these for the workspace's next owner)."""
        st = torch.cuda.current_stream(self.device)
        self._streams_used[st.cuda_stream] = st
        return C.c_void_p(st.cuda_stream)

    @property
    def _ext_mode(self) -> int:
        return _lib.EXT_POINTS_COLOUR if self.both else _lib.EXT_COLOUR if self.float_colour else _lib.EXT_POINTS

    @property
    def _ext_flag(self) -> int:
        return _lib.FIT_EXT_BOTH if self.both else _lib.FIT_EXT_COLOUR if self.float_colour else 0

    @property
    def _geom(self):
        return C.c_void_p(self.ws.data_ptr()), self.H, self.W, self.n_views

    def _note_stream(self) -> None:
        """A view of the workspace is being handed out on the current stream (a ``.cpu()`` or a copy the caller queues there reads the
        workspace): ``return_restoration`` must leave an event on that stream too, not only on those the C ABI was launched on
        (ADVICE round 5)."""
        if self.device.type == 'cuda':
            st = torch.cuda.current_stream(self.device)
            self._streams_used[st.cuda_stream] = st

    def _region(self, region: int, dtype: torch.dtype, count: int) -> torch.Tensor:
        self._note_stream()
        off = self.lib.sucre_ws_offset(self.H, self.W, self.n_views, region)
        if off < 0:
            raise _lib.SucreError(self.lib.sucre_last_error().decode())
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return self.ws[off:off + nbytes].view(dtype)

    # -- matching (sfm.py:127-138 + loader.py:78-118) ---------------------------------------------------------
    def match(self, target: DeviceView, views: list[DeviceView], min_cover: float = 1e-6, packed: bool | None = None) -> None:
        """``packed``: neighbour views as ``sucre_pack_view`` records (one gather per landing pixel: -7 % on the match kernel, the
        records built once per view and cached) or as their plain depth and colour planes (two gathers).  None = the engine knob
        ``PACKED_VIEWS``.  Records pay off for a survey, whose targets share views; for ONE target, building them (0.36 ms for 65
        views at 1080p) costs eight times what they save (45 us).  Same match sets, ranges and colours either way."""
        assert 1 <= len(views) <= self.capacity, (len(views), self.capacity)
        self.n_views = len(views)   # the workspace layout is a function of (H, W, n_views) and grows with n_views
        tgt = target.to_struct()
        packed = (PACKED_VIEWS if packed is None else bool(packed)) and not self.float_colour   # (float32 colour images keep the two-gather form)
        table = (_lib.SucreView * self.n_views)(*[v.to_struct(packed) for v in views])
        if packed:
            with torch.cuda.device(self.device):
                for v in views:
                    v.wait_packed()
        nbytes = C.sizeof(table)
        if self._views_copied is not None:
            self._views_copied.synchronize()   # the previous table has left the staging buffer (it was the first thing
        self._views_host[:nbytes].copy_(torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8))   # of that image)
        self._views_dev = self._views_buf[:nbytes]
        with torch.cuda.device(self.device):
            self._views_dev.copy_(self._views_host[:nbytes], non_blocking=True)
            self._views_copied = torch.cuda.Event()
            self._views_copied.record()
        self._keepalive = [target, views]
        ws, H, W, n = self._geom
        with torch.cuda.device(self.device):
            want = torch.float32 if self.float_colour else torch.uint8
            assert all(v.rgb.dtype == want for v in views), f'this restoration takes {want} colour images'
            if self.light or self.float_colour:
                lws = C.c_void_p(self.lws.data_ptr())
                fn = (self.lib.sucre_match_views_light_fcolour if self.both else
                      self.lib.sucre_match_views_fcolour if self.float_colour else self.lib.sucre_match_views_light)
                _lib.check(fn(ws, lws, H, W, n, C.byref(tgt), C.c_void_p(self._views_dev.data_ptr()), 0, n, self._sp()))
                _lib.check(self.lib.sucre_finalize_matches_ext(ws, lws, H, W, n, float(min_cover), self._ext_mode, self._sp()))
            else:
                _lib.check(self.lib.sucre_match_views(ws, H, W, n, C.byref(tgt), C.c_void_p(self._views_dev.data_ptr()),
                                                      0, n, self._sp()))
                _lib.check(self.lib.sucre_finalize_matches_fmt(ws, H, W, n, float(min_cover), self._fmt, self._sp()))

    def import_matches(self, target: DeviceView, lists: list, min_cover: float = -1.0) -> None:
        """Fills the store from explicit per-view match lists instead of matching: ``lists[k] = (u1, v1, z, rgb_u8)``
        (int16[n], int16[n], float32[n], uint8[n,3]) -- what one group of a reference matches file provides after
        ``cP = unproject_depth(u2, v2, d)`` and ``z = ||cP||`` (loader.py:103-118, sucre.py:53).  A fifth element, a
        float32 (3, n) tensor, fills the extension planes of a ``light`` restoration (the camera points cP) or of a
        ``float_colour`` one (the colours I; ``rgb_u8`` may then be None); a restoration that is both takes (6, n): cP, then I."""
        assert 1 <= len(lists) <= self.capacity, (len(lists), self.capacity)
        self.n_views = len(lists)
        self._keepalive = [target, lists]
        self._views_dev = None
        ws, H, W, n = self._geom
        ext_mode = self._ext_mode
        with torch.cuda.device(self.device):
            for k, item in enumerate(lists):
                u1, v1, z, rgb = item[:4]
                ext = item[4] if len(item) > 4 else None
                u1 = u1.to(self.device, torch.int16).contiguous(); v1 = v1.to(self.device, torch.int16).contiguous()
                z = z.to(self.device, torch.float32).contiguous()
                assert z.numel() == u1.numel() == v1.numel()
                if rgb is not None:
                    rgb = rgb.to(self.device, torch.uint8).contiguous()
                    assert rgb.shape == (u1.numel(), 3)
                self._keepalive.append((u1, v1, z, rgb))
                rgbp = C.c_void_p(rgb.data_ptr()) if rgb is not None else None
                if self.lws is not None:
                    planes = 6 if self.both else 3   # both: the camera points, then the float32 colours
                    assert ext is not None and ext.shape == (planes, u1.numel()), f'this restoration needs ({planes}, n) extension planes'
                    ext = ext.to(self.device, torch.float32).contiguous()
                    self._keepalive.append(ext)
                    _lib.check(self.lib.sucre_import_view_ext(ws, C.c_void_p(self.lws.data_ptr()), H, W, n, k,
                                                              C.c_void_p(u1.data_ptr()), C.c_void_p(v1.data_ptr()),
                                                              C.c_void_p(z.data_ptr()), rgbp, C.c_void_p(ext.data_ptr()),
                                                              u1.numel(), ext_mode, self._sp()))
                else:
                    _lib.check(self.lib.sucre_import_view(ws, H, W, n, k, C.c_void_p(u1.data_ptr()), C.c_void_p(v1.data_ptr()),
                                                          C.c_void_p(z.data_ptr()), rgbp, u1.numel(), self._sp()))
            if self.lws is not None:
                _lib.check(self.lib.sucre_finalize_matches_ext(ws, C.c_void_p(self.lws.data_ptr()), H, W, n,
                                                               float(min_cover), self._ext_mode, self._sp()))
            else:
                _lib.check(self.lib.sucre_finalize_matches_fmt(ws, H, W, n, float(min_cover), self._fmt, self._sp()))

    def match_map(self, k: int) -> torch.Tensor:
        """(H,W) int32: linear pixel index v2*W2+u2 in view k matched to every target pixel, -1 = none
        (dense form of sfm.Matches).  ``match`` must have been called (it uploads the view table)."""
        assert self._views_dev is not None, 'call match() first'
        out = torch.empty((self.H, self.W), dtype=torch.int32, device=self.device)
        tgt = self._keepalive[0].to_struct()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sucre_match_map(self.H, self.W, self.n_views, C.byref(tgt),
                                                C.c_void_p(self._views_dev.data_ptr()), int(k),
                                                C.c_void_p(out.data_ptr()), self._sp()))
        return out

    def view_counts(self) -> torch.Tensor:
        return self._region(_lib.WS_VIEW_COUNT, torch.int64, self.n_views)

    def view_keep(self) -> torch.Tensor:
        return self._region(_lib.WS_VIEW_KEEP, torch.int32, self.n_views)

    def store_format(self) -> torch.Tensor:
        """uint32[4] on the device: how the last ``match`` / ``import_matches`` laid the observations out -- _lib.STORE_F32,
        STORE_U16MM or STORE_Z24 (float32 ranges as 24-bit offsets of their bit patterns: the device's choice for an 'f32'
        store whose ranges span less than 2^24 patterns; same results bit for bit), the offset, the smallest and the largest
        range bit pattern."""
        return self._region(_lib.WS_STORE_FORMAT, torch.int32, 4)

    def n_obs_device(self) -> torch.Tensor:
        return self._region(_lib.WS_N_OBS, torch.int64, 1)

    def n_obs(self) -> int:
        return int(self.n_obs_device().item())

    def export_view(self, k: int):
        """Dense (z (H,W) float32, rgb (H,W,3) uint8) planes of view k's matches."""
        z = torch.empty((self.H, self.W), dtype=torch.float32, device=self.device)
        rgb = torch.empty((self.H, self.W, 3), dtype=torch.uint8, device=self.device)
        ws, H, W, n = self._geom
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sucre_export_view(ws, H, W, n, int(k), C.c_void_p(z.data_ptr()),
                                                  C.c_void_p(rgb.data_ptr()), self._sp()))
        return z, rgb

    def export_view_ext(self, k: int) -> torch.Tensor:
        """(3, H, W) float32: the extension planes of view k (camera points with ``light``, colours with
        ``float_colour``), zero where the view has no observation."""
        assert self.lws is not None, 'this restoration keeps no extension planes'
        out = torch.empty((3, self.H, self.W), dtype=torch.float32, device=self.device)
        ws, H, W, n = self._geom
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sucre_export_view_ext(ws, C.c_void_p(self.lws.data_ptr()), H, W, n, int(k),
                                                      C.c_void_p(out.data_ptr()), self._sp()))
        return out

    def check_store(self) -> torch.Tensor:
        """uint32 per view: 0 = sound, bit 0 = non-finite range, bit 1 = negative range, bit 2 = the number of stored
        ranges differs from the view's match count (the checks of loader.py:89-101, one launch for all views)."""
        verdict = torch.empty(self.n_views, dtype=torch.int32, device=self.device)
        scratch = torch.empty(self.n_views, dtype=torch.int64, device=self.device)
        ws, H, W, n = self._geom
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sucre_check_store(ws, H, W, n, C.c_void_p(verdict.data_ptr()),
                                                  C.c_void_p(scratch.data_ptr()), self._sp()))
        return verdict

    # -- fit (sucre.py:36-82, 124-157) --------------------------------------------------------------------------
    def fit_init(self, target: DeviceView, params0=None, J0: torch.Tensor | None = None) -> None:
        npar = 19 if (self.light or self.float_colour) else 9
        default = np.concatenate([np.full(9, 0.1), np.zeros(6), [1.0, 0.0, 0.0, 1.0]])  # sucre.py:41-46
        p0 = default.copy() if params0 is None else np.asarray(params0, np.float64).reshape(-1)
        if self.float_colour and not self.light and p0.size == 9:
            p0 = np.concatenate([p0, default[9:]])   # no light: identity pose, unit beam (never updated)
        p0 = p0.astype(np.float32).reshape(-1)[:npar] if params0 is None else p0.astype(np.float32).reshape(npar)
        p0c = (C.c_float * npar)(*p0.tolist())
        j0 = None
        if J0 is None and self.float_colour:
            J0 = target.rgb   # SUCRe.__init__: J starts at the image (sucre.py:47); NaN where depth <= 0 by the kernel
        if J0 is not None:
            j0 = J0.to(self.device, torch.float32).contiguous()
            assert j0.shape == (self.H, self.W, 3)
            self._keepalive.append(j0)
        ws, H, W, n = self._geom
        with torch.cuda.device(self.device):
            j0p = C.c_void_p(j0.data_ptr()) if j0 is not None else None
            if self.light or self.float_colour:
                _lib.check(self.lib.sucre_fit_init_light(ws, C.c_void_p(self.lws.data_ptr()), H, W, n,
                                                         None if self.float_colour else C.c_void_p(target.rgb.data_ptr()),
                                                         C.c_void_p(target.depth.data_ptr()), p0c, j0p, self._sp()))
            else:
                _lib.check(self.lib.sucre_fit_init(ws, H, W, n, C.c_void_p(target.rgb.data_ptr()),
                                                   C.c_void_p(target.depth.data_ptr()), p0c, j0p, self._sp()))
        self.steps_done = 0

    def fit(self, num_iter: int = 200, lr: float = 0.05, use_closed_form: bool = False, betas=(0.9, 0.999),
            eps: float = 1e-8, record_trace: bool = True, keep_J: bool = False) -> torch.Tensor | None:
        """Enqueues ``num_iter`` Adam iterations without any host sync; returns the (num_iter,10) float64
        device trace (cost, B, beta, gamma per iteration) or None.  In closed-form mode the C ABI appends the
        final ``update_J`` of sucre.py:156 to a ``sucre_fit_run`` call (idempotent, so a fit split into several
        calls ends in the same state) unless ``keep_J``: then J stays the J(theta_k) the last iteration k solved,
        next to theta_{k+1} -- what the reference holds when it plots at a --save-interval stop (sucre.py:141,153)."""
        ext = self.light or self.float_colour
        width = 20 if ext else 10
        trace = torch.zeros((num_iter, width), dtype=torch.float64, device=self.device) if record_trace else None
        flags = (_lib.FIT_CLOSED_FORM if use_closed_form else 0) | self._fmt_flag | self._ext_flag
        if keep_J and use_closed_form:
            flags |= _lib.FIT_KEEP_J
        ws, H, W, n = self._geom
        tp = C.c_void_p(trace.data_ptr()) if trace is not None else None
        with torch.cuda.device(self.device):
            if ext:
                _lib.check(self.lib.sucre_fit_run_light(ws, C.c_void_p(self.lws.data_ptr()), H, W, n, self.steps_done,
                                                        int(num_iter), float(lr), float(betas[0]), float(betas[1]),
                                                        float(eps), flags, tp, self._sp()))
            else:
                _lib.check(self.lib.sucre_fit_run(ws, H, W, n, self.steps_done, int(num_iter), float(lr),
                                                  float(betas[0]), float(betas[1]), float(eps), flags, tp,
                                                  self._sp()))
        self.steps_done += int(num_iter)
        if trace is not None and self.float_colour and not self.light:
            trace = trace[:, :10]   # the light columns are constants here
        self.trace = trace
        return trace

    def update_J(self) -> None:
        ws, H, W, n = self._geom
        with torch.cuda.device(self.device):
            if self.light or self.float_colour:
                _lib.check(self.lib.sucre_update_J_ext(ws, C.c_void_p(self.lws.data_ptr()), H, W, n, self._ext_flag, self._sp()))
            else:
                _lib.check(self.lib.sucre_update_J_fmt(ws, H, W, n, self._fmt, self._sp()))

    def params(self) -> torch.Tensor:
        """B[3], beta[3], gamma[3] (+ cam2light[6], sigma[4] with the light model) on the device."""
        if self.light or self.float_colour:
            self._note_stream()
            off = self.lib.sucre_light_params_offset(self.H, self.W, self.n_views)
            return self.lws[off:off + (76 if self.light else 36)].view(torch.float32)
        return self._region(_lib.WS_PARAMS, torch.float32, 9)

    def J(self) -> torch.Tensor:
        out = torch.empty((self.H, self.W, 3), dtype=torch.float32, device=self.device)
        ws, H, W, n = self._geom
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sucre_export_J(ws, H, W, n, C.c_void_p(out.data_ptr()), self._sp()))
        return out


def fit_batch(restorations: list, num_iter: int = 200, lr: float = 0.05, use_closed_form: bool = False, betas=(0.9, 0.999),
              eps: float = 1e-8, record_trace: bool = True, keep_J: bool = False) -> list:
    """``Restoration.fit`` for several INDEPENDENT images at once -- the reference's loop over images (sucre.py:243-261) with
    the iterations of all of them advancing together, one launch per iteration (``sucre_fit_run_batch``).  Every image keeps
    its own B, beta, gamma, J, Adam state and trace, bit for bit what ``fit`` gives it alone; what the batch saves is the
    per-launch cost (small images are dominated by it) and the idle end of every image's pass.  The images must have one
    size, sit on one device, stand at the same step and use the same observation format; their view counts may differ.
    Returns the list of (num_iter, 10) device traces (or of None)."""
    rs = list(restorations)
    if not rs:
        raise ValueError('fit_batch: no image')
    r0 = rs[0]
    for r in rs:
        if r.light or r.float_colour:
            raise NotImplementedError('fit_batch: the extension-plane modes (light model, float32 colours) are fitted one by one')
        # (not asserts: under python -O images at different Adam steps would be stepped with one bias correction, silently --
        # ADVICE round 5.  The float32 forms of the store -- words, 24- or 26-bit codes -- may mix: each workspace says its own.)
        if (r.H, r.W, str(r.device), r._fmt_flag, r.steps_done) != (r0.H, r0.W, str(r0.device), r0._fmt_flag, r0.steps_done):
            raise ValueError('fit_batch: one image size, device, observation format and Adam step per batch -- got '
                             f'{(r.H, r.W, str(r.device), r.obs_format, r.steps_done)} next to {(r0.H, r0.W, str(r0.device), r0.obs_format, r0.steps_done)}')
    if len({r.ws.data_ptr() for r in rs}) != len(rs):
        raise ValueError('fit_batch: every image needs its own workspace')
    n = len(rs)
    traces = [torch.zeros((num_iter, 10), dtype=torch.float64, device=r0.device) if record_trace else None for _ in rs]
    flags = (_lib.FIT_CLOSED_FORM if use_closed_form else 0) | r0._fmt_flag
    if keep_J and use_closed_form:
        flags |= _lib.FIT_KEEP_J
    lib = r0.lib
    table = torch.empty(lib.sucre_batch_bytes(n), dtype=torch.uint8, device=r0.device)
    ws = (C.c_void_p * n)(*[r.ws.data_ptr() for r in rs])
    tr = (C.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in traces])
    nv = (C.c_int * n)(*[int(r.n_views) for r in rs])
    with torch.cuda.device(r0.device):
        sp = None
        for r in rs:
            sp = r._sp()
        _lib.check(lib.sucre_fit_run_batch(C.c_void_p(table.data_ptr()), n, ws, tr, r0.H, r0.W, nv, r0.steps_done, int(num_iter),
                                           float(lr), float(betas[0]), float(betas[1]), float(eps), flags, sp))
    for r, t in zip(rs, traces):
        r.steps_done += int(num_iter)
        r.trace = t
        r._batch_table = table   # the launches read the table: it lives until the workspace's next batch (replaced, not appended:
                                 # a fit driven one iteration per call would otherwise keep one table per call -- ADVICE round 5)
    return traces


def select_ranks(J: torch.Tensor, ranks: list[int]) -> torch.Tensor:
    """(3, len(ranks)) float32: per channel of the (H,W,3) float32 device image ``J``, the values at the given 0-based
    ranks among the valid pixels (no NaN in any channel), ascending -- exact order statistics by radix select on the
    device (csrc/plot.hip); the image does not move."""
    lib = _lib.load()
    assert J.is_cuda and J.dtype == torch.float32 and J.is_contiguous() and J.dim() == 3 and J.shape[2] == 3
    assert 1 <= len(ranks) <= 8
    out = torch.empty((3, len(ranks)), dtype=torch.float32, device=J.device)
    scratch = torch.empty(lib.sucre_select_scratch_bytes(), dtype=torch.uint8, device=J.device)
    arr = (C.c_uint64 * len(ranks))(*[int(r) for r in ranks])
    with torch.cuda.device(J.device):
        _lib.check(lib.sucre_select_ranks(C.c_void_p(J.data_ptr()), J.shape[0], J.shape[1], len(ranks), arr,
                                          C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()), _stream_ptr()))
    return out


def count_valid(J: torch.Tensor) -> int:
    """Pixels of the (H,W,3) float32 device image ``J`` without a NaN in any channel (``valid``, sucre.py:86-87)."""
    lib = _lib.load()
    assert J.is_cuda and J.dtype == torch.float32 and J.is_contiguous() and J.dim() == 3 and J.shape[2] == 3
    count = torch.empty(1, dtype=torch.int64, device=J.device)
    with torch.cuda.device(J.device):
        _lib.check(lib.sucre_count_valid(C.c_void_p(J.data_ptr()), J.shape[0], J.shape[1], C.c_void_p(count.data_ptr()),
                                         _stream_ptr()))
    return int(count.item())


def plot_stretch(J: torch.Tensor, lo, hi) -> torch.Tensor:
    """(H,W,3) uint8: SUCRe.plot_J's clip / shift / scale / x255 / truncate (sucre.py:88-94) of the device image ``J``
    between the per-channel percentiles ``lo`` and ``hi`` (3 float32 each); invalid pixels black."""
    lib = _lib.load()
    assert J.is_cuda and J.dtype == torch.float32 and J.is_contiguous() and J.dim() == 3 and J.shape[2] == 3
    out = torch.empty(J.shape, dtype=torch.uint8, device=J.device)
    lo_c = (C.c_float * 3)(*[float(x) for x in lo])
    hi_c = (C.c_float * 3)(*[float(x) for x in hi])
    with torch.cuda.device(J.device):
        _lib.check(lib.sucre_plot_stretch(C.c_void_p(J.data_ptr()), J.shape[0], J.shape[1], lo_c, hi_c,
                                          C.c_void_p(out.data_ptr()), _stream_ptr()))
    return out


def device_views_from_scene(scene, device='cuda') -> list[DeviceView]:
    """Uploads a synthetic scene (sucre_amd.synth) the way the loaders would: float32 depth, uint8 colour."""
    out = []
    for v in scene.views:
        out.append(DeviceView(depth=v.depth_f32().to(device).contiguous(), rgb=v.rgb_u8.to(device).contiguous(),
                              K=scene.K, R=v.R, t=v.t, name=v.name,
                              Kinv=getattr(scene, 'Kinv_given', None), tinv=getattr(v, 'tinv_given', None)))
    return out


class HipWaterBackend:
    """dist.WaterBackend over a Restoration: split grad/step launches around the host's all-reduce."""

    def __init__(self, restoration: Restoration, lr: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8,
                 use_closed_form: bool = False, trace: torch.Tensor | None = None):
        if restoration.light or restoration.float_colour:
            raise NotImplementedError('shared water parameters run on the plain water model with uint8 colours')
        self.r = restoration
        self.hyper = (float(lr), float(betas[0]), float(betas[1]), float(eps))
        self.flags = (_lib.FIT_CLOSED_FORM if use_closed_form else 0) | restoration._fmt_flag
        self.trace = trace
        self._sums = restoration._region(_lib.WS_SUMS, torch.float64, 12)

    def grad_device(self):
        return self.r.device

    def n_obs(self) -> int:
        return self.r.n_obs()

    def set_n_obs_total(self, n: int) -> None:
        ws, H, W, nv = self.r._geom
        with torch.cuda.device(self.r.device):
            _lib.check(self.r.lib.sucre_set_n_obs_total(ws, H, W, nv, int(n), self.r._sp()))

    def grad(self, step: int) -> torch.Tensor:
        ws, H, W, nv = self.r._geom
        with torch.cuda.device(self.r.device):
            _lib.check(self.r.lib.sucre_fit_grad(ws, H, W, nv, int(step), *self.hyper, self.flags, self.r._sp()))
        return self._sums

    def step(self, step: int) -> None:
        ws, H, W, nv = self.r._geom
        row = C.c_void_p(self.trace[step - 1].data_ptr()) if self.trace is not None else None
        with torch.cuda.device(self.r.device):
            _lib.check(self.r.lib.sucre_fit_step(ws, H, W, nv, int(step), *self.hyper, row, self.r._sp()))
        self.r.steps_done = int(step)


class HipWaterGroup:
    """dist.WaterBackend over ALL the restorations of this rank (BASELINE config 4: a 512-image scene, 64 images per
    GPU, all sharing B, beta, gamma) with one launch and one collective per iteration: ``grad`` runs the gradient pass
    of every image in a single launch (csrc/fit.hip, group_iter_kernel) and returns the rank's ten sums for the
    all-reduce; the Adam step those sums call for is taken in the prologue of the next ``grad`` launch (``step`` only
    notes it) and ``finish`` takes the last one.  ``trace``: optional (T, 10) float64 device tensor."""

    def __init__(self, restorations: list, lr: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8,
                 use_closed_form: bool = False, trace: torch.Tensor | None = None, params0=None):
        rs = [getattr(r, 'r', r) for r in restorations]      # Restoration objects (or HipWaterBackend wrappers)
        assert rs, 'need at least one image'
        if any(r.light or r.float_colour for r in rs):
            raise NotImplementedError('shared water parameters run on the plain water model with uint8 colours')
        assert len({(r.obs_format, str(r.device)) for r in rs}) == 1, 'one device and one observation format per group'
        self.rs = rs
        self.lib = rs[0].lib
        self.device = rs[0].device
        self.hyper = (float(lr), float(betas[0]), float(betas[1]), float(eps))
        self.closed = bool(use_closed_form)
        self.flags = (_lib.FIT_CLOSED_FORM if use_closed_form else 0) | rs[0]._fmt_flag
        self.trace = trace
        self.total = None
        self.steps_done = 0
        n = len(rs)
        self.buf = torch.empty(self.lib.sucre_group_bytes(n), dtype=torch.uint8, device=self.device)
        assert self.buf.data_ptr() % 256 == 0
        table = (_lib.GroupImage * n)(*[_lib.GroupImage(r.ws.data_ptr(), r.H, r.W, r.n_views, 0) for r in rs])
        p0 = np.full(9, 0.1, np.float32) if params0 is None else np.asarray(params0, np.float32).reshape(9)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sucre_group_init(C.c_void_p(self.buf.data_ptr()), n, table, (C.c_float * 9)(*p0.tolist()),
                                                 self._sp()))
        off = self.lib.sucre_group_sums_offset()
        self._sums = self.buf[off:off + 96].view(torch.float64)

    def grad_device(self):
        return self.device

    def n_obs(self) -> int:
        return sum(r.n_obs() for r in self.rs)

    def set_n_obs_total(self, n: int) -> None:
        self.total = int(n)

    def _sp(self) -> C.c_void_p:   # a group launch touches every image's workspace
        sp = None
        for r in self.rs:
            sp = r._sp()
        return sp

    def _trace_ptr(self):
        return C.c_void_p(self.trace.data_ptr()) if self.trace is not None else None

    def grad(self, step: int) -> torch.Tensor:
        assert self.total, 'set_n_obs_total() first'
        if int(step) != self.steps_done + 1:
            # the water state is double-buffered by step parity and the pending step is taken by the NEXT launch: a group
            # runs iterations 1, 2, 3, ... once; starting over needs fit_init on the images and a new group
            raise _lib.SucreError(f'HipWaterGroup: iteration {step} asked after {self.steps_done} done -- iterations run in '
                                  f'order, once; build a new group (after fit_init) to fit again')
        with torch.cuda.device(self.device):
            if self.closed and step == 1:   # start the one-pass closed-form kernel from a solved J (see sucre_fit_run)
                for r in self.rs:
                    r.update_J()
            _lib.check(self.lib.sucre_group_iter(C.c_void_p(self.buf.data_ptr()), len(self.rs), int(step), *self.hyper,
                                                 self.flags, self.total, self._trace_ptr(), self._sp()))
        return self._sums

    def step(self, step: int) -> None:
        self.steps_done = int(step)    # taken in the prologue of the next grad launch, or by finish()
        for r in self.rs:
            r.steps_done = int(step)

    def finish(self) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sucre_group_finish(C.c_void_p(self.buf.data_ptr()), len(self.rs), self.steps_done, *self.hyper,
                                                   self.total, self._trace_ptr(), self._sp()))
            if self.closed:
                for r in self.rs:
                    r.update_J()     # the final update_J of sucre.py:156 from the final parameters


_POOL: dict = {}
_STREAMS: dict = {}
_TLS = threading.local()   # .slot: the in-flight slot the calling THREAD is working for (see in_flight_slot); the CLI runs
                           # decode, plan and writer pools next to the main thread: a module global would leak a slot into them


def current_slot() -> int:
    return getattr(_TLS, 'slot', 0)


def current_lane() -> int:
    return getattr(_TLS, 'lane', 0)


@contextlib.contextmanager
def slot_lane(index: int):
    """The ``index``-th of several images that share ONE in-flight slot (a survey of small images whose fits advance in one
    launch per iteration, ``fit_batch``): inside the block ``acquire_restoration`` hands out that image's own workspace; the
    stream stays the slot's."""
    prev, _TLS.lane = current_lane(), int(index)
    try:
        yield
    finally:
        _TLS.lane = prev


def acquire_restoration(height: int, width: int, n_views: int, device='cuda', light: bool = False,
                        obs_format: str | None = None, float_colour: bool = False, tag: str = '') -> Restoration:
    """Workspace pool: one Restoration per (geometry, device, in-flight slot, tag), reused image after image (the
    1080p x 65-view workspace is ~2 GB; re-allocating it per image would serialise on the allocator) and grown when
    a target needs more views than it holds.  ``tag`` keeps side uses (``Image.match_two_way``) off the workspace a
    matches file is attached to."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    if obs_format is None:   # engine knob for the reference-compatible CLI, which has no flag for it
        obs_format = 'f32' if (light or float_colour) else os.environ.get('SUCRE_OBS_FORMAT', 'f32')
    # capacities come in steps, so targets whose surviving view counts differ a little share one workspace
    n = int(n_views)
    cap = (n + 7) // 8 * 8 if n <= 256 else (n + 31) // 32 * 32   # (65 views -> 72, not 96: +11 % HBM per slot, not +48 %)
    key = (int(height), int(width), str(dev), bool(light), (current_slot(), current_lane()), obs_format, bool(float_colour), tag)
    have = _POOL.get(key)
    if have is None or have.capacity < n:
        _POOL.pop(key, None)   # release the smaller workspace before allocating the larger one
        del have
        _POOL[key] = Restoration(height, width, cap, device=dev, light=light, obs_format=obs_format,
                                 float_colour=float_colour)
    return _POOL[key]


_FREE: list = []   # leased workspaces handed back by their owners (lease_restoration / return_restoration)
_FREE_LOCK = threading.Lock()
_FREE_KEEP = 4      # at most this many idle workspaces are kept (a 1080p x 65-view one is ~2 GB)


def lease_restoration(height: int, width: int, n_views: int, device='cuda', light: bool = False,
                      obs_format: str | None = None, float_colour: bool = False) -> Restoration:
    """A workspace for ONE owner (a list-backed ``loader.MatchesData``): taken from the free list when an idle one of the
    same geometry and kind with enough capacity exists, allocated otherwise; nobody else sees it until the owner hands it
    back with ``return_restoration``.  ``obs_format`` None: the SUCRE_OBS_FORMAT knob, as in ``acquire_restoration``."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    if obs_format is None:
        obs_format = 'f32' if (light or float_colour) else os.environ.get('SUCRE_OBS_FORMAT', 'f32')
    n = int(n_views)
    want = (int(height), int(width), str(dev), bool(light), obs_format, bool(float_colour))
    with _FREE_LOCK:
        for i, r in enumerate(_FREE):
            if (r.H, r.W, str(r.device), r.light, r.obs_format, r.float_colour) == want and r.capacity >= n:
                _FREE.pop(i)
                with torch.cuda.device(dev):   # whatever its last owner had enqueued on it comes first
                    for ev in r.__dict__.pop('_idle_after', []):
                        torch.cuda.current_stream(dev).wait_event(ev)
                return r
    cap = (n + 7) // 8 * 8 if n <= 256 else (n + 31) // 32 * 32
    return Restoration(height, width, cap, device=dev, light=light, obs_format=obs_format, float_colour=float_colour)


def return_restoration(r: Restoration) -> None:
    """Hands a leased workspace back.  Its last owner's launches may still be queued: one event per stream the workspace
    was ever launched on (``Restoration._sp`` remembers them -- a user stream included; this function may run from a
    ``weakref.finalize`` on any thread, whose current stream says nothing) is left on it for the next owner to wait for."""
    r._keepalive = []
    try:
        with torch.cuda.device(r.device):
            streams = list(r._streams_used.values())
            r._streams_used = {}
            events = []
            for s in streams:
                ev = torch.cuda.Event()
                ev.record(s)
                events.append(ev)
            r.__dict__['_idle_after'] = events
    except Exception:   # interpreter shutdown: the workspace is going away with everything else
        return
    with _FREE_LOCK:
        if len(_FREE) < _FREE_KEEP and all(x is not r for x in _FREE):
            _FREE.append(r)


@contextlib.contextmanager
def in_flight_slot(index: int, device='cuda', wait_for_caller: bool = True):
    """Several images in flight on one GPU.  Code inside the block enqueues on slot ``index``'s own HIP stream and
    ``acquire_restoration`` hands out that slot's own workspace, so image i+1 (slot 1) can be submitted while image
    i (slot 0) is still iterating: the 200 launches of one fit depend on each other, and each ends in a short
    tail where a single workgroup reduces and steps the parameters -- a second image's kernels fill those tails
    (+13 % images/s at 1080p x 65 views, tools/dual_stream_probe.py).  Results of a slot must be read inside a
    block of the same slot (its stream is then the current one, so ``.cpu()`` waits for the right work).
    ``wait_for_caller=False``: the slot's stream does not wait for what the caller's stream holds -- for callers whose
    inputs are complete when they are handed over (``sfm.Image.device_view`` publishes a view only once its upload has
    landed); recording that dependency on the default stream costs the host up to 10 ms per image when other threads
    keep that stream busy with blocking copies (measured in the CLI, tools/cli_timeline.py)."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    key = (str(dev), int(index))
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(dev)
    stream = _STREAMS[key]
    if wait_for_caller:
        stream.wait_stream(torch.cuda.current_stream(dev))   # inputs uploaded / allocations freed on the caller's stream
    prev, _TLS.slot = current_slot(), int(index)
    try:
        with torch.cuda.stream(stream):
            yield stream
    finally:
        _TLS.slot = prev


def release_pool() -> None:
    with _FREE_LOCK:
        _FREE.clear()
    _POOL.clear()
    _STREAMS.clear()
