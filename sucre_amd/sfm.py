"""Structure-from-motion side of the drop-in: poses, cameras, images, matches and the COLMAP model.

Mirrors the reference's ``sfm`` module (sfm.py:32-238) name for name -- ``Pose``, ``Camera``, ``Image``,
``Matches``, ``COLMAPModel`` with the same constructor arguments, attributes and conventions (float32 tensors,
points ``(3, n)``, ``(u, v)`` = (column, row), world-from-camera poses) -- but the hot loop of
``Image.match_images`` (sfm.py:127-138) is one launch of the HIP engine over every view instead of a Python loop
of eager torch kernels, and its result stays in HBM instead of going to an HDF5 file.

pycolmap is not required: ``COLMAPModel`` reads ``cameras/images.{bin,txt}`` itself (PINHOLE only, as the
reference asserts, sfm.py:192).
"""
from __future__ import annotations

import struct
from pathlib import Path

import numpy as np
import collections
import os
import threading

import torch
from torch import Tensor

from . import loader


class Pose:
    def __init__(self, R: Tensor, t: Tensor):
        """Rigid transform ``x -> R @ x + t``.  R: (3,3), t: (3,1)."""
        self.R = R
        self.t = t

    def inverse(self) -> 'Pose':
        Rt = self.R.T
        return Pose(Rt, -Rt @ self.t)

    def transform(self, P: Tensor) -> Tensor:
        """Applies the pose to points ``P`` of shape (3, n)."""
        return self.R.to(P.device) @ P + self.t.to(P.device)

    def __repr__(self) -> str:
        return f'Pose(R={self.R!r}, t={self.t!r})'


class Camera:
    def __init__(self, camera_id: int, width: int, height: int, K: Tensor):
        """PINHOLE camera: id in the COLMAP model, sensor size in pixels, 3x3 intrinsics."""
        self.id = camera_id
        self.width = width
        self.height = height
        self.K = K

    def __repr__(self) -> str:
        return f'Camera(id={self.id}, width={self.width}, height={self.height}, K={self.K!r})'


def _canonical_device(device) -> torch.device:
    """'cuda' and 'cuda:<current>' name the same GPU: the pixel cache must not treat them as different."""
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


def require_gpu(device, what: str) -> None:
    """The reference's functions default to ``device='cpu'`` (sfm.py:128, sucre.py:176); this engine keeps that default in
    its signatures and refuses it loudly -- a silently different default would hide that there is no CPU path."""
    if torch.device(device).type != 'cuda':
        raise RuntimeError(f"{what}(device={str(device)!r}): the HIP engine runs on a GPU and has no CPU path -- "
                           f"pass device='cuda' (or 'cuda:<index>')")


class _PixelCache:
    """Budget for the pixels kept resident in HBM (uint8 colour + float32 depth = 7 B/pixel per image): least
    recently used images are dropped from the cache once the budget (SUCRE_DEVICE_CACHE_GB, default half of the
    device memory) is exceeded.  Dropping only removes the cache's reference -- a restoration in flight keeps its
    views alive through its own references -- so a survey larger than the GPU degrades to re-decoding, not to an
    out-of-memory error."""

    def __init__(self):
        self.lock = threading.Lock()
        self.entries: 'collections.OrderedDict[int, tuple[Image, int]]' = collections.OrderedDict()
        self.total = 0
        self._budget = None

    def budget(self, dev: torch.device) -> int:
        if self._budget is None:
            gb = os.environ.get('SUCRE_DEVICE_CACHE_GB')
            if gb is not None:
                self._budget = int(float(gb) * 2 ** 30)
            elif dev.type == 'cuda':
                self._budget = torch.cuda.mem_get_info(dev)[1] // 2
            else:
                self._budget = 1 << 62
        return self._budget

    def full(self, dev: torch.device) -> bool:
        return self.total >= self.budget(dev)

    def touch(self, image: 'Image') -> None:
        with self.lock:
            if id(image) in self.entries:
                self.entries.move_to_end(id(image))

    def insert(self, image: 'Image', nbytes: int, dev: torch.device) -> None:
        with self.lock:
            self.entries[id(image)] = (image, nbytes)
            self.entries.move_to_end(id(image))
            self.total += nbytes
            while self.total > self.budget(dev) and len(self.entries) > 1:
                _, (old, n) = self.entries.popitem(last=False)
                old._device_view = None     # benign race with a reader: it holds its own reference or re-decodes
                self.total -= n

    def grow(self, image: 'Image', extra: int, dev: torch.device) -> None:
        """More bytes now hang on ``image``'s entry (the float32 colour twin of its view, 12 B/pixel)."""
        with self.lock:
            ent = self.entries.get(id(image))
            if ent is None:
                return
            self.entries[id(image)] = (image, ent[1] + extra)
            self.total += extra
            while self.total > self.budget(dev) and len(self.entries) > 1:
                key, (old, n) = next(iter(self.entries.items()))
                if old is image:
                    break
                del self.entries[key]
                old._device_view = None
                self.total -= n

    def forget(self, image: 'Image') -> None:
        with self.lock:
            ent = self.entries.pop(id(image), None)
            if ent is not None:
                self.total -= ent[1]


PIXEL_CACHE = _PixelCache()


_STACKED: dict = {}


def _stacked_cameras(image_list):
    """float64 stacks (R, t, K, W, H) of the images' poses and cameras, kept for the list that was asked about last (a
    survey asks about the same list for every target)."""
    key = (len(image_list), id(image_list[0]), id(image_list[-1]), id(image_list[len(image_list) // 2]))
    hit = _STACKED.get('entry')
    if hit is None or hit[0] != key:
        R = np.stack([im.pose.R.numpy() for im in image_list]).astype(np.float64)              # (N, 3, 3) world-from-camera
        t = np.stack([im.pose.t.numpy() for im in image_list]).astype(np.float64)              # (N, 3, 1)
        K = np.stack([im.camera.K.numpy() for im in image_list]).astype(np.float64)
        W2 = np.array([im.camera.width for im in image_list], np.float64).reshape(-1, 1)
        H2 = np.array([im.camera.height for im in image_list], np.float64).reshape(-1, 1)
        hit = (key, (R, t, K, W2, H2))
        _STACKED['entry'] = hit
    return hit[1]


class Image:
    def __init__(self, image_id: int, rgb_path: Path, depth_map_path: Path, pose: Pose, camera: Camera):
        self.id = image_id
        self.name = str(Path(rgb_path).name)
        self.rgb_path = rgb_path
        self.depth_map_path = depth_map_path
        self.pose = pose          # world-from-camera
        self.camera = camera
        self._device_view = None  # (device, engine.DeviceView) cache: pixels stay resident in HBM
        self._device_lock = threading.Lock()   # decode threads and in-flight slots may ask for the same image

    # -- geometry helpers (plain tensor math; used by the output stage and the compatibility shims) -------------
    def unproject_depth(self, u: Tensor, v: Tensor, d: Tensor) -> Tensor:
        """Pixel centres ``(u+0.5, v+0.5)`` at depth ``d`` -> camera-frame points (3, n)."""
        rays = torch.stack([u + 0.5, v + 0.5, torch.ones_like(u)])
        return self.camera.K.inverse().to(rays.device) @ (d * rays)

    def unproject_depth_map(self, depth_map: Tensor, to_world: bool = True) -> tuple[Tensor, Tensor, Tensor]:
        """Valid pixels (depth > 0) of a depth map -> ``(u, v, points)`` in the camera or world frame."""
        v, u = torch.where(depth_map > 0)
        P = self.unproject_depth(u=u, v=v, d=depth_map[v, u])
        return (u, v, self.pose.transform(P)) if to_world else (u, v, P)

    def project_to_view(self, wP: Tensor) -> Tensor:
        """World points (3, n) -> continuous pixel coordinates (2, n) in this image."""
        cp = self.camera.K.to(wP.device) @ self.pose.inverse().transform(wP)
        return cp[:2] / cp[2]

    # -- pixel access ---------------------------------------------------------------------------------------------
    def get_rgb(self) -> Tensor:
        return loader.load_rgb(self.rgb_path, width=self.camera.width, height=self.camera.height)

    def get_depth_map(self) -> Tensor:
        return loader.load_depth_map(self.depth_map_path, width=self.camera.width, height=self.camera.height)

    def device_view(self, device):
        """This image as the engine sees it: uint8 colour + float32 depth resident on ``device`` (cached), and
        the float32 camera/pose matrices.  The files are decoded to their stored integers, uploaded as such
        (7 instead of 20 bytes per pixel over PCIe) and the depth is converted on the GPU with the reference's
        arithmetic, ``float32(float64(mm) / 1000)`` (loader.py:167-170)."""
        from . import engine
        dev = _canonical_device(device)
        with self._device_lock:
            cached = self._device_view
            if cached is not None and cached[0] == dev:
                PIXEL_CACHE.touch(self)
                return cached[1]
            PIXEL_CACHE.forget(self)
            if type(self).get_rgb is not Image.get_rgb or type(self).get_depth_map is not Image.get_depth_map:
                # a subclass supplies the pixels itself (the documented way to feed images that are not files): take
                # them through the same accessors the reference calls (sfm.py:109-113)
                rgb = self.get_rgb().to(torch.float32)
                k255 = rgb.to(torch.float64) * 255
                if bool(((k255 - k255.round()).abs() < 1e-3).all()):
                    rgb = k255.round().to(torch.uint8)          # images as stored: exactly k/255
                depth = self.get_depth_map().to(torch.float32).to(dev).contiguous()
                mm = None
            else:
                rgb = loader.load_rgb_u8(self.rgb_path, width=self.camera.width, height=self.camera.height)
                if rgb is None:   # --image-scale: colours resized in float64 like the reference -> float32 observations
                    rgb = self.get_rgb()
                mm = loader.load_depth_raw(self.depth_map_path, width=self.camera.width, height=self.camera.height)
                if mm is None or dev.type != 'cuda':   # not a 16-bit file: host conversion as the reference does it
                    depth = self.get_depth_map().to(dev).contiguous()
                else:
                    depth = (mm.to(dev).to(torch.float64) / 1000).to(torch.float32).contiguous()
                    # the valid depth range (overlapping_views) from the stored integers, here on the decode thread: the
                    # conversion is monotone, so these are the extremes of `depth` itself -- and the thread that drives
                    # the GPU does not have to launch (and, the first time, load) three torch kernels and wait for them
                    pos = mm[mm > 0]
                    self.__dict__['_depth_range'] = ((float(np.float32(np.float64(int(pos.min())) / 1000)),
                                                      float(np.float32(np.float64(int(pos.max())) / 1000))) if pos.numel() else ())
            view = engine.DeviceView(depth=depth, rgb=rgb.to(dev).contiguous(),
                                     K=self.camera.K, R=self.pose.R, t=self.pose.t, name=self.name)
            if dev.type == 'cuda':  # the cache is shared by every stream (engine.in_flight_slot, decode threads):
                torch.cuda.current_stream(dev).synchronize()   # publish it only once the upload has landed
            self._device_view = (dev, view)
            # (+ 8 bytes per pixel for the packed {depth, colour} records a uint8 view gets when it is first matched against)
            packed = view.depth.numel() * 8 if (view.rgb.dtype == torch.uint8 and dev.type == 'cuda') else 0
            PIXEL_CACHE.insert(self, view.depth.numel() * 4 + view.rgb.numel() * view.rgb.element_size() + packed, dev)
            return view

    def release_device(self) -> None:
        PIXEL_CACHE.forget(self)
        self._device_view = None

    # -- matching ---------------------------------------------------------------------------------------------------
    def depth_range(self, device) -> tuple[float, float] | None:
        """(smallest, largest) valid depth of this image, or None when no pixel is valid (cached)."""
        cached = self.__dict__.get('_depth_range')
        if cached is None:
            depth = self.device_view(device).depth
            valid = depth[depth > 0]
            cached = (float(valid.min()), float(valid.max())) if valid.numel() else ()
            self.__dict__['_depth_range'] = cached
        return cached or None

    def overlapping_views(self, image_list: list['Image'], device: str = 'cuda', margin: float = 2.0) -> list[int]:
        """Indices of the images of ``image_list`` that CAN hold a match of this image; the others provably have none.

        A match needs a valid pixel of this image whose world point projects inside the other image
        (``match_one_way``, sfm.py:115-119).  Every such point lies in the convex hull of this camera's view
        frustum cut at its smallest and largest valid depth (``unproject_depth`` scales the ray by the z-depth,
        sfm.py:90-93).  The other image accepts a camera-frame point ``c = K (R^T (P - t))`` iff
        ``-1 < c0/c2 < W`` and ``-1 < c1/c2 < H`` -- four half-spaces meeting in a cone for ``c2 > 0`` and, because
        the reference never tests ``z > 0``, the mirrored cone for ``c2 < 0``.  If all eight hull corners violate one
        and the same half-space, the hull misses that cone.  Evaluated in float64 with ``margin`` pixels of slack, far
        above the float32 rounding of the kernel's own projection, so no view with a match is ever dropped: a dropped
        view has zero matches and fails ``n / (W H) > min_cover`` for every ``min_cover >= 0`` (sfm.py:136) anyway."""
        rng = self.depth_range(device)
        if rng is None or not image_list:
            return list(range(len(image_list)))
        # numpy, not torch: for a model of a few hundred images torch's CPU kernels wake their whole OpenMP team (one
        # thread per CPU of the machine, spinning after every call), which under a container's CPU quota got the process
        # throttled and made this function take 50-140 ms
        W1, H1 = self.camera.width, self.camera.height
        Kinv = np.linalg.inv(self.camera.K.double().numpy())
        px = np.array([[0.0, W1, 0.0, W1], [0.0, 0.0, H1, H1], [1.0, 1.0, 1.0, 1.0]])
        rays = Kinv @ px                                                        # (3, 4): z component is 1
        cP = np.concatenate([rays * rng[0], rays * rng[1]], axis=1)             # (3, 8) hull corners, camera frame
        wP = self.pose.R.double().numpy() @ cP + self.pose.t.double().numpy()   # world frame
        R, t, K, W2, H2 = _stacked_cameras(image_list)
        rel = wP[None] - t                                                      # (N, 3, 8)
        c = np.einsum('nij,njk->nik', K, np.einsum('nji,njk->nik', R, rel))     # K (R^T (P - t))
        c0, c1, c2 = c[:, 0], c[:, 1], c[:, 2]
        a = np.stack([c0 + (1 + margin) * c2, (W2 + margin) * c2 - c0,
                      c1 + (1 + margin) * c2, (H2 + margin) * c2 - c1], axis=1)          # (N, 4, 8); inside the cone: all > 0
        misses_cone = (a <= 0).all(axis=2).any(axis=1)      # some half-space excludes every corner
        misses_mirror = (a >= 0).all(axis=2).any(axis=1)    # inside the mirrored cone: all < 0
        keep = ~(misses_cone & misses_mirror)
        return np.nonzero(keep)[0].tolist()

    def match_one_way(self, other: 'Image', u1: Tensor, v1: Tensor, wP1: Tensor, device: str = 'cuda') -> 'Matches':
        """Pixels ``(u1, v1)`` of this image (world points ``wP1``) that land inside ``other`` (sfm.py:115-119):
        the continuous projection is truncated towards zero and tested against ``other``'s sensor; nothing checks
        that the point is in front of ``other``.  The projection runs on the GPU (``sucre_project_points``: the match
        kernel's own float32 operation order, bit-identical to the reference's torch CPU arithmetic); the lists come
        back on the device the caller's tensors live on (host tensors are staged through ``device``)."""
        from . import engine
        home = wP1.device
        dev = home if home.type == 'cuda' else _canonical_device(device)
        cam = engine.camera_struct(other.camera.K, other.pose.R, other.pose.t, other.camera.height, other.camera.width)
        q = engine.project_points(cam, wP1.to(dev, torch.float32)).long()
        inside = q >= 0
        q = q[inside]
        W2 = other.camera.width
        u2, v2 = (q % W2).to(home), torch.div(q, W2, rounding_mode='floor').to(home)
        inside = inside.to(home)
        return Matches(image1=self, image2=other, u1=u1[inside], v1=v1[inside], u2=u2, v2=v2)

    def match_images(self, image_list: list['Image'], matches_file: 'loader.MatchesFile', min_cover: float = 0.000001,
                     num_workers: int = 0, device: str = 'cpu', light_model: bool = False):
        """Two-way dense matching of this image against every image of ``image_list`` and preparation of the
        observations the fit consumes (replaces sfm.py:127-138 + loader.py:78-118).  One HIP launch matches all
        views; views failing ``n_matches / (W*H) > min_cover`` are dropped on the device; nothing is written to
        disk unless the matches file is asked to persist.  ``light_model`` (not a reference argument) also keeps the
        camera-frame point of every observation, which the artificial-light model needs.  ``device`` defaults to the
        reference's 'cpu' and must be given as a GPU (``require_gpu``)."""
        from . import engine
        require_gpu(device, 'Image.match_images')
        image_list = list(image_list)
        if min_cover >= 0 and len(image_list) > 1 and os.environ.get('SUCRE_CULL_VIEWS', '1') != '0':
            # images whose field of view cannot contain anything this image sees have no match and would be dropped
            # by the min_cover rule: they are not decoded, uploaded or given workspace (overlapping_views)
            image_list = [image_list[i] for i in self.overlapping_views(image_list, device)] or image_list[:1]
        if len(image_list) > engine.MAX_VIEWS:
            raise RuntimeError(f'{self.name}: {len(image_list)} images can overlap it, more than the {engine.MAX_VIEWS} '
                               f'views one restoration holds; list images to leave out in --filter-images-path')
        loader.prefetch_device_views(image_list + [self], device, num_workers=num_workers)
        views = [im.device_view(device) for im in image_list]
        target = self.device_view(device)
        float_colour = any(v.rgb.dtype == torch.float32 for v in views + [target])   # resized inputs (--image-scale)
        if float_colour:   # (with light_model the engine keeps both the camera points and the float32 colours)
            for im, v in zip(image_list + [self], views + [target]):   # the float32 twins count against the cache budget
                if v.rgb.dtype != torch.float32 and '_float_twin' not in v.__dict__:
                    PIXEL_CACHE.grow(im, v.rgb.numel() * 4, _canonical_device(device))
            views = [v.as_float_colour() for v in views]
            target = target.as_float_colour()
        resto = engine.acquire_restoration(self.camera.height, self.camera.width, len(views), device, light=light_model,
                                           float_colour=float_colour)
        resto.match(target, views, min_cover=min_cover)
        matches_file.attach(resto, target_image=self, image_list=image_list)

    def match_two_way(self, other: 'Image', u1: Tensor = None, v1: Tensor = None, wP1: Tensor = None,
                      u2: Tensor = None, v2: Tensor = None, wP2: Tensor = None, device: str = 'cuda') -> 'Matches':
        """Mutually consistent matches between this image and ``other`` (sfm.py:121-125) as explicit lists.

        With the reference's arguments -- pixels ``(u1, v1)`` of this image with their world points ``wP1`` and
        ``(u2, v2, wP2)`` of ``other``, positionally or by keyword -- this is the reference's own composition on the
        caller's point sets: ``self.match_one_way(other, ...) & other.match_one_way(self, ...)``, so a caller that
        passes a subset of the pixels gets the matches of that subset.  Without them (not a reference call form) both
        images' full valid sets are matched by the fused kernel from the depth maps on the device -- the same
        match set the reference computes from ``unproject_depth_map`` of both images (sfm.py:129-135)."""
        given = [a is not None for a in (u1, v1, wP1, u2, v2, wP2)]
        if any(given):
            if not all(given):
                raise TypeError('match_two_way() takes (other, u1, v1, wP1, u2, v2, wP2) like the reference (sfm.py:121), '
                                'or (other) alone')
            return (self.match_one_way(other, u1=u1, v1=v1, wP1=wP1, device=device)
                    & other.match_one_way(self, u1=u2, v1=v2, wP1=wP2, device=device))
        from . import engine
        views = [other.device_view(device)]
        target = self.device_view(device)
        float_colour = views[0].rgb.dtype == torch.float32 or target.rgb.dtype == torch.float32
        if float_colour:
            views, target = [views[0].as_float_colour()], target.as_float_colour()
        resto = engine.acquire_restoration(self.camera.height, self.camera.width, 1, device, float_colour=float_colour, tag='pair')
        resto.match(target, views, min_cover=-1.0)
        q = resto.match_map(0)
        v1, u1 = torch.where(q >= 0)
        p2 = q[v1, u1].long()
        W2 = other.camera.width
        return Matches(image1=self, image2=other, u1=u1, v1=v1, u2=p2 % W2, v2=torch.div(p2, W2, rounding_mode='floor'))

    def __repr__(self) -> str:
        return f'SfMImage({self.name!r})'


class Matches:
    def __init__(self, image1: Image, image2: Image, u1: Tensor, v1: Tensor, u2: Tensor, v2: Tensor):
        self.image1, self.image2 = image1, image2
        self.u1, self.v1, self.u2, self.v2 = u1, v1, u2, v2

    def map(self) -> Tensor:
        """Dense (H1, W1, 2) map holding ``(v2, u2)`` at every matched ``(v1, u1)`` and -1 elsewhere."""
        out = torch.full((self.image1.camera.height, self.image1.camera.width, 2), -1,
                         device=self.u1.device, dtype=self.u1.dtype)
        out[self.v1, self.u1, 0] = self.v2
        out[self.v1, self.u1, 1] = self.u2
        return out

    def __and__(self, other: 'Matches') -> 'Matches':
        """Keeps the matches of ``self`` whose reverse match in ``other`` points back to the same pixel."""
        back = other.map()[self.v2, self.u2]
        keep = (back[:, 0] == self.v1) & (back[:, 1] == self.u1)
        return Matches(self.image1, self.image2, self.u1[keep], self.v1[keep], self.u2[keep], self.v2[keep])

    def plot(self, step: int = 10000, color: tuple[float, float, float] = None):
        """Debug picture (sfm.py:161-169): the two images side by side, every ``step``-th match joined by a line
        (random colours unless ``color`` is given)."""
        from PIL import Image as PILImage, ImageDraw
        left, right = self.image1.get_rgb(), self.image2.get_rgb()
        canvas = np.zeros((max(left.shape[0], right.shape[0]), left.shape[1] + right.shape[1], 3), np.uint8)
        canvas[:left.shape[0], :left.shape[1]] = np.uint8(left.numpy() * 255)
        canvas[:right.shape[0], left.shape[1]:] = np.uint8(right.numpy() * 255)
        picture = PILImage.fromarray(canvas)
        pen = ImageDraw.Draw(picture)
        rng = np.random.default_rng()
        ends = torch.stack([self.u1, self.v1, self.u2 + left.shape[1], self.v2], dim=1)[::step].cpu().tolist()
        for x1, y1, x2, y2 in ends:
            pen.line([(x1, y1), (x2, y2)], fill=tuple(int(c) for c in rng.integers(0, 256, 3)) if color is None else color, width=3)
        return picture

    def __len__(self) -> int:
        return int(self.u1.shape[0])

    def __repr__(self) -> str:
        return f'Matches(image1={self.image1!r}, image2={self.image2!r}, {len(self)} matches)'


# ---- COLMAP model -------------------------------------------------------------------------------------------------

_PINHOLE_MODEL_ID = 1
_COLMAP_NUM_PARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8, 5: 8, 6: 12, 7: 5, 8: 4, 9: 5, 10: 12}


def quat_to_rotmat(qw: float, qx: float, qy: float, qz: float) -> np.ndarray:
    """Quaternion (COLMAP order: w first) -> 3x3 rotation in float64.  COLMAP normalises the quaternion when it reads
    ``images.txt`` / ``images.bin`` (a zero quaternion becomes the identity), so the input need not be a unit one;
    the matrix is then built in the operation order of Eigen's ``toRotationMatrix``, which is what pycolmap's
    ``cam_from_world.rotation.matrix()`` evaluates (sfm.py:221).  Pinned against scipy's
    ``Rotation.from_quat([x, y, z, w])`` in tests/test_host_logic.py."""
    n = float(np.sqrt(qw * qw + qx * qx + qy * qy + qz * qz))
    if n == 0.0:
        qw, qx, qy, qz = 1.0, 0.0, 0.0, 0.0
    else:
        qw, qx, qy, qz = qw / n, qx / n, qy / n, qz / n
    tx, ty, tz = 2.0 * qx, 2.0 * qy, 2.0 * qz
    twx, twy, twz = tx * qw, ty * qw, tz * qw
    txx, txy, txz = tx * qx, ty * qx, tz * qx
    tyy, tyz, tzz = ty * qy, tz * qy, tz * qz
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1.0 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1.0 - (txx + tyy)]], dtype=np.float64)


def _read_cameras(model_dir: Path) -> dict[int, tuple[str, int, int, list[float]]]:
    cams = {}
    if (model_dir / 'cameras.bin').exists():
        data = (model_dir / 'cameras.bin').read_bytes()
        (n,), off = struct.unpack_from('<Q', data, 0), 8
        for _ in range(n):
            cam_id, model_id, w, h = struct.unpack_from('<iiQQ', data, off)
            off += 24
            npar = _COLMAP_NUM_PARAMS[model_id]
            params = list(struct.unpack_from(f'<{npar}d', data, off))
            off += 8 * npar
            cams[cam_id] = ('PINHOLE' if model_id == _PINHOLE_MODEL_ID else f'MODEL_{model_id}', w, h, params)
    else:
        for line in (model_dir / 'cameras.txt').read_text().splitlines():
            line = line.strip()
            if not line or line.startswith('#'):
                continue
            tok = line.split()
            cams[int(tok[0])] = (tok[1], int(tok[2]), int(tok[3]), [float(x) for x in tok[4:]])
    return cams


def _read_images(model_dir: Path) -> dict[int, tuple[tuple[float, ...], tuple[float, ...], int, str]]:
    images = {}
    if (model_dir / 'images.bin').exists():
        data = (model_dir / 'images.bin').read_bytes()
        (n,), off = struct.unpack_from('<Q', data, 0), 8
        for _ in range(n):
            image_id, = struct.unpack_from('<I', data, off)
            q = struct.unpack_from('<4d', data, off + 4)
            t = struct.unpack_from('<3d', data, off + 36)
            cam_id, = struct.unpack_from('<I', data, off + 60)
            off += 64
            end = data.index(b'\x00', off)
            name = data[off:end].decode()
            off = end + 1
            (n2d,) = struct.unpack_from('<Q', data, off)
            off += 8 + 24 * n2d
            images[image_id] = (q, t, cam_id, name)
    else:
        # COLMAP's text reader: blank lines and lines starting with '#' are skipped wherever a header line is expected;
        # the line after a header is that image's POINTS2D list (taken as it is, even when empty); the name is the rest
        # of the header line after the ninth blank, so it may contain blanks itself
        lines = (model_dir / 'images.txt').read_text().splitlines()
        i = 0
        while i < len(lines):
            head = lines[i].strip()
            i += 1
            if not head or head.startswith('#'):
                continue
            tok = head.split(' ', 9)
            assert len(tok) == 10, f'images.txt: malformed image line {head!r}'
            images[int(tok[0])] = (tuple(float(x) for x in tok[1:5]), tuple(float(x) for x in tok[5:8]), int(tok[8]),
                                   tok[9].strip())
            i += 1  # the POINTS2D line
    return images


class COLMAPModel:
    def __init__(self, model_dir: Path, image_dir: Path, depth_dir: Path, image_scale: float = 1.0):
        """Loads an undistorted COLMAP model (sfm.py:186-238): PINHOLE cameras (intrinsics rescaled by
        ``image_scale``), images with world-from-camera poses, depth maps named ``depth_<stem>.png``."""
        model_dir, image_dir, depth_dir = Path(model_dir), Path(image_dir), Path(depth_dir)
        self.image_scale = image_scale
        self.cameras: dict[int, Camera] = {}
        for cam_id, (model, w, h, params) in sorted(_read_cameras(model_dir).items()):
            assert model == 'PINHOLE', f'Camera {cam_id} ({model}) is not using the PINHOLE model.'
            width, height = int(w * image_scale), int(h * image_scale)
            sw, sh = width / w, height / h
            fx, fy, cx, cy = params
            K = torch.tensor([[fx * sw, 0, cx * sw], [0, fy * sh, cy * sh], [0, 0, 1]], dtype=torch.float32)
            self.cameras[cam_id] = Camera(camera_id=cam_id, width=width, height=height, K=K)

        self.images: dict[int, Image] = {}
        for image_id, (q, t, cam_id, name) in sorted(_read_images(model_dir).items()):
            rgb_path = image_dir / name
            depth_path = (depth_dir / name).with_name('depth_' + rgb_path.stem + '.png')
            cam_from_world = Pose(R=torch.tensor(quat_to_rotmat(*q), dtype=torch.float32),
                                  t=torch.tensor(t, dtype=torch.float32).view(3, 1))
            self.images[image_id] = Image(image_id=image_id, rgb_path=rgb_path, depth_map_path=depth_path,
                                          pose=cam_from_world.inverse(), camera=self.cameras[cam_id])
        self.imagename2id = {im.name: im.id for im in self.images.values()}

    def __getitem__(self, image_name: str) -> Image:
        return self.images[self.imagename2id[image_name]]

    def __repr__(self) -> str:
        return f'COLMAPModel({len(self.images)} images)'


def write_colmap_text(model_dir: Path, K: Tensor, width: int, height: int, names: list[str], poses: list[Pose]) -> None:
    """Writes ``cameras.txt`` / ``images.txt`` / ``points3D.txt`` for a single-camera scene whose images have the
    given world-from-camera poses (used by the synthetic-scene tooling and the CLI tests)."""
    model_dir = Path(model_dir)
    model_dir.mkdir(parents=True, exist_ok=True)
    (model_dir / 'cameras.txt').write_text(
        '# Camera list with one line of data per camera:\n'
        f'1 PINHOLE {width} {height} {float(K[0, 0])!r} {float(K[1, 1])!r} {float(K[0, 2])!r} {float(K[1, 2])!r}\n')
    lines = ['# Image list with two lines of data per image:']
    for i, (name, pose) in enumerate(zip(names, poses), start=1):
        cfw = Pose(pose.R.double(), pose.t.double()).inverse()
        q = rotmat_to_quat(cfw.R.numpy())
        t = cfw.t.numpy().ravel()
        lines.append(f'{i} ' + ' '.join(repr(float(x)) for x in (*q, *t)) + f' 1 {name}')
        lines.append('')
    (model_dir / 'images.txt').write_text('\n'.join(lines) + '\n')
    (model_dir / 'points3D.txt').write_text('# 3D point list (empty)\n')


def rotmat_to_quat(R: np.ndarray) -> np.ndarray:
    """3x3 rotation -> unit quaternion (w, x, y, z), w >= 0."""
    R = np.asarray(R, dtype=np.float64)
    tr = np.trace(R)
    if tr > 0:
        s = 2.0 * np.sqrt(1.0 + tr)
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k])
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = q / np.linalg.norm(q)
    return q if q[0] >= 0 else -q
