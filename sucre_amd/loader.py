"""Data side of the drop-in: pixel loaders and the matches containers.

Mirrors the reference's ``loader`` module (loader.py:33-180): ``MatchesSample``, ``MatchesData``, ``MatchesFile``,
``ImageDataset``, ``load_rgb``, ``load_depth_map``, ``load_image_list`` keep their names and call signatures.
What changes is where the observations live.  The reference spills every match to an HDF5 file (24 B/obs),
re-reads it three times and keeps 28 B/obs on the host, re-uploading everything each Adam iteration
(loader.py:46-49, 68-118).  Here ``MatchesFile`` is a handle on the engine's observation store in HBM
(7 B/obs, written once by the match kernel); ``prepare_matches`` / ``check_integrity`` / ``load_matches`` keep their
meaning but move no data, and ``MatchesData`` wraps the same store.  ``MatchesData.iter`` still yields the
reference's ``(u, v, cP, I)`` batches for code written against it.
"""
from __future__ import annotations

from collections import namedtuple
from concurrent.futures import ThreadPoolExecutor
import os
from pathlib import Path

import numpy as np
import torch
from torch import Tensor
from torch.utils.data import DataLoader, Dataset

from . import _pixelio, h5bridge

MatchesSample = namedtuple('MatchesSample', ['u', 'v', 'cP', 'I'])


# ---- pixel loaders ----------------------------------------------------------------------------------------------------

def _decode_pool():
    """The worker processes, when the CLI has started some (image files are decoded there: PIL's decoder scales to only
    ~4 threads' worth over 16 threads; SUCRE_DECODE_IN_WORKERS=0 keeps decoding in this process's threads)."""
    return _pixelio.POOL if os.environ.get('SUCRE_DECODE_IN_WORKERS', '1') != '0' else None


def _imread_rgb_u8(path: Path) -> np.ndarray:
    """The colour image as stored (loader.py:157)."""
    pool = _decode_pool()
    if pool is not None:
        try:
            return pool.read(path)
        except _pixelio.WorkerLost:
            pass   # the worker died mid-request: decode here instead
    return _pixelio.imread_rgb_u8(path)


def _imread_depth_u16(path: Path) -> np.ndarray:
    pool = _decode_pool()
    if pool is not None:
        try:
            return pool.read(path, depth=True)
        except _pixelio.WorkerLost:
            pass
    return _pixelio.imread_depth_u16(path)


def load_rgb_u8(rgb_path: Path, width: int, height: int) -> Tensor | None:
    """(H,W,3) uint8 colour image: the engine's compact input format (source pixels are exactly k/255).  None when the
    file is not camera-sized: resized colours (--image-scale) are not multiples of 1/255, use ``load_rgb``."""
    rgb = _imread_rgb_u8(Path(rgb_path))
    if rgb.shape[0] != height or rgb.shape[1] != width:
        return None
    return torch.from_numpy(np.array(rgb, dtype=np.uint8, order="C"))


def _area_taps(ssize: int, dsize: int) -> tuple[np.ndarray, np.ndarray]:
    """Source indices and weights of OpenCV's INTER_AREA along one axis for a non-integer shrink factor (its
    ``computeResizeAreaTab``): destination cell ``d`` covers ``[d*scale, (d+1)*scale)`` of the source, whole source
    cells weigh ``1/cell``, the two partly covered ones their covered fraction (dropped below 1e-3), the weights are
    rounded to float32 as OpenCV stores them.  Returns ``(index, weight)`` of shape (dsize, taps), padded with weight 0."""
    scale = 1.0 / (dsize / ssize)
    rows = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        taps = []
        if s1 - f1 > 1e-3:
            taps.append((s1 - 1, np.float32((s1 - f1) / cell)))
        taps += [(sx, np.float32(1.0 / cell)) for sx in range(s1, s2)]
        if f2 - s2 > 1e-3:
            taps.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
        rows.append(taps)
    n = max(len(t) for t in rows)
    idx = np.zeros((dsize, n), np.int64)
    wgt = np.zeros((dsize, n), np.float64)
    for d, taps in enumerate(rows):
        for k, (sx, a) in enumerate(taps):
            idx[d, k], wgt[d, k] = sx, float(a)
    return idx, wgt


def _resize_area(rgb: np.ndarray, width: int, height: int) -> np.ndarray:
    """INTER_AREA of a float64 image for any shrink factor: columns reduced first, then rows, every sum taken in
    source order in float64 -- the order of OpenCV's ``ResizeArea_Invoker`` for a double image."""
    ix, wx = _area_taps(rgb.shape[1], width)
    iy, wy = _area_taps(rgb.shape[0], height)
    cols = np.zeros((rgb.shape[0], width, rgb.shape[2]), np.float64)
    for k in range(ix.shape[1]):
        cols += rgb[:, ix[:, k]] * wx[None, :, k, None]
    out = np.zeros((height, width, rgb.shape[2]), np.float64)
    for k in range(iy.shape[1]):
        out += cols[iy[:, k]] * wy[:, k, None, None]
    return out


def _cubic_taps(ssize: int, dsize: int) -> tuple[np.ndarray, np.ndarray]:
    """Source indices and weights of OpenCV's INTER_CUBIC along one axis (its generic resize path): destination ``d``
    samples the source at ``(d + 0.5) * scale - 0.5`` with the four-tap cubic kernel of parameter A = -0.75, position
    and weights in float32 as OpenCV computes them, taps outside the image clamped to the border pixel.  Returns
    ``(index, weight)`` of shape (dsize, 4)."""
    scale = 1.0 / (dsize / ssize)
    A = np.float32(-0.75)
    idx = np.zeros((dsize, 4), np.int64)
    wgt = np.zeros((dsize, 4), np.float64)
    one = np.float32(1.0)
    for d in range(dsize):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(np.floor(fx))
        x = np.float32(fx - np.float32(sx))
        c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
        c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
        c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
        c3 = one - c0 - c1 - c2
        for j, c in enumerate((c0, c1, c2, c3)):
            idx[d, j] = min(max(sx - 1 + j, 0), ssize - 1)
            wgt[d, j] = float(np.float32(c))
    return idx, wgt


def _resize_cubic(rgb: np.ndarray, width: int, height: int) -> np.ndarray:
    """INTER_CUBIC of a float64 image (what the reference's loader uses when it enlarges, loader.py:158-162): rows
    filtered horizontally first, then vertically, every four-tap sum taken left to right in float64."""
    ix, wx = _cubic_taps(rgb.shape[1], width)
    iy, wy = _cubic_taps(rgb.shape[0], height)
    cols = rgb[:, ix[:, 0]] * wx[None, :, 0, None]
    for k in range(1, 4):
        cols = cols + rgb[:, ix[:, k]] * wx[None, :, k, None]
    out = cols[iy[:, 0]] * wy[:, 0, None, None]
    for k in range(1, 4):
        out = out + cols[iy[:, k]] * wy[:, k, None, None]
    return out


def _resize_rgb(rgb: np.ndarray, width: int, height: int) -> np.ndarray:
    """The reference's resize of the float64 colour image (loader.py:158-162): OpenCV INTER_AREA when shrinking,
    INTER_CUBIC otherwise.  OpenCV is used when it is installed.  Without it, shrinking is restated here from
    OpenCV's published algorithm: integer factors (--image-scale 0.5 / 0.25) the way its fast path does them (the
    block's pixels summed in row-major order, times 1/area), any other factor by `_resize_area`.  The restatement
    could not be checked against OpenCV in the build image (no cv2 there): install opencv-python where bit-identical
    --image-scale output matters.  Enlarging (INTER_CUBIC, `_resize_cubic`) is restated the same way."""
    try:
        import cv2
        return cv2.resize(rgb, (width, height), interpolation=cv2.INTER_AREA if width < rgb.shape[1] else cv2.INTER_CUBIC)
    except ImportError:
        pass
    H0, W0 = rgb.shape[:2]
    if width < W0 and W0 % width == 0 and H0 % height == 0:
        fx, fy = W0 // width, H0 // height
        acc = np.zeros((height, width, rgb.shape[2]), np.float64)
        for dy in range(fy):
            for dx in range(fx):
                acc += rgb[dy::fy, dx::fx]
        return acc * float(np.float32(1.0) / np.float32(fx * fy))   # OpenCV keeps 1/area in float32
    if width < W0 and height <= H0:
        return _resize_area(rgb, width, height)
    if width < W0:   # (narrower but taller: OpenCV would still take INTER_AREA, whose tap table does not enlarge)
        raise NotImplementedError(f'resizing {W0}x{H0} to {width}x{height} needs OpenCV, which is not installed')
    return _resize_cubic(rgb, width, height)


def load_rgb(rgb_path: Path, width: int, height: int) -> Tensor:
    """(H,W,3) float32 colour in [0,1]: ``uint8 / 255`` evaluated in float64 then cast (loader.py:156-163)."""
    rgb = _imread_rgb_u8(Path(rgb_path)) / 255
    if rgb.shape[0] != height or rgb.shape[1] != width:
        rgb = _resize_rgb(rgb, width, height)
    return torch.tensor(rgb, dtype=torch.float32)


def load_depth_map(depth_map_path: Path, width: int, height: int) -> Tensor:
    """(H,W) float32 metres from a 16-bit millimetre PNG: ``uint16 / 1000`` in float64 then cast
    (loader.py:166-170); nearest-neighbour resize when the camera size differs."""
    depth = _imread_depth_u16(Path(depth_map_path)) / 1000
    if depth.shape[0] != height or depth.shape[1] != width:
        rows = (np.arange(height) * (depth.shape[0] / height)).astype(np.int64).clip(0, depth.shape[0] - 1)
        cols = (np.arange(width) * (depth.shape[1] / width)).astype(np.int64).clip(0, depth.shape[1] - 1)
        depth = depth[rows][:, cols]  # cv2.INTER_NEAREST picks floor(dst * scale)
    return torch.tensor(depth, dtype=torch.float32)


def load_depth_raw(depth_map_path: Path, width: int, height: int) -> Tensor | None:
    """(H,W) int32 millimetres exactly as stored in a 16-bit depth PNG (nearest-neighbour resized like
    ``load_depth_map``), for conversion on the GPU; None when the file is not an unsigned 16-bit image."""
    depth = _imread_depth_u16(Path(depth_map_path))
    if depth.dtype != np.uint16 or depth.ndim != 2:
        return None
    if depth.shape[0] != height or depth.shape[1] != width:
        rows = (np.arange(height) * (depth.shape[0] / height)).astype(np.int64).clip(0, depth.shape[0] - 1)
        cols = (np.arange(width) * (depth.shape[1] / width)).astype(np.int64).clip(0, depth.shape[1] - 1)
        depth = depth[rows][:, cols]
    return torch.from_numpy(depth.astype(np.int32))


def effective_cpus() -> int:
    """CPUs this process may actually use: the scheduler affinity capped by the cgroup CPU quota.  In a container
    ``os.cpu_count()`` reports the machine (256 on the MI355X boxes) while the quota may be 16 CPUs; thread pools sized
    by the former get the whole process throttled -- the thread that feeds the GPU included (measured: the CLI's
    per-image rate fell from 22 to 35 ms with 32 PNG-encoding threads under a 16-CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:   # cgroup v2
        quota, period = Path('/sys/fs/cgroup/cpu.max').read_text().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:   # cgroup v1
            quota = int(Path('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read_text())
            period = int(Path('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read_text())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def decode_threads(num_workers: int = 0) -> int:
    """Threads used to decode image files: ``--num-workers`` when given, else SUCRE_DECODE_THREADS (default: the
    CPUs this process may use, at most 16).  PNG/JPEG decoding releases the GIL, and the result does not depend on
    who decoded it."""
    return int(num_workers) if num_workers else max(1, int(os.environ.get('SUCRE_DECODE_THREADS', min(16, effective_cpus()))))


_PREFETCH_POOL: ThreadPoolExecutor | None = None


def prefetch_device_views(images, device, num_workers: int = 0, background: bool = False) -> None:
    """Decodes and uploads the pixels of every image not yet resident on ``device`` (replaces the reference's
    DataLoader prefetch, loader.py:173-180).  ``background=True`` returns at once: the decode threads keep filling
    the per-image caches while the caller matches and fits, and ``Image.device_view`` blocks only on an image
    that is still being decoded (per-image lock)."""
    global _PREFETCH_POOL
    from .sfm import _canonical_device
    dev = _canonical_device(device)
    def missing(im) -> bool:
        cached = im._device_view   # read once: the pixel cache may drop the entry (set it to None) concurrently
        return cached is None or cached[0] != dev
    todo = [im for im in {id(i): i for i in images}.values() if missing(im)]
    if not todo:
        return
    n = decode_threads(num_workers)
    if background:
        from .sfm import PIXEL_CACHE
        if _PREFETCH_POOL is None:
            _PREFETCH_POOL = ThreadPoolExecutor(max_workers=n, thread_name_prefix='sucre-decode')

        def fetch(im):
            if not PIXEL_CACHE.full(dev):   # a scene larger than the cache budget is decoded on demand instead
                im.device_view(device)
        for im in todo:
            _PREFETCH_POOL.submit(fetch, im)
    elif n > 1 and len(todo) > 1:
        with ThreadPoolExecutor(max_workers=n) as pool:
            list(pool.map(lambda im: im.device_view(device), todo))
    else:
        for im in todo:
            im.device_view(device)


_PLAN_POOL: ThreadPoolExecutor | None = None


def prefetch_for_targets(targets, image_list, device, num_workers: int = 0, min_cover: float = 0.0) -> None:
    """Background decode + upload of what the coming restorations will read, in the order they will read it: every
    target and the images of ``image_list`` that can overlap it (``sfm.Image.overlapping_views``; all of them when the
    cull is off).  Returns at once; ``Image.device_view`` blocks only on an image that is still being decoded."""
    global _PLAN_POOL
    if _PLAN_POOL is None:
        _PLAN_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix='sucre-plan')
    cull = min_cover >= 0 and os.environ.get('SUCRE_CULL_VIEWS', '1') != '0'
    image_list = list(image_list)

    targets = list(targets)

    def plan(i):
        from .sfm import PIXEL_CACHE, _canonical_device
        if PIXEL_CACHE.full(_canonical_device(device)):
            return   # a scene larger than the cache budget is decoded on demand instead
        # which images can overlap a target follows from the target's own depth map, so the targets' pixels come first:
        # the next few are decoded side by side by the pool (one after the other in this thread, a 512-image model cost
        # 180 ms per target in decoding alone), this thread then only waits for the one it needs
        prefetch_device_views(targets[i:i + 8], device, num_workers=num_workers, background=True)
        target = targets[i]
        target.device_view(device)
        idx = target.overlapping_views(image_list, device) if cull else range(len(image_list))
        prefetch_device_views([image_list[j] for j in idx], device, num_workers=num_workers, background=True)
    for i in range(len(targets)):
        _PLAN_POOL.submit(plan, i)


class ImageDataset(Dataset):
    """Streams ``(idx, rgb)``, ``(idx, depth)`` or ``(idx, rgb, depth)`` like the reference (loader.py:133-153)."""

    def __init__(self, image_list, return_rgb: bool = True, return_depth_map: bool = True):
        self.images = image_list
        self.return_rgb = return_rgb
        self.return_depth_map = return_depth_map

    def __len__(self) -> int:
        return len(self.images)

    def __getitem__(self, idx):
        im = self.images[idx]
        out = [idx]
        if self.return_rgb:
            out.append(im.get_rgb())
        if self.return_depth_map:
            out.append(im.get_depth_map())
        return tuple(out) if len(out) > 1 else None


def _first(batch):
    return batch[0]


def load_image_list(image_list, return_rgb: bool = True, return_depth_map: bool = True, num_workers: int = 0):
    return DataLoader(ImageDataset(image_list, return_rgb=return_rgb, return_depth_map=return_depth_map),
                      num_workers=num_workers, collate_fn=_first)


# ---- matches ----------------------------------------------------------------------------------------------------------

class MatchesData:
    """Observations of one target image.

    Two flavours share the interface of the reference's class (loader.py:36-53):
    * engine-backed (``MatchesData(restoration=...)``, what ``MatchesFile.load_matches`` returns): the data are
      the HBM observation store; ``sucre.adam`` hands it to the HIP fit directly.
    * list-backed (``MatchesData()`` + ``append``, exactly what the reference's ``load_matches`` builds,
      loader.py:103-118): tensors in the reference's format; ``iter`` works on them as they are, and ``sucre.adam`` /
      ``SUCRe.update_J`` import them into an engine workspace on first use (``to_engine``).
    """

    def __init__(self, restoration=None, image_list=None):
        self.data: list[MatchesSample] = []
        self.restoration = restoration
        self.image_list = image_list

    def append(self, u: Tensor, v: Tensor, cP: Tensor, I: Tensor):
        self.data.append(MatchesSample(u=u, v=v, cP=cP, I=I))

    def _materialise(self) -> list[MatchesSample]:
        """Kept views of the observation store in the reference's sample format, groups in name order
        (h5py iterates groups alphabetically).  ``cP`` is the camera-frame point of loader.py:113: read back from the
        extension planes when the restoration keeps them (light model), else recomputed exactly as the reference
        does, ``unproject_depth(u2, v2, depth2[v2, u2])``, from the explicit correspondences (``match_map``) while the
        matched views are at hand; for a store filled from lists that carried no points, ``(0, 0, ||cP||)`` -- all
        that ``SUCRe.forward`` reads without the light model (sucre.py:53)."""
        r = self.restoration
        keep = r.view_keep().cpu().numpy().astype(bool)
        order = sorted(range(r.n_views), key=lambda k: self.image_list[k].name if self.image_list else k)
        out = []
        for k in order:
            if not keep[k]:
                continue
            z, rgb = r.export_view(k)
            v, u = torch.where(z > 0)
            zz = z[v, u]
            if r.light:
                cP = r.export_view_ext(k)[:, v, u].contiguous()
            elif r._views_dev is not None and self.image_list:
                other = self.image_list[k]
                p2 = r.match_map(k)[v, u].long()
                W2 = other.camera.width
                u2, v2 = p2 % W2, torch.div(p2, W2, rounding_mode='floor')
                cP = other.unproject_depth(u=u2, v=v2, d=other.device_view(r.device).depth[v2, u2])
            else:
                cP = torch.stack([torch.zeros_like(zz), torch.zeros_like(zz), zz])
            if r.float_colour and r.light:   # the colours sit in the second extension set: gathered from the view itself
                other = self.image_list[k]
                p2 = r.match_map(k)[v, u].long()
                W2 = other.camera.width
                rgbf = other.device_view(r.device).as_float_colour().rgb
                I = rgbf[torch.div(p2, W2, rounding_mode='floor'), p2 % W2].T.contiguous()
            elif r.float_colour:
                I = r.export_view_ext(k)[:, v, u].contiguous()
            else:
                I = (rgb[v, u].to(torch.float64) / 255).to(torch.float32).T.contiguous()
            out.append(MatchesSample(u=u.short(), v=v.short(), cP=cP, I=I))
        return out

    def to_engine(self, height: int, width: int, device='cuda', light: bool = False):
        """The engine workspace holding these observations; for a list-backed container the samples are imported
        once (``sucre_import_view[_ext]``): ranges ``z = ||cP||`` in the match kernel's float32 operation order,
        colours as uint8 when every ``I * 255`` is an integer (images as stored) and as float32 otherwise (resized
        images), camera points along when ``light``.  All samples are kept (they passed ``min_cover`` upstream)."""
        if self.restoration is not None:
            return self.restoration
        if not self.data:
            raise RuntimeError('this MatchesData holds no observation')
        from . import engine
        lists, integral = [], True
        for s in self.data:
            I = s.I.to(torch.float32)
            k255 = I.to(torch.float64) * 255
            integral = integral and bool(((k255 - k255.round()).abs() < 1e-3).all()) and bool(((k255 >= 0) & (k255 <= 255)).all())
        for s in self.data:
            cP = s.cP.to(torch.float32)
            z = torch.sqrt((cP[0] * cP[0] + cP[1] * cP[1]) + cP[2] * cP[2])
            I = s.I.to(torch.float32)
            rgb = (I.to(torch.float64) * 255).round().to(torch.uint8).T.contiguous() if integral else None
            ext = (torch.cat([cP, I]).contiguous() if (light and not integral)      # both extension sets
                   else cP.contiguous() if light else (None if integral else I.contiguous()))
            lists.append((s.u, s.v, z, rgb) if ext is None else (s.u, s.v, z, rgb, ext))
        # A workspace of its OWN while this container lives (the reference's MatchesData objects are independent of each
        # other: a workspace shared by geometry would hand md_A the observations md_B imported after it), LEASED from the
        # engine's free list and handed back when the container is dropped -- image after image of a kept-matches run
        # reuses one allocation (and its pinned staging buffer) instead of allocating ~2 GB per image.
        import weakref
        # (obs_format spelled out: a list-backed container is the reference's lossless MatchesData whatever SUCRE_OBS_FORMAT says)
        resto = engine.lease_restoration(height, width, len(lists), device=device, light=light, obs_format='f32', float_colour=not integral)
        weakref.finalize(self, engine.return_restoration, resto)
        resto.import_matches(None, lists)
        self.restoration = resto
        return resto

    def iter(self, batch_size: int = 1, device: str = 'cpu'):
        data = self.data if (self.data or self.restoration is None) else self._materialise()   # appended samples stay as given
        for i in range(0, len(data), batch_size):
            chunk = data[i:i + batch_size]
            yield (torch.hstack([s.u.to(device) for s in chunk]).long(),
                   torch.hstack([s.v.to(device) for s in chunk]).long(),
                   torch.hstack([s.cP.to(device) for s in chunk]),
                   torch.hstack([s.I.to(device) for s in chunk]))

    def __len__(self) -> int:
        if self.restoration is not None and not self.data:
            return self.restoration.n_obs()
        return sum(int(s.u.shape[0]) for s in self.data)


class MatchesFile:
    """Handle on the matches of one target image (loader.py:56-130).

    ``path`` is kept for interface parity; the matches themselves live in the engine's workspace, attached by
    ``sfm.Image.match_images``.  With ``persist=True`` (``--keep-matches``) ``save`` writes the reference's
    dataset names (``<image name>/{u1,v1,u2,v2,d,I}``) to ``path`` -- as HDF5 when h5py is reachable
    (``h5bridge.available``), else as a ``.npz`` next to it; ``on_disk`` / ``load_file`` accept either, so a later run
    consumes the kept matches instead of re-matching (sucre.py:185).
    """

    _groups = None   # view name -> match lists appended one view at a time (save_matches)
    _pending_check = None   # (device verdicts, n_views) between check_integrity(defer=True) and finish_integrity

    def __init__(self, path: Path, colmap_model=None, overwrite: bool = False):
        self.path = Path(path)
        if overwrite:
            self.path.unlink(missing_ok=True)
            self._npz_path.unlink(missing_ok=True)
        self.colmap_model = colmap_model
        self.restoration = None
        self.target_image = None
        self.image_list = None
        self._groups = {}

    @property
    def _npz_path(self) -> Path:
        return self.path.with_suffix('.npz')

    def exists(self) -> bool:
        return self.restoration is not None

    def attach(self, restoration, target_image, image_list) -> None:
        self.restoration = restoration
        self.target_image = target_image
        self.image_list = image_list

    def _need(self):
        if self.restoration is None:
            raise RuntimeError(f'{self.path}: no matches attached; call Image.match_images first')
        return self.restoration

    def get_image_list(self):
        """Images that passed the min_cover rule, in name order (the HDF5 group order of the reference)."""
        keep = self._need().view_keep().cpu().numpy().astype(bool)
        return sorted([im for im, k in zip(self.image_list, keep) if k], key=lambda im: im.name)

    def save_matches(self, matches, d: Tensor):
        """Appends the matches of one view (loader.py:68-76): ``u1, v1, u2, v2`` as int16, the depths ``d`` of the
        matched pixels of ``matches.image2``, colours to be filled by ``prepare_matches``.  The groups are kept in
        memory (``save`` writes them); ``load_matches`` hands them to the engine."""
        if self.restoration is not None:
            raise RuntimeError(f'{self.path}: matches of this file already live in the engine (Image.match_images)')
        if self._groups is None:
            self._groups = {}
        self._groups[matches.image2.name] = dict(u1=matches.u1.short().cpu(), v1=matches.v1.short().cpu(),
                                                 u2=matches.u2.short().cpu(), v2=matches.v2.short().cpu(),
                                                 d=d.detach().to(torch.float32).cpu(), I=None, image=matches.image2)

    def prepare_matches(self, num_workers: int = 0):
        """Colours were gathered by the match kernel (I = rgb2[v2,u2], loader.py:87): nothing left to do -- except for
        groups appended with ``save_matches``, whose colours are gathered here like the reference does."""
        if self._groups and self.restoration is None:
            for g in self._groups.values():
                rgb = g['image'].get_rgb()
                g['I'] = rgb[g['v2'].long(), g['u2'].long()].T.contiguous()
            return
        self._need()

    def check_integrity(self, defer: bool = False):
        """Device-side counterpart of loader.py:89-101: every stored range must be finite and >= 0 and the
        per-view totals must add up to n_obs.  One launch checks all views; the verdicts come back in one read.
        ``defer=True`` (not a reference argument) only enqueues the check and leaves its verdicts on the device:
        ``finish_integrity`` reads them later, so a pipeline can enqueue the fit behind the matching without waiting
        for either (sucre.restore_images)."""
        if self._groups and self.restoration is None:   # host lists: the reference's own checks, loader.py:89-101
            for name, g in self._groups.items():
                for key in ('u1', 'v1', 'u2', 'v2', 'd', 'I'):
                    arr = g[key]
                    assert arr is not None and not torch.isnan(arr.float()).any(), f'In {self.path}, dataset /{name}/{key} contains NaN(s).'
                    if key == 'd':
                        assert bool((arr > 0).all()), f'In {self.path}, dataset /{name}/d contains null of negative depth(s).'
                    else:
                        assert bool((arr >= 0).all()), f'In {self.path}, dataset /{name}/{key} contains invalid value(s).'
            return
        r = self._need()
        self._pending_check = (torch.cat([r.view_keep().to(torch.int64), r.view_counts().to(torch.int64),
                                          r.check_store().to(torch.int64), r.n_obs_device()]), r.n_views)
        if not defer:
            self.finish_integrity()

    def finish_integrity(self) -> int:
        """Reads the verdicts ``check_integrity`` left on the device, raises like it, and returns n_obs."""
        if self._pending_check is None:   # host lists: checked on the spot by check_integrity
            return sum(len(g['u1']) for g in (self._groups or {}).values())
        state, n = self._pending_check
        self._pending_check = None
        state = state.cpu().numpy()
        keep, counts, verdict, n_obs = state[:n].astype(bool), state[n:2 * n], state[2 * n:3 * n], int(state[3 * n])
        assert int(counts[keep].sum()) == n_obs, f'In {self.path}, observation count mismatch.'
        for k in np.nonzero(keep)[0]:
            assert not verdict[k] & 1, f'In {self.path}, view {k} contains NaN(s).'
            assert not verdict[k] & 2, f'In {self.path}, view {k} contains null of negative depth(s).'
            assert not verdict[k] & 4, f'In {self.path}, view {k} lost observations.'
        return n_obs

    def load_matches(self, pin_memory: bool = False) -> MatchesData:
        if self._groups and self.restoration is None:   # appended view by view: the reference's own loop, loader.py:103-118
            md = MatchesData()
            for name in sorted(self._groups):
                g = self._groups[name]
                assert g['I'] is not None, 'call prepare_matches() first'
                md.append(u=g['u1'], v=g['v1'], cP=g['image'].unproject_depth(u=g['u2'], v=g['v2'], d=g['d']), I=g['I'])
            return md
        return MatchesData(restoration=self._need(), image_list=self.image_list)

    def save(self) -> Path:
        """Persists the explicit match lists in the reference's layout (used by --keep-matches)."""
        r = self._need()
        keep = r.view_keep().cpu().numpy().astype(bool)
        groups = {}
        for k, im in enumerate(self.image_list):
            if not keep[k]:
                continue
            q = r.match_map(k)
            v1, u1 = torch.where(q >= 0)
            p2 = q[v1, u1].long()
            W2 = im.camera.width
            u2, v2 = p2 % W2, torch.div(p2, W2, rounding_mode='floor')
            view = im.device_view(r.device)
            I = view.rgb[v2, u2]
            I = (I if I.dtype == torch.float32 else (I.to(torch.float64) / 255).to(torch.float32)).T
            groups[im.name] = dict(u1=u1.short().cpu().numpy(), v1=v1.short().cpu().numpy(),
                                   u2=u2.short().cpu().numpy(), v2=v2.short().cpu().numpy(),
                                   d=view.depth[v2, u2].cpu().numpy(), I=I.cpu().numpy())
        if h5bridge.available():
            return h5bridge.write_groups(self.path, groups)
        np.savez(self._npz_path, **{f'{name}/{key}': val for name, ds in groups.items() for key, val in ds.items()})
        return self._npz_path

    def on_disk(self) -> bool:
        """True when a matches file written earlier (by this engine or by the reference) can be loaded."""
        return (self.path.exists() and h5bridge.available()) or self._npz_path.exists()

    def load_file(self, target_image, device='cuda', light: bool = False) -> None:
        """Consumes an existing HDF5 matches file instead of matching (what the reference does when the file is
        already there, sucre.py:185): every group becomes one view of the engine's store.  cP / z are rebuilt like
        loader.py:113 + sucre.py:53 (float32, same operation order as the match kernel).  ``light``: the store also
        keeps cP per observation, which the artificial-light model needs (the reference derives it from the same
        ``u2, v2, d`` every time it loads the file).  Kept colours that are not multiples of 1/255 (matches of resized
        images, --image-scale) are carried as float32 colours."""
        from . import engine
        if self.path.exists() and h5bridge.available():
            groups = h5bridge.read_groups(self.path)
        else:
            groups = h5bridge.read_npz_groups(self._npz_path)
        float_colour = False
        for ds in groups.values():
            k255 = ds['I'].astype(np.float64) * 255
            float_colour = float_colour or bool(k255.size and np.abs(k255 - np.rint(k255)).max() > 1e-3)
        images, lists = [], []
        for name, ds in groups.items():
            im = self.colmap_model[name]
            for key in h5bridge.DATASETS:
                arr = ds[key]
                assert not np.isnan(arr).any(), f'In {self.path}, dataset /{name}/{key} contains NaN(s).'
            u2, v2 = torch.tensor(ds['u2']), torch.tensor(ds['v2'])
            cP = im.unproject_depth(u=u2, v=v2, d=torch.tensor(ds['d']))
            z = torch.sqrt((cP[0] * cP[0] + cP[1] * cP[1]) + cP[2] * cP[2])
            item = [torch.tensor(ds['u1']), torch.tensor(ds['v1']), z]
            if float_colour and light:   # both extension sets: the camera points, then the float32 colours
                item += [None, torch.cat([cP.to(torch.float32).reshape(3, -1), torch.tensor(ds['I'].astype(np.float32)).reshape(3, -1)])]
            elif float_colour:
                item += [None, torch.tensor(ds['I'].astype(np.float32)).reshape(3, -1)]
            else:
                item.append(torch.tensor(np.rint(ds['I'].astype(np.float64) * 255).astype(np.uint8).T.copy()))
                if light:
                    item.append(cP.to(torch.float32).reshape(3, -1))
            images.append(im)
            lists.append(tuple(item))
        resto = engine.acquire_restoration(target_image.camera.height, target_image.camera.width, len(lists), device,
                                           light=light, float_colour=float_colour)
        target = target_image.device_view(device)
        resto.import_matches(target.as_float_colour() if float_colour else target, lists)
        self.attach(resto, target_image=target_image, image_list=images)

    def __len__(self) -> int:
        if self.restoration is not None:
            return self.restoration.n_obs()
        return sum(int(g['u1'].shape[0]) for g in (self._groups or {}).values())

    def __repr__(self) -> str:
        return f'MatchesFile(path={self.path}, {len(self)} observations)'
