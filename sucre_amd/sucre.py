"""SUCRe restoration: model, optimiser loop, per-image pipeline and command line.

Drop-in for the reference's ``sucre.py`` (sucre.py:35-307): ``SUCRe``, ``adam``, ``restore_image``, ``parse_args``
and the argparse CLI keep their names, arguments, defaults, printed stages and output files
(``<stem>_rgb.png``, ``<stem>_reconstruction.png``, ``<name>.pt`` with keys ``B, beta, gamma, J``).  The 200
full-batch Adam iterations of ``adam`` (sucre.py:138-148) run as 200 back-to-back launches of one fused HIP
kernel over observations that never leave HBM, with no host synchronisation until the trace is read.

    python -m sucre_amd.sucre --image-dir IMAGES --depth-dir DEPTHS --model-dir COLMAP --output-dir OUT \\
        --image-name NAME [...]        # same flags as the reference, see --help

Under ``torchrun`` (WORLD_SIZE > 1) the images to restore are sharded over the ranks, one GPU each.
"""
from __future__ import annotations

import argparse
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch
from PIL import Image as PILImage
from torch import Tensor

from . import _pixelio, dist as sdist
from . import loader, se3, sfm


class SUCRe(torch.nn.Module):
    """Image-formation model ``I = l * (J exp(-beta z) + B (1 - exp(-gamma z)))`` of one image (sucre.py:35-82).

    Parameters ``B, beta, gamma`` are (3,1), initialised to 0.1; ``J`` is an (H,W,3) parameter initialised to the
    image itself with NaN where the depth map is invalid, or -- with ``use_closed_form`` -- a plain attribute
    recomputed from the observations."""

    def __init__(self, image: sfm.Image, light_model: bool = False, use_closed_form: bool = False,
                 _J_by_engine: bool = False):
        """``_J_by_engine`` (engine-internal): leave ``J`` uninitialised on the device -- the fit's first kernel
        (``sucre_fit_init``) writes the very same start values, ``float32(float64(k)/255)`` with NaN where the depth is
        invalid, and ``adam`` copies the fitted J back; saves the pipeline five torch launches and a host wait per image."""
        super().__init__()
        self.image = image
        self.light_model = light_model
        self.use_closed_form = use_closed_form
        self.B = torch.nn.Parameter(torch.full((3, 1), 0.1))
        self.beta = torch.nn.Parameter(torch.full((3, 1), 0.1))
        self.gamma = torch.nn.Parameter(torch.full((3, 1), 0.1))
        if light_model:  # light pose (twist, light-from-camera) and the 2x2 factor of the beam-shape matrix
            self.cam2light = torch.nn.Parameter(torch.zeros(6))
            self.sigma = torch.nn.Parameter(torch.eye(2))
        if not use_closed_form:
            cached = getattr(image, '_device_view', None)
            if _J_by_engine and cached is not None and cached[0].type == 'cuda':
                J = torch.empty(cached[1].rgb.shape, dtype=torch.float32, device=cached[1].rgb.device)
                self._J_unset = True
            elif cached is not None and cached[0].type == 'cuda':
                # the pixels are already resident for the engine: same values as get_rgb() / get_depth_map()
                # (float32(float64(k)/255), loader.py:157-163) without decoding the files a second time
                view = cached[1]
                J = view.rgb.clone() if view.rgb.dtype == torch.float32 else (view.rgb.to(torch.float64) / 255).to(torch.float32)
                J[view.depth <= 0] = torch.nan
            else:
                J = image.get_rgb()
                J[image.get_depth_map() <= 0] = torch.nan
            self.J = torch.nn.Parameter(J)

    # -- model evaluation (output stage / compatibility; the fit itself runs in the HIP engine) -------------------
    def compute_l_z(self, cP: Tensor) -> tuple[float | Tensor, Tensor]:
        """Illumination factor l and total light path z of camera-frame points (3, n) (sucre.py:52-64)."""
        z = cP.norm(dim=0)
        if not self.light_model:
            return 1.0, z
        R, t = se3.exp(self.cam2light)
        lP = R @ cP + t                                   # points in the light's frame
        lp = lP[:2] / lP[2]                               # (2, n) direction from the light axis
        Minv = (self.sigma.T @ self.sigma).inverse()
        q = (lp * (Minv @ lp)).sum(dim=0)                 # lp^T Sigma^-1 lp per point
        return torch.exp(-q / 2), z + lP.norm(dim=0)

    @torch.no_grad()
    def update_J(self, matches_data: loader.MatchesData, force_update: bool = False):
        """Closed-form J from the current water parameters (sucre.py:66-77), computed by the engine."""
        if not (self.use_closed_form or force_update):
            return
        resto = _restoration_of(matches_data, self)
        resto.params().copy_(self.water_vector().to(resto.device))
        resto.update_J()
        self.J = resto.J().to(self.B.device)

    def forward(self, u: Tensor, v: Tensor, cP: Tensor) -> Tensor:
        l, z = self.compute_l_z(cP)
        direct = self.J[v, u].T * torch.exp(-self.beta * z)
        veil = self.B * (1 - torch.exp(-self.gamma * z))
        return l * (direct + veil)

    def water_vector(self) -> Tensor:
        """B, beta, gamma (+ cam2light, sigma with the light model) flattened to the floats the engine stores."""
        parts = [self.B, self.beta, self.gamma] + ([self.cam2light, self.sigma] if self.light_model else [])
        return torch.cat([p.detach().flatten() for p in parts]).float()

    # -- output stage (sucre.py:84-121) -----------------------------------------------------------------------------
    @torch.no_grad()
    def plot_J(self) -> PILImage.Image:
        """Restored image: per-channel 1-99 percentile stretch of the valid pixels, invalid pixels black."""
        if self.J.is_cuda and _NUMPY_2:
            return PILImage.fromarray(self._plot_J_device().cpu().numpy())
        J = self.J.detach().cpu().numpy().copy()
        ok = ~np.isnan(J).any(axis=2)
        vals = J[ok]
        lo, hi = np.percentile(vals, 1, axis=0), np.percentile(vals, 99, axis=0)
        vals = np.clip(vals, lo, hi)
        vals = vals - vals.min(axis=0)
        vals = vals / vals.max(axis=0)
        J[~ok] = 0.0
        J[ok] = vals
        return PILImage.fromarray(np.uint8(J * 255))

    def _plot_J_device(self) -> Tensor:
        """The same stretch with J left on the GPU (uint8 (H,W,3) out).  The two percentiles of every channel are
        numpy's linear interpolation between two order statistics of the valid pixels: the engine finds those four
        values per channel by radix select on the device (``sucre_select_ranks``), the ranks and the interpolation
        are numpy's own float32 arithmetic (``percentile_plan`` / ``percentile_lerp``), so the percentiles are the
        very numbers ``np.percentile`` returns for the whole array.  After the clip the minimum is the lower
        percentile itself and the maximum of the shifted values the float32 difference of the two, so what remains is
        one elementwise pass of IEEE-exact operations (``sucre_plot_stretch``): the same bits as the host path."""
        from . import engine
        J = self.J.detach().contiguous()
        n = engine.count_valid(J)
        if n == 0:
            return torch.zeros(J.shape, dtype=torch.uint8, device=J.device)   # (numpy raises on an empty percentile)
        plan = [percentile_plan(n, q) for q in (1, 99)]
        stats = engine.select_ranks(J, [plan[0][0], plan[0][1], plan[1][0], plan[1][1]]).cpu().numpy()   # (3, 4)
        lo = np.array([percentile_lerp(stats[c, 0], stats[c, 1], plan[0][2]) for c in range(3)], np.float32)
        hi = np.array([percentile_lerp(stats[c, 2], stats[c, 3], plan[1][2]) for c in range(3)], np.float32)
        return engine.plot_stretch(J, lo, hi)

    @torch.no_grad()
    def plot_reconstruction(self) -> PILImage.Image:
        """The image as the fitted model re-synthesises it from J and the depth map."""
        dev = self.B.device
        u, v, cP = self.image.unproject_depth_map(self._depth_on(dev), to_world=False)
        out = torch.zeros((self.image.camera.height, self.image.camera.width, 3), device=dev)
        out[v, u] = self(u=u, v=v, cP=cP).clip(0, 1).T
        if out.is_cuda:   # the float32 product and the truncating cast are the same IEEE operations on either side
            return PILImage.fromarray((out * 255).to(torch.uint8).cpu().numpy())
        return PILImage.fromarray(np.uint8(out.cpu().numpy() * 255))

    @torch.no_grad()
    def plot_l(self) -> PILImage.Image:
        """Illumination factor over the image, jet colour map (sucre.py:96-104)."""
        import matplotlib.pyplot as plt
        dev = self.cam2light.device
        u, v, cP = self.image.unproject_depth_map(self._depth_on(dev), to_world=False)
        l, _ = self.compute_l_z(cP)
        lmap = torch.zeros((self.image.camera.height, self.image.camera.width), device=dev)
        lmap[v, u] = l
        return PILImage.fromarray(np.uint8(plt.colormaps['jet'](lmap.cpu().numpy())[:, :, :3] * 255))

    def _depth_on(self, dev) -> Tensor:
        """The target's depth map on ``dev``: the copy already resident for the engine when there is one (same
        values as ``get_depth_map``), else decoded from the file like the reference does."""
        if dev.type == 'cuda' and hasattr(self.image, 'device_view'):
            return self.image.device_view(dev).depth
        return self.image.get_depth_map().to(dev)

    def save_plots(self, save_dir: Path, iteration: int | None = None):
        stem = Path(self.image.name).stem
        tag = '' if iteration is None else f'_{iteration:04d}'
        _save_png(self.plot_J(), Path(save_dir) / f'{stem}_rgb{tag}.png')
        _save_png(self.plot_reconstruction(), Path(save_dir) / f'{stem}_reconstruction{tag}.png')
        if self.light_model:
            _save_png(self.plot_l(), Path(save_dir) / f'{stem}_vignetting{tag}.png')


def _save_png(img: PILImage.Image, path: Path) -> None:
    """Writes ``img`` as a PNG holding exactly its pixels.  PNG is lossless, so only the encoding effort is a choice:
    8-bit RGB images go through ``_pixelio.encode_rgb`` (Sub-filtered rows, one zlib stream at SUCRE_PNG_COMPRESS_LEVEL,
    default 1: 2.5-4x faster than PIL's encoder at the same level), in a worker process when the CLI has started some;
    anything else, or SUCRE_PNG_WRITER=pil, is saved by PIL (level 6 is PIL's default and what the reference writes)."""
    from . import _pixelio
    level = _pixelio.default_level()
    if img.mode != 'RGB' or os.environ.get('SUCRE_PNG_WRITER', 'zlib') == 'pil':
        img.save(path, compress_level=level)
    else:
        pool = _pixelio.POOL
        if pool is not None:
            try:
                pool.write(path, np.asarray(img), level)
                return
            except _pixelio.WorkerLost as e:
                print(f'warning: {e}; writing {path} in-process')
        _pixelio.write_rgb(str(path), np.asarray(img), level)


# percentile_plan / percentile_lerp restate numpy >= 2, where the quantile, the virtual index and the interpolation of a
# float32 array stay float32.  numpy 1.x runs np.percentile in float64 (and promotes the clipped image): there plot_J takes
# the host path, which calls np.percentile itself, so the picture equals the reference's on the same install either way.
_NUMPY_2 = int(np.__version__.split('.')[0]) >= 2


def percentile_plan(n: int, q: float) -> tuple[int, int, np.floating]:
    """(rank below, rank above, weight) of ``np.percentile(a, q)`` (method 'linear') for a float32 array of ``n``
    values, in numpy's own arithmetic: the quantile ``q / 100`` and the virtual index ``(n - 1) q'`` are formed in
    the array's dtype, float32 (numpy/lib/_function_base_impl.py: percentile, _QuantileMethods['linear'],
    _get_indexes, _get_gamma)."""
    quant = np.asanyarray(np.true_divide(q, np.float32(100)))
    virtual = np.asanyarray((n - 1) * quant)
    below = np.floor(virtual)
    above = below + 1
    if virtual >= n - 1:
        below = above = np.float32(n - 1)
    if virtual < 0:
        below = above = np.float32(0)
    gamma = np.asanyarray(virtual - below, dtype=virtual.dtype)
    return int(below), int(above), gamma[()]


def percentile_lerp(a, b, t):
    """numpy's ``_lerp``: ``a + (b - a) t``, formed from the upper end when ``t >= 0.5``."""
    a, b = np.float32(a), np.float32(b)
    diff = np.subtract(b, a)
    out = np.add(a, diff * t)
    if t >= 0.5:
        out = np.subtract(b, diff * (1 - t))
    return out


def _restoration_of(matches_data: loader.MatchesData, sucre: 'SUCRe | None' = None):
    """The engine workspace behind ``matches_data``: the one ``Image.match_images`` filled, or -- for a container built
    with ``append`` like the reference's ``load_matches`` does (loader.py:103-118) -- one the samples are imported
    into on first use.  There is no CPU path for the fit."""
    resto = getattr(matches_data, 'restoration', None)
    if resto is None and sucre is not None and getattr(matches_data, 'data', None):
        dev = sucre.B.device
        if dev.type != 'cuda':
            raise RuntimeError('the fit runs on the GPU engine: move the model to a cuda device (SUCRe(...).to("cuda"))')
        resto = matches_data.to_engine(sucre.image.camera.height, sucre.image.camera.width, dev, light=sucre.light_model)
    if resto is None:
        raise RuntimeError('this MatchesData holds no observations the engine can use (build it with Image.match_images + '
                           'MatchesFile.load_matches, or append samples to it); there is no CPU fallback for the fit')
    return resto


def _pull_results(sucre: SUCRe, resto) -> None:
    p = resto.params().to(sucre.B.device)
    with torch.no_grad():
        sucre.B.copy_(p[0:3].view(3, 1))
        sucre.beta.copy_(p[3:6].view(3, 1))
        sucre.gamma.copy_(p[6:9].view(3, 1))
        if sucre.light_model:
            sucre.cam2light.copy_(p[9:15])
            sucre.sigma.copy_(p[15:19].view(2, 2))
        J = resto.J().to(sucre.B.device)
        if sucre.use_closed_form:
            sucre.J = J
        else:
            sucre.J.copy_(J)
            sucre._J_unset = False


def _format_trace(trace: np.ndarray, first_iteration: int) -> str:
    """The per-iteration lines of sucre.py:149-152."""
    with np.printoptions(precision=4):
        return '\n'.join(f'iter: {first_iteration + i:04d}, cost: {row[0]:.4e}, B: {row[1:4].astype(np.float32)}, '
                         f'beta: {row[4:7].astype(np.float32)}, gamma: {row[7:10].astype(np.float32)}'
                         for i, row in enumerate(trace))


def _log_trace(trace: np.ndarray, first_iteration: int) -> None:
    print(_format_trace(trace, first_iteration))


def adam(sucre: SUCRe, matches_data: loader.MatchesData, lr: float = 0.05, num_iter: int = 200, batch_size: int = 1,
         save_dir: Path = None, save_interval: int = None, device: str = 'cpu', verbose: bool = True) -> SUCRe:
    """``num_iter`` steps of ``torch.optim.Adam(lr)`` on the least-squares cost (sucre.py:124-157).

    ``batch_size`` is accepted for compatibility: the engine always uses the full batch in one pass, which is
    what the reference's accumulated mini-batch gradients add up to."""
    print(f'Solve least squares with Adam optimizer ({num_iter} iterations).')
    resto = _adam_begin(sucre, matches_data)
    if save_dir is not None and save_interval is not None:
        stops = sorted({min(k + 1, num_iter) for k in range(0, num_iter, save_interval)} | {num_iter})
    else:
        stops = [num_iter]
    done = 0
    for stop in stops:
        snapshot = save_dir is not None and save_interval is not None and (stop - 1) % save_interval == 0
        if stop > done:
            # at a snapshot stop closed-form J stays J(theta_k) of the last iteration k: the pair
            # (J(theta_k), theta_{k+1}) is what the reference plots (sucre.py:141,153-154)
            trace = resto.fit(stop - done, lr=lr, use_closed_form=sucre.use_closed_form, keep_J=snapshot)
            if verbose:
                _log_trace(trace.cpu().numpy(), done)
            done = stop
        if snapshot:
            _pull_results(sucre, resto)
            sucre.save_plots(save_dir=save_dir, iteration=done - 1)
            if sucre.use_closed_form and done == num_iter:
                resto.update_J()   # the final update_J of sucre.py:156, held back for the snapshot
    _pull_results(sucre, resto)
    return sucre


def _adam_begin(sucre: SUCRe, matches_data: loader.MatchesData, params0: np.ndarray | None = None):
    """``params0``: the model's water (and light) parameters when the caller already has them on the host; reading
    them from the module costs a device-to-host copy, i.e. a wait for everything enqueued on the stream."""
    resto = _restoration_of(matches_data, sucre)
    if sucre.light_model and not resto.light:
        raise RuntimeError('these matches were computed without light_model=True: the camera points the light model '
                           'needs were not kept (call Image.match_images(..., light_model=True))')
    target = sucre.image.device_view(resto.device)
    if resto.float_colour:
        target = target.as_float_colour()
    J0 = None if (sucre.use_closed_form or getattr(sucre, '_J_unset', False)) else sucre.J.detach()
    if params0 is None:
        params0 = sucre.water_vector().detach().cpu().numpy()
    resto.fit_init(target, params0=params0, J0=J0)
    return resto


class _Job:
    """One image between ``_restore_submit`` (everything enqueued, nothing waited for) and ``_restore_finish``."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def _restore_submit(image: sfm.Image, colmap_model: sfm.COLMAPModel, output_dir: Path, light_model: bool,
                    use_closed_form: bool, min_cover: float, image_list: list[sfm.Image], lr: float, num_iter: int,
                    params_path: Path, force_compute_matches: bool, num_workers: int, device: str,
                    defer_checks: bool = False) -> _Job:
    """Stages of sucre.py:160-210 up to and including the enqueued fit; the trace and J stay on the device.
    ``defer_checks``: the integrity verdicts and the observation count stay on the device until ``_restore_finish``
    reads them with the results -- waiting for them here would make the host wait for the matching, which shares the
    GPU with the previous image's fit, before it can enqueue this image's fit (measured: 7-37 ms per image)."""
    print(f'Restore {image.name}.')
    output_dir = Path(output_dir)
    matches_path = (output_dir / image.name).with_suffix('.h5')
    matches_file = loader.MatchesFile(matches_path, colmap_model=colmap_model, overwrite=force_compute_matches)
    if image_list is None:
        image_list = list(colmap_model.images.values())

    # The model is built before the matching is enqueued (the reference builds it after, sucre.py:199; the two do not
    # depend on each other): moving its parameters to the device is a host-to-device copy, which the host waits for,
    # and behind the matching on the same stream that wait would last until the matching is done.
    if str(device).startswith('cuda'):
        image.device_view(device)   # J starts from the resident pixels (SUCRe.__init__)
    sucre = SUCRe(image=image, light_model=light_model, use_closed_form=use_closed_form,
                  _J_by_engine=defer_checks and params_path is None)
    if params_path is not None:
        sucre.load_state_dict(torch.load(params_path), strict=False)
    params0 = sucre.water_vector().detach().cpu().numpy()   # still on the host: no wait
    sucre = sucre.to(device)

    reuse = not force_compute_matches and matches_file.on_disk()
    if reuse:   # a kept matches file (ours or the reference's): consumed as is, like sucre.py:185
        try:
            matches_file.load_file(image, device=device, light=light_model)
        except NotImplementedError as e:   # e.g. --light-model on kept matches of resized images
            print(f'{matches_file.path.name} is not reused ({e}).')
            reuse = False
    if not reuse:
        print(f'Compute {image.name} matches.')
        image.match_images(image_list=image_list, matches_file=matches_file, min_cover=min_cover,
                           num_workers=num_workers, device=device, light_model=light_model)
        print('Prepare matches for optimization.')
        matches_file.prepare_matches(num_workers=num_workers)
    print('Check matches integrity.')
    matches_file.check_integrity(defer=defer_checks)
    print('Load matches.')
    matches_data = matches_file.load_matches(pin_memory=False)
    if not defer_checks:
        _report_observations(image, len(matches_data))

    return _Job(image=image, sucre=sucre, matches_file=matches_file, matches_data=matches_data,
                matches_path=matches_path, output_dir=output_dir, lr=lr, num_iter=num_iter, trace=None,
                deferred_checks=defer_checks, params0=params0)


def _report_observations(image: sfm.Image, n_obs: int) -> None:
    print(f'Total of {n_obs} observations.')
    if n_obs == 0:
        raise RuntimeError(f'{image.name}: no observation survived matching; nothing to fit')


def _restore_enqueue_fit(job: _Job) -> None:
    """The whole fit of sucre.py:138-156 enqueued in one go (no intermediate plots): no host synchronisation."""
    print(f'Solve least squares with Adam optimizer ({job.num_iter} iterations).')
    resto = _adam_begin(job.sucre, job.matches_data, params0=job.params0)
    job.trace = resto.fit(job.num_iter, lr=job.lr, use_closed_form=job.sucre.use_closed_form)


def _restore_enqueue_fits(jobs: list) -> None:
    """The fits of several images submitted together.  Images of one size without the extension planes (light model,
    float32 colours) advance in ONE launch per iteration (``engine.fit_batch``: what a small image pays for most is its own
    launches -- 640x480 x 5 views: 16.7 us per image and iteration alone, 7.3 in a launch of 32); every image's results are
    the bits of ``_restore_enqueue_fit``.  Whatever does not fit a batch is fitted by itself."""
    from . import engine
    groups: dict = {}
    singles = []
    for job in jobs:
        resto = _adam_begin(job.sucre, job.matches_data, params0=job.params0)
        job.resto = resto
        if resto.light or resto.float_colour:
            singles.append(job)
        else:
            groups.setdefault((resto.H, resto.W, resto.obs_format, job.num_iter, job.lr, job.sucre.use_closed_form), []).append(job)
    for key, members in groups.items():
        if len(members) == 1:
            singles.extend(members)
            continue
        print(f'Solve least squares with Adam optimizer ({key[3]} iterations), {len(members)} images per launch.')
        traces = engine.fit_batch([j.resto for j in members], key[3], lr=key[4], use_closed_form=key[5])
        for j, t in zip(members, traces):
            j.trace = t
    for job in singles:
        print(f'Solve least squares with Adam optimizer ({job.num_iter} iterations).')
        job.trace = job.resto.fit(job.num_iter, lr=job.lr, use_closed_form=job.sucre.use_closed_form)


def fit_batch_size(images: list, light_model: bool = False, in_flight: int = 2, n_views: int | None = None, device='cuda') -> int:
    """Images per fit launch in a survey (engine knob ``SUCRE_FIT_BATCH``; not a reference flag).  ``auto``: 8 for images of
    less than a megapixel (BASELINE config 1's 640x480: 144 -> 215 Mpix/s; 32 per launch: 219), 1 otherwise -- at 1080p a
    launch of two images is no faster than two launches in flight -- and 1 with the light model.
    Every image of a chunk holds its own workspace, ``fit_batch x in_flight`` of them at once: with ``n_views`` (the largest
    number of views a target can be matched against) the automatic batch is halved until those workspaces fit into 80 % of
    the device memory that is free now -- a 1280x720 survey with hundreds of neighbours keeps fewer images per launch instead
    of running out of memory."""
    env = os.environ.get('SUCRE_FIT_BATCH', 'auto').strip().lower()
    if env != 'auto':
        return max(1, int(env))
    if light_model or not images:
        return 1
    W, H = max(int(im.camera.width) for im in images), max(int(im.camera.height) for im in images)
    batch = 8 if W * H < 1_000_000 else 1
    if batch > 1 and n_views:
        from . import _lib, engine
        n = min(int(n_views), engine.MAX_VIEWS)
        cap = (n + 7) // 8 * 8 if n <= 256 else (n + 31) // 32 * 32    # engine.acquire_restoration's capacity steps
        need = int(_lib.load().sucre_workspace_bytes(H, W, min(cap, engine.MAX_VIEWS)))
        try:
            free = int(torch.cuda.mem_get_info(device)[0]) + int(torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
        except Exception:   # no device to ask: keep the plain rule
            return batch
        while batch > 1 and need * batch * max(1, int(in_flight)) > 0.8 * free:
            batch //= 2
    return batch


def _restore_finish(job: _Job, keep_matches: bool, writers: ThreadPoolExecutor | None = None):
    """Results back from the device (the only wait of the image), then the output files (sucre.py:211-219).  With a
    ``writers`` pool the files are written in the background -- percentile stretch, PNG encoding and the .pt take
    ~1 s of host time per 1080p image, 30x the GPU time -- and the future is returned instead of the model."""
    sucre = job.sucre
    if job.deferred_checks:   # raises what check_integrity / the empty-list test would have raised before the fit
        _report_observations(job.image, job.matches_file.finish_integrity())
    if job.trace is not None:
        job.trace = job.trace.cpu().numpy()
        if writers is None:
            _log_trace(job.trace, 0)
        _pull_results(sucre, _restoration_of(job.matches_data))
    if keep_matches and job.matches_file.restoration._views_dev is not None:   # freshly matched (not loaded from this
        print(f'Keep {job.matches_file.save()}.')                              # very file); needs the live workspace
    if writers is None:
        _write_outputs(job, keep_matches)
        return sucre
    if sucre.B.is_cuda:
        torch.cuda.current_stream(sucre.B.device).synchronize()   # the writer thread works on another stream
    return writers.submit(_write_outputs, job, keep_matches, True)


def _write_outputs(job: _Job, keep_matches: bool, log: bool = False) -> None:
    sucre = job.sucre
    if log and job.trace is not None:   # formatted off the main thread, printed as one block
        print(f'{job.image.name}:\n' + _format_trace(job.trace, 0))
    sucre.save_plots(save_dir=job.output_dir)
    torch.save({**sucre.cpu().state_dict(), 'J': sucre.J.detach().cpu()},
               (job.output_dir / job.image.name).with_suffix('.pt'))
    if not keep_matches and job.matches_path.exists():
        print(f'Erase {job.matches_path}.')
        job.matches_path.unlink()


def restore_image(image: sfm.Image, colmap_model: sfm.COLMAPModel, output_dir: Path, light_model: bool = False,
                  use_closed_form: bool = False, min_cover: float = 0.000001, image_list: list[sfm.Image] = None,
                  lr: float = 0.05, num_iter: int = 200, batch_size: int = 1, save_interval: int = None,
                  params_path: Path = None, force_compute_matches: bool = False, keep_matches: bool = False,
                  num_workers: int = 0, device: str = 'cpu'):
    """Per-image pipeline (sucre.py:160-219): match -> prepare -> check -> load -> fit -> save.  The signature is the
    reference's, default ``device='cpu'`` included -- which this engine refuses (``sfm.require_gpu``): pass a GPU."""
    sfm.require_gpu(device, 'restore_image')
    job = _restore_submit(image, colmap_model, output_dir, light_model, use_closed_form, min_cover, image_list, lr,
                          num_iter, params_path, force_compute_matches, num_workers, device)
    adam(sucre=job.sucre, matches_data=job.matches_data, lr=lr, num_iter=num_iter, batch_size=batch_size,
         save_dir=job.output_dir, save_interval=save_interval, device=device)
    return _restore_finish(job, keep_matches)


def restore_images(images: list[sfm.Image], colmap_model: sfm.COLMAPModel, output_dir: Path, in_flight: int = 2,
                   keep_matches: bool = False, device: str = 'cuda', fit_batch: int | None = None, **kw) -> None:
    """A survey: the same per-image pipeline with ``in_flight`` images on the GPU at once (engine.in_flight_slot)
    and the output files written by background threads (SUCRE_WRITER_THREADS).  While image i iterates,
    image i+1 is matched and submitted and the plots of images < i are encoded, so neither the tails of the fit
    launches nor the PNG encoding leave the GPU idle.  Per-image results are the same bits as ``restore_image``
    (each image has its own workspace and stream; nothing is shared).  ``fit_batch`` consecutive images share a slot and
    their fits one launch per iteration (``fit_batch_size``; same bits again)."""
    pending: list[tuple[int, list]] = []
    written = []
    if fit_batch is None:
        candidates = kw.get('image_list')
        n_views = len(candidates) if candidates is not None else len(colmap_model.images)
        fit_batch = fit_batch_size(images, bool(kw.get('light_model', False)), in_flight=in_flight, n_views=n_views, device=device)
    _restore_pipeline(images, colmap_model, output_dir, in_flight, keep_matches, device, pending, written, kw, max(1, int(fit_batch)))


def _restore_pipeline(images, colmap_model, output_dir, in_flight, keep_matches, device, pending, written, kw, fit_batch: int = 1) -> None:
    from . import engine
    # writer threads: the CPUs this process may use minus two (this thread and the HIP runtime's), at most 32
    with ThreadPoolExecutor(max_workers=max(1, int(os.environ.get('SUCRE_WRITER_THREADS', min(32, loader.effective_cpus() - 2)))),
                            thread_name_prefix='sucre-write') as writers:
        def finish(slot, jobs):
            with engine.in_flight_slot(slot, device, wait_for_caller=False):
                for job in jobs:
                    written.append(_restore_finish(job, keep_matches, writers))
            while len(written) > 64:                  # bound the host memory held by queued outputs (~50 MB each)
                written.pop(0).result()

        chunks = [images[i:i + fit_batch] for i in range(0, len(images), fit_batch)]
        for c, chunk in enumerate(chunks):
            slot = c % in_flight
            while pending and (len(pending) >= in_flight or pending[0][0] == slot):
                finish(*pending.pop(0))
            with engine.in_flight_slot(slot, device, wait_for_caller=False):   # inputs: Image.device_view, complete
                jobs = []
                for k, image in enumerate(chunk):
                    with engine.slot_lane(k):   # every image of the chunk in its own workspace
                        jobs.append(_restore_submit(image, colmap_model, output_dir, device=device, defer_checks=True, **kw))
                if len(jobs) == 1:
                    _restore_enqueue_fit(jobs[0])
                else:
                    _restore_enqueue_fits(jobs)
            pending.append((slot, jobs))
        for slot, jobs in pending:
            finish(slot, jobs)
        for f in written:
            f.result()   # re-raises anything a writer thread hit


def parse_args(args: argparse.Namespace):
    """Runs the CLI request (sucre.py:222-261); with WORLD_SIZE > 1 each rank restores its shard of the images."""
    rank, local_rank, world = sdist.env_rank_world()
    # torch's CPU thread pool follows the machine's CPU count; inside a container with a CPU quota that many spinning
    # threads only get the process throttled (147 CPU-seconds for 64 images on a 256-CPU box with a 16-CPU quota)
    local_world = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    cpus = max(1, loader.effective_cpus() // local_world)   # the ranks of one node share its CPUs (and its quota)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), cpus)))
    device = args.device
    if world > 1 and str(device).startswith('cuda'):
        # one GPU per rank; ranks beyond the GPUs of the box wrap around (two ranks on a one-GPU test box: per-image mode
        # has no collective, so sharing a GPU only shares its time)
        index = local_rank % max(1, torch.cuda.device_count())
        device = f'cuda:{index}'
        torch.cuda.set_device(index)
    print('Loading COLMAP model.')
    colmap_model = sfm.COLMAPModel(model_dir=args.model_dir, image_dir=args.image_dir, depth_dir=args.depth_dir,
                                   image_scale=args.image_scale)
    if args.image_name is not None:
        images = [colmap_model[args.image_name]]
    elif args.image_list is not None:
        images = [colmap_model[name] for name in args.image_list.read_text().splitlines() if name.strip()]
    else:
        lo, hi = args.image_ids
        images = [colmap_model.images[i] for i in range(lo, hi) if i in colmap_model.images]
    images = sdist.shard_images(images, rank, world)

    skipped = set(args.filter_images_path.read_text().splitlines()) if args.filter_images_path else set()
    image_list = [im for im in colmap_model.images.values() if im.name not in skipped]
    args.output_dir.mkdir(parents=True, exist_ok=True)
    in_flight = int(os.environ.get('SUCRE_IMAGES_IN_FLIGHT', '2'))   # engine knob, not a reference flag
    survey = len(images) > 1 and in_flight > 1 and args.save_interval is None and str(device).startswith('cuda')
    if survey:
        # image files are decoded and the result pictures encoded by child processes (_pixelio.WorkerPool: CPU work in
        # the process that drives the GPU slows its launches down, and PIL's decoder does not scale over threads);
        # SUCRE_IO_PROCESSES=0 keeps both in this process's threads
        n_io = int(os.environ.get('SUCRE_IO_PROCESSES', max(1, cpus - 2)))
        if n_io > 0:
            _pixelio.start_pool(n_io)
    # The host's torch work is all tiny (3x3 inverses and products per view, nine-number parameter vectors): with its default of
    # one intra-op thread per core every such call wakes the whole OpenMP team -- 2 ms instead of 0.1 ms per camera on a 16-core
    # box with the decode and writer threads next to it (tools/cli_survey_bench.py: 20 -> 16.5 ms per 640x480 image, 49 -> 41 per
    # 1080p image).  SUCRE_HOST_TORCH_THREADS (1; 0 = leave torch alone); restored on the way out.
    host_threads = int(os.environ.get('SUCRE_HOST_TORCH_THREADS', '1'))
    threads_before = torch.get_num_threads()
    if host_threads > 0:
        torch.set_num_threads(host_threads)
    try:
        _run_request(args, images, image_list, colmap_model, device, survey, in_flight)
    finally:
        _pixelio.stop_pool()
        if host_threads > 0:
            torch.set_num_threads(threads_before)


def _run_request(args, images, image_list, colmap_model, device, survey: bool, in_flight: int) -> None:
    if str(device).startswith('cuda') and images:   # start decoding + uploading the scene now, in the background
        loader.prefetch_for_targets(images, image_list, device, num_workers=args.num_workers, min_cover=args.min_cover)
    if survey:
        restore_images(images, colmap_model, args.output_dir, in_flight=in_flight, keep_matches=args.keep_matches,
                       device=device, light_model=args.light_model, use_closed_form=args.use_closed_form,
                       min_cover=args.min_cover, image_list=image_list, lr=args.learning_rate, num_iter=args.num_iter,
                       params_path=args.params_path, force_compute_matches=args.force_compute_matches,
                       num_workers=args.num_workers)
        return
    for image in images:
        restore_image(image=image, colmap_model=colmap_model, output_dir=args.output_dir,
                      light_model=args.light_model, use_closed_form=args.use_closed_form, min_cover=args.min_cover,
                      image_list=image_list, lr=args.learning_rate, num_iter=args.num_iter,
                      batch_size=args.batch_size, save_interval=args.save_interval, params_path=args.params_path,
                      force_compute_matches=args.force_compute_matches, keep_matches=args.keep_matches,
                      num_workers=args.num_workers, device=device)


def build_parser() -> argparse.ArgumentParser:
    """Same flags, defaults and exclusivity group as the reference CLI (sucre.py:264-305)."""
    p = argparse.ArgumentParser(description='SUCRe (MI355X engine).', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--image-dir', required=True, type=Path, help='directory holding the colour images')
    p.add_argument('--depth-dir', required=True, type=Path, help='directory holding the depth_<stem>.png maps')
    p.add_argument('--model-dir', required=True, type=Path, help='undistorted COLMAP model (cameras/images .bin or .txt)')
    p.add_argument('--output-dir', required=True, type=Path, help='where restored images and parameters go')
    which = p.add_mutually_exclusive_group(required=True)
    which.add_argument('--image-name', type=str, help='restore this single image')
    which.add_argument('--image-list', type=Path, help='text file, one image name per line')
    which.add_argument('--image-ids', type=int, nargs=2, metavar=('MIN_ID', 'MAX_ID'),
                       help='restore COLMAP image ids in [MIN_ID, MAX_ID)')
    p.add_argument('--light-model', action='store_true', help='model artificial lighting (light pose + beam shape)')
    p.add_argument('--use-closed-form', action='store_true',
                   help='solve J in closed form from the water parameters instead of optimising it')
    p.add_argument('--min-cover', type=float, default=0.000001,
                   help='drop a neighbour whose matches cover at most this fraction of the image')
    p.add_argument('--image-scale', type=float, default=1.0, help='rescale factor applied to all images (colour: OpenCV INTER_AREA when shrinking, INTER_CUBIC when '
                        'enlarging -- cv2 is used when installed, otherwise both are restated from its published algorithms, '
                        'unchecked against cv2 itself; depth: nearest neighbour)')
    p.add_argument('--filter-images-path', type=Path, help='text file of image names never used as neighbours')
    p.add_argument('--learning-rate', type=float, default=0.05, help='Adam learning rate')
    p.add_argument('--num-iter', type=int, default=200, help='Adam iterations')
    p.add_argument('--batch-size', type=int, default=5,
                   help='accepted for compatibility; the engine always runs the full batch in one pass')
    p.add_argument('--save-interval', type=int, help='also save the restored image every this many iterations')
    p.add_argument('--params-path', type=Path, help='.pt file to warm-start the model parameters from')
    p.add_argument('--force-compute-matches', action='store_true', help='discard a kept matches file first')
    p.add_argument('--keep-matches', action='store_true', help='export the match lists next to the outputs')
    p.add_argument('--num-workers', type=int, default=0, help='threads decoding images (0 = main thread)')
    p.add_argument('--device', type=str, default='cuda', help='GPU to run on')
    return p


def main(argv=None):
    parse_args(build_parser().parse_args(argv))


if __name__ == '__main__':
    main()
