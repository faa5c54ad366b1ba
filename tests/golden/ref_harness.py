"""Drive the *real* reference (clementinboittiaux/sucre, mounted read-only at /root/reference) on synthetic
scenes.  Only usable in the dev container: the reference never travels to the GPU box, so this module is
imported exclusively by the fixture generators next to it (``gen_golden.py``, ``gen_golden_extras.py``,
``gen_golden_baseline.py``) and by ``tools/time_reference.py`` (the CPU-baseline calibration); no test imports it.

The reference's hot path imports fine once its three I/O-only dependencies (cv2, h5py, pycolmap — all absent
here and never called by the functions we exercise) are registered as empty modules.  Synthetic pixels are
injected by overriding ``sfm.Image.get_rgb/get_depth_map`` (sfm.py:109-113); the HDF5 round trip of
``loader.MatchesFile`` is reproduced by value (int16 casts of loader.py:71-74, ``d = depth2[v2,u2]`` of
sfm.py:137, ``cP = unproject_depth(u2,v2,d)`` of loader.py:113, ``I = rgb2[v2,u2].T`` of loader.py:87, groups in
name order).
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import numpy as np
import torch

REFERENCE_DIR = Path('/root/reference/sucre')


def reference_available() -> bool:
    return (REFERENCE_DIR / 'sucre.py').exists()


def import_reference():
    """Returns the reference modules (sfm, loader, se3, sucre)."""
    if not reference_available():
        raise RuntimeError('reference not mounted at /root/reference')
    for missing in ('cv2', 'h5py', 'pycolmap'):
        if missing not in sys.modules:
            try:
                __import__(missing)
            except ImportError:
                sys.modules[missing] = types.ModuleType(missing)
    own = {k: sys.modules.pop(k) for k in ('sfm', 'loader', 'se3', 'sucre') if k in sys.modules}
    sys.path.insert(0, str(REFERENCE_DIR))
    try:
        import importlib
        mods = tuple(importlib.import_module(n) for n in ('sfm', 'loader', 'se3', 'sucre'))
    finally:
        sys.path.remove(str(REFERENCE_DIR))
        for k in ('sfm', 'loader', 'se3', 'sucre'):
            sys.modules.pop(k, None)
        sys.modules.update(own)
    return mods


def build_reference_images(scene, sfm):
    """One reference ``sfm.Image`` per synthetic view, serving the synthetic pixels."""
    camera = sfm.Camera(camera_id=1, width=scene.width, height=scene.height, K=scene.K.clone())

    class _SynthImage(sfm.Image):
        def __init__(self, image_id, view):
            super().__init__(image_id=image_id, rgb_path=Path(view.name),
                             depth_map_path=Path('depth_' + view.name), pose=sfm.Pose(view.R.clone(), view.t.clone()),
                             camera=camera)
            self._view = view

        def get_rgb(self):
            return self._view.rgb_f32()

        def get_depth_map(self):
            return self._view.depth_f32()

    return [_SynthImage(i + 1, v) for i, v in enumerate(scene.views)]


def reference_matches(scene, min_cover: float = 1e-6):
    """Runs sfm.Image.match_two_way for the target against every view (sfm.py:127-138 without the file).

    Returns (per-view dict list, MatchesData) where the MatchesData is what ``load_matches`` would hand to
    ``sucre.adam``.
    """
    sfm, loader, _, _ = import_reference()
    images = build_reference_images(scene, sfm)
    target = images[scene.target]
    depth1 = target.get_depth_map()
    u1, v1, wP1 = target.unproject_depth_map(depth1, to_world=True)
    per_view = []
    for other in images:
        depth2 = other.get_depth_map()
        u2, v2, wP2 = other.unproject_depth_map(depth2, to_world=True)
        m = target.match_two_way(other, u1=u1, v1=v1, wP1=wP1, u2=u2, v2=v2, wP2=wP2)
        kept = len(m) / (target.camera.width * target.camera.height) > min_cover
        rec = dict(name=other.name, kept=bool(kept), u1=m.u1.short(), v1=m.v1.short(), u2=m.u2.short(),
                   v2=m.v2.short(), d=depth2[m.v2, m.u2])
        per_view.append(rec)
    matches_data = loader.MatchesData()
    for rec, other in sorted(zip(per_view, images), key=lambda p: p[0]['name']):
        if not rec['kept']:
            continue
        cP = other.unproject_depth(u=rec['u2'], v=rec['v2'], d=rec['d'])
        I = other.get_rgb()[rec['v2'].long(), rec['u2'].long()].T.contiguous()
        rec['cP'] = cP
        rec['I'] = I
        matches_data.append(u=rec['u1'], v=rec['v1'], cP=cP, I=I)
    return per_view, matches_data, target


def reference_fit(scene, matches_data, target, num_iter: int, use_closed_form: bool = False,
                  light_model: bool = False, batch_size: int = 5, lr: float = 0.05, snapshots=()):
    """Runs sucre.adam (sucre.py:124-157) and records the exact per-iteration trace through public hooks.

    trace[i] = (cost_i, B, beta, gamma *after* step i).  ``snapshots`` = iteration counts after which J is
    captured (J-parameter mode only).
    """
    _, loader, _, sucre_mod = import_reference()
    model = sucre_mod.SUCRe(image=target, light_model=light_model, use_closed_form=use_closed_form)
    trace = []
    snaps = {}
    state = dict(cost=0.0, batches=None, bi=0, it=0)

    batches_I = [I for (_, _, _, I) in matches_data.iter(batch_size=batch_size)]

    def fwd_hook(module, inputs, output):
        if not torch.is_grad_enabled():
            return
        I = batches_I[state['bi'] % len(batches_I)]
        state['cost'] += torch.square(I - output.detach()).sum().item()
        state['bi'] += 1

    def step_hook(optimizer, args, kwargs):
        state['it'] += 1
        row = [state['cost']] + [float(x) for p in (model.B, model.beta, model.gamma) for x in p.detach().flatten()]
        extra = []
        if light_model:
            extra = [float(x) for x in model.cam2light.detach().flatten()] + \
                    [float(x) for x in model.sigma.detach().flatten()]
        trace.append(row + extra)
        state['cost'] = 0.0
        if state['it'] in snapshots and not use_closed_form:
            snaps[state['it']] = model.J.detach().clone().numpy()

    h1 = model.register_forward_hook(fwd_hook)
    from torch.optim.optimizer import register_optimizer_step_post_hook
    h2 = register_optimizer_step_post_hook(step_hook)
    try:
        sucre_mod.adam(model, matches_data, lr=lr, num_iter=num_iter, batch_size=batch_size, device='cpu')
    finally:
        h1.remove()
        h2.remove()
    J = model.J.detach().numpy().copy()
    return dict(J=J, trace=np.asarray(trace, dtype=np.float64), snaps=snaps, model=model)


def reference_shared_water(scenes, num_iter: int, batch_size: int = 5, lr: float = 0.05, use_closed_form: bool = False):
    """Shared-water composition built from reference classes only: one reference SUCRe
    module per image, their B / beta / gamma attributes bound to the SAME Parameter objects, one
    torch.optim.Adam over the de-duplicated parameter set, every batch loss divided by the TOTAL observation count
    (the per-image loop body of sucre.py:143-146 otherwise unchanged).  ``use_closed_form``: every module re-solves
    its J with its own ``update_J`` at the top of the iteration (sucre.py:140-141) and once more after the last
    step (sucre.py:155-156), exactly as ``sucre.adam`` does for one image."""
    _, loader, _, sucre_mod = import_reference()
    models, datas = [], []
    for scene in scenes:
        _, md, target = reference_matches(scene)
        models.append(sucre_mod.SUCRe(image=target, use_closed_form=use_closed_form))
        datas.append(md)
    for m in models[1:]:
        m.B, m.beta, m.gamma = models[0].B, models[0].beta, models[0].gamma
    params, seen = [], set()
    for m in models:
        for p in m.parameters():
            if id(p) not in seen:
                seen.add(id(p)); params.append(p)
    n_total = sum(len(d) for d in datas)
    opt = torch.optim.Adam(params, lr=lr)
    trace = []
    for _ in range(num_iter):
        opt.zero_grad()
        cost = 0.0
        for m, d in zip(models, datas):
            if use_closed_form:
                m.update_J(d)
            for u, v, cP, I in d.iter(batch_size=batch_size, device='cpu'):
                loss = torch.square(I - m(u=u, v=v, cP=cP)).sum()
                (loss / n_total / 3).backward()
                cost += loss.item()
        opt.step()
        trace.append([cost] + [float(x) for p in (models[0].B, models[0].beta, models[0].gamma) for x in p.detach().flatten()])
    if use_closed_form:
        for m, d in zip(models, datas):
            m.update_J(d)
    return dict(J=[m.J.detach().numpy().copy() for m in models], trace=np.asarray(trace), n_total=n_total)
