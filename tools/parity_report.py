#!/usr/bin/env python3
"""Prints the parity numbers (engine vs oracle vs reference golden) the GPU tests assert on."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import numpy as np, torch
import helpers
from oracle import oracle
from sucre_amd import engine

for name in ('plane_64x48_n4', 'relief_96x64_n6'):
    g = helpers.load_fixture(name)
    sc = g.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    _, samples = helpers.oracle_scene_samples(sc)
    tgt = sc.views[sc.target]
    for closed, key, tkey in ((False, 'J_param_200', 'trace_param'), (True, 'J_closed_200', 'trace_closed')):
        r = engine.Restoration(sc.height, sc.width, len(views))
        r.match(views[sc.target], views)
        r.fit_init(views[sc.target])
        tr = r.fit(200, use_closed_form=closed).cpu().numpy()
        J = r.J().cpu().numpy()
        J0 = None if closed else oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
        Jo, po, to = oracle.fit(sc.height, sc.width, samples, J0, num_iter=200, use_closed_form=closed)
        ref, rt = g[key], g[tkey]
        print(f'{name} closed={closed}: rms(J,ref)={helpers.rms_per_channel(J, ref)} rms(J,oracle)={helpers.rms_per_channel(J, Jo)} '
              f'rms(oracle,ref)={helpers.rms_per_channel(Jo, ref)}')
        print(f'    params |hip-ref| {np.abs(tr[:,1:]-rt[:,1:]).max():.3e} |hip-oracle| {np.abs(tr[:,1:]-to[:,1:]).max():.3e} '
              f'|oracle-ref| {np.abs(to[:,1:]-rt[:,1:]).max():.3e}  cost rel hip-ref {np.abs(tr[:,0]/rt[:,0]-1).max():.3e} oracle-ref {np.abs(to[:,0]/rt[:,0]-1).max():.3e}')
