import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, helpers
from oracle import oracle
from sucre_amd import engine, synth
for name in ('plane_64x48_n4','relief_96x64_n6'):
    g=helpers.load_fixture(name); sc=g.scene
    views=engine.device_views_from_scene(sc,'cuda')
    r=engine.Restoration(sc.height,sc.width,len(views)); r.match(views[sc.target],views); r.fit_init(views[sc.target])
    tr=r.fit(200).cpu().numpy(); J=r.J().cpu().numpy()
    ref=g['J_param_200']
    _,samples=helpers.oracle_scene_samples(sc); tgt=sc.views[sc.target]
    Jo,po,to=oracle.fit(sc.height,sc.width,samples,oracle.init_J(tgt.rgb_u8.numpy(),tgt.depth_f32().numpy()),num_iter=200)
    print(name,'vs reference golden rms',helpers.rms_per_channel(J,ref).max(),'max',np.nanmax(np.abs(J-ref)),'| vs oracle rms',helpers.rms_per_channel(J,Jo).max(),'params',np.abs(tr[-1,1:]-to[-1,1:]).max(), '| oracle vs reference', helpers.rms_per_channel(Jo,ref).max())
sc=synth.make_scene(800,600,8,seed=3)
views=engine.device_views_from_scene(sc,'cuda')
r=engine.Restoration(600,800,len(views)); r.match(views[sc.target],views); r.fit_init(views[sc.target])
tr=r.fit(200).cpu().numpy(); J=r.J().cpu().numpy()
_,samples=helpers.oracle_scene_samples(sc); tgt=sc.views[sc.target]
Jo,po,to=oracle.fit(600,800,samples,oracle.init_J(tgt.rgb_u8.numpy(),tgt.depth_f32().numpy()),num_iter=200)
print('800x600x9 vs oracle rms',helpers.rms_per_channel(J,Jo).max(),'max',np.nanmax(np.abs(J-Jo)),'params',np.abs(tr[-1,1:]-to[-1,1:]).max())
