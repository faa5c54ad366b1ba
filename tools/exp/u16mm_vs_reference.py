"""VERDICT round 5, task 1, on the CPU: the ORACLE fed uint16-millimetre ranges (oracle.quantize_ranges_u16mm: the engine's SUCRE_OBS_U16MM store
restated) against the unquantised REFERENCE goldens at BASELINE sizes -- what the lossy store costs against the reference itself, before
any GPU minute is spent (tests/test_gpu_baseline.py::test_u16mm_store_engine_vs_unquantised_reference runs the engine).
usage: python tools/exp/u16mm_vs_reference.py <baseline fixture name> [T_param T_closed]"""
import sys, time
ROOT = __import__('pathlib').Path(__file__).resolve().parents[2]; sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT))
import numpy as np, helpers
from oracle import oracle
name = sys.argv[1]
b = helpers.load_baseline(name)
sc = b.scene
per_view, samples = helpers.oracle_scene_samples(sc)
q = oracle.quantize_ranges_u16mm(samples)
tgt = sc.views[sc.target]
J0 = oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
Tp, Tc = int(b['T_param']), int(b['T_closed'])
if len(sys.argv) > 2: Tp, Tc = int(sys.argv[2]), int(sys.argv[3])
t=time.time()
J, p, tr = oracle.fit(sc.height, sc.width, q, J0, num_iter=Tp)
print('param time', time.time()-t)
try:
    helpers.check_baseline_fit(b, 'param', J, tr, 1e-4, 1e-4, 1e-4, 'oracle u16mm')
except AssertionError as e: print('FAIL', e)
t=time.time()
Jc, pc, trc = oracle.fit(sc.height, sc.width, q, None, num_iter=Tc, use_closed_form=True)
print('closed time', time.time()-t)
try:
    helpers.check_baseline_fit(b, 'closed', Jc, trc, 1e-4, 1e-4, 1e-4, 'oracle u16mm')
except AssertionError as e: print('FAIL', e)
