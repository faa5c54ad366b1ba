import os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT)]
import numpy as np, torch
from sucre_amd import sfm, sucre, synth, loader, engine
W, H = 1920, 1080
survey = synth.make_survey(W, H, 6, 4, seed=3, device='cuda')
root = Path(tempfile.mkdtemp())
synth.write_to_disk(survey, root)
results = {}
orig_write = sucre._write_outputs
def run(tag, in_flight, no_writers=False, no_prefetch=False):
    engine.release_pool()
    model = sfm.COLMAPModel(root / 'model', root / 'images', root / 'depth')
    images = [model.images[i] for i in range(1, 13)]
    image_list = list(model.images.values())
    out = root / ('out_' + tag); out.mkdir(exist_ok=True)
    finals = {}
    def grab(job, keep, log=False):
        finals[job.image.name] = job.trace[-1].copy()
        if not no_writers:
            orig_write(job, keep, False)
    sucre._write_outputs = grab
    if not no_prefetch:
        loader.prefetch_for_targets(images, image_list, 'cuda', 0, 1e-6)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        if in_flight > 1:
            sucre.restore_images(images, model, out, in_flight=in_flight, device='cuda', light_model=False, use_closed_form=False, min_cover=1e-6,
                                 image_list=image_list, lr=0.05, num_iter=200, params_path=None, force_compute_matches=False, num_workers=0)
        else:
            for im in images:
                job = sucre._restore_submit(im, model, out, False, False, 1e-6, image_list, 0.05, 200, None, False, 0, 'cuda')
                sucre._restore_enqueue_fit(job)
                sucre._restore_finish(job, False)
    return finals
ref = run('ref', 1, no_writers=True, no_prefetch=True)
for tag, kw in (('if2_nowriters_noprefetch', dict(in_flight=2, no_writers=True, no_prefetch=True)),
                ('if2_nowriters', dict(in_flight=2, no_writers=True)),
                ('if2_full', dict(in_flight=2))):
    got = run(tag, **kw)
    bad = [n for n in ref if not np.array_equal(got[n], ref[n], equal_nan=True)]
    print(tag, 'mismatching images', len(bad), 'of', len(ref), flush=True)
