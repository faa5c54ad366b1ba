import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT)]
import torch
from sucre_amd import engine, synth
W, H = 1920, 1080
survey = synth.make_survey(W, H, 6, 4, seed=3, device='cuda')
views = engine.device_views_from_scene(survey, 'cuda')
targets = [7, 8, 9, 10, 13, 14, 15, 16]
jobs = []
for t in targets:
    sel = survey.neighbours(t, 14 + (t % 5))     # different view counts per target
    jobs.append((views[t], [views[q] for q in sel]))
cap = max(len(v) for _, v in jobs)
ref = []
r = engine.Restoration(H, W, cap)
for tgt, vs in jobs:
    r.match(tgt, vs); r.fit_init(tgt); t = r.fit(200); torch.cuda.synchronize()
    ref.append((r.J().clone(), t.clone(), r.n_obs(), r.view_counts().clone()))
del r
for ns in (2, 3):
    restos = [engine.Restoration(H, W, cap) for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    outs = []
    for rep in range(3):
        for i, (tgt, vs) in enumerate(jobs):
            with torch.cuda.stream(streams[i % ns]):
                rr = restos[i % ns]
                rr.match(tgt, vs); rr.fit_init(tgt); t = rr.fit(200)
                outs.append((i, rr.J(), t, rr.view_counts().clone()))
    torch.cuda.synchronize()
    for i, J, t, vc in outs:
        if not (torch.equal(torch.nan_to_num(J), torch.nan_to_num(ref[i][0])) and torch.equal(t, ref[i][1])):
            d = (t != ref[i][1]).any(dim=1).nonzero().flatten()
            r0 = int(d[0])
            cols = (t[r0] != ref[i][1][r0]).nonzero().flatten().tolist()
            print('streams', ns, 'job', i, 'first differing trace row', r0, 'cols', cols, 'rel', float(((t[r0] - ref[i][1][r0]).abs() / ref[i][1][r0].abs()).max()), flush=True)
    print('streams', ns, 'done', flush=True)
