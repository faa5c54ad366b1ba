#!/usr/bin/env python3
"""Times the fit kernel of one or more builds of libsucre_hip on a FABRICATED full-cover workspace
(every tile sees every view), bypassing matching.  Experiment tool, not part of the product or the bench.

usage: python tools/microbench.py [--views 65] [--iters 30] lib1.so [lib2.so ...]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from sucre_amd import _lib  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument('libs', nargs='+')
p.add_argument('--views', type=int, default=65)
p.add_argument('--iters', type=int, default=30)
p.add_argument('--rounds', type=int, default=7)
p.add_argument('--height', type=int, default=1080)
p.add_argument('--width', type=int, default=1920)
p.add_argument('--split', action='store_true', help='use sucre_fit_grad/step (no fused tail)')
args = p.parse_args()
H, W, NV = args.height, args.width, args.views


def bind(path):
    lib = C.CDLL(str(path))
    for name, (res, a) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, a
    return lib


# offsets of private regions: mirror of csrc/layout.h (experiment tool only)
def layout(H, W, nv):
    tx, ty = (W + 15) // 16, (H + 15) // 16
    nt = tx * ty
    o = 0
    offs = {}
    def take(name, b):
        nonlocal o
        offs[name] = o
        o = (o + b + 255) // 256 * 256
    take('obs', nt * nv * 1792); take('cnt', nt * nv * 2)
    take('comp', nt * nv * 1792); take('pcount', nt * 256 * 2); take('pmask', nt * 256 * ((nv + 63) // 64) * 8); take('blockhist', 256 * nt * 4); take('bin_totals', 512 * 4)
    take('perm', nt * 256 * 4); take('invperm', nt * 256 * 4); take('levels', nt * 4); take('full', nt * 4); take('tile_off', nt * 8); take('total_chunks', 8)
    take('view_count', nv * 8); take('view_keep', nv * 4); take('n_obs', 8); take('n_obs_total', 8)
    ng = (min(nt, 1536) + 31) // 32
    take('params', 27 * 4); take('sums', 12 * 8); take('ticket', (1 + ng) * 16 * 4); take('gpartials', 10 * ng * 8); take('partials', nt * 10 * 4)
    take('J', nt * 768 * 4); take('m', nt * 768 * 4); take('v', nt * 768 * 4)
    return nt, offs, o


nt, offs, total = layout(H, W, NV)
dev = torch.device('cuda')
n_obs = nt * NV * 256
byts = 7 * n_obs + 72 * H * W


def make_ws():
    ws = torch.zeros(total, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(0)
    obs = ws[offs['comp']:offs['comp'] + nt * NV * 1792].view(nt * NV, 1792)   # compact store, every level full
    obs[:, :1024].view(torch.float32).copy_(2.5 + torch.rand((nt * NV, 256), device=dev, generator=g))
    obs[:, 1024:].copy_(torch.randint(0, 256, (nt * NV, 768), device=dev, generator=g, dtype=torch.uint8))
    ws[offs['levels']:offs['levels'] + nt * 4].view(torch.int32).fill_(NV)
    ws[offs['full']:offs['full'] + nt * 4].view(torch.int32).fill_(NV)
    ws[offs['tile_off']:offs['tile_off'] + nt * 8].view(torch.int64).copy_(
        torch.arange(nt, device=dev, dtype=torch.int64) * (NV * 1792))
    ident = torch.arange(nt * 256, device=dev, dtype=torch.int32)
    ws[offs['perm']:offs['perm'] + nt * 1024].view(torch.int32).copy_(ident)
    ws[offs['invperm']:offs['invperm'] + nt * 1024].view(torch.int32).copy_(ident)
    ws[offs['n_obs']:offs['n_obs'] + 8].view(torch.int64).fill_(n_obs)
    ws[offs['n_obs_total']:offs['n_obs_total'] + 8].view(torch.int64).fill_(n_obs)
    ws[offs['params']:offs['params'] + 36].view(torch.float32).fill_(0.1)
    ws[offs['J']:offs['J'] + nt * 768 * 4].view(torch.float32).copy_(torch.rand(nt * 768, device=dev, generator=g))
    return ws


ws = make_ws()   # one workspace shared by all builds (same bytes, same addresses)
wsp = C.c_void_p(ws.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = []
for path in args.libs:
    lib = bind(path)
    assert lib.sucre_workspace_bytes(H, W, NV) == total, (path, lib.sucre_workspace_bytes(H, W, NV), total)
    libs.append((Path(path).name, lib))


def run(lib, T, t0):
    if args.split:
        for it in range(T):
            assert lib.sucre_fit_grad(wsp, H, W, NV, t0 + it + 1, 0.05, 0.9, 0.999, 1e-8, 0, st) == 0
            assert lib.sucre_fit_step(wsp, H, W, NV, t0 + it + 1, 0.05, 0.9, 0.999, 1e-8, None, st) == 0
    else:
        assert lib.sucre_fit_run(wsp, H, W, NV, t0, T, 0.05, 0.9, 0.999, 1e-8, 0, None, st) == 0, lib.sucre_last_error()


for _, lib in libs:   # warm-up: clocks, page tables, code objects
    run(lib, 20, 0)
torch.cuda.synchronize()
times = {name: [] for name, _ in libs}
for rnd in range(args.rounds):   # interleaved rounds: order effects average out
    for name, lib in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(lib, args.iters, 20); e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / args.iters)
for name, _ in libs:
    t = sorted(times[name])
    med = t[len(t) // 2]
    print(f'{name:36s} median {med * 1e3:8.1f} us/iter (min {t[0] * 1e3:.1f} max {t[-1] * 1e3:.1f})   '
          f'algorithmic {byts / med / 1e6:8.1f} GB/s   ({byts / 1e6:.0f} MB, n_obs {n_obs})', flush=True)
