#!/usr/bin/env python3
"""Randomised sweep of the output stage on the GPU box: SUCRe.plot_J with J on the device (radix select of the order
statistics + one stretch kernel, csrc/plot.hip) against the host path, which is the reference's own numpy code
(sucre.py:84-94; tests/test_host_logic.py pins it to reference-made images).  Random sizes down to one pixel, NaN
fractions from none to all-but-one, values that repeat (ties around the percentiles), constant channels, negative and
huge values, infinities (a closed-form J that overflowed: the reference counts them as valid).
    python3 tools/plot_sweep.py [n_cases] [seed0]"""
import sys
import time
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sucre_amd import sucre  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 900
    rng = np.random.default_rng(seed0)
    g = torch.Generator().manual_seed(seed0)
    stats = {'cases': 0, 'empty': 0, 'with_inf': 0, 'constant_channel': 0}
    t0 = time.time()
    warnings.simplefilter('ignore')
    for case in range(n_cases):
        H, W = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        if rng.random() < 0.1:
            H, W = int(rng.integers(300, 1200)), int(rng.integers(300, 1200))
        kind = int(rng.integers(0, 6))
        J = torch.rand((H, W, 3), generator=g)
        if kind == 1:
            J = J ** 2 * 1.3 - 0.1                                   # negative values
        elif kind == 2:
            J = torch.round(J * float(rng.choice([2, 8, 64]))) / 8     # heavy ties
        elif kind == 3:
            J = J * float(rng.choice([1e-6, 1e6, 1e30]))
        elif kind == 4:
            J[..., int(rng.integers(0, 3))] = float(rng.normal())      # a constant channel: hi == lo
            stats['constant_channel'] += 1
        nan_frac = float(rng.choice([0.0, 0.03, 0.5, 0.97, 1.0]))
        if nan_frac >= 1.0:
            mask = torch.ones((H, W), dtype=torch.bool)
            if rng.random() < 0.7:
                mask.view(-1)[int(rng.integers(0, H * W))] = False     # all but one
        else:
            mask = torch.rand((H, W), generator=g) < nan_frac
        J[mask] = float('nan')
        if rng.random() < 0.2 and H * W > 3:
            J.view(-1, 3)[int(rng.integers(0, H * W)), int(rng.integers(0, 3))] = float('nan')   # one channel only
        with_inf = rng.random() < 0.15
        if with_inf:
            idx = torch.randint(0, H * W, (max(1, H * W // 50),), generator=g)
            J.view(-1, 3)[idx, int(rng.integers(0, 3))] = float('inf') * (1 if rng.random() < 0.7 else -1)
            stats['with_inf'] += 1
        m = sucre.SUCRe.__new__(sucre.SUCRe)
        torch.nn.Module.__init__(m)
        m.J = J.clone()
        n_valid = int((~torch.isnan(J).any(dim=2)).sum())
        if n_valid == 0:
            stats['empty'] += 1
            m.J = J.cuda()
            dev = np.asarray(m.plot_J())        # (numpy raises on an empty percentile; the device path returns black)
            assert not dev.any(), (case, 'empty')
            continue
        host = np.asarray(m.plot_J())
        m.J = J.cuda()
        dev = np.asarray(m.plot_J())
        bad = int((host != dev).sum())
        assert host.shape == (H, W, 3) and bad == 0, (case, H, W, kind, nan_frac, with_inf, n_valid, bad, host[host != dev][:5], dev[host != dev][:5])
        stats['cases'] += 1
        if (case + 1) % 50 == 0:
            print(f'{case + 1} cases ok, {stats}, {time.time() - t0:.0f}s', flush=True)
    print('plot sweep ok', stats)


if __name__ == '__main__':
    main()
