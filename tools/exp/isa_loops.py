#!/usr/bin/env python3
"""Instruction account of the chunk loops of a fit kernel, from the compiler's own assembly.

  hipcc <the Makefile's flags> -S --cuda-device-only sucre_amd/csrc/fit.hip -o /tmp/fit.s
  python tools/exp/isa_loops.py /tmp/fit.s fit_closed_kernelILb1ELi0ELb0E

Prints every innermost loop (a label that a later s_cbranch jumps back to, with no other loop inside) that holds v_exp_f32
instructions: its vector / scalar / LDS / memory instruction counts by mnemonic."""
import collections
import re
import sys


def main():
    path, needle = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, ln in enumerate(lines) if ln.startswith('_Z') and needle in ln and ln.rstrip().split(':')[0].startswith('_Z') and ':' in ln)
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, ln in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', ln))}
    loops = []
    for i, ln in enumerate(body):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', ln) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)', ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    inner = [(a, b) for a, b in loops if not any((c > a and d < b) for c, d in loops if (c, d) != (a, b))]
    for a, b in inner:
        ops = [ln.split()[0] for ln in body[a:b + 1] if re.match(r'\s+[a-z]', ln) and not ln.strip().startswith(('.', ';'))]
        n_exp = sum(o == 'v_exp_f32_e32' or o.startswith('v_exp_f32') for o in ops)
        if n_exp == 0:
            continue
        hist = collections.Counter(ops)
        vec = sum(n for o, n in hist.items() if o.startswith('v_'))
        sc = sum(n for o, n in hist.items() if o.startswith('s_'))
        ds = sum(n for o, n in hist.items() if o.startswith('ds_'))
        mem = sum(n for o, n in hist.items() if o.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))
        print(f'loop at +{a}..+{b} ({b - a + 1} lines): {len(ops)} instructions = {vec} vector ({n_exp} v_exp_f32) + {sc} scalar + {ds} LDS + {mem} memory')
        for o, n in sorted(hist.items(), key=lambda kv: -kv[1]):
            print(f'    {n:4d}  {o}')


if __name__ == '__main__':
    main()
