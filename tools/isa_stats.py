#!/usr/bin/env python3
"""Instruction census of the device kernels: compiles realrobot.hip to gfx950 assembly and prints, per kernel, the number
of instructions and how many of them are fused multiply-adds (a kernel that must be contraction-free should show none
except under its address arithmetic).  Usage: tools/isa_stats.py [kernel ...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'real_robots_amd', 'csrc', 'realrobot.hip')


def census(extra_flags=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'rr.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-fno-unroll-loops', '-fno-slp-vectorize', '-std=c++17', '--offload-arch=gfx950',
                        '-S', '--cuda-device-only', '-w', *extra_flags, '-o', out, SRC], check=True)
        s = open(out).read()
    res = {}
    for m in re.finditer(r'^(_Z\d+(k_\w+?)\d+\w*):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
        ins = [l.split()[0] for l in m.group(3).split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        res[m.group(2)] = collections.Counter(ins)
    return res


if __name__ == '__main__':
    want = sys.argv[1:]
    for k, c in census().items():
        if want and k not in want:
            continue
        fused = {i: n for i, n in c.items() if re.search(r'fma|fmac|mad_f|mac_f', i)}
        print('%-16s %6d instructions, fused: %s' % (k, sum(c.values()), fused or 'none'))
