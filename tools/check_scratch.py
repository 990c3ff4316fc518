#!/usr/bin/env python3
"""Build gate: compiles realrobot.hip with -Rpass-analysis=kernel-resource-usage and fails if a hot kernel uses
private scratch memory (spills / private arrays) -- see DESIGN.md section 7.

A kernel whose resource summary reports a private segment is looked at in the ISA: the register allocator sometimes
leaves frame slots behind that it no longer uses (spill slots it then served from AGPRs or VGPR lanes).  Such a kernel
passes only if its code has NO scratch access instruction and the summary counts no VGPR spill; it is reported."""
import re
import subprocess
import sys

HOT = ('k_prep_a', 'k_prep_b', 'k_collide', 'k_solve', 'k_raster', 'k_render_list', 'k_shade', 'k_static_copy', 'k_restore', 'k_render_setup',
       'k_ik', 'k_plan_macro')


def main():
    cmd = sys.argv[1:]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    name, bad, seen, vspill = None, [], {}, {}
    for line in out.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
        m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
        if m and name:
            seen[name] = int(m.group(1))
        m = re.search(r'VGPRs Spill: (\d+)', line)
        if m and name:
            vspill[name] = int(m.group(1))
    suspects = [k for k, v in seen.items() if any(h in k for h in HOT) and v > 0]
    unused = {}
    if suspects:
        # the same compile to assembly: does the kernel's code touch its private segment at all?
        acmd = [a for a in cmd if not a.startswith('-Rpass') and a not in ('-shared', '-fPIC')]
        i = acmd.index('-o')
        acmd[i + 1] = '-'
        asm = subprocess.run(acmd + ['-S', '--cuda-device-only'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        for k in suspects:
            m = re.search(r'^%s:[^\n]*\n(.*?)^\.Lfunc_end' % re.escape(k), asm, re.S | re.M)
            body = m.group(1) if m else None
            touches = body is None or re.search(r'\bscratch_(load|store)|\bbuffer_(load|store)\w* .*\boffen\b|\bs_(add|mov)\w* s32\b', body)
            if touches or vspill.get(k, 1) != 0:
                bad.append((k, seen[k]))
            else:
                unused[k] = seen[k]
    for k, v in seen.items():
        print('%-70s scratch %d B/lane' % (k[:70], v))
    if bad or not seen:
        print('FAILED: scratch used by hot kernels:', bad)
        sys.exit(1)
    for k, v in unused.items():
        print('note: %s reports a %d B/lane private segment that its code never accesses (leftover frame slots)' % (k, v))
    print('ok: no scratch in hot kernels')


if __name__ == '__main__':
    main()
