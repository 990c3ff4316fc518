#!/usr/bin/env python3
"""Build gate: compiles realrobot.hip with -Rpass-analysis=kernel-resource-usage and fails if a hot kernel uses
private scratch memory (spills / private arrays) -- see DESIGN.md section 7."""
import re
import subprocess
import sys

HOT = ('k_prep_a', 'k_prep_b', 'k_collide', 'k_solve', 'k_raster', 'k_render_list', 'k_shade', 'k_static_copy', 'k_restore', 'k_render_setup',
       'k_ik', 'k_plan_macro')


def main():
    cmd = sys.argv[1:]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    name, bad, seen = None, [], {}
    for line in out.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
        m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
        if m and name:
            seen[name] = int(m.group(1))
    for k, v in seen.items():
        if any(h in k for h in HOT) and v > 0:
            bad.append((k, v))
    for k, v in seen.items():
        print('%-70s scratch %d B/lane' % (k[:70], v))
    if bad or not seen:
        print('FAILED: scratch used by hot kernels:', bad)
        sys.exit(1)
    print('ok: no scratch in hot kernels')


if __name__ == '__main__':
    main()
