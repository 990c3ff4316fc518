#!/usr/bin/env python3
"""Build gate for the hand-placed DPP instructions in inline asm (rr_solve.inc MOTOR_STEP: `s_nop 1; v_fmac_f32_dpp ...`).

The compiler's hazard recogniser does not look inside inline asm.  The asm's own `s_nop 1` covers the "VALU writes a VGPR ->
DPP reads it" hazard (2 wait states); it does NOT cover "VALU writes EXEC -> DPP" (5 wait states: v_cmpx*, v_readlane/
v_readfirstlane into exec, any VALU instruction with an exec destination).  This gate compiles the translation unit to gfx950
assembly and fails if any instruction that writes EXEC from the VALU sits within the five instructions in front of an
inline-asm block containing a DPP instruction -- i.e. it turns "today's build happens to be safe" into a checked property of
every build.  Usage: check_dpp_hazard.py <hipcc> <flags...> realrobot.hip   (flags as for the library, no -o)."""
import re
import subprocess
import sys


def main():
    cmd = [a for a in sys.argv[1:] if a not in ('-shared', '-fPIC')] + ['-S', '--cuda-device-only', '-o', '-']
    asm = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout.split('\n')
    valu_exec = re.compile(r'^\s*(v_cmpx\w*\s|v_\w+\s+exec(_lo|_hi)?\b|v_read(first)?lane_b32\s+exec)')
    n_blocks, bad = 0, []
    i = 0
    while i < len(asm):
        if '#ASMSTART' in asm[i]:
            j = i + 1
            while j < len(asm) and '#ASMEND' not in asm[j]:
                j += 1
            if any('_dpp' in l or 'row_newbcast' in l for l in asm[i + 1:j]):
                n_blocks += 1
                seen, k = 0, i - 1
                while k >= 0 and seen < 5:                      # the five instructions in front of the block
                    line = asm[k].split(';')[0].strip()
                    k -= 1
                    if not line or line.startswith('.') and not line.endswith(':'):
                        continue
                    if line.endswith(':'):
                        break                                   # a label: a branch target -- predecessors unknown, covered below
                    seen += 1
                    if valu_exec.match(line):
                        bad.append((i + 1, line))
            i = j
        i += 1
    # (a block at a branch target could be reached from a path that ends in a VALU exec write: none of the kernels writes EXEC
    # from the VALU at all -- asserted, so the label case above cannot hide one)
    writers = [(n + 1, l.strip()) for n, l in enumerate(asm) if valu_exec.match(l.split(';')[0])]
    if n_blocks == 0:
        print("check_dpp_hazard: no inline-asm DPP block found (expected the motor rows' v_fmac_f32_dpp)")
        return 1
    if bad or writers:
        for n, l in bad + writers:
            print("check_dpp_hazard: VALU write of EXEC at asm line %d: %s" % (n, l))
        return 1
    print("ok: %d inline-asm DPP blocks, no VALU write of EXEC in the translation unit" % n_blocks)
    return 0


if __name__ == '__main__':
    sys.exit(main())
