"""CPU tests of the build gates that `__graft_entry__.build()` runs: tools/check_dpp_hazard.py on synthetic assembly (a stand-in
"compiler" that prints a given listing) -- the gate must pass the shape the kernels have today and fail on each hazard it exists for."""
import os
import stat
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, 'tools', 'check_dpp_hazard.py')

GOOD = """
k_solve:
	v_fma_f32 v1, v2, v3, v4
	v_cndmask_b32_e32 v5, v6, v7, vcc
	;;#ASMSTART
	s_nop 1
	v_fmac_f32_dpp v8, v9, v10 row_newbcast:3 row_mask:0xf bank_mask:0xf
	;;#ASMEND
	s_and_saveexec_b64 s[0:1], vcc
	v_add_f32_e32 v1, v1, v2
	s_endpgm
"""
BAD_NEAR = GOOD.replace("\tv_cndmask_b32_e32 v5, v6, v7, vcc\n", "\tv_cmpx_lt_f32_e32 v6, v7\n")
BAD_FAR = GOOD.replace("\ts_endpgm\n", "\tv_cmpx_gt_i32_e64 v1, v2\n\ts_endpgm\n")
NO_BLOCK = GOOD.replace(";;#ASMSTART", "; (no asm)").replace(";;#ASMEND", "; (no asm)")


def _run(tmp_path, listing):
    asm = tmp_path / 'listing.s'
    asm.write_text(listing)
    fake = tmp_path / 'fakecc'
    fake.write_text("#!/bin/sh\ncat %s\n" % asm)
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    return subprocess.run([sys.executable, TOOL, str(fake), '-O3', 'realrobot.hip'], capture_output=True, text=True)


def test_dpp_hazard_gate_passes_today_and_catches_each_hazard(tmp_path):
    ok = _run(tmp_path, GOOD)
    assert ok.returncode == 0 and 'ok: 1 inline-asm DPP blocks' in ok.stdout, ok.stdout + ok.stderr
    near = _run(tmp_path, BAD_NEAR)           # a VALU write of EXEC within the five instructions in front of the block
    assert near.returncode == 1 and 'v_cmpx_lt_f32' in near.stdout
    far = _run(tmp_path, BAD_FAR)             # ... or anywhere in the translation unit (a block at a branch target hides its predecessors)
    assert far.returncode == 1 and 'v_cmpx_gt_i32' in far.stdout
    none = _run(tmp_path, NO_BLOCK)           # the gate must notice when it no longer sees what it is there for
    assert none.returncode == 1 and 'no inline-asm DPP block' in none.stdout


def test_archived_object_heavy_class_patch_still_applies():
    """scratch/object_heavy_class.patch is the round-6 experiment as built (NOTEBOOK.md "Round 6 -- measured and dropped"): kept as
    a patch against the kernel sources, which only means something while it applies to them."""
    patch = os.path.join(ROOT, 'scratch', 'object_heavy_class.patch')
    assert os.path.exists(patch)
    r = subprocess.run(['git', 'apply', '--check', patch], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(patch).read()
    assert 'k_solve_oh' in text and 'oh_object_wave' in text
