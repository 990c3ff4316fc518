"""CPU tests of the video composition (reference real_robots/videomaker.py:94-129): the inset resize, the
paste positions of the goal / start insets, the frame schedule and the AVI container.  The camera frames themselves come from the
HIP rasteriser (EnvCamera, tests/test_gpu_round2.py) -- here a stand-in camera returns synthetic frames."""
import os
import struct

import numpy as np

from real_robots_amd import videomaker as vm


def test_resize_area_is_an_exact_box_filter():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(240, 320, 3), dtype=np.uint8)
    out = vm.resize_area(img, 106, 80)              # 240 / 80 = 3 exactly; 320 / 106 is fractional
    assert out.shape == (80, 106, 3) and out.dtype == np.uint8
    rows = img.astype(np.float64).reshape(80, 3, 320, 3).mean(1)
    again = vm.resize_area(np.clip(np.rint(rows), 0, 255).astype(np.uint8), 106, 80)      # columns only: rows already at size
    assert np.abs(out.astype(int) - again.astype(int)).max() <= 1
    flat = np.full((240, 320, 3), 137, np.uint8)
    assert (vm.resize_area(flat, 106, 80) == 137).all()
    exact = vm.resize_area(img, 160, 120)            # 2 x 2 blocks
    want = np.rint(img.astype(np.float64).reshape(120, 2, 160, 2, 3).mean((1, 3)))
    assert np.array_equal(exact, want.astype(np.uint8))


def test_compose_frame_pastes_goal_top_right_and_start_top_left():
    cam = np.full((240, 320, 3), 200, np.uint8)
    goal = vm.make_inset(np.full((240, 320, 3), 10, np.uint8))
    start = vm.make_inset(np.full((128, 128, 3), 90, np.uint8))          # the 128 x 128 benchmark retina works too
    assert goal.shape == start.shape == (80, 106, 3)
    f = vm.compose_frame(cam, goal, start)
    assert (f[:80, :106] == 90).all() and (f[:80, 320 - 106:] == 10).all()      # videomaker.py:120-121: (W - W/3, 0) and (0, 0)
    assert (f[80:] == 200).all() and (f[:80, 106:320 - 106] == 200).all()
    assert (cam == 200).all()                                                  # the camera frame is not written to
    labelled = vm.make_inset(np.full((240, 320, 3), 255, np.uint8), "GOAL")
    assert labelled.shape == (80, 106, 3)
    rows = np.where((labelled != 255).any((1, 2)))[0]
    if len(rows):                                                               # (PIL present: the caption sits around 3/4 of the height)
        assert 50 <= rows.min() and rows.max() <= 75


class _Env:
    intrinsic_timesteps, extrinsic_timesteps, extrinsic_trials = 40, 24, 3


def test_videomaker_schedule_and_avi_container(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    m = vm.VideoMaker(_Env(), intrinsic=range(0, 41), extrinsic={1})
    count = [0]

    def fake_render(env):
        count[0] += 1
        return np.full((240, 320, 3), count[0], np.uint8)
    monkeypatch.setattr(m.camera, 'render', fake_render)
    assert m.frame_freq == 8                                   # 200 Hz simulation, 25 fps (videomaker.py:30-32)
    m.start_intrinsic()
    for s in range(1, 41):
        m.update_intrinsic(s)
    m.end_intrinsic()
    assert count[0] == 5                                       # steps 8, 16, 24, 32, 40
    obs = {'goal': np.full((240, 320, 3), 33, np.uint8), 'retina': np.full((240, 320, 3), 66, np.uint8)}
    for trial in range(3):
        m.start_trial(obs, trial)
        for s in range(1, 25):
            m.extrinsic_trial(obs, None, s, {})
        m.end_trial()
    assert count[0] == 5 + 3                                   # only trial 1 is filmed: steps 8, 16, 24
    assert len(m.files) == 2 and m.files[0].endswith('-intrinsic.avi') and m.files[1].endswith('-trial-1.avi')
    for name, frames in zip(m.files, (5, 3)):
        raw = open(name, 'rb').read()
        assert raw[:4] == b'RIFF' and raw[8:12] == b'AVI ' and struct.unpack('<I', raw[4:8])[0] == len(raw) - 8
        assert raw.count(b'00db') >= frames and len(raw) > frames * 320 * 240 * 3
        total = struct.unpack('<I', raw[48:52])[0]             # avih.dwTotalFrames
        assert total == frames
    # last trial frame: bottom-up BGR rows; the bottom row of the picture is camera (value 8), the top-left corner the start inset (66)
    raw = open(m.files[1], 'rb').read()
    last = np.frombuffer(raw[-320 * 240 * 3:], np.uint8).reshape(240, 320, 3)[::-1, :, ::-1]
    assert (last[239] == 8).all() and (last[0, 0] == 66).all() and (last[0, 319] == 33).all()
    for bad in (5, 1.5):
        try:
            vm.VideoMaker(_Env(), intrinsic=bad)
            assert False
        except Exception as ex:
            assert 'intrinsic' in str(ex)
    assert len(vm.VideoMaker(_Env(), intrinsic=True).intrinsic_frames) == 41 and os.path.exists(m.files[0])
