"""Videos of the intrinsic phase and of extrinsic trials (mirror of real_robots/videomaker.py:11-154).

Frames come from the debug camera of the reference's VideoMaker -- EnvCamera(1.0, 90, -45, 0, [-0.3, 0, .4], fov=90, 320x240)
(videomaker.py:25-26) -- rendered by the HIP rasteriser; an extrinsic trial's frames carry two insets, the goal image top right
and the start image top left, each a third of the frame with its caption (videomaker.py:94-105, 112-125).  The composition is
plain numpy (`make_inset`, `compose_frame`); captions are drawn with PIL when it is importable.  The container is an
uncompressed RGB24 AVI written here (the reference's cv2 / XVID writer is not a dependency of this package).
"""
import struct
import time

import numpy as np

VIDEO_WIDTH, VIDEO_HEIGHT = 320, 240      # videomaker.py:8-9


def resize_area(image, width, height):
    """Area-average resize of an [H, W, C] uint8 image (exact box filter; any ratio)."""
    img = np.asarray(image, dtype=np.float64)
    H, W = img.shape[:2]

    def weights(n_in, n_out):
        edges = np.linspace(0.0, n_in, n_out + 1)
        w = np.zeros((n_out, n_in))
        for o in range(n_out):
            lo, hi = edges[o], edges[o + 1]
            for i in range(int(np.floor(lo)), min(int(np.ceil(hi)), n_in)):
                w[o, i] = min(hi, i + 1) - max(lo, i)
            w[o] /= w[o].sum()
        return w
    out = np.tensordot(weights(H, height), img, axes=(1, 0))
    out = np.tensordot(weights(W, width), out, axes=(1, 1)).transpose(1, 0, 2)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def make_inset(image, text=None):
    """A third-size copy of `image` with `text` centred at three quarters of its height (videomaker.py:94-105)."""
    iw, ih = int(VIDEO_WIDTH / 3), int(VIDEO_HEIGHT / 3)
    inset = resize_area(image, iw, ih)
    if text:
        try:
            from PIL import Image, ImageDraw, ImageFont
            img = Image.fromarray(inset)
            d = ImageDraw.Draw(img)
            font = ImageFont.load_default()
            box = d.textbbox((0, 0), text, font=font)
            w, h = box[2] - box[0], box[3] - box[1]
            d.text((int((iw - w) / 2), int(ih * 0.75 - h / 2)), text, fill=(0, 0, 0), font=font)
            inset = np.asarray(img, dtype=np.uint8)
        except ImportError:
            pass
    return inset


def compose_frame(camera, goal_inset=None, start_inset=None):
    """The trial frame of videomaker.py:119-121: goal inset pasted at (W - W/3, 0), start inset at (0, 0).  Returns a copy."""
    frame = np.array(camera, dtype=np.uint8, copy=True)
    assert frame.shape == (VIDEO_HEIGHT, VIDEO_WIDTH, 3)
    if goal_inset is not None:
        x0 = VIDEO_WIDTH - int(VIDEO_WIDTH / 3)
        frame[:goal_inset.shape[0], x0:x0 + goal_inset.shape[1]] = goal_inset
    if start_inset is not None:
        frame[:start_inset.shape[0], :start_inset.shape[1]] = start_inset
    return frame


class RawAviWriter:
    """Uncompressed RGB24 AVI (RIFF 'AVI ', one video stream, bottom-up BGR DIB frames): playable everywhere, no codec needed."""

    def __init__(self, filename, fps, width, height):
        self.f, self.fps, self.w, self.h, self.n = open(filename, 'wb'), int(fps), int(width), int(height), 0
        self.frame_bytes = self.w * self.h * 3
        self._header()

    def _header(self):
        n, fb = self.n, self.frame_bytes
        strf = struct.pack('<IiiHHIIiiII', 40, self.w, self.h, 1, 24, 0, fb, 0, 0, 0, 0)
        strh = struct.pack('<4s4sIHHIIIIIIIIhhhh', b'vids', b'DIB ', 0, 0, 0, 0, 1, self.fps, 0, n, fb, 0xffffffff, 0, 0, 0, self.w, self.h)
        strl = b'LIST' + struct.pack('<I', 4 + 8 + len(strh) + 8 + len(strf)) + b'strl' + b'strh' + struct.pack('<I', len(strh)) + strh \
            + b'strf' + struct.pack('<I', len(strf)) + strf
        avih = struct.pack('<IIIIIIIIII4I', 1000000 // self.fps, fb * self.fps, 0, 0x10, n, 0, 1, fb, self.w, self.h, 0, 0, 0, 0)
        hdrl = b'LIST' + struct.pack('<I', 4 + 8 + len(avih) + len(strl)) + b'hdrl' + b'avih' + struct.pack('<I', len(avih)) + avih + strl
        movi = 4 + n * (8 + fb)
        self.f.seek(0)
        self.f.write(b'RIFF' + struct.pack('<I', 4 + len(hdrl) + 8 + movi) + b'AVI ' + hdrl + b'LIST' + struct.pack('<I', movi) + b'movi')
        self.data_start = self.f.tell()

    def write(self, rgb):
        frame = np.asarray(rgb, dtype=np.uint8)
        assert frame.shape == (self.h, self.w, 3)
        self.f.seek(self.data_start + self.n * (8 + self.frame_bytes))
        self.f.write(b'00db' + struct.pack('<I', self.frame_bytes) + frame[::-1, :, ::-1].tobytes())
        self.n += 1

    def release(self):
        self._header()           # frame count and sizes, now that they are known
        self.f.close()


class VideoMaker:
    """Method names, arguments and frame schedule of real_robots.videomaker.VideoMaker (videomaker.py:11-126): `intrinsic` /
    `extrinsic` are None / False, True, or an explicit collection (the reference takes `interval` objects: any container with
    `in` works -- ranges of steps, sets of trial numbers)."""

    def __init__(self, env, intrinsic=None, extrinsic=None, debug=False, writer=RawAviWriter):
        from .envs.env import EnvCamera
        self.env = env
        self.camera = EnvCamera(1.0, 90, -45, 0, [-0.3, 0, .4], fov=90, width=VIDEO_WIDTH, height=VIDEO_HEIGHT)
        self.seed = np.random.randint(100000)
        self.video_fps, self.speed_up = 25, 1
        self.frame_freq = int((200.0 / self.video_fps) * self.speed_up)       # one frame per 8 steps of 5 ms: real time at 25 fps
        self.debug, self._writer, self.video, self.files = debug, writer, None, []
        if intrinsic is True:
            intrinsic = self.get_intrinsic_frames()
        elif intrinsic and not hasattr(intrinsic, '__contains__'):
            raise Exception("VideoMaker intrinsic param has to be either None/False, a collection of steps or True")
        if extrinsic is True:
            extrinsic = self.get_extrinsic_trials()
        elif extrinsic and not hasattr(extrinsic, '__contains__'):
            raise Exception("VideoMaker extrinsic param has to be either None/False, a collection of trials or True")
        self.intrinsic_frames, self.extrinsic_trials = intrinsic or (), extrinsic or ()

    def get_intrinsic_frames(self):
        """The first, middle and last minute of the intrinsic phase (videomaker.py:58-63)."""
        n = int(self.env.intrinsic_timesteps)
        one_min = 60 * self.video_fps * self.frame_freq
        spans = [(0, one_min), (n // 2, n // 2 + one_min), (n - one_min, n)]
        return frozenset(s for a, b in spans for s in range(max(0, a), min(n, b) + 1))

    def get_extrinsic_trials(self):
        n = int(self.env.extrinsic_trials)
        return frozenset(np.random.choice(n, min(n, 5), replace=False).tolist()) if n > 0 else frozenset()

    def _open(self, suffix):
        name = "Simulation-{}-y{}-m{}-d{}-h{}-m{}-{}.avi".format(self.seed, *time.strftime("%Y,%m,%d,%H,%M").split(','), suffix)
        self.files.append(name)
        return self._writer(name, self.video_fps, VIDEO_WIDTH, VIDEO_HEIGHT)

    def start_intrinsic(self):
        if len(self.intrinsic_frames) > 0:
            self.video = self._open("intrinsic")

    def update_intrinsic(self, steps):
        if steps in self.intrinsic_frames and steps % self.frame_freq == 0:
            self.video.write(compose_frame(self.camera.render(self.env)))

    def end_intrinsic(self):
        if len(self.intrinsic_frames) > 0 and self.video:
            self.video.release()
            self.video = None

    makeInset = staticmethod(lambda image, text, right=False: make_inset(image, text))

    def start_trial(self, observation, trial_number):
        self.trial_number = trial_number
        if trial_number in self.extrinsic_trials:
            self.video = self._open("trial-{}".format(trial_number))
            self.goal = make_inset(observation['goal'], "GOAL")
            self.start = make_inset(observation['retina'], "START")

    def extrinsic_trial(self, observation, action, steps, score_object):
        if self.trial_number in self.extrinsic_trials and steps % self.frame_freq == 0:
            self.video.write(compose_frame(self.camera.render(self.env), self.goal, self.start))

    def end_trial(self):
        if self.trial_number in self.extrinsic_trials and self.video:
            self.video.release()
            self.video = None
