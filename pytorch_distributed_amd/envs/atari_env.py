"""Atari via ALE with the DeepMind-style preprocessing the reference uses
(core/envs/atari_env.py:19-129): grayscale -> 84x84 bilinear, 4-frame stack,
action repeat 4 with a max over the last two frames, up to 30 no-op starts,
loss-of-life terminals in training mode only, sticky actions off, episode cap
`early_stop`.

ALE (`ale_py` or `atari_py`) and OpenCV are optional imports: the class raises
a clear ImportError when no backend is present, and `make_atari_env` (what the
factory registers) then falls back to SyntheticAtariEnv with a warning so that
`python main.py` stays runnable on a box without ROMs.
"""
import warnings

import numpy as np

from .env import Env


def _find_ale():
    try:
        import ale_py  # noqa: F401
        return "ale_py"
    except Exception:
        pass
    try:
        import atari_py  # noqa: F401
        return "atari_py"
    except Exception:
        return None


def resize_bilinear_u8(img, out_h, out_w):
    """Bilinear resize matching cv2.INTER_LINEAR's half-pixel sampling; used
    when OpenCV is not installed."""
    try:
        import cv2
        return cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR)
    except Exception:
        pass
    in_h, in_w = img.shape[:2]
    ys = np.clip((np.arange(out_h) + 0.5) * in_h / out_h - 0.5, 0, in_h - 1)
    xs = np.clip((np.arange(out_w) + 0.5) * in_w / out_w - 0.5, 0, in_w - 1)
    y0, x0 = np.floor(ys).astype(np.int64), np.floor(xs).astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, in_h - 1), np.minimum(x0 + 1, in_w - 1)
    wy, wx = (ys - y0)[:, None], (xs - x0)[None, :]
    f = img.astype(np.float32)
    top = f[y0][:, x0] * (1 - wx) + f[y0][:, x1] * wx
    bot = f[y1][:, x0] * (1 - wx) + f[y1][:, x1] * wx
    return np.clip(np.rint(top * (1 - wy) + bot * wy), 0, 255).astype(np.uint8)


def _open_ale(game):
    """-> (ALEInterface, rom path) from whichever ALE binding is installed."""
    backend = _find_ale()
    if backend is None:
        raise ImportError("AtariEnv needs ale_py or atari_py; neither is installed")
    if backend == "atari_py":
        import atari_py
        return atari_py.ALEInterface(), atari_py.get_game_path(game)
    import ale_py
    import ale_py.roms as roms
    return ale_py.ALEInterface(), getattr(roms, game.capitalize())


class AtariEnv(Env):
    """The wrapper is three pieces: a frame stack kept as ONE [C, H, W] uint8 array that is shifted in place (the
    observation handed out is a copy of it: no per-step list of arrays, no np.stack), `_repeat` = the action-repeat /
    frame-max unit, and the life bookkeeping that turns a lost life into a training-time terminal followed by a
    no-op "soft reset".

    `ale` may be any object with the ALEInterface methods used here (setInt/setFloat/setBool, loadROM,
    getMinimalActionSet, act, game_over, lives, reset_game, getScreenGrayscale, getScreenRGB): the tests drive
    the wrapper with a scripted one (tests/test_atari_env.py), production passes None and opens the emulator."""

    REPEAT = 4              # emulator frames per agent step (reference atari_env.py:95)
    POOLED = 2              # ... of which the last POOLED are max-pooled into the new observation plane
    MAX_NOOP_STARTS = 30    # reference atari_env.py:123

    def __init__(self, args, process_ind=0, num_envs_per_process=1, ale=None):
        self._stack = None
        super().__init__(args, process_ind, num_envs_per_process)
        assert self.num_envs_per_process == 1
        self.ale, rom = (ale, self.game) if ale is not None else _open_ale(self.game)
        for setter, key, value in ((self.ale.setInt, "random_seed", self.seed),
                                   (self.ale.setInt, "max_num_frames", self.early_stop),
                                   (self.ale.setFloat, "repeat_action_probability", 0),     # sticky actions off
                                   (self.ale.setInt, "frame_skip", 0),
                                   (self.ale.setBool, "color_averaging", False)):
            setter(key, value)
        self.ale.loadROM(rom)                               # after the options
        self.actions = dict(enumerate(self.ale.getMinimalActionSet()))
        self.lives = 0
        self._soft_reset_due = False        # a life was lost in training mode and the game goes on
        self._stack = np.zeros((self.state_cha, self.state_hei, self.state_wid), dtype=np.uint8)
        self._pool = np.zeros((self.POOLED, self.state_hei, self.state_wid), dtype=np.uint8)

    # ---- the reference's attribute names, for code and tests written against them ----
    @property
    def just_died(self):
        return self._soft_reset_due

    @property
    def exp_state1(self):
        return self._stack

    @exp_state1.setter
    def exp_state1(self, value):            # Env._reset_experience assigns None: the stack is cleared instead
        if self._stack is not None:
            self._stack[:] = 0 if value is None else value

    # ---- frames ----
    def _capture_ale(self):
        return resize_bilinear_u8(self.ale.getScreenGrayscale().squeeze(), self.state_hei, self.state_wid)

    def _push_plane(self, plane):
        s = self._stack
        s[:-1] = s[1:]
        s[-1] = plane

    def _preprocess_state(self, state):
        return np.array(state, dtype=np.uint8, copy=True)

    def _preprocess_action(self, action):
        return np.asarray(action).reshape(self.action_shape)

    @property
    def norm_val(self):
        return 255.0

    @property
    def state_shape(self):
        return [self.state_cha, self.state_hei, self.state_wid]

    @property
    def action_shape(self):
        return 1

    @property
    def action_space(self):
        return len(self.actions)

    def render(self):
        """Show the emulator's RGB screen (reference atari_env.py:82-84); a no-op without OpenCV / a display."""
        try:
            import cv2
            cv2.imshow("screen", self.ale.getScreenRGB()[:, :, ::-1])
            cv2.waitKey(1)
            return True
        except Exception:
            return False

    def close(self):
        """reference atari_env.py:86-87"""
        try:
            import cv2
            cv2.destroyAllWindows()
        except Exception:
            pass

    # ---- one agent step ----
    def _repeat(self, ale_action):
        """REPEAT emulator frames of one action, stopping at game over -> (summed reward, game over).  The screens of
        the last POOLED repeats land in self._pool (a repeat that never ran leaves its plane black)."""
        self._pool[:] = 0
        total, over = 0.0, False
        first_pooled = self.REPEAT - self.POOLED
        for k in range(self.REPEAT):
            total += self.ale.act(ale_action)
            if k >= first_pooled:
                self._pool[k - first_pooled] = self._capture_ale()
            over = bool(self.ale.game_over())
            if over:
                break
        return total, over

    def step(self, action):
        self.exp_action = self._preprocess_action(action)
        index = int(np.asarray(action).reshape(-1)[0])
        self.exp_reward, over = self._repeat(self.actions.get(index))
        self._push_plane(self._pool.max(axis=0))
        terminal = over
        if self.training:           # a lost life ends the episode for the learner, not for the emulator
            now = self.ale.lives()
            if 0 < now < self.lives:
                self._soft_reset_due = not over
                terminal = True
            self.lives = now
        self.exp_terminal1 = terminal
        return self._get_experience()

    def reset(self):
        if self._soft_reset_due:    # the game goes on: one no-op, the frame stack is kept
            self._soft_reset_due = False
            self.ale.act(0)
        else:
            self._reset_experience()
            self.ale.reset_game()
            noops = int(self.rng.randint(self.MAX_NOOP_STARTS))
            while noops > 0:
                noops -= 1
                self.ale.act(0)     # raw action 0 is the no-op
                if self.ale.game_over():
                    self.ale.reset_game()
        self._push_plane(self._capture_ale())
        self.lives = self.ale.lives()
        return self._get_experience()


def make_atari_env(args, process_ind=0, num_envs_per_process=1):
    """Factory entry for env_type "atari": real ALE when importable, otherwise
    Atari-shaped synthetic frames (with a warning)."""
    if _find_ale() is not None:
        return AtariEnv(args, process_ind, num_envs_per_process)
    from .synthetic_env import SyntheticAtariEnv
    if process_ind == 0:
        warnings.warn("ALE is not installed: env_type 'atari' is using SyntheticAtariEnv frames")
    return SyntheticAtariEnv(args, process_ind, num_envs_per_process)
