"""AtariEnv's wrapper semantics against a scripted emulator (no ROMs here): what the reference's wrapper does in
/root/reference/core/envs/atari_env.py:89-129 -- action repeat 4 with the max over the frames of repeats 3 and 4,
rewards summed over the repeats, early exit on game over, loss of life = terminal in training mode only (and never
when the count reaches 0), a single no-op instead of a game reset after a lost life, up to 30 no-op starts, and a
4-frame uint8 stack that a full reset zeroes."""
import numpy as np
import pytest

from pytorch_distributed_amd.envs.atari_env import AtariEnv
from pytorch_distributed_amd.utils.options import Options


class FakeALE(object):
    """Scripted ALEInterface: frame number f shows a 210x160 screen filled with value f; life losses and the game
    over arrive at the emulator frames given."""

    def __init__(self, lives=3, lose_life_at=(), game_over_at=None, reward_at=None):
        self.start_lives, self.lose_life_at, self.game_over_at = lives, set(lose_life_at), game_over_at
        self.reward_at = reward_at or {}
        self.options, self.rom, self.acts, self.resets = {}, None, [], 0
        self.frame, self._lives, self._over = 0, lives, False

    def setInt(self, k, v):
        assert self.rom is None, "options go in before the ROM (reference atari_env.py:26)"
        self.options[k] = v
    setFloat = setBool = setInt

    def loadROM(self, rom):
        self.rom = rom

    def getMinimalActionSet(self):
        return [0, 1, 3, 4, 11, 12]

    def act(self, a):
        assert not self._over, "act() after game over without a reset"
        self.acts.append(a)
        self.frame += 1
        if self.frame in self.lose_life_at:
            self._lives -= 1
        if self.game_over_at is not None and self.frame >= self.game_over_at:
            self._over = True
        return self.reward_at.get(self.frame, 0.0)

    def game_over(self):
        return self._over

    def lives(self):
        return self._lives

    def reset_game(self):
        self.resets += 1
        self.frame, self._lives, self._over = 0, self.start_lives, False
        self.acts.append("reset")

    def getScreenGrayscale(self):
        # bright on odd frames, dark on even ones: the max over two consecutive frames is the odd one
        v = (100 + self.frame) if self.frame % 2 else self.frame
        return np.full((210, 160, 1), min(v, 255), dtype=np.uint8)

    def getScreenRGB(self):
        return np.zeros((210, 160, 3), dtype=np.uint8)


def make_env(ale, seed=5):
    p = Options(config=2).env_params          # an atari config: 4 x 84 x 84 states, early_stop 12500
    p.seed = seed
    env = AtariEnv(p, 0, 1, ale=ale)
    return env


def action(i):
    return np.array([[i]])


def test_options_are_set_before_the_rom_and_the_action_set_is_minimal():
    ale = FakeALE()
    env = make_env(ale)
    assert ale.rom is not None and ale.options["repeat_action_probability"] == 0 and ale.options["frame_skip"] == 0
    assert ale.options["color_averaging"] is False and ale.options["max_num_frames"] == env.early_stop
    assert env.action_space == 6 and env.action_shape == 1 and env.state_shape == [4, 84, 84] and env.norm_val == 255.0


def test_reset_zeroes_the_stack_plays_up_to_30_noops_and_appends_one_frame():
    ale = FakeALE()
    env = make_env(ale)
    seen = set()
    for _ in range(40):
        n0 = len(ale.acts)
        exp = env.reset()
        played = ale.acts[n0:]
        assert played[0] == "reset" and all(a == 0 for a in played[1:]) and len(played) - 1 < 30
        seen.add(len(played) - 1)
        s = exp.state1
        assert s.dtype == np.uint8 and s.shape == (4, 84, 84)
        assert not s[:3].any()                              # a full reset clears the older frames
        f = ale.frame
        assert (s[3] == ((100 + f) if f % 2 else f)).all()  # the newest slot is the current screen
        assert exp.state0 is None and exp.reward is None and exp.terminal1 is None
        assert env.lives == 3
    assert len(seen) > 10 and min(seen) >= 0 and max(seen) <= 29


def test_step_repeats_the_action_4_times_sums_the_rewards_and_keeps_the_max_of_the_last_two_frames():
    ale = FakeALE(reward_at={})
    env = make_env(ale)
    env.reset()
    f0 = ale.frame
    ale.reward_at = {f0 + 1: 1.0, f0 + 2: 0.5, f0 + 4: 2.0, f0 + 5: 100.0}
    n0 = len(ale.acts)
    exp = env.step(action(2))
    assert ale.acts[n0:] == [3, 3, 3, 3]                    # index 2 of the minimal set, four times
    assert exp.reward == 3.5 and exp.terminal1 is False
    assert exp.action.shape == (1,) or exp.action.shape == () or exp.action.size == 1
    f3, f4 = f0 + 3, f0 + 4
    val = lambda f: (100 + f) if f % 2 else f
    assert (exp.state1[3] == max(val(f3), val(f4))).all()   # max over repeats 3 and 4, not the last frame alone
    prev = env.exp_state1[2]
    assert (exp.state1[2] == prev).all() and exp.state1.shape == (4, 84, 84)


def test_game_over_ends_the_repeat_loop_early_and_is_terminal_in_both_modes():
    for training in (True, False):
        ale = FakeALE(game_over_at=10 ** 9)
        env = make_env(ale)
        env.train() if training else env.eval()
        env.reset()
        ale.game_over_at = ale.frame + 2
        n0 = len(ale.acts)
        exp = env.step(action(0))
        assert len(ale.acts) - n0 == 2 and exp.terminal1 is True and env.just_died is False
        n0 = len(ale.acts)
        ale.game_over_at = 10 ** 9
        env.reset()
        assert ale.acts[n0] == "reset"                       # a real game over: full reset


def test_life_loss_is_terminal_only_in_training_and_is_followed_by_a_noop_not_a_reset():
    ale = FakeALE(lives=3)
    env = make_env(ale)
    env.train()
    env.reset()
    ale.lose_life_at = {ale.frame + 2}
    exp = env.step(action(1))
    assert exp.terminal1 is True and env.just_died is True and env.lives == 2
    stack_before = exp.state1.copy()
    n0, resets = len(ale.acts), ale.resets
    exp = env.reset()
    assert ale.acts[n0:] == [0] and ale.resets == resets     # one no-op, the game goes on
    assert env.just_died is False and env.lives == 2
    assert (exp.state1[:3] == stack_before[1:]).all()        # the stack is kept, one frame appended
    # the next step without a life loss is not terminal
    assert env.step(action(1)).terminal1 is False

    ale = FakeALE(lives=3)
    env = make_env(ale)
    env.eval()
    env.reset()
    ale.lose_life_at = {ale.frame + 2}
    exp = env.step(action(1))
    assert exp.terminal1 is False and env.just_died is False  # evaluation: only game over ends an episode


def test_losing_the_last_life_is_left_to_game_over():
    # lives going to 0 is not a "life lost" terminal (reference: `lives > 0`, the Q*bert case): game_over decides
    ale = FakeALE(lives=1)
    env = make_env(ale)
    env.train()
    env.reset()
    ale.lose_life_at = {ale.frame + 1}
    exp = env.step(action(0))
    assert exp.terminal1 is False and env.just_died is False and env.lives == 0
    # life lost and game over in the same step: terminal, but no just_died (a full reset follows)
    ale = FakeALE(lives=2)
    env = make_env(ale)
    env.train()
    env.reset()
    ale.lose_life_at = {ale.frame + 1}
    ale.game_over_at = ale.frame + 1
    exp = env.step(action(0))
    assert exp.terminal1 is True and env.just_died is False


def test_render_and_close_do_not_need_a_display():
    env = make_env(FakeALE())
    assert env.render() in (True, False)
    env.close()


def test_resize_matches_half_pixel_bilinear_on_a_ramp():
    from pytorch_distributed_amd.envs.atari_env import resize_bilinear_u8
    img = np.tile(np.arange(160, dtype=np.uint8)[None, :], (210, 1))
    out = resize_bilinear_u8(img, 84, 84)
    assert out.shape == (84, 84) and out.dtype == np.uint8
    want = np.clip((np.arange(84) + 0.5) * 160 / 84 - 0.5, 0, 159)
    assert np.abs(out[10].astype(np.float64) - want).max() <= 1.0
