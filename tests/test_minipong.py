"""MiniPong (envs/minipong.py): wrapper semantics as the reference's AtariEnv (/root/reference/core/envs/atari_env.py:
89-129: repeat 4, reward sum, max of the last two frames, 4-frame uint8 stack, +-1 per point, first to 21) and the
skill ladder the learning benchmark relies on -- random play loses, ball tracking is about even, aimed returns win."""
import numpy as np

from pytorch_distributed_amd.envs import minipong as mp
from pytorch_distributed_amd.utils.factory import EnvsDict
from pytorch_distributed_amd.utils.options import Options


def _env(seed=0):
    opt = Options(config=8)
    assert opt.env_type == "synthetic" and opt.game == "minipong"
    return EnvsDict[opt.env_type](opt.env_params, seed)


def _play(env, policy, games):
    scores = []
    for _ in range(games):
        exp = env.reset()
        total, done, n = 0.0, False, 0
        while not done:
            exp = env.step(np.array([[policy(env.game_impl, env.rng)]]))
            total += exp.reward
            done = exp.terminal1
            n += 1
            assert n < 50000
        sa, so = env.score
        assert max(sa, so) == 21 and total == sa - so        # first to 21; the rewards ARE the score
        scores.append(total)
    return float(np.mean(scores))


def _predict_y(g):
    t = (mp.AGENT_X - mp.BALL - g.ball_x) / g.vx
    lo, span = mp.TOP, mp.BOTTOM - mp.BALL - mp.TOP
    y = (g.ball_y + g.vy * t - lo) % (2 * span)
    return lo + (2 * span - y if y > span else y)


def _tracker(g, rng):
    d = g.ball_y + 1 - (g.agent_y + 6)
    return 0 if abs(d) < 2 else (3 if d > 0 else 2)


def _aimer(g, rng):
    if g.vx <= 0:
        d = 42 - (g.agent_y + 6)
    else:       # meet the ball 3.5 px off the paddle's centre, on the side that sends it away from the opponent
        off = 3.5 if g.opp_y + 6 > 42 else -3.5
        d = (_predict_y(g) + 1 + off) - (g.agent_y + 6)
    return 0 if abs(d) < 1.0 else (3 if d > 0 else 2)


def test_shapes_and_wrapper_semantics():
    env = _env()
    assert env.state_shape == [4, 84, 84] and env.action_space == 6 and env.action_shape == 1 and env.norm_val == 255.0
    exp = env.reset()
    assert exp.state1.shape == (4, 84, 84) and exp.state1.dtype == np.uint8
    assert not exp.state1[:3].any() and exp.state1[3].max() == mp.PADDLE_A      # three blank frames, then the court
    g = env.game_impl
    y0 = g.agent_y
    exp = env.step(np.array([[3]]))                                              # DOWN, repeated for 4 frames
    assert g.agent_y == min(y0 + 4 * mp.AGENT_SPEED, mp.BOTTOM - mp.PADDLE_H)
    assert exp.action.shape == (1,) and exp.reward == 0.0 and exp.terminal1 is False
    # the newest frame is the max of the last two emulator frames: the moving paddle leaves a 14-row trace
    col = exp.state1[3][:, mp.AGENT_X]
    assert int((col == mp.PADDLE_A).sum()) == mp.PADDLE_H + int(mp.AGENT_SPEED)
    # train / eval see the same game (no lives in Pong): both modes keep the action repeat of 4
    env.eval()
    assert env.action_repetition == 4


def test_skill_ladder_random_loses_tracking_is_even_aiming_wins():
    rnd = _play(_env(1), lambda g, rng: int(rng.randint(6)), 2)
    trk = _play(_env(2), _tracker, 2)
    aim = _play(_env(3), _aimer, 3)
    assert rnd <= -17, rnd
    assert -8 <= trk <= 14, trk
    assert aim >= 15 and aim > trk + 3, (aim, trk)


def test_deterministic_given_the_seed():
    a = _play(_env(5), _tracker, 1)
    b = _play(_env(5), _tracker, 1)
    assert a == b
