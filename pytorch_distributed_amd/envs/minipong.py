"""MiniPong: a built-in pixel Pong, so that "wall-clock to Pong score" -- the one result the reference publishes
(/root/reference/README.md:5-10, assets/dqn_pong.png) -- has a stand-in that runs with no ALE and no ROM.

It is NOT Atari Pong (parity with the emulator is unpinned: neither atari_py nor ale_py is installed here).  It is a
game of the same shape, played through the same wrapper semantics as the reference's AtariEnv
(/root/reference/core/envs/atari_env.py:89-129):

* two paddles and a ball on an 84x84 court, the agent on the right, a scripted opponent on the left;
* Pong's minimal action set (6 actions: NOOP, FIRE, UP, DOWN, UP+FIRE, DOWN+FIRE);
* every agent step repeats the action for 4 emulator frames, sums the rewards, and shows the element-wise maximum
  of the last two frames; the state is the stack of the last 4 such frames, uint8, norm_val 255;
* reward +1 / -1 per point, first to 21 ends the episode (score range -21 .. +21), up to 30 random no-op frames at
  the start of an episode; there are no lives, so training and evaluation mode see the same terminals.

Physics (per emulator frame, all deterministic but the serve): the ball moves by (vx, vy); it bounces off the top
and bottom walls; a paddle returns it with a vertical speed set by WHERE it hit the paddle (centre: flat, edges:
steep) and every return speeds the rally up by 8 % (capped at 2x).  The opponent follows the ball only while it
approaches and only at `OPP_SPEED` pixels per frame, so flat returns come back forever and steep or fast ones score:
a policy has to learn first to meet the ball (stop losing 21-0) and then to hit it off-centre (start winning).
Scripted yardsticks (tests/test_minipong.py): random play loses ~20-0, a paddle that merely tracks the ball ends a few
points up after very long games, one that predicts the ball and meets it half-way off-centre wins by ~+18.
"""
from collections import deque

import numpy as np

from .env import Env

COURT = 84
TOP, BOTTOM = 8, 76                 # playing field rows [TOP, BOTTOM)
PADDLE_H, PADDLE_W = 12, 2
BALL = 2
AGENT_X, OPP_X = 74, 8              # left pixel column of each paddle
AGENT_SPEED = 2.0                   # pixels per emulator frame
OPP_SPEED = 1.6
BASE_VX = 1.6
MAX_VY = 2.4                        # vertical speed for a hit on the very edge of a paddle (at rally speed 1)
BG, WALL, PADDLE_A, PADDLE_O, BALL_V = 44, 120, 148, 147, 236
ACTION_DY = (0.0, 0.0, -1.0, 1.0, -1.0, 1.0)     # NOOP FIRE UP DOWN UPFIRE DOWNFIRE (Pong's minimal set)


class MiniPongGame(object):
    """The 'emulator': act(a) advances ONE frame and returns the reward; screen() renders 84x84 uint8."""

    def __init__(self, rng, points_to_win=21):
        self.rng = rng
        self.points_to_win = int(points_to_win)
        self.background = np.full((COURT, COURT), BG, dtype=np.uint8)
        self.background[TOP - 2:TOP, :] = WALL
        self.background[BOTTOM:BOTTOM + 2, :] = WALL
        self.reset_game()

    def reset_game(self):
        self.score_agent = self.score_opp = 0
        self.agent_y = self.opp_y = 0.5 * (TOP + BOTTOM - PADDLE_H)
        self._serve(toward_agent=bool(self.rng.randint(2)))

    def _serve(self, toward_agent):
        self.ball_x = 0.5 * (COURT - BALL)
        self.ball_y = float(self.rng.uniform(TOP + 8, BOTTOM - 8 - BALL))
        self.rally = 1.0
        self.vx = BASE_VX if toward_agent else -BASE_VX
        self.vy = float(self.rng.uniform(-1.0, 1.0))
        self.wait = 12                  # frames before the ball is put in play (as after a point in Pong)

    def game_over(self):
        return max(self.score_agent, self.score_opp) >= self.points_to_win

    @staticmethod
    def _clip_paddle(y):
        return float(min(max(y, TOP), BOTTOM - PADDLE_H))

    def _bounce_off(self, paddle_y):
        """New vertical speed after a paddle hit, by the ball's offset from the paddle centre."""
        off = (self.ball_y + 0.5 * BALL - (paddle_y + 0.5 * PADDLE_H)) / (0.5 * (PADDLE_H + BALL))
        self.rally = min(self.rally * 1.08, 2.0)
        return float(np.clip(off, -1.0, 1.0)) * MAX_VY * self.rally

    def act(self, a):
        self.agent_y = self._clip_paddle(self.agent_y + AGENT_SPEED * ACTION_DY[int(a)])
        # the opponent: follows the ball while it approaches, drifts back to the middle otherwise
        if self.vx < 0 and self.wait == 0:
            want = self.ball_y + 0.5 * BALL - 0.5 * PADDLE_H
        else:
            want = 0.5 * (TOP + BOTTOM - PADDLE_H)
        self.opp_y = self._clip_paddle(self.opp_y + float(np.clip(want - self.opp_y, -OPP_SPEED, OPP_SPEED)))
        if self.wait > 0:
            self.wait -= 1
            return 0.0
        self.ball_x += self.vx
        self.ball_y += self.vy
        if self.ball_y < TOP:
            self.ball_y, self.vy = 2 * TOP - self.ball_y, -self.vy
        elif self.ball_y > BOTTOM - BALL:
            self.ball_y, self.vy = 2 * (BOTTOM - BALL) - self.ball_y, -self.vy
        # paddles (the ball is returned when it reaches a paddle's face while overlapping it vertically)
        if self.vx > 0 and self.ball_x + BALL >= AGENT_X and self.ball_x + BALL - self.vx <= AGENT_X + PADDLE_W:
            if self.agent_y - BALL < self.ball_y < self.agent_y + PADDLE_H:
                self.vy = self._bounce_off(self.agent_y)
                self.vx = -BASE_VX * self.rally
                self.ball_x = AGENT_X - BALL
        elif self.vx < 0 and self.ball_x <= OPP_X + PADDLE_W and self.ball_x - self.vx >= OPP_X:
            if self.opp_y - BALL < self.ball_y < self.opp_y + PADDLE_H:
                self.vy = self._bounce_off(self.opp_y)
                self.vx = BASE_VX * self.rally
                self.ball_x = OPP_X + PADDLE_W
        if self.ball_x > COURT:             # past the agent
            self.score_opp += 1
            self._serve(toward_agent=True)
            return -1.0
        if self.ball_x < -BALL:             # past the opponent
            self.score_agent += 1
            self._serve(toward_agent=False)
            return 1.0
        return 0.0

    def screen(self):
        f = self.background.copy()
        ay, oy = int(round(self.agent_y)), int(round(self.opp_y))
        f[ay:ay + PADDLE_H, AGENT_X:AGENT_X + PADDLE_W] = PADDLE_A
        f[oy:oy + PADDLE_H, OPP_X:OPP_X + PADDLE_W] = PADDLE_O
        if self.wait == 0:
            bx, by = int(round(self.ball_x)), int(round(self.ball_y))
            if -BALL < bx < COURT:
                f[max(by, 0):by + BALL, max(bx, 0):bx + BALL] = BALL_V
        return f


class MiniPongEnv(Env):
    """Env face of MiniPongGame with the reference AtariEnv's step / reset semantics (see the module docstring)."""

    def __init__(self, args, process_ind=0, num_envs_per_process=1):
        super().__init__(args, process_ind, num_envs_per_process)
        assert (self.state_hei or COURT) == COURT and (self.state_wid or COURT) == COURT, "MiniPong renders 84x84"
        self.game_impl = MiniPongGame(self.rng, points_to_win=int(getattr(args, "minipong_points", 21)))
        self.frames = deque(maxlen=self.state_cha)

    @property
    def norm_val(self):
        return 255.0

    @property
    def state_shape(self):
        return [self.state_cha, COURT, COURT]

    @property
    def action_shape(self):
        return 1

    @property
    def action_space(self):
        return len(ACTION_DY)

    def _state(self):
        return np.stack(self.frames, axis=0)

    def reset(self):
        self._reset_experience()
        g = self.game_impl
        g.reset_game()
        for _ in range(int(self.rng.randint(30))):      # random no-op starts (atari_env.py:126-129)
            g.act(0)
        self.frames.clear()
        for _ in range(self.state_cha):
            self.frames.append(np.zeros((COURT, COURT), dtype=np.uint8))
        self.frames.append(g.screen())
        self.exp_state1 = self._state()
        return self._get_experience()

    def step(self, action):
        a = int(np.asarray(action).reshape(-1)[0])
        self.exp_action = np.asarray(action).reshape(self.action_shape)
        g = self.game_impl
        reward, done = 0.0, False
        prev = None
        for t in range(self.action_repetition):
            reward += g.act(a)
            if t == self.action_repetition - 2:
                prev = g.screen()
            done = g.game_over()
            if done:
                break
        last = g.screen()
        self.frames.append(last if prev is None else np.maximum(prev, last))
        self.exp_reward = reward
        self.exp_terminal1 = bool(done)
        self.exp_state1 = self._state()
        return self._get_experience()

    @property
    def score(self):
        return self.game_impl.score_agent, self.game_impl.score_opp
