"""n-step transition assembly with the reference actors' exact bookkeeping
(core/single_processes/dqn_actor.py:55-61,110-163):

* windows hold nstep+2 states and nstep+1 actions/rewards/terminals;
* once three states are buffered, every step emits the *lagged* transition
  (s[0], a[0], sum_{i<len(r)-1} gamma^i r_i, gamma^(len(s)-2), s[-2], t[-2]) --
  one step late so that max_a Q(s[-2]) is known for the initial priority;
* when the episode ends exactly one more transition is emitted, reaching the
  true last state: from s[1] if the window is full, else from s[0];
* windows never straddle episodes;
* with PER the initial priority is |R + gamma^k max_a Q(s_k) - Q(s0, a0)| using
  the actor's own (stale) network.
"""
import ctypes
from collections import deque

import numpy as np


class NStepAssembler(object):
    def __init__(self, nstep, gamma, enable_per=False):
        self.nstep = int(nstep)
        self.gamma = float(gamma)
        self.enable_per = bool(enable_per)
        self.states = deque(maxlen=self.nstep + 2)
        self.actions = deque(maxlen=self.nstep + 1)
        self.rewards = deque(maxlen=self.nstep + 1)
        self.terminals = deque(maxlen=self.nstep + 1)
        self.qvalues = deque(maxlen=self.nstep + 1)
        self.max_qvalues = deque(maxlen=self.nstep + 1)

    def reset(self, state):
        for d in (self.states, self.actions, self.rewards, self.terminals, self.qvalues, self.max_qvalues):
            d.clear()
        self.states.append(state)

    def _ret(self, start, stop, shift):
        return float(sum(self.rewards[i] * self.gamma ** (i - shift) for i in range(start, stop)))

    def _priority(self, ret, gamma_sn, max_q, q0):
        if not self.enable_per or max_q is None or q0 is None:
            return 0.0
        return abs(ret + gamma_sn * max_q - q0)

    def push(self, action, reward, state1, terminal1, qvalue=None, max_qvalue=None):
        """Record one env step; returns [(transition, priority)] (0 or 1 items)."""
        self.states.append(state1)
        self.actions.append(action)
        self.rewards.append(reward)
        self.terminals.append(terminal1)
        self.qvalues.append(qvalue)
        self.max_qvalues.append(max_qvalue)
        if len(self.states) < 3:
            return []
        ret = self._ret(0, len(self.rewards) - 1, 0)
        gamma_sn = self.gamma ** (len(self.states) - 2)
        prio = self._priority(ret, gamma_sn, self.max_qvalues[-1], self.qvalues[0])
        return [((self.states[0], self.actions[0], [ret], [gamma_sn],
                  self.states[-2], self.terminals[-2]), prio)]

    def flush(self, max_qvalue_last=None):
        """Episode ended: emit the one closing transition."""
        if not self.actions:
            return []
        if len(self.states) >= self.nstep + 2:
            first = 1
            ret = self._ret(1, len(self.rewards), 1)
            gamma_sn = self.gamma ** (len(self.states) - 2)
        else:
            first = 0
            ret = self._ret(0, len(self.rewards), 0)
            gamma_sn = self.gamma ** (len(self.states) - 1)
        prio = self._priority(ret, gamma_sn, max_qvalue_last, self.qvalues[first])
        return [((self.states[first], self.actions[first], [ret], [gamma_sn],
                  self.states[-1], self.terminals[-1]), prio)]


class PythonFeeder(object):
    """NStepAssembler + memory.feed(): works with every memory type."""

    def __init__(self, memory, nstep, gamma, enable_per=False):
        self.memory = memory
        self.assembler = NStepAssembler(nstep, gamma, enable_per)

    def reset(self, state):
        self.assembler.reset(state)

    def push(self, action, reward, state1, terminal1, qvalue=None, max_qvalue=None):
        n = 0
        for transition, priority in self.assembler.push(action, reward, state1, terminal1, qvalue, max_qvalue):
            self.memory.feed(transition, priority)
            n += 1
        return n

    def flush(self, max_qvalue_last=None):
        n = 0
        for transition, priority in self.assembler.flush(max_qvalue_last):
            self.memory.feed(transition, priority)
            n += 1
        return n


class NativeFeeder(object):
    """The same bookkeeping in C (parallel/csrc/shm_ring.cpp, pda_nstep_*): the window of states
    lives in the native object and finished transitions go straight into the ingestion ring as
    packed records.  For ring-fed memories (memory_type "device*")."""

    def __init__(self, memory, nstep, gamma, enable_per=False, timeout_s=30.0):
        from ..parallel import hostlib
        self.L = hostlib.lib()
        c = memory.codec
        self.codec, self.ring = c, memory.ring
        self.state_np = c.state_np
        self.timeout_ms = int(timeout_s * 1000)
        self.dedup = bool(getattr(memory, "dedup", False))
        if self.dedup:
            # frame-deduplicated runs (parallel/csrc/frame_runs.h): the assembler sends each frame once
            rc = memory.run_codec
            self.h = self.L.pda_nstep_create_dedup(int(nstep), float(gamma), 1 if enable_per else 0, c.state_bytes,
                                                   rc.window, c.action_shape, c.num_tasks, rc.plane_cap, rc.trans_cap,
                                                   int(memory.max_age_ms))
        else:
            self.h = self.L.pda_nstep_create(int(nstep), float(gamma), 1 if enable_per else 0, c.state_bytes,
                                             c.state_stride, c.action_shape, c.num_tasks, c.record_bytes)
        if not self.h:
            raise ValueError("n-step assembler: inconsistent record layout")
        self._act = np.zeros(c.action_shape, dtype=np.float32)
        # per-step constants of push(): what the actor loop would otherwise re-derive for every env step (this object
        # is built inside the actor process, so the ring's mapping and these addresses stay put)
        self._act_ptr = self._act.ctypes.data
        self._ring_base = self.ring.base
        self._push = self.L.pda_nstep_push

    def _state(self, s):
        if type(s) is np.ndarray and s.dtype == self.state_np and s.flags.c_contiguous:
            return s
        return np.ascontiguousarray(s, dtype=self.state_np)

    def reset(self, state):
        s = self._state(state)
        self.L.pda_nstep_reset(self.h, s.ctypes.data)

    def _check(self, rc):
        if rc == -1:
            raise RuntimeError("replay ingestion ring stayed full for %d s (is a learner draining it?)"
                               % (self.timeout_ms // 1000))
        if rc < 0:
            raise ValueError("record larger than ring payload")
        return rc

    def push(self, action, reward, state1, terminal1, qvalue=None, max_qvalue=None):
        s = self._state(state1)
        self._act[:] = action.reshape(-1) if type(action) is np.ndarray else np.asarray(action).reshape(-1)
        if qvalue is not None and max_qvalue is not None:
            rc = self._push(self.h, self._ring_base, self._act_ptr, reward, s.ctypes.data, 1 if terminal1 else 0,
                            qvalue, max_qvalue, 1, self.timeout_ms)
        else:
            rc = self._push(self.h, self._ring_base, self._act_ptr, reward, s.ctypes.data, 1 if terminal1 else 0,
                            0.0, 0.0, 0, self.timeout_ms)
        return rc if rc >= 0 else self._check(rc)

    def flush(self, max_qvalue_last=None):
        has = max_qvalue_last is not None
        return self._check(self.L.pda_nstep_flush(self.h, self.ring.base, float(max_qvalue_last) if has else 0.0,
                                                  int(has), self.timeout_ms))

    def ship(self):
        """Send the open run now (frame-deduplicated records; a no-op otherwise)."""
        return self._check(self.L.pda_nstep_ship(self.h, self.ring.base, self.timeout_ms))

    def run_stats(self):
        out = (ctypes.c_uint64 * 3)()
        self.L.pda_nstep_run_stats(self.h, out)
        return {"runs": int(out[0]), "planes": int(out[1]), "transitions": int(out[2])}

    def __del__(self):
        try:
            if self.h:
                self.L.pda_nstep_destroy(self.h)
                self.h = None
        except Exception:
            pass


def make_feeder(memory, nstep, gamma, enable_per=False, native=True):
    if (native or getattr(memory, "dedup", False)) and hasattr(memory, "ring") and hasattr(memory, "codec"):
        return NativeFeeder(memory, nstep, gamma, enable_per)       # (deduplication happens in the native assembler)
    return PythonFeeder(memory, nstep, gamma, enable_per)
