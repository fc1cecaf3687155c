"""Distributed DDPG learner step at the reference's precision: fp32 end to end on gfx950's fp32-input matrix
instruction (csrc/mlp_f32.hip), selected by `precision="fp32"` (the option's default).

The reference learner is plain fp32 PyTorch (/root/reference/core/single_processes/ddpg_learner.py:66-95; models
/root/reference/core/models/ddpg_mlp_model.py:16-72).  `DdpgMlpEngine` (ops/mlp_engine.py) multiplies bf16 operands
(gradients within ~6 % of fp32); this engine runs the same step -- same replay binding, same head semantics, same
optimizer launch, same `MlpStepProgram` -- with fp32 operands everywhere:

  launch 1   {actor.0, actor'.0, critic.0 (both passes), critic'.0} reading the replay rows through the sampled
             indices; extra workgroups of the same launch copy s0, the replay action (into the critic's concat rows)
             and reward / discount / terminal of the sampled rows into compact buffers for the later launches
  launch 2   {actor.2, actor'.2}; its epilogue leaves per-tile partial dot products of actor.4 (1-17 outputs)
  launch 3   {critic.1 x 3 passes}: the loader finishes actor.4 (tanh(bias + partials)) into the concat columns --
             torch.cat is two column ranges of one row --; the epilogue leaves the partials of the Q layer
  launch 4   critic head in the prologue (Q values, n-step target, both losses, dL/dQ, PER leaves), then
             d[h|a] = (dL/dQ w3 tanh') W_c1 x tanh' (+ the pre-activation penalty on the actor rows)
  launch 5   dha1 = ((dz3 W_a4) tanh') W_a2 x tanh': the narrow layer's data gradient is formed in the loader
  launch 6   all six weight (+ bias) gradients as ONE job list, written in the master layout; bumps the step counters
  launch 7   the shared flat-buffer Adam launch (+ Polyak + next index draw); it also keeps the bf16 shadows of the
             actor layers current, which is what the hosted inference loop reads

7 launches per step, 76-817 workgroups each; 36 us per update at batch 64 = 27.5 k updates/s (the first, 11-launch
version of this engine: 59 us; the bf16 fused kernel: 68 us), profiles/r3_ddpg_fp32_step.txt.
"""
import ctypes

import torch

from . import lib
from .mlp_engine import DDPG_LAYERS, MlpWeights, _EngineBase, _p, _stream
from .optim import AdamExtras, FlatAdam, polyak_

_vp = ctypes.c_void_p
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
KIND_FWD, KIND_DGRAD, KIND_WGRAD = 0, 1, 2


class _Job(ctypes.Structure):           # mirrors F32Job in csrc/mlp_f32.hip
    _fields_ = [("a", _vp), ("a_si", ctypes.c_long), ("a_sr", ctypes.c_long),
                ("b", _vp), ("b_sr", ctypes.c_long), ("b_sj", ctypes.c_long),
                ("I", ctypes.c_int), ("J", ctypes.c_int), ("R", ctypes.c_int),
                ("c", _vp), ("c_ld", ctypes.c_long), ("c_col0", ctypes.c_int), ("c_scale", ctypes.c_float),
                ("bias", _vp), ("yin", _vp), ("yin_ld", ctypes.c_long),
                ("act", ctypes.c_int), ("kind", ctypes.c_int), ("gb", _vp),
                ("x_src", _vp), ("x_ld", ctypes.c_long), ("x_col0", ctypes.c_int), ("x_rows", ctypes.c_int),
                ("x_coef", ctypes.c_float), ("x_out", _vp), ("x_out_ld", ctypes.c_long),
                ("a_idx", _vp), ("a_mod", ctypes.c_int),
                ("nar_w", _vp), ("nar_ld", ctypes.c_long), ("nar_n", ctypes.c_int), ("npart", _vp),
                ("t_part", _vp), ("t_bias", _vp), ("t_z", _vp), ("t_col0", ctypes.c_int), ("t_rows", ctypes.c_int),
                ("t_tiles", ctypes.c_int), ("t_I", ctypes.c_int), ("t_n", ctypes.c_int),
                ("p_p", _vp), ("p_q", _vp), ("p_qld", ctypes.c_long), ("p_y", _vp), ("p_yld", ctypes.c_long),
                ("p_out", _vp), ("p_outld", ctypes.c_long),
                ("p_n", ctypes.c_int), ("p_act", ctypes.c_int), ("p_head", ctypes.c_int),
                ("tiles_j", ctypes.c_int), ("tiles_i", ctypes.c_int), ("tile0", ctypes.c_int)]


class _StageArgs(ctypes.Structure):     # mirrors F32StageArgs
    _fields_ = [("s0", _vp), ("ld_s", ctypes.c_long), ("tails", _vp), ("tail_ld", ctypes.c_int),
                ("off_action", ctypes.c_int), ("idx", _vp), ("B", ctypes.c_int), ("D", ctypes.c_int),
                ("A", ctypes.c_int), ("x0", _vp), ("ld_x0", ctypes.c_long),
                ("cat", _vp), ("ld_cat", ctypes.c_long), ("c1", ctypes.c_int),
                ("hd", _vp), ("off_reward", ctypes.c_int), ("off_gamma", ctypes.c_int), ("off_term", ctypes.c_int)]


class _HeadArgs(ctypes.Structure):      # mirrors F32Head
    _fields_ = [("qp_on", _vp), ("qp_tg", _vp), ("q_tiles", ctypes.c_int),
                ("b3_on", _vp), ("b3_tg", _vp),
                ("hd", _vp), ("idx", _vp), ("isw", _vp), ("dq", _vp),
                ("za", _vp), ("A", ctypes.c_int), ("actor_l2", ctypes.c_float),
                ("q_out", _vp), ("td_out", _vp), ("loss_sum", _vp), ("leaf_sum", _vp), ("leaf_min", _vp),
                ("prio_eps", ctypes.c_float), ("prio_alpha", ctypes.c_float),
                ("B", ctypes.c_int), ("leak", ctypes.c_int), ("actor_warmup", ctypes.c_int), ("opt_step", _vp)]


MAX_JOBS = 6


class _Launch(ctypes.Structure):        # mirrors F32Launch
    _fields_ = [("j", _Job * MAX_JOBS), ("n", ctypes.c_int), ("tiles", ctypes.c_int), ("stage_blocks", ctypes.c_int),
                ("pad_", ctypes.c_int), ("stage", _StageArgs), ("head", _HeadArgs),
                ("bump_counter", _vp), ("bump_step", _vp)]


_ready = False


def _L():
    global _ready
    L = lib()
    if not _ready:
        from . import mlp_engine
        mlp_engine._L()
        L.pda_mlp_f32_launch.argtypes = [_vp, _vp]
        L.pda_mlp_f32_launch.restype = ctypes.c_int
        L.pda_mlp_f32_job_bytes.restype = ctypes.c_int
        L.pda_mlp_f32_launch_bytes.restype = ctypes.c_int
        assert L.pda_mlp_f32_job_bytes() == ctypes.sizeof(_Job), (L.pda_mlp_f32_job_bytes(), ctypes.sizeof(_Job))
        assert L.pda_mlp_f32_launch_bytes() == ctypes.sizeof(_Launch), (L.pda_mlp_f32_launch_bytes(),
                                                                        ctypes.sizeof(_Launch))
        _ready = True
    return L


def _pad4(n):
    return (int(n) + 3) // 4 * 4


def _tiles(n):
    return (int(n) + 15) // 16


def fwd_job(x, w, bias, out, rows, N, K, out_col0=0, act=ACT_TANH, x_ld=None, gather=None, gather_mod=0, scale=1.0):
    """out[:, out_col0 : out_col0 + N] = act(scale * x[:rows, :K] @ w.T + bias); x / out: fp32 2-D buffers, w: [N][K].
    gather / gather_mod: row i of x is x[gather[i % gather_mod]] (replay rows through the sampled indices)."""
    j = _Job()
    j.c_scale = float(scale)
    j.a, j.a_si, j.a_sr = x.data_ptr(), (x.stride(0) if x_ld is None else int(x_ld)), 1
    j.b, j.b_sr, j.b_sj = w.data_ptr(), 1, w.stride(0)
    j.I, j.J, j.R = int(rows), int(N), int(K)
    j.c, j.c_ld, j.c_col0 = out.data_ptr(), out.stride(0), int(out_col0)
    j.bias = _p(bias)
    j.a_idx, j.a_mod = _p(gather), int(gather_mod)
    j.act, j.kind = act, KIND_FWD
    return j


def with_narrow(j, w_next, n_next, part):
    """The forward job also leaves, per tile column, its share of the next (narrow) layer's dot products:
    part[tile_j][i][k] = sum_{j in tile} y[i][j] w_next[k][j]."""
    assert part.numel() >= _tiles(j.J) * j.I * int(n_next)
    j.nar_w, j.nar_ld, j.nar_n, j.npart = w_next.data_ptr(), w_next.stride(0), int(n_next), part.data_ptr()
    return j


def with_tail(j, part, bias, col0, rows, tiles, part_rows, n, z=None):
    """Columns col0 .. col0 + n of the job's A rows < `rows` are tanh(bias + sum over tiles of part): the narrow layer
    finished in the loader (and stored back into A by the first tile column; z: its pre-activation)."""
    j.t_part, j.t_bias, j.t_z = part.data_ptr(), bias.data_ptr(), _p(z)
    j.t_col0, j.t_rows, j.t_tiles, j.t_I, j.t_n = int(col0), int(rows), int(tiles), int(part_rows), int(n)
    return j


def dgrad_job(dy, w, yin, dx, rows, Nout, Kin, act=ACT_TANH):
    """dx[:rows, :Kin] = (dy[:rows, :Nout] @ w) * act'(yin); w: [Nout][Kin] fp32 master.  dy None: the job gets a
    producer (with_producer) instead of an A operand in memory."""
    j = _Job()
    j.c_scale = 1.0
    if dy is not None:
        j.a, j.a_si, j.a_sr = dy.data_ptr(), dy.stride(0), 1
    j.b, j.b_sr, j.b_sj = w.data_ptr(), w.stride(0), 1
    j.I, j.J, j.R = int(rows), int(Kin), int(Nout)
    j.c, j.c_ld, j.c_col0 = dx.data_ptr(), dx.stride(0), 0
    if yin is not None:
        j.yin, j.yin_ld = yin.data_ptr(), yin.stride(0)
    j.act, j.kind = act, KIND_DGRAD
    return j


def with_producer(j, p, w_narrow, n, y, out, act=ACT_TANH, head=False):
    """A(i, r) = (sum_{k<n} p[i][k] w_narrow[k][r]) * act'(y[i][r]), stored to out by the first tile column;
    head=True: p is dL/dQ from the critic head that runs at the top of the job's workgroups (n = 1)."""
    j.p_p, j.p_q, j.p_qld = _p(p), w_narrow.data_ptr(), w_narrow.stride(0)
    j.p_y, j.p_yld, j.p_out, j.p_outld = y.data_ptr(), y.stride(0), out.data_ptr(), out.stride(0)
    j.p_n, j.p_act, j.p_head = int(n), act, 1 if head else 0
    return j


def wgrad_job(dy, x, gw, gb, rows, Nout, Kin, scale=1.0):
    """gw[Nout][Kin] = scale * dy[:rows, :Nout].T @ x[:rows, :Kin], gb[Nout] = column sums of dy (views of the flat
    gradient)."""
    j = _Job()
    j.c_scale = float(scale)
    j.a, j.a_si, j.a_sr = dy.data_ptr(), 1, dy.stride(0)
    j.b, j.b_sr, j.b_sj = x.data_ptr(), x.stride(0), 1
    j.I, j.J, j.R = int(Nout), int(Kin), int(rows)
    j.c, j.c_ld, j.c_col0 = gw.data_ptr(), int(Kin), 0
    j.gb = _p(gb)
    j.kind = KIND_WGRAD
    return j


class JobList(object):
    """A fixed list of jobs = one launch.  Built once (pointers are stable), launched every step."""

    def __init__(self, jobs, stage=None, stage_blocks=0, head=None, bump_counter=None, bump_step=None):
        assert 0 < len(jobs) <= MAX_JOBS
        self.launch = _Launch()
        self.launch.n = len(jobs)
        for k, j in enumerate(jobs):
            self.launch.j[k] = j
        if stage is not None:
            self.launch.stage, self.launch.stage_blocks = stage, int(stage_blocks)
        if head is not None:
            self.launch.head = head
        self.launch.bump_counter, self.launch.bump_step = _p(bump_counter), _p(bump_step)

    def run(self):
        rc = _L().pda_mlp_f32_launch(ctypes.cast(ctypes.pointer(self.launch), _vp), _stream())
        if rc != 0:
            raise RuntimeError("pda_mlp_f32_launch failed: %d" % rc)


class DdpgMlpEngineF32(_EngineBase):
    precision = "fp32"

    def __init__(self, model, target_model, batch_size, lr=1e-4, weight_decay=0.0, clip=40.0, tau=1e-3,
                 per=None, seed=0, tail_offsets=(0, 1, 2, 3), leak=True, actor_warmup=0, actor_l2=0.0):
        self.actor_warmup, self.actor_l2 = int(actor_warmup), float(actor_l2)
        # (the actor layers keep their bf16 shadows: the inference loop hosted in the learner reads them)
        self.online = MlpWeights(model, DDPG_LAYERS, with_grad=True, fp32_only=("critic.0.0", "critic.1.0", "critic.1.2"),
                                 opt_chunk=256)
        self.target = MlpWeights(target_model, DDPG_LAYERS, with_grad=False,
                                 fp32_only=("critic.0.0", "critic.1.0", "critic.1.2"), opt_chunk=256)
        on = self.online
        dev = on.flat.device
        self.H1, self.D, _ = on.shape["actor.0"]
        self.H2 = on.shape["actor.2"][0]
        self.A = on.shape["actor.4"][0]
        self.C1 = on.shape["critic.0.0"][0]
        self.C2, self.CK, _ = on.shape["critic.1.0"]
        assert self.CK == self.C1 + self.A and on.shape["critic.1.2"][0] == 1
        self._init_common(batch_size, dev, per, seed, tail_offsets, n_losses=2)
        self.tau, self.leak = float(tau), bool(leak)
        self.target_update = self.tau
        B = self.B
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        Dp, H1p, H2p, CKp, C2p = (_pad4(v) for v in (self.D, self.H1, self.H2, self.CK, self.C2))
        self.x0 = z(2 * B, Dp)
        self.ha1 = [z(B, H1p), z(B, H1p)]          # online(s0), target(s_n)
        self.ha2 = [z(B, H2p), z(B, H2p)]
        self.cat_on, self.cat_tg = z(2 * B, CKp), z(B, CKp)
        self.h2 = z(3 * B, C2p)
        self.za = z(B, self.A)                      # the online actor's pre-tanh output
        # narrow layers, finished by their consumers: per tile column partial dot products
        self.zpart = [z(_tiles(self.H2), B, self.A), z(_tiles(self.H2), B, self.A)]     # actor.4: online, target
        self.qpart = [z(_tiles(self.C2), 2 * B), z(_tiles(self.C2), B)]                 # critic.1.2: online x2, target
        self.dh2, self.dq = z(2 * B, C2p), z(2 * B, 1)
        self.dcat, self.dz3 = z(2 * B, CKp), z(B, self.A)
        self.dha2, self.dha1 = z(B, H2p), z(B, H1p)
        self.q = z(3 * B)
        self.hd = z(B, 4)                           # reward, discount, terminal flag of the sampled rows
        self.grads = model.flat_grads
        self.grads.zero_()
        self.gw, self.gb = on.grad_views(self.grads)
        self.adam = FlatAdam(on.flat, self.grads, on.table, lr=lr, weight_decay=weight_decay, clip=clip,
                             shadow=on.shadow)
        polyak_(self.target.flat, on.flat, 1.0, self.target.table, self.target.shadow)
        self.fused = True           # MlpStepProgram: fused_forward_backward + wgrad_all + one optimizer launch
        self._lists = {}
        _L()

    def bind_replay(self, *a, **k):
        self._lists = {}
        return super().bind_replay(*a, **k)

    def make_staging(self, *a, **k):
        self._lists = {}
        return super().make_staging(*a, **k)

    # ---- the step's launches, built once per source (replay through indices | staged batch) ------------
    def _build(self, staged):
        B, on, tg = self.B, self.online, self.target
        w, wt, b, bt = on.w32, tg.w32, on.b, tg.b
        D, H1, H2, A, C1, C2, CK = self.D, self.H1, self.H2, self.A, self.C1, self.C2, self.CK
        s0, s1, tails, idx, tree = self._sources(staged)
        oa, orw, og, ot = self.tail_offsets
        ld_s = int(s0.stride(0))
        st = _StageArgs(s0=_p(s0), ld_s=ld_s, tails=_p(tails), tail_ld=int(tails.stride(0)), off_action=oa,
                        idx=_p(idx), B=B, D=D, A=A, x0=_p(self.x0), ld_x0=self.x0.stride(0), cat=_p(self.cat_on),
                        ld_cat=self.cat_on.stride(0), c1=C1, hd=_p(self.hd), off_reward=orw, off_gamma=og, off_term=ot)
        g = dict(x_ld=ld_s, gather=idx, gather_mod=B)
        fwd1 = JobList([fwd_job(s0, w["actor.0"], b["actor.0"], self.ha1[0], B, H1, D, **g),
                        fwd_job(s1, wt["actor.0"], bt["actor.0"], self.ha1[1], B, H1, D, **g),
                        fwd_job(s0, w["critic.0.0"], b["critic.0.0"], self.cat_on, 2 * B, C1, D, **g),
                        fwd_job(s1, wt["critic.0.0"], bt["critic.0.0"], self.cat_tg, B, C1, D, **g)],
                       stage=st, stage_blocks=min(256, (B * (D + A + 3) + 255) // 256))
        fwd2 = JobList([with_narrow(fwd_job(self.ha1[0], w["actor.2"], b["actor.2"], self.ha2[0], B, H2, H1),
                                    w["actor.4"], A, self.zpart[0]),
                        with_narrow(fwd_job(self.ha1[1], wt["actor.2"], bt["actor.2"], self.ha2[1], B, H2, H1),
                                    wt["actor.4"], A, self.zpart[1])])
        t2 = _tiles(H2)
        fwd4 = JobList([
            with_narrow(with_tail(fwd_job(self.cat_on, w["critic.1.0"], b["critic.1.0"], self.h2, 2 * B, C2, CK),
                                  self.zpart[0], b["actor.4"], C1, B, t2, B, A, z=self.za),
                        w["critic.1.2"], 1, self.qpart[0]),
            with_narrow(with_tail(fwd_job(self.cat_tg, wt["critic.1.0"], bt["critic.1.0"], self.h2[2 * B:], B, C2, CK),
                                  self.zpart[1], bt["actor.4"], C1, B, t2, B, A),
                        wt["critic.1.2"], 1, self.qpart[1])])
        head = _HeadArgs(qp_on=_p(self.qpart[0]), qp_tg=_p(self.qpart[1]), q_tiles=_tiles(C2),
                         b3_on=_p(b["critic.1.2"]), b3_tg=_p(bt["critic.1.2"]),
                         hd=_p(self.hd), idx=_p(idx), isw=_p(self.isw) if self.per is not None else 0, dq=_p(self.dq),
                         za=_p(self.za), A=A, actor_l2=self.actor_l2, q_out=_p(self.q), td_out=_p(self.td),
                         loss_sum=_p(self.loss_acc),
                         leaf_sum=_p(tree.sum) if tree is not None else 0,
                         leaf_min=_p(tree.min) if tree is not None else 0,
                         prio_eps=float(self.per["eps"]) if self.per else 0.0,
                         prio_alpha=float(self.per["alpha"]) if self.per else 1.0,
                         B=B, leak=1 if self.leak else 0, actor_warmup=self.actor_warmup,
                         opt_step=_p(self.adam.step_dev))
        d1 = with_producer(dgrad_job(None, w["critic.1.0"], self.cat_on, self.dcat, 2 * B, C2, CK),
                           None, w["critic.1.2"], 1, self.h2, self.dh2, head=True)
        d1.x_col0, d1.x_rows = C1, B
        d1.x_out, d1.x_out_ld = self.dz3.data_ptr(), self.dz3.stride(0)
        if self.actor_l2 > 0:
            d1.x_src, d1.x_ld, d1.x_coef = self.za.data_ptr(), self.za.stride(0), 2.0 * self.actor_l2 / B
        bwd1 = JobList([d1], head=head)
        bwd3 = JobList([with_producer(dgrad_job(None, w["actor.2"], self.ha1[0], self.dha1, B, H2, H1),
                                      self.dz3, w["actor.4"], A, self.ha2[0], self.dha2)])
        r0, rows = (0, 2 * B) if self.leak else (B, B)      # critic: both passes (reference) or the critic loss only
        gw, gb = self.gw, self.gb
        wgrad = JobList([
            wgrad_job(self.dq[r0:], self.h2[r0:], gw["critic.1.2"], gb["critic.1.2"], rows, 1, C2),
            wgrad_job(self.dh2[r0:], self.cat_on[r0:], gw["critic.1.0"], gb["critic.1.0"], rows, C2, CK),
            wgrad_job(self.dcat[r0:], self.x0[r0:], gw["critic.0.0"], gb["critic.0.0"], rows, C1, D),
            wgrad_job(self.dz3, self.ha2[0], gw["actor.4"], gb["actor.4"], B, A, H2),
            wgrad_job(self.dha2, self.ha1[0], gw["actor.2"], gb["actor.2"], B, H2, H1),
            wgrad_job(self.dha1, self.x0, gw["actor.0"], gb["actor.0"], B, H1, D)],
            bump_counter=self.counter, bump_step=self.adam.step_dev)
        return {"fwdbwd": [fwd1, fwd2, fwd4, bwd1, bwd3], "wgrad": wgrad}

    def _get_lists(self, staged):
        key = bool(staged)
        if key not in self._lists:
            self._lists[key] = self._build(key)
        return self._lists[key]

    def fused_forward_backward(self, staged=False):
        for jl in self._get_lists(staged)["fwdbwd"]:
            jl.run()
        self._last_staged = bool(staged)

    def wgrad_all(self):
        self._get_lists(getattr(self, "_last_staged", False))["wgrad"].run()

    # (non-fused names, for callers that drive the pieces themselves)
    def forward_head(self, staged=False):
        self.fused_forward_backward(staged)

    def backward(self, staged=False):
        self.wgrad_all()

    def fused_extras(self, staged):
        x = AdamExtras()
        x.tgt, x.tgt_shadow, x.tau = _p(self.target.flat), _p(self.target.shadow), float(self.tau)
        if self.presamples(staged):
            x.next_idx, x.next_w, x.next_B = _p(self.idx), _p(self.isw), self.B
            x.size, x.seed, x.counter = _p(self.size_dev), self.seed, _p(self.counter)
        return x

    def presamples(self, staged):
        return (not staged) and self.replay is not None and not (self.replay[3] is not None and self.per is not None)

    def update(self, staged=False, tick=False, extras=None):
        if extras is not None:
            self.adam.step(tick=tick, extras=extras)
        else:
            self.adam.step(tick=tick)
            polyak_(self.target.flat, self.online.flat, self.tau, self.target.table, self.target.shadow)
        self._finish_update(staged)

    def hard_target_update(self):
        polyak_(self.target.flat, self.online.flat, 1.0, self.target.table, self.target.shadow)


# =====================================================================================
# MLP DQN at fp32 (dqn-mlp: D -> H -> H -> H -> A, relu)
# =====================================================================================
class DqnMlpEngineF32(_EngineBase):
    """Distributed-DQN learner step for the MLP Q-network in fp32 end to end -- the reference learner's arithmetic
    (/root/reference/core/single_processes/dqn_learner.py:60-96, core/models/dqn_mlp_model.py) -- on the same job-list
    kernels as DdpgMlpEngineF32:

      launch 1   critic.0 of {online(s0), target(s_n)[, online(s_n)]} reading the replay rows through the sampled
                 indices (x 1 / norm_val in the epilogue); extra workgroups stage s0 for the weight gradient
      launch 2-3 critic.2, critic.4 (the last one leaves the pre-activation: the head adds bias + relu)
      launch 4   dqn_head_kernel (csrc/rl_kernels.hip): fc head, double / plain n-step target, loss, dQ, PER leaves,
                 fp32 outputs
      launch 5-6 the two data gradients;  launch 7: all four weight (+ bias) gradients;  launch 8: Adam + index draw
    """
    precision = "fp32"

    def __init__(self, model, target_model, batch_size, lr=1e-4, weight_decay=0.0, clip=float("inf"), double=False,
                 target_update=250, per=None, seed=0, tail_offsets=(0, 1, 2, 3)):
        from .dqn_engine import HID_SUPPORTED
        from .mlp_engine import DQN_MLP_LAYERS
        self.online = MlpWeights(model, DQN_MLP_LAYERS, with_grad=True, fp32_only=DQN_MLP_LAYERS, opt_chunk=256)
        self.target = MlpWeights(target_model, DQN_MLP_LAYERS, with_grad=False, fp32_only=DQN_MLP_LAYERS,
                                 opt_chunk=256)
        on = self.online
        dev = on.flat.device
        self.HID, self.D, _ = on.shape["critic.0"]
        self.A = on.shape["critic.6"][0]
        if self.HID not in HID_SUPPORTED or self.A > 32 or self.HID % 4:
            raise ValueError("unsupported dqn-mlp head shape")
        self._init_common(batch_size, dev, per, seed, tail_offsets, n_losses=1, per_sample_loss=True)
        self.norm_val = float(model.norm_val)
        self.double = bool(double)
        self.target_update = target_update
        self.groups = 3 if self.double else 2
        B, H = self.B, self.HID
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.x0 = z(2 * B, _pad4(self.D))           # (the stage block writes rows b and B + b)
        self.hd = z(B, 4)
        self.h1 = [z(B, H) for _ in range(self.groups)]
        self.h2 = [z(B, H) for _ in range(self.groups)]
        self.h3pre = [z(B, H) for _ in range(self.groups)]
        self.h3, self.dh3 = z(B, H), z(B, H)
        self.dq_ld = max(16, (self.A + 7) // 8 * 8)
        self.dq = z(B, self.dq_ld)
        self.dh2, self.dh1 = z(B, H), z(B, H)
        self.q = z(B, self.A)
        self.grads = model.flat_grads
        self.grads.zero_()
        self.gw, self.gb = on.grad_views(self.grads)
        self.adam = FlatAdam(on.flat, self.grads, on.table, lr=lr, weight_decay=weight_decay, clip=clip,
                             shadow=on.shadow)
        polyak_(self.target.flat, on.flat, 1.0, self.target.table, self.target.shadow)
        self.fused = True
        self._lists = {}
        _L()
        from . import dqn_engine
        dqn_engine._L()

    def bind_replay(self, *a, **k):
        self._lists = {}
        return super().bind_replay(*a, **k)

    def make_staging(self, *a, **k):
        self._lists = {}
        return super().make_staging(*a, **k)

    def _build(self, staged):
        from .dqn_engine import _HeadArgs
        B, H, D, A, on, tg = self.B, self.HID, self.D, self.A, self.online, self.target
        s0, s1, tails, idx, tree = self._sources(staged)
        oa, orw, og, ot = self.tail_offsets
        ld_s, sc = int(s0.stride(0)), 1.0 / self.norm_val
        nets = [on, tg] + ([on] if self.double else [])
        srcs = [s0, s1] + ([s1] if self.double else [])
        st = _StageArgs(s0=_p(s0), ld_s=ld_s, tails=_p(tails), tail_ld=int(tails.stride(0)), off_action=oa,
                        idx=_p(idx), B=B, D=D, A=0, x0=_p(self.x0), ld_x0=self.x0.stride(0), cat=_p(self.x0),
                        ld_cat=self.x0.stride(0), c1=0, hd=_p(self.hd), off_reward=orw, off_gamma=og, off_term=ot)
        g = dict(x_ld=ld_s, gather=idx, gather_mod=B, act=ACT_RELU, scale=sc)
        fwd1 = JobList([fwd_job(src, n.w32["critic.0"], n.b["critic.0"], self.h1[i], B, H, D, **g)
                        for i, (n, src) in enumerate(zip(nets, srcs))],
                       stage=st, stage_blocks=min(256, (B * (D + 3) + 255) // 256))
        fwd2 = JobList([fwd_job(self.h1[i], n.w32["critic.2"], n.b["critic.2"], self.h2[i], B, H, H, act=ACT_RELU)
                        for i, n in enumerate(nets)])
        fwd3 = JobList([fwd_job(self.h2[i], n.w32["critic.4"], None, self.h3pre[i], B, H, H, act=ACT_NONE)
                        for i, n in enumerate(nets)])
        head = _HeadArgs(
            h1pre_on0=_p(self.h3pre[0]), h1pre_tg1=_p(self.h3pre[1]),
            h1pre_on1=_p(self.h3pre[2]) if self.double else 0,
            b4_on=_p(on.b["critic.4"]), b4_tg=_p(tg.b["critic.4"]),
            w5_on=_p(on.w32["critic.6"]), b5_on=_p(on.b["critic.6"]),
            w5_tg=_p(tg.w32["critic.6"]), b5_tg=_p(tg.b["critic.6"]),
            tails=_p(tails), tail_ld=int(tails.stride(0)), off_action=oa, off_reward=orw, off_gamma=og, off_term=ot,
            idx=_p(idx), isw=_p(self.isw) if self.per is not None else 0,
            h1_on0=0, dh1=0, dq=0, dq_ld=self.dq_ld,
            q_out=_p(self.q), td_out=_p(self.td), loss_sum=_p(self.loss_acc),
            leaf_sum=_p(tree.sum) if tree is not None else 0, leaf_min=_p(tree.min) if tree is not None else 0,
            prio_eps=float(self.per["eps"]) if self.per else 0.0,
            prio_alpha=float(self.per["alpha"]) if self.per else 1.0,
            B=B, HID=H, A=A, rng_counter=_p(self.counter), opt_step=_p(self.adam.step_dev),
            h1_f32=_p(self.h3), dh1_f32=_p(self.dh3), dq_f32=_p(self.dq))
        w = on.w32
        bwd1 = JobList([dgrad_job(self.dh3, w["critic.4"], self.h2[0], self.dh2, B, H, H, act=ACT_RELU)])
        bwd2 = JobList([dgrad_job(self.dh2, w["critic.2"], self.h1[0], self.dh1, B, H, H, act=ACT_RELU)])
        gw, gb = self.gw, self.gb
        wgrad = JobList([wgrad_job(self.dq, self.h3, gw["critic.6"], gb["critic.6"], B, A, H),
                         wgrad_job(self.dh3, self.h2[0], gw["critic.4"], gb["critic.4"], B, H, H),
                         wgrad_job(self.dh2, self.h1[0], gw["critic.2"], gb["critic.2"], B, H, H),
                         wgrad_job(self.dh1, self.x0, gw["critic.0"], gb["critic.0"], B, H, D, scale=sc)])
        return {"fwd": [fwd1, fwd2, fwd3], "head": head, "bwd": [bwd1, bwd2], "wgrad": wgrad}

    def _get_lists(self, staged):
        key = bool(staged)
        if key not in self._lists:
            self._lists[key] = self._build(key)
        return self._lists[key]

    def fused_forward_backward(self, staged=False):
        ls = self._get_lists(staged)
        for jl in ls["fwd"]:
            jl.run()
        lib().pda_dqn_head(ctypes.byref(ls["head"]), _stream())
        for jl in ls["bwd"]:
            jl.run()
        self._last_staged = bool(staged)

    def wgrad_all(self):
        self._get_lists(getattr(self, "_last_staged", False))["wgrad"].run()

    def forward_head(self, staged=False):
        self.fused_forward_backward(staged)

    def backward(self, staged=False):
        self.wgrad_all()

    def fused_extras(self, staged):
        x = AdamExtras()
        if self.presamples(staged):
            x.next_idx, x.next_w, x.next_B = _p(self.idx), _p(self.isw), self.B
            x.size, x.seed, x.counter = _p(self.size_dev), self.seed, _p(self.counter)
        return x

    def presamples(self, staged):
        return (not staged) and self.replay is not None and not (self.replay[3] is not None and self.per is not None)

    def update(self, staged=False, tick=False, extras=None):
        if extras is not None and extras.next_idx:
            self.adam.step(tick=tick, extras=extras)
        else:
            self.adam.step(tick=tick)
        if self.target_update < 1.0:
            polyak_(self.target.flat, self.online.flat, float(self.target_update), self.target.table,
                    self.target.shadow)
        self._finish_update(staged)

    def hard_target_update(self):
        polyak_(self.target.flat, self.online.flat, 1.0, self.target.table, self.target.shadow)
