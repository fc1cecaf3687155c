"""Distributed DDPG learner step at the reference's precision: fp32 end to end on gfx950's fp32-input matrix
instruction (csrc/mlp_f32.hip), selected by `precision="fp32"` (the option's default).

The reference learner is plain fp32 PyTorch (/root/reference/core/single_processes/ddpg_learner.py:66-95; models
/root/reference/core/models/ddpg_mlp_model.py:16-72).  `DdpgMlpEngine` (ops/mlp_engine.py) multiplies bf16 operands
(gradients within ~6 % of fp32); this engine runs the same step -- same replay binding, same head semantics, same
optimizer launch, same `MlpStepProgram` -- with fp32 operands everywhere:

  stage      ONE launch gathers the sampled replay rows into padded fp32 operands (s0 twice, s_n, the replay action
             straight into the critic's concat rows)
  forward    4 launches, each a LIST of 16x16-tiled fp32 GEMMs over the master weights themselves:
             {actor.0, actor'.0, critic.0 (both passes), critic'.0} -> {actor.2, actor'.2} -> {actor.4, actor'.4}
             -> {critic.1 x 3 passes}; torch.cat is two column ranges of one row, as in the bf16 engine
  head       critic output layer, n-step target, both losses, dL/dh2 of both passes, PER leaves (one wave per sample)
  backward   d[h|a] = dh2 W_c1 (x tanh'; + the pre-activation penalty on the actor rows) -> dha2 -> dha1: 3 launches
  wgrad      all six weight (+ bias) gradients as ONE job list, written in the master layout
  update     the shared flat-buffer Adam launch (+ Polyak + next index draw); it also keeps the bf16 shadows of the
             actor layers current, which is what the hosted inference loop reads

11 launches per step, every one of 30-250 workgroups (the bf16 fused kernel: 4 workgroups).
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
                ("c", _vp), ("c_ld", ctypes.c_long), ("c_col0", ctypes.c_int),
                ("bias", _vp), ("z", _vp), ("z_ld", ctypes.c_long),
                ("yin", _vp), ("yin_ld", ctypes.c_long),
                ("act", ctypes.c_int), ("kind", ctypes.c_int), ("gb", _vp),
                ("x_src", _vp), ("x_ld", ctypes.c_long), ("x_col0", ctypes.c_int), ("x_rows", ctypes.c_int),
                ("x_coef", ctypes.c_float), ("x_gate", _vp), ("x_out", _vp), ("x_out_ld", ctypes.c_long),
                ("tiles_j", ctypes.c_int), ("tile0", ctypes.c_int), ("a_vec", ctypes.c_int), ("b_vec", ctypes.c_int)]


class _StageArgs(ctypes.Structure):     # mirrors F32StageArgs
    _fields_ = [("s0", _vp), ("s1", _vp), ("ld_s", ctypes.c_long), ("tails", _vp), ("tail_ld", ctypes.c_int),
                ("off_action", ctypes.c_int), ("idx", _vp), ("B", ctypes.c_int), ("D", ctypes.c_int),
                ("A", ctypes.c_int), ("x0", _vp), ("ld_x0", ctypes.c_long), ("x1", _vp), ("ld_x1", ctypes.c_long),
                ("cat", _vp), ("ld_cat", ctypes.c_long), ("c1", ctypes.c_int), ("scale", ctypes.c_float)]


class _HeadArgs(ctypes.Structure):      # mirrors DdpgHeadF32Args
    _fields_ = [("h2", _vp), ("ldh", ctypes.c_long),
                ("w3_on", _vp), ("b3_on", _vp), ("w3_tg", _vp), ("b3_tg", _vp),
                ("tails", _vp), ("tail_ld", ctypes.c_int), ("off_reward", ctypes.c_int), ("off_gamma", ctypes.c_int),
                ("off_term", ctypes.c_int), ("idx", _vp), ("isw", _vp), ("dh2", _vp), ("dq", _vp),
                ("za", _vp), ("A", ctypes.c_int), ("actor_l2", ctypes.c_float), ("gate", _vp),
                ("q_out", _vp), ("td_out", _vp), ("loss_sum", _vp), ("leaf_sum", _vp), ("leaf_min", _vp),
                ("prio_eps", ctypes.c_float), ("prio_alpha", ctypes.c_float),
                ("B", ctypes.c_int), ("H", ctypes.c_int), ("leak", ctypes.c_int), ("actor_warmup", ctypes.c_int),
                ("rng_counter", _vp), ("opt_step", _vp)]


_ready = False


def _L():
    global _ready
    L = lib()
    if not _ready:
        from . import mlp_engine
        mlp_engine._L()
        L.pda_mlp_f32_jobs.argtypes = [_vp, ctypes.c_int, _vp]
        L.pda_mlp_f32_jobs.restype = ctypes.c_int
        L.pda_mlp_f32_job_bytes.restype = ctypes.c_int
        assert L.pda_mlp_f32_job_bytes() == ctypes.sizeof(_Job), (L.pda_mlp_f32_job_bytes(), ctypes.sizeof(_Job))
        L.pda_mlp_f32_stage.argtypes = [ctypes.POINTER(_StageArgs), _vp]
        L.pda_mlp_f32_stage.restype = None
        L.pda_ddpg_head_f32.argtypes = [ctypes.POINTER(_HeadArgs), _vp]
        L.pda_ddpg_head_f32.restype = None
        _ready = True
    return L


def _pad4(n):
    return (int(n) + 3) // 4 * 4


def fwd_job(x, w, bias, out, rows, N, K, out_col0=0, act=ACT_TANH, z=None):
    """out[:, out_col0 : out_col0 + N] = act(x[:rows, :K] @ w.T + bias); x / out: padded fp32 2-D buffers, w: [N][K]."""
    j = _Job()
    j.a, j.a_si, j.a_sr = x.data_ptr(), x.stride(0), 1
    j.b, j.b_sr, j.b_sj = w.data_ptr(), 1, w.stride(0)
    j.I, j.J, j.R = int(rows), int(N), int(K)
    j.c, j.c_ld, j.c_col0 = out.data_ptr(), out.stride(0), int(out_col0)
    j.bias = _p(bias)
    if z is not None:
        j.z, j.z_ld = z.data_ptr(), z.stride(0)
    j.act, j.kind = act, KIND_FWD
    return j


def dgrad_job(dy, w, yin, dx, rows, Nout, Kin, act=ACT_TANH):
    """dx[:rows, :Kin] = (dy[:rows, :Nout] @ w) * act'(yin); w: [Nout][Kin] fp32 master."""
    j = _Job()
    j.a, j.a_si, j.a_sr = dy.data_ptr(), dy.stride(0), 1
    j.b, j.b_sr, j.b_sj = w.data_ptr(), w.stride(0), 1
    j.I, j.J, j.R = int(rows), int(Kin), int(Nout)
    j.c, j.c_ld, j.c_col0 = dx.data_ptr(), dx.stride(0), 0
    if yin is not None:
        j.yin, j.yin_ld = yin.data_ptr(), yin.stride(0)
    j.act, j.kind = act, KIND_DGRAD
    return j


def wgrad_job(dy, x, gw, gb, rows, Nout, Kin):
    """gw[Nout][Kin] = dy[:rows, :Nout].T @ x[:rows, :Kin], gb[Nout] = column sums of dy (views of the flat gradient)."""
    j = _Job()
    j.a, j.a_si, j.a_sr = dy.data_ptr(), 1, dy.stride(0)
    j.b, j.b_sr, j.b_sj = x.data_ptr(), x.stride(0), 1
    j.I, j.J, j.R = int(Nout), int(Kin), int(rows)
    j.c, j.c_ld, j.c_col0 = gw.data_ptr(), int(Kin), 0
    j.gb = _p(gb)
    j.kind = KIND_WGRAD
    return j


class JobList(object):
    """A fixed list of jobs = one launch.  Built once (pointers are stable), launched every step."""

    def __init__(self, jobs):
        assert 0 < len(jobs) <= 8
        self.n = len(jobs)
        self.arr = (_Job * self.n)(*jobs)

    def run(self):
        rc = _L().pda_mlp_f32_jobs(ctypes.cast(self.arr, _vp), self.n, _stream())
        if rc != 0:
            raise RuntimeError("pda_mlp_f32_jobs failed: %d" % rc)


class DdpgMlpEngineF32(_EngineBase):
    precision = "fp32"

    def __init__(self, model, target_model, batch_size, lr=1e-4, weight_decay=0.0, clip=40.0, tau=1e-3,
                 per=None, seed=0, tail_offsets=(0, 1, 2, 3), leak=True, actor_warmup=0, actor_l2=0.0):
        self.actor_warmup, self.actor_l2 = int(actor_warmup), float(actor_l2)
        # (the actor layers keep their bf16 shadows: the inference loop hosted in the learner reads them)
        self.online = MlpWeights(model, DDPG_LAYERS, with_grad=True, fp32_only=("critic.0.0", "critic.1.0", "critic.1.2"))
        self.target = MlpWeights(target_model, DDPG_LAYERS, with_grad=False,
                                 fp32_only=("critic.0.0", "critic.1.0", "critic.1.2"))
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
        self.x0, self.x1 = z(2 * B, Dp), z(B, Dp)
        self.ha1 = [z(B, H1p), z(B, H1p)]          # online(s0), target(s_n)
        self.ha2 = [z(B, H2p), z(B, H2p)]
        self.cat_on, self.cat_tg = z(2 * B, CKp), z(B, CKp)
        self.h2 = z(3 * B, C2p)
        self.za = z(B, self.A)                      # the online actor's pre-tanh output
        self.dh2, self.dq = z(2 * B, C2p), z(2 * B, 1)
        self.dcat, self.dz3 = z(2 * B, CKp), z(B, self.A)
        self.dha2, self.dha1 = z(B, H2p), z(B, H1p)
        self.gate = z(1)
        self.q = z(3 * B)
        self.grads = model.flat_grads
        self.grads.zero_()
        self.gw, self.gb = on.grad_views(self.grads)
        self.adam = FlatAdam(on.flat, self.grads, on.table, lr=lr, weight_decay=weight_decay, clip=clip,
                             shadow=on.shadow)
        polyak_(self.target.flat, on.flat, 1.0, self.target.table, self.target.shadow)
        self.fused = True           # MlpStepProgram: fused_forward_backward + wgrad_all + one optimizer launch
        self._lists = None
        _L()

    # ---- the step's launches, built once -----------------------------------------------------------
    def _build(self):
        B, on, tg = self.B, self.online, self.target
        w, wt, b, bt = on.w32, tg.w32, on.b, tg.b
        D, H1, H2, A, C1, C2, CK = self.D, self.H1, self.H2, self.A, self.C1, self.C2, self.CK
        x0a = self.x0[:B]
        fwd1 = JobList([fwd_job(x0a, w["actor.0"], b["actor.0"], self.ha1[0], B, H1, D),
                        fwd_job(self.x1, wt["actor.0"], bt["actor.0"], self.ha1[1], B, H1, D),
                        fwd_job(self.x0, w["critic.0.0"], b["critic.0.0"], self.cat_on, 2 * B, C1, D),
                        fwd_job(self.x1, wt["critic.0.0"], bt["critic.0.0"], self.cat_tg, B, C1, D)])
        fwd2 = JobList([fwd_job(self.ha1[0], w["actor.2"], b["actor.2"], self.ha2[0], B, H2, H1),
                        fwd_job(self.ha1[1], wt["actor.2"], bt["actor.2"], self.ha2[1], B, H2, H1)])
        fwd3 = JobList([fwd_job(self.ha2[0], w["actor.4"], b["actor.4"], self.cat_on, B, A, H2, out_col0=C1, z=self.za),
                        fwd_job(self.ha2[1], wt["actor.4"], bt["actor.4"], self.cat_tg, B, A, H2, out_col0=C1)])
        fwd4 = JobList([fwd_job(self.cat_on, w["critic.1.0"], b["critic.1.0"], self.h2, 2 * B, C2, CK),
                        fwd_job(self.cat_tg, wt["critic.1.0"], bt["critic.1.0"], self.h2[2 * B:], B, C2, CK)])
        d1 = dgrad_job(self.dh2, w["critic.1.0"], self.cat_on, self.dcat, 2 * B, C2, CK)
        d1.x_col0, d1.x_rows = C1, B
        d1.x_out, d1.x_out_ld = self.dz3.data_ptr(), self.dz3.stride(0)
        d1.x_gate = self.gate.data_ptr()
        if self.actor_l2 > 0:
            d1.x_src, d1.x_ld, d1.x_coef = self.za.data_ptr(), self.za.stride(0), 2.0 * self.actor_l2 / B
        bwd1 = JobList([d1])
        bwd2 = JobList([dgrad_job(self.dz3, w["actor.4"], self.ha2[0], self.dha2, B, A, H2)])
        bwd3 = JobList([dgrad_job(self.dha2, w["actor.2"], self.ha1[0], self.dha1, B, H2, H1)])
        r0, rows = (0, 2 * B) if self.leak else (B, B)      # critic: both passes (reference) or the critic loss only
        gw, gb = self.gw, self.gb
        wgrad = JobList([
            wgrad_job(self.dq[r0:], self.h2[r0:], gw["critic.1.2"], gb["critic.1.2"], rows, 1, C2),
            wgrad_job(self.dh2[r0:], self.cat_on[r0:], gw["critic.1.0"], gb["critic.1.0"], rows, C2, CK),
            wgrad_job(self.dcat[r0:], self.x0[r0:], gw["critic.0.0"], gb["critic.0.0"], rows, C1, D),
            wgrad_job(self.dz3, self.ha2[0], gw["actor.4"], gb["actor.4"], B, A, H2),
            wgrad_job(self.dha2, self.ha1[0], gw["actor.2"], gb["actor.2"], B, H2, H1),
            wgrad_job(self.dha1, self.x0, gw["actor.0"], gb["actor.0"], B, H1, D)])
        self._lists = {"fwd": [fwd1, fwd2, fwd3, fwd4], "bwd": [bwd1, bwd2, bwd3], "wgrad": wgrad}

    def fused_forward_backward(self, staged=False):
        if self._lists is None:
            self._build()
        B, on, tg = self.B, self.online, self.target
        s0, s1, tails, idx, tree = self._sources(staged)
        oa, orw, og, ot = self.tail_offsets
        st = _StageArgs(s0=_p(s0), s1=_p(s1), ld_s=int(s0.stride(0)), tails=_p(tails), tail_ld=int(tails.stride(0)),
                        off_action=oa, idx=_p(idx), B=B, D=self.D, A=self.A, x0=_p(self.x0),
                        ld_x0=self.x0.stride(0), x1=_p(self.x1), ld_x1=self.x1.stride(0), cat=_p(self.cat_on),
                        ld_cat=self.cat_on.stride(0), c1=self.C1, scale=1.0)
        L = _L()
        L.pda_mlp_f32_stage(ctypes.byref(st), _stream())
        for jl in self._lists["fwd"]:
            jl.run()
        h = _HeadArgs(h2=_p(self.h2), ldh=self.h2.stride(0),
                      w3_on=_p(on.w32["critic.1.2"]), b3_on=_p(on.b["critic.1.2"]),
                      w3_tg=_p(tg.w32["critic.1.2"]), b3_tg=_p(tg.b["critic.1.2"]),
                      tails=_p(tails), tail_ld=int(tails.stride(0)), off_reward=orw, off_gamma=og, off_term=ot,
                      idx=_p(idx), isw=_p(self.isw) if self.per is not None else 0,
                      dh2=_p(self.dh2), dq=_p(self.dq), za=_p(self.za), A=self.A, actor_l2=self.actor_l2,
                      gate=_p(self.gate), q_out=_p(self.q), td_out=_p(self.td), loss_sum=_p(self.loss_acc),
                      leaf_sum=_p(tree.sum) if tree is not None else 0, leaf_min=_p(tree.min) if tree is not None else 0,
                      prio_eps=float(self.per["eps"]) if self.per else 0.0,
                      prio_alpha=float(self.per["alpha"]) if self.per else 1.0,
                      B=B, H=self.C2, leak=1 if self.leak else 0, actor_warmup=self.actor_warmup,
                      rng_counter=_p(self.counter), opt_step=_p(self.adam.step_dev))
        L.pda_ddpg_head_f32(ctypes.byref(h), _stream())
        for jl in self._lists["bwd"]:
            jl.run()

    def wgrad_all(self):
        self._lists["wgrad"].run()

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
