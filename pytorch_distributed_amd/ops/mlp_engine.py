"""MLP agents on hand-written gfx950 kernels: DDPG actor/critic and MLP-DQN
learner steps plus batched actor inference (kernels K5, K6, K12-K16 of
SURVEY.md section 2.3).

The reference DDPG step (core/single_processes/ddpg_learner.py:50-106) is two
autograd passes of ~60 tiny eager launches behind a host-side batch gather.
Here a step is ~20 launches captured in one hipGraph, all on data resident in HBM:

  sample    uniform counter-hash draw or radix-64 sum-tree descent (PER)
  stage     ONE launch gathers the sampled fp32 replay rows into padded bf16
            GEMM operands (s0, s_n, and the replay action straight into the
            critic's concat buffer)
  forward   dense layers on the MFMA GEMM family; online and target networks
            (and the two online critic passes) are problem groups of the SAME
            launch.  torch.cat((h, a)) is never materialised: the first critic
            layer and the actor's last layer write side by side into one padded
            [B][400+A] buffer that the second critic layer reads.
  head      critic output layer + n-step TD target + actor loss + weighted
            critic loss + the two dL/dh that start the backward passes + PER
            leaves, one wave64 per sample (csrc/mlp_kernels.hip)
  backward  the actor-loss pass Q(s0, pi(s0)) and the critic-loss pass Q(s0, a)
            are stacked into ONE 2B-row backward through the critic, which is
            exactly the reference's gradient accumulation (its actor loss
            leaves gradients in the critic, ddpg_learner.py:66-86); with
            compat_ddpg_grad_leak=False the critic weight gradients only read
            the critic-loss rows.  The tanh' mask of the concat buffer's action
            columns is the actor's output tanh', so dgrad of the second critic
            layer lands directly on the actor's last pre-activation.
  update    fused clip(+-40)+Adam, fused Polyak, tree repair

MLP-DQN (dqn-mlp, D->256->256->256->A) reuses the DQN head kernel with HID=256.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib, nn
from .optim import FlatAdam, SegmentTable, polyak_

_vp = ctypes.c_void_p
BF16 = torch.bfloat16


def _p(t):
    return t.data_ptr() if t is not None else 0


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _pad8(n):
    return (int(n) + 7) // 8 * 8


class _StageJob(ctypes.Structure):
    _fields_ = [("src", _vp), ("ld_src", ctypes.c_long), ("src_col", ctypes.c_int), ("ncols", ctypes.c_int),
                ("dst", _vp), ("ld_dst", ctypes.c_long), ("dst_row0", ctypes.c_int), ("scale", ctypes.c_float)]


class _StageJobs(ctypes.Structure):
    _fields_ = [("j", _StageJob * 6)]


class _DdpgHeadArgs(ctypes.Structure):      # mirrors DdpgHeadArgs in csrc/mlp_kernels.hip
    _fields_ = [("h2", _vp), ("ldh", ctypes.c_long),
                ("w3_on", _vp), ("b3_on", _vp), ("w3_tg", _vp), ("b3_tg", _vp),
                ("tails", _vp), ("tail_ld", ctypes.c_int), ("off_reward", ctypes.c_int),
                ("off_gamma", ctypes.c_int), ("off_term", ctypes.c_int),
                ("idx", _vp), ("isw", _vp),
                ("dh2", _vp), ("dq", _vp), ("dq_ld", ctypes.c_int),
                ("q_out", _vp), ("td_out", _vp), ("loss_sum", _vp),
                ("leaf_sum", _vp), ("leaf_min", _vp), ("prio_eps", ctypes.c_float), ("prio_alpha", ctypes.c_float),
                ("B", ctypes.c_int), ("H", ctypes.c_int), ("leak", ctypes.c_int), ("actor_warmup", ctypes.c_int),
                ("rng_counter", _vp), ("opt_step", _vp)]


class _Layer(ctypes.Structure):             # mirrors Layer in csrc/mlp_fused.hip
    _fields_ = [("w", _vp), ("b", _vp), ("N", ctypes.c_int), ("K", ctypes.c_int), ("Kp", ctypes.c_int)]


class _DdpgFusedArgs(ctypes.Structure):     # mirrors DdpgFusedArgs in csrc/mlp_fused.hip
    _fields_ = [("s0", _vp), ("s1", _vp), ("ld_s", ctypes.c_long), ("tails", _vp), ("tail_ld", ctypes.c_int),
                ("off_action", ctypes.c_int), ("off_reward", ctypes.c_int), ("off_gamma", ctypes.c_int),
                ("off_term", ctypes.c_int), ("idx", _vp), ("isw", _vp),
                ("a0", _Layer * 2), ("a2", _Layer * 2), ("a4", _Layer * 2), ("c0", _Layer * 2), ("c1", _Layer * 2),
                ("w3", _vp * 2), ("b3", _vp * 2), ("c1T", _Layer), ("a2T", _Layer),
                ("x0", _vp), ("ld_x0", ctypes.c_long), ("ha1", _vp), ("ld_ha1", ctypes.c_long),
                ("ha2", _vp), ("ld_ha2", ctypes.c_long), ("cat", _vp), ("ld_cat", ctypes.c_long),
                ("h2", _vp), ("ld_h2", ctypes.c_long), ("dh2", _vp), ("dq", _vp), ("dq_ld", ctypes.c_int),
                ("dcat", _vp), ("dha2", _vp), ("dha1", _vp),
                ("q_out", _vp), ("td_out", _vp), ("loss_sum", _vp), ("leaf_sum", _vp), ("leaf_min", _vp),
                ("prio_eps", ctypes.c_float), ("prio_alpha", ctypes.c_float),
                ("B", ctypes.c_int), ("D", ctypes.c_int), ("A", ctypes.c_int), ("C1", ctypes.c_int),
                ("leak", ctypes.c_int), ("actor_warmup", ctypes.c_int), ("rng_counter", _vp), ("opt_step", _vp),
                ("actor_l2", ctypes.c_float), ("prof", _vp)]


_ready = False


def _L():
    global _ready
    L = lib()
    if not _ready:
        from . import dqn_engine
        dqn_engine._L()      # pda_dqn_head / pda_uniform_* prototypes
        L.pda_mlp_stage.argtypes = [ctypes.POINTER(_StageJobs), ctypes.c_int, _vp, ctypes.c_int, ctypes.c_int, _vp]
        L.pda_mlp_stage.restype = None
        L.pda_ddpg_head.argtypes = [ctypes.POINTER(_DdpgHeadArgs), _vp]
        L.pda_ddpg_head.restype = None
        L.pda_bf16_rows_to_f32.argtypes = [_vp, ctypes.c_long, _vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, _vp]
        L.pda_bf16_rows_to_f32.restype = None
        L.pda_ddpg_fused.argtypes = [ctypes.POINTER(_DdpgFusedArgs), _vp]
        L.pda_ddpg_fused.restype = None
        L.pda_ddpg_fused_supported.argtypes = [ctypes.c_int] * 6
        L.pda_ddpg_fused_supported.restype = ctypes.c_int
        L.pda_linear_wgrad_n.argtypes = [ctypes.c_int] + [_vp] * 10 + [_vp]
        L.pda_linear_wgrad_n.restype = None
        _ready = True
    return L


def stage_rows(jobs, idx, B):
    """jobs: [(src fp32 2-D tensor, src_col, ncols, dst bf16 2-D tensor/view, dst_row0, scale)]."""
    assert 0 < len(jobs) <= 6
    blk = _StageJobs()
    for i, (src, col, ncols, dst, row0, scale) in enumerate(jobs):
        assert src.dtype == torch.float32 and dst.dtype == BF16
        blk.j[i] = _StageJob(src.data_ptr(), int(src.stride(0)), int(col), int(ncols), dst.data_ptr(),
                             int(dst.stride(0)), int(row0), float(scale))
    _L().pda_mlp_stage(ctypes.byref(blk), len(jobs), _vp(_p(idx)), int(B), max(j[2] for j in jobs), _stream())


class MlpWeights(object):
    """Flat fp32 parameters of one network + zero-padded bf16 shadows [N][pad8(K)] of every
    matrix that feeds an MFMA GEMM.  Gradients/moments use the same [N][K] order."""

    def __init__(self, model, layers, with_grad, fp32_only=(), transposed=(), opt_chunk=None):
        """opt_chunk=256: the optimizer walks these segments one element per thread (SegmentTable "medium" chunks).
        transposed: layers that also keep a bf16 copy [K][pad8(N)] behind the row-major shadows (the fused
        backward's B operand, csrc/mlp_fused.hip); the optimizer launch maintains it (AdamExtras.tinfo)."""
        self.model = model
        self.flat = model.flatten_(with_grad=with_grad)
        dev = self.flat.device
        segs = {name: (off, shape) for name, off, shape in model.param_segments()}
        self.layers = list(layers)
        shapes = {}
        total = 0
        for name in self.layers:
            _, (n, k) = segs[name + ".weight"]
            shapes[name] = (n, k, _pad8(k))
            if name not in fp32_only:
                total += n * _pad8(k)
        self.transposed = tuple(transposed)
        t_total = sum(shapes[name][1] * _pad8(shapes[name][0]) for name in self.transposed)
        self.shadow = torch.zeros(max(total + t_total, 8), dtype=BF16, device=dev)
        self.wT = {}
        toff = total
        self.w, self.b, self.w32, self.off, self.shape = {}, {}, {}, {}, shapes
        entries, soff = [], 0
        covered = 0
        for name in self.layers:
            woff, _ = segs[name + ".weight"]
            boff, _ = segs[name + ".bias"]
            n, k, kp = shapes[name]
            ent = {"start": woff, "count": n * k, "cols": k, "medium": opt_chunk == 256}
            if name not in fp32_only:
                ent.update(C=1, P=k, shadow_off=soff, shadow_ld=kp, shadow=True)
                self.w[name] = self.shadow[soff:soff + n * kp].view(n, kp)
                soff += n * kp
            self.w32[name] = self.flat[woff:woff + n * k].view(n, k)
            self.b[name] = self.flat[boff:boff + n]
            self.off[name] = (woff, n * k, boff, n)
            entries.append(ent)
            entries.append({"start": boff, "count": n, "medium": opt_chunk == 256})
            covered += n * k + n
        assert covered == self.flat.numel(), "layer list does not cover the model's parameters"
        self.table = SegmentTable(entries, dev)
        tinfo = torch.zeros(len(entries), 2, dtype=torch.int32)
        for name in self.transposed:
            n, k, _ = shapes[name]
            npad = _pad8(n)
            self.wT[name] = self.shadow[toff:toff + k * npad].view(k, npad)
            tinfo[2 * self.layers.index(name)] = torch.tensor([toff, npad], dtype=torch.int32)
            toff += k * npad
        self.tinfo = tinfo.to(dev) if self.transposed else None
        self.refresh()

    def refresh(self):
        polyak_(self.flat, self.flat, 1.0, self.table, self.shadow)
        for name in self.transposed:
            n = self.shape[name][0]
            self.wT[name][:, :n].copy_(self.w32[name].t())

    def grad_views(self, flat_grads):
        gw, gb = {}, {}
        for name in self.layers:
            woff, wn, boff, bn = self.off[name]
            gw[name] = flat_grads[woff:woff + wn]
            gb[name] = flat_grads[boff:boff + bn]
        return gw, gb


class _EngineBase(object):
    """Sampling state, replay binding and bookkeeping shared by the two MLP engines."""

    def _init_common(self, B, dev, per, seed, tail_offsets, n_losses, per_sample_loss=False):
        self.B, self.device, self.per = int(B), dev, per
        self.idx = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.isw = torch.ones(self.B, dtype=torch.float32, device=dev)
        self.uniforms = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.td = torch.zeros(self.B, dtype=torch.float32, device=dev)
        # pda_dqn_head accumulates its loss per sample (no atomics): one slot per batch row
        self.per_sample_loss = per_sample_loss
        self.loss_acc = torch.zeros(self.B if per_sample_loss else n_losses, dtype=torch.float32, device=dev)
        self.loss_steps = 0
        self.size_dev = torch.zeros(4, dtype=torch.int32, device=dev)    # {rows, first row, ring length, -}: csrc/common.h window_pick
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.seed = int(seed) & 0xFFFFFFFFFFFF
        self.tail_offsets = tuple(int(o) for o in tail_offsets)
        self.replay = None
        self.step_count = 0

    def bind_replay(self, state0s, state1s, tails, tree=None):
        assert state0s.dtype == torch.float32 and state0s.is_cuda and tails.dtype == torch.float32
        m = state0s.shape[0]
        self.replay = (state0s.view(m, -1), state1s.view(m, -1), tails, tree)
        return self

    def make_staging(self, tail_floats):
        dev = self.device
        self.stage_s0 = torch.zeros(self.B, self.D, dtype=torch.float32, device=dev)
        self.stage_s1 = torch.zeros_like(self.stage_s0)
        self.stage_tails = torch.zeros(self.B, tail_floats, dtype=torch.float32, device=dev)
        return self

    def _sources(self, staged):
        if staged:
            return self.stage_s0, self.stage_s1, self.stage_tails, None, None
        s0, s1, tails, tree = self.replay
        return s0, s1, tails, self.idx, (tree if self.per is not None else None)

    def sample(self):
        _, _, _, tree = self.replay
        L = _L()
        if tree is not None and self.per is not None:
            if getattr(tree, "native", False):      # stratum uniforms drawn inside the descent kernel: one launch
                tree.sample_rng(self.B, float(self.per["beta"]), self.seed, self.counter, self.idx, self.isw)
            else:
                L.pda_uniform_f32(_p(self.uniforms), self.B, self.seed, _p(self.counter), _stream())
                tree.sample(self.uniforms, float(self.per["beta"]), self.idx, self.isw)
        else:
            L.pda_uniform_idx(_p(self.idx), _p(self.isw), self.B, _p(self.size_dev), self.seed, _p(self.counter),
                              _stream())

    def set_size(self, n):
        self.size_dev[:1].fill_(int(n))

    def _finish_update(self, staged):
        if not staged and self.per is not None and self.replay[3] is not None:
            self.replay[3].repair(self.idx, self.B)

    def mean_losses(self, lagged=False):
        if lagged and self.loss_acc.is_cuda:        # no host synchronisation: one read-out period behind
            if getattr(self, "_loss_readout", None) is None:
                from .optim import LaggedReadout
                self._loss_readout = LaggedReadout(self.loss_acc)
            vals, n = self._loss_readout.read(self.loss_steps)
            self.loss_steps = 0
            return [float(vals.sum()) / n] if self.per_sample_loss else (vals / n).tolist()
        n = max(self.loss_steps, 1)
        vals = [float(self.loss_acc.sum()) / n] if self.per_sample_loss else (self.loss_acc / n).tolist()
        self.loss_acc.zero_()
        self.loss_steps = 0
        return vals


# =====================================================================================
# DDPG
# =====================================================================================
DDPG_LAYERS = ("actor.0", "actor.2", "actor.4", "critic.0.0", "critic.1.0", "critic.1.2")


class DdpgMlpEngine(_EngineBase):
    def __init__(self, model, target_model, batch_size, lr=1e-4, weight_decay=0.0, clip=40.0, tau=1e-3,
                 per=None, seed=0, tail_offsets=(0, 1, 2, 3), leak=True, actor_warmup=0, actor_l2=0.0):
        self.actor_warmup = int(actor_warmup)
        self.actor_l2 = float(actor_l2)
        self.online = MlpWeights(model, DDPG_LAYERS, with_grad=True, fp32_only=("critic.1.2",),
                                 transposed=("critic.1.0", "actor.2"))
        self.target = MlpWeights(target_model, DDPG_LAYERS, with_grad=False, fp32_only=("critic.1.2",))
        on = self.online
        dev = on.flat.device
        self.H1, self.D, self.Dp = on.shape["actor.0"]           # actor hidden 1, obs dim
        self.H2 = on.shape["actor.2"][0]
        self.A = on.shape["actor.4"][0]
        self.C1 = on.shape["critic.0.0"][0]                      # critic hidden 1 (400)
        self.C2, self.CK, self.CKp = on.shape["critic.1.0"]      # 300, 400 + A, padded
        assert self.CK == self.C1 + self.A and on.shape["critic.1.2"][0] == 1
        assert self.C1 % 8 == 0, "concat folding needs the action columns 16-byte aligned"
        self._init_common(batch_size, dev, per, seed, tail_offsets, n_losses=2)
        self.tau, self.leak = float(tau), bool(leak)
        self.target_update = self.tau
        B = self.B
        z = lambda *s, dtype=BF16: torch.zeros(*s, dtype=dtype, device=dev)
        H1p, H2p, C2p = _pad8(self.H1), _pad8(self.H2), _pad8(self.C2)
        self.x0 = z(2 * B, self.Dp)          # s0 twice: rows [0,B) actor-loss pass, [B,2B) critic-loss pass
        self.x1 = z(B, self.Dp)
        self.ha1 = [z(B, H1p), z(B, H1p)]    # online(s0), target(s_n)
        self.ha2 = [z(B, H2p), z(B, H2p)]
        self.cat_on = z(2 * B, self.CKp)     # [tanh(critic.0.0) | action]: rows [0,B) use pi(s0), [B,2B) the replay action
        self.cat_tg = z(B, self.CKp)
        self.h2 = z(3 * B, C2p)
        self.dh2 = z(2 * B, C2p)
        self.dq = z(2 * B, 8)
        self.dcat = z(2 * B, self.CKp)
        self.dha2 = z(B, H2p)
        self.dha1 = z(B, H1p)
        self.q = torch.zeros(3 * B, dtype=torch.float32, device=dev)
        self.grads = model.flat_grads
        self.grads.zero_()
        self.gw, self.gb = on.grad_views(self.grads)
        self.adam = FlatAdam(on.flat, self.grads, on.table, lr=lr, weight_decay=weight_decay, clip=clip,
                             shadow=on.shadow)
        polyak_(self.target.flat, on.flat, 1.0, self.target.table, self.target.shadow)
        # one launch for forward + head + data gradients, one for all weight gradients, one for Adam + Polyak +
        # the next index draw (csrc/mlp_fused.hip); shapes it does not cover run the layer-by-layer launches
        self.fused = bool(_L().pda_ddpg_fused_supported(self.D, self.H1, self.H2, self.A, self.C1, self.C2))
        self._wgrad_jobs = None
        if self.actor_l2 > 0 and not self.fused:
            raise ValueError("actor_preact_l2 is implemented by the fused DDPG kernel only (unsupported layer shape)")

    # ---- fused step (3 launches with uniform sampling) --------------------------------------------
    def _layer(self, w, name):
        n, k, kp = w.shape[name]
        return _Layer(_p(w.w[name]), _p(w.b[name]), n, k, kp)

    def fused_forward_backward(self, staged=False):
        B, on, tg = self.B, self.online, self.target
        s0, s1, tails, idx, tree = self._sources(staged)
        oa, orw, og, ot = self.tail_offsets
        a = _DdpgFusedArgs()
        a.s0, a.s1, a.ld_s = _p(s0), _p(s1), int(s0.stride(0))
        a.tails, a.tail_ld = _p(tails), int(tails.stride(0))
        a.off_action, a.off_reward, a.off_gamma, a.off_term = oa, orw, og, ot
        a.idx, a.isw = _p(idx), (_p(self.isw) if self.per is not None else 0)
        for i, w in enumerate((on, tg)):
            a.a0[i], a.a2[i], a.a4[i] = self._layer(w, "actor.0"), self._layer(w, "actor.2"), self._layer(w, "actor.4")
            a.c0[i], a.c1[i] = self._layer(w, "critic.0.0"), self._layer(w, "critic.1.0")
            a.w3[i], a.b3[i] = _p(w.w32["critic.1.2"]), _p(w.b["critic.1.2"])
        a.c1T = _Layer(_p(on.wT["critic.1.0"]), 0, self.CK, self.C2, _pad8(self.C2))
        a.a2T = _Layer(_p(on.wT["actor.2"]), 0, self.H1, self.H2, _pad8(self.H2))
        a.x0, a.ld_x0 = _p(self.x0), self.Dp
        a.ha1, a.ld_ha1 = _p(self.ha1[0]), self.ha1[0].shape[1]
        a.ha2, a.ld_ha2 = _p(self.ha2[0]), self.ha2[0].shape[1]
        a.cat, a.ld_cat = _p(self.cat_on), self.CKp
        a.h2, a.ld_h2 = _p(self.h2), self.h2.shape[1]
        a.dh2, a.dq, a.dq_ld = _p(self.dh2), _p(self.dq), 8
        a.dcat, a.dha2, a.dha1 = _p(self.dcat), _p(self.dha2), _p(self.dha1)
        a.q_out, a.td_out, a.loss_sum = _p(self.q), _p(self.td), _p(self.loss_acc)
        a.leaf_sum = _p(tree.sum) if tree is not None else 0
        a.leaf_min = _p(tree.min) if tree is not None else 0
        a.prio_eps = float(self.per["eps"]) if self.per else 0.0
        a.prio_alpha = float(self.per["alpha"]) if self.per else 1.0
        a.B, a.D, a.A, a.C1 = B, self.D, self.A, self.C1
        a.leak, a.actor_warmup = (1 if self.leak else 0), self.actor_warmup
        a.rng_counter, a.opt_step = _p(self.counter), _p(self.adam.step_dev)
        a.actor_l2 = self.actor_l2
        a.prof = _p(getattr(self, "prof", None))
        _L().pda_ddpg_fused(ctypes.byref(a), _stream())

    def wgrad_all(self):
        """The six weight (+ bias) gradients as problem groups of ONE GEMM launch."""
        if self._wgrad_jobs is None:
            B = self.B
            H1p, H2p, C2p, CKp = self.ha1[0].shape[1], self.ha2[0].shape[1], self.h2.shape[1], self.CKp
            r0, rows = (0, 2 * B) if self.leak else (B, B)      # critic: both passes (reference) or the critic loss only
            dz3 = self.dcat[:B, self.C1:]
            jobs = [  # dy, x, layer, rows, N, K, lddy, ldx
                (self.dq[r0:], self.h2[r0:], "critic.1.2", rows, 1, self.C2, 8, C2p),
                (self.dh2[r0:], self.cat_on[r0:], "critic.1.0", rows, self.C2, self.CK, C2p, CKp),
                (self.dcat[r0:], self.x0[r0:], "critic.0.0", rows, self.C1, self.D, CKp, self.Dp),
                (dz3, self.ha2[0], "actor.4", B, self.A, self.H2, CKp, H2p),
                (self.dha2, self.ha1[0], "actor.2", B, self.H2, self.H1, H2p, H1p),
                (self.dha1, self.x0, "actor.0", B, self.H1, self.D, H1p, self.Dp)]
            n = len(jobs)
            vp, ci, cl = _vp * n, ctypes.c_int * n, ctypes.c_long * n
            self._wgrad_jobs = (n, vp(*[_p(j[0]) for j in jobs]), vp(*[_p(j[1]) for j in jobs]),
                                vp(*[_p(self.gw[j[2]]) for j in jobs]), vp(*[_p(self.gb[j[2]]) for j in jobs]),
                                ci(*[j[3] for j in jobs]), ci(*[j[4] for j in jobs]), ci(*[j[5] for j in jobs]),
                                cl(*[j[6] for j in jobs]), cl(*[j[7] for j in jobs]), cl(*[j[5] for j in jobs]))
        _L().pda_linear_wgrad_n(*self._wgrad_jobs, _stream())

    def fused_extras(self, staged):
        """Transposed shadows, the Polyak update of the target network and (uniform sampling from device replay)
        the next step's index draw, all inside the optimizer launch."""
        from .optim import AdamExtras
        x = AdamExtras()
        x.tinfo = _p(self.online.tinfo)
        x.tgt, x.tgt_shadow, x.tau = _p(self.target.flat), _p(self.target.shadow), float(self.tau)
        if self.presamples(staged):
            x.next_idx, x.next_w, x.next_B = _p(self.idx), _p(self.isw), self.B
            x.size, x.seed, x.counter = _p(self.size_dev), self.seed, _p(self.counter)
        return x

    def presamples(self, staged):
        return (not staged) and self.replay is not None and not (self.replay[3] is not None and self.per is not None)

    # ---- step pieces ---------------------------------------------------------------------------
    def forward_head(self, staged=False):
        B, on, tg = self.B, self.online, self.target
        s0, s1, tails, idx, tree = self._sources(staged)
        act_off = self.tail_offsets[0]
        act_view = self.cat_on[:, self.C1:]
        stage_rows([(s0, 0, self.D, self.x0, 0, 1.0), (s0, 0, self.D, self.x0, B, 1.0),
                    (s1, 0, self.D, self.x1, 0, 1.0),
                    (tails, act_off, self.A, act_view, B, 1.0)], idx, B)
        H1p, H2p, C2p, CKp = self.ha1[0].shape[1], self.ha2[0].shape[1], self.h2.shape[1], self.CKp
        x0a = self.x0[:B]
        # actor: online(s0) and target(s_n) as two groups per launch
        nn.linear_fwd([x0a, self.x1], [on.w["actor.0"], tg.w["actor.0"]], [on.b["actor.0"], tg.b["actor.0"]],
                      self.ha1, None, B, self.H1, self.D, self.Dp, self.Dp, H1p, 0, act=nn.ACT_TANH)
        nn.linear_fwd(self.ha1, [on.w["actor.2"], tg.w["actor.2"]], [on.b["actor.2"], tg.b["actor.2"]],
                      self.ha2, None, B, self.H2, self.H1, H1p, H1p, H2p, 0, act=nn.ACT_TANH)
        nn.linear_fwd(self.ha2, [on.w["actor.4"], tg.w["actor.4"]], [on.b["actor.4"], tg.b["actor.4"]],
                      [self.cat_on[:B, self.C1:], self.cat_tg[:, self.C1:]], None, B, self.A, self.H2, H2p, H2p,
                      CKp, 0, act=nn.ACT_TANH)
        # critic layer 1 writes the left part of the concat buffers
        nn.linear_fwd([x0a, self.x0[B:], self.x1],
                      [on.w["critic.0.0"], on.w["critic.0.0"], tg.w["critic.0.0"]],
                      [on.b["critic.0.0"], on.b["critic.0.0"], tg.b["critic.0.0"]],
                      [self.cat_on[:B], self.cat_on[B:], self.cat_tg], None, B, self.C1, self.D, self.Dp, self.Dp,
                      CKp, 0, act=nn.ACT_TANH)
        nn.linear_fwd([self.cat_on[:B], self.cat_on[B:], self.cat_tg],
                      [on.w["critic.1.0"], on.w["critic.1.0"], tg.w["critic.1.0"]],
                      [on.b["critic.1.0"], on.b["critic.1.0"], tg.b["critic.1.0"]],
                      [self.h2[:B], self.h2[B:2 * B], self.h2[2 * B:]], None, B, self.C2, self.CK, CKp, CKp,
                      C2p, 0, act=nn.ACT_TANH)
        _, orw, og, ot = self.tail_offsets
        args = _DdpgHeadArgs(
            h2=_p(self.h2), ldh=C2p,
            w3_on=_p(on.w32["critic.1.2"]), b3_on=_p(on.b["critic.1.2"]),
            w3_tg=_p(tg.w32["critic.1.2"]), b3_tg=_p(tg.b["critic.1.2"]),
            tails=_p(tails), tail_ld=int(tails.stride(0)), off_reward=orw, off_gamma=og, off_term=ot,
            idx=_p(idx), isw=_p(self.isw) if self.per is not None else 0,
            dh2=_p(self.dh2), dq=_p(self.dq), dq_ld=8, q_out=_p(self.q), td_out=_p(self.td),
            loss_sum=_p(self.loss_acc),
            leaf_sum=_p(tree.sum) if tree is not None else 0, leaf_min=_p(tree.min) if tree is not None else 0,
            prio_eps=float(self.per["eps"]) if self.per else 0.0,
            prio_alpha=float(self.per["alpha"]) if self.per else 1.0,
            B=B, H=self.C2, leak=1 if self.leak else 0, actor_warmup=self.actor_warmup,
            rng_counter=_p(self.counter), opt_step=_p(self.adam.step_dev))
        _L().pda_ddpg_head(ctypes.byref(args), _stream())

    def backward(self, staged=False):
        B, on = self.B, self.online
        H1p, H2p, C2p, CKp = self.ha1[0].shape[1], self.ha2[0].shape[1], self.h2.shape[1], self.CKp
        gw, gb = self.gw, self.gb
        # critic weight gradients read both passes (reference behaviour) or only the critic-loss rows
        r0, rows = (0, 2 * B) if self.leak else (B, B)
        nn.linear_wgrad(self.dq[r0:], self.h2[r0:], gw["critic.1.2"], gb["critic.1.2"], rows, 1, self.C2,
                        8, C2p, self.C2)
        nn.linear_wgrad(self.dh2[r0:], self.cat_on[r0:], gw["critic.1.0"], gb["critic.1.0"], rows, self.C2, self.CK,
                        C2p, CKp, self.CK)
        nn.linear_dgrad(self.dh2, on.w["critic.1.0"], self.cat_on, self.dcat, 2 * B, self.C2, self.CK, C2p, CKp,
                        CKp, CKp, act=nn.ACT_TANH)
        nn.linear_wgrad(self.dcat[r0:], self.x0[r0:], gw["critic.0.0"], gb["critic.0.0"], rows, self.C1, self.D,
                        CKp, self.Dp, self.D)
        # actor: dL/d(pre-tanh action) already sits in the action columns of dcat rows [0,B)
        dz3 = self.dcat[:B, self.C1:]
        nn.linear_wgrad(dz3, self.ha2[0], gw["actor.4"], gb["actor.4"], B, self.A, self.H2, CKp, H2p, self.H2)
        nn.linear_dgrad(dz3, on.w["actor.4"], self.ha2[0], self.dha2, B, self.A, self.H2, CKp, H2p, H2p, H2p,
                        act=nn.ACT_TANH)
        nn.linear_wgrad(self.dha2, self.ha1[0], gw["actor.2"], gb["actor.2"], B, self.H2, self.H1, H2p, H1p, self.H1)
        nn.linear_dgrad(self.dha2, on.w["actor.2"], self.ha1[0], self.dha1, B, self.H2, self.H1, H2p, H1p, H1p, H1p,
                        act=nn.ACT_TANH)
        nn.linear_wgrad(self.dha1, self.x0, gw["actor.0"], gb["actor.0"], B, self.H1, self.D, H1p, self.Dp, self.D)

    def update(self, staged=False, tick=False, extras=None):
        if extras is not None:          # Polyak (and the transposed shadows) ride in the optimizer launch
            self.adam.step(tick=tick, extras=extras)
        else:
            self.adam.step(tick=tick, extras=self._tinfo_extras())
            polyak_(self.target.flat, self.online.flat, self.tau, self.target.table, self.target.shadow)
        self._finish_update(staged)

    def _tinfo_extras(self):
        if getattr(self, "_tx", None) is None:
            from .optim import AdamExtras
            self._tx = AdamExtras()
            self._tx.tinfo = _p(self.online.tinfo)
        return self._tx

    def hard_target_update(self):
        polyak_(self.target.flat, self.online.flat, 1.0, self.target.table, self.target.shadow)


# =====================================================================================
# MLP DQN
# =====================================================================================
DQN_MLP_LAYERS = ("critic.0", "critic.2", "critic.4", "critic.6")


class DqnMlpEngine(_EngineBase):
    def __init__(self, model, target_model, batch_size, lr=1e-4, weight_decay=0.0, clip=np.inf, double=False,
                 target_update=250, per=None, seed=0, tail_offsets=(0, 1, 2, 3)):
        from .dqn_engine import HID_SUPPORTED
        self.online = MlpWeights(model, DQN_MLP_LAYERS, with_grad=True, fp32_only=("critic.6",))
        self.target = MlpWeights(target_model, DQN_MLP_LAYERS, with_grad=False, fp32_only=("critic.6",))
        on = self.online
        dev = on.flat.device
        self.HID, self.D, self.Dp = on.shape["critic.0"]
        self.A = on.shape["critic.6"][0]
        if self.HID not in HID_SUPPORTED or self.A > 64:
            raise ValueError("unsupported dqn-mlp head shape")
        self._init_common(batch_size, dev, per, seed, tail_offsets, n_losses=1, per_sample_loss=True)
        self.norm_val = float(model.norm_val)
        self.double = bool(double)
        self.target_update = target_update
        self.groups = 3 if self.double else 2
        B, H = self.B, self.HID
        z = lambda *s, dtype=BF16: torch.zeros(*s, dtype=dtype, device=dev)
        self.x0, self.x1 = z(B, self.Dp), z(B, self.Dp)
        self.h1 = [z(B, H) for _ in range(self.groups)]
        self.h2 = [z(B, H) for _ in range(self.groups)]
        self.h3pre = [z(B, H, dtype=torch.float32) for _ in range(self.groups)]
        self.h3 = z(B, H)
        self.dh3 = z(B, H)
        self.dq_ld = max(16, _pad8(self.A))
        self.dq = z(B, self.dq_ld)
        self.dh2, self.dh1 = z(B, H), z(B, H)
        self.q = torch.zeros(B, self.A, dtype=torch.float32, device=dev)
        self.grads = model.flat_grads
        self.grads.zero_()
        self.gw, self.gb = on.grad_views(self.grads)
        self.adam = FlatAdam(on.flat, self.grads, on.table, lr=lr, weight_decay=weight_decay, clip=clip,
                             shadow=on.shadow)
        polyak_(self.target.flat, on.flat, 1.0, self.target.table, self.target.shadow)

    def forward_head(self, staged=False):
        from .dqn_engine import _HeadArgs
        B, H, on, tg = self.B, self.HID, self.online, self.target
        s0, s1, tails, idx, tree = self._sources(staged)
        sc = 1.0 / self.norm_val
        stage_rows([(s0, 0, self.D, self.x0, 0, sc), (s1, 0, self.D, self.x1, 0, sc)], idx, B)
        nets = [on, tg] + ([on] if self.double else [])
        xs = [self.x0, self.x1] + ([self.x1] if self.double else [])
        n = len(nets)
        nn.linear_fwd(xs, [w.w["critic.0"] for w in nets], [w.b["critic.0"] for w in nets], self.h1[:n], None,
                      B, H, self.D, self.Dp, self.Dp, H, 0, act=nn.ACT_RELU)
        nn.linear_fwd(self.h1[:n], [w.w["critic.2"] for w in nets], [w.b["critic.2"] for w in nets], self.h2[:n],
                      None, B, H, H, H, H, H, 0, act=nn.ACT_RELU)
        nn.linear_fwd(self.h2[:n], [w.w["critic.4"] for w in nets], None, None, self.h3pre[:n],
                      B, H, H, H, H, 0, H, act=nn.ACT_NONE)
        oa, orw, og, ot = self.tail_offsets
        args = _HeadArgs(
            h1pre_on0=_p(self.h3pre[0]), h1pre_tg1=_p(self.h3pre[1]),
            h1pre_on1=_p(self.h3pre[2]) if self.double else 0,
            b4_on=_p(on.b["critic.4"]), b4_tg=_p(tg.b["critic.4"]),
            w5_on=_p(on.w32["critic.6"]), b5_on=_p(on.b["critic.6"]),
            w5_tg=_p(tg.w32["critic.6"]), b5_tg=_p(tg.b["critic.6"]),
            tails=_p(tails), tail_ld=int(tails.stride(0)), off_action=oa, off_reward=orw, off_gamma=og, off_term=ot,
            idx=_p(idx), isw=_p(self.isw) if self.per is not None else 0,
            h1_on0=_p(self.h3), dh1=_p(self.dh3), dq=_p(self.dq), dq_ld=self.dq_ld,
            q_out=_p(self.q), td_out=_p(self.td), loss_sum=_p(self.loss_acc),
            leaf_sum=_p(tree.sum) if tree is not None else 0, leaf_min=_p(tree.min) if tree is not None else 0,
            prio_eps=float(self.per["eps"]) if self.per else 0.0,
            prio_alpha=float(self.per["alpha"]) if self.per else 1.0,
            B=B, HID=H, A=self.A, rng_counter=_p(self.counter), opt_step=_p(self.adam.step_dev))
        _L().pda_dqn_head(ctypes.byref(args), _stream())

    def backward(self, staged=False):
        B, H, on = self.B, self.HID, self.online
        gw, gb = self.gw, self.gb
        nn.linear_wgrad(self.dq, self.h3, gw["critic.6"], gb["critic.6"], B, self.A, H, self.dq_ld, H, H)
        nn.linear_wgrad(self.dh3, self.h2[0], gw["critic.4"], gb["critic.4"], B, H, H, H, H, H)
        nn.linear_dgrad(self.dh3, on.w["critic.4"], self.h2[0], self.dh2, B, H, H, H, H, H, H, act=nn.ACT_RELU)
        nn.linear_wgrad(self.dh2, self.h1[0], gw["critic.2"], gb["critic.2"], B, H, H, H, H, H)
        nn.linear_dgrad(self.dh2, on.w["critic.2"], self.h1[0], self.dh1, B, H, H, H, H, H, H, act=nn.ACT_RELU)
        nn.linear_wgrad(self.dh1, self.x0, gw["critic.0"], gb["critic.0"], B, H, self.D, H, self.Dp, self.D,
                        scale=1.0)

    def update(self, staged=False, tick=False):
        self.adam.step(tick=tick)
        if self.target_update < 1.0:
            polyak_(self.target.flat, self.online.flat, float(self.target_update), self.target.table,
                    self.target.shadow)
        self._finish_update(staged)

    def hard_target_update(self):
        polyak_(self.target.flat, self.online.flat, 1.0, self.target.table, self.target.shadow)


# =====================================================================================
# step program + learner adapter
# =====================================================================================
class MlpStepProgram(object):
    """[sample, stage, forward, head, backward] | all-reduce(flat grads) | [adam, target, tree repair];
    one hipGraph per bracket (a single graph when there is one rank)."""

    def __init__(self, engine, comm=None, use_graph=True, staged=False, force_dp=False, graph_steps=1):
        self.e, self.comm, self.staged = engine, comm, staged
        self.world = comm.world_size if comm is not None else 1
        self.dp = comm is not None and (self.world > 1 or force_dp)
        self.use_graph = use_graph
        self.graphs = None
        self.calls = 0
        # updates per graph replay (single rank, device replay): see ops/dqn_engine.py:DqnStepProgram
        k = int(graph_steps or 1)
        if self.dp or staged or not use_graph:
            k = 1
        tu = engine.target_update
        if k > 1 and tu >= 1.0:
            while k > 1 and int(tu) % k != 0:
                k -= 1
        self.graph_steps, self.ahead = max(1, k), 0
        if self.dp:     # the all-reduce is a SUM; Adam applies the 1/world_size
            engine.adam.set_grad_scale(1.0 / self.world)
        self.fused = bool(getattr(engine, "fused", False))
        self.extras = engine.fused_extras(staged) if self.fused else None
        self.presampled = False

    def _part_a(self):
        e = self.e
        if not self.staged and not (self.fused and self.presampled and e.presamples(self.staged)):
            e.sample()
        if self.fused:
            e.fused_forward_backward(self.staged)
            e.wgrad_all()
        else:
            e.forward_head(self.staged)
            e.backward(self.staged)

    def _part_c(self):
        if self.fused:
            self.e.update(self.staged, extras=self.extras)
            self.presampled = True
        else:
            self.e.update(self.staged)

    def _all(self):
        self._part_a()
        self._part_c()

    def _capture(self, fns):
        graphs = []
        for fn in fns:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                fn()
            graphs.append(gr)
        return graphs

    def step(self):
        e = self.e
        if self.ahead > 0:              # already performed by the last multi-step replay
            self.ahead -= 1
            self.calls += 1
            return
        done = 1
        if not self.use_graph or self.calls == 0:
            if not self.dp:
                self._all()
            else:
                self._part_a()
                self.comm.all_reduce_sum_(e.grads)
                self._part_c()
        else:
            if self.graphs is None:
                def k_steps():
                    for _ in range(self.graph_steps):
                        self._all()
                self.graphs = self._capture([self._part_a, self._part_c] if self.dp else [k_steps])
            if not self.dp:
                self.graphs[0].replay()
                done = self.graph_steps
                self.ahead = done - 1
            else:
                self.graphs[0].replay()
                self.comm.all_reduce_sum_(e.grads)
                self.graphs[1].replay()
        self.calls += 1
        e.loss_steps += done
        e.step_count += done - 1
        if e.target_update >= 1.0 and e.step_count % int(e.target_update) == 0:
            e.hard_target_update()
        e.step_count += 1


class MlpNativeCore(object):
    """Learner-loop adapter (agents/learners.py) for the two MLP engines."""

    def __init__(self, kind, args, model, target_model, memory, comm=None, use_graph=True):
        ap = args.agent_params
        self.kind, self.args, self.ap, self.memory, self.comm = kind, args, ap, memory, comm
        self.model, self.target_model = model, target_model
        per = None
        if memory.enable_per:
            per = {"eps": memory.priority_eps, "alpha": memory.priority_exponent, "beta": memory.priority_weight}
        a, k = int(memory.action_shape), int(memory.reward_shape)
        rank = comm.rank if comm is not None else 0
        seed = args.seed * 7919 + 104729 * rank + 17
        offs = (0, a, a + k, a + 2 * k)
        precision = str(getattr(args, "precision", "fp32"))
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16', not %r" % (precision,))
        self.precision = precision
        if kind == "ddpg":
            # precision="fp32" (the default, the reference learner's arithmetic): fp32 operands on the fp32-input MFMA
            # (ops/mlp_f32_engine.py); "bf16": the fused bf16 kernel
            if precision == "fp32":
                from .mlp_f32_engine import DdpgMlpEngineF32 as engine_cls
            else:
                engine_cls = DdpgMlpEngine
            self.engine = engine_cls(model, target_model, ap.batch_size, lr=ap.lr, weight_decay=ap.weight_decay,
                                     clip=ap.clip_grad, tau=ap.target_model_update, per=per, seed=seed,
                                     tail_offsets=offs, leak=getattr(ap, "compat_ddpg_grad_leak", True),
                                     actor_warmup=getattr(ap, "actor_warmup_steps", 0),
                                     actor_l2=float(getattr(ap, "actor_preact_l2", 0.0) or 0.0))
            assert self.engine.A == a
        else:
            if precision == "fp32":
                from .mlp_f32_engine import DqnMlpEngineF32 as dqn_cls
            else:
                dqn_cls = DqnMlpEngine
            self.engine = dqn_cls(model, target_model, ap.batch_size, lr=ap.lr, weight_decay=ap.weight_decay,
                                       clip=ap.clip_grad, double=ap.enable_double,
                                       target_update=ap.target_model_update, per=per, seed=seed, tail_offsets=offs)
        self.on_device = hasattr(memory, "tails") and getattr(memory, "attached", False) and memory.tails.is_cuda
        if self.on_device:
            self.engine.bind_replay(memory.state0s, memory.state1s, memory.tails, memory.tree)
        else:
            self.engine.make_staging(a + 3 * k + 1)
        self.program = MlpStepProgram(self.engine, comm, use_graph=use_graph, staged=not self.on_device,
                                      force_dp=bool(getattr(args, "force_dp", False)),
                                      graph_steps=int(getattr(args, "graph_steps", 1)))
        self._last_size = -1
        self.step_count = 0

    def step(self):
        e = self.engine
        if self.on_device:
            n = self.memory.size
            if n != self._last_size:
                e.set_size(n)
                self._last_size = n
            self.program.step()
        else:
            batch, inds, weights = self.memory.sample_with_priorities(self.ap.batch_size)
            s0, a, r, g, s1, t = batch
            e.stage_s0.copy_(s0.reshape(e.B, -1).float(), non_blocking=True)
            e.stage_s1.copy_(s1.reshape(e.B, -1).float(), non_blocking=True)
            tails = torch.cat([a.float().reshape(e.B, -1), r.float().reshape(e.B, -1), g.float().reshape(e.B, -1),
                               t.float().reshape(e.B, -1)], dim=1)
            e.stage_tails[:, :tails.shape[1]].copy_(tails, non_blocking=True)
            e.isw.copy_(weights.reshape(-1).float(), non_blocking=True)
            self.program.step()
            if self.memory.enable_per and inds is not None:
                self.memory.update_priorities(inds, e.td.abs().cpu())
        self.step_count += 1

    def flat_params(self):
        return self.engine.online.flat

    def loss_values(self, lagged=False):
        vals = self.engine.mean_losses(lagged=lagged)
        if self.kind == "ddpg":
            return {"actor_loss": vals[0], "critic_loss": vals[1]}
        return {"critic_loss": vals[0]}


def make_mlp_native_core(kind, args, model, target_model, memory, comm):
    """The native core for an MLP agent (fp32: ops/mlp_f32_engine.py, bf16: the engines above), or None when the
    layer shapes are outside what the native engines cover (the caller then uses the fp32 torch core)."""
    precision = str(getattr(args, "precision", "fp32"))
    if precision not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16', not %r" % (precision,))
    if kind == "ddpg" and args.model_type == "ddpg-mlp":
        return MlpNativeCore("ddpg", args, model, target_model, memory, comm)
    if kind == "dqn" and args.model_type == "dqn-mlp":
        try:
            return MlpNativeCore("dqn", args, model, target_model, memory, comm)
        except ValueError:
            return None
    return None


# =====================================================================================
# batched actor inference
# =====================================================================================
class DdpgActorInference(object):
    """pi(s) for a batch of observations (inference server): three grouped-GEMM launches."""

    def __init__(self, model, max_batch):
        self.model = model
        self.max_batch = int(max_batch)
        self._build()

    def _build(self):
        self.weights = w = MlpWeights(self.model, DDPG_LAYERS, with_grad=False, fp32_only=("critic.1.2",))
        dev = w.flat.device
        self.H1, self.D, self.Dp = w.shape["actor.0"]
        self.H2, self.A = w.shape["actor.2"][0], w.shape["actor.4"][0]
        n = self.max_batch
        z = lambda *s, dtype=BF16: torch.zeros(*s, dtype=dtype, device=dev)
        self.x, self.h1, self.h2 = z(n, self.Dp), z(n, _pad8(self.H1)), z(n, _pad8(self.H2))
        self.a = z(n, 8 if self.A <= 8 else _pad8(self.A))
        self.out = torch.zeros(n, self.A, dtype=torch.float32, device=dev)

    def refresh_weights(self):
        if self.weights.flat.data_ptr() != self.model.flat_params.data_ptr():
            self._build()
        else:
            self.weights.refresh()

    @torch.no_grad()
    def act(self, obs):
        """obs fp32 (n, D) on the device -> actions fp32 (n, A) on the device."""
        n = int(obs.shape[0])
        obs = obs.reshape(n, -1).float().contiguous()
        assert n <= self.max_batch and obs.shape[1] == self.D
        w = self.weights
        H1p, H2p = self.h1.shape[1], self.h2.shape[1]
        stage_rows([(obs, 0, self.D, self.x, 0, 1.0)], None, n)
        nn.linear_fwd([self.x], [w.w["actor.0"]], [w.b["actor.0"]], [self.h1], None, n, self.H1, self.D, self.Dp,
                      self.Dp, H1p, 0, act=nn.ACT_TANH)
        nn.linear_fwd([self.h1], [w.w["actor.2"]], [w.b["actor.2"]], [self.h2], None, n, self.H2, self.H1, H1p, H1p,
                      H2p, 0, act=nn.ACT_TANH)
        nn.linear_fwd([self.h2], [w.w["actor.4"]], [w.b["actor.4"]], [self.a], None, n, self.A, self.H2, H2p, H2p,
                      self.a.shape[1], 0, act=nn.ACT_TANH)
        _L().pda_bf16_rows_to_f32(_p(self.a), int(self.a.shape[1]), _p(self.out), self.A, n, self.A, _stream())
        return self.out[:n]
