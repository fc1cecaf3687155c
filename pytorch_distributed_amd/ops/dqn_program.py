"""The DQN learner step as a program: how an engine's step pieces are sequenced, captured into hipGraphs, run ahead of
the host, and -- data parallel -- cut around the gradient all-reduces (reference: the learner loop body,
/root/reference/core/single_processes/dqn_learner.py:50-101; num_learners, /root/reference/main.py:82-94)."""
import torch

from ._dqn_base import BF16, _Range


class DqnStepProgram(object):
    """Sequences an engine's step and captures it into hipGraphs.

    One rank: one graph per step.  Data parallel (one process per GPU, RCCL): the step is cut
    where the collectives go, and both all-reduces run on a side stream under other work:

      main   [A: sample, forward, head, fc wgrad] [B: conv backward ........] [Adam(fc tail)] [Adam(conv), target, tree]
      comm                                 all-reduce(fc tail, 95 % of bytes)   all-reduce(conv grads)

    The all-reduce is a SUM; the 1/world_size is Adam's grad_scale, so no scaling launch exists."""

    def __init__(self, engine, comm=None, use_graph=True, staged=False, force_dp=False, graph_steps=1):
        self.e, self.comm, self.staged = engine, comm, staged
        self.world = comm.world_size if comm is not None else 1
        self.dp = comm is not None and (self.world > 1 or force_dp)
        self.use_graph = use_graph
        self.graphs = None
        self.dp_graph_mode = None
        self.calls = 0
        # Updates per graph replay.  Between two replays the GPU idles ~8 us (graph completion + next launch: the
        # kernels inside a graph run back to back, tools/step_gaps.py), 9 % of a 0.09 ms step; a graph of K steps
        # pays it once per K.  The program then runs AHEAD of its caller: the first step() of a group performs K
        # updates, the next K - 1 calls return at once.  Only for the HBM-replay programs (single rank, or data
        # parallel with the collectives captured in the graph), and K is cut
        # down to a divisor of the hard target-update period so that those updates still fall between replays.
        self.knobs = knobs = engine.knobs
        _Range.enabled = bool(knobs.roctx)
        k = int(graph_steps or 1)
        if staged or not use_graph:
            k = 1
        tu = engine.target_update
        if k > 1 and tu >= 1.0:
            while k > 1 and int(tu) % k != 0:
                k -= 1
        self.graph_steps, self.ahead = max(1, k), 0
        # How far the host may run ahead of the GPU, in graph replays.  Unbounded, the launch queue fills up and --
        # with the inference loop's thread and the ingestion copies driving the same device -- every launch then
        # waits inside the runtime: the learner fell from 8.1 k to 4.9 k updates/s inside the job when its one
        # periodic host synchronisation (the loss read-out) was removed.  Three replays in flight keep the GPU fed
        # (a replay is ~0.5 ms of work, a launch ~20 us) and the queue short.
        self.max_inflight = max(1, int(knobs.max_inflight_replays))
        self._inflight = []
        # single rank: the split-K slab sums and the next step's index draw ride in the optimizer launch
        self.fused_tail = not self.dp
        self.presampled = False
        self.extras = engine.fused_extras(staged) if self.fused_tail else None
        self.dp_extras = None
        if self.dp and engine.presamples(staged):
            self.dp_extras = engine.fused_extras(staged, slabs=False)
        first_tail_seg = 2 * engine.online.ORDER.index("critic.1.0")     # (weight, bias) segment pairs per layer
        self.tail_table = engine.online.table.subset(lambda s: s >= first_tail_seg)
        self.head_table = engine.online.table.subset(lambda s: s < first_tail_seg)
        # (Measured and removed: the fc tail's optimizer update co-launched with the conv2/conv3 weight-gradient
        # GEMM -- 11.76k vs 12.51k updates/s at batch 128, the Adam workgroups' HBM streams slow the GEMM's operand
        # loads by more than the overlap returns -- and the same update forked onto a side stream under the conv
        # backward: 9.1k vs 10.6k fp32-class, 11.1k vs 13.5k bf16.)
        if self.dp:
            e = engine
            e.adam.set_grad_scale(1.0 / self.world)
            self.comm_stream = torch.cuda.Stream(device=e.device)
            self.fc_done, self.conv_done = torch.cuda.Event(), torch.cuda.Event()
            self.tail_done, self.head_done = torch.cuda.Event(), torch.cuda.Event()
            self.tail = e.grads[e.tail_start:]
            self.head = e.grads[:e.tail_start]
            # the fc tail is 95 % of the gradient bytes (6.4 MB fp32): optionally all-reduced as bf16 -- half the
            # bytes on the xGMI links, which at ~0.1 ms of compute per step is what bounds data-parallel scaling.
            # The all-reduce itself: the one-kernel xGMI peer all-reduce (parallel/peer_allreduce.py: every rank maps
            # every peer's hand-off region through hipIpc; two hops over all links, replayable inside the step graph)
            # when it passes its self-test on every rank, else the process group's (RCCL ring / gloo).
            # Knobs dp_allreduce = auto | peer | rccl, dp_grad_dtype = measure | fp32 | bf16 (ops/knobs.py): with
            # "measure" the candidates are timed here, on the real bucket, and the fastest wins.
            from ..parallel.peer_allreduce import choose_grad_allreduce, make_grad_allreduce
            kind, wg = knobs.dp_allreduce, knobs.peer_workgroups
            if not hasattr(comm, "backend") or not self.tail.is_cuda:
                kind = "rccl"           # FakeComm / CPU tensors: the communicator's own reduce
            self.comm_report, self._spare_allreduces = {"us": {}, "chosen": None}, []
            if knobs.dp_grad_dtype == "measure" and kind != "rccl":
                # (exact fp32 wires for the fp32-class engine: plain and packed; the bf16 engine may also round the wire)
                wires = ("fp32", "fp32p") if e.x3 else ("fp32", "fp32p", "bf16")
                self.ar_tail, self.comm_report, self._spare_allreduces = choose_grad_allreduce(
                    comm, self.tail, kind, wires, wg)
                head_kind = "rccl" if self.ar_tail.kind == "rccl" else kind
                self.grad_dtype = "bf16" if self.ar_tail.kind == "peer-bf16" else "fp32"
            else:
                self.grad_dtype = "fp32" if knobs.dp_grad_dtype == "measure" else knobs.dp_grad_dtype
                self.ar_tail = make_grad_allreduce(comm, self.tail, kind, wire=self.grad_dtype, workgroups=wg)
                head_kind = kind
                self.comm_report["chosen"] = self.ar_tail.kind
            self.ar_head = make_grad_allreduce(comm, self.head, head_kind, workgroups=wg)
            self.allreduce_kind = self.ar_tail.kind
            self.peer = self.allreduce_kind.startswith("peer")
            if getattr(comm, "backend", None) != "nccl" and not self.peer:
                self.grad_dtype = "fp32"        # host back-ends (gloo rehearsals, tests) reduce the fp32 buffer
            # (process-group all-reduce with a bf16 wire: two conversion launches around the collective -- ~20 us per
            # step in the 1-rank rehearsal, 9.2k vs 7.8k updates/s -- so it only pays where the wire saves more)
            self.tail16 = (torch.empty_like(self.tail, dtype=BF16)
                           if self.grad_dtype == "bf16" and not self.peer else None)

    def _part_a(self):
        if not self.staged and not (self.presampled and self.e.presamples(self.staged)):
            with _Range("learner/sample"):
                self.e.sample()
        with _Range("learner/forward+td_head"):
            self.e.forward_head(self.staged)
        with _Range("learner/backward_fc"):
            self.e.backward_fc()

    def _part_b(self):
        with _Range("learner/backward_conv"):
            self.e.backward_conv(self.staged)

    def _part_c(self):
        with _Range("learner/adam+target+tree"):
            self.e.update(self.staged)

    def _whole(self):
        e = self.e
        if not self.staged and not (self.presampled and e.presamples(self.staged)):
            with _Range("learner/sample"):
                e.sample()
        with _Range("learner/forward+td_head"):
            e.forward_head(self.staged)
        with _Range("learner/backward_fc"):
            e.backward_fc()
        with _Range("learner/backward_conv"):
            e.backward_conv(self.staged, reduce=False)
        with _Range("learner/adam+target+tree"):
            e.update(self.staged, extras=self.extras)      # + slab sums + the next step's index draw
        self.presampled = True

    def _capture(self, fns):
        graphs = []
        for fn in fns:
            gr = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may touch the runtime while this thread captures
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                fn()
            graphs.append(gr)
        return graphs

    def _run_dp(self, a, b, probe=None):
        """One data-parallel step on the current stream + the comm stream.  Works eagerly, with
        a/b replaying captured graphs, and -- because every cross-stream edge is an event --
        inside a stream capture, where the comm stream becomes a forked branch of ONE graph.
        probe: dict of timing events (measure_comm) recorded at the hand-over points of an EAGER step."""
        e, cur = self.e, torch.cuda.current_stream()

        def mark(name, stream):
            if probe is not None:
                probe[name].record(stream)

        mark("step0", cur)
        a()
        self.fc_done.record(cur)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.fc_done)
            mark("tail0", self.comm_stream)
            if self.tail16 is not None:
                self.tail16.copy_(self.tail)
                self.comm.all_reduce_sum_(self.tail16)
                self.tail.copy_(self.tail16)
            else:
                self.ar_tail.all_reduce_sum_(self.tail)     # (peer kernel: bf16 wire conversion happens in it)
            mark("tail1", self.comm_stream)
            self.tail_done.record(self.comm_stream)
        b()     # (incl. the conv bucket's slab sums; measured and removed: that launch on the comm stream, in front of
        # the bucket's all-reduce and beside the fc tail's optimizer update -- 9.0 k vs 9.9 k updates/s in the one-rank
        # rehearsal: it competes with the optimizer for bandwidth and delays the all-reduce the second update waits for)
        self.conv_done.record(cur)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.conv_done)
            mark("head0", self.comm_stream)
            self.ar_head.all_reduce_sum_(self.head)
            mark("head1", self.comm_stream)
            self.head_done.record(self.comm_stream)
        mark("wait_tail", cur)
        cur.wait_event(self.tail_done)
        e.adam.step(table=self.tail_table)          # fc tail update runs under the conv all-reduce
        mark("wait_head", cur)
        cur.wait_event(self.head_done)
        e.adam.step(table=self.head_table, extras=self.dp_extras)     # + next step's index draw (+ tree repair)
        x = self.dp_extras
        e.post_update(self.staged, repair_from=((0 if x.rep_count else 2) if (x is not None and x.rep_sum) else 1))
        mark("step1", cur)
        self.presampled = self.dp_extras is not None

    def measure_comm(self, steps=20):
        """Where a data-parallel step's communication goes, from `steps` EAGER steps with timing events at the
        hand-over points (these steps are real updates; call it outside any timed region).  -> dict, us per step:
          allreduce_tail_us / allreduce_head_us   the two gradient buckets' all-reduces, on the comm stream
          exposed_tail_us / exposed_head_us       how long the main stream had nothing to do but wait for each
                                                  (0 = the all-reduce was hidden under the conv backward / the fc
                                                  tail's optimizer update)
          step_us                                 the whole eager step (launch-bound: slower than the graph replay)"""
        if not self.dp:
            return {}
        names = ("step0", "tail0", "tail1", "head0", "head1", "wait_tail", "wait_head", "step1")
        acc = dict.fromkeys(("allreduce_tail_us", "allreduce_head_us", "exposed_tail_us", "exposed_head_us",
                             "step_us"), 0.0)
        dev = self.e.device
        for _ in range(int(steps)):
            probe = {n: torch.cuda.Event(enable_timing=True) for n in names}
            self._run_dp(self._part_a, self._part_b, probe=probe)
            torch.cuda.synchronize(dev)
            acc["allreduce_tail_us"] += 1e3 * probe["tail0"].elapsed_time(probe["tail1"])
            acc["allreduce_head_us"] += 1e3 * probe["head0"].elapsed_time(probe["head1"])
            acc["exposed_tail_us"] += max(0.0, 1e3 * probe["wait_tail"].elapsed_time(probe["tail1"]))
            acc["exposed_head_us"] += max(0.0, 1e3 * probe["wait_head"].elapsed_time(probe["head1"]))
            acc["step_us"] += 1e3 * probe["step0"].elapsed_time(probe["step1"])
            self.e.loss_steps += 1
            self.e.step_count += 1
        out = {k: round(v / max(int(steps), 1), 2) for k, v in acc.items()}
        out["steps"] = int(steps)
        out["note"] = ("eager steps with events; the exposed figures are upper bounds for the graph-replayed step, "
                       "whose compute side is shorter")
        return out

    def _dp_whole(self):
        self._run_dp(self._part_a, self._part_b)

    # ---- fail-safe: the peer kernel's waits are bounded; a wait that ran out leaves an error word ---------------------
    def comm_error(self):
        """0, or the error word of a peer all-reduce whose wait timed out on THIS rank (synchronises the device).
        The learner loop MAX-reduces it over the ranks every `stop_agree_every` steps (agents/learners.py), so a
        timed-out wait never feeds Adam for longer than that."""
        if not self.dp:
            return 0
        err = 0
        for ar in (self.ar_tail, self.ar_head):
            if hasattr(ar, "error"):
                err = max(err, abs(int(ar.error())))
        return err

    def fall_back_to_collectives(self):
        """Collective: every rank leaves the peer kernel for the process group's all-reduce (RCCL / gloo).  The
        updates since the failed wait used incomplete gradient sums, differently on each rank, so rank 0's
        parameters, target network and optimizer state are re-broadcast before the step graphs are rebuilt."""
        from ..parallel.peer_allreduce import _RcclAllReduce
        e, comm = self.e, self.comm
        torch.cuda.synchronize(e.device)
        for name in ("ar_tail", "ar_head"):
            ar = getattr(self, name)
            if hasattr(ar, "clear_error"):
                ar.clear_error()
            ar.close()
            setattr(self, name, _RcclAllReduce(comm))
        for ar in getattr(self, "_spare_allreduces", []):
            ar.close()
        self._spare_allreduces = []
        self.allreduce_kind = self.ar_tail.kind
        self.peer = False
        self.grad_dtype = "fp32"
        self.tail16 = None
        step = e.adam.step_dev.to(torch.float32).reshape(-1)
        for buf in (e.online.flat, e.target.flat, e.adam.m, e.adam.v, step):
            comm.broadcast_(buf, src=0)
        e.adam.step_dev.copy_(step.reshape(e.adam.step_dev.shape).to(e.adam.step_dev.dtype))
        e.online.refresh()
        e.target.refresh()
        self.graphs, self.calls, self.ahead = None, 0, 0
        self.presampled = False
        torch.cuda.synchronize(e.device)

    def _capture_dp(self):
        """Preferred: the whole step, collectives included, as ONE graph (one host call per
        step: 4899 vs 3810 updates/s in the 1-rank RCCL rehearsal).  RCCL's intra-node
        collectives are plain kernels and capture like any other launch; host-driven back-ends
        (gloo) cannot be captured, and a refused capture poisons the process, so the choice is
        made up front: knob dp_graph = whole | split (PDA_DP_GRAPH), or by back-end.  bench.py sets the variable from
        a pre-flight child process (parallel/preflight.py) that tries the capture first.
        Every rank takes the same decision (MIN over ranks)."""
        mode = self.knobs.dp_graph
        if self.peer and mode != "split":
            mode = "whole"      # the peer all-reduce is a plain kernel launch: nothing host-driven is left in the step
        if mode == "auto":
            # RCCL inside a graph only when a pre-flight child has shown that this node captures and replays it
            # (parallel/preflight.py sets PDA_DP_GRAPH=whole): a refused capture poisons the process
            mode = "split"
        whole = self.comm.all_agree(mode == "whole")
        if whole:
            self.dp_graph_mode = "whole"

            def k_steps():
                for _ in range(self.graph_steps):
                    self._dp_whole()
            return self._capture([k_steps])
        self.dp_graph_mode = "split"
        self.graph_steps = 1            # collectives stay host-driven: one step per pair of graphs
        return self._capture([self._part_a, self._part_b])

    def _throttle(self):
        """Block (host only) until at most max_inflight - 1 earlier replays are still queued or running."""
        ev = None
        if len(self._inflight) >= self.max_inflight:
            ev = self._inflight.pop(0)
            ev.synchronize()
        return ev if ev is not None else torch.cuda.Event()

    def _launched(self, ev):
        ev.record()
        self._inflight.append(ev)

    def step(self):
        e = self.e
        if self.ahead > 0:              # this update already ran as part of the last multi-step replay
            self.ahead -= 1
            self.calls += 1
            return
        ev = self._throttle() if (self.use_graph and e.device.type == "cuda") else None
        try:
            self._step_launch()
        finally:
            if ev is not None:
                self._launched(ev)

    def _step_launch(self):
        e = self.e
        first = (not self.use_graph) or self.calls == 0
        done = 1
        # the first step runs eagerly: it loads every code object (and lets RCCL set up its
        # channels), so nothing is lazily initialised while a stream is capturing
        if not self.dp:
            if first:
                self._whole()
            else:
                if self.graphs is None:
                    def k_steps():
                        for _ in range(self.graph_steps):
                            self._whole()
                    self.graphs = self._capture([k_steps])
                self.graphs[0].replay()
                done = self.graph_steps
                self.ahead = done - 1
        else:
            if first:
                self._dp_whole()
            else:
                if self.graphs is None:
                    self.graphs = self._capture_dp()
                if len(self.graphs) == 1:
                    self.graphs[0].replay()
                    done = self.graph_steps
                    self.ahead = done - 1
                else:
                    self._run_dp(self.graphs[0].replay, self.graphs[1].replay)
        self.calls += 1
        e.loss_steps += done
        # (with K updates per replay the period is a multiple of K and the first step ran alone, so a due
        # update is always the last one of the replay)
        e.step_count += done - 1
        if e.target_update >= 1.0 and e.step_count % int(e.target_update) == 0:
            e.hard_target_update()
        e.step_count += 1


