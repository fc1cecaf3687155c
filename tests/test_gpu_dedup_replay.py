"""Frame-deduplicated replay on the GPU (memories/dedup_memory.py): the run-unpack kernel against what the actors
sent, and the learner step reading frame windows of the plane store (ops/nn.py FrameWindows, csrc/nature_cnn.h
frame_index) against the same step reading rows of full frame stacks -- bit for bit, both engines, small and large
batch (the stack kernels and the whole-image conv1 kernels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def native():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from pytorch_distributed_amd import ops
    ops.require_native()


def options(memory_type, size, batch=32, precision="fp32", **mp):
    from pytorch_distributed_amd.launcher import prepare_options
    from pytorch_distributed_amd.utils.options import Options
    ov = {"device": "cuda", "memory_type": memory_type, "memory_params.memory_size": size,
          "memory_params.replay_device": "cuda", "agent_params.batch_size": batch, "precision": precision,
          "graph_steps": 1}
    ov.update({"memory_params." + k: v for k, v in mp.items()})
    return prepare_options(Options(config=2, overrides=ov))


def test_unpack_kernel_holds_what_the_actors_sent():
    from test_dedup_memory import check_rows, play
    from pytorch_distributed_amd.agents.nstep import NStepAssembler, make_feeder
    from pytorch_distributed_amd.utils.factory import MemoriesDict
    opt = options("device-dedup-per", 1024)
    m = MemoriesDict[opt.memory_type](opt.memory_params)
    rng = np.random.RandomState(0)
    f = make_feeder(m, 5, 0.99, enable_per=True)
    m.attach("cuda")
    expected = []
    for _ in range(6):
        expected += play(f, rng, 6, 50, NStepAssembler(5, 0.99, enable_per=True), per=True)
        for _ in range(200):
            m.drain()
            torch.cuda.synchronize()
            if m.trans_linear == len(expected):
                break
    assert m.trans_linear == len(expected) > 300
    n = min(len(expected), 1024)
    lin = list(range(len(expected) - n, len(expected)))

    class Cpu(object):          # check_rows reads through _gather
        def _gather(self, idx):
            return tuple(t.cpu() for t in m._gather(idx.cuda()))
    check_rows(Cpu(), [j % 1024 for j in lin], [expected[j] for j in lin])
    w = m.window_dev.cpu().tolist()
    assert tuple(w[:3]) == m.window() == (n, (len(expected) - n) % 1024, 1024)
    # initial priorities went into the tree
    pr = torch.tensor([p for _, p in expected[-n:]], dtype=torch.float32)
    want = (pr.abs() + m.priority_eps) ** m.priority_exponent
    got = m.tree.get(torch.tensor([j % 1024 for j in lin], device="cuda")).cpu()
    assert torch.allclose(got, want, rtol=1e-4)
    m.close()


@pytest.mark.parametrize("precision,batch,per", [("fp32", 32, False), ("bf16", 32, True), ("fp32", 512, False),
                                                 ("bf16", 512, False)])
def test_step_on_frame_windows_equals_step_on_full_stacks(precision, batch, per):
    from pytorch_distributed_amd.ops.dqn_engine import DqnNativeCore
    from pytorch_distributed_amd.utils.factory import MemoriesDict, ModelsDict
    n = 1024
    kind = "-per" if per else ""
    g = torch.Generator(device="cuda").manual_seed(5)
    s = torch.randint(0, 256, (n + 1, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
    a = torch.randint(0, 6, (n, 1), device="cuda", generator=g)
    r = torch.randn(n, 1, device="cuda", generator=g).clamp_(-1, 1)
    t = (torch.rand(n, 1, device="cuda", generator=g) < 0.05).float()
    pr = torch.rand(n, device="cuda", generator=g) + 0.05
    flats = []
    for memory_type, mp in (("device" + kind, {}), ("device-dedup" + kind, {"dedup_plane_slack": 8.5})):
        opt = options(memory_type, n, batch, precision, **mp)
        memory = MemoriesDict[opt.memory_type](opt.memory_params)
        memory.attach("cuda")
        memory.feed_batch(s[:-1], a, r, torch.full((n, 1), 0.95, device="cuda"), s[1:], t, priorities=pr)
        assert memory.size == n
        torch.manual_seed(3)
        model = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space, 1).cuda()
        target = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space, 1).cuda()
        core = DqnNativeCore(opt, model, target, memory)
        assert core.dedup == ("dedup" in memory_type)
        for _ in range(6):
            core.step()
        torch.cuda.synchronize()
        flats.append((core.flat_params().clone(), core.engine.idx.clone(), core.engine.td.clone()))
        memory.close()
    assert torch.equal(flats[0][1], flats[1][1])        # the same rows were drawn
    assert torch.equal(flats[0][2], flats[1][2])        # ... gave the same TD errors
    assert torch.equal(flats[0][0], flats[1][0])        # ... and the same parameters after six updates


def test_synthetic_fill_and_window_draw():
    from pytorch_distributed_amd.utils.factory import MemoriesDict
    opt = options("device-dedup", 4096)
    m = MemoriesDict[opt.memory_type](opt.memory_params)
    m.attach("cuda")
    m.fill_synthetic(4096, nstep=5, num_actions=6)
    assert m.window_dev.cpu().tolist()[:3] == [4096, 0, 4096]
    s0, a, r, g, s1, t = m.sample(16)
    assert s0.shape == (16, 4, 84, 84) and s0.dtype == torch.uint8 and float(a.max()) <= 5
    # plane p + 1 of s0's window is plane p of the window one step later
    i = torch.arange(8, device="cuda", dtype=torch.int32)
    assert torch.equal(m.states(m.ref0, i)[1:, 0], m.states(m.ref0, i)[:-1, 1])
    m.close()


@pytest.mark.parametrize("size", [4096, 50000, 300000])
def test_prioritized_step_keeps_the_tree_consistent_without_repair_launches_of_its_own(size):
    """The first repair level rides in the optimizer launch; trees of up to three levels are completed there (the last
    repair block recomputes the top level), larger ones get their top level re-derived by the sampling kernel
    (ops/segtree.py lazy_top): after some updates every inner node still is the sum / min of its children."""
    from pytorch_distributed_amd.ops.dqn_engine import DqnNativeCore
    from pytorch_distributed_amd.ops.segtree import RADIX
    from pytorch_distributed_amd.utils.factory import MemoriesDict, ModelsDict
    opt = options("device-dedup-per", size, 64, "bf16")
    memory = MemoriesDict[opt.memory_type](opt.memory_params)
    memory.attach("cuda")
    memory.fill_synthetic(size, nstep=5, num_actions=6)
    torch.manual_seed(0)
    model = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space, 1).cuda()
    target = ModelsDict[opt.model_type](opt.model_params, opt.norm_val, opt.state_shape, opt.action_space, 1).cuda()
    core = DqnNativeCore(opt, model, target, memory)
    tree = memory.tree
    assert tree.depth == {4096: 2, 50000: 3, 300000: 4}[size] and tree.lazy_top == (tree.depth >= 4)
    before = tree.sum[:size].clone()
    for _ in range(12):
        core.step()
    if tree.lazy_top:
        core.engine.sample()        # the next step's draw refreshes the stored top level
    torch.cuda.synchronize()
    assert not torch.equal(before, tree.sum[:size])         # priorities were written back
    for lvl in range(1, tree.depth):
        avail = tree.sizes[lvl - 1] // RADIX        # nodes of this level that have (whole rows of) children
        kids_s = tree.sum[tree.offsets[lvl - 1]:tree.offsets[lvl - 1] + avail * RADIX]
        kids_m = tree.min[tree.offsets[lvl - 1]:tree.offsets[lvl - 1] + avail * RADIX]
        want_s = kids_s[:avail * RADIX].view(avail, RADIX).double().sum(1)
        want_m = kids_m[:avail * RADIX].view(avail, RADIX).min(1)[0]
        got_s = tree.sum[tree.offsets[lvl]:tree.offsets[lvl] + avail].double()
        got_m = tree.min[tree.offsets[lvl]:tree.offsets[lvl] + avail]
        assert torch.allclose(got_s, want_s, rtol=1e-5, atol=1e-6), lvl
        assert torch.equal(got_m, want_m), lvl
    assert abs(float(tree.total()) - float(tree.sum[:size].double().sum())) < 1e-3 * float(tree.total())
    memory.close()


@pytest.mark.parametrize("memory_type", ["device-per", "device-dedup-per"])
def test_ingest_writes_leaves_in_the_unpack_kernel_and_repairs_in_one_launch(memory_type):
    """Prioritized ingestion on the GPU: the unpack kernel writes (|p| + eps)^alpha into both trees and
    DeviceSumMinTree.repair_range repairs above the chunk's consecutive leaves in one single-workgroup launch, across
    ring wrap-arounds; afterwards every inner node (up to the level the sampler re-derives) is the sum / min of its
    children and the leaves are what the actors' initial priorities say."""
    from test_dedup_memory import play
    from pytorch_distributed_amd.agents.nstep import NStepAssembler, make_feeder
    from pytorch_distributed_amd.ops.segtree import RADIX
    from pytorch_distributed_amd.utils.factory import MemoriesDict
    size = 4500                 # depth 3; the ring wraps below
    opt = options(memory_type, size, **({"ingest_chunk": 64, "ingest_ring_slots": 1024} if "dedup" not in memory_type
                                        else {}))
    m = MemoriesDict[opt.memory_type](opt.memory_params)
    rng = np.random.RandomState(1)
    f = make_feeder(m, 5, 0.99, enable_per=True)
    m.attach("cuda")
    expected = []
    for _ in range(24):
        expected += play(f, rng, 8, 90, NStepAssembler(5, 0.99, enable_per=True), per=True)
        for _ in range(400):
            m.drain()
            torch.cuda.synchronize()
            if m.total_fed.value == len(expected):
                break
        assert m.total_fed.value == len(expected)
    assert len(expected) > size + 2000
    tree = m.tree
    lin = list(range(len(expected) - size, len(expected)))
    pr = torch.tensor([expected[j][1] for j in lin], dtype=torch.float32)
    want = (pr.abs() + m.priority_eps) ** m.priority_exponent
    got = tree.get(torch.tensor([j % size for j in lin], device="cuda")).cpu()
    assert torch.allclose(got, want, rtol=1e-4)
    for lvl in range(1, tree.depth - 1):
        avail = tree.sizes[lvl - 1] // RADIX
        kids_s = tree.sum[tree.offsets[lvl - 1]:tree.offsets[lvl - 1] + avail * RADIX].view(avail, RADIX)
        kids_m = tree.min[tree.offsets[lvl - 1]:tree.offsets[lvl - 1] + avail * RADIX].view(avail, RADIX)
        assert torch.allclose(tree.sum[tree.offsets[lvl]:tree.offsets[lvl] + avail].double(), kids_s.double().sum(1),
                              rtol=1e-5)
        assert torch.equal(tree.min[tree.offsets[lvl]:tree.offsets[lvl] + avail], kids_m.min(1)[0])
    assert abs(float(tree.total()) - float(want.double().sum())) < 1e-3 * float(want.sum())
    idx, w = m.sample_indices(256)
    assert bool((idx >= 0).all()) and bool((idx < size).all()) and bool(torch.isfinite(w).all())
    m.close()
