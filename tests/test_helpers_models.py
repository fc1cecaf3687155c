import numpy as np
import pytest
import torch

from pytorch_distributed_amd.utils.helpers import Experience, reset_experience, update_target_model
from pytorch_distributed_amd.utils.random_process import OrnsteinUhlenbeckProcess
from pytorch_distributed_amd.utils.checkpoint import save_state_dict, load_state_dict
from pytorch_distributed_amd.utils.options import Options
from pytorch_distributed_amd.models.dqn_cnn_model import DqnCnnModel
from pytorch_distributed_amd.models.dqn_mlp_model import DqnMlpModel
from pytorch_distributed_amd.models.ddpg_mlp_model import DdpgMlpModel


def cnn(actions=6):
    return DqnCnnModel(Options(config=1).model_params, 255.0, [4, 84, 84], actions, 1)


def ddpg(obs=3, act=1):
    return DdpgMlpModel(Options(config=3).model_params, 1.0, [1, 1, obs], act, act)


def test_experience_tuple():
    e = reset_experience()
    assert e == Experience(None, None, None, None, False)
    assert Experience._fields == ("state0", "action", "reward", "state1", "terminal1")


def test_dqn_cnn_keys_shapes_and_count():
    sd = cnn().state_dict()
    expect = {"critic.0.0.weight": (32, 4, 8, 8), "critic.0.0.bias": (32,),
              "critic.0.2.weight": (64, 32, 4, 4), "critic.0.2.bias": (64,),
              "critic.0.4.weight": (64, 64, 3, 3), "critic.0.4.bias": (64,),
              "critic.1.0.weight": (512, 3136), "critic.1.0.bias": (512,),
              "critic.1.2.weight": (6, 512), "critic.1.2.bias": (6,)}
    assert {k: tuple(v.shape) for k, v in sd.items()} == expect
    assert sum(v.numel() for v in sd.values()) == 1687206


def test_ddpg_keys_shapes():
    sd = ddpg().state_dict()
    expect = {"actor.0.weight": (300, 3), "actor.0.bias": (300,), "actor.2.weight": (200, 300),
              "actor.2.bias": (200,), "actor.4.weight": (1, 200), "actor.4.bias": (1,),
              "critic.0.0.weight": (400, 3), "critic.0.0.bias": (400,),
              "critic.1.0.weight": (300, 401), "critic.1.0.bias": (300,),
              "critic.1.2.weight": (1, 300), "critic.1.2.bias": (1,)}
    assert {k: tuple(v.shape) for k, v in sd.items()} == expect
    m = ddpg()
    assert m.actor[4].weight.abs().max() <= 3e-3 and m.critic[1][2].weight.abs().max() <= 3e-3
    assert m.actor[0].bias.abs().sum() == 0


def test_dqn_mlp_keys():
    m = DqnMlpModel(Options(config=6).model_params, 1.0, [1, 1, 5], 2, 1)
    assert list(m.state_dict()) == ["critic.%d.%s" % (i, n) for i in (0, 2, 4, 6) for n in ("weight", "bias")]
    a, q, mq = m.get_action(np.zeros((1, 1, 5), np.float32), enable_per=True, eps=0.0)
    assert a.shape == (1, 1) and q == mq


def test_get_action_contract():
    m = cnn()
    s = np.random.randint(0, 255, (4, 84, 84)).astype(np.uint8)
    a, q, mq = m.get_action(s, enable_per=False, eps=0.0)
    assert a.shape == (1, 1) and q == mq and isinstance(q, float)
    rng = np.random.RandomState(0)
    a, q, mq = m.get_action(s, enable_per=False, eps=1.0, rng=rng)      # random: forward skipped
    assert q is None and mq is None and a.shape == (1, 1)
    a, q, mq = m.get_action(s, enable_per=True, eps=1.0, rng=rng)       # PER: q of the random action
    assert q is not None and mq >= q
    d = ddpg(3, 2)
    a, q, mq = d.get_action(np.zeros((1, 1, 3), np.float32), noise=np.array([0.5, -0.5]))
    assert a.shape == (1, 2) and (q, mq) == (0.0, 0.0)


def test_flatten_keeps_state_dict_and_views():
    m = cnn()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    flat = m.flatten_()
    assert flat.numel() == 1687206
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd0[k])
    flat.zero_()
    assert all(p.abs().sum() == 0 for p in m.parameters())
    x = torch.randint(0, 255, (2, 4, 84, 84), dtype=torch.uint8)
    m.flat_params.normal_(0, 0.01)
    m(x).sum().backward()
    assert m.flat_grads.abs().sum() > 0
    segs = m.param_segments()
    assert segs[0] == ("critic.0.0.weight", 0, (32, 4, 8, 8)) and segs[1][1] == 8192


def test_target_update_hard_and_soft():
    a, b = ddpg(), ddpg()
    update_target_model(a, b)       # default: hard copy
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
    with torch.no_grad():
        for p in a.parameters():
            p.add_(1.0)
    update_target_model(a, b, 250, learner_step=3)      # not a multiple: unchanged
    assert not torch.equal(next(a.parameters()), next(b.parameters()))
    update_target_model(a, b, 250, learner_step=500)
    assert torch.equal(next(a.parameters()), next(b.parameters()))
    with torch.no_grad():
        for p in a.parameters():
            p.add_(1.0)
    before = [q.clone() for q in b.parameters()]
    update_target_model(a, b, 1e-3)
    for p, q, q0 in zip(a.parameters(), b.parameters(), before):
        assert torch.allclose(q, q0 * (1 - 1e-3) + p * 1e-3, atol=1e-6)


def test_checkpoint_roundtrip_and_format(tmp_path):
    m = cnn()
    path = str(tmp_path / "models" / "box_1.pth")
    save_state_dict(m, path)
    sd = load_state_dict(path)
    assert list(sd) == list(m.state_dict())
    assert all(v.device.type == "cpu" for v in sd.values())
    m2 = cnn()
    m2.load_state_dict(sd)
    x = torch.randint(0, 255, (3, 4, 84, 84), dtype=torch.uint8)
    assert torch.equal(m(x), m2(x))
    raw = torch.load(path, weights_only=True)      # plain OrderedDict[str, Tensor], like the reference
    assert isinstance(raw, dict) and all(torch.is_tensor(v) for v in raw.values())


def test_ou_process_statistics():
    p = OrnsteinUhlenbeckProcess(theta=0.15, sigma=0.3, size=1, seed=0)
    xs = np.array([p.sample()[0] for _ in range(20000)])
    assert abs(xs.mean()) < 0.3
    # stationary std of dx = -theta x dt + sigma sqrt(dt) N: sigma / sqrt(2 theta)
    assert 0.3 < xs[5000:].std() < 0.8
    assert p.current_sigma == 0.3       # sigma_min=None: never anneals (reference quirk)
    p.reset_states()
    assert p.x_prev[0] == 0
