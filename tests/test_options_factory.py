"""Options/factory wiring (unit tier)."""
import pickle

import numpy as np

from pytorch_distributed_amd.utils.options import Options, CONFIGS, parse_overrides
from pytorch_distributed_amd.utils import factory


def test_reference_defaults_dqn():
    o = Options(config=0)
    ap = o.agent_params
    assert (o.agent_type, o.env_type, o.game, o.memory_type, o.model_type) == tuple(CONFIGS[0])
    assert o.num_actors == 8 and o.num_learners == 1 and o.num_envs_per_actor == 1 and o.seed == 100
    assert ap.steps == 500000 and ap.gamma == 0.99 and ap.lr == 1e-4 and ap.weight_decay == 0.0
    assert ap.clip_grad == np.inf and ap.actor_sync_freq == 100 and ap.logger_freq == 15
    assert ap.actor_freq == 250 and ap.learner_freq == 100 and ap.evaluator_freq == 30
    assert ap.evaluator_nepisodes == 2 and ap.tester_nepisodes == 50
    assert ap.learn_start == 5000 and ap.batch_size == 128 and ap.target_model_update == 250
    assert ap.nstep == 5 and ap.enable_double is False and ap.eps == 0.4 and ap.eps_alpha == 7
    assert o.memory_params.memory_size == 50000 and o.memory_params.enable_per is False
    assert o.memory_params.priority_exponent == 0.5 and o.memory_params.priority_weight == 0.4
    assert (o.env_params.state_cha, o.env_params.state_hei, o.env_params.state_wid) == (4, 84, 84)
    assert o.env_params.early_stop == 12500
    assert o.model_name.endswith("/models/%s_%s.pth" % (o.machine, o.timestamp))
    assert o.log_dir.endswith("/logs/%s_%s/" % (o.machine, o.timestamp))


def test_reference_defaults_ddpg():
    o = Options(config=3)
    ap = o.agent_params
    assert o.agent_type == "ddpg" and o.model_type == "ddpg-mlp"
    assert ap.clip_grad == 40.0 and ap.actor_sync_freq == 400 and ap.actor_freq == 2500
    assert ap.learner_freq == 1000 and ap.evaluator_freq == 60 and ap.learn_start == 250
    assert ap.batch_size == 64 and ap.target_model_update == 1e-3 and ap.nstep == 5
    assert ap.random_process.__name__ == "OrnsteinUhlenbeckProcess"
    assert (o.env_params.state_cha, o.env_params.state_hei) == (1, 1)


def test_mode2_points_model_file_at_model_name():
    o = Options(config=1, overrides={"mode": 2})
    assert o.model_file == o.model_name and o.visualize is False


def test_overrides_and_presets():
    o = Options(config=1)
    assert o.num_actors == 2 and o.agent_params.nstep == 1 and o.device == "cpu"
    o = Options(config=4)
    assert o.num_actors == 256 and o.memory_params.enable_per and o.memory_params.memory_size == 2 * 1024 * 1024
    o = Options(config=5)
    assert o.agent_params.batch_size == 4096
    o = Options(config=0, overrides={"num_actors": 3, "agent_params.steps": 77, "machine": "box"})
    assert o.num_actors == 3 and o.env_params.num_actors == 3 and o.agent_params.steps == 77
    assert "box_" in o.model_name and "box_" in o.agent_params.log_dir
    assert parse_overrides("a=1,b.c=x,d=inf,e=None") == {"a": 1, "b.c": "x", "d": float("inf"), "e": None}


def test_options_pickle_roundtrip():
    o = Options(config=3, overrides={"num_actors": 2})
    o2 = pickle.loads(pickle.dumps(o))
    assert o2.agent_params.batch_size == 64 and o2.num_actors == 2
    assert o2.agent_params.optim is o.agent_params.optim


def test_factory_names_match_reference():
    for name in ("GlobalLogsDict", "ActorLogsDict", "LearnerLogsDict", "EvaluatorLogsDict", "LoggersDict",
                 "ActorsDict", "LearnersDict", "EvaluatorsDict", "TestersDict", "EnvsDict", "MemoriesDict",
                 "ModelsDict"):
        assert isinstance(getattr(factory, name), dict), name
    assert set(factory.ActorsDict) == {"dqn", "ddpg"}
    assert {"atari", "synthetic", "pendulum"} <= set(factory.EnvsDict)
    assert {"shared", "none", "device", "device-per", "device-dedup", "device-dedup-per"} <= set(factory.MemoriesDict)
    o9 = Options(config=9, overrides={"device": "cpu"})     # Ape-X row on frame-deduplicated replay
    assert o9.memory_type == "device-dedup-per" and o9.memory_params.enable_per
    assert o9.memory_params.memory_size == 2 * 1024 * 1024 and o9.memory_params.dedup_plane_cap == 96
    assert {"dqn-cnn", "dqn-mlp", "ddpg-mlp"} == set(factory.ModelsDict)
    assert factory.MemoriesDict["none"] is None


def test_reference_import_paths():
    from utils.options import Options as O2            # noqa: F401
    from utils.factory import ModelsDict               # noqa: F401
    from utils.helpers import Experience, update_target_model   # noqa: F401
    from core.models.dqn_cnn_model import DqnCnnModel  # noqa: F401
    from core.single_processes.logs import GlobalLogs  # noqa: F401
    from core.memories.shared_memory import SharedMemory   # noqa: F401
    assert O2 is Options


def test_engine_knobs_resolve_from_options_and_environment(monkeypatch):
    """ops/knobs.py is the one place that reads PDA_<NAME> tuning variables: defaults < Options.engine_knobs < env."""
    from pytorch_distributed_amd.ops.knobs import DEFAULTS, resolve
    from pytorch_distributed_amd.utils.options import Options
    for name in DEFAULTS:
        monkeypatch.delenv("PDA_" + name.upper(), raising=False)
    k = resolve()
    assert k.as_dict() == DEFAULTS and k.dp_grad_dtype == "measure"
    opt = Options(config=1, overrides={"engine_knobs.fc1_blocks": 128, "engine_knobs.dp_graph": "split"})
    k = resolve(opt)
    assert k.fc1_blocks == 128 and k.dp_graph == "split" and k.pair_blocks == DEFAULTS["pair_blocks"]
    monkeypatch.setenv("PDA_FC1_BLOCKS", "64")
    monkeypatch.setenv("PDA_DP_ALLREDUCE", "PEER")
    k = resolve(opt)
    assert k.fc1_blocks == 64 and k.dp_allreduce == "peer"
    assert resolve(opt, fc1_blocks=32).fc1_blocks == 32
    import pytest
    with pytest.raises(ValueError):
        resolve(Options(config=1, overrides={"engine_knobs.no_such_knob": 1}))
    with pytest.raises(ValueError):
        resolve(dp_graph="sideways")
    with pytest.raises(ValueError):
        resolve(wgrad_blocks=-1)
