"""parallel/preflight.py: the throw-away child that answers "may this node capture the process group's collectives
into a hipGraph?" -- on a box without a GPU the child fails, and the answer must be "auto" (host-driven collectives,
but a peer-kernel step stays one graph), never a forced split; an explicit knob wins without starting a child."""
import os

from pytorch_distributed_amd.parallel.preflight import choose_dp_graph_mode


def test_failed_preflight_leaves_the_choice_to_the_step_program(monkeypatch):
    monkeypatch.delenv("PDA_DP_GRAPH", raising=False)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29871")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "")      # no GPU for the child, whatever the box has
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert choose_dp_graph_mode(timeout_s=120) == "auto"


def test_explicit_knob_needs_no_child(monkeypatch):
    for forced in ("whole", "split"):
        monkeypatch.setenv("PDA_DP_GRAPH", forced)
        assert choose_dp_graph_mode(timeout_s=1) == forced
    assert "PDA_DP_GRAPH" in os.environ
