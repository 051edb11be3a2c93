"""State-dict layout of the PyTorch-side networks vs the reference's (tests/golden/checkpoint_layout.json, dumped
from keisei.training.models.resnet_tower.ActorCriticResTower / keisei.core.neural_network.ActorCritic), so that
reference checkpoints load into this package's models."""
import json
import os

import pytest
import torch

from shogidrl_amd.models import ResTowerActorCritic, TinyActorCritic, model_factory

G = os.path.join(os.path.dirname(__file__), "golden")
LAYOUT = json.load(open(os.path.join(G, "checkpoint_layout.json")))


def _layout(m):
    return [[k, list(v.shape)] for k, v in m.state_dict().items()]


def test_resnet_tower_state_dict_matches_reference():
    assert _layout(ResTowerActorCritic(46, 13527, 9, 256, 0.25)) == LAYOUT["resnet_9x256_se025"]
    assert _layout(ResTowerActorCritic(46, 13527, 2, 16, None)) == LAYOUT["resnet_2x16_nose"]
    m = model_factory("resnet", tower_depth=9, tower_width=256, se_ratio=0.25)
    assert sum(p.numel() for p in m.parameters()) == 13241300


def test_tiny_cnn_state_dict_matches_reference():
    assert _layout(TinyActorCritic(46, 13527)) == LAYOUT["tiny_cnn"]
    with pytest.raises(ValueError):
        model_factory("no_such_model")


def test_forward_shapes_cpu():
    m = ResTowerActorCritic(46, 13527, 1, 8, 0.25)
    logits, value = m(torch.zeros(3, 46, 9, 9))
    assert logits.shape == (3, 13527) and value.shape == (3,) and logits.dtype == torch.float32


@pytest.mark.gpu
def test_agent_checkpoint_round_trip_and_reference_keys(tmp_path):
    from shogidrl_amd.agent import VecPPOAgent
    a = VecPPOAgent(ResTowerActorCritic(46, 13527, 2, 16, None), device="cuda:0")
    f = str(tmp_path / "ck.pth")
    a.save_model(f, global_timestep=123, total_episodes_completed=4, stats_to_save={"black_wins": 1, "white_wins": 2, "draws": 1})
    ck = torch.load(f, weights_only=True)
    assert sorted(ck.keys()) == LAYOUT["save_model_keys"]
    assert sorted(ck["optimizer_state_dict"].keys()) == LAYOUT["optimizer_state_dict_keys"]
    b = VecPPOAgent(ResTowerActorCritic(46, 13527, 2, 16, None), device="cuda:0")
    res = b.load_model(f)
    assert res["global_timestep"] == 123 and res["black_wins"] == 1 and res["draws"] == 1 and "error" not in res
    for (k1, v1), (k2, v2) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert b.load_model(str(tmp_path / "missing.pth"))["error"] == "File not found"


def test_update_graph_option_is_refused_without_the_runtime_setting(monkeypatch):
    """PPOConfig.graph_minibatch_step needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (profiles/r02_update_graph_corruption.txt): the
    constructor says so before it touches the device"""
    from shogidrl_amd.agent import PPOConfig, VecPPOAgent
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        VecPPOAgent(TinyActorCritic(), PPOConfig(graph_minibatch_step=True), device="cpu")
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        VecPPOAgent(TinyActorCritic(), PPOConfig(graph_minibatch_step=True), device="cpu")
