"""Host-side pieces of the PPO counterpart: networks, action distribution, statistics, save/load.
The training loop itself needs the HIP physics and is covered by the GPU test `test_ppo_improves_pendulum_reward`."""
import math

import numpy as np
import pytest
import torch

from ambersim_amd.learning.architectures import MLP
from ambersim_amd.rl.helpers import NormalTanhDistribution, PPONetworksWrapper, make_ppo_networks
from ambersim_amd.rl.ppo import PPOResult, RunningStatistics


def test_mlp_shapes_and_options():
    """reference tests/learning/test_architectures.py: output size = last layer size, bias/activate_final honoured."""
    m = MLP((8, 8, 2))
    assert m(torch.zeros(5, 3)).shape == (5, 2)
    assert [tuple(l.weight.shape) for l in m.layers] == [(8, 3), (8, 8), (2, 8)]
    assert all(l.bias is None for l in MLP((4, 2), bias=False).layers)
    assert (MLP((4, 3), activate_final=True)(torch.randn(10, 2)) >= 0).all()
    import pickle
    m2 = pickle.loads(pickle.dumps(m))  # "pickle-able" is the reference's stated reason for its MLP
    x = torch.randn(4, 3)
    assert torch.equal(m(x), m2(x))


def test_network_wrapper_checks_sizes():
    """reference rl/helpers.py:45-70: wrong policy / value widths are refused."""
    ok = PPONetworksWrapper(MLP((16, 2)), MLP((16, 1))).make_ppo_networks(observation_size=3, action_size=1)
    assert ok.parametric_action_distribution.param_size == 2
    with pytest.raises(AssertionError):
        PPONetworksWrapper(MLP((16, 3)), MLP((16, 1))).make_ppo_networks(3, 1)
    with pytest.raises(AssertionError):
        PPONetworksWrapper(MLP((16, 2)), MLP((16, 2))).make_ppo_networks(3, 1)
    nets = make_ppo_networks(3, 1, policy_hidden_layer_sizes=(64,) * 3)
    assert [l.out_features for l in nets.policy_network.layers] == [64, 64, 64, 2]


def test_tanh_normal_log_prob_matches_torch():
    d = NormalTanhDistribution(2)
    p = torch.randn(50, 4, generator=torch.Generator().manual_seed(1))  # seeded: the torch reference saturates for extreme draws (tanh(raw) = +-1)
    raw = d.sample_raw(p, torch.Generator().manual_seed(0))
    loc, scale = d._split(p)
    import torch.distributions as D
    ref = D.TransformedDistribution(D.Normal(loc, scale), [D.transforms.TanhTransform()]).log_prob(torch.tanh(raw)).sum(-1)
    assert torch.allclose(d.log_prob(p, raw), ref, atol=1e-4)
    assert d.mode(p).abs().max() <= 1 and (scale > 0).all()


def test_running_statistics_and_checkpoint(tmp_path):
    rs = RunningStatistics(3)
    x = torch.randn(1000, 3, generator=torch.Generator().manual_seed(2)) * torch.tensor([1.0, 5.0, 0.1]) + torch.tensor([0.0, 2.0, -1.0])
    for chunk in x.split(100):
        rs.update(chunk)
    assert torch.allclose(rs.mean.float(), x.mean(0), atol=1e-5) and torch.allclose(rs.std.float(), x.std(0, unbiased=False), atol=1e-4)
    assert rs.normalize(x).abs().max() <= 5.0
    nets = make_ppo_networks(3, 1)
    nets.policy_network(torch.zeros(1, 3)); nets.value_network(torch.zeros(1, 3))
    res = PPOResult(nets, rs, [])
    res.save(str(tmp_path / "ckpt.pt"))
    nets2 = make_ppo_networks(3, 1)
    nets2.policy_network(torch.zeros(1, 3)); nets2.value_network(torch.zeros(1, 3))
    res2 = PPOResult(nets2, RunningStatistics(3), [])
    res2.load(str(tmp_path / "ckpt.pt"))
    obs = torch.randn(6, 3)
    assert torch.equal(res.make_inference_fn()(obs), res2.make_inference_fn()(obs))


def test_mlp_construction_and_pickle(tmp_path):
    """reference tests/test_mlp.py:12-76: parameter count = sum of weights and biases; a pickled MLP gives the same output."""
    import pickle
    input_size, layer_sizes = 3, (2, 3, 4)
    mlp = MLP(layer_sizes=layer_sizes, bias=True)
    out = mlp(torch.randn(input_size))
    assert out.shape[-1] == layer_sizes[-1]
    sizes = (input_size,) + layer_sizes
    expected = sum(sizes[i] * sizes[i + 1] + sizes[i + 1] for i in range(len(sizes) - 1))
    assert sum(p.numel() for p in mlp.parameters()) == expected
    x = torch.ones(3)
    with open(tmp_path / "mlp.pkl", "wb") as f:
        pickle.dump(mlp, f)
    with open(tmp_path / "mlp.pkl", "rb") as f:
        new_mlp = pickle.load(f)
    assert torch.allclose(mlp(x), new_mlp(x))


def test_ppo_wrapper_and_io(tmp_path):
    """reference tests/test_ppo_networks.py:15-90 with the same layer sizes."""
    import pickle
    from ambersim_amd.rl.helpers import PPONetworks
    with pytest.raises(AssertionError):  # policy width must be mean + scale of the action distribution
        PPONetworksWrapper(policy_network=MLP(layer_sizes=(512, 2)), value_network=MLP(layer_sizes=(512, 1)),
                           action_distribution=NormalTanhDistribution).make_ppo_networks(observation_size=3, action_size=2)
    with pytest.raises(AssertionError):  # the value network must be scalar
        PPONetworksWrapper(policy_network=MLP(layer_sizes=(512, 4)), value_network=MLP(layer_sizes=(512, 2)),
                           action_distribution=NormalTanhDistribution).make_ppo_networks(observation_size=3, action_size=2)
    wrapper = PPONetworksWrapper(policy_network=MLP(layer_sizes=(512, 4)), value_network=MLP(layer_sizes=(512, 1)), action_distribution=NormalTanhDistribution)
    nets = wrapper.make_ppo_networks(observation_size=3, action_size=2)
    assert isinstance(nets, PPONetworks)
    with open(tmp_path / "ppo_networks.pkl", "wb") as f:
        pickle.dump(wrapper, f)
    with open(tmp_path / "ppo_networks.pkl", "rb") as f:
        new_wrapper = pickle.load(f)
    new_nets = new_wrapper.make_ppo_networks(observation_size=3, action_size=2)
    assert isinstance(new_nets, PPONetworks)
    same = lambda a, b: [(k, tuple(v.shape)) for k, v in a.state_dict().items()] == [(k, tuple(v.shape)) for k, v in b.state_dict().items()]
    assert same(nets.policy_network, new_nets.policy_network) and same(nets.value_network, new_nets.value_network)
    obs = torch.randn(5, 3)
    assert torch.equal(nets.policy_network(obs), new_nets.policy_network(obs))


def test_introspection_names(mj_models):
    from ambersim_amd.utils.introspection_utils import get_actuator_names, get_equality_names, get_geom_names, get_joint_names
    p = mj_models["pendulum"]
    assert get_actuator_names(p) == ["pendulum_joint_actuator"] and get_equality_names(p) == [] and len(get_joint_names(p)) == 1
    assert len(get_geom_names(mj_models["quadruped"])) == mj_models["quadruped"].ngeom
