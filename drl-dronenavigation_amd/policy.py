"""The policy/value network the reference trains, as a plain torch module for the rollout collector and benches.

Reference: PBDroneSimulator.setup_agent builds SB3 `PPO(ActorCriticPolicy, policy_kwargs=dict(activation_fn=Tanh,
net_arch=dict(pi=[512, 512, 256], vf=[512, 512, 256])))` (Sol/Model/PBDroneSimulator.py:251-286): separate
actor and critic MLPs over the 13-float observation, a linear action head with a state-independent log-std
(diagonal Gaussian, log_std_init = 0) and a linear value head; SB3 initialises orthogonally (gain sqrt(2) for the
trunk, 0.01 for the action head, 1 for the value head) [3P-recall].  The GEMMs go to rocBLAS/hipBLASLt through
torch (plain library GEMMs); this module is host-side plumbing around the step kernel, not part of its roofline.
"""
import math

import torch
from torch import nn


def _mlp(sizes, act):
    layers = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        layers += [nn.Linear(a, b), act()]
    return nn.Sequential(*layers)


class MlpActorCritic(nn.Module):
    def __init__(self, obs_dim=13, act_dim=4, pi=(512, 512, 256), vf=(512, 512, 256), log_std_init=0.0,
                 ortho_init=True, trunk_dtype=None):
        """trunk_dtype=torch.bfloat16 runs the two trunks on the bf16 matrix cores (hipBLASLt); the heads, the
        Gaussian sample and the log-probability stay float32 (the useful action band is 0.0073 wide around 0.09:
        bf16 would leave 15 distinct thrust commands).  Default None = float32 throughout, as SB3."""
        super().__init__()
        self.trunk_dtype = trunk_dtype
        self.pi = _mlp((obs_dim,) + tuple(pi), nn.Tanh)
        self.vf = _mlp((obs_dim,) + tuple(vf), nn.Tanh)
        self.action_net = nn.Linear(pi[-1], act_dim)
        self.value_net = nn.Linear(vf[-1], 1)
        self.log_std = nn.Parameter(torch.full((act_dim,), float(log_std_init)))
        if ortho_init:
            for mod, gain in ((self.pi, math.sqrt(2)), (self.vf, math.sqrt(2)), (self.action_net, 0.01), (self.value_net, 1.0)):
                for m in mod.modules():
                    if isinstance(m, nn.Linear):
                        nn.init.orthogonal_(m.weight, gain=gain)
                        nn.init.zeros_(m.bias)

    def _trunk(self, net, obs):
        if self.trunk_dtype is None:
            return net(obs)
        with torch.autocast(device_type=obs.device.type, dtype=self.trunk_dtype):
            return net(obs).float()

    def _dist(self, obs):
        return self.action_net(self._trunk(self.pi, obs)), self.log_std.expand(obs.shape[0], -1)

    @staticmethod
    def _log_prob(actions, mean, log_std):
        z = (actions - mean) * torch.exp(-log_std)
        return (-0.5 * z * z - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)

    def forward(self, obs, deterministic=False):
        """obs [N, obs_dim] -> (actions [N, act_dim] (unclipped, as SB3 stores them), values [N], log_prob [N])."""
        mean, log_std = self._dist(obs)
        actions = mean if deterministic else mean + torch.exp(log_std) * torch.randn_like(mean)
        return actions, self.value_net(self._trunk(self.vf, obs)).squeeze(-1), self._log_prob(actions, mean, log_std)

    def predict_values(self, obs):
        return self.value_net(self._trunk(self.vf, obs)).squeeze(-1)

    def evaluate_actions(self, obs, actions):
        mean, log_std = self._dist(obs)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + log_std).sum(-1)
        return self.predict_values(obs), self._log_prob(actions, mean, log_std), entropy


LOG_STD_MIN, LOG_STD_MAX = -20.0, 2.0          # SB3 sac/policies.py [3P-recall]


class SacActor(nn.Module):
    """The actor of the reference's SAC agent: SB3 `SAC("MlpPolicy", policy_kwargs=dict(activation_fn=ReLU,
    net_arch=dict(qf=[256, 256, 128], pi=[256, 256]), share_features_extractor=False), use_sde=False)`
    (Sol/Model/PBDroneSimulator.py:297-338).  SB3's Actor [3P-recall]: latent_pi = MLP(obs -> 256 -> 256, ReLU), two linear
    heads `mu` and `log_std` (clamped to [-20, 2]), action = tanh(mu + exp(log_std) eps): a squashed Gaussian in (-1, 1);
    default torch initialisation (SAC does not use SB3's orthogonal init).  Only the actor takes part in collection
    (BASELINE config 5); the critics belong to the learner."""

    def __init__(self, obs_dim=13, act_dim=4, pi=(256, 256)):
        super().__init__()
        self.latent_pi = _mlp((obs_dim,) + tuple(pi), nn.ReLU)
        self.mu = nn.Linear(pi[-1], act_dim)
        self.log_std = nn.Linear(pi[-1], act_dim)

    def mean_log_std(self, obs):
        z = self.latent_pi(obs)
        return self.mu(z), torch.clamp(self.log_std(z), LOG_STD_MIN, LOG_STD_MAX)

    @staticmethod
    def squash(mean, log_std, eps=None):
        """actions in (-1, 1) and their log-probability (SquashedDiagGaussianDistribution, epsilon = 1e-6)."""
        pre = mean if eps is None else mean + torch.exp(log_std) * eps
        act = torch.tanh(pre)
        z = (pre - mean) * torch.exp(-log_std)
        logp = (-0.5 * z * z - log_std - 0.5 * math.log(2 * math.pi)).sum(-1) - torch.log(1.0 - act * act + 1e-6).sum(-1)
        return act, logp

    def forward(self, obs, deterministic=False):
        mean, log_std = self.mean_log_std(obs)
        return self.squash(mean, log_std, None if deterministic else torch.randn_like(mean))[0]
