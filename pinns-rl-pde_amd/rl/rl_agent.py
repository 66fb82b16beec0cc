"""The part of `pinnrl/rl/rl_agent.py` that the collocation path consumes: the DQN policy network and the
epsilon-greedy scorer that `generate_collocation_points(strategy="adaptive")` calls once per batch
(pinnrl/pdes/pde_base.py:961-1073).

Mirrored: `DQNNetwork` (rl_agent.py:15-88: parameter layout, `state_dict` keys and theta_0 under a seed),
`RLAgent.__init__` (rl_agent.py:139-212: fields, policy + target networks, RNG consumption),
`RLAgent.select_action` (rl_agent.py:214-229) and `RLAgent.update_epsilon` (rl_agent.py:557-566).
Behaviour kept as upstream: the policy network is never put in eval mode, so its Dropout(0.1)
is active while it scores the grid; the explore branch returns a (1, 1) tensor, which makes the sampler's
multinomial collapse every point onto grid cell 0.

Not mirrored (out of scope: nothing in pinnrl ever trains the agent): replay buffer, optimiser step, plotting.

The scorer is a 2 -> H -> H -> 1 MLP over <= 10^4 grid points once per batch: plain torch ops on the
agent's device (a few microseconds of work; not a hot-path kernel).
"""

from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn


class DQNNetwork(nn.Module):  # rl_agent.py:15-88
    def __init__(self, state_dim: int, action_dim: int, hidden_dim: int, num_layers: int = 3, dropout: float = 0.1):
        super().__init__()
        layers = [nn.Sequential(nn.Linear(state_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.ReLU(), nn.Dropout(dropout))]
        for _ in range(num_layers - 2):
            layers.append(nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.ReLU(),
                                        nn.Dropout(dropout)))
        layers.append(nn.Linear(hidden_dim, action_dim))
        self.layers = nn.Sequential(*layers)
        self._init_weights()

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight, gain=1.0)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.layers(x)


class RLAgent:  # rl_agent.py:139-229, 557-566
    def __init__(self, state_dim: int, action_dim: int, hidden_dim: int, learning_rate: float = 0.0001,
                 gamma: float = 0.99, epsilon_start: float = 1.0, epsilon_end: float = 0.01,
                 epsilon_decay: float = 0.995, memory_size: int = 10000, batch_size: int = 64,
                 target_update: int = 100, reward_weights: Optional[Dict[str, float]] = None,
                 device: Optional[torch.device] = None):
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.state_dim, self.action_dim, self.hidden_dim = state_dim, action_dim, hidden_dim
        self.learning_rate, self.gamma = learning_rate, gamma
        self.epsilon, self.epsilon_end, self.epsilon_decay = epsilon_start, epsilon_end, epsilon_decay
        self.memory_size, self.batch_size, self.target_update = memory_size, batch_size, target_update
        self.reward_weights = reward_weights or {"residual": 1.0, "boundary": 1.0, "initial": 1.0, "exploration": 0.1}
        # theta_0 comes from the CPU generator (then moved), so a seed gives the reference's CPU-built weights
        self.policy_net = DQNNetwork(state_dim, action_dim, hidden_dim).to(self.device)
        self.target_net = DQNNetwork(state_dim, action_dim, hidden_dim).to(self.device)
        self.target_net.load_state_dict(self.policy_net.state_dict())
        self.steps = 0
        self.episode_rewards = []
        self.episode_reward = 0

    def select_action(self, state: torch.Tensor) -> torch.Tensor:
        if torch.rand(1).to(self.device).item() > self.epsilon:
            with torch.no_grad():
                return self.policy_net(state.to(self.device)).view(1, -1)
        return torch.rand(1, 1, device=self.device)

    def action_probabilities(self, state: torch.Tensor) -> torch.Tensor:
        """Device-side form of `select_action` + the sampler's `abs / sum` (pde_base.py:1021-1030) with FIXED shapes and no
        host round trip, so that the adaptive sampler can sit inside a captured step: a (G*G,) probability vector that is
        |Q| / sum |Q| on the exploit branch and all mass on grid cell 0 on the explore branch — exactly what the
        reference's (1, 1) explore action does to its one-category multinomial.  The branch is chosen on the
        device: `rand > epsilon` with epsilon read from `self.epsilon` (a float, or a device scalar once
        `epsilon_tensor()` has been called).  The policy network keeps its mode (train: dropout active, as in the
        reference, which never calls `.eval()`)."""
        state = state.to(self.device)
        with torch.no_grad():
            q = self.policy_net(state).reshape(-1).abs()
            probs = q / q.sum()
            eps = self._eps_dev if getattr(self, "_eps_dev", None) is not None else torch.full((), float(self.epsilon), device=self.device)
            exploit = torch.rand((), device=self.device) > eps
            onehot = (torch.arange(probs.numel(), device=self.device) == 0).to(probs.dtype)  # no host scalar: capturable
            return torch.where(exploit, probs, onehot)

    def epsilon_tensor(self) -> torch.Tensor:
        """Move the exploration rate to a device scalar (decayed on the device by `update_epsilon_device`)."""
        if getattr(self, "_eps_dev", None) is None:
            self._eps_dev = torch.full((), float(self.epsilon), device=self.device)
        return self._eps_dev

    def update_epsilon_device(self) -> None:  # update_epsilon without a host value (rl_agent.py:557-566)
        e = self.epsilon_tensor()
        e.copy_(torch.clamp(e * self.epsilon_decay, min=self.epsilon_end))

    def update_epsilon(self, epoch: int = None):
        self.epsilon = max(self.epsilon_end, self.epsilon * self.epsilon_decay)
        return self.epsilon
