"""RL-driven collocation sampling — same import path as `pinnrl.rl`."""

from .rl_agent import DQNNetwork, RLAgent  # noqa: F401
