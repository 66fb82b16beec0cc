"""CPU oracle for the collocation-point hot path — TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU restatement of the reference algorithm
(`pinnrl` 0.3.1: network forward + autograd-of-autograd PDE residual + loss
gradient).  It exists so that the hand-written HIP path can be checked against
the reference's numerics on machines where `/root/reference` is absent (the GPU
box).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it; nothing under `pinns-rl-pde_amd/` does, and the product
path raises when its HIP library is missing instead of falling back to this.

Parity status: PINNED.  `oracle/make_golden.py` (run in the build container,
where the reference is importable) asserts that every function here reproduces
the imported reference on seeded inputs, then writes the vectors to
`tests/golden/` — see that script and `tests/test_oracle_golden.py`.
"""

from .reference_path import (  # noqa: F401
    AgentState,
    ArchSpec,
    PdeSpec,
    apply_loss_fn,
    compute_derivatives,
    compute_loss_terms,
    compute_loss_terms_heat,
    compute_residual,
    composite_layer_norm,
    dqn_forward,
    dqn_init_state_dict,
    init_state_dict,
    make_agent,
    network_forward,
    rar_probabilities,
    residual_loss_and_grad,
    sample_adaptive,
    sample_stratified,
    sample_uniform,
    select_action,
)
