"""consolver_amd -- MI355X-native ConsistencySolver sampling engine (hot path only).

Public surface mirrors the reference's plugin API for this path:
``PPOScheduler`` (scheduler_ppo.py), ``FMPPOScheduler`` (edit_ppo/scheduler_fmppo.py),
``FactorNetPPO`` (factor_net_ppo.py / edit_ppo/factor_net_ppo.py) and the denoiser behind
``unet(latents, t, encoder_hidden_states=..., return_dict=False)[0]``.  Everything computes in
hand-written HIP kernels behind ``include/consolver_hip.h``; there is no CPU fallback.
"""
from .scheduling_ppo import PPOScheduler, SolverOutput  # noqa: F401
from ._scheduler_base import HAVE_DIFFUSERS, SolverConfig  # noqa: F401
from .scheduling_fmppo import FMPPOScheduler  # noqa: F401
from .factor_net import FactorNetPPO, FluxFactorNetPPO  # noqa: F401

__version__ = "0.1.0"
