"""MI355X-native batched ODE stepping behind exciting-environments' vmap_step / vmap_sim_ahead API.

Usage mirrors the reference package (README.md:15-33):

    import exciting_environments_amd as excenvs
    from exciting_environments_amd import EnvironmentRegistry
    env = EnvironmentRegistry.PENDULUM.make(batch_size=5, tau=2e-2)
    obs, state = env.vmap_reset()
    obs, state = env.vmap_step(state, actions)            # one fused HIP launch
    observations, states, last_state = env.vmap_sim_ahead(state, actions, env.tau, env.tau)  # one persistent launch
"""
from .core_env import CoreEnvironment
from .envs import Acrobot, CartPole, FluidTank, MassSpringDamper, MotorVariant, Pendulum, PMSM, load_pmsm_lut, prepare_pmsm_lut
from .registration import EnvironmentRegistry
from .gym_wrapper import GymWrapper
from .solvers import Euler, RK4, Tsit5
from .stepper import Stepper
from .utils import MinMaxNormalization, dump_sim_properties_to_json, load_sim_properties_from_json
from . import random, tree, utils

__all__ = [
    "CoreEnvironment", "Acrobot", "CartPole", "FluidTank", "MassSpringDamper", "Pendulum", "PMSM", "MotorVariant", "prepare_pmsm_lut", "load_pmsm_lut",
    "EnvironmentRegistry", "GymWrapper", "Stepper", "Euler", "RK4", "Tsit5", "MinMaxNormalization", "dump_sim_properties_to_json",
    "load_sim_properties_from_json", "random", "tree", "utils",
]
