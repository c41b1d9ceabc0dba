"""`exciting_environments.fluid_tank` of the reference (its `__init__.py`): the same import path here."""
from .envs import FluidTank

__all__ = ["FluidTank"]
