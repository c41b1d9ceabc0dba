"""`exciting_environments.pmsm` of the reference (its `__init__.py`): the same import path here."""
from .envs import MotorVariant, PMSM

__all__ = ["MotorVariant", "PMSM"]
