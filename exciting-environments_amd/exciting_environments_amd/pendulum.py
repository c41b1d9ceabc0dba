"""`exciting_environments.pendulum` of the reference (its `__init__.py`): the same import path here."""
from .envs import Pendulum

__all__ = ["Pendulum"]
