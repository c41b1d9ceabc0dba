"""`exciting_environments.acrobot` of the reference (its `__init__.py`): the same import path here."""
from .envs import Acrobot

__all__ = ["Acrobot"]
