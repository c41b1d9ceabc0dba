"""`exciting_environments.cart_pole` of the reference (its `__init__.py`): the same import path here."""
from .envs import CartPole

__all__ = ["CartPole"]
