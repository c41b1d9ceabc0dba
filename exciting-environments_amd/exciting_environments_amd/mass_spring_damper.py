"""`exciting_environments.mass_spring_damper` of the reference (its `__init__.py`): the same import path here."""
from .envs import MassSpringDamper

__all__ = ["MassSpringDamper"]
