"""Environment factory with the reference's public surface: ``EnvironmentRegistry.<NAME>.make(**kwargs)``
(reference exciting_environments/registration.py:13-32). Member names and string ids are the reference's; the
lookup is a plain table keyed by member name."""
import enum

from . import envs as _envs

_ENV_CLASS_BY_MEMBER = {
    "PENDULUM": _envs.Pendulum,
    "MASS_SPRING_DAMPER": _envs.MassSpringDamper,
    "CART_POLE": _envs.CartPole,
    "ACROBOT": _envs.Acrobot,
    "FLUID_TANK": _envs.FluidTank,
    "PMSM": _envs.PMSM,
}


class EnvironmentRegistry(enum.Enum):
    """The six ODE environments. Iteration order follows the reference (tests iterate ``list(EnvironmentRegistry)``)."""

    CART_POLE = "CartPole-v0"
    MASS_SPRING_DAMPER = "MassSpringDamper-v0"
    PENDULUM = "Pendulum-v0"
    FLUID_TANK = "FluidTank-v0"
    PMSM = "PMSM-v0"
    ACROBOT = "Acrobot-v0"

    @property
    def env_class(self):
        try:
            return _ENV_CLASS_BY_MEMBER[self.name]
        except KeyError:  # pragma: no cover - every member has an entry
            raise ValueError(f"Unknown environment: {self}") from None

    def make(self, **env_kwargs):
        """Instantiate the environment; keyword arguments are forwarded to its constructor."""
        return self.env_class(**env_kwargs)
