"""Solver tokens accepted by the `solver=` keyword (stand-ins for diffrax.Euler() / diffrax.Tsit5();
diffrax objects cannot exist here). Fixed step only, like every call site in the reference
(e.g. pendulum_env.py:184, 226-235)."""


class _Solver:
    id = -1
    name = "?"
    # diffrax keeps a solver state only for FSAL ("first same as last") Runge-Kutta methods: (first_step, f0) with f0 shaped like
    # the ODE state; Euler's and the RK4 extension's is None. The environments mirror that in Additions.solver_state.
    fsal = False

    def __repr__(self):
        return f"{type(self).__name__}()"

    def __eq__(self, other):
        return isinstance(other, _Solver) and other.id == self.id

    def __hash__(self):
        return hash(self.id)


class Euler(_Solver):
    """y1 = y0 + f(y0, u) * dt (diffrax.Euler.step)."""

    id, name = 0, "euler"


class RK4(_Solver):
    """Classic 4-stage Runge-Kutta (build-side extension)."""

    id, name = 1, "rk4"


class Tsit5(_Solver):
    """Tsitouras 5(4), 5th-order solution at fixed step (error estimate unused, as under ConstantStepSize)."""

    id, name = 2, "tsit5"
    fsal = True
