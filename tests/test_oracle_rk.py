"""RK4 / Tsit5 in the oracle are "parity unpinned" (the reference has no test or fixture for any solver
other than Euler). They are pinned here by (i) the Butcher order conditions of the tableau
and (ii) the observed convergence order against the closed-form mass-spring-damper solution."""
import itertools

import numpy as np
import pytest
from scipy.linalg import expm

import oracle

TSIT5_C = np.array([0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0])
TSIT5_A = np.zeros((6, 6))
TSIT5_A[1, :1] = [0.161]
TSIT5_A[2, :2] = [-0.008480655492356989, 0.335480655492357]
TSIT5_A[3, :3] = [2.8971530571054935, -6.359448489975075, 4.3622954328695815]
TSIT5_A[4, :4] = [5.325864828439257, -11.74888356406283, 7.4955393428898365, -0.09249506636175525]
TSIT5_A[5, :5] = [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383]
TSIT5_B = np.array([0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
                    2.324710524099774])


def _order_conditions(A, b, c, order):
    """Rooted-tree order conditions up to `order` (1,1,2,4,9 trees for orders 1..5)."""
    e = np.ones_like(c)
    Ac, Ac2, Ac3, AAc = A @ c, A @ c**2, A @ c**3, A @ (A @ c)
    conds = [(b @ e, 1.0)]
    if order >= 2:
        conds += [(b @ c, 1 / 2)]
    if order >= 3:
        conds += [(b @ c**2, 1 / 3), (b @ Ac, 1 / 6)]
    if order >= 4:
        conds += [(b @ c**3, 1 / 4), (b @ (c * Ac), 1 / 8), (b @ Ac2, 1 / 12), (b @ AAc, 1 / 24)]
    if order >= 5:
        conds += [(b @ c**4, 1 / 5), (b @ (c**2 * Ac), 1 / 10), (b @ (c * Ac2), 1 / 15), (b @ (Ac * Ac), 1 / 20),
                  (b @ (c * AAc), 1 / 30), (b @ Ac3, 1 / 20), (b @ (A @ (c * Ac)), 1 / 40),
                  (b @ (A @ Ac2), 1 / 60), (b @ (A @ AAc), 1 / 120)]
    return conds


def test_tsit5_tableau_order_conditions():
    assert np.allclose(TSIT5_A.sum(1), TSIT5_C, atol=1e-15)
    conds = _order_conditions(TSIT5_A, TSIT5_B, TSIT5_C, 5)
    assert len(conds) == 17
    for got, want in conds:
        assert abs(got - want) < 5e-15


def test_rk4_tableau_order_conditions():
    A = np.zeros((4, 4)); A[1, 0] = 0.5; A[2, 1] = 0.5; A[3, 2] = 1.0
    b = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6]); c = np.array([0, 0.5, 0.5, 1.0])
    for got, want in _order_conditions(A, b, c, 4):
        assert abs(got - want) < 1e-15


def _msd_exact(x0, v0, u, d, k, m, T):
    M = np.array([[0, 1, 0], [-k / m, -d / m, 1 / m], [0, 0, 0]])
    return (expm(M * T) @ np.array([x0, v0, u]))[:2]


@pytest.mark.parametrize("solver,order", [("euler", 1), ("rk4", 4), ("tsit5", 5)])
def test_convergence_order_on_mass_spring_damper(solver, order):
    params = {"d": 1.0, "k": 100.0, "m": 1.0}
    pn = {"deflection": (-10, 10), "velocity": (-10, 10)}
    an = {"force": (-20, 20)}
    props, keep = oracle.make_props("mass_spring_damper", params, pn, an, np.float64, 1)
    T, a = 0.5, 0.35
    exact = _msd_exact(1.0, -2.0, oracle.denormalize(a, -20, 20), 1.0, 100.0, 1.0, T)
    errs = []
    ns = [50, 100, 200] if order > 1 else [2000, 4000, 8000]
    for n in ns:
        st = [np.array([1.0]), np.array([-2.0])]
        _, _, last = oracle.sim_ahead("mass_spring_damper", solver, st, np.full((1, n, 1), a), props, T / n)
        errs.append(np.hypot(last[0][0] - exact[0], last[1][0] - exact[1]))
    rates = [np.log2(errs[i] / errs[i + 1]) for i in range(2)]
    for r in rates:
        assert order - 0.25 < r < order + 0.4, (solver, errs, rates)


def test_c1_stage_sees_next_action_only_in_ahead_semantics():
    """core_env.py:435-439 warning: with higher-order solvers sim_ahead's c=1 stages read the next action.
    With piecewise-constant actions that differ, SEM_AHEAD != SEM_STEP for rk4/tsit5 but == for euler;
    with a constant action sequence both agree bit for bit."""
    params = {"d": 1.0, "k": 100.0, "m": 1.0}
    props, keep = oracle.make_props("mass_spring_damper", params, {"deflection": (-10, 10), "velocity": (-10, 10)},
                                    {"force": (-20, 20)}, np.float64, 2)
    st = [np.array([0.3, -0.2]), np.array([0.0, 1.0])]
    rng = np.random.default_rng(1)
    varying = rng.uniform(-1, 1, (2, 40, 1))
    const = np.full((2, 40, 1), 0.25)
    for solver in ("euler", "rk4", "tsit5"):
        o_s, _, _ = oracle.sim_ahead("mass_spring_damper", solver, st, varying, props, 1e-2, semantics=oracle.SEM_STEP)
        o_a, _, _ = oracle.sim_ahead("mass_spring_damper", solver, st, varying, props, 1e-2, semantics=oracle.SEM_AHEAD)
        if solver == "euler":
            assert np.array_equal(o_s, o_a)
        else:
            assert not np.array_equal(o_s, o_a)
        c_s, _, _ = oracle.sim_ahead("mass_spring_damper", solver, st, const, props, 1e-2, semantics=oracle.SEM_STEP)
        c_a, _, _ = oracle.sim_ahead("mass_spring_damper", solver, st, const, props, 1e-2, semantics=oracle.SEM_AHEAD)
        assert np.array_equal(c_s, c_a)


def test_rk4_ahead_first_step_by_hand():
    """One SEM_AHEAD RK4 step of the MSD system computed by hand: stages 1-3 see action 0, stage 4 (c=1) action 1."""
    d, k, m, dt = 1.0, 100.0, 1.0, 1e-2
    props, keep = oracle.make_props("mass_spring_damper", {"d": d, "k": k, "m": m},
                                    {"deflection": (-10, 10), "velocity": (-10, 10)}, {"force": (-20, 20)},
                                    np.float64, 1)
    a = np.array([[[0.5], [-0.75]]])
    u0, u1 = oracle.denormalize(0.5, -20, 20), oracle.denormalize(-0.75, -20, 20)
    f = lambda y, u: np.array([y[1], (u - d * y[1] - k * y[0]) / m])
    y0 = np.array([0.3, -0.2])
    k1 = f(y0, u0) * dt
    k2 = f(y0 + 0.5 * k1, u0) * dt
    k3 = f(y0 + 0.5 * k2, u0) * dt
    k4s, k4a = f(y0 + k3, u0) * dt, f(y0 + k3, u1) * dt
    for sem, k4 in ((oracle.SEM_STEP, k4s), (oracle.SEM_AHEAD, k4a)):
        want = y0 + (k1 / 6 + k2 / 3 + k3 / 3 + k4 / 6)
        _, straj, _ = oracle.sim_ahead("mass_spring_damper", "rk4", [y0[:1].copy(), y0[1:].copy()], a, props, dt,
                                       semantics=sem)
        got = np.array([straj[0][0, 1], straj[1][0, 1]])
        assert np.allclose(got, want, rtol=1e-14, atol=0)


@pytest.mark.parametrize("env,solver,tol", [("pendulum", "tsit5", 2e-10), ("pendulum", "rk4", 5e-8), ("acrobot", "tsit5", 5e-9),
                                            ("cartpole", "tsit5", 5e-9)])
def test_nonlinear_systems_against_independent_scipy_integration(env, solver, tol):
    """Constant action, 200 fixed steps: the oracle's RK trajectory vs scipy's DOP853 (rtol 1e-13) on the same vector
    field written independently here from the reference formulas (pendulum_env.py:144-150, acrobot_env.py:171-197,
    cart_pole_env.py:159-180)."""
    from scipy.integrate import solve_ivp
    from conftest import load_golden

    g = load_golden(env)
    p = g["params"]
    a_norm = 0.3
    (lo, hi), = g["act_norm"].values()
    u = oracle.denormalize(a_norm, lo, hi)
    if env == "pendulum":
        y0 = np.array([0.4, -0.3])
        f = lambda t, y: [y[1], (u + p["l"] * p["m"] * p["g"] * np.sin(y[0])) / (p["m"] * p["l"] ** 2)]
    elif env == "acrobot":
        y0 = np.array([0.5, -0.4, 0.2, 0.1])

        def f(t, y):
            th1, th2, w1, w2 = y
            m1, m2, l1, lc1, lc2, I1, I2, gg = p["m_1"], p["m_2"], p["l_1"], p["l_c1"], p["l_c2"], p["I_1"], p["I_2"], p["g"]
            d11 = m1 * lc1**2 + m2 * (l1**2 + lc2**2 + 2 * l1 * lc2 * np.cos(th2)) + I1 + I2
            d12 = m2 * (lc2**2 + l1 * lc2 * np.cos(th2)) + I2
            d22 = m2 * lc2**2 + I2
            h1 = -m2 * l1 * lc2 * np.sin(th2) * w2**2 - 2 * m2 * l1 * lc2 * np.sin(th2) * w1 * w2
            h2 = m2 * l1 * lc2 * np.sin(th2) * w1**2
            phi1 = (m1 * lc1 + m2 * l1) * gg * np.cos(th1 + np.pi / 2) + m2 * lc2 * gg * np.cos(th1 + th2 + np.pi / 2)
            phi2 = m2 * lc2 * gg * np.cos(th1 + th2 + np.pi / 2)
            dw1 = 1 / (d12 - d22 / d12 * d11) * (u + d22 / d12 * (h1 + phi1) - h2 - phi2)
            dw2 = (-d11 * dw1 - h1 - phi1) / d12
            return [w1, w2, dw1, dw2]
    else:
        y0 = np.array([0.1, 0.5, 0.3, -0.2])

        def f(t, y):
            x, v, th, w = y
            mp, mc, l, mup, muc, gg = p["m_p"], p["m_c"], p["l"], p["mu_p"], p["mu_c"], p["g"]
            dw = (gg * np.sin(th) + np.cos(th) * ((-u - mp * l * w**2 * np.sin(th) + muc * np.sign(v)) / (mc + mp))
                  - (mup * w) / (mp * l)) / (l * (4 / 3 - (mp * np.cos(th) ** 2) / (mc + mp)))
            dv = (u + mp * l * (w**2 * np.sin(th) - dw * np.cos(th)) - muc * np.sign(v)) / (mc + mp)
            return [v, dv, w, dw]
    n, tau = 200, 1e-3
    ref = solve_ivp(f, (0, n * tau), y0, method="DOP853", rtol=1e-13, atol=1e-15).y[:, -1]
    pn = {k: (-1e3, 1e3) for k in g["phys_norm"]}  # wide box: no observation saturation matters here
    props, keep = oracle.make_props(env, p, pn, g["act_norm"], np.float64, 1)
    st = [np.array([v]) for v in y0]
    _, straj, last = oracle.sim_ahead(env, solver, st, np.full((1, n, 1), a_norm), props, tau, semantics=oracle.SEM_AHEAD)
    got = np.array([s[0, -1] for s in straj])
    # SEM_AHEAD saves wrapped angles; the short horizon keeps them inside (-pi, pi)
    assert np.abs(got - ref).max() < tol, (got, ref)
