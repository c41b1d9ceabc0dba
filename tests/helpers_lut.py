"""Synthetic PMSM look-up tables for the saturated-model tests (the reference's motor data files are not shipped)."""
import numpy as np


def linear_lut(l_d, l_q, psi_p, i_d_range=(-250.0, 0.0), i_q_range=(-250.0, 250.0), n_d=26, n_q=51):
    """Tables that encode the linear dq model exactly: L_dd = l_d, L_qq = l_q, Psi_d = l_d*i_d + psi_p, Psi_q = l_q*i_q."""
    i_d = np.linspace(*i_d_range, n_d)[None]
    i_q = np.linspace(*i_q_range, n_q)[None]
    ID, IQ = np.meshgrid(i_d[0], i_q[0])  # (n_q, n_d): rows follow i_q like the reference's tables
    return dict(i_d_vec=i_d, i_q_vec=i_q, L_dd=np.full_like(ID, l_d), L_dq=np.zeros_like(ID), L_qd=np.zeros_like(ID),
                L_qq=np.full_like(ID, l_q), Psi_d=l_d * ID + psi_p, Psi_q=l_q * IQ)


def saturating_lut(seed=0, n_d=26, n_q=51, holes=True):
    """A smooth saturating machine (inductances fall with current, small cross-coupling) with NaN corners like the SEW file."""
    lut = linear_lut(0.37e-3, 1.2e-3, 65.6e-3, n_d=n_d, n_q=n_q)
    ID, IQ = np.meshgrid(lut["i_d_vec"][0], lut["i_q_vec"][0])
    sat = 1.0 / (1.0 + (ID / 300.0) ** 2 + (IQ / 280.0) ** 2)
    lut["L_dd"] = 0.37e-3 * (0.6 + 0.4 * sat)
    lut["L_qq"] = 1.2e-3 * (0.5 + 0.5 * sat)
    lut["L_dq"] = 2e-5 * np.tanh(ID / 100.0) * np.tanh(IQ / 100.0)
    lut["L_qd"] = lut["L_dq"].copy()
    lut["Psi_d"] = 65.6e-3 + 0.37e-3 * 300.0 * np.arctan(ID / 300.0)
    lut["Psi_q"] = 1.2e-3 * 280.0 * np.arctan(IQ / 280.0)
    if holes:
        rng = np.random.default_rng(seed)
        for q in ("L_dd", "L_dq", "L_qd", "L_qq", "Psi_d", "Psi_q"):
            m = np.array(lut[q])
            m[:3, :2] = np.nan
            m[-2:, -4:] = np.nan
            m[rng.integers(5, n_q - 5), rng.integers(3, n_d - 3)] = np.nan
            lut[q] = m
    return lut


def sew_shaped_lut():
    """A synthetic machine on the grid of the reference's SEW file (LUT_SEW_jax_grad.mat: i_d_vec 18 points in [-16, 1] A,
    i_q_vec 33 points in [-16, 16] A, tables (33, 18)) with its NaN pattern: everything outside the current-limit circle is
    NaN — a larger hole in the differential inductances than in the flux linkages (the file has 188 vs 142 NaNs). The values
    are made up (the motor data is not redistributed); the SHAPES, the negative-to-positive i_d range and the hole geometry
    are what prepare_pmsm_lut / the kernels have to cope with."""
    i_d = np.linspace(-16.0, 1.0, 18)[None]
    i_q = np.linspace(-16.0, 16.0, 33)[None]
    ID, IQ = np.meshgrid(i_d[0], i_q[0])  # (33, 18)
    sat = 1.0 / (1.0 + (ID / 20.0) ** 2 + (IQ / 18.0) ** 2)
    lut = dict(i_d_vec=i_d, i_q_vec=i_q,
               L_dd=1.44e-3 * (1.0 + 1.2 * sat), L_qq=1.44e-3 * (0.7 + 1.6 * sat),
               L_dq=2e-4 * np.tanh(ID / 8.0) * np.tanh(IQ / 8.0), L_qd=3e-4 * np.tanh(ID / 8.0) * np.tanh(IQ / 8.0),
               Psi_d=0.09 + 1.44e-3 * 20.0 * np.arctan(ID / 20.0), Psi_q=1.44e-3 * 18.0 * np.arctan(IQ / 18.0))
    r = np.sqrt(ID ** 2 + IQ ** 2)
    for q, lim in (("Psi_d", 16.3), ("Psi_q", 16.3), ("L_dd", 15.2), ("L_dq", 15.2), ("L_qd", 15.2), ("L_qq", 15.2)):
        m = np.array(lut[q])
        m[r > lim] = np.nan
        lut[q] = m
    return lut
