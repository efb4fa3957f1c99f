"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's continuum opacity models: numpy
for the element-wise formulas, oracle/pb_oracle.c for the pieces that are C in the
reference (_alkali.c, _spline.c).  Pinned by tests/golden/g7_continuum.npz (outputs of the
real reference classes)."""
import ctypes as C

import numpy as np

from . import oracle as _o

c_dp = C.POINTER(C.c_double)

# pyratbay/constants/astrophysical_constants.py:67-131 (scipy.constants, CODATA 2018)
H = 6.62607015e-27
K = 1.380649e-16
LS = 29979245800.0
BAR = 1e6
AMAGAT = 2.6867801117984436e+19


def _p(a):
    return a.ctypes.data_as(c_dp)


# ---- opacity/rayleigh/rayleigh.py:37-83 ----
def rayleigh_cross_section(wn, species):
    if species in ('H', 'H2'):
        c = {'H': (5.799e-45, 1.422e-54, 2.784e-64),
             'H2': (8.140e-45, 1.280e-54, 1.610e-64)}[species]
        return c[0] * wn**4.0 + c[1] * wn**6.0 + c[2] * wn**8.0
    if species == 'He':
        c = (5.484e-46, 2.440e-11, 5.940e-42, 2.900e-11)
        return c[0] * wn**4 * (1.0 + c[1] * wn**2 + c[2] * wn**4 / (1 - c[3] * wn**2))**2.0
    if species == 'e-':
        return np.tile(6.653e-25, len(wn))
    raise ValueError(species)


# ---- opacity/clouds/lecavelier.py:64-100 ----
def lecavelier_cross_section(wn, pars, s0=5.31e-27, l0=3.5e-5):
    return 10.0**pars[0] * s0 * (wn * l0)**(-pars[1])


def nominal_density(pressure_bar, temperature):
    return pressure_bar * BAR / temperature / K


# ---- opacity/clouds/gray.py:44-75 ----
def gray_layer_cross_section(pressure_bar, pars, s0=5.31e-27):
    p_top, p_bottom = 10**pars[2], 10**pars[1]
    mask = (pressure_bar >= p_bottom) & (pressure_bar <= p_top)
    cs = np.zeros(len(pressure_bar))
    cs[mask] = 10**pars[0] * s0
    return cs


# ---- opacity/cia.py:119-215 with src_c/_spline.c ----
def second_deriv(yin, xin):
    yin, xin = np.ascontiguousarray(yin, float), np.ascontiguousarray(xin, float)
    out = np.empty(len(yin))
    _o.lib().orc_second_deriv(_p(out), _p(yin), _p(xin), C.c_int(len(yin)))
    return out


def splinterp_1D(yin, xin, y2nd, xout, extrap):
    yin, xin, y2nd, xout = (np.ascontiguousarray(a, float) for a in (yin, xin, y2nd, xout))
    out = np.empty(len(xout))
    _o.lib().orc_splinterp_1D(_p(out), _p(yin), _p(xin), _p(y2nd), C.c_int(len(xin)),
                              _p(xout), C.c_int(len(xout)), C.c_double(extrap))
    return out


def cia_cross_section(tab, temps, temperature, lo, hi):
    """calc_cross_section: linear interpolation in temperature of the tabulated values."""
    tab, temps, temperature = (np.ascontiguousarray(a, float) for a in (tab, temps, temperature))
    dcs_dt = np.ascontiguousarray(np.diff(tab, axis=0) / np.expand_dims(np.ediff1d(temps), axis=1))
    out = np.zeros((len(temperature), tab.shape[1]))
    rc = _o.lib().orc_lin_interp_2D(_p(out), _p(tab), _p(temps), _p(dcs_dt),
                                    C.c_int(len(temps)), C.c_int(tab.shape[1]),
                                    _p(temperature), C.c_int(len(temperature)), C.c_int(lo),
                                    C.c_int(hi))
    if rc:
        raise ValueError('temperature outside the CIA table')
    return out


# ---- opacity/hydrogen_ion.py:33-276 (John 1988) ----
WN0_BF = 6090.5


def hminus_sigma_bf(wn):
    c_bf = [152.519, 49.534, -118.858, 92.536, -34.194, 4.982]
    mask = wn > WN0_BF
    reduced_wl = 1e-2 * np.sqrt(wn[mask] - WN0_BF)
    f_lambda = np.zeros(np.sum(mask))
    for n in range(6):
        f_lambda += c_bf[n] * reduced_wl**n
    sigma = np.zeros(len(wn))
    sigma[mask] = 1.0e-6 * (reduced_wl / wn[mask])**3.0 * f_lambda
    return sigma


def hminus_ff_factors(wn):
    """Rows F_i[w] multiplying beta_i = sqrt(5040/T)**(i+2), i = 0..5 (zero where the
    reference's short/long-wavelength branch does not use that power)."""
    wl = 1e4 / wn
    short = [[518.1021, 473.2636, -482.2089, 115.5291], [-734.8666, 1443.4137, -737.1616, 169.6374],
             [1021.1775, -1977.3395, 1096.8827, -245.649], [-479.0721, 922.3575, -521.1341, 114.243],
             [93.1373, -178.9275, 101.7963, -21.9972], [-6.4285, 12.3600, -7.0571, 1.5097]]
    long_ = [[2483.346, -3449.889, 2200.040, -696.271, 88.283],
             [285.827, -1158.382, 2427.719, -1841.400, 444.517],
             [-2054.291, 8746.523, -13651.105, 8624.970, -1863.864],
             [2827.776, -11485.632, 16755.524, -10051.530, 2095.288],
             [-1341.537, 5303.609, -7510.494, 4400.067, -901.788],
             [208.952, -812.939, 1132.738, -655.020, 132.985]]
    wl_crit = 0.3645
    sw = (0.182 < wl) & (wl < wl_crit)
    lw = wl >= wl_crit
    F = np.zeros((6, len(wn)))

    def poly(co, i, x):
        return (co[0][i] * x**2.0 + co[1][i] + co[2][i] / x + co[3][i] / x**2.0
                + co[4][i] / x**3.0 + co[5][i] / x**4.0)
    for i in range(4):
        F[i, sw] = 1.0e-29 * poly(short, i, wl[sw])
    for i in range(5):
        F[i + 1, lw] = 1.0e-29 * poly(long_, i, wl[lw])
    return F, sw, lw


def hminus_cross_sections(wn, temperature):
    """(bound-free, free-free) [L,W] in cm5 / H / electron."""
    temperature = np.asarray(temperature, float)
    alpha = H * LS / K
    t2 = np.expand_dims(temperature, axis=1)
    bf = (0.75 * t2**-1.5 * K * np.exp(WN0_BF * alpha / t2) * (1.0 - np.exp(-wn * alpha / t2))
          * hminus_sigma_bf(wn))
    tc = np.clip(temperature, 1000.0, 10080.0)
    beta = np.array([np.sqrt(5040.0 / tc)**(i + 2) for i in range(6)])     # [6, L]
    F, sw, lw = hminus_ff_factors(wn)
    sigma = np.zeros((len(wn), len(tc)))
    sigma[sw] = np.sum(beta[0:4] * np.expand_dims(F[0:4, sw].T, axis=2), axis=1)
    sigma[lw] = np.sum(beta[1:6] * np.expand_dims(F[1:6, lw].T, axis=2), axis=1)
    sigma *= K * tc
    return bf, sigma.T


# ---- opacity/alkali/alkali.py:122-160 -> src_c/_alkali.c:30-106 ----
def alkali_cross_section(pressure_barye, wn, temp, voigt_det, detuning, mass, lpar, Z, cutoff,
                         wn0, gf):
    pressure_barye, wn, temp, voigt_det, wn0, gf = (
        np.ascontiguousarray(a, float) for a in (pressure_barye, wn, temp, voigt_det, wn0, gf))
    ec = np.zeros((len(temp), len(wn)))
    _o.lib().orc_alkali_cross_section(
        _p(ec), _p(pressure_barye), _p(wn), _p(temp), _p(voigt_det), C.c_double(detuning),
        C.c_double(mass), C.c_double(lpar), C.c_double(Z), C.c_double(cutoff), _p(wn0), _p(gf),
        C.c_int(len(wn0)), C.c_int(len(temp)), C.c_int(len(wn)))
    return ec
