"""TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/liboracle.so (the plain-C restatement of the reference
hot path, oracle/pb_oracle.c) plus NumPy restatements of the small host pre-computes
that feed it.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product package pyratbay_amd never does.

Function names and argument order follow the reference extension modules
(pyratbay.lib.*) so the parity tests read like the reference's.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)


class ExtStats(C.Structure):
    _fields_ = [('ofactor', C.c_int32), ('nadd', C.c_int32),
                ('nskip', C.c_int32), ('neval', C.c_int32)]


def build(force=False):
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'pb_oracle.c')
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(['make', '-C', _HERE, 'liboracle.so'],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_trapezoid.restype = C.c_double
        _LIB.orc_simps.restype = C.c_double
        _LIB.orc_voigt_point.restype = C.c_double
        _LIB.orc_voigt_point.argtypes = [C.c_double] * 3
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_ip)


# --------------------------------------------------------------------------
# vprofile.grid(profile, psize, index, lorentz, doppler, dwn, verb)
# --------------------------------------------------------------------------
def voigt_grid(profile, psize, index, lorentz, doppler, dwn, verb=0):
    """In-place like the reference (vprofile.c:42-114): fills profile, resets the
    zero entries of psize and writes index."""
    ps, psp = _i(psize)
    ix, ixp = _i(index)
    lor, lorp = _d(lorentz)
    dop, dopp = _d(doppler)
    assert profile.dtype == np.float64 and profile.flags.c_contiguous
    status = lib().orc_voigt_grid(
        profile.ctypes.data_as(c_dp), C.c_int64(profile.size), psp, ixp,
        lorp, C.c_int(len(lor)), dopp, C.c_int(len(dop)), C.c_double(dwn))
    if status != 1:
        raise RuntimeError(f'orc_voigt_grid status {status}')
    psize[...] = ps.reshape(np.shape(psize))
    index[...] = ix.reshape(np.shape(index))
    return 1


# --------------------------------------------------------------------------
# _extcoeff.extinction(...)
# --------------------------------------------------------------------------
def extinction(ext, profile, psize, pindex, lorentz, doppler, wn, own, divisors,
               moldensity, molrad, molmass, isoimol, isomass, isoratio, isoz, isoiext,
               lwn, elow, gf, lid, cutoff, ethresh, temp, verb=0, add=0, resolution=0,
               return_stats=False):
    """Same positional signature as the reference (_extcoeff.c:114-123);
    writes ext[nextinct, nwave] in place."""
    assert ext.dtype == np.float64 and ext.flags.c_contiguous and ext.ndim == 2
    prof, profp = _d(profile)
    ps, psp = _i(psize)
    pi, pip = _i(pindex)
    lor, lorp = _d(lorentz)
    dop, dopp = _d(doppler)
    wn_, wnp = _d(wn)
    own_, ownp = _d(own)
    div, divp = _i(divisors)
    md, mdp = _d(moldensity)
    mr, mrp = _d(molrad)
    mm, mmp = _d(molmass)
    ii, iip = _i(isoimol)
    im, imp = _d(isomass)
    ir, irp = _d(isoratio)
    iz, izp = _d(isoz)
    ie, iep = _i(isoiext)
    lw, lwp = _d(lwn)
    el, elp = _d(elow)
    g, gp = _d(gf)
    li, lip = _i(lid)
    stats = ExtStats()
    lib().orc_extinction(
        ext.ctypes.data_as(c_dp), C.c_int(ext.shape[0]), C.c_int(ext.shape[1]),
        profp, psp, pip, lorp, C.c_int(len(lor)), dopp, C.c_int(len(dop)),
        wnp, ownp, C.c_int64(len(own_)), divp, C.c_int(len(div)),
        mdp, mrp, mmp, C.c_int(len(mm)),
        iip, imp, irp, izp, iep, C.c_int(len(im)),
        lwp, elp, gp, lip, C.c_int64(len(lw)),
        C.c_double(cutoff), C.c_double(ethresh), C.c_double(temp),
        C.c_int(int(add)), C.c_int(int(resolution)), C.byref(stats))
    if return_stats:
        return dict(ofactor=stats.ofactor, nadd=stats.nadd, nskip=stats.nskip,
                    neval=stats.neval)
    return 1


def interp_ec(extinction_, etable, ttable, temperatures, density, lay1, lay2):
    """_extcoeff.c:367-418; accumulates into extinction_[nlayers, nwave]."""
    return _interp(extinction_, etable, ttable, temperatures, density, lay1, lay2, 0)


def interp_ec_per_mol(extinction_, etable, ttable, temperatures, density, lay1, lay2):
    """_extcoeff.c:422-472; accumulates into extinction_[nmol, nlayers, nwave]."""
    return _interp(extinction_, etable, ttable, temperatures, density, lay1, lay2, 1)


def _interp(ext, etable, ttable, temperatures, density, lay1, lay2, per_mol):
    assert ext.dtype == np.float64 and ext.flags.c_contiguous
    et, etp = _d(etable)
    tt, ttp = _d(ttable)
    te, tep = _d(temperatures)
    de, dep = _d(density)
    nmol, ntemp, nlayers, nwave = et.shape
    lib().orc_interp_ec(ext.ctypes.data_as(c_dp), etp, ttp, tep, dep,
                        C.c_int(nmol), C.c_int(ntemp), C.c_int(nlayers), C.c_int(nwave),
                        C.c_int(lay1), C.c_int(lay2), C.c_int(per_mol))
    return 1


# --------------------------------------------------------------------------
# _trapezoid.*
# --------------------------------------------------------------------------
def trapezoid(data, intervals):
    d, dp = _d(data)
    h, hp = _d(intervals)
    if len(h) < 1:
        return 0.0
    return lib().orc_trapezoid(dp, hp, C.c_int(len(h)))


def trapezoid2D(data, intervals, nint):
    d, dp = _d(data)
    h, hp = _d(intervals)
    n, np_ = _i(nint)
    out = np.empty(d.shape[1])
    lib().orc_trapezoid2D(out.ctypes.data_as(c_dp), dp, hp, np_, C.c_int(d.shape[1]))
    return out


def cumulative_sum(output, data, intervals, threshold):
    d, dp = _d(data)
    h, hp = _d(intervals)
    assert output.dtype == np.float64 and output.flags.c_contiguous
    return lib().orc_cumulative_sum(output.ctypes.data_as(c_dp), dp, hp,
                                    C.c_int(len(h)), C.c_double(threshold))


def plane_parallel_optical_depth(depth, ideep, extinction_, intervals, maxdepth,
                                 itop, ibottom):
    assert depth.dtype == np.float64 and depth.flags.c_contiguous
    ec, ecp = _d(extinction_)
    h, hp = _d(intervals)
    idp, idpp = _i(ideep)
    nlayers, nwave = depth.shape
    lib().orc_plane_parallel_optical_depth(
        depth.ctypes.data_as(c_dp), idpp, ecp, hp, C.c_double(maxdepth),
        C.c_int(itop), C.c_int(ibottom), C.c_int(nlayers), C.c_int(nwave))
    ideep[...] = idp
    return None


def optdepth(data, intervals, taumax, ideep, ilay):
    d, dp = _d(data)
    h, hp = _d(intervals)
    idp, idpp = _i(ideep)
    nwave = d.shape[1]
    tau = np.empty(nwave)
    lib().orc_optdepth(tau.ctypes.data_as(c_dp), dp, hp, C.c_int(len(h)),
                       C.c_double(taumax), idpp, C.c_int(ilay), C.c_int(nwave))
    ideep[...] = idp
    return tau


def intensity(tau, ideep, planck, mu, rtop):
    t, tp = _d(tau)
    b, bp = _d(planck)
    m, mp = _d(mu)
    idp, idpp = _i(ideep)
    nlayers, nwave = t.shape
    out = np.empty((len(m), nwave))
    lib().orc_intensity(out.ctypes.data_as(c_dp), tp, idpp, bp, mp, C.c_int(len(m)),
                        C.c_int(rtop), C.c_int(nlayers), C.c_int(nwave))
    return out


# --------------------------------------------------------------------------
# _blackbody.*
# --------------------------------------------------------------------------
def blackbody_wn_2D(wn, temp, B=None, last=None):
    w, wp = _d(wn)
    t, tp = _d(temp)
    ret = B is None
    if B is None:
        B = np.empty((len(t), len(w)))
    assert B.dtype == np.float64 and B.flags.c_contiguous
    if last is None:
        lp = None
    else:
        l_, lp = _i(last)
    lib().orc_blackbody_wn_2D(B.ctypes.data_as(c_dp), wp, C.c_int(len(w)), tp,
                              C.c_int(len(t)), lp)
    return B if ret else 1


def blackbody_wn(wn, temp, B=None):
    w, wp = _d(wn)
    ret = B is None
    if B is None:
        B = np.empty(len(w))
    lib().orc_blackbody_wn(B.ctypes.data_as(c_dp), wp, C.c_int(len(w)), C.c_double(temp))
    return B if ret else 1


def exp1(x):
    """scipy.special.exp1 for real arguments (xsf/expint.h:22-52)."""
    f = lib().orc_exp1
    f.restype = C.c_double
    x = np.asarray(x, np.float64)
    return np.array([f(C.c_double(v)) for v in x.ravel()]).reshape(x.shape)


def internal_flux(wn, tint):
    """f_int of pyrat/spectrum.py:475-478."""
    w, wp = _d(wn)
    out = np.empty(len(w))
    lib().orc_internal_flux(out.ctypes.data_as(c_dp), wp, C.c_int(len(w)), C.c_double(tint))
    return out


def two_stream(depth, wn, temp, f_int, flux_top=None, rtop=0):
    """(flux_down, flux_up) of pyrat/spectrum.py:454-522 from depth[L,W]."""
    d, dp = _d(depth)
    nlayers, nwave = d.shape
    B = blackbody_wn_2D(wn, temp)
    fi, fip = _d(f_int)
    if flux_top is None:
        ftp = None
    else:
        ft, ftp = _d(flux_top)
    down = np.empty((nlayers, nwave))
    up = np.empty((nlayers, nwave))
    lib().orc_two_stream(down.ctypes.data_as(c_dp), up.ctypes.data_as(c_dp), dp,
                         B.ctypes.data_as(c_dp), fip, ftp, C.c_int(rtop), C.c_int(nlayers),
                         C.c_int(nwave))
    return down, up


# --------------------------------------------------------------------------
# _simpson.*
# --------------------------------------------------------------------------
def geth(h):
    h_, hp = _d(h)
    n = len(h_)
    if n == 0:
        return [0, 0, 0]
    hsum = np.empty(n // 2)
    hratio = np.empty(n // 2)
    hfactor = np.empty(n // 2)
    lib().orc_geth(hp, C.c_int(n), hsum.ctypes.data_as(c_dp),
                   hratio.ctypes.data_as(c_dp), hfactor.ctypes.data_as(c_dp))
    return [hsum, hratio, hfactor]


def simps(y, h, hsum, hratio, hfactor):
    y_, yp = _d(y)
    h_, hp = _d(h)
    a, ap = _d(hsum)
    b, bp = _d(hratio)
    c, cp = _d(hfactor)
    return lib().orc_simps(yp, C.c_int(len(y_)), hp, ap, bp, cp)


def simps2D(y, h, nint, hsum, hratio, hfactor):
    y_, yp = _d(y)
    h_, hp = _d(h)
    n, np_ = _i(nint)
    a, ap = _d(hsum)
    b, bp = _d(hratio)
    c, cp = _d(hfactor)
    out = np.empty(y_.shape[1])
    lib().orc_simps2D(out.ctypes.data_as(c_dp), yp, C.c_int(y_.shape[1]), hp, np_,
                      ap, bp, cp)
    return out


# --------------------------------------------------------------------------
# cutils.*, _indices.*
# --------------------------------------------------------------------------
def ediff(arr):
    a, ap = _d(arr)
    out = np.empty(max(len(a) - 1, 0))
    lib().orc_ediff(out.ctypes.data_as(c_dp), ap, C.c_int(len(a)))
    return out


def arrbinsearch(values, array):
    v, vp = _d(values)
    a, ap = _d(array)
    out = np.empty(len(v), np.int32)
    lib().orc_arrbinsearch(out.ctypes.data_as(c_ip), vp, C.c_int(len(v)), ap,
                           C.c_int(len(a)))
    return out


def ifirst(data, default_ret=-1):
    d, dp = _i(data)
    return lib().orc_ifirst(dp, C.c_int(len(d)), C.c_int(default_ret))


def ilast(data, default_ret=-1):
    d, dp = _i(data)
    return lib().orc_ilast(dp, C.c_int(len(d)), C.c_int(default_ret))


# --------------------------------------------------------------------------
# Host pre-computes of the callers (NumPy restatements)
# --------------------------------------------------------------------------
def transit_path(radius, nskip=0):
    """pyratbay/atmosphere/atmosphere.py:737-802: chord segments between
    concentric shells for each impact parameter."""
    rad = np.asarray(radius, float)[nskip:]
    # The reference squares SCALARS (`rad[i]**2`: libm pow), and pow(x, 2) is not always the
    # correctly rounded x*x that NumPy's array power computes (0.09 % of values differ by one
    # ulp).  Every square here is the scalar pow, like there: mixing the two forms makes
    # rad[r]**2 - rad[r]**2 non-zero, and its square root NaN, for one atmosphere in ~30.
    sq = np.array([x**2 for x in rad.tolist()], float)
    path = [np.empty(0) for _ in range(nskip)]
    for r in range(len(rad)):
        path.append(np.sqrt(sq[:r] - sq[r]) - np.sqrt(sq[1:r + 1] - sq[r]))
    return path


def divisors(number):
    """pyratbay/tools/tools.py:314-323."""
    return np.array([i for i in range(1, int(number) + 1) if number % i == 0], int)


def voigt_sizes(lorentz, doppler, extent, cutoff, ownstep, onwave, dlratio):
    """pyratbay/pyrat/voigt.py:109-130: half-sizes (0 = cell not computed)."""
    lorentz = np.asarray(lorentz)
    doppler = np.asarray(doppler)
    size = np.zeros((len(lorentz), len(doppler)), int)
    for i in range(len(lorentz)):
        pwidth = extent * (0.5346 * lorentz[i]
                           + np.sqrt(0.2166 * lorentz[i]**2 + doppler**2))
        if cutoff > 0:
            pwidth = np.minimum(pwidth, cutoff)
        psize = 1 + 2 * np.asarray(pwidth / ownstep + 0.5, int)
        psize = np.clip(psize, 3, 1 + 2 * onwave)
        skip = doppler / lorentz[i] < dlratio
        skip[0] = False
        psize[skip] = 0
        size[i] = psize // 2
    return size


def optical_depth_transit(ec, radius, itop, ibottom, maxdepth):
    """pyratbay/opacity/optic_depth.py:103-112 (transit branch)."""
    nlayers, nwave = ec.shape
    raypath = transit_path(radius, itop)
    depth = np.zeros((nlayers, nwave))
    ideep = np.full(nwave, -1, np.int32)
    r = itop
    for r in range(itop, ibottom):
        depth[r] = optdepth(ec[itop:r + 1], raypath[r], maxdepth, ideep, r)
    ideep[ideep < 0] = r
    return depth, ideep


def transmission(depth, radius, rstar, ideep, itop):
    """pyratbay/spectrum/radiative_transfer.py:57-71 (no cloud deck)."""
    nlay = ideep - itop + 1
    h = np.ediff1d(radius[itop:])
    integ = np.exp(-depth[itop:]) * np.expand_dims(radius[itop:], 1)
    spectrum = trapezoid2D(integ, h, nlay - 1)
    return (radius[itop]**2 + 2 * spectrum) / rstar**2


def transmission_deck(depth, radius, rstar, ideep, itop, deck_rsurf=None, deck_itop=None):
    """radiative_transfer.py:17-71 with the cloud-deck branch (numpy + orc_trapezoid2D)."""
    depth = np.asarray(depth, float)
    radius = np.asarray(radius, float)
    nlayers = np.asarray(ideep) - itop + 1
    h = np.ediff1d(radius[itop:])
    integ = np.exp(-depth[itop:]) * np.expand_dims(radius[itop:], 1)
    if deck_rsurf is not None and deck_itop > itop:
        h[deck_itop - itop - 1] = deck_rsurf - radius[deck_itop - 1]
        k = deck_itop - itop
        x_lo, x_hi = radius[deck_itop], radius[deck_itop - 1]      # interp1d sorts x
        slope = (integ[k - 1] - integ[k]) / (x_hi - x_lo)
        integ[k] = slope * (deck_rsurf - x_lo) + integ[k]
    spectrum = trapezoid2D(np.ascontiguousarray(integ), h, (nlayers - 1).astype(np.int32))
    return (radius[itop]**2 + 2 * spectrum) / rstar**2


def emission_deck(depth, ideep, wn, temp, mu, weights, rtop, cloud_tsurf=None, cloud_itop=None):
    """plane_parallel_rt (radiative_transfer.py:74-139) + the quadrature sum."""
    B = blackbody_wn_2D(wn, temp)
    ideep = np.asarray(ideep)
    if cloud_tsurf is not None:
        B[cloud_itop] = blackbody_wn(wn, cloud_tsurf)
        ideep = np.clip(ideep, 0, cloud_itop)
    inten = intensity(depth, ideep.astype(np.int32), B, mu, rtop)
    return np.sum(inten * np.asarray(weights)[:, None], axis=0)


def emission_observables(flux, rt_path, starflux=None, rplanet=None, rstar=None,
                         f_dilution=None):
    """What spectrum() makes of a plane-parallel flux after the radiative transfer
    (pyrat/spectrum.py:394-405) -> (spectrum, fplanet): the dilution factor on every emission-type
    path, the planet-to-star flux ratio on the eclipse ones."""
    fplanet = np.array(flux, float)
    if f_dilution is not None:
        fplanet *= f_dilution
    spectrum = fplanet
    if rt_path in ('eclipse', 'eclipse_two_stream'):
        fstar_rprs = 1 / np.asarray(starflux) * (rplanet / rstar)**2
        spectrum = fplanet * fstar_rprs
    return spectrum, fplanet


def f_lambda_units(spectrum, wn, rplanet, distance):
    """eval()'s conversion of an emission spectrum from erg s-1 cm-2 cm at the planet to
    W m-2 um-1 at the observer (pyrat_obj.py:323-329; pc.um = 1e-4)."""
    return 10.0 * np.asarray(spectrum) * (rplanet / distance * np.asarray(wn) * 1.0e-4)**2


def band_integrate(spectrum, wn, bands):
    """Pyrat.band_integrate without the eclipse factor (pyrat_obj.py:649-660): per band
    PassBand.integrate (spec_tools.py:193-233).  bands: (idx, response, height, photon_counting)."""
    out = np.zeros(len(bands))
    for b, (idx, response, height, counting) in enumerate(bands):
        w = wn[idx]
        if counting:
            wl = 1.0 / (w * 1.0e-4)
            out[b] = np.trapezoid(wl * spectrum[idx] * response, w) * height
        else:
            out[b] = np.trapezoid(spectrum[idx] * response, w) * height
    return out


def eclipse_bandflux(bandflux, rplanet, rstar, bandflux_star):
    """The eclipse branch of Pyrat.band_integrate (pyrat_obj.py:662-665)."""
    rprs = rplanet / rstar
    return np.asarray(bandflux) * (rprs**2.0 / np.asarray(bandflux_star))


# ---------------------------------------------------------------------------------------------
# Loader of sampled cross sections (numpy restatement; test infrastructure)
# ---------------------------------------------------------------------------------------------
def wn_mask(wn, wn_min, wn_max, tol=1.0e-8):
    """pyratbay/spectrum/spec_tools.py:778-814: samples within [wn_min, wn_max] with a tolerance
    of tol x the sampling step at the edges."""
    wn = np.asarray(wn, float)
    mask = (wn >= wn_min) & (wn <= wn_max)
    if np.sum(mask) < 2:
        min_dwn = max_dwn = 0
    else:
        min_dwn = np.abs(np.ediff1d(wn[mask][0:2]))
        max_dwn = np.abs(np.ediff1d(wn[mask][-2:]))
    return (wn >= wn_min - min_dwn * tol) & (wn <= wn_max + max_dwn * tol)


def interpolate_opacity(cs, temp, press, temperature=None, pressure=None):
    """pyratbay/tools/tools.py:1026-1107 on arrays: cs[ntemp, nlayers, nwave] tabulated at
    (temp, press) -> the same on (temperature, pressure): linear in log(cs) over log(p), then over
    T, constant beyond the table; untouched when both grids agree with the table's to 1 %."""
    cs = np.asarray(cs, float)
    resample_p = pressure is not None and (
        len(press) != len(pressure) or np.any(np.abs(1.0 - press / pressure) > 0.01))
    resample_t = temperature is not None and (
        len(temp) != len(temperature) or np.any(np.abs(1.0 - temp / temperature) > 0.01))
    if not resample_p and not resample_t:
        return cs
    with np.errstate(divide='ignore'):
        log_cs = np.log(cs)
    log_cs[~np.isfinite(log_cs)] = -230.0
    if resample_p:
        x, xt = np.log(pressure), np.log(press)
        out = np.empty((log_cs.shape[0], len(x), log_cs.shape[2]))
        for t in range(log_cs.shape[0]):
            for w in range(log_cs.shape[2]):
                out[t, :, w] = np.interp(x, xt, log_cs[t, :, w])
        log_cs = out
    if resample_t:
        out = np.empty((len(temperature), log_cs.shape[1], log_cs.shape[2]))
        for p in range(log_cs.shape[1]):
            for w in range(log_cs.shape[2]):
                out[:, p, w] = np.interp(temperature, temp, log_cs[:, p, w])
        log_cs = out
    return np.exp(log_cs)


def line_sample_table(tables, temperature=None, pressure=None, min_wn=0.0, max_wn=np.inf,
                      wl_thinning=1):
    """Line_Sample.__init__ (opacity/line_sampling.py:104-275) on in-memory tables: `tables` is
    a list of (species, temp, press, wn, cs[ntemp, nlayers, nwave]) in file order.  Returns
    (species[nspec], temp, press, wn, cs_table[nspec, ntemp, nlayers, nwave]); a species that
    appears in several files is the SUM of its files."""
    _, temp0, press0, wn0, _ = tables[0]
    temp = np.asarray(temp0 if temperature is None else temperature, float)
    press = np.asarray(press0 if pressure is None else pressure, float)
    mask0 = wn_mask(wn0, min_wn, max_wn)
    wn = np.asarray(wn0)[mask0][::wl_thinning]
    species, index = [], []
    for sp, _, ptab, wtab, _ in tables:
        m = wn_mask(wtab, min_wn, max_wn)
        w = np.asarray(wtab)[m][::wl_thinning]
        if len(w) != len(wn) or np.any(np.abs(1.0 - w / wn) > 0.01):
            raise ValueError('wavenumber arrays of the cross-section files do not match')
        if np.amax(press) / np.amax(ptab) - 1 > 1e-3:
            raise ValueError('Pressure profile extends beyond the maximum tabulated pressure')
        if sp not in species:
            species.append(sp)
        index.append(species.index(sp))
    table = np.zeros((len(species), len(temp), len(press), len(wn)))
    for (sp, ttab, ptab, wtab, cs), idx in zip(tables, index):
        m = wn_mask(wtab, min_wn, max_wn)
        sub = np.asarray(cs)[:, :, m][:, :, ::wl_thinning]
        table[idx] += interpolate_opacity(sub, np.asarray(ttab, float), np.asarray(ptab, float),
                                          temp, press)
    return np.array(species), temp, press, wn, table
