"""Continuum opacity models on the device.

Host side mirrors the reference's model classes -- same names, constructor arguments and
parameters -- and keeps what depends only on the spectral grid (cross sections, tables
resampled to the grid) resident in HBM.  What depends on the atmosphere is evaluated by
ONE fused kernel (pb_continuum) plus one windowed kernel per alkali species
(pb_alkali_cross_section), accumulating into the extinction coefficient that the
line-by-line or table stage left on the device.

    models = [Kurucz(wn, 'H2'), Lecavelier(pressure, wn), Collision_Induced(path, wn=wn), ...]
    cont = Continuum(wn, pressure, models)
    cont.add(ec, temp, {'H2': n_H2, 'He': n_He, ...})      # ec[L,W] += all terms

Reference: pyratbay/opacity/rayleigh/rayleigh.py, clouds/lecavelier.py, clouds/gray.py,
cia.py, hydrogen_ion.py, alkali/alkali.py; io/io.py:866-950 (read_cs)."""
import ctypes as C

import numpy as np
import torch

from ._capi import call, hptr
from .engine import dev, _ptr, _stream

# pyratbay/constants/astrophysical_constants.py:67-131 (scipy.constants, CODATA 2018)
H = 6.62607015e-27
K = 1.380649e-16
LS = 29979245800.0
BAR = 1e6
AMU = 1.6605390666e-24
AMAGAT = 2.6867801117984436e+19
UM = 1e-4


def _grid(wn, wl):
    if (wn is None) == (wl is None):
        raise ValueError('Either provide wavelength or wavenumber array, not both')
    return np.asarray(1.0 / (np.asarray(wl) * UM) if wn is None else wn, float)


class Kurucz:
    """Rayleigh scattering of H, He, H2 or e- (rayleigh.py:13-107)."""

    def __init__(self, wn, species):
        self.name = f'rayleigh_{species}'
        self.species = species
        self.wn = np.asarray(wn, float)
        wn = self.wn
        if species in ('H', 'H2'):
            c = {'H': (5.799e-45, 1.422e-54, 2.784e-64),
                 'H2': (8.140e-45, 1.280e-54, 1.610e-64)}[species]
            self.cross_section = c[0] * wn**4.0 + c[1] * wn**6.0 + c[2] * wn**8.0
        elif species == 'He':
            c = (5.484e-46, 2.440e-11, 5.940e-42, 2.900e-11)
            self.cross_section = c[0] * wn**4 * (
                1.0 + c[1] * wn**2 + c[2] * wn**4 / (1 - c[3] * wn**2))**2.0
        elif species == 'e-':
            self.cross_section = np.tile(6.653e-25, len(wn))
        else:
            raise ValueError(f"no Rayleigh model for '{species}'")

    def rank1(self, pressure, temperature, density):
        return self.cross_section, density[self.species]


class Lecavelier:
    """Rayleigh-like haze, kappa = 10**pars[0] * s0 * (wn*l0)**(-pars[1]) on a nominal
    density p/kT (lecavelier.py:14-100)."""

    def __init__(self, pressure, wl=None, wn=None):
        self.name = 'lecavelier'
        self.pressure = np.asarray(pressure, float)
        self.wn = _grid(wn, wl)
        self.pars = [0.0, -4.0]
        self.s0, self.l0 = 5.31e-27, 3.5e-5
        self.calc_cross_section()

    def calc_cross_section(self, pars=None):
        if pars is not None:
            self.pars[:] = pars
        self.cross_section = 10.0**self.pars[0] * self.s0 * (self.wn * self.l0)**(-self.pars[1])
        return self.cross_section

    def rank1(self, pressure, temperature, density):
        return self.cross_section, self.pressure * BAR / temperature / K


class CCSgray:
    """Constant-cross-section gray cloud between two pressures (gray.py:17-90)."""

    def __init__(self, pressure, wn):
        self.name = 'ccsgray'
        self.pressure = np.asarray(pressure, float)
        self.wn = np.asarray(wn, float)
        self.pars = [0.0, -4.0, 2.0]
        self.s0 = 5.31e-27

    def calc_cross_section(self):
        p_top, p_bottom = 10**self.pars[2], 10**self.pars[1]
        mask = (self.pressure >= p_bottom) & (self.pressure <= p_top)
        cs = np.zeros(len(self.pressure))
        cs[mask] = 10**self.pars[0] * self.s0
        return cs

    def rank1(self, pressure, temperature, density):
        return np.ones(len(self.wn)), self.calc_cross_section() * (
            self.pressure * BAR / temperature / K)


class Deck:
    """Instantly opaque gray cloud deck at pressure 10**pars[0] bar (gray.py:92-150).  It adds
    nothing to ec; it sets the bottom of the optical-depth integration (ibottom = itop + 1)
    and the cloud-top radius / temperature that the radiative transfer uses."""

    def __init__(self, pressure, wn):
        self.name = 'deck'
        self.pressure = np.asarray(pressure, float)
        self.wn = np.asarray(wn, float)
        self.pars = [-1.0]
        self.itop, self.rsurf, self.tsurf = None, 0.0, 0.0

    def calc_extinction_coefficient(self, radius, temperature, pars=None):
        if pars is not None:
            self.pars[:] = pars
        ptop = 10**self.pars[0]
        nlayers = len(self.pressure)
        if ptop >= self.pressure[-1]:
            self.itop = nlayers - 1
        elif ptop < self.pressure[0]:
            self.itop = 1
        else:
            self.itop = int(np.where(self.pressure >= ptop)[0][0])
        # scipy.interpolate.interp1d (linear) in pressure
        self.tsurf = float(np.interp(ptop, self.pressure, temperature))
        self.rsurf = float(np.interp(ptop, self.pressure, radius))
        return self.itop, self.rsurf, self.tsurf


def read_cs(csfile):
    """Cross-section table file (io/io.py:866-950): '@SPECIES', '@TEMPERATURES', '@DATA'
    blocks -> (cs[ntemp, nwave], species, temps, wn)."""
    species = temps = None
    rows = []
    with open(csfile) as f:
        lines = iter(f.readlines())
    for line in lines:
        line = line.strip()
        if line == '@SPECIES':
            species = next(lines).split()
        elif line == '@TEMPERATURES':
            temps = np.array(next(lines).split(), float)
        elif line == '@DATA':
            break
    for line in lines:
        line = line.strip()
        if line and not line.startswith('#'):
            rows.append(np.array(line.split(), float))
    data = np.array(rows)
    return data[:, 1:].T.copy(), species, temps, data[:, 0].copy()


def second_deriv(yin, xin):
    """src_c/_spline.c:25-74, including its (xin[i+1] - YIN[i-1]) denominator."""
    n = len(yin) - 1
    y2 = np.zeros(n + 1)
    u = np.zeros(max(n, 1))
    for i in range(1, n):
        sig = (xin[i] - xin[i - 1]) / (xin[i + 1] - yin[i - 1])
        p = sig * y2[i - 1] + 2.0
        y2[i] = (sig - 1.0) / p
        u[i] = ((yin[i + 1] - yin[i]) / (xin[i + 1] - xin[i])
                - (yin[i] - yin[i - 1]) / (xin[i] - xin[i - 1]))
        u[i] = (6.0 * u[i] / (xin[i + 1] - xin[i - 1]) - sig * u[i - 1]) / p
    for i in range(n - 1, -1, -1):
        y2[i] = y2[i] * y2[i + 1] + u[i]
    return y2


def splinterp_1D(yin, xin, y2nd, xout, extrap):
    """src_c/_spline.c:95-131 + include/spline.h:6-35 (vectorised; same bracket rule)."""
    xout = np.asarray(xout, float)
    out = np.full(len(xout), float(extrap))
    inside = (xout >= xin[0]) & (xout <= xin[-1])
    x = xout[inside]
    i = np.clip(np.searchsorted(xin, x, side='right') - 1, 0, len(xin) - 2)
    dx = xin[i + 1] - xin[i]
    a = (xin[i + 1] - x) / dx
    b = (x - xin[i]) / dx
    out[inside] = (a * yin[i] + b * yin[i + 1]
                   + ((a * a * a - a) * y2nd[i] + (b * b * b - b) * y2nd[i + 1]) * dx * dx / 6.0)
    return out


class Collision_Induced:
    """CIA table resampled to the model grid at construction (cia.py:20-117); linear in
    temperature per evaluation, on the device."""

    def __init__(self, cia_file=None, *, wn=None, wl=None, table=None):
        absorption, species, temps, tab_wn = table if table is not None else read_cs(cia_file)
        self.cia_file = cia_file
        self.species = list(species)
        self.nspec = len(self.species)
        self.name = 'CIA ' + '-'.join(self.species)
        order = np.argsort(temps)
        absorption = np.asarray(absorption, float)[order]
        self.temps = np.asarray(temps, float)[order]
        self.ntemp = len(self.temps)
        self.tmin, self.tmax = self.temps.min(), self.temps.max()
        if wl is not None and wn is not None:
            raise ValueError('Either provide wl or wn array for CIA, not both')
        if wl is not None:
            wn = 1.0 / (np.asarray(wl) * UM)
        if wn is None:
            self.wn = tab_wn
            cross_section = absorption
        else:
            self.wn = np.asarray(wn, float)
            grid = self.wn[::-1] if self.wn[1] < self.wn[0] else self.wn
            tab = tab_wn[::-1] if tab_wn[1] < tab_wn[0] else tab_wn
            cross_section = np.zeros((self.ntemp, len(self.wn)))
            for j in range(self.ntemp):
                ddev = second_deriv(absorption[j], tab)
                cross_section[j] = splinterp_1D(absorption[j], tab, ddev, grid, 0.0)
            if self.wn[1] < self.wn[0]:
                cross_section = np.fliplr(cross_section)
        self.nwave = len(self.wn)
        self.tab_cross_section = cross_section / AMAGAT**self.nspec
        good = np.where((self.wn >= tab_wn.min()) & (self.wn <= tab_wn.max()))[0]
        self._wn_lo_idx, self._wn_hi_idx = int(good[0]), int(good[-1]) + 1


class Hydrogen_Ion:
    """H- bound-free and free-free opacity (hydrogen_ion.py:17-276; John 1988)."""
    species = ['H', 'e-']

    def __init__(self, wn):
        self.name = 'H- bound-free/free-free'
        self.wn = np.asarray(wn, float)
        wn0 = 6090.5
        c_bf = [152.519, 49.534, -118.858, 92.536, -34.194, 4.982]
        mask = self.wn > wn0
        reduced_wl = 1e-2 * np.sqrt(self.wn[mask] - wn0)
        f_lambda = np.zeros(np.sum(mask))
        for n in range(6):
            f_lambda += c_bf[n] * reduced_wl**n
        self.sigma_bf = np.zeros(len(self.wn))
        self.sigma_bf[mask] = 1.0e-6 * (reduced_wl / self.wn[mask])**3.0 * f_lambda
        # Equation (6): rows multiplying sqrt(5040/T)**(i+2), i = 0..5
        wl = 1e4 / self.wn
        short = [[518.1021, 473.2636, -482.2089, 115.5291],
                 [-734.8666, 1443.4137, -737.1616, 169.6374],
                 [1021.1775, -1977.3395, 1096.8827, -245.649],
                 [-479.0721, 922.3575, -521.1341, 114.243],
                 [93.1373, -178.9275, 101.7963, -21.9972],
                 [-6.4285, 12.3600, -7.0571, 1.5097]]
        long_ = [[2483.346, -3449.889, 2200.040, -696.271, 88.283],
                 [285.827, -1158.382, 2427.719, -1841.400, 444.517],
                 [-2054.291, 8746.523, -13651.105, 8624.970, -1863.864],
                 [2827.776, -11485.632, 16755.524, -10051.530, 2095.288],
                 [-1341.537, 5303.609, -7510.494, 4400.067, -901.788],
                 [208.952, -812.939, 1132.738, -655.020, 132.985]]
        sw = (0.182 < wl) & (wl < 0.3645)
        lw = wl >= 0.3645

        def poly(co, i, x):
            return (co[0][i] * x**2.0 + co[1][i] + co[2][i] / x + co[3][i] / x**2.0
                    + co[4][i] / x**3.0 + co[5][i] / x**4.0)
        self.ff_factors = np.zeros((6, len(self.wn)))
        for i in range(4):
            self.ff_factors[i, sw] = 1.0e-29 * poly(short, i, wl[sw])
        for i in range(5):
            self.ff_factors[i + 1, lw] = 1.0e-29 * poly(long_, i, wl[lw])


class VanderWaals:
    """Alkali resonance doublet, Burrows et al. (2000) (alkali.py:28-262)."""

    def __init__(self, pressure, wn, cutoff):
        self.pressure = np.asarray(pressure, float)
        self.wn = np.asarray(wn, float)
        self.cutoff = cutoff
        self.nlines = len(self.wn0)

    def voigt_det(self, temperature):
        """Voigt value at the detuning distance, [nlayers, nlines] (alkali.py:56-89);
        broadening.Voigt.eval (broadening.py:231-260): Faddeeva function when
        hwhm_L/hwhm_G < 0.1, else the four-term rational approximation."""
        from scipy.special import wofz
        temperature = np.asarray(temperature, float)
        dsigma = self.detuning * (temperature / 500.0)**0.6
        lor = self.lpar * (temperature / 2000.0)**(-0.7) * self.pressure * BAR / 1.01e6
        A = np.array([[-1.2150, -1.3509, -1.2150, -1.3509]]).T
        B = np.array([[1.2359, 0.3786, -1.2359, -0.3786]]).T
        Cc = np.array([[-0.3085, 0.5906, -0.3085, 0.5906]]).T
        D = np.array([[0.0210, -1.1858, -0.0210, 1.1858]]).T
        out = np.zeros((len(temperature), self.nlines))
        ln2 = np.sqrt(np.log(2))
        for j, wn0 in enumerate(self.wn0):
            hg = np.sqrt(2 * K * temperature / (self.mass * AMU)) * wn0 / LS
            x = wn0 + dsigma
            # Faddeeva branch
            sigma = hg / ln2
            z = (x + 1j * lor - wn0) / sigma
            faddeeva = wofz(z).real / (sigma * np.sqrt(np.pi))
            # rational branch (all layers at once; rows = the four terms)
            X = (x - wn0) * ln2 / hg
            Y = lor * ln2 / hg
            V = np.sum((Cc * (Y - A) + D * (X - B)) / ((Y - A)**2 + (X - B)**2), axis=0)
            rational = V * np.sqrt(np.pi * np.log(2.0)) / (np.pi * hg)
            out[:, j] = np.where(lor / hg < 0.1, faddeeva, rational)
        return out


class SodiumVdW(VanderWaals):
    def __init__(self, pressure, *, wn=None, wl=None, cutoff=4500.0):
        self.name, self.species = 'sodium_vdw', 'Na'
        self.wn0, self.gf = [16960.87, 16978.07], [0.65464, 1.30918]
        self.lpar, self.Z, self.detuning, self.mass = 0.071, 2.0, 30.0, 22.989769
        super().__init__(pressure, _grid(wn, wl), cutoff)


class PotassiumVdW(VanderWaals):
    def __init__(self, pressure, *, wn=None, wl=None, cutoff=4500.0):
        self.name, self.species = 'potassium_vdw', 'K'
        self.wn0, self.gf = [12988.76, 13046.486], [0.701455, 1.40929]
        self.lpar, self.Z, self.detuning, self.mass = 0.14, 2.0, 20.0, 39.0983
        super().__init__(pressure, _grid(wn, wl), cutoff)


class Continuum:
    """All continuum terms of a run, resident on the device; add(ec, temp, density) is one
    fused pass over ec plus one pass per alkali species."""

    def __init__(self, wn, pressure, models):
        self.wn_h = np.asarray(wn, float)
        self.pressure = np.asarray(pressure, float)
        self.nwave = len(self.wn_h)
        self.wn = dev(self.wn_h)
        self.rank1 = [m for m in models if hasattr(m, 'rank1')]
        self.cia = [m for m in models if isinstance(m, Collision_Induced)]
        self.hminus = [m for m in models if isinstance(m, Hydrogen_Ion)]
        self.alkali = [m for m in models if isinstance(m, VanderWaals)]
        if len(self.cia) > 4 or len(self.hminus) > 1:
            raise ValueError('at most 4 CIA tables and one H- model per Continuum')
        self.cia_tab = [dev(m.tab_cross_section) for m in self.cia]
        self.cia_temps = [dev(m.temps) for m in self.cia]
        if self.hminus:
            self.hm_sigma_bf = dev(self.hminus[0].sigma_bf)
            self.hm_ff = dev(self.hminus[0].ff_factors)
        self.pressure_barye = dev(self.pressure * BAR)
        self._cs_key = None

    def _rank1_cross_sections(self):
        """[nrank1, nwave] on the device; uploaded again only when a model's parameters
        (Lecavelier: scale and exponent) change."""
        key = tuple((id(m), tuple(np.ravel(getattr(m, 'pars', ())).tolist())) for m in self.rank1)
        if key != self._cs_key:
            rows = []
            for m in self.rank1:
                if isinstance(m, Lecavelier):
                    m.calc_cross_section()
                rows.append(np.ones(self.nwave) if isinstance(m, CCSgray) else m.cross_section)
            self._cs_d = dev(np.array(rows).reshape(len(rows), self.nwave))
            self._cs_key = key
        return self._cs_d

    def add(self, ec, temperature, density):
        """ec[L,W] (device, float64) += every term.  temperature[L] and the number
        densities {species: n[L]} (molecules cm-3) are host arrays: L values each; they
        travel to the device in ONE packed upload per call."""
        temperature = np.asarray(temperature, float)
        nlayers = len(temperature)
        assert ec.shape == (nlayers, self.nwave) and ec.is_contiguous()
        for m in self.cia:
            if np.any(temperature < m.tmin) or np.any(temperature > m.tmax):
                raise ValueError('Invalid temperature, values must be in the '
                                 f'{m.tmin:.1f}-{m.tmax:.1f} K range')
        nr1, ncia = len(self.rank1), len(self.cia)
        # per-layer factors, packed: temperature | rank-1 | CIA | H- | alkali densities
        parts = [temperature]
        parts += [np.asarray(m.rank1(self.pressure, temperature, density)[1], float)
                  for m in self.rank1]
        parts += [np.prod([density[s] for s in m.species], axis=0) for m in self.cia]
        if self.hminus:
            parts.append(np.asarray(density['H'], float) * np.asarray(density['e-'], float))
        parts += [np.asarray(density[m.species], float) for m in self.alkali]
        # ... | the alkali models' Voigt values at the detuning distance [nlayers, nlines] each
        vds = [np.ascontiguousarray(m.voigt_det(temperature), float).ravel() for m in self.alkali]
        packed = dev(np.concatenate([np.broadcast_to(p, nlayers) for p in parts] + vds))
        row = [packed[i * nlayers:(i + 1) * nlayers] for i in range(len(parts))]
        temp_d = row[0]
        f_d = packed[nlayers:(1 + nr1) * nlayers] if nr1 else None
        cia_f_d = packed[(1 + nr1) * nlayers:(1 + nr1 + ncia) * nlayers] if ncia else None
        nxt = 1 + nr1 + ncia
        cs_d = self._rank1_cross_sections() if nr1 else None
        tabs = (C.c_void_p * max(ncia, 1))(*[t.data_ptr() for t in self.cia_tab])
        temps = (C.c_void_p * max(ncia, 1))(*[t.data_ptr() for t in self.cia_temps])
        ntemp = np.array([m.ntemp for m in self.cia] or [0], np.int32)
        lo = np.array([m._wn_lo_idx for m in self.cia] or [0], np.int32)
        hi = np.array([m._wn_hi_idx for m in self.cia] or [0], np.int32)
        hm = (None, None, None)
        if self.hminus:
            hm = (self.hm_sigma_bf, self.hm_ff, row[nxt])
            nxt += 1
        if nr1 or ncia or self.hminus:
            call('pb_continuum', _ptr(ec), _ptr(self.wn), _ptr(temp_d), nlayers, self.nwave,
                 nr1, _ptr(cs_d), _ptr(f_d), ncia,
                 C.cast(tabs, C.c_void_p) if ncia else None,
                 C.cast(temps, C.c_void_p) if ncia else None,
                 hptr(ntemp), hptr(lo), hptr(hi), _ptr(cia_f_d), _ptr(hm[0]), _ptr(hm[1]),
                 _ptr(hm[2]), _stream())
        keep = [packed]
        vd_at = len(parts) * nlayers
        for ia, m in enumerate(self.alkali):
            vd = packed[vd_at:vd_at + len(vds[ia])]
            vd_at += len(vds[ia])
            dens_d = row[nxt + ia]
            wn0, gf = np.array(m.wn0, float), np.array(m.gf, float)
            call('pb_alkali_cross_section', _ptr(ec), _ptr(self.pressure_barye), _ptr(self.wn),
                 _ptr(temp_d), _ptr(vd), float(m.detuning), float(m.mass), float(m.lpar),
                 float(m.Z), float(m.cutoff), hptr(wn0), hptr(gf), m.nlines, _ptr(dens_d),
                 nlayers, self.nwave, _stream())
        del keep        # stream-ordered allocator: safe to release after the launches
        return ec
