"""Deterministic synthetic workloads of the shapes named in BASELINE.json.

Host-side NumPy only.  The spectral-grid and Voigt-grid set-up mirror what the
reference's callers do before they reach the native kernels, so that the arrays
handed to the hot path have the same structure as in a real `pyrat.run()`:

* spectral grid + fine grid + divisors : pyratbay/pyrat/spectrum.py:203-228,
  pyratbay/tools/tools.py:314-323
* Voigt width grids and half-sizes      : pyratbay/pyrat/voigt.py:27-130,
  pyratbay/opacity/broadening/broadening.py:367-498
* number densities                      : pyratbay/atmosphere/atmosphere.py:629-664
* line-list ordering (isotope, then wn) : pyratbay/pyrat/line_by_line.py:298-482

Inputs are synthetic: uniform-random line positions,
Elow ~ U(0, 8000) cm-1, log10 gf ~ U(-12, -6), Z(T) = 1 + T**1.5/10.
"""
import numpy as np

# CGS constants used only to *generate inputs* (CODATA-2018 like the reference's
# Python layer, pyratbay/constants/astrophysical_constants.py:67-109)
K_B = 1.380649e-16
AMU = 1.66053906660e-24
C_LIGHT = 2.99792458e10
G_GRAV = 6.67430e-8
BAR = 1e6
ANGSTROM = 1e-8
RJUP = 7.1492e9
RSUN = 6.957e10
MJUP = 1.8982e30

# name: (mass amu, collision radius Angstrom)
SPECIES = {
    'H2': (2.01588, 1.445), 'He': (4.002602, 1.40), 'H2O': (18.01528, 1.60),
    'CO': (28.0101, 1.69), 'CO2': (44.0095, 1.90), 'CH4': (16.0425, 2.00),
}
HCN = np.array([
    1, 2, 4, 6, 12, 24, 36, 48, 60, 120, 180, 240, 360, 720, 840,
    1260, 1680, 2160, 2520, 5040, 7560, 10080, 15120, 20160, 25200,
    27720, 45360, 50400, 55440, 83160, 110880, 221760, 277200])


def divisors(number):
    """All integer divisors of number, ascending (tools.py:314-323)."""
    number = int(number)
    divs = [i for i in range(1, number // 2 + 1) if number % i == 0]
    divs.append(number)
    return np.asarray(divs, np.int32)


def spectral_grid(wnlow, wnhigh, wnstep, wnosamp=None):
    """Constant-step output grid wn and fine grid own (spectrum.py:182-228).  wnlow / wnhigh are
    kept in the result: the reference selects the lines of [spec.wnlow, spec.wnhigh]
    (pyrat/opacity.py:46-47, 105), and wnhigh is the configured boundary, not wn[-1]."""
    if wnosamp is None:
        wnosamp = int(HCN[wnstep / HCN <= 0.0004][0])
    nwave = int((wnhigh - wnlow) / wnstep) + 1
    wn = wnlow + np.arange(nwave) * wnstep
    ownstep = wnstep / wnosamp
    onwave = int(np.ceil((wn[-1] - wnlow) / ownstep)) + 1
    own = wnlow + np.arange(onwave) * ownstep
    return dict(wn=wn, own=own, wnstep=wnstep, ownstep=ownstep, nwave=nwave,
                onwave=onwave, wnosamp=int(wnosamp), divisors=divisors(wnosamp),
                wnlow=float(wnlow), wnhigh=float(wnhigh))


def resolution_grid(wnlow, wnhigh, resolution, wnstep=1.0, wnosamp=None):
    """Constant-resolving-power output grid (spec_tools.py:499-504) over the constant-step fine
    grid the reference keeps in that mode: wnstep stays the config's wnstep (default 1.0 cm-1)
    and only sets ownstep = wnstep / wnosamp (spectrum.py:184-222)."""
    if wnosamp is None:
        wnosamp = int(HCN[wnstep / HCN <= 0.0004][0])
    f = 0.5 / resolution
    g = (1.0 + f) / (1.0 - f)
    nwave = int(np.ceil(-np.log(wnlow / wnhigh) / np.log(g)))
    wn = wnlow * g**np.arange(nwave)
    ownstep = wnstep / wnosamp
    onwave = int(np.ceil((wn[-1] - wnlow) / ownstep)) + 1
    own = wnlow + np.arange(onwave) * ownstep
    return dict(wn=wn, own=own, wnstep=wnstep, ownstep=ownstep, nwave=nwave, onwave=onwave,
                wnosamp=int(wnosamp), divisors=divisors(wnosamp), resolution=float(resolution),
                wnlow=float(wnlow), wnhigh=float(wnhigh))


def wlstep_grid(wl_low, wl_high, wlstep, wnstep=1.0, wnosamp=None):
    """Constant-wavelength-step output grid (the reference's `wlstep` mode, spectrum.py:205-210:
    wl = arange(wl_low, wl_high, wlstep), wn = 1 / flip(wl), wnlow = wn[0]; all three in cm) over
    the constant-step fine grid of step wnstep / wnosamp, like resolution_grid.  The extinction
    is interpolated onto it from the layers' dynamic grids exactly as in `resolution` mode
    (extinction.py:163: interpolate = resolution is not None or wlstep is not None)."""
    if wnosamp is None:
        wnosamp = int(HCN[wnstep / HCN <= 0.0004][0])
    wl = np.arange(wl_low, wl_high, wlstep)
    wn = 1.0 / np.flip(wl)
    wnlow = wn[0]
    ownstep = wnstep / wnosamp
    onwave = int(np.ceil((wn[-1] - wnlow) / ownstep)) + 1
    own = wnlow + np.arange(onwave) * ownstep
    return dict(wn=wn, own=own, wnstep=wnstep, ownstep=ownstep, nwave=len(wn), onwave=onwave,
                wnosamp=int(wnosamp), divisors=divisors(wnosamp), wlstep=float(wlstep),
                interpolate=True, wnlow=float(wnlow), wnhigh=1.0 / wl_low)


def voigt_widths(wn, press_bar, masses, radii_cm, nlor, ndop, tmin=100.0, tmax=3000.0):
    """log-spaced Lorentz/Doppler HWHM samples (voigt.py:27-105 with the
    H2-dominated estimates of broadening.py:411-497)."""
    h2_rad, h2_mass = 1.445e-8, 2.01588
    min_mass, max_mass = np.amin(masses), np.amax(masses)
    min_rad, max_rad = np.amin(radii_cm), np.amax(radii_cm)
    dmin = np.sqrt(2 * np.log(2) * K_B * tmin / (max_mass * AMU)) * np.amin(wn) / C_LIGHT
    dmax = np.sqrt(2 * np.log(2) * K_B * tmax / (min_mass * AMU)) * np.amax(wn) / C_LIGHT
    lmin = (np.sqrt(2 / (np.pi * K_B * tmax * AMU)) * np.amin(press_bar) * BAR
            * (h2_rad + min_rad)**2 / C_LIGHT * np.sqrt(1 / max_mass + 1 / h2_mass))
    lmax = (np.sqrt(2 / (np.pi * K_B * tmin * AMU)) * np.amax(press_bar) * BAR
            * (h2_rad + max_rad)**2 / C_LIGHT * np.sqrt(1 / min_mass + 1 / h2_mass))
    doppler = np.logspace(np.log10(dmin), np.log10(dmax), ndop)
    lorentz = np.logspace(np.log10(lmin), np.log10(lmax), nlor)
    return lorentz, doppler


def voigt_sizes(lorentz, doppler, extent, cutoff, ownstep, onwave, dlratio=0.1):
    """Profile half-sizes in fine-grid samples; 0 marks a cell that is not computed
    and will alias the previous Doppler column (voigt.py:109-130)."""
    size = np.zeros((len(lorentz), len(doppler)), np.int64)
    for i, lor in enumerate(lorentz):
        pwidth = extent * (0.5346 * lor + np.sqrt(0.2166 * lor**2 + doppler**2))
        if cutoff > 0:
            pwidth = np.minimum(pwidth, cutoff)
        psize = 1 + 2 * np.asarray(pwidth / ownstep + 0.5, int)
        psize = np.clip(psize, 3, 1 + 2 * onwave)
        skip = doppler / lor < dlratio
        skip[0] = False
        psize[skip] = 0
        size[i] = psize // 2
    return size


def band_positions(rng, n, wnlow, wnhigh, nbands=8, contrast=300.0, in_bands=0.9,
                   duplicates=0.02):
    """n line positions with BAND STRUCTURE instead of a uniform density: a uniform background
    holding (1 - in_bands) of the lines plus `nbands` Gaussian band heads (random centres, equal
    shares) whose peak line density is `contrast` times the background's -- real molecular lists
    (the reference's LBL tests run on HITRAN, tests/configs/spectrum_transmission_test.cfg) have
    10^2-10^3 x density contrasts between band heads and the gaps -- and a fraction `duplicates`
    of the band lines at EXACTLY the wavenumber of another line (blended lines at a head: the
    co-adding of _extcoeff.c:248-262 and equal sort keys).  Sorted."""
    span = wnhigh - wnlow
    nbg = int(round(n * (1.0 - in_bands)))
    nb = n - nbg
    sigma = in_bands * span / (nbands * np.sqrt(2 * np.pi) * max(contrast - 1.0, 1e-9)
                               * max(1.0 - in_bands, 1e-9))
    centres = rng.uniform(wnlow + 3 * sigma, wnhigh - 3 * sigma, nbands)
    which = rng.integers(0, nbands, nb)
    pos = centres[which] + sigma * rng.standard_normal(nb)
    bad = (pos < wnlow) | (pos > wnhigh)
    pos[bad] = rng.uniform(wnlow, wnhigh, int(bad.sum()))          # (tails beyond the grid)
    ndup = int(duplicates * nb)
    if ndup and nb > 1:
        dst = rng.choice(nb, ndup, replace=False)
        pos[dst] = pos[rng.integers(0, nb, ndup)]
    return np.sort(np.concatenate([pos, rng.uniform(wnlow, wnhigh, nbg)]))


def synthetic_lines(nlines, wnlow, wnhigh, niso=1, seed=42,
                    ratios=(0.997, 2e-3, 4e-4, 3e-4), bands=None):
    """Line list sorted by isotope then wavenumber, as the TLI reader returns it.  bands: None =
    uniform positions; a dict of band_positions' keywords (or True for its
    defaults) = band heads with a 10^2-10^3 x density contrast and duplicated positions."""
    rng = np.random.default_rng(seed)
    frac = np.asarray(ratios[:niso], float)
    frac = frac / frac.sum()
    # most lines belong to the main isotopologue, but give the minor ones enough
    counts = np.maximum((nlines * np.maximum(frac, 0.05 if niso > 1 else 1.0)
                         / np.sum(np.maximum(frac, 0.05 if niso > 1 else 1.0))).astype(int), 1)
    if counts.sum() > nlines:
        # fewer lines than isotopes (or nearly): one line each for as many isotopes as there are lines
        counts = (np.arange(niso) < nlines).astype(int) if nlines < niso else counts
        while counts.sum() > nlines:
            counts[np.argmax(counts)] -= 1
    counts[0] += nlines - counts.sum()
    lwn, lid = [], []
    for i, c in enumerate(counts):
        if bands:
            kw = bands if isinstance(bands, dict) else {}
            lwn.append(band_positions(rng, int(c), wnlow, wnhigh, **kw))
        else:
            lwn.append(np.sort(rng.uniform(wnlow, wnhigh, c)))
        lid.append(np.full(c, i, np.int32))
    lwn = np.concatenate(lwn)
    lid = np.concatenate(lid)
    elow = rng.uniform(0.0, 8000.0, nlines)
    gf = 10.0**rng.uniform(-12.0, -6.0, nlines)
    return dict(lwn=lwn, elow=elow, gf=gf, lid=lid)


def partition_function(temp):
    return 1.0 + np.asarray(temp, float)**1.5 / 10.0


def synthetic_atmosphere(nlayers, species=('H2', 'He', 'H2O'), vmr=(0.85, 0.149, 4e-4),
                         ptop=1e-6, pbottom=1e2, t_top=1000.0, t_bottom=1700.0,
                         mplanet=0.6 * MJUP, rplanet=1.0 * RJUP, refpressure=0.1,
                         rstar=1.27 * RSUN):
    """Layers from top (low p) to bottom, smooth T(p), uniform VMRs, ideal-gas number
    densities and a hydrostatic radius profile (g = GM/r**2)."""
    press = np.logspace(np.log10(ptop), np.log10(pbottom), nlayers)     # bar
    x = (np.log10(press) - np.log10(ptop)) / (np.log10(pbottom) - np.log10(ptop))
    temp = t_top + (t_bottom - t_top) * 0.5 * (1 - np.cos(np.pi * x))
    vmr = np.tile(np.asarray(vmr, float), (nlayers, 1))
    dens = vmr * np.expand_dims(press / temp, 1) * BAR / K_B             # cm-3
    mol_mass = np.array([SPECIES[s][0] for s in species])
    mol_radius = np.array([SPECIES[s][1] for s in species]) * ANGSTROM   # cm
    mu = np.sum(vmr * mol_mass, axis=1)
    # hydrostatic: d(1/r) = k T/(mu amu G M) dln p, anchored at refpressure
    lnp = np.log(press)
    integrand = K_B * temp / (mu * AMU * G_GRAV * mplanet)
    cum = np.concatenate([[0.0], np.cumsum(0.5 * (integrand[1:] + integrand[:-1])
                                           * np.diff(lnp))])
    cum_ref = np.interp(np.log(refpressure), lnp, cum)
    radius = 1.0 / (1.0 / rplanet + (cum - cum_ref))
    return dict(press=press, temp=temp, vmr=vmr, dens=dens, radius=radius,
                mol_mass=mol_mass, mol_radius=mol_radius, species=list(species),
                rstar=rstar, nlayers=nlayers)


def lbl_case(nwave_target, nlayers, nlines, *, wnlow=4000.0, wnstep=0.05, wnosamp=None,
             niso=1, nlor=100, ndop=50, extent=300.0, cutoff=25.0, dlratio=0.1,
             seed=42, species=('H2', 'He', 'H2O'), vmr=(0.85, 0.149, 4e-4),
             line_species_index=2, iso_masses=None, ptop=1e-6, pbottom=1e2,
             line_species=None, resolution=None, bands=None):
    """Everything the LBL hot path needs for one synthetic spectrum.

    The line-carrying species is species[line_species_index]; its `niso` isotopes
    have masses m, m+1, ... and the HITEMP-style ratios of SURVEY.md section 8d.

    line_species = tuple of species names (BASELINE config 4: several line lists at
    once): every one of them carries `nlines` lines (seed + its position) and `niso`
    isotopes; the isotopes are numbered species-major, the line list is the
    concatenation of the per-species lists (each sorted by isotope, then wavenumber --
    the order in which line_by_line.py:298-482 concatenates its TLI files) and
    isoiext[i] = position of the isotope's species, so that add=0 gives one extinction
    row per species (pyrat/extinction.py:170-213, _extcoeff.c:203-226,265-272)."""
    wnhigh = wnlow + (nwave_target - 1) * wnstep
    if resolution is not None:
        # `resolution` mode: nwave_target and wnstep only fix the upper boundary; the fine grid
        # follows the reference's default rule (wnstep 1.0 -> wnosamp 2520) unless wnosamp is given
        grid = resolution_grid(wnlow, wnhigh, resolution, 1.0, wnosamp)
    else:
        grid = spectral_grid(wnlow, wnhigh + 0.5 * wnstep, wnstep, wnosamp)
    atm = synthetic_atmosphere(nlayers, species, vmr, ptop=ptop, pbottom=pbottom)
    if line_species is None:
        carriers = [line_species_index]
    else:
        carriers = [list(species).index(s) for s in line_species]
        assert iso_masses is None, 'iso_masses applies to a single line species'
    base_ratios = np.array((0.997, 2e-3, 4e-4, 3e-4)[:niso]) if niso > 1 else np.array([1.0])
    parts, isoimol, isomass, isoratio, isoiext = [], [], [], [], []
    for k, imol in enumerate(carriers):
        ln = synthetic_lines(nlines, grid['wn'][0], grid['wn'][-1], niso, seed + k, bands=bands)
        ln['lid'] = ln['lid'] + np.int32(k * niso)
        parts.append(ln)
        m0 = atm['mol_mass'][imol]
        isoimol += [imol] * niso
        isomass += list(m0 + np.arange(niso)) if iso_masses is None else list(iso_masses)
        isoratio += list(base_ratios)
        isoiext += [k] * niso
    lines = {key: np.concatenate([p[key] for p in parts]) for key in ('lwn', 'elow', 'gf', 'lid')}
    ntot = niso * len(carriers)
    iso = dict(
        isoimol=np.asarray(isoimol, np.int32),
        isomass=np.asarray(isomass, float),
        isoratio=np.asarray(isoratio, float),
        isoiext=np.asarray(isoiext, np.int32),
        isoz=partition_function(atm['temp'])[None, :].repeat(ntot, 0),  # [niso, L]
    )
    lorentz, doppler = voigt_widths(
        grid['wn'], atm['press'], atm['mol_mass'][carriers], atm['mol_radius'][carriers],
        nlor, ndop)
    size = voigt_sizes(lorentz, doppler, extent, cutoff, grid['ownstep'],
                       grid['onwave'], dlratio)
    voigt = dict(lorentz=lorentz, doppler=doppler, size=size, extent=extent,
                 cutoff=cutoff, dlratio=dlratio)
    return dict(grid=grid, atm=atm, lines=lines, iso=iso, voigt=voigt,
                ethresh=1e-30, maxdepth=10.0)
