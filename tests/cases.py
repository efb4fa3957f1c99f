"""Seeded small inputs shared by the golden-vector generator and the parity tests.

Pure NumPy; nothing here computes expected values."""
import os

import numpy as np
import pytest

from pyratbay_amd import synth

# The measured dead ends (gather modes scatter / rounds / wave, the one-pass and layers-outer
# matrix transit kernels, predicted `resolution` run plans) live in libpbhip_exp.so only
# (`make -C pyratbay_amd/csrc EXPERIMENTS=1`, PB_LIBPBHIP=pyratbay_amd/libpbhip_exp.so); their
# tests carry the marker `gpu_experiments` and are deselected otherwise (tests/conftest.py).
EXPERIMENTS = os.environ.get('PB_LIBPBHIP', '').endswith('libpbhip_exp.so')
EXPERIMENTAL_GATHERS = ('scatter', 'rounds', 'wave')


def exp(*values):
    """pytest.param(...) marked gpu_experiments."""
    return pytest.param(*values, marks=pytest.mark.gpu_experiments)


def gathers(*modes):
    """Parametrisation over gather modes: the experimental ones marked gpu_experiments."""
    return [exp(m) if m in EXPERIMENTAL_GATHERS else m for m in modes]


def live(*modes):
    """The gather modes of a loop inside a test that the loaded library carries."""
    return tuple(m for m in modes if EXPERIMENTS or m not in EXPERIMENTAL_GATHERS)


def voigt_case():
    """6 x 4 width grid at a coarse step: contains skipped cells (size 0), cells whose
    step resolves the Doppler core (two-point mean), oversampled/Simpson cells and one
    cell above 99999 samples (QUICK point sampling) -- voigt.h:235-290."""
    lorentz = np.array([1e-4, 2e-3, 0.02, 0.3, 2.0, 6.0])
    doppler = np.array([4e-3, 0.012, 0.05, 0.7])
    dwn = 2.5e-4
    size = np.array([
        [40, 60, 400, 1500],
        [35, 0, 420, 900],
        [300, 0, 0, 2000],
        [2000, 0, 0, 2500],
        [3000, 0, 0, 0],
        [50001, 0, 0, 0],
    ])
    return dict(lorentz=lorentz, doppler=doppler, dwn=dwn, size=size)


def extinction_inputs(resolution=False, seed=7):
    """~2400 lines, 3 isotopes of two species, coarse fine-grid; includes exact
    duplicates / near-coincident lines (co-adding), lines outside the grid and
    an exact tie for the nearest fine-grid index."""
    rng = np.random.default_rng(seed)
    wnlow, wnstep, osamp = 5000.0, 0.05, 12
    nwave = 801
    if resolution:
        # constant resolving power output grid; fine grid keeps wnstep=1/osamp rule
        R = 120000.0
        wn = wnlow * np.exp(np.arange(nwave) / R)
        ownstep = wnstep / osamp
        onwave = int(np.ceil((wn[-1] - wnlow) / ownstep)) + 1
        own = wnlow + np.arange(onwave) * ownstep
    else:
        g = synth.spectral_grid(wnlow, wnlow + (nwave - 1) * wnstep + 0.01, wnstep, osamp)
        wn, own, ownstep, onwave = g['wn'], g['own'], g['ownstep'], g['onwave']
    divisors = synth.divisors(osamp)

    niso = 3
    counts = [1500, 500, 400]
    lwn, lid = [], []
    for i, c in enumerate(counts):
        v = rng.uniform(own[0] - 1.0, own[-1] + 1.0, c)      # some out of range
        # near-coincident pairs -> co-added lines
        v[:c // 10] = v[c // 10:2 * (c // 10)] + rng.uniform(-0.4, 0.4, c // 10) * ownstep
        # an exact mid-point tie between two fine samples
        v[-1] = own[1234 + i] + 0.5 * ownstep
        lwn.append(np.sort(v))
        lid.append(np.full(c, i, np.int32))
    lwn = np.concatenate(lwn)
    lid = np.concatenate(lid)
    nlines = len(lwn)
    elow = rng.uniform(0, 6000.0, nlines)
    gf = 10.0**rng.uniform(-9, -4, nlines)

    atm = synth.synthetic_atmosphere(9, ('H2', 'He', 'H2O', 'CO'),
                                     (0.85, 0.1485, 1e-3, 5e-4), ptop=1e-5, pbottom=50.0)
    iso = dict(
        isoimol=np.array([2, 2, 3], np.int32),
        isomass=np.array([18.01, 20.01, 28.01]),
        isoratio=np.array([0.997, 0.002, 0.99]),
        isoiext=np.array([0, 0, 1], np.int32),
    )
    lorentz, doppler = synth.voigt_widths(
        wn, atm['press'], atm['mol_mass'][2:4], atm['mol_radius'][2:4], 14, 7)
    extent, cutoff = 40.0, 6.0
    size = synth.voigt_sizes(lorentz, doppler, extent, cutoff, ownstep, onwave, 0.1)
    return dict(wn=wn, own=own, divisors=divisors, lwn=lwn, lid=lid, elow=elow, gf=gf,
                atm=atm, iso=iso, lorentz=lorentz, doppler=doppler, size=size,
                cutoff=cutoff, nspec=2)


def iso_z(temp, niso):
    """Synthetic partition functions, one per isotope."""
    return (1.0 + temp**1.5 / 10.0) * (1.0 + 0.1 * np.arange(niso))


def extinction_variants():
    """(layer, add, cutoff_on, ethresh, skip_iso) combinations of fixture G2."""
    out = []
    for add in (0, 1):
        for cut in (0, 1):
            for eth in (1e-30, 1e-3):
                out.append((4, add, cut, eth, 0))
    for layer in (0, 8):
        for add in (0, 1):
            out.append((layer, add, 1, 1e-30, 0))
    out.append((4, 0, 1, 1e-30, 1))      # isoiext = -1 for one isotope
    out.append((2, 1, 1, 1e-6, 1))
    return out


def column_case(seed=3, nlayers=24, nwave=96):
    """Random ec field spanning optically thin..thick columns, plus geometry."""
    rng = np.random.default_rng(seed)
    radius = np.linspace(8.0e9, 7.0e9, nlayers) + rng.uniform(-1e6, 1e6, nlayers)
    radius = np.sort(radius)[::-1].copy()
    press = np.logspace(-6, 2, nlayers)
    scale = 10.0**rng.uniform(-14, -8.5, nwave)
    ec = press[:, None]**0.9 * scale[None, :] * rng.uniform(0.5, 1.5, (nlayers, nwave))
    ec[:, :4] = 0.0                                           # transparent columns
    temp = np.linspace(900.0, 1900.0, nlayers) + rng.uniform(-30, 30, nlayers)
    wn = np.linspace(2000.0, 9000.0, nwave)
    mu = np.cos(np.radians([0.0, 20.0, 40.0, 60.0, 80.0]))
    return dict(radius=radius, ec=ec, temp=temp, wn=wn, mu=mu, rstar=8.8e10,
                nlayers=nlayers, nwave=nwave)


def table_case(seed=11):
    rng = np.random.default_rng(seed)
    nmol, ntemp, nlayers, nwave = 3, 5, 7, 50
    ttable = np.array([300.0, 700.0, 1100.0, 1800.0, 3000.0])
    etable = 10.0**rng.uniform(-30, -20, (nmol, ntemp, nlayers, nwave))
    temps = np.array([300.0, 450.0, 700.0, 1099.999, 1800.0, 2999.0, 3000.0])
    dens = 10.0**rng.uniform(8, 18, (nlayers, nmol))
    return dict(etable=etable, ttable=ttable, temps=temps, dens=dens)
