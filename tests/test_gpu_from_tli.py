"""TLI file + atmosphere -> spectrum with nothing taken from a fixture but the configuration
(the partition-function interpolation of line_by_line.py:156-158, 219-222 and the
isotope bookkeeping of :120-200 in front of the HIP path).  The mock H2O file was written by the
reference's own `runmode = tli` (G13); G6 is the reference's pb.run() on that file."""
import os

import numpy as np
import pytest

from pyratbay_amd import tli

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TLI = os.path.join(HERE, 'g13_mock_h2o.tli')


def _config(g):
    """The run's configuration as the reference's cfg file states it (grid, Voigt keys) and its
    atmosphere; NOT the line arrays, isotope tables or partition functions."""
    grid = dict(wn=g['wn'], own=g['own'], ownstep=float(g['ownstep']), onwave=int(g['onwave']),
                nwave=len(g['wn']), wnosamp=int(g['wnosamp']), divisors=g['divisors'],
                wnstep=float(g['wn'][1] - g['wn'][0]))
    species = [f'X{i}' for i in range(len(g['mol_mass']))]
    species[int(g['iso_atm_index'][0])] = 'H2O'
    atm = dict(temp=g['temp'], dens=g['dens'], radius=g['radius'], press=g['press'],
               species=species, mol_mass=g['mol_mass'], mol_radius=g['mol_radius'],
               rstar=float(g['rstar']))
    voigt = dict(extent=float(g['extent']), cutoff=float(g['cutoff']),
                 dlratio=float(g['dlratio']), lorentz=g['lorentz'], doppler=g['doppler'])
    return grid, atm, voigt


@pytest.mark.parametrize('rt', ['transit', 'emission'])
def test_g6_spectra_from_the_tli_file_and_the_atmosphere_alone(golden, rt):
    from pyratbay_amd import engine
    g = golden(f'g6_e2e_{rt}')
    grid, atm, voigt = _config(g)
    kw = {}
    if rt == 'emission':
        kw = dict(quadrature_mu=g['quadrature_mu'], quadrature_weights=g['quadrature_weights'])
    m = engine.LBLSpectrum.from_tli(TLI, atm, grid, ethresh=float(g['ethresh']),
                                    maxdepth=float(g['maxdepth']), rt_path=rt,
                                    itop=int(g['rtop']), **voigt, **kw)
    # what from_tli derived from the file equals what the reference derived from it
    c = m.case
    assert np.array_equal(c['lines']['lwn'], g['lwn']) and np.array_equal(c['lines']['lid'], g['isoid'])
    assert np.array_equal(c['iso']['isoimol'], g['iso_atm_index'])
    assert np.array_equal(c['iso']['isoiext'], g['iso_mol_index'])
    assert np.array_equal(c['iso']['isomass'], g['iso_mass'])
    assert np.array_equal(c['iso']['isoratio'], g['iso_ratio'])
    assert np.array_equal(c['iso']['isoz'], g['iso_pf'])
    # (the reference's `size_out` is vprofile.grid's OUTPUT: aliased cells filled in; the input
    # marks them 0)
    computed = c['voigt']['size'] > 0
    assert np.array_equal(c['voigt']['size'][computed], g['size_out'][computed])
    spec = m.run().cpu().numpy()
    ec = m.ec.view(m.nlayers, m.wcount).cpu().numpy()
    assert np.array_equal(ec == 0, g['ec'] == 0)
    np.testing.assert_allclose(ec, g['ec'], rtol=1e-10)
    np.testing.assert_allclose(spec, g['spectrum'], rtol=1e-10)
    # a new temperature profile: Z(T) on the device == the host restatement, bit for bit, and
    # the run equals the one that is handed that isoz explicitly
    temp2 = g['temp'] * 1.07 + 11.0
    dens2 = g['dens'] * (g['temp'] / temp2)[:, None]
    m.set_atmosphere(temp2, dens2)
    want_z = tli.iso_partition(m.databases, temp2)
    assert np.array_equal(m.isoz.cpu().numpy(), want_z)
    s_dev = m.run().cpu().numpy().copy()
    m.set_atmosphere(temp2, dens2, want_z)
    assert np.array_equal(m.run().cpu().numpy(), s_dev)
    assert np.max(np.abs(s_dev / spec - 1)) > 1e-6          # (the atmosphere did change)
    # outside the table: the reference's interp1d raises
    with pytest.raises(ValueError):
        m.set_atmosphere(np.full_like(temp2, 7000.0), dens2)


def test_from_tli_default_width_grids_and_species_errors(golden):
    """Width grids derived from the atmosphere (voigt.py:27-105) instead of given: the table
    differs from the fixture's in the 10th digit of the widths, the spectrum stays within 1e-6."""
    from pyratbay_amd import engine
    g = golden('g6_e2e_transit')
    grid, atm, voigt = _config(g)
    voigt.pop('lorentz'), voigt.pop('doppler')
    m = engine.LBLSpectrum.from_tli([TLI], atm, grid, nlor=len(g['lorentz']),
                                    ndop=len(g['doppler']), ethresh=float(g['ethresh']),
                                    maxdepth=float(g['maxdepth']), itop=int(g['rtop']), **voigt)
    np.testing.assert_allclose(m.run().cpu().numpy(), g['spectrum'], rtol=1e-6)
    bad = dict(atm, species=[f'X{i}' for i in range(len(g['mol_mass']))])
    with pytest.raises(ValueError, match='not present in the atmosphere'):
        engine.LBLSpectrum.from_tli(TLI, bad, grid, **voigt)


def test_partition_table_for_a_batch_of_walkers(golden):
    """The device form for batched walkers: temp[walkers, L] -> Z[niso, walkers, L], bit-equal to
    the host restatement; check=False leaves NaN for walkers outside the table, no exception."""
    import torch
    from pyratbay_amd import engine
    dbs = tli.read_tli(os.path.join(HERE, 'g13_two_db.tli'))[0]
    pt = engine.PartitionTable(dbs)
    rng = np.random.default_rng(3)
    lo = max(d['temperatures'][0] for d in dbs)
    hi = min(d['temperatures'][-1] for d in dbs)
    temps = rng.uniform(lo, hi, (37, 51))
    temps[0, :3] = [lo, hi, dbs[0]['temperatures'][5]]          # the ends and a node
    z = pt.evaluate(engine.dev(temps)).cpu().numpy()
    assert z.shape == (pt.niso, 37, 51)
    assert np.array_equal(z, tli.iso_partition(dbs, temps.ravel()).reshape(z.shape))
    temps[5, 7] = hi + 1.0
    with pytest.raises(ValueError):
        pt.evaluate(engine.dev(temps))
    z = pt.evaluate(engine.dev(temps), check=False).cpu().numpy()
    # (NaN for the isotopes of the databases whose table ends below that temperature)
    outside = np.concatenate([np.full(len(d['isotopes']), d['temperatures'][-1] < hi + 1.0)
                              for d in dbs])
    assert outside.any()
    assert np.array_equal(np.isnan(z[:, 5, 7]), outside) and np.isnan(z).sum() == outside.sum()
    torch.cuda.synchronize()


def test_get_ec_of_single_layers(golden):
    """Pyrat.get_ec(layer) of the real package on the G6 transit run (fixture G20,
    make_golden_get_ec.py): the per-species extinction of single layers (add = 0 cross sections x
    the species' density) from the TLI file + atmosphere alone."""
    from pyratbay_amd import engine
    g = golden('g6_e2e_transit')
    g20 = golden('g20_get_ec')
    grid, atm, voigt = _config(g)
    model = engine.LBLSpectrum.from_tli(TLI, atm, grid, ethresh=float(g['ethresh']),
                                        maxdepth=float(g['maxdepth']), rt_path='transit',
                                        itop=int(g['rtop']), **voigt)
    for layer in g20['layers']:
        ec, labels = model.get_ec(int(layer))
        assert labels == [str(x) for x in g20[f'labels_{layer}']] == ['H2O']
        want = g20[f'ec_{layer}']
        got = ec.cpu().numpy()
        assert got.shape == want.shape
        assert np.array_equal(got == 0, want == 0)
        np.testing.assert_allclose(got, want, rtol=1e-12)
    with pytest.raises(Exception, match='layer'):
        model.get_ec(len(g['temp']))
