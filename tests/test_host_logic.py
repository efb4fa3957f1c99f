"""Host-side pre-computes of the product (pyratbay_amd.synth / engine) against the
oracle's independent restatements and hand-checked values.  CPU only."""
import numpy as np

from pyratbay_amd import synth


def test_divisors():
    assert list(synth.divisors(12)) == [1, 2, 3, 4, 6, 12]
    assert list(synth.divisors(1)) == [1]
    assert list(synth.divisors(720))[-3:] == [240, 360, 720]


def test_spectral_grid_matches_reference_rule():
    g = synth.spectral_grid(4000.0, 4010.0, 0.05)
    # default oversampling: first highly-composite number with step/osamp <= 4e-4
    assert g['wnosamp'] == 180
    assert g['nwave'] == 201 and g['wn'][0] == 4000.0
    assert g['onwave'] == 200 * 180 + 1
    assert abs(g['own'][180] - g['wn'][1]) < 1e-9
    g = synth.spectral_grid(4000.0, 4010.0, 0.005)
    assert g['wnosamp'] == 24


def test_transit_path_docstring_example(orc):
    """Example of pyratbay/atmosphere/atmosphere.py:752-776."""
    import importlib
    radius = np.linspace(5.0, 1.0, 5)
    want = [[], [3.0], [1.35424869, 2.64575131], [1.11847408, 1.22803364, 2.23606798],
            [1.02599614, 1.04455622, 1.09637632, 1.73205081]]
    # engine imports torch; only its host helpers are used here
    engine = importlib.import_module('pyratbay_amd.engine')
    for fn in (orc.transit_path, engine.transit_path):
        path = fn(radius)
        for p, w in zip(path, want):
            np.testing.assert_allclose(p, w, rtol=1e-8)
        skipped = fn(radius, nskip=1)
        assert len(skipped[0]) == 0 and len(skipped[1]) == 0
        np.testing.assert_allclose(skipped[2], [2.64575131], rtol=1e-8)
    packed = engine.pack_raypath(engine.transit_path(radius, 1), 1)
    assert len(packed) == 0 + 1 + 2 + 3
    np.testing.assert_allclose(packed[:1], [2.64575131], rtol=1e-8)


def test_voigt_sizes_agree_with_oracle_restatement(orc):
    lor = np.logspace(-6, 1, 15)
    dop = np.logspace(-3, -1, 6)
    a = synth.voigt_sizes(lor, dop, 100.0, 25.0, 1e-3, 50000, 0.1)
    b = orc.voigt_sizes(lor, dop, 100.0, 25.0, 1e-3, 50000, 0.1)
    assert np.array_equal(a, b)
    assert (a[:, 0] > 0).all() and (a == 0).any()
    assert a.max() <= 25000


def test_synthetic_case_shapes():
    c = synth.lbl_case(1001, 7, 500, wnosamp=12, nlor=6, ndop=4, niso=2)
    g, atm, ln, iso = c['grid'], c['atm'], c['lines'], c['iso']
    assert g['nwave'] == 1001 and atm['dens'].shape == (7, 3)
    assert iso['isoz'].shape == (2, 7)
    assert len(ln['lwn']) == 500
    # sorted by isotope, then wavenumber (TLI order)
    for i in range(2):
        v = ln['lwn'][ln['lid'] == i]
        assert np.all(np.diff(v) >= 0)
    assert np.all(np.diff(ln['lid']) >= 0)
    assert np.all(np.diff(atm['radius']) < 0)      # top to bottom
