"""Voigt table built by the HIP kernel against the golden table of the compiled
reference and against the oracle.  Needs an MI355X.

Tolerance: rtol 2e-12.  The reference accumulates the Region-I series in x87 long
double; the kernel is binary64 (<= 4e-15), device sin/cos/exp add a few
ulp that the cancellation in Region I amplifies."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu
RTOL = 2e-12


@pytest.fixture(scope='module')
def eng():
    from pyratbay_amd import engine
    engine.require_gpu()
    return engine


@pytest.mark.parametrize('osamp', [1, 7, 12])
def test_g1_voigt_golden(eng, golden, osamp):
    g = golden('g1_voigt')
    vt = eng.VoigtTable.build(g['lorentz'], g['doppler'], g['size_in'], float(g['dwn']), osamp)
    assert np.array_equal(vt.size, g['size_out'])
    assert np.array_equal(vt.index, g['index_out'])
    assert vt.nprofile == int(g['used'])
    flat = vt.flat()
    q0, qn, st = int(g['quick_start']), int(g['quick_n']), int(g['quick_stride'])
    err = np.max(np.abs(flat[:q0] / g['profile_head'] - 1))
    print(f'osamp={osamp}: max rel err vs reference table = {err:.2e}')
    np.testing.assert_allclose(flat[:q0], g['profile_head'], rtol=RTOL)
    np.testing.assert_allclose(flat[q0:q0 + qn:st], g['quick_sub'], rtol=RTOL)
    np.testing.assert_allclose(flat[q0:q0 + qn].sum(), g['quick_sum'], rtol=1e-11)


def test_from_flat_round_trip(eng, golden, orc):
    """reference-layout table -> phase-major -> reference layout is the identity."""
    c = cases.extinction_inputs()
    size = c['size'].copy()
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    orc.voigt_grid(profile, size, index, c['lorentz'], c['doppler'], c['own'][1] - c['own'][0])
    for osamp in (12, 5):
        vt = eng.VoigtTable.from_flat(profile, size, index, c['lorentz'], c['doppler'], osamp)
        assert np.array_equal(vt.size, size) and np.array_equal(vt.index, index)
        back = vt.flat()
        assert np.array_equal(back, profile[:vt.nprofile])


def test_realistic_grid_vs_oracle(eng, orc):
    """A width grid as voigt.py builds it (log-spaced, dlratio skipping)."""
    from pyratbay_amd import synth
    case = synth.lbl_case(2001, 10, 10, wnosamp=60, nlor=16, ndop=8, extent=60.0, cutoff=12.0)
    vg, g = case['voigt'], case['grid']
    size = vg['size'].copy()
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    orc.voigt_grid(profile, size, index, vg['lorentz'], vg['doppler'], g['ownstep'])
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], 60)
    assert np.array_equal(vt.size, size) and np.array_equal(vt.index, index)
    np.testing.assert_allclose(vt.flat(), profile[:vt.nprofile], rtol=RTOL)


def test_bad_arguments(eng):
    from pyratbay_amd._capi import PbError
    with pytest.raises(PbError):
        eng.VoigtTable.build([1e-3], [1e-2, 2e-2], [[0, 5]], 1e-3, 4)   # size 0 in column 0
    with pytest.raises(PbError):
        eng.VoigtTable.build([1e-3], [1e-2], [[-3]], 1e-3, 4)
