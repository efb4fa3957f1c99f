"""End-to-end golden vectors from the REAL reference package (tests/golden/g6_e2e_*.npz,
made by tests/golden/make_golden_e2e.py: pb.run() of a line-by-line transmission and an
emission spectrum on the reference's mock HITRAN H2O list and test atmosphere).

They pin, beyond the kernels, the callers' glue that this repo restates: Voigt-grid
sizing (pyrat/voigt.py:109-130), the fine grid and divisors, transit_path, the per-layer
call sequence, quadrature weights.  CPU test: the oracle; GPU test: the HIP path."""
import numpy as np
import pytest

from pyratbay_amd import synth


def load(golden, rt):
    return golden(f'g6_e2e_{rt}')


def voigt_inputs(g):
    size_in = synth.voigt_sizes(g['lorentz'], g['doppler'], float(g['extent']),
                                float(g['cutoff']), float(g['ownstep']), int(g['onwave']),
                                float(g['dlratio']))
    return size_in


@pytest.mark.parametrize('rt', ['transit', 'emission'])
def test_oracle_reproduces_reference_run(orc, golden, rt):
    g = load(golden, rt)
    # grid bookkeeping restated in synth.py
    assert np.array_equal(synth.divisors(int(g['wnosamp'])), g['divisors'])
    size = voigt_inputs(g)
    index = np.zeros_like(size)
    profile = np.zeros(np.sum(2 * size + 1))
    orc.voigt_grid(profile, size, index, g['lorentz'], g['doppler'], float(g['ownstep']))
    assert np.array_equal(size, g['size_out']) and np.array_equal(index, g['index_out'])
    assert len(profile) == int(g['nprofile'])
    np.testing.assert_allclose(profile[::37], g['profile_sub'], rtol=1e-13)
    L, W = g['ec'].shape
    ec = np.zeros((L, W))
    for layer in range(L):
        row = np.zeros((1, W))
        orc.extinction(row, profile, size, index, g['lorentz'], g['doppler'], g['wn'],
                       g['own'], g['divisors'], g['dens'][layer], g['mol_radius'],
                       g['mol_mass'], g['iso_atm_index'], g['iso_mass'], g['iso_ratio'],
                       g['iso_pf'][:, layer].copy(), g['iso_mol_index'], g['lwn'], g['elow'],
                       g['gf'], g['isoid'], float(g['cutoff']), float(g['ethresh']),
                       g['temp'][layer], 0, 1, 0)
        ec[layer] = row[0]
    np.testing.assert_allclose(ec, g['ec'], rtol=1e-12)
    itop = int(g['rtop'])
    if rt == 'transit':
        depth, ideep = orc.optical_depth_transit(ec, g['radius'], itop, L, float(g['maxdepth']))
        assert np.array_equal(ideep, g['ideep'])
        np.testing.assert_allclose(depth, g['depth'], rtol=1e-12)
        spec = orc.transmission(depth, g['radius'], float(g['rstar']), ideep, itop)
    else:
        depth = np.zeros((L, W))
        ideep = np.full(W, L - 1, np.int32)
        orc.plane_parallel_optical_depth(depth, ideep, ec, -orc.ediff(g['radius']),
                                         float(g['maxdepth']), itop, L)
        assert np.array_equal(ideep, g['ideep'])
        np.testing.assert_allclose(depth, g['depth'], rtol=1e-12)
        B = orc.blackbody_wn_2D(g['wn'], g['temp'])
        inten = orc.intensity(depth, ideep, B, g['quadrature_mu'], itop)
        np.testing.assert_allclose(inten, g['intensity'], rtol=1e-12)
        spec = np.sum(inten * g['quadrature_weights'][:, None], axis=0)
    np.testing.assert_allclose(spec, g['spectrum'], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('gather', ['global', 'staged'])
@pytest.mark.parametrize('rt', ['transit', 'emission'])
def test_hip_path_reproduces_reference_run(golden, rt, gather):
    """Same inputs through libpbhip: Voigt table built on the device, all layers in one
    extinction call, device-resident depth and spectrum.  rtol 1e-10 (north_star: 1e-6)."""
    from pyratbay_amd import engine
    engine.require_gpu()
    g = load(golden, rt)
    L, W = g['ec'].shape
    vt = engine.VoigtTable.build(g['lorentz'], g['doppler'], voigt_inputs(g),
                                 float(g['ownstep']), int(g['wnosamp']))
    assert np.array_equal(vt.size, g['size_out']) and np.array_equal(vt.index, g['index_out'])
    ll = engine.LineList(g['lwn'], g['elow'], g['gf'], g['isoid'], len(g['iso_mass']),
                         g['own'])
    lbl = engine.LBL(vt, ll, g['wn'], g['divisors'], g['mol_radius'], g['mol_mass'],
                     g['iso_atm_index'], g['iso_mass'], g['iso_ratio'], g['iso_mol_index'],
                     float(g['cutoff']), float(g['ethresh']), max_layers=L)
    lbl.set_gather_mode(gather)
    ec = lbl.extinction(engine.dev(g['temp']), engine.dev(g['dens']), engine.dev(g['iso_pf']),
                        add=True).view(L, W)
    want = g['ec']
    got = ec.cpu().numpy()
    assert np.array_equal(got == 0, want == 0)
    np.testing.assert_allclose(got, want, rtol=1e-10)
    itop = int(g['rtop'])
    radius = engine.dev(g['radius'])
    if rt == 'transit':
        path = engine.dev(engine.pack_raypath(engine.transit_path(g['radius'], itop), itop))
        spec, depth, ideep = engine.transit_spectrum(ec, path, radius, float(g['rstar']), itop,
                                                     L, float(g['maxdepth']))
    else:
        depth, ideep = engine.plane_parallel_optical_depth(
            ec, engine.dev(-np.diff(g['radius'])), itop, L, float(g['maxdepth']))
        spec = engine.emission_flux(depth, ideep, engine.dev(g['wn']), engine.dev(g['temp']),
                                    engine.dev(g['quadrature_mu']),
                                    engine.dev(g['quadrature_weights']), itop)
    assert np.array_equal(ideep.cpu().numpy(), g['ideep'])
    np.testing.assert_allclose(depth.cpu().numpy(), g['depth'], rtol=1e-10)
    np.testing.assert_allclose(spec.cpu().numpy(), g['spectrum'], rtol=1e-10)
    print(f'{rt}/{gather}: spectrum max rel err vs pb.run() = '
          f'{np.max(np.abs(spec.cpu().numpy() / g["spectrum"] - 1)):.2e}')
