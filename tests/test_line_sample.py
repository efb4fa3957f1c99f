"""Loader of sampled cross sections (the reference's Line_Sample.__init__ +
tools.interpolate_opacity: several opacity files -> one table on the run's grid).  Fixture g15
comes from the reference package itself (tests/golden/make_golden_line_sample.py): the oracle's
numpy restatement is pinned to it on the CPU, the product (table assembled on the device by
pb_resample_cross_section) on the GPU."""
import os

import numpy as np
import pytest

CASES = ('native', 'newp', 'newpt', 'window')


def _tables(g):
    names = [str(x) for x in g['file_order']]
    return names, [(str(g[n + '_species']), g[n + '_temp'], g[n + '_press'], g['wn'], g[n + '_cs'])
                   for n in names]


def _kwargs(g, case):
    kw = {}
    for k in ('pressure', 'temperature', 'min_wn', 'max_wn', 'wl_thinning'):
        key = f'{case}_arg_{k}'
        if key in g:
            kw[k] = g[key] if g[key].ndim else g[key].item()
    return kw


@pytest.mark.parametrize('case', CASES)
def test_oracle_loader_equals_the_reference(golden, orc, case):
    g = golden('g15_line_sample')
    _, tabs = _tables(g)
    sp, t, p, w, tab = orc.line_sample_table([tabs[i] for i in g[f'{case}_files']],
                                             **_kwargs(g, case))
    assert [str(x) for x in sp] == [str(x) for x in g[f'{case}_species']]
    assert np.array_equal(w, g[f'{case}_wn']) and np.array_equal(t, g[f'{case}_temp'])
    assert np.array_equal(p, g[f'{case}_press'])
    np.testing.assert_allclose(tab, g[f'{case}_cs_table'], rtol=1e-12)
    if case == 'native':
        assert np.array_equal(tab, g[f'{case}_cs_table'])          # no resampling: untouched


def test_oracle_loader_errors(golden, orc):
    g = golden('g15_line_sample')
    _, tabs = _tables(g)
    with pytest.raises(ValueError, match='beyond the maximum tabulated pressure'):
        orc.line_sample_table([tabs[0]], pressure=np.logspace(-5, 1.5, 7))
    assert 'beyond the maximum tabulated pressure' in str(g['beyond_table_error'])
    bad = list(tabs[1])
    bad[3] = g['wn'][:-3]
    bad[4] = bad[4][:, :, :-3]
    with pytest.raises(ValueError, match='do not match'):
        orc.line_sample_table([tabs[0], tuple(bad)])


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES)
def test_load_cross_sections_equals_the_reference(golden, orc, tmp_path, case):
    """The product's loader on files written in the reference's format: species order, grids,
    and the table against the reference's own result (1e-12; untouched values bit for bit), the
    same errors, and the table usable by TableSpectrum."""
    import torch
    from pyratbay_amd import opacity_table as ot
    g = golden('g15_line_sample')
    names, tabs = _tables(g)
    paths = []
    for n, (sp, t, p, w, cs) in zip(names, tabs):
        path = os.path.join(tmp_path, f'cross_section_{n}.npz')
        ot.write_opacity(path, sp, t, p, w, cs)
        paths.append(path)
    cs = ot.load_cross_sections([paths[i] for i in g[f'{case}_files']], **_kwargs(g, case))
    assert [str(x) for x in cs.species] == [str(x) for x in g[f'{case}_species']]
    assert np.array_equal(cs.wn, g[f'{case}_wn']) and np.array_equal(cs.temp, g[f'{case}_temp'])
    got = cs.cs_table.cpu().numpy()
    np.testing.assert_allclose(got, g[f'{case}_cs_table'], rtol=1e-12)
    if case == 'native':
        assert np.array_equal(got, g[f'{case}_cs_table'])
    with pytest.raises(ValueError, match='beyond the maximum tabulated pressure'):
        ot.load_cross_sections([paths[0]], pressure=np.logspace(-5, 1.5, 7))
    if case == 'newpt':
        # the table feeds the retrieval path: one eval() on it equals interp_ec on the same table
        nl = cs.nlayers
        radius = np.linspace(7.2e9, 6.9e9, nl)
        model = cs.table_spectrum(radius, 8.8e10)
        temp = np.linspace(700.0, 1600.0, nl)
        dens = np.abs(np.random.default_rng(3).normal(1e12, 1e11, (nl, cs.nspec)))
        spec = model.eval(temp, dens).cpu().numpy()
        ec = np.zeros((nl, cs.nwave))
        orc.interp_ec(ec, got, cs.temp, temp, dens, 0, nl)
        depth, ideep = orc.optical_depth_transit(ec, radius, 0, nl, 10.0)
        want = orc.transmission(depth, radius, 8.8e10, ideep, 0)
        np.testing.assert_allclose(spec, want, rtol=1e-10)
