"""Cross-section tables: the reference's `runmode = opacity` on the GPU and its .npz file.

compute_opacity() restates pyratbay/pyrat/extinction.py:14-126: for every temperature of
the grid and every pressure layer, the per-species cross section (cm2 molecule-1) =
`_extcoeff.extinction(..., add=0)` with ideal-gas densities at (T, p) and the partition
functions at T.  The reference deals the ntemp*nlayers cells round-robin to forked
processes; here they are the "layers" of ONE batched pb_lbl_extinction call (chunked to a
memory budget).  write_opacity/read_opacity keep the file format of
pyratbay/io/io.py:570-694 (keys species, temperature, pressure, wavenumber, opacity,
units).
"""
import numpy as np
import torch

from . import engine
from .synth import BAR, K_B

UNITS = {
    'temperature': 'K',
    'pressure': 'bar',
    'wavenumber': 'cm-1',
    'cross section': 'cm2 molecule-1',
}


def compute_opacity(lbl, temp_grid, press_bar, vmr, iso_pf, chunk_bytes=8 << 30,
                    out=None):
    """lbl: engine.LBL plan (built with max_layers >= the chunk size; a plan of the `resolution`
    mode -- the usual grid of such tables -- should have set_gather_mode('dynamic'): the cells of
    a chunk are then walked in runs of equal oversampling factor on the layers' dynamic grids,
    5x the direct gather);
    temp_grid[ntemp] K; press_bar[nlayers]; vmr[nlayers, nmol];
    iso_pf[niso, ntemp] partition functions at temp_grid.
    Returns etable[nrows, ntemp, nlayers, nwave] (device tensor; nrows = species rows)."""
    temp_grid = np.asarray(temp_grid, float)
    press_bar = np.asarray(press_bar, float)
    vmr = np.asarray(vmr, float)
    iso_pf = np.asarray(iso_pf, float)
    ntemp, nlayers = len(temp_grid), len(press_bar)
    ncell = ntemp * nlayers
    # cell c = itemp*nlayers + ilayer, like extinction.py:171-176
    temps = np.repeat(temp_grid, nlayers)
    dens = (np.tile(vmr * press_bar[:, None], (ntemp, 1)) * BAR / (K_B * temps[:, None]))
    z = np.repeat(iso_pf, nlayers, axis=1)                      # [niso, ncell]
    rows = lbl.nrows_sep
    per_cell = rows * lbl.nwave * 8
    step = int(max(1, min(lbl.max_layers, chunk_bytes // per_cell, ncell)))
    if out is None:
        out = torch.empty((rows, ntemp, nlayers, lbl.nwave), dtype=torch.float64,
                          device='cuda')
    flat = out.view(rows, ncell, lbl.nwave)
    for c0 in range(0, ncell, step):
        c1 = min(c0 + step, ncell)
        ext = lbl.extinction(engine.dev(temps[c0:c1]), engine.dev(dens[c0:c1]),
                             engine.dev(z[:, c0:c1]), add=False)     # [cells, rows, W]
        flat[:, c0:c1] = ext.permute(1, 0, 2)
    return out


def write_opacity(ofile, species, temp, press, wn, opacity):
    """Same file as pyratbay.io.write_opacity: opacity[ntemp, nlayers, nwave] of ONE
    species, pressures in bar."""
    if not isinstance(species, str):
        raise ValueError("'species' input must be a string")
    if isinstance(opacity, torch.Tensor):
        opacity = opacity.cpu().numpy()
    np.savez(ofile, species=[species], temperature=temp, pressure=press, wavenumber=wn,
             opacity=opacity, units=UNITS)


def read_opacity(ofile, extract='all'):
    """pyratbay.io.read_opacity for the .npz flavour (io.py:609-694)."""
    with np.load(ofile, allow_pickle=True) as f:
        if len(f['species']) > 1:
            raise ValueError('Opacity files must contain a single species')
        species = str(f['species'][0])
        temp, press, wn = f['temperature'], f['pressure'], f['wavenumber']
        opacity = None
        if extract in ('opacity', 'all'):
            opacity = f['opacity']
            if np.ndim(opacity) == 4:
                opacity = opacity[0]
        units = np.ndarray.item(f['units']) if 'units' in f else None
    if units is None:                       # pyratbay < 2.0 stored barye
        press = press / BAR
        units = dict(UNITS)
    if extract == 'opacity':
        return opacity
    if extract == 'arrays':
        return species, temp, press, wn
    return units, species, temp, press, wn, opacity


# ---------------------------------------------------------------------------------------------
# Loader of sampled cross sections: what Line_Sample.__init__ builds from a list of opacity files
# (pyratbay/opacity/line_sampling.py:60-275), with the table assembled on the device
# ---------------------------------------------------------------------------------------------
def _wn_mask(wn, wn_min, wn_max, tol=1.0e-8):
    """pyratbay/spectrum/spec_tools.py:778-814."""
    wn = np.asarray(wn, float)
    mask = (wn >= wn_min) & (wn <= wn_max)
    if np.sum(mask) < 2:
        min_dwn = max_dwn = 0
    else:
        min_dwn = np.abs(np.ediff1d(wn[mask][0:2]))
        max_dwn = np.abs(np.ediff1d(wn[mask][-2:]))
    return (wn >= wn_min - min_dwn * tol) & (wn <= wn_max + max_dwn * tol)


def _brackets(x_table, x_new):
    """(perm, lo, a): the permutation that sorts x_table ascending, and for every x_new the lower
    node IN THE SORTED TABLE and the weight of the node above it.  The reference resamples with
    scipy's interp1d(assume_sorted=False, fill_value=(y[first], y[last])) (tools.py:1086-1104):
    a descending or unsorted tabulated axis is sorted first, and a value beyond the table takes
    the table's FIRST entry in FILE order when below the smallest node and its LAST entry in
    file order when above the largest one (the end nodes themselves for an ascending file)."""
    x_table, x_new = np.asarray(x_table, float), np.asarray(x_new, float)
    perm = np.argsort(x_table, kind='stable')
    xs = x_table[perm]
    if np.any(np.diff(xs) <= 0):
        raise ValueError('tabulated grid holds a value twice')
    n = len(xs)
    inv = np.empty(n, np.intp)
    inv[perm] = np.arange(n)                          # sorted position of every file entry
    lo = np.clip(np.searchsorted(xs, x_new, side='right') - 1, 0, n - 1)
    lo = np.where(x_new >= xs[-1], n - 1, lo)
    hi = np.minimum(lo + 1, n - 1)
    span = xs[hi] - xs[lo]
    with np.errstate(divide='ignore', invalid='ignore'):
        a = np.where(span > 0, (x_new - xs[lo]) / span, 0.0)
    a = np.where((x_new <= xs[0]) | (x_new >= xs[-1]), 0.0, a)
    lo = np.where(x_new < xs[0], inv[0], lo)          # fill_value[0] = the file's first entry
    lo = np.where(x_new > xs[-1], inv[n - 1], lo)     # fill_value[1] = the file's last entry
    return perm, lo.astype(np.int32), a


class CrossSections:
    """species[nspec], temp[ntemp] (K), press[nlayers] (bar), wn[nwave] (cm-1) and
    cs_table[nspec, ntemp, nlayers, nwave] (cm2 molecule-1, device tensor): the attributes of
    the reference's Line_Sample that the retrieval path reads."""

    def __init__(self, species, temp, press, wn, cs_table):
        self.species, self.temp, self.press, self.wn = species, temp, press, wn
        self.cs_table = cs_table
        self.nspec, self.ntemp, self.nlayers, self.nwave = cs_table.shape
        self.tmin, self.tmax = float(np.amin(temp)), float(np.amax(temp))

    def table_spectrum(self, radius, rstar, **kw):
        """An engine.TableSpectrum on this table (the path of Pyrat.eval())."""
        return engine.TableSpectrum(self.cs_table, self.temp, self.wn, radius, rstar, **kw)


def load_cross_sections(cs_files, temperature=None, pressure=None, min_wn=None, max_wn=None,
                        min_wl=None, max_wl=None, wl_thinning=1):
    """Line_Sample.__init__ without its isotope-ratio parameters: read the opacity files, keep
    the wavenumber window [min_wn, max_wn] (or wavelengths in microns) thinned by wl_thinning,
    bring every file onto (temperature, pressure) -- default: the first file's -- with
    tools.interpolate_opacity's rule, ADD the files of a species.  Same errors as the reference
    for mismatching wavenumber grids and for a pressure profile beyond a table.  The table is
    assembled on the device (pb_resample_cross_section): files are uploaded one at a time."""
    if isinstance(cs_files, str):
        cs_files = [cs_files]
    if min_wn is not None and max_wl is not None:
        raise ValueError('Either define min_wn or max_wl, not both')
    if max_wn is not None and min_wl is not None:
        raise ValueError('Either define min_wl or max_wn, not both')
    um = 1e-4
    if min_wn is None:
        min_wn = 0.0 if max_wl is None else 1.0 / (max_wl * um)
    if max_wn is None:
        max_wn = np.inf if min_wl is None else 1.0 / (min_wl * um)
    _, temp0, press0, wn0 = read_opacity(cs_files[0], extract='arrays')
    temp = np.asarray(temp0 if temperature is None else temperature, float)
    press = np.asarray(press0 if pressure is None else pressure, float)
    wn = np.asarray(wn0)[_wn_mask(wn0, min_wn, max_wn)][::wl_thinning]
    species, index, masks, grids = [], [], [], []
    for f in cs_files:
        sp, ttab, ptab, wtab = read_opacity(f, extract='arrays')
        mask = _wn_mask(wtab, min_wn, max_wn)
        w = np.asarray(wtab)[mask][::wl_thinning]
        if len(w) != len(wn) or np.any(np.abs(1.0 - w / wn) > 0.01):
            raise ValueError(f"Wavenumber array of cross-section file '{f}' does not match "
                             'with previous arrays')
        if np.amax(press) / np.amax(ptab) - 1 > 1e-3:
            raise ValueError('Pressure profile extends beyond the maximum tabulated pressure')
        if sp not in species:
            species.append(sp)
        index.append(species.index(sp))
        masks.append(mask)
        grids.append((np.asarray(ttab, float), np.asarray(ptab, float)))
    engine.require_gpu()
    table = torch.zeros((len(species), len(temp), len(press), len(wn)), dtype=torch.float64,
                        device='cuda')
    for f, idx, mask, (ttab, ptab) in zip(cs_files, index, masks, grids):
        cs = read_opacity(f, extract='opacity')
        resample_p = len(ptab) != len(press) or np.any(np.abs(1.0 - ptab / press) > 0.01)
        resample_t = len(ttab) != len(temp) or np.any(np.abs(1.0 - ttab / temp) > 0.01)
        resample = bool(resample_p or resample_t)
        if resample:
            pperm, plo, pa = _brackets(np.log(ptab), np.log(press)) if resample_p else (
                None, np.arange(len(press), dtype=np.int32), np.zeros(len(press)))
            tperm, tlo, ta = _brackets(ttab, temp) if resample_t else (
                None, np.arange(len(temp), dtype=np.int32), np.zeros(len(temp)))
            # a descending / unsorted tabulated axis: the brackets index the SORTED table
            if pperm is not None and np.any(np.diff(pperm) != 1):
                cs = cs[:, pperm]
            if tperm is not None and np.any(np.diff(tperm) != 1):
                cs = cs[tperm]
        else:
            plo, pa = np.arange(len(press), dtype=np.int32), np.zeros(len(press))
            tlo, ta = np.arange(len(temp), dtype=np.int32), np.zeros(len(temp))
        wsel = np.flatnonzero(mask)[::wl_thinning].astype(np.int32)
        # (named, so that every upload outlives the call: a c_void_p does not keep its tensor)
        cs_d, dst = engine.dev(cs), table[idx]
        wsel_d, tlo_d, plo_d = (engine.dev(x, torch.int32) for x in (wsel, tlo, plo))
        ta_d, pa_d = engine.dev(ta), engine.dev(pa)
        engine.call('pb_resample_cross_section', engine._ptr(dst), engine._ptr(cs_d),
                    engine._ptr(wsel_d), engine._ptr(tlo_d), engine._ptr(ta_d),
                    engine._ptr(plo_d), engine._ptr(pa_d), cs.shape[0], cs.shape[1],
                    cs.shape[2], len(temp), len(press), len(wn), int(resample), 1,
                    engine._stream())
        torch.cuda.synchronize()
    return CrossSections(np.array(species), temp, press, wn, table)
