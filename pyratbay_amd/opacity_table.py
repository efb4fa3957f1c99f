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
    """lbl: engine.LBL plan (built with max_layers >= the chunk size);
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
