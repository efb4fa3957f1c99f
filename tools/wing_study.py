#!/usr/bin/env python3
"""Reduced-work far wings: what would it cost, what would it save?  (a
STUDY -- nothing here is on the product path or selectable from it.)

The extinction of a layer is a sum of shifted profiles, ext[j] = sum_lines k P(j - pos).  At C3 / C4 a
profile is 10 001 output samples wide (cutoff 25 cm-1 at 0.005 cm-1 steps) and the gather is bound by
the one LDS read per (line, sample) product (53 % of the LDS peak): only fewer products can help.
Beyond a few half-widths a Voigt profile is smooth, so its wings can be summed on a COARSER grid and
interpolated.  The split must itself be smooth (a hard cut at |delta| = dc is a jump the coarse grid
cannot represent):

    P = P (1 - w) + P w,   w(delta) = 0 for |delta| <= dc, smootherstep up to 1 at dc + T
    core  = sum k [P (1 - w)]  on the full grid, windows of +-(dc + T) only
    wings = sum k [P w]        on every s-th sample of the grid, then cubic (4-point Lagrange)
                               interpolation to the full grid

Both sums run through the PRODUCT's kernels unchanged (two Voigt tables made from the reference
table with VoigtTable.from_flat, a plan with cutoff = dc + T on the full grid, a plan on the sub-grid
wn[::s] with oversampling factor osamp * s): the prototype measures the error of the idea and the time
of its two gathers, not a tuned implementation.

usage: python tools/wing_study.py [nwave] [nlayers] [nlines] [s] [dc_cm] [T_cm]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smootherstep(x):
    x = np.clip(x, 0.0, 1.0)
    return x * x * x * (x * (6.0 * x - 15.0) + 10.0)


def split_tables(profile, size, index, ownstep, dc, T):
    """(core, wing) copies of the reference-layout table: P (1 - w) and P w per cell."""
    core, wing = profile.copy(), profile.copy()
    done = set()
    for cell in range(size.size):
        half = int(size.flat[cell])
        i0 = int(index.flat[cell])
        if half == 0 or i0 in done:
            continue                                        # aliased cell: shares another cell's samples
        done.add(i0)
        d = np.abs(np.arange(-half, half + 1)) * ownstep
        w = smootherstep((d - dc) / T)
        seg = slice(i0, i0 + 2 * half + 1)
        core[seg] = profile[seg] * (1.0 - w)
        wing[seg] = profile[seg] * w
    return core, wing


def lagrange4(y, s, n):
    """Cubic interpolation of y (samples at 0, s, 2s, ...) to 0 .. n-1 along the last axis: 4-point
    Lagrange on the stencil around each target, one-sided at the ends."""
    import torch
    m = y.shape[-1]
    j = torch.arange(n, device=y.device)
    i1 = torch.clamp(j // s, 1, m - 3)                      # stencil i1-1 .. i1+2
    t = (j - i1 * s).double() / s                           # in [0, 1) away from the ends
    w0 = -t * (t - 1) * (t - 2) / 6
    w1 = (t + 1) * (t - 1) * (t - 2) / 2
    w2 = -(t + 1) * t * (t - 2) / 2
    w3 = (t + 1) * t * (t - 1) / 6
    return (y[..., i1 - 1] * w0 + y[..., i1] * w1 + y[..., i1 + 1] * w2 + y[..., i1 + 2] * w3)


def main():
    import torch
    from pyratbay_amd import engine as eng, synth
    a = sys.argv[1:]
    nwave = int(a[0]) if len(a) > 0 else 200001
    nlayers = int(a[1]) if len(a) > 1 else 16
    nlines = int(a[2]) if len(a) > 2 else 200000
    s = int(a[3]) if len(a) > 3 else 4
    dc = float(a[4]) if len(a) > 4 else 1.5
    T = float(a[5]) if len(a) > 5 else 1.5
    # the C3 grid (0.005 cm-1, wnosamp 24, cutoff 25) on a shorter range; layers spread over the
    # whole atmosphere so that Doppler cores and 100-bar Lorentz wings are both present
    case = synth.lbl_case(nwave, nlayers, nlines, wnstep=0.005, niso=4, seed=42)
    g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
    osamp = g['wnosamp']
    assert (nwave - 1) % s == 0, 'nwave - 1 must be a multiple of the stride'
    vt = eng.VoigtTable.build(vg['lorentz'], vg['doppler'], vg['size'], g['ownstep'], osamp, True)
    profile, size, index = vt.flat(), np.array(vt.size), np.array(vt.index)
    ll = eng.LineList(ln['lwn'], ln['elow'], ln['gf'], ln['lid'], len(iso['isomass']), g['own'])
    t, d, z = eng.dev(atm['temp']), eng.dev(atm['dens']), eng.dev(iso['isoz'])

    def plan(table, wn, divisors, cutoff):
        return eng.LBL(table, ll, wn, divisors, atm['mol_radius'], atm['mol_mass'], iso['isoimol'],
                       iso['isomass'], iso['isoratio'], iso['isoiext'], cutoff, case['ethresh'],
                       max_layers=nlayers)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) / reps

    exact_plan = plan(vt, g['wn'], g['divisors'], vg['cutoff'])
    exact, t_exact = timed(lambda: exact_plan.extinction(t, d, z))
    exact = exact[:, 0]
    core_p, wing_p = split_tables(profile, size, index, g['ownstep'], dc, T)
    vt_core = eng.VoigtTable.from_flat(core_p, size, index, vg['lorentz'], vg['doppler'], osamp)
    vt_wing = eng.VoigtTable.from_flat(wing_p, size, index, vg['lorentz'], vg['doppler'], osamp * s)
    core_plan = plan(vt_core, g['wn'], g['divisors'], dc + T)
    wing_plan = plan(vt_wing, g['wn'][::s].copy(), synth.divisors(osamp * s), vg['cutoff'])
    core, t_core = timed(lambda: core_plan.extinction(t, d, z))
    wing, t_wing = timed(lambda: wing_plan.extinction(t, d, z))
    both, t_interp = timed(lambda: core[:, 0] + lagrange4(wing[:, 0], s, nwave))
    err = ((both - exact).abs() / exact.abs().clamp_min(1e-300))
    # the same split WITHOUT the sub-grid (s = 1): what the partition alone costs in accuracy
    wing1_plan = plan(eng.VoigtTable.from_flat(wing_p, size, index, vg['lorentz'], vg['doppler'],
                                               osamp), g['wn'], g['divisors'], vg['cutoff'])
    full_split = core[:, 0] + wing1_plan.extinction(t, d, z)[:, 0]
    err1 = ((full_split - exact).abs() / exact.abs().clamp_min(1e-300))
    # spectra: transit and emission from both extinctions
    rad = eng.dev(atm['radius'])
    path = eng.dev(eng.pack_raypath(eng.transit_path(atm['radius'], 0), 0))
    sp = {}
    for name, ec in (('exact', exact), ('wings', both)):
        ec = ec.contiguous()
        tr = eng.transit_spectrum(ec, path, rad, float(atm['rstar']), 0, nlayers, 10.0)[0]
        dep, ideep = eng.plane_parallel_optical_depth(ec, eng.dev(-np.diff(atm['radius'])), 0,
                                                      nlayers, 10.0)
        mu, wts = eng.default_quadrature()
        em = eng.emission_flux(dep, ideep, eng.dev(g['wn']), t, eng.dev(mu), eng.dev(wts), 0)
        sp[name] = (tr.clone(), em.clone())
    e_tr = ((sp['wings'][0] - sp['exact'][0]).abs() / sp['exact'][0].abs()).max().item()
    e_em = ((sp['wings'][1] - sp['exact'][1]).abs() / sp['exact'][1].abs()).max().item()
    print(f'grid {nwave} x {nlayers} layers, {nlines} lines, stride {s}, core +-{dc} cm-1, '
          f'transition {T} cm-1 (kernels: exact {exact_plan.last_gather_kernel}, core '
          f'{core_plan.last_gather_kernel}, wings {wing_plan.last_gather_kernel})')
    print(f'  extinction stage: exact {1e3 * t_exact:.2f} ms; core {1e3 * t_core:.2f} + wings '
          f'{1e3 * t_wing:.2f} + interpolation {1e3 * t_interp:.2f} ms = '
          f'{1e3 * (t_core + t_wing + t_interp):.2f} ms ({t_exact / (t_core + t_wing + t_interp):.2f} x)')
    per_layer = err.max(dim=1).values.cpu().numpy()
    print(f'  ec rel. error: max {err.max().item():.2e}, median of the layer maxima '
          f'{np.median(per_layer):.2e}; the partition alone (s = 1): {err1.max().item():.2e}')
    print('  per layer:', ' '.join(f'{v:.1e}' for v in per_layer))
    print(f'  spectrum rel. error: transit {e_tr:.2e}, emission {e_em:.2e}')


if __name__ == '__main__':
    main()
