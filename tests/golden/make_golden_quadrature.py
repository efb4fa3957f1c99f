"""Fixture G18: what the reference computes for `quadrature = n`, n = 1 ... 16
(pyratbay/pyrat/spectrum.py:41-49): SciPy's Gauss-Legendre nodes and weights
(scipy.special.p_roots, SciPy 1.15.3 in the build container) and the quadrature_mu / weights the
Spectrum object derives from them.  Numbers only.

    python tests/golden/make_golden_quadrature.py
"""
import os

import numpy as np
import scipy
import scipy.special as ss

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'g18_p_roots.npz')


def main():
    out = {'scipy_version': np.array(scipy.__version__)}
    for n in range(1, 17):
        qnodes, qweights = ss.p_roots(n)
        out[f'nodes_{n}'], out[f'qweights_{n}'] = qnodes, qweights
        qnodes = 0.5 * (qnodes + 1.0)
        out[f'mu_{n}'] = np.sqrt(qnodes)
        out[f'weights_{n}'] = 0.5 * np.pi * qweights
    np.savez(OUT, **out)
    print(OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
