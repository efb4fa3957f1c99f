#!/usr/bin/env python3
"""Fixture G14: log-likelihoods from the reference's own `Loglike.__call__`
(pyratbay/tools/retrieval_tools.py:73-104) for a batch of band-integrated models, including a
rejected model (non-finite band flux -> -1e98).  Build container only.

    python tests/golden/make_golden_loglike.py

The Loglike object is created without its constructor (which needs a whole Pyrat run) and
given exactly the attributes __call__ reads: data, uncert, params, ifree, ishare, pstep, func.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_e2e import reference_package, REF      # noqa: E402


def main():
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference')
    work = tempfile.mkdtemp(prefix='pb_ll_')
    try:
        reference_package(work)
        from pyratbay.tools.retrieval_tools import Loglike
        rng = np.random.default_rng(14)
        nw, nb = 9, 57
        data = rng.uniform(1e-3, 2e-3, nb)
        uncert = rng.uniform(1e-5, 5e-5, nb)
        models = data[None, :] * (1 + rng.normal(0, 0.02, (nw, nb)))
        models[3, 10] = np.inf                   # eval()'s reject value
        models[6, 0] = np.nan
        models[7] = data                         # perfect fit: the normalisation term alone
        ll = object.__new__(Loglike)
        ll.data, ll.uncert = data, uncert
        ll.params = np.zeros(3)
        ll.ifree = np.array([0, 1, 2])
        ll.ishare = np.array([], int)
        ll.pstep = np.ones(3)
        ll._dt_snapshot = 0.0
        out = np.zeros(nw)
        for w in range(nw):
            ll.func = lambda params, retmodel=False, w=w: models[w]
            out[w] = ll(np.array([0.1 * w, 0.2, 0.3]))
        np.savez_compressed(os.path.join(HERE, 'g14_loglike.npz'), data=data, uncert=uncert,
                            models=models, loglike=out)
        print(out)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
