#!/usr/bin/env python3
"""Fixture G20: `Pyrat.get_ec(layer)` of the real package for the line-by-line model
(pyrat_obj.py:700-719 -> pyrat/opacity.py:260-305 -> line_by_line.py:224-230 ->
extinction.py:170-213 with add = False): the per-species extinction (cm-1) of single layers of
the G6 transit run.  Build container only:

    python tests/golden/make_golden_get_ec.py

The run is make_golden_e2e.py's transit run (same TLI file, same configuration: checked against
g6_e2e_transit.npz here), so the fixture holds only what get_ec returned; the inputs are G6's."""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_e2e as e2e                      # noqa: E402

LAYERS = (3, 24, 47)


def main():
    if not os.path.isdir(e2e.REF):
        sys.exit('needs /root/reference')
    work = tempfile.mkdtemp(prefix='pb_getec_')
    try:
        pb = e2e.reference_package(work)

        def run(cfg_text, name, **kw):
            cfg = os.path.join(work, name + '.cfg')
            with open(cfg, 'w') as f:
                f.write(cfg_text.format(work=work, ref=e2e.REF, **kw))
            return pb.run(cfg)

        run(e2e.CFG_TLI, 'tli')
        pyrat = run(e2e.CFG_SPEC, 'spec_transit', rt='transit')
        g6 = np.load(os.path.join(HERE, 'g6_e2e_transit.npz'))
        lbl = pyrat.opacity.models[pyrat.opacity.models_type.index('lbl')]
        assert np.array_equal(lbl.ec, g6['ec']) and np.array_equal(pyrat.atm.d, g6['dens'])
        store = dict(layers=np.array(LAYERS))
        for layer in LAYERS:
            ec, labels = pyrat.get_ec(layer)
            store[f'ec_{layer}'] = ec
            store[f'labels_{layer}'] = np.array(labels)
            print(layer, labels, ec.shape, float(ec.max()))
            # (the run's own total at that layer is the sum over the species)
            np.testing.assert_allclose(ec.sum(axis=0), lbl.ec[layer], rtol=1e-12)
        store['species'] = np.array(list(lbl.species))
        np.savez_compressed(os.path.join(HERE, 'g20_get_ec.npz'), **store)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
