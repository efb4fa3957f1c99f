#!/bin/bash
# usage: tools/fuzz_all.sh [scale] -- the randomised parity campaign in one GPU call (every sweep
# against the oracle / the compiled reference under oracle/_ref); prints each sweep's last lines.
# scale 1 is ~6 minutes on one MI355X box.
s=${1:-1}
run() { echo "== $*"; timeout -k 10 1000 python "$@" 2>&1 | grep -v amdgpu.ids | tail -2; echo "exit ${PIPESTATUS[0]}"; }
run tools/fuzz_sweep.py ${FUZZ_SEED0:-20000} $((2500 * s))
run tools/fuzz_ext.py $((4000 * s))
run tools/fuzz_more.py $((4000 * s))
run tools/fuzz_pipeline.py $((1200 * s)) 7000
run tools/fuzz_table.py $((2000 * s))
run tools/fuzz_rt.py $((2000 * s))
run tools/fuzz_continuum.py $((3000 * s))
run tools/fuzz_dropin.py $((8000 * s))
run tools/fuzz_opacity.py $((2000 * s))
run tools/fuzz_r3.py $((3000 * s))
run tools/fuzz_resdyn.py $((3000 * s))
run tools/fuzz_r4.py $((400 * s))
run tools/fuzz_r5.py $((300 * s))
