"""A/B of who requests the row DMA in k_ext_staged (experiments build: PB_STAGE_PROBE=11|12 select
kIssue = 1|2, see the kernel's comment): gather-kernel time per launch and a hash of ec (the
arithmetic does not depend on who moves the row: the bits must be equal).

    PB_LIBPBHIP=$PWD/pyratbay_amd/libpbhip_exp.so python tools/issue_ab.py [workload ...]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(name, steps):
    import torch
    import bench
    from pyratbay_amd import engine
    w = bench.WORKLOADS[name]
    model = engine.LBLSpectrum(bench.make_case(w), rt_path=w.get('rt_path', 'transit'))
    model.lbl.set_gather_mode('staged')
    for _ in range(3):
        model.extinction()
    torch.cuda.synchronize()
    model.lbl.timing_begin(steps)
    for _ in range(steps):
        model.extinction()
    torch.cuda.synchronize()
    ms, n = model.lbl.timing_end()
    h = hashlib.sha256(model.ec.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({'gather_ms': ms / max(n, 1), 'ec_sha': h}), flush=True)


if __name__ == '__main__':
    if sys.argv[1:2] == ['--child']:
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    for name in sys.argv[1:] or ['c2']:
        steps = 20 if 'c2' in name else 3
        base = None
        for rep in range(2):
            for mode, label in ((None, 'one requester (default)'), ('11', 'one requester, descriptor a turn ahead'),
                                ('12', 'every wavefront its slice')):
                env = dict(os.environ)
                env.pop('PB_STAGE_PROBE', None)
                if mode:
                    env['PB_STAGE_PROBE'] = mode
                p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', name, str(steps)],
                                   env=env, capture_output=True, text=True, cwd=ROOT)
                if p.returncode:
                    print(name, label, 'FAILED', p.stderr[-500:])
                    continue
                d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
                base = base or d['ec_sha']
                print(f"{name:8s} {label:45s} {d['gather_ms']:.4f} ms  ec {'==' if d['ec_sha'] == base else '!='} default",
                      flush=True)
