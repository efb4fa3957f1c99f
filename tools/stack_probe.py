"""VERDICT round 3, item 4 ("two spectra per launch"): a middle 1/8 shard of a workload, DEPTH
contexts in flight, no collectives -- one atmosphere per launch (LBLSpectrum as dist.ShardPipeline
runs it) against STACK atmospheres stacked as STACK x L layers of ONE two-phase extinction call
(layer state, records, gather, combine once) + STACK transit calls.  A probe: the stacked form is
not a mode of the pipeline.  usage: STACK=2 DEPTH=3 python tools/stack_probe.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np
import torch
import bench
from pyratbay_amd import engine, dist as pbdist

world = int(os.environ.get('WORLD', '8'))
r, depth = world // 2, int(os.environ.get('DEPTH', '3'))
K = int(os.environ.get('STACK', '2'))
case = bench.make_case(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'c2'])
g, atm, ln, iso, vg = (case[k] for k in ('grid', 'atm', 'lines', 'iso', 'voigt'))
nwave, L = g['nwave'], atm['nlayers']
b = pbdist.uniform_bounds(nwave, world) if world > 1 else np.array([0, nwave])
w0, wc = int(b[r]), int(b[r + 1] - b[r])
streams = engine.side_streams(depth)

def timed(submit, per, n=300):
    n = max(20, n // (1 if world > 1 else 4))
    for _ in range(max(10, n // 5)): submit()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): submit()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / (n * per)

# --- one atmosphere per launch
first = engine.LBLSpectrum(case, wbegin=w0, wcount=wc, timestamps=False)
models = [first] + [engine.LBLSpectrum(case, wbegin=w0, wcount=wc, voigt=first.voigt, lines=first.lines, timestamps=False) for _ in range(depth - 1)]
for m in models:
    m.lbl.set_concurrency(depth)
    m.kmax_exchange = lambda t: None
cnt = [0]
def submit1():
    j = cnt[0] % depth; cnt[0] += 1
    with torch.cuda.stream(streams[j]):
        models[j].run()
t1 = timed(submit1, 1)
ref = models[0].run().clone()
print('one atmosphere per launch   %.4f ms per spectrum' % t1, flush=True)

# --- two atmospheres stacked
class Pair:
    def __init__(self):
        self.lbl = engine.LBL(first.voigt, first.lines, g['wn'], g['divisors'], atm['mol_radius'], atm['mol_mass'],
                              iso['isoimol'], iso['isomass'], iso['isoratio'], iso['isoiext'], vg['cutoff'],
                              case['ethresh'], max_layers=K * L)
        self.lbl.set_concurrency(depth)
        self.temp = engine.dev(np.concatenate([atm['temp']] * K))
        self.dens = engine.dev(np.concatenate([atm['dens']] * K))
        self.isoz = engine.dev(np.concatenate([iso['isoz']] * K, axis=1))
        self.ec = torch.empty((K * L, 1, wc), dtype=torch.float64, device='cuda')
        self.out = [torch.empty(wc, dtype=torch.float64, device='cuda') for _ in range(K)]
    def run(self):
        self.lbl.extinction_begin(self.temp, self.dens, self.isoz, add=True, out=self.ec, wbegin=w0, wcount=wc)
        self.lbl.extinction_end()
        for h in range(K):
            engine.transit_spectrum(self.ec[h * L:(h + 1) * L].view(L, wc), first.raypath, first.radius,
                                    first.rstar, 0, L, first.maxdepth, out=self.out[h])
        return self.out
pairs = [Pair() for _ in range(depth)]
def submit2():
    j = cnt[0] % depth; cnt[0] += 1
    with torch.cuda.stream(streams[j]):
        pairs[j].run()
t2 = timed(submit2, K, 300 // K)
o = pairs[0].run()
torch.cuda.synchronize()
print(f'{K} atmospheres per launch, {depth} in flight  ' + '%.4f ms per spectrum   (max rel diff %.2e)' %
      (t2, float(((o[0] - ref).abs() / ref.abs()).max())), flush=True)
t1b = timed(submit1, 1)
print('one atmosphere per launch   %.4f ms per spectrum (again)' % t1b)
