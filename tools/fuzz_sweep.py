"""One-off extended sweep of the randomly drawn parity cases (the committed tests keep 10 + 8
seeds): extinction configurations and column problems for seeds beyond the committed ones.
usage: python tools/fuzz_sweep.py [first] [count]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


class _Patch:
    """the two methods of pytest's monkeypatch the tests use"""
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k):
        os.environ.pop(k, None)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    from oracle import oracle
    oracle.lib()
    from pyratbay_amd import engine
    engine.require_gpu()
    import test_gpu_extinction as te
    import test_gpu_columns as tc
    # the tests derive their generators from fixed offsets + seed: any seed is a new case
    bad = []
    for seed in range(first, first + count):
        for name, fn, args in (('extinction', te.test_random_configurations, (engine, oracle, seed)),
                               ('columns', tc.test_random_rt_configurations, (engine, oracle, seed))):
            try:
                fn(*args)
            except Exception:                              # noqa: BLE001
                bad.append((name, seed))
                print(f'FAIL {name} seed {seed}')
                traceback.print_exc(limit=3)
        if (seed - first) % 10 == 9:
            print(f'seeds {first}..{seed} done, {len(bad)} failures', flush=True)
    print('failures:', bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
