"""Developer stress run (GPU box): more seeds of the randomised scene set that tests/test_gpu_parity.py::
test_randomised_stress_scenes runs under pytest -m gpu (same scene generator, same bars).
usage: python tools/stress_parity.py [scenes per seed] [first seed] [number of seeds] [verbose: 1 = print the
fp64 evidence of every tensor]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import raster_oracle as orc
from test_gpu_parity import stress_case


def main(n=40, seed=0, seeds=1, verbose=0):
    bad = 0
    orc.build()
    for s in range(seed, seed + seeds):
        rng = np.random.default_rng(s)
        for it in range(n):
            try:
                line = stress_case(orc, rng, verbose=bool(verbose))
            except AssertionError as e:
                bad += 1
                line = "FAIL " + str(e)[:300]
            print(f"[{s}/{it}] {line}", flush=True)
    print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:])) else 0)
