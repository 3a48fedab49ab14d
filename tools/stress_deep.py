"""Developer stress run (GPU box): the randomised stress scenes with the deep-tile-list variants forced
(scr_debug_force_deep_lists: no gm_index array, per-Gaussian record flags).  usage: python tools/stress_deep.py [scenes per seed] [first seed] [seeds]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import raster_oracle as orc
from splatco_amd import _C
from test_gpu_parity import stress_case


def main(n=10, seed=20, seeds=4):
    orc.build()
    _C.check(_C.lib.scr_debug_force_deep_lists(1))
    bad = 0
    for s in range(seed, seed + seeds):
        rng = np.random.default_rng(s)
        for it in range(n):
            try:
                line = stress_case(orc, rng)
            except AssertionError as e:
                bad += 1
                line = "FAIL " + str(e)[:300]
            print(f"[deep {s}/{it}] {line[:200]}", flush=True)
    print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:])) else 0)
