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
    bad, ratios, fr_big, fr_small, b_scenes = 0, [], [], [], {}
    orc.build()
    for s in range(seed, seed + seeds):
        rng = np.random.default_rng(s)
        for it in range(n):
            info = {}
            try:
                line = stress_case(orc, rng, verbose=bool(verbose), info=info)
            except AssertionError as e:
                bad += 1
                line = "FAIL " + str(e)[:2500]
            rs = {k: round(v["ratio"], 2) for k, v in info.get("branch_b", {}).items()}
            ratios += list(rs.values())
            if info.get("branch_b"):
                b_scenes[s] = b_scenes.get(s, 0) + 1
                frac = max(v["unpinned_rows"] / max(v["visible_rows"], 1) for v in info["branch_b"].values())
                (fr_big if info.get("visible", 0) >= 1000 else fr_small).append(round(frac, 3))
            print(f"[{s}/{it}] {line[:2600] if line.startswith('FAIL') else line[:260]}" + (f" || branch (b) ratios device / max(oracle, noise): {rs}" if rs else ""), flush=True)
    ratios.sort()
    if ratios:
        print(f"branch (b) ratios over {len(ratios)} (scene, tensor) pairs: median {ratios[len(ratios) // 2]:.2f}, p90 {ratios[int(len(ratios) * 0.9)]:.2f}, "
              f"max {ratios[-1]:.2f}")
    fr_big.sort(); fr_small.sort()
    print(f"branch (b) scenes per seed (of {n}): max {max(b_scenes.values(), default=0)}; fraction of visible rows binary32 does not pin, scenes with >= 1000 visible Gaussians: "
          f"{len(fr_big)} scenes, median {fr_big[len(fr_big) // 2] if fr_big else 0}, max {fr_big[-1] if fr_big else 0}; smaller scenes: {len(fr_small)}, "
          f"median {fr_small[len(fr_small) // 2] if fr_small else 0}, max {fr_small[-1] if fr_small else 0}")
    print("failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:])) else 0)
