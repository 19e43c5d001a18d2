"""Generates the committed fixtures under tests/golden/.  Run in the build container (needs
/root/reference for the real bamg):  python tests/golden/make_golden.py

  bamg_connectivity.npz  REAL contrib/bamg BamgConvertMeshx output on the seeded 'tiny' mesh
                         (inputs + both tables)  -- pins nxs_mesh_connectivity and the oracle.
  oracle_tiny.npz        oracle (liboracle.so) state on the 'tiny' toy case after 1 sub-step, 1 step
                         and 3 steps (beyond a few steps the algorithm amplifies 1-ulp differences to O(1), see
                         tests/test_oracle_sensitivity.py): a regression net for the oracle itself and size-0 cost
                         expected values for the GPU path.  NOT reference output: the reference's
                         model/ cannot be built here.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from nextsim_amd import mesh as M  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

KEYS = ("VT", "UM", "UT", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick", "snow_thick", "ridge_ratio",
        "conc_young", "h_young", "hs_young", "conc_myi", "thick_myi")


def main():
    lm = M.localize(cases.global_mesh("tiny"), 1)[0]
    assert O.bamg_shim() is not None, "build oracle/_ref first (make -C oracle ref)"
    nec, nc = O.bamg_connectivity(lm.indices, lm.coord_x, lm.coord_y)
    np.savez_compressed(os.path.join(HERE, "bamg_connectivity.npz"), indices=lm.indices, num_nodes=lm.num_nodes,
                        x=lm.coord_x, y=lm.coord_y, nec=nec, nc=nc)

    out = {}
    for tag, nsteps, over in (("sub1", 1, dict(substeps=1, dtime_step=200. / 120.)), ("step1", 1, {}), ("step3", 3, {})):
        gm, p, g, lms, fields = cases.make_case("tiny", **over)
        r = O.OracleRank(lms[0], p, fields[0])
        for _ in range(nsteps):
            r.step()
        for k in KEYS:
            out[f"{tag}_{k}"] = r.arr[k]
    np.savez_compressed(os.path.join(HERE, "oracle_tiny.npz"), **out)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
