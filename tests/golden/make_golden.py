#!/usr/bin/env python3
"""Regenerates tests/golden/solves.json from the CPU oracle.

The reference ships no golden outputs for the AL path and cannot be built here
(SURVEY.md 8(c)), so these fixtures are the build's own: iteration counts and
residual histories of the oracle on the seeded cases of tests/cases.py.  They
pin the oracle against regressions and give the GPU tests committed expected
values.  Floats are stored as hex strings (exact)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    out = {}
    for name in cases.ALL_CASES:
        pb, cfg = cases.case(name)
        osys = cases.oracle_system(pb, cfg)
        rhs = cases.prepared_rhs(osys, pb, cfg)
        rc, x, res, hist = osys.solve(cfg, rhs)
        assert rc == 0, (name, rc)
        out[name] = {
            "block_sizes": pb.block_sizes,
            "outer_iterations": res.outer_iterations,
            "inner_iterations": res.inner_iterations,
            "mp_iterations": res.mp_iterations,
            "rational_iterations": res.rational_iterations,
            "mass_iterations": res.mass_iterations,
            "lambda_max": float(res.lambda_max).hex(),
            "history": [float(h).hex() for h in hist],
            "x_block_norms": [float((b * b).sum() ** 0.5).hex() for b in x],
        }
        print(name, pb.block_sizes, "outer", res.outer_iterations, "inner", res.inner_iterations)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "solves.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
