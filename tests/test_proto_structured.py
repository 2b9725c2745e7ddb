"""The numpy model of the kernel's structured dual active-set solver (tools/proto_structured.py)
against the dense oracle: documents that slack elimination + capped multipliers solve the same QP."""
import os
import sys

import numpy as np
import pytest

from conftest import make_case, load_golden, golden_step_inputs, ROOT
from oracle import Oracle

sys.path.insert(0, os.path.join(ROOT, "tools"))
from proto_structured import solve_ab_step  # noqa: E402


@pytest.mark.parametrize("tree,N", [("ABO", 20), ("ORIG", 20), ("ABO", 30)])
def test_structured_solver_equals_dense_qp(tree, N):
    OPT, V, s_tv, v_tv = make_case(tree, N)
    G = load_golden(f"{tree.lower()}_abmpc")
    orc = Oracle(OPT, V)
    for k in (0, 148, 222, 555, 740):
        inp = golden_step_inputs(G, s_tv, v_tv, k)
        r = orc.ab_step(**inp, want_dense=True)
        prob, qp, st, xi = solve_ab_step(OPT, V, inp["s"], inp["v"], inp["a_prev"], inp["t0"], inp["s_tv"],
                                         inp["v_tv"], inp["a_tv_prev"])
        assert st == 0
        a_ref = r["x"][0::5]
        scale = 1.0 + np.abs(a_ref).max()
        assert np.abs(qp.a - a_ref).max() < 1e-8 * scale
        xs = np.array([[xi[(g, kk)] for g in "vhsf"] for kk in range(N)])
        xr = np.stack([r["x"][1 + i::5] for i in range(4)], 1)
        assert np.abs(xs - xr).max() < 1e-8 * (1.0 + np.abs(xr).max())
