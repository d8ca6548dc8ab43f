#!/usr/bin/env python3
"""Generates tests/golden/srbd_qp_golden.npz with THIS repo's fp64 oracle (oracle/srbd_oracle.py).

The reference's implementation of the path is an absent submodule and the reference holds no fixtures (see the
oracle header), so these vectors pin the build's own specification: inputs, the assembled QP, the exact QP optimum
(independent active-set KKT solve, residuals <= 1e-9) and the ADMM twin's iterate.  Re-run only when the
specification (DESIGN.md "Problem specification") changes:   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import srbd_oracle as orc  # noqa: E402

CASES = [  # name, N, schedule, seed, index in the seeded batch
    ("n10_single_a", 10, "single", 4242, 0),
    ("n10_single_b", 10, "single", 4242, 1),
    ("n10_double", 10, "double", 4243, 0),
    ("n10_mixed", 10, "mixed", 4244, 2),
    ("n8_mixed", 8, "mixed", 4245, 0),
    ("n4_single", 4, "single", 4246, 1),
]


def main():
    p = orc.SrbdParams()
    out = {"params_json": np.array(repr(sorted(p.as_dict().items())))}
    for name, N, sched, seed, idx in CASES:
        x0, xr, ft, ct = orc.synthetic_batch(4, N, seed, sched)
        x0, xr, ft, ct = x0[idx], xr[idx], ft[idx], ct[idx]
        qp = orc.build_qp(p, x0, xr, ft, ct)
        xs, ys = orc.solve_reference(p, qp)
        kr = orc.kkt_residuals(qp["P"], qp["q"], qp["A"], qp["l"], qp["u"], xs, ys)
        assert max(kr.values()) < 1e-8, (name, kr)
        tw = orc.update(p, x0, xr, ft, ct)
        tw_full = orc.update(orc.SrbdParams(eliminate_swing=False), x0, xr, ft, ct)
        out[f"{name}/x0"] = x0; out[f"{name}/x_ref"] = xr; out[f"{name}/foot"] = ft; out[f"{name}/contact"] = ct
        out[f"{name}/q"] = qp["q"]; out[f"{name}/l"] = qp["l"]; out[f"{name}/u"] = qp["u"]
        if name in ("n10_single_a", "n4_single"):
            out[f"{name}/P"] = qp["P"]
        else:
            out[f"{name}/P_diag"] = np.diag(qp["P"]).copy(); out[f"{name}/P_rowsum"] = qp["P"].sum(1)
        out[f"{name}/u_exact"] = (xs * p.force_scale).reshape(N, 12)
        out[f"{name}/y_exact"] = ys
        out[f"{name}/x_exact"] = orc.rollout(qp, x0, xs, p.force_scale)
        out[f"{name}/u_admm"] = tw["u"]; out[f"{name}/iters_admm"] = np.int32(tw["iters"])
        out[f"{name}/u_admm_full"] = tw_full["u"]; out[f"{name}/iters_admm_full"] = np.int32(tw_full["iters"])
        print(name, "iters", tw["iters"], tw_full["iters"], "max|u_admm-u_exact|", np.abs(tw["u"] - out[f"{name}/u_exact"]).max())
    np.savez_compressed(os.path.join(HERE, "srbd_qp_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "srbd_qp_golden.npz"))


if __name__ == "__main__":
    main()
