"""Golden vectors for the ALS-CG application (BASELINE config 5), produced by the REFERENCE's own
als_conjugate_gradients.cpp (oracle/_ref/ref_driver als ...): ground truth = the case's S values, embeddings
initialised from the case's A and B, `steps` x {cg_optimizer(Amat, iters); cg_optimizer(Bmat, iters)}.
Run where /root/reference exists:  python tests/golden/make_golden_als.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hnh_testlib as T  # noqa: E402
from oracle import refrun as RR  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
STEPS, ITERS = 1, 5
VARIANTS = [("15d_fusion1", 1, 1), ("15d_fusion2", 1, 1), ("15d_fusion2", 4, 2), ("15d_fusion1", 4, 1), ("15d_sparse", 4, 1),
            ("25d_dense_replicate", 4, 1), ("25d_sparse_replicate", 8, 2)]


def main():
    manifest = {}
    for name in ("er8_r16", "ragged_r8"):
        case = T.case_inputs(name)
        args = (case["M"], case["N"], case["rows"], case["cols"], case["vals"], case["R"], case["A"], case["B"])
        canon = RR.als(*args, "15d_fusion1", 1, 1, STEPS, ITERS)
        np.savez_compressed(os.path.join(HERE, "als_%s.npz" % name), A=canon["A"], B=canon["B"], residuals=canon["residuals"])
        dev = {}
        for alg, p, c in VARIANTS:
            if not T.valid_config(alg, p, c, case["R"]):
                continue
            res = RR.als(*args, alg, p, c, STEPS, ITERS)
            dev["%s p%d c%d" % (alg, p, c)] = {"A": T.rel(res["A"], canon["A"]), "B": T.rel(res["B"], canon["B"]),
                                              "residuals": T.rel(res["residuals"], canon["residuals"])}
            print(name, alg, p, c, dev["%s p%d c%d" % (alg, p, c)], flush=True)
        manifest[name] = {"steps": STEPS, "cg_iters": ITERS, "residuals": canon["residuals"].tolist(), "deviation_from_canonical": dev}
    path = os.path.join(HERE, "als_manifest.json")
    try:  # (what the asserting tests observed against these vectors — HNH_OBSERVED_LOG, see tests/hnh_testlib.py — is kept across regeneration)
        with open(path) as f:
            old = json.load(f)
        if "observed_vs_reference" in old:
            manifest["observed_vs_reference"] = old["observed_vs_reference"]
    except (OSError, ValueError):
        pass
    with open(path, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
