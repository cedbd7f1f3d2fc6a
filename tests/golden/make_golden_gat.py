"""Golden vectors for the GAT forward pass (gat.hpp), produced by the REFERENCE's own gat.hpp through
oracle/_ref/ref_driver gat ... (weights and alpha installed through public members; the reference itself leaves
them zero / uninitialised).  Only schedules that do not split R are meaningful (gat.hpp:88 multiplies local column
slices).  `quirk_fusion2_p4_c2` records what the reference computes for local-kernel-fusion with c > 1, where the
SpMM accumulates on top of the gathered SDDMM operand (15D_dense_shift.hpp:306-314 is skipped by
initial_replicate = false, gat.hpp:100) — kept to show that the mirror reproduces the reference bit for bit there too.
Run where /root/reference exists:  python tests/golden/make_golden_gat.py"""
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


def main():
    case = T.case_inputs("er8_r16")
    x = case["A"] * T.GAT_INPUT_SCALE
    args = (case["M"], case["rows"], case["cols"], case["R"], x)
    canon = RR.gat(*args, "15d_fusion1", 1, 1, T.GAT_ALPHA, T.GAT_LAYERS)
    dev = {}
    for alg, p, c in [("15d_fusion1", 4, 1), ("15d_fusion1", 4, 2), ("15d_fusion1", 8, 2), ("15d_fusion2", 1, 1), ("15d_fusion2", 4, 1)]:
        res = RR.gat(*args, alg, p, c, T.GAT_ALPHA, T.GAT_LAYERS)
        dev["%s p%d c%d" % (alg, p, c)] = T.rel(res, canon)
        print(alg, p, c, dev["%s p%d c%d" % (alg, p, c)], flush=True)
    quirk = RR.gat(*args, "15d_fusion2", 4, 2, T.GAT_ALPHA, T.GAT_LAYERS)
    print("fusion2 p4 c2 vs canonical:", T.rel(quirk, canon))
    np.savez_compressed(os.path.join(HERE, "gat_er8_r16.npz"), out=canon, quirk_fusion2_p4_c2=quirk)
    with open(os.path.join(HERE, "gat_manifest.json"), "w") as f:
        json.dump({"layers": T.GAT_LAYERS, "alpha": T.GAT_ALPHA, "input_scale": T.GAT_INPUT_SCALE,
                   "deviation_from_canonical": dev, "fusion2_p4_c2_vs_canonical": T.rel(quirk, canon)}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
