"""Generates tests/golden/*.npz by running the REFERENCE ITSELF (oracle/_ref/ref_driver = the
unmodified /root/reference sources + real MKL/MPICH, built by oracle/build_ref.sh) on seeded inputs.

Run here (the container that has /root/reference):   python tests/golden/make_golden.py
The GPU box has no /root/reference; tests only read the committed .npz / .json files.

Per case the canonical outputs come from `15d_fusion1` at p = 1 (plus `15d_fusion2` at p = 1 for the
local-kernel-fusion fusedSpMM, which ignores Svalues — SURVEY Appendix C #4).  Every other
(algorithm, p, c) is executed too and its maximum relative deviation from the canonical output is
recorded in manifest.json: distribution-independence of the results is the reference's own
correctness criterion (scratch.cpp:26-76).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O, refrun as RR  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (M, N, draws, R, seed)
    "er8_r16": (256, 256, 256 * 8, 16, 12345),          # square, powers of two
    "ragged_r8": (250, 250, 1500, 8, 777),             # M not divisible by p: padded row blocks
    "rect_r16": (192, 320, 2400, 16, 4242),            # non-square S
    "tiny_r8": (64, 64, 9, 8, 99),                     # almost empty: most blocks have no nonzeros
}
GRIDS = {
    "15d_fusion1": [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 2)],
    "15d_fusion2": [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (4, 4), (8, 2)],
    "15d_sparse": [(1, 1), (2, 1), (2, 2), (4, 1), (4, 2), (8, 2)],
    "25d_dense_replicate": [(1, 1), (2, 2), (4, 1), (8, 2)],
    "25d_sparse_replicate": [(1, 1), (2, 2), (4, 1), (8, 2)],
}
DENSE = ("spmmA", "spmmB", "fusedA", "fusedB")
SPARSE = ("sddmmA", "sddmmB", "fusedA_buf", "fusedB_buf")


def inputs(name):
    m, n, draws, r, seed = CASES[name]
    rows, cols = O.erdos_renyi_mn(m, n, draws, seed)
    vals = O.sparse_values(rows, cols, n, seed + 1)
    return m, n, r, rows, cols, vals, O.dense_fill(m, r, seed + 2), O.dense_fill(n, r, seed + 3)


def rel(x, y):
    return float(np.max(np.abs(x - y)) / max(float(np.max(np.abs(y))), 1e-300)) if x.size else 0.0


def main():
    manifest = {}
    for name in CASES:
        m, n, r, rows, cols, vals, a, b = inputs(name)
        keys = rows * n + cols
        canon = RR.dump(m, n, rows, cols, vals, r, a, b, "15d_fusion1", 1, 1)
        canon2 = RR.dump(m, n, rows, cols, vals, r, a, b, "15d_fusion2", 1, 1)
        for s in SPARSE:
            assert np.array_equal(canon[s][0], keys), "coordinate probe disagrees with the input"
        out = {k: canon[k] for k in DENSE}
        out.update({k: canon[k][1] for k in SPARSE})
        out["fusedA_fusion2"] = canon2["fusedA"]
        out["fusedB_fusion2"] = canon2["fusedB"]
        fp = RR.fingerprints(m, n, rows, cols, r, "15d_fusion1", 1, 1)
        out["fingerprints"] = np.array([fp["sddmm"], fp["spmmA"], fp["spmmB"]])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        dev = {}
        for alg, grids in GRIDS.items():
            for p, c in grids:
                if alg == "15d_sparse" and m != n:
                    continue  # reference quirk: block column count assumes square S (15D_sparse_shift.hpp:132)
                sq = int(round((p // c) ** 0.5))
                if alg.startswith("25d") and r % (sq * (c if alg == "25d_sparse_replicate" else 1)) != 0:
                    continue
                if alg == "15d_sparse" and r % (p // c) != 0:
                    continue
                try:
                    res = RR.dump(m, n, rows, cols, vals, r, a, b, alg, p, c)
                except Exception as e:  # record, do not hide
                    dev["%s p%d c%d" % (alg, p, c)] = "FAILED: %s" % str(e)[:200]
                    continue
                d = {}
                for k in DENSE:
                    ref = out[k + "_fusion2"] if (alg == "15d_fusion2" and k.startswith("fused")) else out[k]
                    d[k] = rel(res[k], ref)
                for k in SPARSE:
                    if alg == "15d_fusion2" and k.endswith("_buf"):
                        d[k] = "not filled by the reference (15D_dense_shift.hpp:250-251)"
                        continue
                    d[k] = rel(res[k][1], out[k]) if np.array_equal(res[k][0], keys) else "KEY MISMATCH"
                fpv = RR.fingerprints(m, n, rows, cols, r, alg, p, c)
                d["fingerprint_rel"] = max(rel(np.array([fpv["sddmm"], fpv["spmmA"], fpv["spmmB"]])[i:i + 1],
                                               out["fingerprints"][i:i + 1]) for i in range(3))
                dev["%s p%d c%d" % (alg, p, c)] = d
                print(name, alg, p, c, d, flush=True)
        manifest[name] = {"M": m, "N": n, "R": r, "nnz": int(len(rows)), "seed": CASES[name][4],
                          "draws": CASES[name][2], "deviation_from_canonical": dev}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
