"""Golden numbers at the sizes bench.py reports, produced by the REFERENCE ITSELF (oracle/_ref/ref_driver = its unmodified
sources + MKL / MPICH) on the build container's host cores, so that the GPU box does not have to spend minutes of host time
re-running it on every test run (tests/test_fullsize_gpu.py, tests/test_schedules_gpu.py fall back to running it live where a
fixture is missing, or always with HNH_LIVE_REFERENCE=1):

  fullsize_reference.json
      "config2_fingerprints"   scratch.cpp:26-76 trio (sddmmA / spmmA / spmmB under dummyInitialize, S = 1) at BASELINE config 2's
                               full size: ER 2^20 x 2^20, edge factor 96 -> 100 658 766 nonzeros (host generator, seed 12345), R = 128
      "at_scale_fingerprints"  the same trio at 2^18, edge factor 32 (8.4e6 nonzeros), R = 128, for 15d_fusion2 and 15d_fusion1
  fullsize_cfg5_als.npz
      one alternating ALS step (cg_optimizer(Amat, 2); cg_optimizer(Bmat, 2), als_conjugate_gradients.cpp) at config 2's matrix,
      R = 128, from hashed initial factors and ground truth: the residuals before / after, 512 evenly spaced rows of the resulting
      A and B, and the column sums of both (every row contributes)

Run where /root/reference exists (about 15 minutes on 8 cores, ~30 GB of memory):  python tests/golden/make_golden_fullsize.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from distributed_sddmm_amd import api as H  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import refrun as RR  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE_ROWS = 512


def main():
    which = set(sys.argv[1:]) or {"at_scale", "config2", "als"}
    path = os.path.join(HERE, "fullsize_reference.json")
    rec = json.load(open(path)) if os.path.exists(path) else {}
    threads = min(32, os.cpu_count() or 1)
    if "at_scale" in which:
        logm, ef, r = 18, 32, 128
        m = 1 << logm
        rows, cols = O.erdos_renyi_mn(m, m, m * ef, 12345)
        out = {"logm": logm, "edge_factor": ef, "R": r, "nnz": int(len(rows)), "seed": 12345}
        for alg in ("15d_fusion2", "15d_fusion1"):
            ref = RR.fingerprints(m, m, rows, cols, r, alg, 1, 1, timeout=1800)
            out[alg] = [ref["sddmm"], ref["spmmA"], ref["spmmB"]]
            print("at scale", alg, out[alg], flush=True)
        rec["at_scale_fingerprints"] = out
        json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
    if which & {"config2", "als"}:
        logm, ef, r = 20, 96, 128
        m = 1 << logm
        t = time.time()
        rows, cols = H.generate_er(m, m, m * ef, 12345)
        assert len(rows) == 100658766
        print("generated %d nonzeros in %.0f s" % (len(rows), time.time() - t), flush=True)
    if "config2" in which:
        t = time.time()
        ref = RR.fingerprints(m, m, rows, cols, r, "15d_fusion2", 1, 1, timeout=7200, threads=threads)
        rec["config2_fingerprints"] = {"logm": logm, "edge_factor": ef, "R": r, "nnz": int(len(rows)), "seed": 12345, "alg": "15d_fusion2",
                                       "values": [ref["sddmm"], ref["spmmA"], ref["spmmB"]], "host_threads": threads,
                                       "seconds": round(time.time() - t)}
        print("config 2", rec["config2_fingerprints"], flush=True)
        json.dump(rec, open(path, "w"), indent=1, sort_keys=True)
    if "als" in which:
        t = time.time()
        vals = O.sparse_values(rows, cols, m, 5)
        a0, b0 = O.dense_fill(m, r, 11), O.dense_fill(m, r, 12)
        ref = RR.als(m, m, rows, cols, vals, r, a0, b0, "15d_fusion2", 1, 1, steps=1, cg_iters=2, timeout=14400, threads=threads)
        idx = np.arange(0, m, m // SAMPLE_ROWS)[:SAMPLE_ROWS]
        np.savez_compressed(os.path.join(HERE, "fullsize_cfg5_als.npz"), rows=idx, A=ref["A"][idx], B=ref["B"][idx], colsum_A=ref["A"].sum(axis=0),
                            colsum_B=ref["B"].sum(axis=0), residuals=ref["residuals"], absmax=np.array([np.abs(ref["A"]).max(), np.abs(ref["B"]).max()]))
        rec["config5_als"] = {"logm": logm, "edge_factor": ef, "R": r, "steps": 1, "cg_iters": 2, "value_seed": 5, "a_seed": 11, "b_seed": 12,
                              "residuals": ref["residuals"].tolist(), "host_threads": threads, "seconds": round(time.time() - t)}
        print("als", rec["config5_als"], flush=True)
        json.dump(rec, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
