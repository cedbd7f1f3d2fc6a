"""hnh_tuples_* (the setup pipeline's primitives) of the oracle's C test double, against numpy — the same body runs on
the GPU in test_tuples_gpu.py; and the HNH_HOST_SETUP=1 pipeline (the reference's host algorithm restated) must give
the same operator results as the device pipeline."""
import ctypes as C

import numpy as np
import pytest

import hnh_testlib as T
import cg_common
import tuples_common
from distributed_sddmm_amd import _kernels as K
from distributed_sddmm_amd import api as H


class HostArray:
    def __init__(self, a):
        self.a = np.ascontiguousarray(a).copy()
        self.ptr = self.a.ctypes.data_as(C.c_void_p)

    def get(self):
        return self.a

    def free(self):
        pass


class DoubleApi:
    def __init__(self):
        self.lib = C.CDLL(T.ORACLE_BACKEND)
        for name, (res, args) in K.SIGNATURES.items():
            f = getattr(self.lib, name)
            f.restype, f.argtypes = res, args
        self.h = C.c_void_p()
        assert self.lib.hnh_ctx_create(0, C.byref(self.h)) == 0

    def upload(self, a):
        return HostArray(a)

    def check(self, rc, what):
        assert rc == 0, what


def test_c_test_double_tuple_primitives():
    tuples_common.run(DoubleApi())


def test_c_test_double_round6_primitives():
    tuples_common.run_round6_primitives(DoubleApi())


def test_world_identities_say_where_every_rank_runs():
    """World::identities() (hnh_world_identities): one record per rank — pid, device ordinal, the GPU's PCI bus id — the same list on every
    rank; bench.py builds its workload sentence from the number of distinct bus ids."""
    import os
    assert H.load_backend(T.ORACLE_BACKEND) == "oracle-cpu-test-double"
    res = H.run_spmd(3, lambda w: w.identities())
    assert res[0] == res[1] == res[2] and [r["rank"] for r in res[0]] == [0, 1, 2]
    assert all(r["pid"] == os.getpid() and r["device_ordinal"] == 0 and r["pci_bus_id"] for r in res[0]) and "comm_count" not in res[0][0]


@pytest.mark.parametrize("R,hubs,windows,standalone", [(16, False, 0, False), (7, False, 3, False), (32, True, 0, False), (16, False, 0, True)])
def test_c_test_double_folded_cg_iteration(R, hubs, windows, standalone):
    """hnh_cg_update as the test double performs it vs the reference's statement sequence in numpy (the GPU runs the same body)."""
    cg_common.run(DoubleApi(), R, hubs=hubs, windows=windows, standalone=standalone)


@pytest.mark.parametrize("R,hubs,windows", [(16, False, 0), (7, False, 3), (32, True, 0)])
def test_c_test_double_relu_delivery(R, hubs, windows):
    """relu_dst of hnh_fused_extras as the test double performs it vs numpy (the GPU runs the same body)."""
    cg_common.run_relu(DoubleApi(), R, hubs=hubs, windows=windows)


@pytest.mark.parametrize("alg,p,c", [("15d_fusion2", 4, 2), ("15d_fusion1", 4, 1), ("15d_sparse", 4, 2), ("25d_dense_replicate", 8, 2),
                                     ("25d_sparse_replicate", 8, 2)])
def test_host_setup_pipeline_still_matches_the_reference(monkeypatch, alg, p, c):
    H.load_backend(T.ORACLE_BACKEND)
    monkeypatch.setenv("HNH_HOST_SETUP", "1")
    for name in ("er8_r16", "rect_r16"):
        case = T.case_inputs(name)
        if not T.valid_config(alg, p, c, case["R"]):
            continue
        per_rank = H.run_spmd(p, lambda w: T.run_all_ops(w, alg, c, case))
        T.check_against_golden(T.assemble(per_rank, case), per_rank, case, alg)
