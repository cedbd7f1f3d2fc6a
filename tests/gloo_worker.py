"""Worker of tests/test_gloo_world.py: one process per rank, torch.distributed (gloo, CPU) as the transport
of a CallbackWorld.  The kernel ABI is served by the oracle's C test double (no GPU here); what is under
test is the multi-PROCESS path of the host layer: redistribution all-to-all, grid communicators, ring
shifts, fiber collectives — the same C++ code that runs over RCCL on the GPUs."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hnh_testlib as T  # noqa: E402
from distributed_sddmm_amd import api as H  # noqa: E402


def as_tensor(ptr, nbytes):
    return torch.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=torch.uint8)


def make_callbacks():
    def sendrecv(user, sendbuf, sendbytes, dst, recvbuf, recvbytes, src):
        try:
            reqs = []
            if sendbytes:
                reqs.append(dist.isend(as_tensor(sendbuf, sendbytes), dst))
            if recvbytes:
                reqs.append(dist.irecv(as_tensor(recvbuf, recvbytes), src))
            for r in reqs:
                r.wait()
            return 0
        except Exception as e:  # noqa: BLE001
            print("sendrecv callback failed:", e, flush=True)
            return 1

    def barrier(user):
        dist.barrier()
        return 0

    def allgather(user, send, recv, nbytes):
        try:
            out = as_tensor(recv, nbytes * dist.get_world_size())
            dist.all_gather_into_tensor(out, as_tensor(send, nbytes).clone())
            return 0
        except Exception as e:  # noqa: BLE001
            print("allgather callback failed:", e, flush=True)
            return 1

    cb = H.CommCallbacks(None, H.SENDRECV_CB(sendrecv), H.BARRIER_CB(barrier), H.ALLGATHER_CB(allgather))
    return cb


def main():
    rank, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=n)
    assert H.load_backend(T.ORACLE_BACKEND) == "oracle-cpu-test-double"
    cb = make_callbacks()
    world = H.World.callback(rank, n, 0, cb)
    case_name, configs = sys.argv[1], sys.argv[2]
    if case_name == "cfg1":  # BASELINE config 1: ER 2^16 x 2^16, edge factor 16, R = 16 (checked against the oracle)
        from oracle import oracle as O
        rows, cols = O.erdos_renyi(16, 16)
        case = T.make_case("cfg1", 1 << 16, 1 << 16, 16, rows, cols)
    else:
        case = T.case_inputs(case_name)
    failures = []
    for item in configs.split(";"):
        alg, c = item.split(":")
        if alg.startswith("als@"):  # ALS-CG (R-split all-reduce, hold hint) over the multi-process transport, vs the reference's golden
            out = T.run_als(world, alg[4:], int(c), case, 1, 5)  # steps / CG iterations of tests/golden/make_golden_als.py
            gathered = [None] * n
            dist.all_gather_object(gathered, out)
            if rank == 0:
                try:
                    T.check_als_against_golden(gathered, case)
                except AssertionError as e:
                    failures.append("%s c=%s: %r" % (alg, c, e))
            continue
        if alg.startswith("gat@"):  # GAT forward (fused head under local kernel fusion) vs the reference's golden features
            out = T.run_gat(world, alg[4:], int(c), case)
            gathered = [None] * n
            dist.all_gather_object(gathered, out)
            if rank == 0:
                try:
                    got = T.assemble_dense(gathered, "gat", "subA", case["M"], T.GAT_LAYERS[-1][1] * T.GAT_LAYERS[-1][2])
                    gold = dict(np.load(os.path.join(T.GOLDEN, "gat_er8_r16.npz")))["out"]
                    assert T.rel(got, gold) <= T.TOL
                except AssertionError as e:
                    failures.append("%s c=%s: %r" % (alg, c, e))
            continue
        out = T.run_all_ops(world, alg, int(c), case)
        gathered = [None] * n
        dist.all_gather_object(gathered, out)
        if rank == 0:
            try:
                if case_name == "cfg1":
                    T.check_against_oracle(T.assemble(gathered, case), case, alg)
                else:
                    T.check_against_golden(T.assemble(gathered, case), gathered, case, alg)
                assert gathered[0]["alg_info"]["transport"] == "callback"
            except AssertionError as e:
                failures.append("%s c=%s: %r" % (alg, c, e))
    vals, ok = world.grid_probe(n, 1, 1, 1)
    assert ok and vals[0] == rank
    world.close()
    dist.barrier()
    if rank == 0:
        print("GLOO_FAIL " + " | ".join(failures) if failures else "GLOO_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
