"""Worker of tests/test_multigpu_gpu.py: one process per GPU, the PRODUCT transport (RcclWorld: explicit-peer
ncclSend/ncclRecv groups on one communicator over xGMI) and the HIP kernels; torch.distributed (gloo) only carries
the unique id and gathers the per-rank results for the comparison with the reference's golden vectors."""
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


def main():
    rank, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    dist.init_process_group(backend="gloo", rank=rank, world_size=n)
    # HNH_RCCL_WORKER_DOUBLE=1 (tests/test_rccl_emulation_cpu.py): the same worker without GPUs — the kernel test double, whose RCCL section
    # emulates RCCL's calling contract between the processes of this host
    double = os.environ.get("HNH_RCCL_WORKER_DOUBLE") == "1"
    if double:
        device, backend = 0, "oracle-cpu-test-double"
        assert H.load_backend(T.ORACLE_BACKEND) == backend
    else:
        device, backend = rank % torch.cuda.device_count(), "hip-gfx950"
        torch.cuda.set_device(device)
        assert H.load_backend(None) == backend
    ident = [H.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0)
    world = H.World.rccl(rank, n, device, ident[0])
    case_name, configs = sys.argv[1], sys.argv[2]
    case = T.case_inputs(case_name)
    failures = []
    for what in range(len(H.World.PREFLIGHT)):
        err = world.preflight(what, 1 << 14)
        if not err <= 1e-9:
            failures.append("preflight %s: %r" % (H.World.PREFLIGHT[what], err))
    for item in configs.split(";"):
        alg, c, mode, chunks = item.split(":")
        os.environ["HNH_RING_MODE"] = mode
        os.environ["HNH_MESH_CHUNKS"] = chunks
        if alg.startswith("als@"):
            out = T.run_als(world, alg[4:], int(c), case, 1, 5)
            gathered = [None] * n
            dist.all_gather_object(gathered, out)
            if rank == 0:
                try:
                    T.check_als_against_golden(gathered, case)
                except AssertionError as e:
                    failures.append("%s: %r" % (item, e))
            continue
        out = T.run_all_ops(world, alg, int(c), case)
        gathered = [None] * n
        dist.all_gather_object(gathered, out)
        if rank == 0:
            try:
                T.check_against_golden(T.assemble(gathered, case), gathered, case, alg)
                assert gathered[0]["alg_info"]["transport"] == "rccl" and gathered[0]["alg_info"]["backend"] == backend
            except AssertionError as e:
                failures.append("%s: %r" % (item, e))
    world.close()
    dist.barrier()
    if rank == 0:
        print("RCCL_FAIL " + " | ".join(failures) if failures else "RCCL_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
