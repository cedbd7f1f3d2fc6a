"""Worker of tests/test_ipc_world_cpu.py and tests/test_multigpu_gpu.py: one PROCESS per rank over the ipc-pull transport
(IpcWorld: shared-memory control plane, receivers copy out of their peers' mapped buffers, stream-ordered by flag words).
No torch.distributed: the ranks meet in the transport's own shared-memory session, every rank leaves its results in a file
and rank 0 compares them with the reference's golden vectors.
    HNH_TEST_BACKEND=oracle : the C test double (process_vm_readv stands in for the mapped peer memory) — CPU suite
    HNH_TEST_BACKEND=hip    : the HIP library; all ranks share whatever GPUs are visible (rank % device count)"""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, n = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    session, outdir, case_name, configs = sys.argv[1:5]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    hip = os.environ.get("HNH_TEST_BACKEND", "oracle") == "hip"
    device = 0
    if hip:
        import torch  # first: one HIP runtime per process
        device = rank % torch.cuda.device_count()
        torch.cuda.set_device(device)
    import numpy as np
    import hnh_testlib as T
    from distributed_sddmm_amd import api as H
    if hip:
        assert H.load_backend(None) == "hip-gfx950"
    else:
        assert H.load_backend(T.ORACLE_BACKEND) == "oracle-cpu-test-double"
    world = H.World.ipc(rank, n, device, session)
    case = T.case_inputs(case_name)
    failures = []

    def gather(tag, out):
        with open(os.path.join(outdir, "%s_rank%d.pkl" % (tag, rank)), "wb") as f:
            pickle.dump(out, f)
        world.barrier()
        if rank != 0:
            return None
        got = []
        for r in range(n):
            with open(os.path.join(outdir, "%s_rank%d.pkl" % (tag, r)), "rb") as f:
                got.append(pickle.load(f))
        return got

    for what in range(len(H.World.PREFLIGHT)):
        err = world.preflight(what, 1 << 13)
        if not err <= 1e-9:
            failures.append("preflight %s: %r" % (H.World.PREFLIGHT[what], err))
    for i, item in enumerate(configs.split(";")):
        alg, c, mode, chunks = item.split(":")
        os.environ["HNH_RING_MODE"] = mode
        os.environ["HNH_MESH_CHUNKS"] = chunks
        if alg.startswith("als@"):
            gathered = gather("cfg%d" % i, T.run_als(world, alg[4:], int(c), case, 1, 5))
            if rank == 0:
                try:
                    T.check_als_against_golden(gathered, case)
                except AssertionError as e:
                    failures.append("%s: %r" % (item, e))
            continue
        gathered = gather("cfg%d" % i, T.run_all_ops(world, alg, int(c), case))
        if rank == 0:
            try:
                T.check_against_golden(T.assemble(gathered, case), gathered, case, alg)
                assert gathered[0]["alg_info"]["transport"] == "ipc-pull", gathered[0]["alg_info"]["transport"]
                if hip:
                    assert gathered[0]["alg_info"]["backend"] == "hip-gfx950"
            except AssertionError as e:
                failures.append("%s: %r" % (item, e))
    world.barrier()
    world.close()
    if rank == 0:
        print("IPC_FAIL " + " | ".join(failures) if failures else "IPC_OK", flush=True)


if __name__ == "__main__":
    main()
