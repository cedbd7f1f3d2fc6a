"""Worker of tests/test_multigpu_gpu.py::test_bench_processes_share_one_gpu: bench.run() — the real benchmark driver with the
real HIP kernels and device-resident data — on several PROCESSES that share the one GPU of a test box.  RCCL refuses two ranks
on one device, so the transport is a CallbackWorld whose send/recv stage device buffers through the host and travel over gloo:
everything bench.py --gpus N does on a multi-GPU node except the RCCL calls themselves (preflight, set-up through the device
all-to-all, replication factor / route search, timed steps, row/column-keyed result check) runs as it will there."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HIP = C.CDLL("libamdhip64.so")
HIP.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
HIP.hipMemcpy.restype = C.c_int
MEMCPY_DEFAULT = 4  # hipMemcpyDefault: the runtime works out on which side each pointer lives (the callbacks get both kinds)


def staged_callbacks(H):
    def to_host(ptr, nbytes):
        buf = np.empty(nbytes, dtype=np.uint8)
        if nbytes and HIP.hipMemcpy(buf.ctypes.data, ptr, nbytes, MEMCPY_DEFAULT) != 0:
            raise RuntimeError("hipMemcpy to the staging buffer failed")
        return torch.from_numpy(buf)

    def sendrecv(user, sendbuf, sendbytes, dst, recvbuf, recvbytes, src):
        try:
            reqs, incoming = [], None
            if sendbytes:
                reqs.append(dist.isend(to_host(sendbuf, sendbytes), dst))
            if recvbytes:
                incoming = torch.empty(recvbytes, dtype=torch.uint8)
                reqs.append(dist.irecv(incoming, src))
            for r in reqs:
                r.wait()
            if incoming is not None and HIP.hipMemcpy(recvbuf, incoming.numpy().ctypes.data, recvbytes, MEMCPY_DEFAULT) != 0:
                raise RuntimeError("hipMemcpy from the staging buffer failed")
            return 0
        except Exception as e:  # noqa: BLE001
            print("sendrecv callback failed:", e, flush=True)
            return 1

    def barrier(user):
        dist.barrier()
        return 0

    def allgather(user, send, recv, nbytes):  # host buffers
        try:
            out = torch.empty(nbytes * dist.get_world_size(), dtype=torch.uint8)
            dist.all_gather_into_tensor(out, to_host(send, nbytes))
            C.memmove(recv, out.numpy().ctypes.data, out.numel())
            return 0
        except Exception as e:  # noqa: BLE001
            print("allgather callback failed:", e, flush=True)
            return 1

    return H.CommCallbacks(None, H.SENDRECV_CB(sendrecv), H.BARRIER_CB(barrier), H.ALLGATHER_CB(allgather))


def shared_gpu_world(H, dist_, rank, n, local_rank):
    torch.cuda.set_device(0)
    assert H.load_backend(None) == "hip-gfx950"
    if n == 1:
        return H.World.single(0), torch.cuda.synchronize
    shared_gpu_world.cb = staged_callbacks(H)  # keep the ctypes thunks alive
    return H.World.callback(rank, n, 0, shared_gpu_world.cb), torch.cuda.synchronize


if __name__ == "__main__":
    args = bench.parse(sys.argv[1:])
    assert args.gpus == int(os.environ["WORLD_SIZE"])
    bench.run(args, make_world=shared_gpu_world)
