"""Worker of tests/test_bench_cpu.py::test_the_multi_gpu_product_path_*: bench.run() through its PRODUCT branch for several GPUs —
device selection, transport trials in child processes, creation of every usable transport in the benchmark process, preflight, the
default route, the search over transports x replication factors x routes, the final measurement — with three things replaced from
outside: the kernel library (the CPU test double instead of the HIP one), the device selection (no GPU here), and the script the
transport trials start (this one, so that the children get the same replacements).  The transports are the product's own classes:
RcclWorld over the double's emulation of RCCL between processes, IpcWorld over its process_vm_readv pull."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
import hnh_testlib as T  # noqa: E402
from distributed_sddmm_amd import api as H  # noqa: E402

real_load = H.load_backend


def load_backend(path=None):
    return real_load(path or T.ORACLE_BACKEND)


H.load_backend = load_backend
bench.PRODUCT_BACKEND = "oracle-cpu-test-double"
bench.PROBE_SCRIPT = os.path.abspath(__file__)
bench.visible_device = lambda rank, n, local_rank: (0, n)
torch.cuda.synchronize = lambda *a, **k: None

if os.environ.get("BENCH_PRODUCT_BREAK") == "rccl":  # one transport that cannot be created on this "node": its trial must say so, the run goes on
    real_make = bench.make_gpu_transport

    def make_gpu_transport(H_, dist, rank, n, device, name):
        if name == "rccl":
            raise RuntimeError("RCCL is broken on this node (test)")
        return real_make(H_, dist, rank, n, device, name)
    bench.make_gpu_transport = make_gpu_transport

if __name__ == "__main__":
    if os.environ.get("BENCH_PRODUCT_BREAK") == "rccl-hang" and "--probe-transport" in sys.argv and sys.argv[sys.argv.index("--probe-transport") + 1] == "rccl":
        import time
        time.sleep(3600)  # a transport whose trial never answers: the parent ends it at --probe-timeout
    bench.main()
