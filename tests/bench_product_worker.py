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
from benchlib import transports as BT  # noqa: E402  (where bench.py's transport hooks live)
import hnh_testlib as T  # noqa: E402
from distributed_sddmm_amd import api as H  # noqa: E402

real_load = H.load_backend


def load_backend(path=None):
    return real_load(path or T.ORACLE_BACKEND)


H.load_backend = load_backend
BT.PRODUCT_BACKEND = "oracle-cpu-test-double"
BT.PROBE_SCRIPT = os.path.abspath(__file__)
BT.visible_device = lambda rank, n, local_rank: (0, n)
torch.cuda.synchronize = lambda *a, **k: None

if os.environ.get("BENCH_PRODUCT_BREAK") == "rccl":  # one transport that cannot be created on this "node": its trial must say so, the run goes on
    real_make = BT.make_gpu_transport

    def make_gpu_transport(H_, dist, rank, n, device, name):
        if name == "rccl":
            raise RuntimeError("RCCL is broken on this node (test)")
        return real_make(H_, dist, rank, n, device, name)
    BT.make_gpu_transport = make_gpu_transport

if os.environ.get("BENCH_PRODUCT_BREAK") == "rccl-preflight" and "--probe-transport" not in sys.argv:
    # a transport that passes its child-process trial and then delivers wrong data in the benchmark process's own preflight
    real_make2, real_preflight = BT.make_gpu_transport, BT.run_preflight

    def make_tagged(H_, dist, rank, n, device, name):
        w = real_make2(H_, dist, rank, n, device, name)
        w._test_transport = name
        return w

    def preflight(H_, world, count, dog=None):
        if getattr(world, "_test_transport", None) == "rccl" and int(os.environ["RANK"]) == 1:
            raise RuntimeError("preflight: ring sendrecv delivered wrong data (test)")
        return real_preflight(H_, world, count, dog)
    BT.make_gpu_transport, BT.run_preflight = make_tagged, preflight

if os.environ.get("BENCH_PRODUCT_BREAK") == "ipc-hang-late" and "--probe-transport" not in sys.argv:
    # the SECOND transport hangs while it is created in the benchmark process, on one rank (its peers wait for it inside the transport)
    real_make3 = BT.make_gpu_transport

    def make_late(H_, dist, rank, n, device, name):
        if name == "ipc" and rank == 1:
            import time
            time.sleep(3600)
        return real_make3(H_, dist, rank, n, device, name)
    BT.make_gpu_transport = make_late

if os.environ.get("BENCH_PRODUCT_BREAK") == "rccl-dies-in-search" and "--probe-transport" not in sys.argv:
    # the transports work until the search: there a candidate raises on one rank (the other rank is inside the transport, waiting)
    real_build = bench.Bench.build

    def build(self, route):
        if route[3] == "4" and int(os.environ["RANK"]) == 1:  # (whichever transport won the first stage)
            raise RuntimeError("transport error mid-search (test)")
        return real_build(self, route)
    bench.Bench.build = build

if os.environ.get("BENCH_PRODUCT_SLOW") and "--probe-transport" not in sys.argv:
    # every candidate of the route search takes this many seconds longer (a node whose candidates are slow: the time budget's test)
    from benchlib import search as BS
    real_try = BS.try_route

    def slow_try(b, route, calls=5):
        import time
        time.sleep(float(os.environ["BENCH_PRODUCT_SLOW"]))
        return real_try(b, route, calls)
    BS.try_route = slow_try

if __name__ == "__main__":
    if os.environ.get("BENCH_PRODUCT_BREAK") == "all-trials" and "--probe-transport" in sys.argv:
        sys.exit(9)  # the trial machinery itself is broken on this "node": every child fails before it gets anywhere
    if os.environ.get("BENCH_PRODUCT_BREAK") == "rccl-hang" and "--probe-transport" in sys.argv and sys.argv[sys.argv.index("--probe-transport") + 1] == "rccl":
        import time
        time.sleep(3600)  # a transport whose trial never answers: the parent ends it at --probe-timeout
    bench.main()
