"""Device-to-device transports of a multi-GPU run: device selection, creation (RCCL over xGMI; ipc-pull with copy engines or a pull
kernel), the preflight of every primitive the schedules use, and the child-process trials that keep a transport which cannot
initialise, delivers wrong data or hangs on this node out of the benchmark process."""
import argparse
import os
import sys
import time

from . import common

PRODUCT_BACKEND = "hip-gfx950"  # the only kernel library the product path accepts: there is no CPU mode here.  (The CPU test of the
PROBE_SCRIPT = os.path.join(common.ROOT, "bench.py")  # multi-GPU control flow, tests/bench_product_worker.py, replaces both — and the device
#                                                       selection — FROM OUTSIDE, with the kernel test double and its emulated transports.)


def visible_device(rank, n, local_rank):
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    ndev = torch.cuda.device_count()
    if n > 1 and 1 < ndev < n:
        raise SystemExit("bench.py --gpus %d: this process sees %d GPUs; one process per GPU needs either all %d visible to every rank "
                         "(LOCAL_RANK picks one) or exactly one per rank (launcher-side isolation)" % (n, ndev, n))
    device = local_rank % ndev
    torch.cuda.set_device(device)
    return device, ndev


def make_gpu_transport(H, dist, rank, n, device, name):
    """One process per GPU.  rccl: explicit-peer send/recv groups over xGMI (unique id handed round through torch.distributed);
    ipc / ipc-kernel: receivers pull out of their peers' mapped buffers — copy engines on forked streams / one gather-copy kernel."""
    if name == "rccl":
        ident = [H.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        return H.World.rccl(rank, n, device, ident[0])
    session = [H.ipc_session_id() if rank == 0 else None]
    dist.broadcast_object_list(session, src=0)
    os.environ["HNH_IPC_PULL"] = "kernel" if name == "ipc-kernel" else "engine"
    return H.World.ipc(rank, n, device, session[0])


def gpu_world(H, dist, rank, n, local_rank):
    """The default transport of the product: RCCL over xGMI (N = 1: no transport at all)."""
    import torch
    device, ndev = visible_device(rank, n, local_rank)
    assert H.load_backend(None) == PRODUCT_BACKEND
    if n == 1:
        return H.World.single(device), torch.cuda.synchronize
    try:
        return make_gpu_transport(H, dist, rank, n, device, "rccl"), torch.cuda.synchronize
    except Exception as e:
        raise SystemExit("bench.py --gpus %d, rank %d on device %d of %d visible: the RCCL communicator could not be created: %s\n"
                         "(\"invalid usage\" here usually means two ranks share one physical GPU, which RCCL refuses)" % (n, rank, device, ndev, e))


def run_preflight(H, world, count, dog=None):
    """Every transport primitive the schedules use, on small buffers with known contents; returns {name: max deviation}."""
    res = {}
    for what, name in enumerate(H.World.PREFLIGHT):
        if dog is not None:
            dog.phase("preflight: " + name, 150.0)  # (longer than HNH_IPC_WAIT_S: a transport with a time limit of its own reports instead of being cut off)
        res[name] = world.preflight(what, count)
        if dog is not None:
            dog.done()
        if not res[name] <= 1e-9:
            raise RuntimeError("preflight: %s delivered wrong data (max deviation %.3e)" % (name, res[name]))
    return res


def probe_main(args):
    """Child process of probe_transports(): create the transport, run the preflight and one small keyed fusedSpMM over it.
    Exit code 0 = usable.  Whatever goes wrong here — an exception, a hang the parent ends — stays in this process."""
    import torch
    import torch.distributed as dist
    from distributed_sddmm_amd import api as H
    from . import checks, guards, timed
    rank, n, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group(backend="gloo", rank=rank, world_size=n)
    device, _ = visible_device(rank, n, local_rank)
    assert H.load_backend(None) == PRODUCT_BACKEND
    world = make_gpu_transport(H, dist, rank, n, device, args.probe_transport)
    run_preflight(H, world, 1 << 16)
    wl = common.Workload("er", 12, 8)
    b = timed.Bench(argparse.Namespace(**dict(vars(args), r=32, alg="15d_fusion2", app="vanilla", steps=1, warmup=0, no_check=False)), H, torch, dist, rank, n,
                    guards.Watchdog(rank, 0, False), wl)
    b.add_transport(args.probe_transport, world, torch.cuda.synchronize)
    b.build((args.probe_transport, 1, "mesh" if n > 1 else "none", "2"))
    chk = checks.check(b)
    b.free_current()
    b.close_transports()
    dist.barrier()
    dist.destroy_process_group()
    if not chk["ok"]:
        sys.stderr.write("[bench.py transport probe] rank %d, %s: the keyed result check failed: %r\n" % (rank, args.probe_transport, chk))
        sys.exit(4)
    sys.exit(0)


def probe_transports(args, dist, rank, n, names):
    """Tries each transport in a CHILD process per rank (own rendezvous port) under a time limit, so that a transport that
    cannot initialise, delivers wrong data or hangs on this node never gets into this process.  Returns {name: "ok" | reason}."""
    import socket
    import subprocess
    verdicts = {}
    for name in names:
        port = [None]
        if rank == 0:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port[0] = sk.getsockname()[1]
        dist.broadcast_object_list(port, src=0)
        env = dict(os.environ, MASTER_PORT=str(port[0]), MASTER_ADDR="127.0.0.1")
        env.pop("HNH_BENCH_STATUS_DIR", None)
        # under torch.distributed.run the workers are told to use the AGENT's store (TORCHELASTIC_USE_AGENT_STORE): the children
        # rendezvous on a port of their own, where rank 0's child has to host the store itself
        for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
            env.pop(k)
        cmd = [sys.executable, PROBE_SCRIPT, "--gpus", str(n), "--probe-transport", name]
        if args.nchannels:
            cmd += ["--nchannels", str(args.nchannels)]
        t0 = time.perf_counter()
        try:
            res = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=args.probe_timeout,
                                 preexec_fn=common.unblock_signals)
            mine = "ok" if res.returncode == 0 else "rank %d: exit code %d: %s" % (rank, res.returncode, (res.stderr or "").strip().splitlines()[-1][:200] if (res.stderr or "").strip() else "")
        except subprocess.TimeoutExpired:
            mine = "rank %d: no answer within %.0f s (hang)" % (rank, args.probe_timeout)
        everyone = [None] * n
        dist.all_gather_object(everyone, mine)
        bad = [v for v in everyone if v != "ok"]
        verdicts[name] = "ok (%.0f s)" % (time.perf_counter() - t0) if not bad else bad[0]
    return verdicts
