import os
import sys

import pytest

# Idle OpenMP workers of the host library sleep instead of spinning (read by libgomp when it is loaded, i.e. before any test module
# imports numpy / torch / the host library): a process that had created a CU-masked stream (the HNH_COMM_CUS tests) was seen to hang
# in the HIP runtime's exit handler in 4 of 80 runs while a large spinning pool was alive, never with a passive one
# (profiles/r04_masked_stream_exit_hang.log) — the test process should not be the 81st.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
# The kernel test double checks the host layer's STREAM PROTOCOL while it computes (oracle/hnh_stream_order.h): two calls that touch the same
# bytes of "device" memory from different streams, at least one writing, with no event / synchronisation path between them, are a race —
# which a synchronous CPU run could never show as a wrong number.  On by default for every test process (and the processes they start, which
# exit with status 86 and the report if they saw one); this process reports at the end of the session (below).  HNH_ORDER_CHECK=0 turns it off.
os.environ.setdefault("HNH_ORDER_CHECK", "1")

# bench.py writes the full record of a run beside its (bounded) line: tests keep it out of the tree
if "HNH_BENCH_RECORD" not in os.environ:
    import tempfile
    os.environ["HNH_BENCH_RECORD"] = os.path.join(tempfile.mkdtemp(prefix="hnh_bench_record_"), "bench_secondary.json")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """The test double's happens-before checker of the stream protocol (oracle/hnh_stream_order.h) ran beside every test of this
    process: every race it saw is printed and fails the session."""
    if os.environ.get("HNH_ORDER_CHECK", "0") in ("", "0"):
        return
    import ctypes
    path = os.path.join(ROOT, "oracle", "liboracle_backend.so")
    # only when a test of this session loaded the double: a `-m gpu` session never does (its tests assert the HIP backend), and mapping
    # the checker's library there just to ask for an empty report would put an oracle file among the libraries the GPU run loaded
    try:
        with open("/proc/self/maps") as f:
            mapped = any(os.path.basename(path) in line for line in f)
    except OSError:
        mapped = os.path.exists(path)
    if not mapped:
        return
    lib = ctypes.CDLL(path)
    lib.hnh_oracle_order_report.restype = ctypes.c_long
    lib.hnh_oracle_order_accesses.restype = ctypes.c_long
    buf = ctypes.create_string_buffer(16384)
    races = lib.hnh_oracle_order_report(buf, 16384)
    print("\n[stream-order checker] %d accesses checked in this process, %d races" % (lib.hnh_oracle_order_accesses(), races))
    if races:
        print(buf.value.decode())
        session.exitstatus = 1
