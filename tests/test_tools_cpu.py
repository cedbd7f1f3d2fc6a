"""The measurement tools that take `--backend` run end to end on the CPU test double (tiny sizes; the numbers mean nothing):
keeps tools/rank_share_probe.py and tools/overlap_probe.py from rotting between the GPU sessions that need them."""
import os
import subprocess
import sys

import hnh_testlib as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_tool(name, *args):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("HNH_MESH_CHUNKS", "HNH_MESH_TAPER", "HNH_PACE_LINK_GBPS", "HNH_PACE_COPY", "HNH_FORCE_WINDOWS"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name), "--backend", T.ORACLE_BACKEND, "--logm", "10", "--ef", "8", "--r", "16",
                          "--p", "4", "--iters", "2", *args], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    return res.stdout


def test_rank_share_probe_runs_on_the_test_double():
    out = run_tool("rank_share_probe.py", "--chunks", "1,4")
    lines = [ln for ln in out.splitlines() if "rank 0 alone" in ln]
    assert len(lines) == 2 and "chunks=1" in lines[0] and "chunks=4" in lines[1]


def test_overlap_probe_runs_on_the_test_double():
    out = run_tool("overlap_probe.py", "--chunks", "2", "--tapers", "3,2,1;1,2,2,2,1,1", "--pace", "60", "--copy-wgs", "0,2")
    assert out.count("unpaced call") == 6  # three shapes, without and with the paced copies
    assert "taper 3,2,1:" in out and "taper 1,2,2,2,1,1:" in out and "Q=2:" in out
    assert out.count("GB/s/link:") == 6  # one rate per shape and variant
    assert out.count("the paced transfers also move their bytes: 2 throttled workgroups per link") == 3


def test_accumulator_overlap_probe_runs_on_the_test_double():
    out = run_tool("overlap_probe_accumulator.py", "--gbps", "0,60")
    assert "15d_fusion1 spmmA" in out and "loopback copies only" in out and out.count(" ms") >= 4
