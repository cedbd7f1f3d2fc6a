import os
import sys

import pytest

# Idle OpenMP workers of the host library sleep instead of spinning (read by libgomp when it is loaded, i.e. before any test module
# imports numpy / torch / the host library): a process that had created a CU-masked stream (the HNH_COMM_CUS tests) was seen to hang
# in the HIP runtime's exit handler in 4 of 80 runs while a large spinning pool was alive, never with a passive one
# (profiles/r04_masked_stream_exit_hang.log) — the test process should not be the 81st.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
