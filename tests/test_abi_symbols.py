"""CPU-only: the HIP C-ABI library loads here (no GPU needed to dlopen it) and exports every symbol that
include/hnh_kernels.h declares; creating a context without a GPU fails loudly instead of falling back."""
import os
import re

import pytest

from distributed_sddmm_amd import _kernels as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(hnh_[a-z0-9_]+)\s*\(", txt))


def test_kernel_library_exports_every_declared_symbol():
    lib = K.load()
    declared = declared_symbols("hnh_kernels.h")
    assert declared, "no declarations parsed"
    assert declared == set(K.SIGNATURES), (declared ^ set(K.SIGNATURES))
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.hnh_backend_name() == b"hip-gfx950"


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(K.HnhError):
        K.Ctx(0)
