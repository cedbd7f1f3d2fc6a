"""hnh_tuples_* on the GPU (rocPRIM radix sort + streaming kernels) against numpy; same body as the CPU double's test."""
import pytest

import tuples_common

pytestmark = pytest.mark.gpu


def test_hip_tuple_primitives():
    from distributed_sddmm_amd import _kernels as K
    ctx = K.Ctx(0)
    assert K.load().hnh_backend_name() == b"hip-gfx950"
    tuples_common.run(ctx)
    ctx.close()


def test_hip_round6_primitives():
    from distributed_sddmm_amd import _kernels as K
    ctx = K.Ctx(0)
    tuples_common.run_round6_primitives(ctx)
    ctx.close()


def test_sort_at_scale_is_a_permutation_in_order():
    """1e7 tuples (size-independent properties): sorted by key, same multiset of values."""
    import ctypes as C
    import numpy as np
    from distributed_sddmm_amd import _kernels as K
    ctx = K.Ctx(0)
    n = 10_000_000
    t = tuples_common.make_tuples(n, 1 << 20, 1 << 20, 11)
    d = ctx.upload(t)
    key = K.TupleKey(K.KEY_COL_ROW, 0, 0, 0, 0, None, 0)
    ctx.check(ctx.lib.hnh_tuples_sort(ctx.h, d.ptr, n, C.byref(key), 52, 0), "sort")
    got = d.get().view(K.TUPLE_DTYPE).reshape(-1)
    k = tuples_common.key_of(got, K.KEY_COL_ROW)
    assert np.all(k[1:] >= k[:-1])
    assert abs(float(got["value"].sum()) - float(t["value"].sum())) < 1e-6 and np.array_equal(np.sort(k), np.sort(tuples_common.key_of(t, K.KEY_COL_ROW)))
    d.free(); ctx.close()
