"""Development tool: what the throttled copy kernel behind HNH_PACE_COPY (hnh_stream_paced_copy) manages ALONE on the GPU, per
workgroup — the calibration for reading tools/overlap_probe.py --copy-wgs (its copy loop is far weaker than an RCCL channel)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributed_sddmm_amd import _kernels as K
ctx = K.Ctx(0); lib = ctx.lib
slice_bytes = 32 << 20
src = K.DevArray(ctx, (slice_bytes // 8,), np.float64); dst = K.DevArray(ctx, (7 * slice_bytes // 8,), np.float64)
lib.hnh_fill_f64(ctx.h, src.ptr, slice_bytes // 8, 1.0, 0); ctx.sync()
for wgs in (1, 2, 4, 8, 16):
    ctx.check(lib.hnh_stream_paced_copy(ctx.h, 1, dst.ptr, src.ptr, slice_bytes, 7, 0.0, wgs), "pc"); ctx.sync(1)
    t = time.perf_counter()
    for _ in range(5):
        ctx.check(lib.hnh_stream_paced_copy(ctx.h, 1, dst.ptr, src.ptr, slice_bytes, 7, 0.0, wgs), "pc")
    ctx.sync(1); dt = (time.perf_counter() - t) / 5
    print("standalone paced_copy, 7 slices x 32 MiB, %2d workgroups per slice: %.3f ms -> %.1f GB/s per slice, %.1f GB/s per workgroup (read side)" % (wgs, dt * 1e3, slice_bytes / dt / 1e9, slice_bytes / dt / 1e9 / wgs))
