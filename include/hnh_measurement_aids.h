/*
 * hnh_measurement_aids.h — NOT part of the drop-in boundary.  Two stand-ins for an xGMI transfer of known duration, exported by
 * libhnh_kernels.so for the overlap measurements on a single GPU (tools/overlap_probe.py, tools/rank_share_probe.py,
 * tools/paced_copy_calibration.py).  The product host library (libhnh_host.so) never calls them: the schedule code that does
 * is compiled only into libhnh_host_aids.so (-DHNH_MEASUREMENT_AIDS, csrc/build_host.sh), which the tools load explicitly.
 * No counterpart in the reference.
 */
#ifndef HNH_MEASUREMENT_AIDS_H
#define HNH_MEASUREMENT_AIDS_H
#include "hnh_kernels.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Holds `stream` for `microseconds` (one idle-spinning wave on the constant 100 MHz clock: no memory traffic, one CU slot).
 * Measurement aid: the paced stand-in for an xGMI transfer of known duration when a rank's fetch/compute overlap is
 * timed on a single GPU (HNH_PACE_LINK_GBPS, tools/overlap_probe.py). */
int hnh_stream_delay_us(hnh_ctx* ctx, int stream, double microseconds);
/* The same stand-in WITH the memory traffic and the compute units a transfer costs its receiver and sender: `nslices` slices of
 * `slice_bytes` are copied src -> dst_base + k * slice_bytes (k = 0 .. nslices-1; every slice reads the same source, as a rank's
 * block goes to all its peers) by `wgs_per_slice` workgroups each, throttled so that a slice takes `microseconds` (the modelled
 * link time).  HNH_PACE_COPY in the overlap measurement. */
int hnh_stream_paced_copy(hnh_ctx* ctx, int stream, void* dst_base, const void* src, size_t slice_bytes, int nslices,
                          double microseconds, int wgs_per_slice);

/* A transfer of modelled duration that INCLUDES the real work enqueued between the two marks (the loopback transport's copy):
 * begin stamps the device clock on `stream`, end holds the stream until `microseconds` have passed since that stamp — the
 * transfer takes max(real work, modelled time), as a link that is slower than the local copy would. */
int hnh_stream_pace_begin(hnh_ctx* ctx, int stream);
int hnh_stream_pace_end(hnh_ctx* ctx, int stream, double microseconds);

#ifdef __cplusplus
}
#endif
#endif /* HNH_MEASUREMENT_AIDS_H */
