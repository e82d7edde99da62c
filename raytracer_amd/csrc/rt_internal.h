// rt_internal.h — entry points of the render core that only the library's own host mirror
// uses (not part of the C ABI in include/rt_core.h).
#pragma once
#include <stdint.h>

#include "../../include/rt_core.h"

namespace rti {
// rt_render_device for the whole frame (u8, no statistics) with one difference: a camera the
// scene has not rendered before is issued WITHOUT the host read-back per recursion level when
// the scene has rendered some frame of the same size, depth and mode: grids and buffers are
// sized from that frame's ray-queue lengths plus a quarter, the kernels take the true lengths
// from the device.  The call never blocks then, so frames of a camera sweep overlap like
// repeated frames do.  If a queue outgrows its estimate the image is incomplete:
// estimated_frame_result() says so once the stream has been synchronised, and the caller
// renders the frame again with rt_render_device.
int render_device_estimated(rt_scene*, const rt_camera*, int depth, int mode, int pipeline,
                            uint8_t* d_rgb_u8, void* hip_stream);
// After synchronising `hip_stream`: *complete = 0 if the last frame issued on it was issued on
// an estimate that did not hold, 1 otherwise (its queue lengths are then remembered, so the same
// camera is afterwards issued like any frame seen before).
int estimated_frame_result(rt_scene*, void* hip_stream, int* complete);
// Slot i (0..7) of the scene's own for camera sweeps: a stream, a device image and a pinned host
// image of at least `bytes` bytes each, created on first use, grown when needed, freed with the
// scene.  (The ray queues of the wavefront pipeline are kept per stream: a sweep that brought
// new streams with every call would leave the scene holding queues of streams that are gone;
// and pinning host memory costs milliseconds per call.)
int sweep_slot(rt_scene*, int i, size_t bytes, void** stream, uint8_t** device, uint8_t** pinned);
// How many frames of this scene were issued on an estimate so far, and how many of those had to
// be rendered again.
void estimate_counters(const rt_scene*, long long* issued, long long* missed);
}  // namespace rti
