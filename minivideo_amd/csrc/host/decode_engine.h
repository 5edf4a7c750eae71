// decode_engine.h -- the pipelined whole-path decoder behind mvhp_engine_* (include/minivideo_hotpath.h) and
// minivideo_decode(): entropy threads -> page-locked chunks -> H2D -> batched reconstruction -> D2H -> sink.
//
// Replaces, as one pipeline, the reference's serial loop: NAL loop h264.c:76-188 -> decode_slice h264_slice.c:64-109
// -> macroblock loop h264_slice.c:1046-1139 -> export_idr export.c:618-767.
//
// The engine is plain C++ over a small table of device operations (DeviceApi), so that the threading can be built and
// checked on a CPU-only box against a stub device (tools/engine_harness.cpp, tests/test_engine_harness.py: ThreadSanitizer + AddressSanitizer); the product's table is HIP
// (csrc/hip/hotpath_abi.hip) and there is no CPU implementation of it in the library.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

#include "minivideo_hotpath.h"

struct mvhp_stream;

namespace mvengine {

struct DevCtx;   // one device context: a reconstruction context plus its upload / compute / download queues

// Every operation blocks its calling thread until the device has finished it; the engine runs one thread per queue
// and context, which is what overlaps upload(k+1), kernel(k) and download(k-1).  *ms = device-side duration.
struct DeviceApi {
    int    (*device_count)();
    void  *(*host_alloc)(size_t bytes);           // page-locked
    void   (*host_free)(void *p);
    DevCtx *(*ctx_create)(int device, std::string &err);
    void   (*ctx_destroy)(DevCtx *c);
    void  *(*dev_alloc)(DevCtx *c, size_t bytes);
    void   (*dev_free)(DevCtx *c, void *p);
    size_t (*dev_free_bytes)(DevCtx *c);
    // n pieces in one go (one piece per picture: compact pictures have individual sizes)
    int    (*h2d)(DevCtx *c, int n, void *const *d_dst, const void *const *h_src, const size_t *bytes, float *ms, std::string &err);
    // n pieces in one go (the planes and the RGB of an output chunk: one wait instead of two)
    int    (*d2h)(DevCtx *c, int n, void *const *h_dst, const void *const *d_src, const size_t *bytes, float *ms, std::string &err);
    // compact pictures (`stride` bytes apart) -> packed records in d_packed (scratch) -> planes (+ RGB)
    int    (*recon)(DevCtx *c, const mvhp_stream_params_t *p, const void *d_compact, size_t stride, void *d_packed,
                    int n_pictures, uint8_t *d_yuv, uint8_t *d_rgb, float *ms, int *layout, int *waves, std::string &err);
    // optional (may be NULL): `sets` x 4 batch buffers {compact, records, planes, RGB} inside one arena, records / planes / RGB
    // each in a group of the device's memory regions of its own (mvhp_placed_alloc_sets); ptrs[s * 4 + i]; nullptr = failed
    void  *(*placed_alloc)(DevCtx *c, int sets, const size_t bytes[4], void **ptrs);
    void   (*placed_free)(DevCtx *c, void *arena);
};

class Engine;

Engine *engine_create(const DeviceApi &api, const mvhp_engine_opts_t *opts, std::string &err);
void    engine_destroy(Engine *e);
void    engine_release_picture(Engine *e, int seq);
int     effective_cores();   // hardware threads cut down to the container's CPU quota
int     engine_decode(Engine *e, const mvhp_stream &s, const int *order, int n_order, int wanted, int out_mask,
                      mvhp_picture_sink_t sink, void *user, mvhp_decode_stats_t *stats, std::string &err);

} // namespace mvengine

// the product's device table (HIP); defined in csrc/hip/hotpath_abi.hip
const mvengine::DeviceApi &mvhp_hip_device_api();
