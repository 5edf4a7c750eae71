// hotpath_abi.hip -- C-ABI (include/minivideo_hotpath.h) over the HIP kernels.
// There is deliberately NO CPU fallback here: without a usable HIP device every
// reconstruction entry point fails with MVHP_FAILURE and a message.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>

#include "decode_engine.h"
#include "minivideo_hotpath.h"
#include "recon_kernels.h"

namespace {

thread_local char g_err[512] = "";

void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    fprintf(stderr, "[minivideo-hip] %s\n", g_err);
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MVHP_FAILURE;                                                       \
        }                                                                              \
    } while (0)

bool params_ok(const mvhp_stream_params_t *p)
{
    if (!p) return false;
    if (p->width_mbs == 0 || p->height_mbs == 0) return false;
    if (p->width_mbs > 1024 || p->height_mbs > 1024) return false; // LDS line buffer: 32 B per MB column
    if (p->chroma_qp_index_offset < -12 || p->chroma_qp_index_offset > 12) return false;
    if (p->second_chroma_qp_index_offset < -12 || p->second_chroma_qp_index_offset > 12) return false;
    return true;
}

} // namespace

struct mvhp_ctx {
    int          device;
    hipStream_t  stream;
    uint32_t    *d_err;
    int          waves;       // 0 = auto
    int          layout;      // MVHP_LAYOUT_*
    int          fused_color; // 1 = RGB written by the reconstruction kernel's epilogue (default)
    int          n_cus;
    size_t       max_lds;
    int          last_layout, last_waves;   // what the last reconstruction launch used
    // wide launches (MVHP_LAYOUT_WIDE / QUAD_WIDE): ticket counter, seam granules, the tag of the last launch.  One set per
    // context: two wide launches of one context never overlap (a launch on another stream waits for the previous one).
    uint32_t    *d_ticket;
    uint32_t     ticket_base;               // value of *d_ticket once every launch issued so far has run
    uint32_t     wide_epoch;
    void        *d_seam;
    size_t       d_seam_bytes;
    hipEvent_t   wide_done;                 // recorded behind the last wide launch
    hipStream_t  wide_stream;               // the stream it ran on
    int          ticket_skew_once;          // test hook (mvhp_debug_skew_next_ticket_base): added to the next wide launch's base
    // staging for the host convenience path
    void        *d_packed;
    size_t       d_packed_bytes;
    uint8_t     *d_yuv;
    size_t       d_yuv_bytes;
    uint8_t     *d_rgb;
    size_t       d_rgb_bytes;
};

extern "C" {

MVHP_EXPORT const char *mvhp_last_error(void) { return g_err; }

MVHP_EXPORT size_t mvhp_packed_frame_bytes(const mvhp_stream_params_t *p)
{
    return p ? (size_t)p->width_mbs * p->height_mbs * MVHP_MB_BYTES : 0;
}
MVHP_EXPORT size_t mvhp_yuv_frame_bytes(const mvhp_stream_params_t *p)
{
    return p ? (size_t)p->width_mbs * p->height_mbs * 384 : 0;
}
MVHP_EXPORT size_t mvhp_rgb_frame_bytes(const mvhp_stream_params_t *p)
{
    return p ? (size_t)p->width_mbs * p->height_mbs * 768 : 0;
}

MVHP_EXPORT int mvhp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

MVHP_EXPORT int mvhp_create(int device, mvhp_ctx_t **out)
{
    if (!out) return MVHP_FAILURE;
    *out = nullptr;
    int n = mvhp_device_count();
    if (n <= 0) {
        set_err("no HIP device available: the reconstruction path has no CPU fallback");
        return MVHP_FAILURE;
    }
    if (device < 0 || device >= n) {
        set_err("device %d out of range (0..%d)", device, n - 1);
        return MVHP_FAILURE;
    }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    mvhp_ctx *c = new mvhp_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->fused_color = 1;
    if (const char *e = getenv("MINIVIDEO_LAYOUT")) { // tuning / test override, speed only
        if (!strcmp(e, "rows")) c->layout = MVHP_LAYOUT_ROWS;
        else if (!strcmp(e, "quad")) c->layout = MVHP_LAYOUT_QUAD;
        else if (!strcmp(e, "oct")) c->layout = MVHP_LAYOUT_OCT;
        else if (!strcmp(e, "wide")) c->layout = MVHP_LAYOUT_WIDE;
        else if (!strcmp(e, "quad_wide")) c->layout = MVHP_LAYOUT_QUAD_WIDE;
        else if (!strcmp(e, "pipe")) c->layout = MVHP_LAYOUT_PIPE;
        else if (!strcmp(e, "pipe1")) c->layout = MVHP_LAYOUT_PIPE1;
    }
    c->n_cus = prop.multiProcessorCount;
    c->max_lds = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 65536;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&c->d_err, 2 * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(c->d_err, 0, 2 * sizeof(uint32_t)) != hipSuccess ||
        hipEventCreateWithFlags(&c->wide_done, hipEventDisableTiming) != hipSuccess) {
        set_err("context allocation failed on device %d", device);
        delete c;
        return MVHP_FAILURE;
    }
    c->d_ticket = c->d_err + 1;   // (the error word and the ticket counter share one small allocation)
    *out = c;
    return MVHP_SUCCESS;
}

MVHP_EXPORT void mvhp_destroy(mvhp_ctx_t *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->d_packed) hipFree(c->d_packed);
    if (c->d_yuv) hipFree(c->d_yuv);
    if (c->d_rgb) hipFree(c->d_rgb);
    if (c->d_err) hipFree(c->d_err);
    if (c->d_seam) hipFree(c->d_seam);
    if (c->wide_done) hipEventDestroy(c->wide_done);
    hipStreamDestroy(c->stream);
    delete c;
}

MVHP_EXPORT int mvhp_set_waves_per_picture(mvhp_ctx_t *c, int waves)
{
    if (!c || !(waves == 0 || waves == 1 || waves == 2 || waves == 4 || waves == 6 || waves == 8 || waves == 12 || waves == 16)) return MVHP_FAILURE;
    c->waves = waves;
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_set_layout(mvhp_ctx_t *c, int layout)
{
    if (!c || layout < MVHP_LAYOUT_AUTO || layout >= MVHP_LAYOUT_COUNT) return MVHP_FAILURE;
    c->layout = layout;
    return MVHP_SUCCESS;
}

MVHP_EXPORT void *mvhp_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {   // portable: every device may DMA from / to it
        set_err("hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

MVHP_EXPORT void mvhp_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

MVHP_EXPORT int mvhp_set_fused_color(mvhp_ctx_t *c, int on)
{
    if (!c) return MVHP_FAILURE;
    c->fused_color = on ? 1 : 0;
    return MVHP_SUCCESS;
}

// Which kernel form a batch runs on: speed only, results identical.
//   Few pictures: ONE picture (one group of four) spread over several workgroups, bands of four macroblock rows each
//   ("wide" forms: SURVEY 7 step 5's "grid = F x PicHeightInMbs wavefronts"); many pictures: one workgroup per group of
//   four / eight.  Measured on 1080p (tools/layout_crossover.py, profiles/r04l_crossover_{base,high}.log; ms per launch):
//     Baseline      1      4     16     64    128    256    512    768   1024
//     rows        2.45   2.47   2.49   2.53   2.55   2.58   5.00     -    8.86    one workgroup per picture (rounds 1-3)
//     quad        2.73   3.69   3.69   3.71   3.71   3.75   3.96   4.27   4.72    ... per four pictures
//     wide        1.00   1.00   1.01   1.21   1.50   2.34   4.16   6.06   8.00    one picture in 17 bands
//     quad_wide   1.05   1.38   1.38   1.43   1.61   1.93   2.80   3.89   5.05    four pictures in 17 bands
//     pipe        0.59   0.78   0.79   0.95   1.23   1.88   3.28   4.72   6.27    ... three waves per row: residuals / luma / chroma + output
//     pipe1       0.64   0.64   0.65   0.92   1.44   2.61   5.01     -      -     one picture per wavefront, three waves per row
//     High        1      4     16     64    128    256    512    768   1024
//     wide        1.03   1.04   1.05   1.30   1.62   2.50   4.46   6.49   8.57
//     quad_wide   1.18   1.91   1.91   1.99   2.20   2.61   3.70   5.12   6.47    (the four pictures of a wavefront run their
//     pipe        0.67   1.18   1.19   1.41   1.85   2.72   4.69     -      -      three luma paths one after the other)
//     pipe1       0.60   0.60   0.63   0.93   1.50   2.74   5.31     -      -
//     quad        3.03   5.17   5.17   5.20   5.20   5.21   5.24   5.34   5.60
//   ONE Baseline picture: the quarters of a wavefront hold the same picture (no divergence): pipe (0.57 against pipe1's 0.65 ms;
//   at three pictures 0.73 against 0.65).  pipe1 has no lock
//   step at all and its Intra4x4 chain takes ten dependent steps instead of sixteen, but needs three resident waves per ROW.
//   720p and 2160p: profiles/r04l_crossover_{high720,base2160,high2160,high2160b}.log.  Small batches with slices / scaling
//   matrices: pipe1 (it reconstructs them as the one-picture kernel does), larger ones wide.
static int pick_layout(const mvhp_ctx *c, const mvhp_stream_params_t *p, int n_frames)
{
    int layout = c->layout;
    // pictures of several slices and scaling matrices (MVHP_STREAM_SPEC streams): the one-picture kernel,
    // where a neighbour's availability is a per-wavefront scalar and LevelScale is a table in LDS -- whatever was asked for;
    // in bands at every batch size (2.44 against 2.58 ms at 256 pictures, 8.5 against 8.9 at 1024) unless "rows" is forced
    const bool pipe1_fits = mvhp::recon_pipe1_lds_bytes((int)p->width_mbs, 1) <= c->max_lds;
    if (p->flags & (MVHP_PARAM_SLICES | MVHP_PARAM_SCALING)) {
        if (layout == MVHP_LAYOUT_ROWS || layout == MVHP_LAYOUT_WIDE) return layout;
        if (layout == MVHP_LAYOUT_PIPE1) return pipe1_fits ? MVHP_LAYOUT_PIPE1 : MVHP_LAYOUT_WIDE;
        return (pipe1_fits && (double)n_frames * (double)p->height_mbs <= 40.0 * c->n_cus) ? MVHP_LAYOUT_PIPE1 : MVHP_LAYOUT_WIDE;
    }
    if (layout == MVHP_LAYOUT_AUTO) {
        const double cus = (double)c->n_cus;
        const double row_waves = (double)n_frames * (double)p->height_mbs;
        const bool may8 = (p->flags & MVHP_PARAM_MAY_HAVE_8X8) != 0;
        const bool pipe_fits = mvhp::recon_pipe_lds_bytes((int)p->width_mbs, 1) <= c->max_lds;
        // round 4, after the wave priorities went in (profiles/r04l_crossover_*.log: 720p, 1080p and 2160p, both profiles): what
        // decides between the one-picture forms is ROW-WAVES (three waves per row have to be resident), what decides between the
        // four-picture forms is PICTURES (a round of the unbanded kernel is 4 x CUs pictures whatever their size):
        //   Baseline  pipe (1 picture) | pipe1 up to 18 x CUs row-waves | pipe up to 76 x CUs row-waves (rows of 240: 1.15 x CUs pictures) | quad_wide | round model
        //   High                       pipe1 up to 46 (rows of > 160 macroblocks: 40) x CUs row-waves | wide up to 76 x CUs row-waves (rows of 240: 1.2 x CUs pictures) | quad_wide | round model
        // quad_wide against a first round of quad: 0.84 x 4 x CUs pictures at 120 macroblocks per row (720p: 0.80), 0.65 at 240
        const double wide_rows = fmax(0.0, ((double)p->width_mbs - 120.0) / 120.0);   // 0 at 1080p, 1 at 2160p
        const double qw_share = fmin(0.84, fmax(0.60, 0.84 - (may8 ? 0.19 : 0.06) * wide_rows));
        // four pictures per wavefront in bands against the forms below them: 76 x CUs row-waves on rows of up to 160 macroblocks
        // (720p: 450 pictures, 1080p: 300), at most 2 x CUs pictures; on longer rows 1.15 / 1.2 x CUs pictures (r04r_grid*.log)
        const bool below_qw = (p->width_mbs <= 160) ? (row_waves <= 76.0 * cus && n_frames <= 2.0 * cus) : (n_frames <= (may8 ? 1.2 : 1.15) * cus);
        if (pipe_fits && !may8 && n_frames <= 1) {
            layout = MVHP_LAYOUT_PIPE;
        } else if (pipe1_fits && row_waves <= (may8 ? (p->width_mbs <= 160 ? 46.0 : 40.0) : 18.0) * cus) {
            layout = MVHP_LAYOUT_PIPE1;
        } else if (pipe_fits && !may8 && below_qw) {
            layout = MVHP_LAYOUT_PIPE;
        } else if (may8 ? below_qw : (!pipe_fits && row_waves <= 34.0 * cus)) {
            layout = MVHP_LAYOUT_WIDE;
        } else if (n_frames <= qw_share * 4.0 * cus) {
            layout = MVHP_LAYOUT_QUAD_WIDE;
        } else {
            // A launch is a number of "rounds" of one workgroup per CU (the batch kernels fill a CU with one workgroup), in
            // units of one full round of the four-picture kernel (5.4 ms for 4 * CUs pictures of 1080p): the four-picture
            // kernel 0.77 with one workgroup on the device .. 1.0 with all CUs busy; the eight-picture kernel 1.48 .. 1.85
            // (8 * CUs pictures; round 4, with the priorities: 1.45 .. 1.75).  (1100 pictures: quad 8.5 / oct 7.4 ms, 2048: 8.7 / 7.9, 2560: 13.7 / 16.1.)
            auto rounds = [&](double per_round, double lo, double hi) {
                const double full = floor(n_frames / per_round), rem = n_frames - full * per_round;
                return full * hi + (rem > 0 ? lo + (hi - lo) * rem / per_round : 0.0);
            };
            // (2160p High: one 16-wave workgroup per CU, a partial round costs a whole one: 1300 pictures 40.4 ms = 2 x 20)
            const double t_quad = rounds(4 * cus, (may8 && p->width_mbs > 160) ? 1.0 : 0.77, 1.0);
            const bool oct_fits = mvhp::recon_oct_lds_bytes((int)p->width_mbs, 8) <= c->max_lds;   // with six waves it loses to quad
            const double t_oct = oct_fits ? (may8 ? rounds(8 * cus, 1.8, 2.0) : rounds(8 * cus, 1.45, 1.75)) : 1e30;   // (High: 10.3 against 5.3 ms per round)
            // ... and the banded four-picture form, whose time is linear in the pictures (8-row bands at these sizes): between one
            // and two rounds it beats both (1100 x 1080p: 5.15 ms against 8.5 / 7.4; profiles/r04q_crossover_big*.log); per round
            // 1.0 (Baseline) / 1.05 (High) at 120 macroblocks per row, 1.14 / 1.49 at 240
            const double t_qw = (n_frames / (4.0 * cus)) * (may8 ? 1.05 + 0.44 * wide_rows : 1.0 + 0.14 * wide_rows);
            layout = (n_frames > 4 * cus && t_qw < t_quad && t_qw < t_oct) ? MVHP_LAYOUT_QUAD_WIDE : (t_oct < t_quad) ? MVHP_LAYOUT_OCT : MVHP_LAYOUT_QUAD;
        }
    }
    // the batch kernels address a workgroup's pictures with 32-bit offsets and keep one line buffer per picture in LDS
    const size_t mbs = (size_t)p->width_mbs * p->height_mbs;
    if (layout == MVHP_LAYOUT_OCT && (mbs > ((size_t)1 << 19) || mvhp::recon_oct_lds_bytes((int)p->width_mbs, 4) > c->max_lds))
        layout = MVHP_LAYOUT_QUAD;
    if (layout == MVHP_LAYOUT_QUAD && mvhp::recon_quad_lds_bytes((int)p->width_mbs, 4) > c->max_lds) layout = MVHP_LAYOUT_ROWS;
    if (layout == MVHP_LAYOUT_PIPE1 && mvhp::recon_pipe1_lds_bytes((int)p->width_mbs, 1) > c->max_lds) layout = MVHP_LAYOUT_WIDE;
    if (layout == MVHP_LAYOUT_PIPE && (mbs > ((size_t)1 << 20) || mvhp::recon_pipe_lds_bytes((int)p->width_mbs, 1) > c->max_lds))
        layout = MVHP_LAYOUT_QUAD_WIDE;
    if (layout == MVHP_LAYOUT_QUAD_WIDE && (mbs > ((size_t)1 << 20) || mvhp::recon_quad_lds_bytes((int)p->width_mbs, 4) > c->max_lds))
        layout = MVHP_LAYOUT_WIDE;
    return layout;
}

static int pick_waves(const mvhp_ctx *c, const mvhp_stream_params_t *p, int n_frames, int layout)
{
    int nw = c->waves;
    if (layout == MVHP_LAYOUT_PIPE || layout == MVHP_LAYOUT_PIPE1) {
        // rows per band (three wavefronts each), built for 1, 2 and 4: 4 unless asked; the one-picture form at the upper end of its
        // range (more than 0.44 x CUs pictures) packs better with single rows (150 x 1080p High: 1.61 against 1.68 ms; 64: the same;
        // 16: 0.77 against 0.63 -- profiles/r04o_pipe1_rows.log), but not on rows of 240 macroblocks, where a seam per row costs
        // more (75 x 2160p: 3.58 against 3.29 -- profiles/r04q_crossover_pipe1_rows.log)
        if (nw == 0) nw = (layout == MVHP_LAYOUT_PIPE1 && n_frames > 0.44 * c->n_cus && p->width_mbs <= 160) ? 1 : 4;
        nw = (nw >= 4) ? 4 : (nw >= 2 ? 2 : 1);
        while (nw > 1 && (layout == MVHP_LAYOUT_PIPE ? mvhp::recon_pipe_lds_bytes((int)p->width_mbs, nw)
                                                      : mvhp::recon_pipe1_lds_bytes((int)p->width_mbs, nw)) > c->max_lds) nw /= 2;
        return nw;
    }
    if (layout == MVHP_LAYOUT_WIDE) return 4;   // rows per band (built for 4: the finest grain, 17 bands per 1080p picture)
    if (layout == MVHP_LAYOUT_QUAD_WIDE) {
        // rows per band, built for 4 and 8: 8-wave workgroups fit two to a CU (LDS) = 16 waves, 4-wave ones three = 12;
        // the finer grain is the faster one on Baseline at every batch size measured (512 x 1080p: 2.83 against 2.95 ms), the
        // coarser one on High from ~1.5 x CUs pictures on (640 pictures: 4.13 against 4.28; profiles/r04o_qw48_*.log)
        if (nw == 0) nw = (p->width_mbs <= 160 && (((p->flags & MVHP_PARAM_MAY_HAVE_8X8) && n_frames >= 1.5 * c->n_cus) || n_frames > 3.4 * c->n_cus)) ? 8 : 4;   // (rows of 240: 4 everywhere)
        nw = (nw >= 8) ? 8 : 4;
        if (nw == 8 && mvhp::recon_quad_lds_bytes((int)p->width_mbs, 8) > c->max_lds) nw = 4;
        return nw;
    }
    if (layout == MVHP_LAYOUT_OCT) {
        // speed only: built for 4, 6 and 8 waves; one workgroup per CU (LDS)
        static const int opts[3] = {8, 6, 4};
        if (nw == 0) nw = 8;
        for (int k = 0; k < 3; k++) {
            const int o = opts[k];
            if (o > nw) continue;
            if (o > 4 && ((o + 1) / 2 >= (int)p->height_mbs || mvhp::recon_oct_lds_bytes((int)p->width_mbs, o) > c->max_lds)) continue;
            return o;
        }
        return 4;
    }
    if (layout == MVHP_LAYOUT_QUAD) {
        // speed only: built for 4, 6, 8, 12 and 16 waves; 8-wave workgroups fit two to a CU (LDS, 128 VGPRs) = 16 waves
        // per CU; when only one workgroup per CU will be resident (few workgroups, or wide pictures whose four line
        // buffers leave LDS for one), it should bring the 16 waves itself
        static const int opts[5] = {16, 12, 8, 6, 4};
        if (nw == 0) {
            const int groups = (n_frames + 3) / 4;
            const bool two_fit = 2 * mvhp::recon_quad_lds_bytes((int)p->width_mbs, 8) <= c->max_lds;
            nw = (groups >= 2 * c->n_cus && two_fit) ? 8 : 16;
        }
        for (int k = 0; k < 5; k++) {
            const int o = opts[k];
            if (o > nw) continue;
            if (o > 4 && ((o + 1) / 2 >= (int)p->height_mbs || mvhp::recon_quad_lds_bytes((int)p->width_mbs, o) > c->max_lds)) continue;
            return o;
        }
        return 4;
    }
    // speed only (DESIGN.md "waves per picture"): 8-wave workgroups fit three to a CU (LDS) = 24 waves/CU,
    // 16-wave workgroups one to a CU; small batches need the wider workgroup to occupy the chip.
    if (nw == 0) nw = (n_frames >= 384) ? 8 : 16;
    if (nw < 4 || nw == 6) nw = 4;   // (1 and 2 are rows per band of the pipe form)
    if (nw == 12) nw = 8;
    while (nw > 4 && (nw / 2) >= (int)p->height_mbs) nw /= 2;
    while (nw > 4 && mvhp::recon_lds_bytes((int)p->width_mbs, nw) > c->max_lds) nw /= 2;
    return nw;
}

// Everything a wide launch needs besides the batch: seam granules for its band boundaries (grown on demand, zeroed once: a
// granule counts when its tag equals the launch's epoch, and epochs never repeat), the ticket base, ordering behind the
// context's previous wide launch when that ran on another stream (they share the counter and the seams).
static int wide_prepare(mvhp_ctx *c, mvhp::ReconArgs &a, size_t seam_bytes, uint32_t units, hipStream_t st)
{
    if (c->wide_stream && c->wide_stream != st) HIP_TRY(hipStreamWaitEvent(st, c->wide_done, 0));
    if (seam_bytes > c->d_seam_bytes) {
        if (c->d_seam) {
            HIP_TRY(hipEventSynchronize(c->wide_done));   // (the previous wide launch may still read the old one)
            HIP_TRY(hipFree(c->d_seam));
            c->d_seam = nullptr;
            c->d_seam_bytes = 0;
        }
        const size_t want = seam_bytes + seam_bytes / 4;
        HIP_TRY(hipMalloc(&c->d_seam, want));
        c->d_seam_bytes = want;
        HIP_TRY(hipMemsetAsync(c->d_seam, 0, want, st));
    }
    if (++c->wide_epoch == 0) {   // 2^32 launches later: tags would repeat
        if (c->d_seam) HIP_TRY(hipMemsetAsync(c->d_seam, 0, c->d_seam_bytes, st));
        c->wide_epoch = 1;
    }
    a.wide_ticket = c->d_ticket;
    a.wide_base = c->ticket_base + (uint32_t)c->ticket_skew_once;
    c->ticket_skew_once = 0;
    a.wide_epoch = c->wide_epoch;
    a.seam = (unsigned long long *)c->d_seam;
    (void)units;
    c->wide_stream = st;
    return MVHP_SUCCESS;
}

static int launch_all(mvhp_ctx *c, const mvhp_stream_params_t *p, const void *d_packed, int n_frames,
                      uint8_t *d_yuv, uint8_t *d_rgb, hipStream_t st, bool recon, bool color)
{
    if (recon) {
        mvhp::ReconArgs a;
        a.packed = (const uint8_t *)d_packed;
        a.yuv = d_yuv;
        a.rgb = (c->fused_color && color) ? d_rgb : nullptr;
        a.err = c->d_err;
        a.width_mbs = (int)p->width_mbs;
        a.height_mbs = (int)p->height_mbs;
        a.cqp_off_cb = p->chroma_qp_index_offset;
        a.cqp_off_cr = p->second_chroma_qp_index_offset;
        a.n_frames = n_frames;
        a.dc_shift_from = (p->flags & MVHP_PARAM_SPEC_LUMA_DC) ? 36 : 37;
        a.slices = (p->flags & MVHP_PARAM_SLICES) ? 1 : 0;
        a.scaling = (p->flags & MVHP_PARAM_SCALING) ? 1 : 0;
        if (a.scaling) {
            memcpy(a.weights, p->scaling4, 48);
            memcpy(a.weights + 48, p->scaling8, 64);
        } else {
            memset(a.weights, 16, sizeof(a.weights));
        }
        a.wide_ticket = nullptr;
        a.wide_base = a.wide_epoch = 0;
        a.seam = nullptr;
        const int layout = pick_layout(c, p, n_frames);
        const int nw = pick_waves(c, p, n_frames, layout);
        c->last_layout = layout;
        c->last_waves = nw;
        if (layout == MVHP_LAYOUT_WIDE) {
            if (mvhp::recon_lds_bytes(a.width_mbs, nw) > c->max_lds) {
                set_err("picture too wide for the LDS line buffer (%u macroblocks)", p->width_mbs);
                return MVHP_UNSUPPORTED;
            }
            const int rc = wide_prepare(c, a, mvhp::recon_wide_seam_bytes(a.width_mbs, a.height_mbs, n_frames, nw),
                                        (uint32_t)n_frames * (uint32_t)((a.height_mbs + nw - 1) / nw), st);
            if (rc != MVHP_SUCCESS) return rc;
            HIP_TRY(mvhp::launch_recon_wide(a, n_frames, nw, st));
            c->ticket_base += (uint32_t)n_frames * (uint32_t)((a.height_mbs + nw - 1) / nw);   // every workgroup takes one ticket
            HIP_TRY(hipEventRecord(c->wide_done, st));
        } else if (layout == MVHP_LAYOUT_PIPE) {
            const int rc = wide_prepare(c, a, mvhp::recon_wide_seam_bytes(a.width_mbs, a.height_mbs, n_frames, nw), 0, st);
            if (rc != MVHP_SUCCESS) return rc;
            HIP_TRY(mvhp::launch_recon_pipe(a, nw, st));
            c->ticket_base += (uint32_t)((n_frames + 3) / 4) * (uint32_t)((a.height_mbs + nw - 1) / nw);
            HIP_TRY(hipEventRecord(c->wide_done, st));
        } else if (layout == MVHP_LAYOUT_PIPE1) {
            const int rc = wide_prepare(c, a, mvhp::recon_wide_seam_bytes(a.width_mbs, a.height_mbs, n_frames, nw), 0, st);
            if (rc != MVHP_SUCCESS) return rc;
            HIP_TRY(mvhp::launch_recon_pipe1(a, nw, st));
            c->ticket_base += (uint32_t)n_frames * (uint32_t)((a.height_mbs + nw - 1) / nw);
            HIP_TRY(hipEventRecord(c->wide_done, st));
        } else if (layout == MVHP_LAYOUT_QUAD_WIDE) {
            const int rc = wide_prepare(c, a, mvhp::recon_wide_seam_bytes(a.width_mbs, a.height_mbs, n_frames, nw), 0, st);
            if (rc != MVHP_SUCCESS) return rc;
            HIP_TRY(mvhp::launch_recon_quad_wide(a, nw, st));
            c->ticket_base += (uint32_t)((n_frames + 3) / 4) * (uint32_t)((a.height_mbs + nw - 1) / nw);
            HIP_TRY(hipEventRecord(c->wide_done, st));
        } else if (layout == MVHP_LAYOUT_OCT) {
            HIP_TRY(mvhp::launch_recon_oct(a, nw, st));
        } else if (layout == MVHP_LAYOUT_QUAD) {
            HIP_TRY(mvhp::launch_recon_quad(a, nw, st));
        } else {
            if (mvhp::recon_lds_bytes(a.width_mbs, nw) > c->max_lds) {
                set_err("picture too wide for the LDS line buffer (%u macroblocks)", p->width_mbs);
                return MVHP_UNSUPPORTED;
            }
            HIP_TRY(mvhp::launch_recon(a, n_frames, nw, st));
        }
    }
    if (color && d_rgb && !(recon && c->fused_color)) {
        mvhp::ColorArgs ca;
        ca.yuv = d_yuv;
        ca.rgb = d_rgb;
        ca.width_mbs = (int)p->width_mbs;
        ca.height_mbs = (int)p->height_mbs;
        ca.n_frames = n_frames;
        HIP_TRY(mvhp::launch_color(ca, st));
    }
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_recon_batch_dev(mvhp_ctx_t *c, const mvhp_stream_params_t *p, const void *d_packed,
                                     int n_frames, uint8_t *d_yuv, uint8_t *d_rgb, void *stream)
{
    if (!c || !params_ok(p) || !d_packed || !d_yuv || n_frames <= 0) {
        set_err("mvhp_recon_batch_dev: invalid argument");
        return MVHP_FAILURE;
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    return launch_all(c, p, d_packed, n_frames, d_yuv, d_rgb, st, true, true);
}

MVHP_EXPORT int mvhp_expand_compact_dev(mvhp_ctx_t *c, const mvhp_stream_params_t *p, const void *d_compact, size_t stride,
                                        int n_pictures, void *d_packed, void *stream)
{
    if (!c || !params_ok(p) || !d_compact || !d_packed || n_pictures <= 0 || (stride & 3) ||
        stride < (size_t)p->width_mbs * p->height_mbs * 4 + MVHP_MB_HEADER_BYTES) {
        set_err("mvhp_expand_compact_dev: invalid argument");
        return MVHP_FAILURE;
    }
    HIP_TRY(hipSetDevice(c->device));
    mvhp::ExpandArgs a;
    a.compact = (const uint8_t *)d_compact;
    a.stride = stride;
    a.packed = (uint8_t *)d_packed;
    a.mbs = (int)(p->width_mbs * p->height_mbs);
    a.n_pictures = n_pictures;
    HIP_TRY(mvhp::launch_expand(a, stream ? (hipStream_t)stream : c->stream));
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_recon_stages_dev(mvhp_ctx_t *c, const mvhp_stream_params_t *p, const void *d_packed,
                                      int n_frames, uint8_t *d_yuv, uint8_t *d_rgb, void *stream, int stages)
{
    if (!c || !params_ok(p) || !d_packed || !d_yuv || n_frames <= 0 || (stages & ~3) || !stages) {
        set_err("mvhp_recon_stages_dev: invalid argument");
        return MVHP_FAILURE;
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    return launch_all(c, p, d_packed, n_frames, d_yuv, d_rgb, st, (stages & 1) != 0, (stages & 2) != 0);
}

static int ensure(void **ptr, size_t *have, size_t need)
{
    if (*have >= need) return MVHP_SUCCESS;
    if (*ptr) hipFree(*ptr);
    *ptr = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(ptr, need));
    *have = need;
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_recon_batch_host(mvhp_ctx_t *c, const mvhp_stream_params_t *p, const void *h_packed,
                                      int n_frames, uint8_t *h_yuv, uint8_t *h_rgb)
{
    if (!c || !params_ok(p) || !h_packed || !h_yuv || n_frames <= 0) {
        set_err("mvhp_recon_batch_host: invalid argument");
        return MVHP_FAILURE;
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t pb = mvhp_packed_frame_bytes(p) * n_frames;
    const size_t yb = mvhp_yuv_frame_bytes(p) * n_frames;
    const size_t rb = mvhp_rgb_frame_bytes(p) * n_frames;
    if (ensure(&c->d_packed, &c->d_packed_bytes, pb) != MVHP_SUCCESS) return MVHP_FAILURE;
    if (ensure((void **)&c->d_yuv, &c->d_yuv_bytes, yb) != MVHP_SUCCESS) return MVHP_FAILURE;
    if (h_rgb && ensure((void **)&c->d_rgb, &c->d_rgb_bytes, rb) != MVHP_SUCCESS) return MVHP_FAILURE;
    HIP_TRY(hipMemsetAsync(c->d_err, 0, sizeof(uint32_t), c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_packed, h_packed, pb, hipMemcpyHostToDevice, c->stream));
    int rc = launch_all(c, p, c->d_packed, n_frames, c->d_yuv, h_rgb ? c->d_rgb : nullptr, c->stream, true, true);
    if (rc != MVHP_SUCCESS) return rc;
    HIP_TRY(hipMemcpyAsync(h_yuv, c->d_yuv, yb, hipMemcpyDeviceToHost, c->stream));
    if (h_rgb) HIP_TRY(hipMemcpyAsync(h_rgb, c->d_rgb, rb, hipMemcpyDeviceToHost, c->stream));
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, c->d_err, sizeof(err), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (err) {
        set_err("reconstruction kernel reported error word 0x%x (%s)", err,
                (err & 2u) ? "a workgroup's ticket lay outside the launch" : "row dependency wait timed out");
        return MVHP_FAILURE;
    }
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_sync_check(mvhp_ctx_t *c, void *stream)
{
    if (!c) return MVHP_FAILURE;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, c->d_err, sizeof(err), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (err) {
        set_err("reconstruction kernel reported error word 0x%x (%s)", err,
                (err & 2u) ? "a workgroup's ticket lay outside the launch" : "row dependency wait timed out");
        hipMemset(c->d_err, 0, sizeof(uint32_t));
        return MVHP_FAILURE;
    }
    return MVHP_SUCCESS;
}

/* Test hook (tests/test_gpu_wide.py): the next wide launch of `c` hands out its units `delta` off, i.e. one unit is never
 * reconstructed and one ticket falls outside the launch -- what a corrupted ticket counter would look like.  The launch must END
 * (bounded waits) with the error word set, and mvhp_sync_check() must say so. */
MVHP_EXPORT int mvhp_debug_skew_next_ticket_base(mvhp_ctx_t *c, int delta)
{
    if (!c) return MVHP_FAILURE;
    c->ticket_skew_once = delta;
    return MVHP_SUCCESS;
}

MVHP_EXPORT int mvhp_last_launch_info(const mvhp_ctx_t *c, int *layout, int *waves)
{
    if (!c) return MVHP_FAILURE;
    if (layout) *layout = c->last_layout;
    if (waves) *waves = c->last_waves;
    return MVHP_SUCCESS;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// The decode engine's device table (csrc/host/decode_engine.h): every operation runs on its own HIP stream of the
// context and blocks the calling engine thread until the device has finished it; durations come from HIP events
// recorded on that stream.
// ---------------------------------------------------------------------------------------------------------------
namespace mvengine {
struct DevCtx {
    mvhp_ctx   *c = nullptr;
    hipStream_t up = nullptr, down = nullptr;
    hipEvent_t  ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // begin/end per queue: upload, compute, download
};
} // namespace mvengine

namespace {

using mvengine::DevCtx;

#define ENG_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                               \
            return MVHP_FAILURE;                                                                          \
        }                                                                                                 \
    } while (0)

DevCtx *eng_ctx_create(int device, std::string &err)
{
    mvhp_ctx_t *c = nullptr;
    if (mvhp_create(device, &c) != MVHP_SUCCESS) { err = mvhp_last_error(); return nullptr; }
    DevCtx *d = new DevCtx();
    d->c = c;
    bool ok = hipStreamCreateWithFlags(&d->up, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&d->down, hipStreamNonBlocking) == hipSuccess;
    // blocking-sync events: the engine's copy / launch threads sleep in hipEventSynchronize instead of spinning on a core
    // the entropy threads could use
    for (int i = 0; i < 6 && ok; i++) ok = hipEventCreateWithFlags(&d->ev[i], hipEventBlockingSync) == hipSuccess;
    if (!ok) {
        err = "stream / event creation failed on device " + std::to_string(device);
        for (int i = 0; i < 6; i++) if (d->ev[i]) hipEventDestroy(d->ev[i]);
        if (d->up) hipStreamDestroy(d->up);
        if (d->down) hipStreamDestroy(d->down);
        mvhp_destroy(c);
        delete d;
        return nullptr;
    }
    return d;
}

void eng_ctx_destroy(DevCtx *d)
{
    if (!d) return;
    hipSetDevice(d->c->device);
    hipStreamSynchronize(d->up);
    hipStreamSynchronize(d->down);
    for (int i = 0; i < 6; i++) hipEventDestroy(d->ev[i]);
    hipStreamDestroy(d->up);
    hipStreamDestroy(d->down);
    mvhp_destroy(d->c);
    delete d;
}

void *eng_dev_alloc(DevCtx *d, size_t bytes)
{
    void *p = nullptr;
    if (hipSetDevice(d->c->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}

void eng_dev_free(DevCtx *d, void *p)
{
    hipSetDevice(d->c->device);
    (void)hipFree(p);
}

size_t eng_dev_free_bytes(DevCtx *d)
{
    size_t fr = 0, tot = 0;
    if (hipSetDevice(d->c->device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return 0;
    return fr;
}

int eng_copy(DevCtx *d, hipStream_t st, hipEvent_t e0, hipEvent_t e1, void *dst, const void *src, size_t bytes, hipMemcpyKind kind,
             float *ms, std::string &err)
{
    ENG_TRY(hipSetDevice(d->c->device));
    ENG_TRY(hipEventRecord(e0, st));
    ENG_TRY(hipMemcpyAsync(dst, src, bytes, kind, st));
    ENG_TRY(hipEventRecord(e1, st));
    ENG_TRY(hipEventSynchronize(e1));
    if (ms) ENG_TRY(hipEventElapsedTime(ms, e0, e1));
    return MVHP_SUCCESS;
}

int eng_h2d(DevCtx *d, int n, void *const *dst, const void *const *src, const size_t *bytes, float *ms, std::string &err)
{
    ENG_TRY(hipSetDevice(d->c->device));
    ENG_TRY(hipEventRecord(d->ev[0], d->up));
    for (int i = 0; i < n; i++)
        if (bytes[i]) ENG_TRY(hipMemcpyAsync(dst[i], src[i], bytes[i], hipMemcpyHostToDevice, d->up));
    ENG_TRY(hipEventRecord(d->ev[1], d->up));
    ENG_TRY(hipEventSynchronize(d->ev[1]));
    if (ms) ENG_TRY(hipEventElapsedTime(ms, d->ev[0], d->ev[1]));
    return MVHP_SUCCESS;
}

int eng_d2h(DevCtx *d, int n, void *const *dst, const void *const *src, const size_t *bytes, float *ms, std::string &err)
{
    ENG_TRY(hipSetDevice(d->c->device));
    ENG_TRY(hipEventRecord(d->ev[4], d->down));
    for (int i = 0; i < n; i++)
        if (bytes[i]) ENG_TRY(hipMemcpyAsync(dst[i], src[i], bytes[i], hipMemcpyDeviceToHost, d->down));
    ENG_TRY(hipEventRecord(d->ev[5], d->down));
    ENG_TRY(hipEventSynchronize(d->ev[5]));
    if (ms) ENG_TRY(hipEventElapsedTime(ms, d->ev[4], d->ev[5]));
    return MVHP_SUCCESS;
}

int eng_recon(DevCtx *d, const mvhp_stream_params_t *p, const void *d_compact, size_t stride, void *d_packed, int n, uint8_t *d_yuv,
              uint8_t *d_rgb, float *ms, int *layout, int *waves, std::string &err)
{
    mvhp_ctx *c = d->c;
    if (!params_ok(p) || !d_compact || !d_packed || !d_yuv || n <= 0) { err = "reconstruction: invalid argument"; return MVHP_FAILURE; }
    ENG_TRY(hipSetDevice(c->device));
    ENG_TRY(hipEventRecord(d->ev[2], c->stream));
    if (mvhp_expand_compact_dev(c, p, d_compact, stride, n, d_packed, c->stream) != MVHP_SUCCESS) { err = mvhp_last_error(); return MVHP_FAILURE; }
    const int rc = launch_all(c, p, d_packed, n, d_yuv, d_rgb, c->stream, true, true);
    if (rc != MVHP_SUCCESS) { err = mvhp_last_error(); return rc; }
    ENG_TRY(hipEventRecord(d->ev[3], c->stream));
    uint32_t ew = 0;
    ENG_TRY(hipEventSynchronize(d->ev[3]));   // (sleeps: blocking-sync event)
    ENG_TRY(hipMemcpyAsync(&ew, c->d_err, sizeof(ew), hipMemcpyDeviceToHost, c->stream));
    ENG_TRY(hipStreamSynchronize(c->stream));
    if (ms) ENG_TRY(hipEventElapsedTime(ms, d->ev[2], d->ev[3]));
    if (layout) *layout = c->last_layout;
    if (waves) *waves = c->last_waves;
    if (ew) {
        (void)hipMemsetAsync(c->d_err, 0, sizeof(uint32_t), c->stream);
        err = "reconstruction kernel reported error word (row dependency wait timed out)";
        return MVHP_FAILURE;
    }
    return MVHP_SUCCESS;
}

// the engine's batch buffers from one placed arena (MINIVIDEO_PLACED=1): records / planes / RGB of a batch in three groups of
// the device's memory regions (placement.hip); the compact staging area goes wherever room is left
void *eng_placed_alloc(DevCtx *d, int sets, const size_t bytes[4], void **ptrs)
{
    static const uint8_t any_group[4] = {1, 0, 0, 0};   // {compact, records, planes, RGB}
    void *arena = nullptr;
    int found = 0;
    // an arena sized from the need, not "everything that is free": twice the buffers (room to choose blocks of the
    // right group) + 16 GB, inside the library's cap -- what the device keeps free stays available to ordinary allocations
    // (another picture shape, a second context on the device)
    const size_t blk = (size_t)4 << 30;   // (placement.hip hands out runs of 4-GB blocks)
    size_t need = 0;
    for (int i = 0; i < 4; i++) need += (bytes[i] + blk - 1) / blk * blk;
    need *= (size_t)sets;
    size_t want = need + need / 2 + ((size_t)16 << 30);
    size_t cap = (size_t)200 << 30;
    if (const char *e = getenv("MVHP_PLACED_ARENA_GB")) { const long gb = atol(e); if (gb >= 8 && gb <= 256) cap = (size_t)gb << 30; }
    size_t fr = 0, tot = 0;
    if (hipSetDevice(d->c->device) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess) {
        const size_t reserve = (size_t)24 << 30;
        cap = std::min(cap, fr > reserve ? fr - reserve : (size_t)0);
    }
    want = std::min(want, cap);
    if (want < need) return nullptr;
    if (mvhp_placed_alloc_sets(d->c->device, sets, 4, bytes, any_group, want, ptrs, &arena, nullptr, &found) != MVHP_SUCCESS) return nullptr;
    return arena;
}

void eng_placed_free(DevCtx *d, void *arena)
{
    (void)d;
    mvhp_placed_free(arena);
}

const mvengine::DeviceApi g_hip_api = {
    mvhp_device_count, mvhp_host_alloc, mvhp_host_free, eng_ctx_create, eng_ctx_destroy, eng_dev_alloc, eng_dev_free,
    eng_dev_free_bytes, eng_h2d, eng_d2h, eng_recon, eng_placed_alloc, eng_placed_free,
};

} // namespace

const mvengine::DeviceApi &mvhp_hip_device_api() { return g_hip_api; }
