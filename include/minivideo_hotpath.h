/*
 * minivideo_hotpath.h -- C-ABI of the MI355X-native H.264 intra reconstruction
 * hot path (libminivideo.so).  Plain pointers and sizes only; no torch / HIP
 * types in any signature (a HIP stream is passed as an opaque void*).
 *
 * What each entry point replaces in the reference (paths relative to
 * minivideo/src/ of misterk72/MiniVideo):
 *
 *   mvhp_parse_annexb / mvhp_index_annexb
 *       demuxer/esparser/esparser.c:40-143      (es_fileParse)
 *       decoder/h264/h264.c:76-188              (NAL loop)
 *       decoder/h264/h264_nalu.c:109-249        (header, emulation prevention)
 *       decoder/h264/h264_parameterset.c:123-397, 812-942 (SPS, PPS)
 *       decoder/h264/h264_slice.c:156-334, 1013-1142      (slice header, MB loop)
 *       decoder/h264/h264_macroblock.c:75-313   (macroblock_layer, minus :278-281)
 *       decoder/h264/h264_cavlc.c:79-346, h264_cabac.c:138-325 (residual blocks)
 *       decoder/h264/h264_intra_prediction.c:196-290, 977-1083 (pred-mode derivation)
 *   mvhp_recon_batch_dev / mvhp_recon_batch_host
 *       decoder/h264/h264_macroblock.c:278-281  (intra_prediction_process call)
 *       decoder/h264/h264_intra_prediction.c:112-2564 (all sample prediction)
 *       decoder/h264/h264_transform.c:121-1610  (dequant, IDCT, DC transforms,
 *                                                picture construction)
 *       export.c:65-188, export_utils.c:117-198 (planar YCbCr gather)
 *       export_utils.c:209-324                  (mb_to_rgb colour conversion)
 *
 * The packed macroblock record below is the build's replacement for the
 * reference's Macroblock_t (decoder/h264/h264_macroblock_struct.h:209-319) as
 * the interface between entropy decoding (host) and reconstruction (GPU).
 */
#ifndef MINIVIDEO_HOTPATH_H
#define MINIVIDEO_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVHP_EXPORT __attribute__((visibility("default")))

/* Return codes follow the reference convention (typedef.h:40-42). */
#define MVHP_SUCCESS      1
#define MVHP_FAILURE      0
#define MVHP_UNSUPPORTED (-1)

/* ---------------------------------------------------------------------------
 * Packed macroblock record: 32-byte header + 384 int16 coefficients = 800 B.
 *
 * Coefficient area (int16, little endian), already inverse-scanned
 * (zig-zag -> raster, h264_transform.c:440-480 is a pure permutation that the
 * host applies while scattering entropy-decoder output):
 *   [  0..255]  luma. mb_kind I4x4 / I16x16: 16 blocks in luma4x4BlkIdx
 *               order (h264_spatial.c:210), each 16 coefficients c[row][col]
 *               row-major.  I16x16: slot 0 of block b holds the *untransformed*
 *               Intra16x16 DC level c1[i][j] with {i,j}=raster position of b
 *               (h264_transform.c:180-186); the kernel runs the 4x4 Hadamard.
 *               mb_kind I8x8: 4 blocks in luma8x8BlkIdx order, each 64
 *               coefficients c[row][col] row-major.
 *   [256..319]  Cb: 4 blocks (chroma4x4BlkIdx order), 16 coefficients each;
 *               slot 0 of block k holds the untransformed chroma DC level k
 *               (h264_transform.c:313-316,354).
 *   [320..383]  Cr, same layout.
 * ------------------------------------------------------------------------- */
#define MVHP_MB_HEADER_BYTES  32
#define MVHP_MB_COEFS         384
#define MVHP_MB_BYTES         800

#define MVHP_KIND_I4x4   0
#define MVHP_KIND_I8x8   1
#define MVHP_KIND_I16x16 2
/* I_PCM (h264_macroblock.c:151-154 returns UNSUPPORTED, ipcm_construction_process h264_intra_prediction.c:2585-2630 is dead
 * code in the reference): only streams opened with MVHP_STREAM_SPEC produce it (outside parity).  The
 * coefficient area then holds the 384 SAMPLES, one byte each, arranged so that every owner of a 32-byte piece of the record
 * finds whole rows in it: for j = 0..7, bytes [64j, 64j+32) = luma rows 2j and 2j+1 (16 samples each), bytes [64j+32, 64j+40) =
 * Cb row j, [64j+40, 64j+48) = Cr row j, [64j+48, 64j+64) = 0; the chroma part of the area (bytes 512..767) is 0.
 * nz_mask = 0, qp_y = 0 (QP'Y of an I_PCM macroblock, 7.4.5). */
#define MVHP_KIND_IPCM   3

typedef struct mvhp_mb_header {
    uint8_t  mb_kind;          /* MVHP_KIND_*  (MbPartPredMode, h264_macroblock_struct.h:33-44) */
    uint8_t  qp_y;             /* QP'Y of this macroblock (h264_macroblock.c:263-269)           */
    uint8_t  cbp;              /* bits 0-3 CodedBlockPatternLuma, bits 4-5 ...Chroma            */
    uint8_t  chroma_pred_mode; /* IntraChromaPredMode 0=DC 1=H 2=V 3=Plane                      */
    uint8_t  i16_pred_mode;    /* Intra16x16PredMode 0=V 1=H 2=DC 3=Plane                       */
    uint8_t  flags;            /* reserved (TransformBypassModeFlag is never set for 8-bit)     */
    uint8_t  unavail;          /* MVHP_UNAVAIL_*: neighbouring macroblocks that exist by geometry but lie in ANOTHER SLICE
                                  (6.4.8: not available); always 0 for the reference's one-slice pictures -- set only by
                                  streams opened with MVHP_STREAM_SPEC, announced by MVHP_PARAM_SLICES   */
    uint8_t  reserved0;
    uint32_t nz_mask;          /* bit b (0-15): luma 4x4 block b (or 8x8 block b>>2) has a
                                  non-zero level; bit 16+k: Cb block k; bit 20+k: Cr block k.
                                  DC levels count for the block whose slot 0 they occupy.
                                  A hint only: the kernel may skip all-zero blocks.         */
    uint8_t  pred_mode[16];    /* final Intra4x4PredMode[16] or Intra8x8PredMode[4]             */
    uint32_t reserved1;
} mvhp_mb_header_t;

#define MVHP_UNAVAIL_A 1u   /* left        (mbAddrA, h264_spatial.c:333-416) */
#define MVHP_UNAVAIL_B 2u   /* above       (mbAddrB) */
#define MVHP_UNAVAIL_C 4u   /* above right (mbAddrC) */
#define MVHP_UNAVAIL_D 8u   /* above left  (mbAddrD) */

/* ---------------------------------------------------------------------------
 * Compact pictures: the transfer format between the host front end and the GPU (PCIe carries it instead of the packed
 * records, of which most int16 slots are zero: ~140 instead of 800 bytes per macroblock on dense content).  One picture =
 *     uint32 mb_off[W*H]          byte offset of each macroblock's compact record, counted from the end of this table
 *     compact records, 4-byte aligned, in macroblock order:
 *         mvhp_mb_header_t        as in the packed record, with reserved1 = number of entries that follow
 *         uint32 entry[n]         one per non-zero level, in the order the entropy decoder delivered them:
 *                                 bits 0-15 = int16 slot of the coefficient area (0..383), bits 16-31 = the level
 *       or, for a macroblock of more than MVHP_COMPACT_MAX_ENTRIES levels, header.flags bit 0 set, reserved1 = 0 and
 *       the 768-byte coefficient area itself.
 * mvhp_expand_compact_dev() turns it into packed records on the device (a memory-bound pass of < 1 KB per macroblock);
 * the packed record stays the input format of the reconstruction kernels.
 * ------------------------------------------------------------------------- */
#define MVHP_COMPACT_MAX_ENTRIES  191
#define MVHP_COMPACT_MB_BYTES_MAX 804    /* 4 (offset) + 32 + 768: a picture needs at most W*H times this ...          */
#define MVHP_COMPACT_SLACK_BYTES  1536   /* ... plus this (entries of one macroblock before it is found to be dense)   */

/* Parameters shared by every picture of one batch (one SPS/PPS pair). */
typedef struct mvhp_stream_params {
    uint32_t width_mbs;                      /* PicWidthInMbs                          */
    uint32_t height_mbs;                     /* PicHeightInMapUnits (frame MBs only)   */
    int32_t  chroma_qp_index_offset;         /* PPS, Cb (h264_transform.c:611-618)     */
    int32_t  second_chroma_qp_index_offset;  /* PPS, Cr                                */
    uint32_t flags;                          /* MVHP_PARAM_*                            */
    /* MVHP_PARAM_SCALING only (ignored otherwise): the weight matrices in force for the batch's pictures, after the
     * fall-back rules of 7.4.2.1.1 / 7.4.2.2 (h264_parameterset.c:723-736, 904-923 parse them; the reference's own LevelScale
     * is only right for flat lists), in RASTER order [i*4+j] / [i*8+j]: LevelScale4x4[plane][m][i][j] =
     * scaling4[plane][i*4+j] * normAdjust4x4(m,i,j) (h264_transform.c:645-741). */
    uint8_t  scaling4[3][16];                /* Intra Y, Cb, Cr                          */
    uint8_t  scaling8[64];                   /* Intra Y 8x8                              */
} mvhp_stream_params_t;

/* flags: the batch may contain Intra8x8 macroblocks (PPS transform_8x8_mode_flag).  A hint for the kernel choice
 * only -- every kernel reconstructs every macroblock kind; callers that build records themselves may leave it 0. */
#define MVHP_PARAM_MAY_HAVE_8X8 1u
/* flags: Intra16x16 luma DC dequantisation by the standard's rule (qP >= 36 takes the left-shift branch, 8.5.10) instead
 * of the reference's `qP > 36` (h264_transform.c:797-808), i.e. without the reference's QP'Y = 36 defect.  Results
 * differ from the reference exactly on Intra16x16 macroblocks at QP'Y = 36.  Set by streams opened with
 * MVHP_STREAM_SPEC (outside the parity contract, opt-in). */
#define MVHP_PARAM_SPEC_LUMA_DC 2u
/* flags (set only for streams opened with MVHP_STREAM_SPEC; outside the parity contract -- the reference
 * decodes none of these correctly):
 * MVHP_PARAM_SLICES   pictures of several slices: records carry mvhp_mb_header_t::unavail.  Reconstructed by the one-picture-
 *                     per-workgroup kernel, where availability is a per-wavefront scalar (the batch kernels keep eight / four
 *                     pictures in lock step at one macroblock position and derive availability from the position alone).
 * MVHP_PARAM_SCALING  scaling4 / scaling8 hold non-flat weight matrices (SPS / PPS scaling lists).  Same kernel. */
#define MVHP_PARAM_SLICES  4u
#define MVHP_PARAM_SCALING 8u

/* Bytes of one reconstructed picture: planar Y | Cb | Cr of the *uncropped*
 * coded size (export.c:80-81), and interleaved RGB8. */
MVHP_EXPORT size_t mvhp_packed_frame_bytes(const mvhp_stream_params_t *p);
MVHP_EXPORT size_t mvhp_yuv_frame_bytes(const mvhp_stream_params_t *p);
MVHP_EXPORT size_t mvhp_rgb_frame_bytes(const mvhp_stream_params_t *p);

/* ---------------------------------------------------------------------------
 * Host front end (no GPU involved): Annex-B bytes -> packed pictures.
 * ------------------------------------------------------------------------- */
typedef struct mvhp_stream mvhp_stream_t;   /* parsed elementary stream */

/* Index + parse parameter sets of an Annex-B buffer held in memory.
 * The buffer must outlive the handle. */
MVHP_EXPORT int  mvhp_stream_open(const uint8_t *data, size_t size, mvhp_stream_t **out);
/* The same with flags.  0 = reference parity (everything above).  MVHP_STREAM_SPEC (opt-in; also
 * chosen by minivideo_decode when the environment has MINIVIDEO_SPEC=1): index the stream the way the standard
 * defines it instead of the way esparser.c:40-143 does -- three-byte start codes (Annex B), slice / SPS / PPS NAL units
 * of any nal_ref_idc, no 32-byte blind tail -- and reconstruct Intra16x16 at QP'Y = 36 by the standard's rule
 * (MVHP_PARAM_SPEC_LUMA_DC); pictures of several slices (MVHP_PARAM_SLICES), SPS / PPS scaling lists (MVHP_PARAM_SCALING) and
 * I_PCM macroblocks (MVHP_KIND_IPCM) are decoded by the standard's rules. */
#define MVHP_STREAM_SPEC 1u
MVHP_EXPORT int  mvhp_stream_open_ex(const uint8_t *data, size_t size, uint32_t flags, mvhp_stream_t **out);
/* Same for an ISO-BMFF (MP4/MOV) buffer: avcC parameter sets + the IDR NAL units of the sync samples of the first
 * video track (replaces demuxer/mp4/mp4.c:2587 mp4_fileParse for the thumbnail path). */
MVHP_EXPORT int  mvhp_stream_open_mp4(const uint8_t *data, size_t size, mvhp_stream_t **out);
MVHP_EXPORT void mvhp_stream_close(mvhp_stream_t *s);
MVHP_EXPORT int  mvhp_stream_idr_count(const mvhp_stream_t *s);
/* Parameters in force for IDR picture `idr` (valid after mvhp_stream_open). */
MVHP_EXPORT int  mvhp_stream_params(const mvhp_stream_t *s, int idr, mvhp_stream_params_t *out);
/* Entropy-decode IDR picture `idr` into `packed` (mvhp_packed_frame_bytes()).
 * Thread-safe for distinct `idr` on the same handle. */
MVHP_EXPORT int  mvhp_stream_decode_packed(const mvhp_stream_t *s, int idr, void *packed, size_t packed_bytes);
/* The same into the compact transfer format; `cap` >= W*H * MVHP_COMPACT_MB_BYTES_MAX + MVHP_COMPACT_SLACK_BYTES,
 * *used = bytes written. */
MVHP_EXPORT int  mvhp_stream_decode_compact(const mvhp_stream_t *s, int idr, void *buf, size_t cap, size_t *used);
MVHP_EXPORT const char *mvhp_stream_last_error(void);

/* ---------------------------------------------------------------------------
 * GPU reconstruction.
 * ------------------------------------------------------------------------- */
typedef struct mvhp_ctx mvhp_ctx_t;

MVHP_EXPORT int  mvhp_device_count(void);
MVHP_EXPORT int  mvhp_create(int device, mvhp_ctx_t **out);
MVHP_EXPORT void mvhp_destroy(mvhp_ctx_t *ctx);
MVHP_EXPORT const char *mvhp_last_error(void);

/* Reconstruct n_frames pictures whose packed records are resident in device
 * memory.  d_yuv receives n_frames * mvhp_yuv_frame_bytes(); d_rgb (may be
 * NULL) receives n_frames * mvhp_rgb_frame_bytes().  `stream` is a hipStream_t
 * (NULL = the context's own stream).  Asynchronous with respect to the host. */
MVHP_EXPORT int  mvhp_recon_batch_dev(mvhp_ctx_t *ctx, const mvhp_stream_params_t *p,
                                      const void *d_packed, int n_frames,
                                      uint8_t *d_yuv, uint8_t *d_rgb, void *stream);

/* n_pictures compact pictures (see "Compact pictures" above), `stride` bytes apart in device memory, -> packed records
 * (n_pictures * mvhp_packed_frame_bytes()) in device memory.  Asynchronous on `stream` (NULL = the context's own). */
MVHP_EXPORT int  mvhp_expand_compact_dev(mvhp_ctx_t *ctx, const mvhp_stream_params_t *p, const void *d_compact,
                                         size_t stride, int n_pictures, void *d_packed, void *stream);

/* Same, but only the stages selected by `stages` (bit 0: reconstruction kernel,
 * bit 1: colour kernel) -- lets a caller bracket each kernel with its own events. */
#define MVHP_STAGE_RECON 1
#define MVHP_STAGE_COLOR 2
MVHP_EXPORT int  mvhp_recon_stages_dev(mvhp_ctx_t *ctx, const mvhp_stream_params_t *p,
                                       const void *d_packed, int n_frames,
                                       uint8_t *d_yuv, uint8_t *d_rgb, void *stream, int stages);

/* Page-locked host memory for the host-buffer entry points (H2D / D2H at full PCIe rate). */
MVHP_EXPORT void *mvhp_host_alloc(size_t bytes);
MVHP_EXPORT void  mvhp_host_free(void *p);

/* Wait for `stream` (NULL = context stream) and report any error the kernels
 * flagged since the last check. */
MVHP_EXPORT int  mvhp_sync_check(mvhp_ctx_t *ctx, void *stream);

/* Host-buffer convenience: H2D, reconstruct, D2H, synchronise. */
MVHP_EXPORT int  mvhp_recon_batch_host(mvhp_ctx_t *ctx, const mvhp_stream_params_t *p,
                                       const void *h_packed, int n_frames,
                                       uint8_t *h_yuv, uint8_t *h_rgb);

/* Test hook: the next launch of a wide kernel form hands out its work units `delta` off (see hotpath_abi.hip).  Never in products. */
MVHP_EXPORT int  mvhp_debug_skew_next_ticket_base(mvhp_ctx_t *ctx, int delta);

/* What the last reconstruction launch of this context used (speed-only choices of the launcher):
 * *layout = MVHP_LAYOUT_ROWS/QUAD/OCT, *waves = wavefronts per workgroup.  Either pointer may be NULL. */
MVHP_EXPORT int  mvhp_last_launch_info(const mvhp_ctx_t *ctx, int *layout, int *waves);

/* 1 (default): the reconstruction kernel converts to RGB in its epilogue when d_rgb is given;
 * 0: a separate colour kernel reads the planes back.  Speed only, never results. */
MVHP_EXPORT int  mvhp_set_fused_color(mvhp_ctx_t *ctx, int on);

/* Tuning knob (speed only, never results): waves per picture workgroup
 * (4, 6, 8, 12 or 16; a layout that is not built for the value takes the next smaller one);
 * 0 = choose from batch size. */
MVHP_EXPORT int  mvhp_set_waves_per_picture(mvhp_ctx_t *ctx, int waves);

/* Tuning knob (speed only, never results): how pictures map onto workgroups.
 * MVHP_LAYOUT_ROWS: one picture per workgroup, one wavefront per macroblock row (fills the chip from
 * ~256 pictures); MVHP_LAYOUT_QUAD: four pictures per workgroup, 16 lanes per picture (fewer
 * instructions per macroblock, wants >= ~768 pictures); MVHP_LAYOUT_OCT: eight pictures per workgroup, 8 lanes per
 * picture (fewest instructions, wants >= ~2048 pictures); MVHP_LAYOUT_AUTO chooses from the batch size. */
#define MVHP_LAYOUT_AUTO 0
#define MVHP_LAYOUT_ROWS 1
#define MVHP_LAYOUT_QUAD 2
#define MVHP_LAYOUT_OCT  3
/* The "wide" forms spread ONE picture (one group of four) over several workgroups -- bands of macroblock rows on different
 * CUs, the rows between two bands handed over through global memory -- so that a handful of pictures fills the chip:
 * MVHP_LAYOUT_WIDE = the one-picture kernel in bands (1 .. ~64 pictures; also every small batch with slices / scaling
 * matrices), MVHP_LAYOUT_QUAD_WIDE = the four-picture kernel in bands (up to ~1024 pictures). */
#define MVHP_LAYOUT_WIDE      4
#define MVHP_LAYOUT_QUAD_WIDE 5
/* MVHP_LAYOUT_PIPE: four pictures over several workgroups as QUAD_WIDE, and every macroblock row worked on by THREE wavefronts
 * in a pipeline (residuals / prediction / write-out): the shortest macroblock step, i.e. the lowest latency of a small batch. */
#define MVHP_LAYOUT_PIPE      6
/* MVHP_LAYOUT_PIPE1: the same pipeline with ONE picture per wavefront (nothing runs in lock step with another picture): the
 * lowest latency of a handful of pictures on any profile; also reconstructs slices / scaling matrices. */
#define MVHP_LAYOUT_PIPE1     7
#define MVHP_LAYOUT_COUNT     8
MVHP_EXPORT int  mvhp_set_layout(mvhp_ctx_t *ctx, int layout);

/* ---------------------------------------------------------------------------
 * Decode engine: the whole split path as one pipelined call --
 *   host threads entropy-decode pictures (h264.c:76-188 NAL loop, h264_slice.c:1046-1139 macroblock loop)
 *   into page-locked chunks -> H2D -> batched reconstruction kernel -> D2H into page-locked chunks ->
 *   `sink` called once per picture, in the order of `order` (export.c:618-767 is what minivideo_decode's sink does).
 * Pictures are independent, so contexts (one per HIP device; several per device when `contexts` exceeds the
 * device count, e.g. to exercise the multi-device path on one GPU) pull whole batches from one queue; no collective.
 * A batch that fails on one context is entropy-decoded again and re-queued once to another context.
 * ------------------------------------------------------------------------- */
typedef struct mvhp_engine mvhp_engine_t;

typedef struct mvhp_engine_opts {
    int32_t contexts;        /* 0 = one per visible HIP device (env MINIVIDEO_GPUS caps it, MINIVIDEO_FAKE_GPUS sets it) */
    int32_t host_threads;    /* entropy threads; 0 = hardware concurrency (env MINIVIDEO_HOST_THREADS)                */
    int32_t batch_pictures;  /* pictures per kernel launch; 0 = auto (device fill, memory budget)  (MINIVIDEO_BATCH)  */
    int32_t chunk_pictures;  /* pictures per H2D / D2H transfer; 0 = auto (~64 MiB of records)                       */
    int32_t fail_context;    /* test hook: the first batch launched on this context reports a failure; -1 = off       */
    int32_t first_device;    /* context k runs on HIP device (first_device + k) % device count (one process per GPU:  */
                             /* contexts = 1, first_device = LOCAL_RANK)                                              */
    int32_t reserved[2];     /* [0] bit 0: the contexts' batch buffers come from one placed arena each (mvhp_placed_alloc_sets;
                                also env MINIVIDEO_PLACED=1) -- for engines that live long: the arena takes seconds to get */
} mvhp_engine_opts_t;

typedef struct mvhp_decode_stats {
    uint32_t pictures_issued;      /* pictures handed to the entropy stage (a re-queued picture counts twice)       */
    uint32_t pictures_ok;          /* pictures the sink accepted                                                     */
    uint32_t pictures_failed;      /* parse / device / sink failures delivered to the sink                           */
    uint32_t batches;              /* kernel launches                                                                */
    uint32_t batches_requeued;     /* batches that failed on one context and were re-queued to another               */
    uint32_t contexts;
    uint32_t host_threads;
    uint32_t launches_by_layout[4];/* indexed by MVHP_LAYOUT_AUTO .. MVHP_LAYOUT_OCT; the wide forms: launches_wide below   */
    uint32_t max_batch_pictures;
    double   wall_s;               /* whole call                                                                     */
    double   entropy_busy_s;       /* summed over host threads                                                       */
    double   h2d_s, kernel_s, d2h_s; /* device-side durations (HIP events), summed over contexts                     */
    double   sink_s;               /* time inside the sink callback                                                  */
    uint64_t stream_bytes;         /* NAL bytes entropy-decoded                                                      */
    uint64_t h2d_bytes, d2h_bytes;
    /* where a COLD call's time goes (an engine keeps its pools: the second call of the same shape allocates nothing)  */
    double   host_alloc_s;         /* page-locking host memory (summed over the threads that did it)                 */
    double   dev_alloc_s;          /* device allocations                                                             */
    double   first_launch_s;       /* the first reconstruction call of each context: code-object load + first launch */
    double   first_picture_s;      /* from the call to the first picture at the sink                                 */
    uint64_t host_alloc_bytes, dev_alloc_bytes;
    uint32_t placed_buffers;       /* 1: the device batch buffers come from mvhp_placed_alloc (MINIVIDEO_PLACED=1)   */
    uint32_t reserved;
    uint32_t launches_wide[4];     /* launches on MVHP_LAYOUT_WIDE, _QUAD_WIDE, _PIPE, _PIPE1 (launches_by_layout: 0..3)   */
} mvhp_decode_stats_t;

/* Called on the calling thread, once per picture, in the order of `order`.  rc = MVHP_SUCCESS: yuv (and rgb when
 * asked for) point into page-locked memory valid during the call.  Otherwise yuv = rgb = NULL and err says why.
 * Return 1: picture accepted (counts towards `wanted`); 0: not accepted (counts as a failure); -1: stop decoding;
 * 2: accepted AND kept -- yuv / rgb stay valid after the call returns, until mvhp_engine_release_picture(e, seq), which any
 * thread may call (a pool of file writers: minivideo_decode).  Kept pictures occupy the engine's output chunks, so keep few
 * (the pipeline waits for a free chunk); mvhp_engine_decode does not return before the last kept picture has been released. */
typedef int (*mvhp_picture_sink_t)(void *user, int seq, int idr, int rc, const char *err,
                                   const mvhp_stream_params_t *p, const uint8_t *yuv, const uint8_t *rgb);

MVHP_EXPORT int  mvhp_engine_create(const mvhp_engine_opts_t *opts /* may be NULL */, mvhp_engine_t **out);
MVHP_EXPORT void mvhp_engine_destroy(mvhp_engine_t *e);
/* `want_rgb` of mvhp_engine_decode: 0 = planes only; MVHP_OUT_RGB = planes and RGB; MVHP_OUT_RGB_ONLY = RGB only (the planes
 * are still reconstructed on the device -- RGB is made from them -- but not downloaded: the sink gets yuv = NULL). */
#define MVHP_OUT_RGB      1
#define MVHP_OUT_RGB_ONLY 3
/* Decode the pictures order[0..n_order) of `s` (IDR indices) until `wanted` of them have been accepted by the sink
 * (the reference stops after picture_number IDRs, h264.c:173-179: no more pictures than needed are entropy-decoded).
 * sink may be NULL (every reconstructed picture counts as accepted).  Returns MVHP_SUCCESS when `wanted` pictures
 * were accepted, or when the list ended after at least one. */
MVHP_EXPORT int  mvhp_engine_decode(mvhp_engine_t *e, const mvhp_stream_t *s, const int *order, int n_order, int wanted,
                                    int want_rgb, mvhp_picture_sink_t sink, void *user, mvhp_decode_stats_t *stats);
/* Gives back a picture the sink kept (verdict 2) during the running mvhp_engine_decode call; anything else is ignored. */
MVHP_EXPORT void mvhp_engine_release_picture(mvhp_engine_t *e, int seq);

/* ---- memory placement (MI355X: device memory alternates, in regions of tens of GB, between two halves of the memory
 * system; DESIGN.md 3 "Placement") ---- */
/* Time concurrent streaming writes over two device windows of `bytes` each (both are overwritten): two windows in the
 * same half take about twice as long per pass as two windows in different halves. */
MVHP_EXPORT int  mvhp_probe_pair(int device, void *d_a, void *d_b, size_t bytes, int reps, float *ms_per_pass);
/* `count` (<= 8) device buffers of at least bytes[i] inside ONE allocation (arena_bytes = 0: what is free less 24 GB, at most
 * 200 GB or MVHP_PLACED_ARENA_GB from the environment), placed -- as far as the arena shows several groups -- so that every buffer lies in a group of its own, the largest
 * choosing first: records, planes and RGB of a batch in three different groups is the fastest placement there is
 * (tools/placement_predict.py).  For long-lived batch buffers: the large allocation takes seconds (the driver clears it).
 * groups_of[i] (may be NULL): group of buffer i, -1 = straddles; *groups_found (may be NULL): groups seen in the arena.
 * MVHP_FAILURE: not enough memory -- use ordinary allocations. */
MVHP_EXPORT int  mvhp_placed_alloc(int device, int count, const size_t *bytes, size_t arena_bytes, void **d_ptrs, void **arena,
                                   int *groups_of, int *groups_found);
/* The same for a pipeline's batch buffers: `sets` copies of `count` (<= 8) buffers, d_ptrs[s * count + i] = buffer i of set s;
 * buffer i of every set lies in the group chosen for i (a launch reads / writes the buffers of ONE set: its records, planes and
 * RGB are in three different groups); any_group[i] != 0 (may be NULL = none): buffer i goes wherever room is left.  At most
 * four buffers may ask for a group of their own.  The decode engine uses it when MINIVIDEO_PLACED=1. */
MVHP_EXPORT int  mvhp_placed_alloc_sets(int device, int sets, int count, const size_t *bytes, const uint8_t *any_group,
                                        size_t arena_bytes, void **d_ptrs, void **arena, int *groups_of, int *groups_found);
MVHP_EXPORT void mvhp_placed_free(void *arena);

#ifdef __cplusplus
}
#endif
#endif /* MINIVIDEO_HOTPATH_H */
