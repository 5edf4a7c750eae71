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

typedef struct mvhp_mb_header {
    uint8_t  mb_kind;          /* MVHP_KIND_*  (MbPartPredMode, h264_macroblock_struct.h:33-44) */
    uint8_t  qp_y;             /* QP'Y of this macroblock (h264_macroblock.c:263-269)           */
    uint8_t  cbp;              /* bits 0-3 CodedBlockPatternLuma, bits 4-5 ...Chroma            */
    uint8_t  chroma_pred_mode; /* IntraChromaPredMode 0=DC 1=H 2=V 3=Plane                      */
    uint8_t  i16_pred_mode;    /* Intra16x16PredMode 0=V 1=H 2=DC 3=Plane                       */
    uint8_t  flags;            /* reserved (TransformBypassModeFlag is never set for 8-bit)     */
    uint16_t reserved0;
    uint32_t nz_mask;          /* bit b (0-15): luma 4x4 block b (or 8x8 block b>>2) has a
                                  non-zero level; bit 16+k: Cb block k; bit 20+k: Cr block k.
                                  DC levels count for the block whose slot 0 they occupy.
                                  A hint only: the kernel may skip all-zero blocks.         */
    uint8_t  pred_mode[16];    /* final Intra4x4PredMode[16] or Intra8x8PredMode[4]             */
    uint32_t reserved1;
} mvhp_mb_header_t;

/* Parameters shared by every picture of one batch (one SPS/PPS pair). */
typedef struct mvhp_stream_params {
    uint32_t width_mbs;                      /* PicWidthInMbs                          */
    uint32_t height_mbs;                     /* PicHeightInMapUnits (frame MBs only)   */
    int32_t  chroma_qp_index_offset;         /* PPS, Cb (h264_transform.c:611-618)     */
    int32_t  second_chroma_qp_index_offset;  /* PPS, Cr                                */
    uint32_t flags;                          /* MVHP_PARAM_* hints (speed only)        */
} mvhp_stream_params_t;

/* flags: the batch may contain Intra8x8 macroblocks (PPS transform_8x8_mode_flag).  A hint for the kernel choice
 * only -- every kernel reconstructs every macroblock kind; callers that build records themselves may leave it 0. */
#define MVHP_PARAM_MAY_HAVE_8X8 1u

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

/* Measurement helper for bench.py: launches the same work `iters` times on
 * `stream`, bracketed by HIP events recorded on that stream; returns the mean
 * milliseconds per launch of the reconstruction kernel and (when d_rgb is not
 * NULL) of the colour kernel. */
MVHP_EXPORT int  mvhp_time_recon(mvhp_ctx_t *ctx, const mvhp_stream_params_t *p,
                                 const void *d_packed, int n_frames,
                                 uint8_t *d_yuv, uint8_t *d_rgb, void *stream,
                                 int iters, float *ms_recon, float *ms_color);

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
MVHP_EXPORT int  mvhp_set_layout(mvhp_ctx_t *ctx, int layout);

#ifdef __cplusplus
}
#endif
#endif /* MINIVIDEO_HOTPATH_H */
