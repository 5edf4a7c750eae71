// recon_kernels.h -- launch interface between the C-ABI layer and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mvhp {

struct ReconArgs {
    const uint8_t *packed;   // n_frames * W*H * 800 B packed macroblock records
    uint8_t       *yuv;      // n_frames * W*H * 384 B planar Y|Cb|Cr
    uint8_t       *rgb;      // n_frames * W*H * 768 B RGB8 written by the fused colour epilogue, or NULL
    uint32_t      *err;      // device word: bit0 = dependency wait timed out
    int            width_mbs, height_mbs;
    int            cqp_off_cb, cqp_off_cr;
    int            n_frames;
    int            dc_shift_from;   // Intra16x16 luma DC takes the left-shift branch from this qP on: 37 = the reference's
                                    // `qP > 36` (h264_transform.c:797), 36 = the standard (MVHP_PARAM_SPEC_LUMA_DC)
    // SURVEY 8f row f4 (MVHP_STREAM_SPEC streams; the one-picture-per-workgroup kernel only -- the launcher routes there):
    int            slices;          // MVHP_PARAM_SLICES: mvhp_mb_header_t::unavail is honoured
    int            scaling;         // MVHP_PARAM_SCALING: weights[] below instead of Flat_4x4_16 / Flat_8x8_16
    uint8_t        weights[112];    // scaling4[3][16] | scaling8[64], raster order
    // Wide launches (launch_recon_wide / launch_recon_quad_wide): a picture's macroblock rows are spread over several
    // workgroups ("bands" of NW consecutive rows), so that a handful of pictures fills the chip.
    uint32_t      *wide_ticket;     // device counter; a workgroup's unit (picture, band) = atomicAdd(ticket, 1) - wide_base:
                                    // units are taken in the order workgroups START, so the band a workgroup waits for is
                                    // always held by a workgroup that is already running (no assumption on dispatch order)
    uint32_t       wide_base;       // value of *wide_ticket when this launch starts (the counter is never reset)
    uint32_t       wide_epoch;      // tag of this launch's seam granules (never 0, differs from every tag left in `seam`)
    unsigned long long *seam;       // [picture][seam = band boundary][macroblock column][8] granules {bottom-row dword, tag}:
                                    // the last row of a band hands its bottom samples (16 luma + 8 Cb + 8 Cr per column) to the
                                    // first row of the next band, which runs on another CU
};

// granules per macroblock column of a seam, and their size
constexpr int SEAM_GRANULES = 8;
inline size_t recon_wide_seam_bytes(int width_mbs, int height_mbs, int n_frames, int nw)
{
    const int bands = (height_mbs + nw - 1) / nw;
    return (size_t)n_frames * (size_t)(bands > 1 ? bands - 1 : 0) * width_mbs * SEAM_GRANULES * sizeof(unsigned long long);
}

struct ExpandArgs {
    const uint8_t *compact;  // n_pictures compact pictures, `stride` bytes apart
    size_t         stride;
    uint8_t       *packed;   // n_pictures * mbs * 800 B packed records (output)
    int            mbs, n_pictures;
};

struct ColorArgs {
    const uint8_t *yuv;
    uint8_t       *rgb;
    int            width_mbs, height_mbs, n_frames;
};

size_t     recon_lds_bytes(int width_mbs, int nw);
hipError_t launch_recon(const ReconArgs &a, int n_frames, int nw, hipStream_t stream);
// the same kernel with a picture's rows in bands of `nw` (4) over several workgroups (wide_* and seam of ReconArgs set)
hipError_t launch_recon_wide(const ReconArgs &a, int n_frames, int nw, hipStream_t stream);
// four pictures per workgroup, 16 lanes per picture (recon_quad.hip)
size_t     recon_quad_lds_bytes(int width_mbs, int nw);
hipError_t launch_recon_quad(const ReconArgs &a, int nw, hipStream_t stream);
// ... in bands of `nw` (4 or 8) rows over several workgroups (seams sized for 4 * ceil(n_frames / 4) pictures)
hipError_t launch_recon_quad_wide(const ReconArgs &a, int nw, hipStream_t stream);
// four pictures per wavefront in bands of four rows, three wavefronts per row (recon_pipe.hip); seams as the wide form with nw = 4
size_t     recon_pipe_lds_bytes(int width_mbs, int rows);
hipError_t launch_recon_pipe(const ReconArgs &a, int rows, hipStream_t stream);   // rows per band: 1, 2 or 4
// the same pipeline with one picture per wavefront (recon_pipe1.hip); handles slices / scaling matrices
size_t     recon_pipe1_lds_bytes(int width_mbs, int rows);
hipError_t launch_recon_pipe1(const ReconArgs &a, int rows, hipStream_t stream);
// eight pictures per workgroup, 8 lanes per picture (recon_oct.hip)
size_t     recon_oct_lds_bytes(int width_mbs, int nw);
hipError_t launch_recon_oct(const ReconArgs &a, int nw, hipStream_t stream);
hipError_t launch_color(const ColorArgs &a, hipStream_t stream);
hipError_t launch_expand(const ExpandArgs &a, hipStream_t stream);

} // namespace mvhp
