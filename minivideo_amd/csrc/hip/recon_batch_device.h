// recon_batch_device.h -- device pieces shared by the batch kernels (recon_quad.hip: four pictures per wavefront,
// recon_oct.hip: eight): the per-picture LDS block, the per-workgroup tables, packed add/clip, the packed-16-bit
// RGB conversion, plane-prediction helpers, cross-lane helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "recon_device.h"

namespace mvhp {

// per-quarter (= per-picture, per-wave) LDS state
struct __attribute__((aligned(16))) QLds {
    union {
        int32_t scr[128];    // Intra8x8: two 8x8 blocks of row-transformed coefficients (transpose scratch)
        int16_t res[256];    // Intra4x4: [blk][sample] ; Intra8x8: [blk8][column][row]
    };
    uint8_t T[17 * 32 + 16]; // luma tile: row 0 = top neighbours; byte 15 = left/corner, 16..31 samples;
                             // row 0 bytes 32..39 = up-right neighbours
    uint8_t TC[2][9 * 16];   // chroma tiles: row 0 = top; byte 7 = left/corner, 8..15 samples
    uint8_t Lcol[16];        // compact left neighbour column (luma)
    uint8_t LcolC[2][8];     // compact left neighbour columns (Cb, Cr)
    uint8_t E8[32];          // filtered Intra8x8 edge, see recon_device.h mode_entry()
    uint8_t SC[2][8 * 24];   // output strip, chroma: rows of the three parked macroblocks (the fourth flushes from registers)
#ifdef MVHP_QLDS_PAD
    uint8_t pad[MVHP_QLDS_PAD];  // measurement builds: bank offset between the pictures of a wavefront
#endif
};                           // 1808 B: pictures land 452 dwords apart (different banks)
#ifndef MVHP_QLDS_PAD
static_assert(sizeof(QLds) == 1808, "QLds layout");
#endif

struct __attribute__((aligned(16))) QTables {
    int      progress[16];   // macroblocks completed by wave w (monotonic over its rows)
    int      abort_flag;
    int      pad[3];
    int4     q4[52];         // per qP: LevelScale4x4 classes (0,0) (1,1) (other), pre-shifted left by max(qP/6-4,0);
                             // w = shr | rnd << 8 | (qP/6) << 16 | (qP%6) << 24, shr = max(4-qP/6,0), rnd = (1<<shr)>>1
    int      ls0[52];        // LevelScale4x4(qP%6,0,0), unshifted (DC transforms)
    int      ls8[36];        // LevelScale8x8 classes (h264.c:438-446)
    uint8_t  qpc[64];        // Table 8-15 (h264_transform.c:71): qPI -> QPc
    uint32_t tap4[2 * 9 * 16];
    uint32_t tap8[9 * 64];
};

#ifndef MVHP_RGB_HINT
#define MVHP_RGB_HINT ""   // cache-policy suffix of the RGB stores (measurement builds try " nt" / " sc1")
#endif
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));   // native vectors: usable as inline-asm register operands
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int4 as_int4(v4i v) { return make_int4(v.x, v.y, v.z, v.w); }

__device__ __forceinline__ int pk_add_sat(int a, int b)
{
    return __builtin_bit_cast(int, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
// two int16 -> two uint8 with unsigned saturation, in the low 16 bits
__device__ __forceinline__ uint32_t sat_pk_u8(int v)
{
    uint32_t o;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(o) : "v"(v));
    return o;
}

// level * LevelScale + addend with the 16-bit level taken straight out of its packed word (v_mad_i32_i16: signed 16-bit
// operands selected by op_sel, 32-bit addend): one instruction where extracting the half and a 24-bit multiply were two.
// `ls` must fit int16 (LevelScale << shift is at most 4096 for 4x4 and 2304 for 8x8).
template <int HI>
__device__ __forceinline__ int mad_level(int pk, int ls, int addend)
{
    int d;
    if (HI) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(pk), "v"(ls), "v"(addend));
    else asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(pk), "v"(ls), "v"(addend));
    return d;
}

// Prediction (four row words of four samples) + residual (eight packed int16 pairs) -> tile.
__device__ __forceinline__ void emit_block(uint8_t *dst, int pitch, const uint32_t pw[4], const int r2[8])
{
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const int lo = (int)__builtin_amdgcn_perm(0u, pw[y], 0x0c010c00u);
        const int hi = (int)__builtin_amdgcn_perm(0u, pw[y], 0x0c030c02u);
        const uint32_t a = sat_pk_u8(pk_add_sat(lo, r2[2 * y]));
        const uint32_t b = sat_pk_u8(pk_add_sat(hi, r2[2 * y + 1]));
        *reinterpret_cast<uint32_t *>(dst + y * pitch) = a | (b << 16);
    }
}

// The same with the residual packed by ROW pairs: r2[i], i = 4*y + x (y = 0, 1), holds the residuals of samples
// (x, y) and (x, y + 2) -- the pairing of the eight-pictures kernel, where one lane predicts those two samples of
// every Intra4x4 block.
__device__ __forceinline__ void emit_block_ypairs(uint8_t *dst, int pitch, const uint32_t pw[4], const int r2[8])
{
    uint32_t o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int x = i & 3, y = i >> 2;
        const uint32_t sel = (uint32_t)x | (0x0cu << 8) | ((uint32_t)(4 + x) << 16) | (0x0cu << 24);
        const int pp = (int)__builtin_amdgcn_perm(pw[y + 2], pw[y], sel);          // pred(x,y) | pred(x,y+2) << 16
        o[i] = sat_pk_u8(pk_add_sat(pp, r2[i]));                                   // byte 0: (x,y), byte 1: (x,y+2)
    }
#pragma unroll
    for (int y = 0; y < 2; y++) {
        const uint32_t t01 = __builtin_amdgcn_perm(o[4 * y + 1], o[4 * y + 0], 0x05010400u);
        const uint32_t t23 = __builtin_amdgcn_perm(o[4 * y + 3], o[4 * y + 2], 0x05010400u);
        *reinterpret_cast<uint32_t *>(dst + y * pitch) = __builtin_amdgcn_perm(t23, t01, 0x05040100u);
        *reinterpret_cast<uint32_t *>(dst + (y + 2) * pitch) = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
    }
}

// Four plane-prediction samples clip255((v + k*b) >> 5), k = 0..3, as one row word (v_ashr_pk_u8_i32 shifts,
// saturates to 0..255 and packs two samples).
__device__ __forceinline__ uint32_t plane_row(int v, int b)
{
    const uint32_t lo = (uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(v, v + b, 5);
    const uint32_t hi = (uint16_t)__builtin_amdgcn_ashr_pk_u8_i32(v + 2 * b, v + 3 * b, 5);
    return lo | (hi << 16);
}

// 16 samples of one row -> 48 bytes of RGB: 2x1 nearest chroma and the integer formula of export_utils.c:300-302,
// in packed 16-bit arithmetic (two samples per instruction).  The reference's products are rewritten so that they
// fit 16 bits -- (298 l) >> 8 == (149 l) >> 7, (408 c) >> 8 == (204 c) >> 7, (516 c) >> 8 == (129 c) >> 6 for every
// byte l, c; 100 c and 208 c fit as they are -- and every sum stays inside int16, so v_sat_pk_u8_i16 is the clip.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 bytes01(uint32_t w) { return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, w, 0x0c010c00u)); }
__device__ __forceinline__ u16x2 bytes23(uint32_t w) { return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(0u, w, 0x0c030c02u)); }
__device__ __forceinline__ uint32_t sat_pk_u8v(s16x2 v) { return sat_pk_u8(__builtin_bit_cast(int, v)); }
// four samples (one luma word, their two chroma samples as a 16-bit pair per plane) -> three dwords of RGB
__device__ __forceinline__ void rgb4(uint32_t yw, u16x2 cb, u16x2 cr, int &d0, int &d1, int &d2)
{
    const s16x2 rt = __builtin_bit_cast(s16x2, (u16x2)((cr * (unsigned short)204) >> 7)) - (short)222;
    const s16x2 gt = (short)135 - __builtin_bit_cast(s16x2, (u16x2)((cb * (unsigned short)100) >> 8)) -
                     __builtin_bit_cast(s16x2, (u16x2)((cr * (unsigned short)208) >> 8));
    const s16x2 bt = __builtin_bit_cast(s16x2, (u16x2)((cb * (unsigned short)129) >> 6)) - (short)276;
    const s16x2 ly01 = __builtin_bit_cast(s16x2, (u16x2)((bytes01(yw) * (unsigned short)149) >> 7));
    const s16x2 ly23 = __builtin_bit_cast(s16x2, (u16x2)((bytes23(yw) * (unsigned short)149) >> 7));
    const s16x2 rtl = __builtin_shufflevector(rt, rt, 0, 0), rth = __builtin_shufflevector(rt, rt, 1, 1);
    const s16x2 gtl = __builtin_shufflevector(gt, gt, 0, 0), gth = __builtin_shufflevector(gt, gt, 1, 1);
    const s16x2 btl = __builtin_shufflevector(bt, bt, 0, 0), bth = __builtin_shufflevector(bt, bt, 1, 1);
    const uint32_t RA = sat_pk_u8v(ly01 + rtl), GA = sat_pk_u8v(ly01 + gtl), BA = sat_pk_u8v(ly01 + btl);
    const uint32_t RB = sat_pk_u8v(ly23 + rth), GB = sat_pk_u8v(ly23 + gth), BB = sat_pk_u8v(ly23 + bth);
    const uint32_t W1 = GA | (BA << 16), W2 = RB | (GB << 16);
    d0 = (int)__builtin_amdgcn_perm(W1, RA, 0x01060400u);   // R0 G0 B0 R1
    d1 = (int)__builtin_amdgcn_perm(W2, W1, 0x06040301u);   // G1 B1 R2 G2
    d2 = (int)__builtin_amdgcn_perm(BB, W2, 0x05030104u);   // B2 R3 G3 B3
}
__device__ __forceinline__ void rgb16(const uint4 yv, const uint2 cbv, const uint2 crv, v4i &o0, v4i &o1, v4i &o2)
{
    int d[12];
    rgb4(yv.x, bytes01(cbv.x), bytes01(crv.x), d[0], d[1], d[2]);
    rgb4(yv.y, bytes23(cbv.x), bytes23(crv.x), d[3], d[4], d[5]);
    rgb4(yv.z, bytes01(cbv.y), bytes01(crv.y), d[6], d[7], d[8]);
    rgb4(yv.w, bytes23(cbv.y), bytes23(crv.y), d[9], d[10], d[11]);
    o0 = v4i{d[0], d[1], d[2], d[3]};
    o1 = v4i{d[4], d[5], d[6], d[7]};
    o2 = v4i{d[8], d[9], d[10], d[11]};
}

// The same split in two: the chroma terms of a pair of chroma samples (four luma samples wide), and their use on one
// luma word.  Two luma rows share a chroma row (export_utils.c:278-279), so a lane that owns both rows of a pair computes
// the terms once.
struct RgbTerms {
    s16x2 rtl, rth, gtl, gth, btl, bth;
};
__device__ __forceinline__ RgbTerms rgb_terms(u16x2 cb, u16x2 cr)
{
    const s16x2 rt = __builtin_bit_cast(s16x2, (u16x2)((cr * (unsigned short)204) >> 7)) - (short)222;
    const s16x2 gt = (short)135 - __builtin_bit_cast(s16x2, (u16x2)((cb * (unsigned short)100) >> 8)) -
                     __builtin_bit_cast(s16x2, (u16x2)((cr * (unsigned short)208) >> 8));
    const s16x2 bt = __builtin_bit_cast(s16x2, (u16x2)((cb * (unsigned short)129) >> 6)) - (short)276;
    RgbTerms t;
    t.rtl = __builtin_shufflevector(rt, rt, 0, 0); t.rth = __builtin_shufflevector(rt, rt, 1, 1);
    t.gtl = __builtin_shufflevector(gt, gt, 0, 0); t.gth = __builtin_shufflevector(gt, gt, 1, 1);
    t.btl = __builtin_shufflevector(bt, bt, 0, 0); t.bth = __builtin_shufflevector(bt, bt, 1, 1);
    return t;
}
__device__ __forceinline__ void rgb4_apply(uint32_t yw, const RgbTerms &t, int &d0, int &d1, int &d2)
{
    const s16x2 ly01 = __builtin_bit_cast(s16x2, (u16x2)((bytes01(yw) * (unsigned short)149) >> 7));
    const s16x2 ly23 = __builtin_bit_cast(s16x2, (u16x2)((bytes23(yw) * (unsigned short)149) >> 7));
    const uint32_t RA = sat_pk_u8v(ly01 + t.rtl), GA = sat_pk_u8v(ly01 + t.gtl), BA = sat_pk_u8v(ly01 + t.btl);
    const uint32_t RB = sat_pk_u8v(ly23 + t.rth), GB = sat_pk_u8v(ly23 + t.gth), BB = sat_pk_u8v(ly23 + t.bth);
    const uint32_t W1 = GA | (BA << 16), W2 = RB | (GB << 16);
    d0 = (int)__builtin_amdgcn_perm(W1, RA, 0x01060400u);   // R0 G0 B0 R1
    d1 = (int)__builtin_amdgcn_perm(W2, W1, 0x06040301u);   // G1 B1 R2 G2
    d2 = (int)__builtin_amdgcn_perm(BB, W2, 0x05030104u);   // B2 R3 G3 B3
}
// two luma rows of 16 samples over one chroma row -> 2 x 48 bytes of RGB
__device__ __forceinline__ void rgb16x2(const uint4 ya, const uint4 yb, const uint2 cbv, const uint2 crv, v4i &a0, v4i &a1,
                                        v4i &a2, v4i &b0, v4i &b1, v4i &b2)
{
    int da[12], db[12];
    const uint32_t yaw[4] = {ya.x, ya.y, ya.z, ya.w}, ybw[4] = {yb.x, yb.y, yb.z, yb.w};
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const uint32_t cbw = (g < 2) ? cbv.x : cbv.y, crw = (g < 2) ? crv.x : crv.y;
        const RgbTerms t = (g & 1) ? rgb_terms(bytes23(cbw), bytes23(crw)) : rgb_terms(bytes01(cbw), bytes01(crw));
        rgb4_apply(yaw[g], t, da[3 * g], da[3 * g + 1], da[3 * g + 2]);
        rgb4_apply(ybw[g], t, db[3 * g], db[3 * g + 1], db[3 * g + 2]);
    }
    a0 = v4i{da[0], da[1], da[2], da[3]}; a1 = v4i{da[4], da[5], da[6], da[7]}; a2 = v4i{da[8], da[9], da[10], da[11]};
    b0 = v4i{db[0], db[1], db[2], db[3]}; b1 = v4i{db[4], db[5], db[6], db[7]}; b2 = v4i{db[8], db[9], db[10], db[11]};
}

// Plane-prediction gradient (h264_intra_prediction.c:2064-2080, :2491-2504): sum over i of (i+1) * (e[h+i] - e[h-2-i])
// with e[-1] = the corner, for 16 edge samples (h = 8, i < 8) as four byte dot products.
__device__ __forceinline__ int plane_grad16(const uint4 e, uint32_t cor)
{
    const uint32_t pos = __builtin_amdgcn_udot4(e.w, 0x08070605u, __builtin_amdgcn_udot4(e.z, 0x04030201u, 0u, false), false);
    const uint32_t e3456 = __builtin_amdgcn_alignbyte(e.y, e.x, 3);   // e[3], e[4], e[5], e[6]
    const uint32_t c012 = (e.x << 8) | cor;                           // corner, e[0], e[1], e[2]
    const uint32_t neg = __builtin_amdgcn_udot4(c012, 0x05060708u, __builtin_amdgcn_udot4(e3456, 0x01020304u, 0u, false), false);
    return (int)pos - (int)neg;
}
// ... for 8 edge samples (h = 4, i < 4)
__device__ __forceinline__ int plane_grad8(const uint2 e, uint32_t cor)
{
    const uint32_t pos = __builtin_amdgcn_udot4(e.y, 0x04030201u, 0u, false);
    const uint32_t c012 = (e.x << 8) | cor;                           // corner, e[0], e[1], e[2]
    const uint32_t neg = __builtin_amdgcn_udot4(c012, 0x01020304u, 0u, false);
    return (int)pos - (int)neg;
}

// quad_perm DPP controls
#define DPP_XOR1 0xB1  // [1,0,3,2]
#define DPP_XOR2 0x4E  // [2,3,0,1]
template <int CTRL>
__device__ __forceinline__ int dpp_quad(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
// value of lane k of this lane's quarter (qbase4 = byte address of the quarter's lane 0 = (lane & 48) * 4)
__device__ __forceinline__ uint32_t quarter_bcast(uint32_t v, int qbase4, int k)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute(qbase4 + k * 4, (int)v);
}

// 4-point transform with the matrix of h264_transform.c:62-68 (rows ++++, ++--, +--+, +-+-) across four
// lanes: `p` = the value of the lane whose index differs in the low index bit, then the lanes holding the
// pair sums / differences are fetched with ds_bpermute.  idx = this lane's index along the dimension.
__device__ __forceinline__ int had4_lanes(int x, int p, int idx, int addrP, int addrQ)
{
    const int t = (idx & 1) ? (p - x) : (x + p);       // idx 0: a = x0+x1, 1: b = x0-x1, 2: c = x2+x3, 3: e = x2-x3
    const int P = __builtin_amdgcn_ds_bpermute(addrP, t);
    const int Q = __builtin_amdgcn_ds_bpermute(addrQ, t);
    return (idx == 1 || idx == 2) ? (P - Q) : (P + Q); // a+c, a-c, b-e, b+e
}


} // namespace mvhp
