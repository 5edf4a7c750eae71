"""Hand-derived known answers for the reconstruction stages, one per family -- in the style of SURVEY.md Appendix A,
but WITHOUT a reference run behind them: every expected sample below is worked out on paper from the reference's source
text (file:line cited at each step) for records small enough to follow.  They pin the oracle (oracle/recon_ref.c) and,
under -m gpu, the three HIP kernel layouts, against an independent derivation; they do NOT lift "parity unpinned"
(DESIGN.md 5): only reference output could, and the reference cannot be built here.

Conventions.  Flat macroblocks are made the Appendix-A way: Intra16x16, DC prediction, one luma DC level k at QP'Y 28
adds exactly k to every luma sample [f = k in all 16 positions (h264_transform.c:62-68, 771-782); LevelScale4x4(4,0,0)
= 16*16 = 256; qP/6 = 4 < 6: dcY = (k*256 + 2) >> 2 = 64k (:803-808); a block whose only coefficient is d00 = 64k gives
(64k + 32) >> 6 = k everywhere (:1145-1191)] -- this is the k = 3 -> 131 of Appendix A, which the reference produced.
One chroma DC level k at QP 28 adds 2k to the plane [f = k x 4; dcC = ((k*256) << 4) >> 5 = 128k (:924-936);
(128k + 32) >> 6 = 2k].  Macroblock x = 0 / y = 0 neighbours are unavailable: DC prediction falls back to the side
that exists, or 128 (h264_intra_prediction.c:2017-2083, 2357-2437).
"""
import numpy as np
import pytest

from minivideo_amd.hotpath import StreamParams
from oracle import loader

I4, I8, I16 = 0, 1, 2
# luma4x4BlkIdx -> (x, y) of its top-left sample (h264_spatial.c:210)
BLK_XY = [(((b >> 2) & 1) * 8 + (b & 1) * 4, (b >> 3) * 8 + ((b >> 1) & 1) * 4) for b in range(16)]


def mb(kind, qp=28, i16=2, cmode=0, modes=(), luma=None, cb=None, cr=None):
    """One 800-byte packed record (include/minivideo_hotpath.h).  luma / cb / cr: {(block, row, col): level}; for an
    Intra16x16 DC level or a chroma DC level use (block, 0, 0) -- slot 0 of the block it feeds."""
    rec = np.zeros(800, np.uint8)
    rec[0], rec[1], rec[3], rec[4] = kind, qp, cmode, i16
    for i, m in enumerate(modes):
        rec[12 + i] = m
    coef = rec[32:].view(np.int16)
    nz = 0
    for (b, r, c), v in (luma or {}).items():
        coef[(b * 64 if kind == I8 else b * 16) + (r * 8 + c if kind == I8 else r * 4 + c)] = v
        nz |= (0xf << (4 * b)) if kind == I8 else (1 << b)
    for pl, d in ((0, cb), (1, cr)):
        for (b, r, c), v in (d or {}).items():
            coef[256 + pl * 64 + b * 16 + r * 4 + c] = v
            nz |= 1 << (16 + 4 * pl + b)
    rec[8:12] = np.array([nz], np.uint32).view(np.uint8)
    return rec


def flat(k, ck=0, crk=0):
    """Intra16x16 / chroma DC prediction, luma DC level k, Cb / Cr DC levels ck / crk at QP 28: adds k (2ck, 2crk)."""
    return mb(I16, 28, i16=2, cmode=0, luma={(0, 0, 0): k} if k else None,
              cb={(0, 0, 0): ck} if ck else None, cr={(0, 0, 0): crk} if crk else None)


def planes(yuv, W, H):
    y = yuv[:W * H * 256].reshape(H * 16, W * 16)
    cb = yuv[W * H * 256:W * H * 320].reshape(H * 8, W * 8)
    cr = yuv[W * H * 320:].reshape(H * 8, W * 8)
    return y, cb, cr


CASES = {}


def case(fn):
    CASES[fn.__name__] = fn
    return fn


# ---------------------------------------------------------------------------------------------------------------
@case
def ac4x4_above_and_below_the_shift_split():
    """transform_4x4_residual (h264_transform.c:1049-1191), one AC level c[0][1] = 2 in block 0 of a lone Intra4x4
    macroblock (block 0 has no neighbours: DC prediction = 128, h264_intra_prediction.c:568-632).
    quant4x4 (:1100-1134): position (0,1) is neither (even,even) nor (odd,odd): LevelScale = 16 * v[qP%6][2].
      QP 28: qP%6 = 4 -> v = 20 -> LS = 320; qP >= 24: d = (2*320) << (4-4) = 640.
      QP 20: qP%6 = 2 -> v = 16 -> LS = 256; qP < 24:  d = (2*256 + 2^(3-3)) >> (4-3) = 513 >> 1 = 256.
    idct4x4 (:1145-1191), only d01 set: row 0: e = (0, 0, d>>1, d) -> f = (d, d>>1, -(d>>1), -d); other rows 0; the
    column pass copies row 0 into every row; r = (f + 32) >> 6 (arithmetic shift):
      d = 640: (672, 352, -288, -608) >> 6 = (10, 5, -5, -10) -> 138 133 123 118
      d = 256: (288, 160,  -96, -224) >> 6 = ( 4, 2, -2,  -4) -> 132 130 126 124"""
    out = []
    for qp, row in ((28, (138, 133, 123, 118)), (20, (132, 130, 126, 124))):
        rec = mb(I4, qp, modes=[2] * 16, luma={(0, 0, 1): 2})
        out.append((f"QP {qp}", (1, 1), [rec], {"y": ((0, 0), np.array([row] * 4))}))
    return out


@case
def chroma_dc_and_ac():
    """transform4x4_chroma (h264_transform.c:286-402): Cb DC level 4 + one AC level c[1][0] = 1 in Cb block 0, QP'Y 28
    (QPc = 28, :598-637), chroma DC prediction without neighbours = 128.
    DC: c = [[4,0],[0,0]] -> f = 4 everywhere (:988-1005); dcC = ((4*256) << 4) >> 5 = 512 (:924-936) = d00 of every
    Cb block: blocks 1-3 get (512+32) >> 6 = 8 -> 136.
    Block 0 also has d10 = 1 * LS(4; 1,0) = 320 (keep_dc rule :1126-1129 leaves d00 = 512).  Row pass: row 0 -> 512 x4,
    row 1 -> 320 x4.  Column pass per column: g0 = g1 = 512, g2 = 320 >> 1 = 160, g3 = 320 -> h = (832, 672, 352, 192);
    (h + 32) >> 6 = (13, 11, 6, 3) -> rows 141, 139, 134, 131.  Cr untouched = 128; luma (Intra16x16 DC, nothing coded)
    = 128."""
    rec = mb(I16, 28, i16=2, cmode=0, cb={(0, 0, 0): 4, (0, 1, 0): 1})
    cb = np.full((8, 8), 136)
    cb[0:4, 0:4] = np.array([[141] * 4, [139] * 4, [134] * 4, [131] * 4])
    return [("", (1, 1), [rec], {"y": ((0, 0), np.full((16, 16), 128)), "cb": ((0, 0), cb), "cr": ((0, 0), np.full((8, 8), 128))})]


@case
def residual8x8_dc_below_and_above_36_and_ac():
    """transform_8x8_residual (h264_transform.c:1205-1383) in 8x8 block 0 of a lone Intra8x8 macroblock (DC prediction
    without neighbours = 128, h264_intra_prediction.c:1435-1500).  LevelScale8x8 = 16 * v8x8[qP%6][class] (h264.c:438-446):
    qP%6 = 4: class (0,0) = 32 -> 512; class "row%4==0, col odd" = 30 -> 480.
      DC level 3, QP 28 (< 36): d00 = (3*512 + 2^(5-4)) >> (6-4) = 1538 >> 2 = 384 (:1256-1284); a lone d00 passes both
        butterflies unchanged (:1308-1378): r = (384 + 32) >> 6 = 6 -> 134.
      DC level 1, QP 40 (>= 36): d00 = (1*512) << (6-6) = 512 -> (512 + 32) >> 6 = 8 -> 136.
      AC level c[0][1] = 1, QP 40: d01 = 480.  Row butterfly with only d1 set (:1308-1342): e3 = 480, e5 = -480,
        e7 = 480 + 240 = 720, e1 = 0; f1 = e1 + (e7>>2) = 180, f3 = e3 + (e5>>2) = 360, f5 = (e3>>2) - e5 = 600,
        f7 = e7 - (e1>>2) = 720; out = (f0+f7, f2+f5, f4+f3, f6+f1, f6-f1, f4-f3, f2-f5, f0-f7)
        = (720, 600, 360, 180, -180, -360, -600, -720); columns copy it down; (v + 32) >> 6 =
        (11, 9, 6, 3, -3, -6, -9, -11) -> 139 137 134 131 125 122 119 117 in every row."""
    out = []
    for name, qp, lv, row in (("DC QP28", 28, {(0, 0, 0): 3}, [134] * 8), ("DC QP40", 40, {(0, 0, 0): 1}, [136] * 8),
                              ("AC QP40", 40, {(0, 0, 1): 1}, [139, 137, 134, 131, 125, 122, 119, 117])):
        rec = mb(I8, qp, modes=[2] * 4, luma=lv)
        out.append((name, (1, 1), [rec], {"y": ((0, 0), np.array([row] * 8))}))
    return out


@case
def plane_prediction_luma_and_chroma():
    """Intra_16x16_Plane (h264_intra_prediction.c:2096-2141) and Intra_Chroma_Plane (:2524-2564) in macroblock 3 of a
    2x2 picture whose other macroblocks are flat: MB0 = 128 (corner p[-1,-1]), MB1 (above) = 128 + 16 = 144 luma,
    128 + 2*6 = 140 Cb; MB2 (left) = 128 - 8 = 120 luma, 128 - 2*4 = 120 Cb; Cr = 128 everywhere.
    [MB1: DC prediction from its left neighbour MB0 = (16*128 + 8) >> 4 = 128, + level; MB2: from its top neighbour.]
    Luma: H = sum_{i<8} (i+1)(p[8+i,-1] - p[6-i,-1]); the top row is flat, only i = 7 reaches the corner:
      H = 8*(144-128) = 128, V = 8*(120-128) = -64; a = 16*(p[-1,15] + p[15,-1]) = 16*264 = 4224;
      b = (5H + 32) >> 6 = 672 >> 6 = 10; c = (5V + 32) >> 6 = -288 >> 6 = -5;
      pred[x,y] = (a + b(x-7) + c(y-7) + 16) >> 5 = (4205 + 10x - 5y) >> 5   (no clipping in range).
    Cb: H = 4*(140-128) = 48, V = 4*(120-128) = -32; a = 16*(120+140) = 4160; b = (34*48 + 32) >> 6 = 1664 >> 6 = 26;
      c = (34*(-32) + 32) >> 6 = -1056 >> 6 = -17; pred = (4160 + 26(x-3) - 17(y-3) + 16) >> 5 = (4149 + 26x - 17y) >> 5.
    Cr: flat 128 around -> H = V = 0, a = 4096: (4096 + 16) >> 5 = 128."""
    recs = [flat(0), flat(16, ck=6), flat(-8, ck=-4), mb(I16, 28, i16=3, cmode=3)]
    x, y = np.meshgrid(np.arange(16), np.arange(16))
    cx, cy = np.meshgrid(np.arange(8), np.arange(8))
    return [("", (2, 2), recs, {"y": ((16, 16), (4205 + 10 * x - 5 * y) >> 5),
                                "cb": ((8, 8), (4149 + 26 * cx - 17 * cy) >> 5),
                                "cr": ((8, 8), np.full((8, 8), 128))})]


def _three_tone_4x4(mode):
    """Block 0 of macroblock 4 in a 3x2 picture of flat neighbours: corner p[-1,-1] = a (MB0), top p[0..7,-1] = b (MB1;
    block 0's up-right samples lie in MB1 too), left p[-1,0..3] = c (MB3).  Each mode function
    (h264_intra_prediction.c:496-960) evaluated on that edge; (p + 2q + r + 2) >> 2 with two equal taps written out."""
    a, b, c = 128, 148, 108
    tb, tc = (a + 3 * b + 2) >> 2, (a + 3 * c + 2) >> 2     # one tap on the corner, two/one on a flat side: 143, 113
    mid = (b + 2 * a + c + 2) >> 2                           # :704 (x == y): 128
    P = np.zeros((4, 4), int)                                # [y][x]
    for y in range(4):
        for x in range(4):
            if mode == 0: v = b                              # Vertical :496-518
            elif mode == 1: v = c                            # Horizontal :531-553
            elif mode == 2: v = (4 * b + 4 * c + 4) >> 3     # DC, both sides :583-596 = 128
            elif mode == 3: v = b                            # Diagonal_Down_Left :647-677: top and up-right are all b
            elif mode == 4:                                  # Diagonal_Down_Right :690-723
                v = mid if x == y else ((tb if x - y == 1 else b) if x > y else (tc if y - x == 1 else c))
            elif mode == 5:                                  # Vertical_Right :736-778, zVR = 2x - y
                z = 2 * x - y
                if z >= 0 and z % 2 == 0:                    # (p[x-(y>>1)-1,-1] + p[x-(y>>1),-1] + 1) >> 1
                    v = (a + b + 1) >> 1 if x - (y >> 1) == 0 else b
                elif z >= 0:                                 # three taps on the top row, leftmost may be the corner
                    v = tb if x - (y >> 1) - 2 == -1 else b
                elif z == -1: v = mid                        # (p[-1,0] + 2 p[-1,-1] + p[0,-1] + 2) >> 2
                else: v = tc if y - 2 * x - 3 == -1 else c   # (p[-1,y-2x-1] + 2 p[-1,y-2x-2] + p[-1,y-2x-3] + 2) >> 2: y=2,x=0 -> (c, c, corner)
            elif mode == 6:                                  # Horizontal_Down :791-833, zHD = 2y - x
                z = 2 * y - x
                if z >= 0 and z % 2 == 0: v = (a + c + 1) >> 1 if y - (x >> 1) == 0 else c
                elif z >= 0: v = tc if y - (x >> 1) - 2 == -1 else c
                elif z == -1: v = mid
                else: v = tb if x - 2 * y - 3 == -1 else b   # (p[x-2y-1,-1] + 2 p[x-2y-2,-1] + p[x-2y-3,-1] + 2) >> 2: x=2,y=0
            elif mode == 7: v = b                            # Vertical_Left :846-876: top / up-right only
            else: v = c                                      # Horizontal_Up :889-960: left column only
            P[y, x] = v
    return P


@case
def intra4x4_nine_modes_on_a_three_tone_edge():
    """see _three_tone_4x4.  MB0/MB1/MB2 flat 128 / 148 / 148, MB3 flat 108 (DC levels 0, +20, 0 relative ..., each
    predicted from the flat neighbour it has), MB4 = Intra4x4 with block 0 in the mode under test and DC elsewhere."""
    out = []
    for mode in range(9):
        # MB1 = 128 + 20 (left neighbour MB0 = 128); MB2 = 148 + 0 (left neighbour MB1); MB3 = 128 - 20 (top neighbour MB0)
        recs = [flat(0), flat(20), flat(0), flat(-20), mb(I4, 28, modes=[mode] + [2] * 15), flat(0)]
        out.append((f"mode {mode}", (3, 2), recs, {"y": ((16, 16), _three_tone_4x4(mode))}))
    return out


@case
def intra4x4_up_right_rules():
    """Up-right samples of an Intra4x4 block (h264_intra_prediction.c:400-439), Diagonal_Down_Left (:647-677):
    pred[x,y] = (p[x+y] + 2 p[x+y+1] + p[x+y+2] + 2) >> 2 over the eight samples p[0..7,-1] (x = y = 3: (p6 + 3 p7 + 2) >> 2).
    3x2 picture, MB1 (above MB4) = 148, MB2 (above right) = 108, MB4 Intra4x4, every block Vertical (copies 148 down)
    except the block under test.
      block 5 (x 12-15, y 0-3): up-right = macroblock C = MB2 (by geometry): p = 148 x4, 108 x4:
        x+y = 0,1 -> 148; 2 -> (148+296+108+2)>>2 = 138; 3 -> (148+216+108+2)>>2 = 118; >= 4 -> 108.
      block 3 (x 4-7, y 4-7): up-right is declared unavailable (:410-412, blkIdx 3 and 11) -> p[4..7] = p[3] (:431-439):
        everything 148 -- although the samples up-right of it (block 6) do not exist yet anyway.
      block 1 (x 4-7, y 0-3): up-right = MB1, x 8-11 = 148: everything 148 (available and equal)."""
    recs0 = [flat(0), flat(20), flat(-40), flat(0), None, flat(0)]     # MB2 = 148 - 40 = 108 (left neighbour MB1)
    out = []
    ddl = np.array([[148, 148, 138, 118], [148, 138, 118, 108], [138, 118, 108, 108], [118, 108, 108, 108]])
    for blk, exp in ((5, ddl), (3, np.full((4, 4), 148)), (1, np.full((4, 4), 148))):
        modes = [0] * 16
        modes[blk] = 3
        recs = list(recs0)
        recs[4] = mb(I4, 28, modes=modes)
        bx, by = BLK_XY[blk]
        out.append((f"block {blk}", (3, 2), recs, {"y": ((16 + bx, 16 + by), exp)}))
    return out


@case
def intra8x8_filtered_corner():
    """Intra_8x8_sample_filtering (h264_intra_prediction.c:1295-1353) on the three-tone edge of 8x8 block 0 of MB4 (3x2
    picture: corner a = 128, top b = 148 over 16 samples, left c = 108), then Vertical / Horizontal / DC on the FILTERED edge:
      p'[-1,-1] = (p[0,-1] + 2 p[-1,-1] + p[-1,0] + 2) >> 2 = (148 + 256 + 108 + 2) >> 2 = 128      (:1331-1345, both sides there)
      p'[0,-1]  = (p[-1,-1] + 2 p[0,-1] + p[1,-1] + 2) >> 2 = (128 + 444 + 2) >> 2 = 143            (:1302-1306)
      p'[x,-1]  = 148 for x = 1..15 (flat; x = 15: (p14 + 3 p15 + 2) >> 2 = 148)
      p'[-1,0]  = (p[-1,-1] + 2 p[-1,0] + p[-1,1] + 2) >> 2 = (128 + 324 + 2) >> 2 = 113            (:1347-1351)
      p'[-1,y]  = 108 for y = 1..7
    Vertical (:1366-1388): column x = p'[x,-1] = 143, 148 x7.  Horizontal (:1401-1423): row y = p'[-1,y] = 113, 108 x7.
    DC (:1435-1500): (sum p'[0..7,-1] + sum p'[-1,0..7] + 8) >> 4 = (143 + 7*148 + 113 + 7*108 + 8) >> 4 = 2056 >> 4 = 128."""
    out = []
    v = np.array([[143] + [148] * 7] * 8)
    h = np.array([[113] * 8] + [[108] * 8] * 7)
    for mode, exp in ((0, v), (1, h), (2, np.full((8, 8), 128))):
        recs = [flat(0), flat(20), flat(0), flat(-20), mb(I8, 28, modes=[mode, 2, 2, 2]), flat(0)]
        out.append((f"mode {mode}", (3, 2), recs, {"y": ((16, 16), exp)}))
    return out


def _all_cases():
    for name, fn in CASES.items():
        for sub, (W, H), recs, expect in fn():
            yield pytest.param(W, H, np.stack(recs)[None], expect, id=f"{name}[{sub}]" if sub else name)


def _compare(yuv, W, H, expect):
    y, cb, cr = planes(yuv, W, H)
    for plane, ((x0, y0), exp) in expect.items():
        got = {"y": y, "cb": cb, "cr": cr}[plane][y0:y0 + exp.shape[0], x0:x0 + exp.shape[1]]
        assert np.array_equal(got, exp), (plane, got, exp)


@pytest.mark.parametrize("W,H,rec,expect", list(_all_cases()))
def test_oracle_against_hand_derivation(W, H, rec, expect):
    yuv, _ = loader.recon(StreamParams(W, H, 0, 0, 1), rec, 1)
    _compare(yuv, W, H, expect)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,rec,expect", list(_all_cases()))
def test_gpu_against_hand_derivation(hot, W, H, rec, expect):
    yuv, _ = hot.recon_host(StreamParams(W, H, 0, 0, 1), rec, 1)
    _compare(yuv, W, H, expect)
