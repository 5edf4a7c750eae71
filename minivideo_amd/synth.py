"""Synthetic packed-macroblock pictures (numpy), for tests and for bench.py.

Generates the *output of the host entropy decoder* directly -- random, but legal,
packed records in the layout of include/minivideo_hotpath.h -- so that the GPU
reconstruction path can be exercised and measured without a bitstream.  Content
classes follow SURVEY.md section 8(d): "dense" (about half the 4x4 blocks coded,
several levels each) and "light" (Intra16x16 with one DC level / Intra4x4
without residual).  Every prediction mode drawn is one whose neighbours are
available at that position unless ``illegal_modes`` is set (then the
reference's "predict 0" behaviour is exercised as well).
"""
import numpy as np

from .hotpath import StreamParams, MB_BYTES

KIND_I4x4, KIND_I8x8, KIND_I16x16 = 0, 1, 2

# luma4x4BlkIdx -> (xO, yO) in samples
_BLK_XY = [(((b >> 2) & 1) * 8 + (b & 1) * 4, (b >> 3) * 8 + ((b >> 1) & 1) * 4) for b in range(16)]


def _avail(W, H):
    x = np.arange(W)[None, :].repeat(H, 0)
    y = np.arange(H)[:, None].repeat(W, 1)
    A = (x > 0).reshape(-1)
    B = (y > 0).reshape(-1)
    Cc = ((y > 0) & (x < W - 1)).reshape(-1)
    D = ((x > 0) & (y > 0)).reshape(-1)
    return A, B, Cc, D


def _choose(rng, allowed):
    """allowed: bool (..., M) -> random index of an allowed entry along the last axis."""
    score = rng.random(allowed.shape)
    score = np.where(allowed, score, -1.0)
    return score.argmax(-1).astype(np.uint8)


def _nxn_allowed(left, up, upleft, illegal):
    al = np.zeros(left.shape + (9,), dtype=bool)
    if illegal:
        al[...] = True
        return al
    al[..., 2] = True
    for m in (0, 3, 7):
        al[..., m] = up
    for m in (1, 8):
        al[..., m] = left
    for m in (4, 5, 6):
        al[..., m] = left & up & upleft
    return al


def synth_packed(width_mbs, height_mbs, n_frames, seed=1, profile="baseline", density="dense",
                 qp_range=(20, 40), cqp_offsets=(0, 0), illegal_modes=False, allow_qp36_i16=False, kinds=None):
    """Returns (StreamParams, packed[n_frames, W*H, 800] uint8).
    kinds = (P(Intra16x16), P(Intra8x8 | not Intra16x16)) overrides the profile's mix (measurement aid: content ablations)."""
    rng = np.random.default_rng(seed)
    W, H, F = int(width_mbs), int(height_mbs), int(n_frames)
    N = W * H
    A, B, Cc, D = _avail(W, H)
    A, B, Cc, D = (np.broadcast_to(v, (F, N)) for v in (A, B, Cc, D))

    # ---- macroblock kinds ----
    u = rng.random((F, N))
    if kinds is not None:
        kind = np.where(u < kinds[0], KIND_I16x16, KIND_I4x4)
        v = rng.random((F, N))
        kind = np.where((kind == KIND_I4x4) & (v < kinds[1]), KIND_I8x8, kind)
    elif density == "light":
        kind = np.where(u < 0.5, KIND_I16x16, KIND_I4x4)
    else:
        kind = np.where(u < 0.4, KIND_I16x16, KIND_I4x4)
        if profile == "high":
            v = rng.random((F, N))
            kind = np.where((kind == KIND_I4x4) & (v < 0.5), KIND_I8x8, kind)
    kind = kind.astype(np.uint8)

    # ---- prediction modes ----
    pred = np.zeros((F, N, 16), dtype=np.uint8)
    for b, (xO, yO) in enumerate(_BLK_XY):
        left = A | (xO > 0)
        up = B | (yO > 0)
        if xO > 0:
            upleft = B | (yO > 0)
        else:
            upleft = A if yO > 0 else D
        pred[..., b] = _choose(rng, _nxn_allowed(left, up, upleft, illegal_modes))
    pred8 = np.zeros((F, N, 4), dtype=np.uint8)
    for b in range(4):
        xO, yO = (b & 1) * 8, (b >> 1) * 8
        left = A | (xO > 0)
        up = B | (yO > 0)
        if xO > 0:
            upleft = B | (yO > 0)
        else:
            upleft = A if yO > 0 else D
        pred8[..., b] = _choose(rng, _nxn_allowed(left, up, upleft, illegal_modes))
    is8 = kind == KIND_I8x8
    pred[is8, :4] = pred8[is8]
    pred[is8, 4:] = 0
    pred[kind == KIND_I16x16] = 0

    al16 = np.zeros((F, N, 4), dtype=bool)
    al16[..., 2] = True
    al16[..., 0] = B | illegal_modes
    al16[..., 1] = A | illegal_modes
    al16[..., 3] = (A & B) | illegal_modes
    i16mode = _choose(rng, al16)
    alc = np.zeros((F, N, 4), dtype=bool)
    alc[..., 0] = True
    alc[..., 1] = A | illegal_modes
    alc[..., 2] = B | illegal_modes
    alc[..., 3] = (A & B) | illegal_modes
    cmode = _choose(rng, alc)

    # ---- QP ----
    qp = rng.integers(qp_range[0], qp_range[1] + 1, size=(F, N)).astype(np.uint8)
    if not allow_qp36_i16:
        qp = np.where((kind == KIND_I16x16) & (qp == 36), 37, qp).astype(np.uint8)

    # ---- levels ----
    coef = np.zeros((F, N, 384), dtype=np.int16)
    cbp_l = np.zeros((F, N), dtype=np.uint8)
    cbp_c = np.zeros((F, N), dtype=np.uint8)
    if density == "light":
        dcv = rng.choice(np.array([-3, -2, 2, 3], dtype=np.int16), size=(F, N))
        slot = rng.integers(0, 16, size=(F, N))
        sel = kind == KIND_I16x16
        fi, ni = np.nonzero(sel)
        coef[fi, ni, slot[sel] * 16] = dcv[sel]
    else:
        mag = np.minimum(rng.geometric(0.5, size=(F, N, 384)), 32).astype(np.int16)
        sign = np.where(rng.random((F, N, 384)) < 0.5, -1, 1).astype(np.int16)
        # per-coefficient density decays with frequency index inside the block
        pos = np.arange(384) % 16
        p_nz = (0.55 * np.exp(-pos / 6.0))[None, None, :]
        nzm = rng.random((F, N, 384)) < p_nz
        # block coded flags
        blk_coded = rng.random((F, N, 24)) < 0.5
        nzm &= np.repeat(blk_coded, 16, axis=-1)
        # luma 8x8 cbp
        cbp_bits = rng.random((F, N, 4)) < 0.6
        i16 = kind == KIND_I16x16
        all_or_none = rng.random((F, N)) < 0.5
        cbp_bits = np.where(i16[..., None], all_or_none[..., None], cbp_bits)
        luma_gate = np.repeat(cbp_bits, 64, axis=-1)  # blocks 4k..4k+3 belong to 8x8 block k
        lum = nzm[..., :256] & luma_gate
        # Intra16x16: DC levels (slot 0) are coded independently of cbp
        dc_slots = np.zeros(256, dtype=bool)
        dc_slots[::16] = True
        dc_nz = rng.random((F, N, 256)) < 0.35
        lum = np.where(i16[..., None] & dc_slots[None, None, :], dc_nz, lum)
        # chroma cbp: 0 none, 1 DC only, 2 DC+AC
        cc = rng.choice(np.array([0, 1, 2], dtype=np.uint8), p=[0.4, 0.3, 0.3], size=(F, N))
        cdc = np.zeros(128, dtype=bool)
        cdc[::16] = True
        chroma_dc_nz = rng.random((F, N, 128)) < 0.5
        chr_ = np.where(cdc[None, None, :], chroma_dc_nz & (cc[..., None] >= 1), nzm[..., 256:] & (cc[..., None] == 2))
        full = np.concatenate([lum, chr_], axis=-1)
        coef = np.where(full, mag * sign, 0).astype(np.int16)
        cbp_l = (cbp_bits * np.array([1, 2, 4, 8])).sum(-1).astype(np.uint8)
        cbp_c = cc

    # ---- nz_mask ----
    blk_nz = (coef.reshape(F, N, 24, 16) != 0).any(-1)
    luma_nz = blk_nz[..., :16].copy()
    g8 = luma_nz.reshape(F, N, 4, 4).any(-1)
    luma_nz = np.where(is8[..., None], np.repeat(g8, 4, axis=-1), luma_nz)
    bits = np.concatenate([luma_nz, blk_nz[..., 16:]], axis=-1)
    nz_mask = (bits.astype(np.uint32) << np.arange(24, dtype=np.uint32)).sum(-1).astype(np.uint32)

    # ---- assemble records ----
    rec = np.zeros((F, N, MB_BYTES), dtype=np.uint8)
    rec[..., 0] = kind
    rec[..., 1] = qp
    rec[..., 2] = cbp_l | (cbp_c << 4)
    rec[..., 3] = cmode
    rec[..., 4] = np.where(kind == KIND_I16x16, i16mode, 0)
    rec[..., 8:12] = nz_mask.view(np.uint8).reshape(F, N, 4)
    rec[..., 12:28] = pred
    rec[..., 32:] = coef.view(np.uint8).reshape(F, N, 768)

    # flags bit 0 = MVHP_PARAM_MAY_HAVE_8X8, as the front end sets it from the PPS (a hint for the choice of kernel form only)
    params = StreamParams(W, H, int(cqp_offsets[0]), int(cqp_offsets[1]), 1 if bool(np.any(kind == KIND_I8x8)) else 0)
    return params, rec
