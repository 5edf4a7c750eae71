"""Minimal MP4 muxer for tests: wraps the NAL units of an Annex-B stream (as produced by minivideo_amd.gen) into
ftyp / mdat / moov with one H.264 video track (avc1 + avcC, stts, stss, stsc, stsz, stco|co64)."""
import struct


def split_annexb(data):
    b = bytes(data)
    pos, nals = [], []
    i = 0
    while True:
        j = b.find(b"\x00\x00\x00\x01", i)
        if j < 0:
            break
        pos.append(j + 4)
        i = j + 4
    for k, p in enumerate(pos):
        end = pos[k + 1] - 4 if k + 1 < len(pos) else len(b)
        nal = b[p:end].rstrip(b"\x00") if k + 1 == len(pos) else b[p:end]
        nals.append(nal)
    return nals


def box(typ, payload):
    return struct.pack(">I4s", 8 + len(payload), typ) + payload


def full(typ, version, flags, payload):
    return box(typ, struct.pack(">I", (version << 24) | flags) + payload)


def mux(annexb, width, height, samples_per_chunk=1, use_co64=False, moov_first=False, extra_non_sync=False,
        inband_params=False, length_size=4):
    nals = split_annexb(annexb)
    sps = [n for n in nals if n[0] & 31 == 7][:1]
    pps = [n for n in nals if n[0] & 31 == 8][:1]
    idr = [n for n in nals if n[0] & 31 == 5]

    def pfx(n):
        return len(n).to_bytes(length_size, "big") + n

    samples, sync = [], []
    for k, n in enumerate(idr):
        body = b""
        if inband_params and k % 2 == 1:
            body += pfx(sps[0]) + pfx(pps[0])
        body += pfx(b"\x06\x05\x01\xaa\x80")          # an SEI NAL in front of the slice
        body += pfx(n)
        samples.append(body)
        sync.append(True)
        if extra_non_sync:
            samples.append(pfx(b"\x41\x9a\x00\x10\x20"))   # a (fake) non-IDR slice sample
            sync.append(False)
    ftyp = box(b"ftyp", b"isom" + struct.pack(">I", 512) + b"isomiso2avc1mp41")
    mdat_payload = b"".join(samples)

    def build_moov(mdat_data_offset):
        offs, o = [], mdat_data_offset
        for s in samples:
            offs.append(o)
            o += len(s)
        chunks = [offs[i] for i in range(0, len(samples), samples_per_chunk)]
        avcC = box(b"avcC", bytes([1, sps[0][1], sps[0][2], sps[0][3], 0xFC | (length_size - 1), 0xE0 | 1]) +
                   struct.pack(">H", len(sps[0])) + sps[0] + bytes([1]) + struct.pack(">H", len(pps[0])) + pps[0])
        visual = (b"\x00" * 6 + struct.pack(">H", 1) + b"\x00" * 16 + struct.pack(">HH", width, height) +
                  struct.pack(">II", 0x00480000, 0x00480000) + b"\x00" * 4 + struct.pack(">H", 1) + b"\x00" * 32 +
                  struct.pack(">H", 24) + struct.pack(">h", -1))
        avc1 = box(b"avc1", visual + avcC)
        stsd = full(b"stsd", 0, 0, struct.pack(">I", 1) + avc1)
        stts = full(b"stts", 0, 0, struct.pack(">III", 1, len(samples), 1000))
        ss = [i + 1 for i, s in enumerate(sync) if s]
        stss = full(b"stss", 0, 0, struct.pack(">I", len(ss)) + b"".join(struct.pack(">I", x) for x in ss))
        n_full, rem = divmod(len(samples), samples_per_chunk)
        runs = [(1, samples_per_chunk, 1)] if n_full else []
        if rem:
            runs.append((n_full + 1, rem, 1))
        stsc = full(b"stsc", 0, 0, struct.pack(">I", len(runs)) + b"".join(struct.pack(">III", *r) for r in runs))
        stsz = full(b"stsz", 0, 0, struct.pack(">II", 0, len(samples)) + b"".join(struct.pack(">I", len(s)) for s in samples))
        if use_co64:
            stco = full(b"co64", 0, 0, struct.pack(">I", len(chunks)) + b"".join(struct.pack(">Q", c) for c in chunks))
        else:
            stco = full(b"stco", 0, 0, struct.pack(">I", len(chunks)) + b"".join(struct.pack(">I", c) for c in chunks))
        stbl = box(b"stbl", stsd + stts + (stss if not all(sync) or True else b"") + stsc + stsz + stco)
        vmhd = full(b"vmhd", 0, 1, b"\x00" * 8)
        dref = full(b"dref", 0, 0, struct.pack(">I", 1) + full(b"url ", 0, 1, b""))
        minf = box(b"minf", vmhd + box(b"dinf", dref) + stbl)
        hdlr = full(b"hdlr", 0, 0, b"\x00" * 4 + b"vide" + b"\x00" * 12 + b"VideoHandler\x00")
        mdhd = full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, 25000, 1000 * len(samples), 0x55C4, 0))
        mdia = box(b"mdia", mdhd + hdlr + minf)
        tkhd = full(b"tkhd", 0, 7, struct.pack(">IIIII", 0, 0, 1, 0, 40 * len(samples)) + b"\x00" * 8 +
                    struct.pack(">hhhh", 0, 0, 0, 0) + struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000) +
                    struct.pack(">II", width << 16, height << 16))
        trak = box(b"trak", tkhd + mdia)
        mvhd = full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, 1000, 40 * len(samples)) + struct.pack(">IH", 0x10000, 0x100) +
                    b"\x00" * 10 + struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000) + b"\x00" * 24 +
                    struct.pack(">I", 2))
        return box(b"moov", mvhd + trak)

    if moov_first:
        moov = build_moov(0)
        moov = build_moov(len(ftyp) + len(moov) + 8)
        return ftyp + moov + box(b"mdat", mdat_payload)
    mdat = box(b"mdat", mdat_payload)
    return ftyp + mdat + build_moov(len(ftyp) + 8)
