#!/usr/bin/env python3
"""Writes the two known-answer Annex-B streams of SURVEY.md Appendix A (bytes quoted
there; they were hand-encoded by the survey session and decoded by the real
MiniVideo reference: -f yuv420 -> 512 x 0x83 then 256 x 0x80,
md5 2d87b01fcabfeea0afc04f99c5b30083; RGB (134,133,134))."""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = {
    "kat_cavlc_2mb.264": ("00 00 00 01 67 42 00 1e f9 72 00 00 00 01 68 ce 38 80 "
                          "00 00 00 01 65 88 84 02 13 14 da e0", "e94883181bcd50ea61aacf46d1e73540"),
    "kat_cabac_2mb.264": ("00 00 00 01 67 4d 00 1e f9 72 00 00 00 01 68 ee 38 80 "
                          "00 00 00 01 65 88 84 02 7f fd 09 bc 0e 67", "7b2b5a7fbbabea0511523ae2b5a9359c"),
}
for name, (hx, md5) in KATS.items():
    data = bytes.fromhex(hx.replace(" ", "")) + bytes(64)
    assert hashlib.md5(data).hexdigest() == md5, name
    open(os.path.join(HERE, name), "wb").write(data)
    print(name, len(data), md5)
