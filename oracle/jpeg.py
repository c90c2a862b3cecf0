"""Oracle: baseline-JPEG decode to the BGR uint8 image `cv2.imread` returns.  TEST INFRASTRUCTURE ONLY.

Path row N3: /root/reference/src/utils/make_submit.py:56-67 and
/root/reference/src/utils/export_line_result.py:176-177 call `cv2.imread(path)` on every frame.  The decoder
itself is a third-party dependency that is NOT in /root/reference: opencv-python==4.7.0.72
(/root/reference/requirements.txt:5), which bundles libjpeg-turbo 2.1.x and runs it with the library
defaults -- dct_method JDCT_ISLOW, do_fancy_upsampling TRUE, output JCS_RGB swapped to BGR.  This file
restates the published libjpeg-turbo algorithms for that configuration:
    entropy decode      jdhuff.c   decode_mcu_slow / HUFF_EXTEND, restart handling jdhuff.c process_restart
    inverse DCT         jidctint.c jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2), range limit jdmaster.c
                        prepare_range_limit_table (index masked with RANGE_MASK = 1023)
    chroma upsampling   jdsample.c h2v2_fancy_upsample / h2v1_fancy_upsample (triangle filter, alternating bias),
                        bottom/top row replication of jdmainct.c (set_bottom_pointers)
    colour conversion   jdcolor.c  build_ycc_rgb_table / ycc_rgb_convert (16-bit fixed point)
Pinned by tests/golden/jpeg_*.npz: JPEG byte streams written by Pillow's encoder and the pixels the
libjpeg-turbo inside Pillow decodes from them (tools/make_golden.py gen_jpeg; cv2 is not installed in the build
image, and both wrap the same library with the same defaults).

Scope: 8-bit sequential DCT (SOF0/SOF1), Huffman, one interleaved scan, 1 component (grey -> B=G=R as
cv2.IMREAD_COLOR) or 3 components YCbCr with luma sampling 1x1 / 2x1 / 2x2 and chroma 1x1.  Anything else raises
(progressive, arithmetic coding, CMYK, 4:4:0, multi-scan); EXIF orientation is not applied (SoccerNet frames
carry none).
"""
from typing import Dict, List

import numpy as np

ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
    54, 47, 55, 62, 63], dtype=np.int64)            # zigzag position -> natural (row-major) position


class JpegError(ValueError):
    pass


def _u16(d, i):
    return (d[i] << 8) | d[i + 1]


def parse(data: bytes) -> Dict:
    """Marker segments up to the entropy-coded scan (ITU T.81 annex B)."""
    d = memoryview(data)
    if len(d) < 4 or d[0] != 0xFF or d[1] != 0xD8:
        raise JpegError("not a JPEG (no SOI)")
    qt: Dict[int, np.ndarray] = {}
    huff: Dict = {}
    frame = None
    restart = 0
    i = 2
    while True:
        if i + 4 > len(d):
            raise JpegError("truncated before SOS")
        if d[i] != 0xFF:
            raise JpegError("marker expected")
        m = d[i + 1]
        if m == 0xFF:
            i += 1
            continue
        if m == 0xD8 or (0xD0 <= m <= 0xD7) or m == 0x01:
            i += 2
            continue
        L = _u16(d, i + 2)
        seg = d[i + 4:i + 2 + L]
        if len(seg) != L - 2:
            raise JpegError("truncated segment")
        if m == 0xDB:
            j = 0
            while j < len(seg):
                pq, tq = seg[j] >> 4, seg[j] & 15
                j += 1
                t = np.zeros(64, np.int64)
                for k in range(64):
                    if pq:
                        t[ZIGZAG[k]] = _u16(seg, j); j += 2
                    else:
                        t[ZIGZAG[k]] = seg[j]; j += 1
                qt[tq] = t
        elif m == 0xC4:
            j = 0
            while j < len(seg):
                tc, th = seg[j] >> 4, seg[j] & 15
                counts = [seg[j + 1 + k] for k in range(16)]
                j += 17
                n = sum(counts)
                huff[(tc, th)] = (counts, [seg[j + k] for k in range(n)])
                j += n
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise JpegError("only 8-bit precision")
            frame = dict(height=_u16(seg, 1), width=_u16(seg, 3),
                         comps=[dict(id=seg[6 + 3 * c], h=seg[7 + 3 * c] >> 4, v=seg[7 + 3 * c] & 15, tq=seg[8 + 3 * c])
                                for c in range(seg[5])])
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegError("unsupported JPEG process (SOF%d)" % (m - 0xC0))
        elif m == 0xDD:
            restart = _u16(seg, 0)
        elif m == 0xDA:
            if frame is None:
                raise JpegError("SOS before SOF")
            ns = seg[0]
            if ns != len(frame["comps"]):
                raise JpegError("multi-scan files are not supported")
            for c in range(ns):
                comp = frame["comps"][c]
                if seg[1 + 2 * c] != comp["id"]:
                    raise JpegError("scan component order")
                comp["td"], comp["ta"] = seg[2 + 2 * c] >> 4, seg[2 + 2 * c] & 15
            frame.update(qt=qt, huff=huff, restart=restart, scan=bytes(d[i + 2 + L:]))
            break
        elif m == 0xD9:
            raise JpegError("EOI before SOS")
        i += 2 + L
    comps = frame["comps"]
    if len(comps) == 1:
        comps[0]["h"] = comps[0]["v"] = 1               # a single-component scan is never interleaved
    elif len(comps) == 3:
        if (comps[1]["h"], comps[1]["v"], comps[2]["h"], comps[2]["v"]) != (1, 1, 1, 1) or \
                (comps[0]["h"], comps[0]["v"]) not in ((1, 1), (2, 1), (2, 2)):
            raise JpegError("unsupported sampling factors")
    else:
        raise JpegError("unsupported component count")
    if frame["width"] == 0 or frame["height"] == 0:
        raise JpegError("empty frame")
    return frame


class _Bits:
    """Entropy-coded segment reader: removes 0xFF00 stuffing, stops at markers (jdhuff.c jpeg_fill_bit_buffer)."""

    def __init__(self, data: bytes):
        self.d, self.i, self.acc, self.n = data, 0, 0, 0

    def _fill(self):
        while self.n <= 24:
            b = 0
            if self.i < len(self.d):
                b = self.d[self.i]
                if b == 0xFF:
                    nxt = self.d[self.i + 1] if self.i + 1 < len(self.d) else 0xD9
                    if nxt == 0:
                        self.i += 2
                    else:
                        b = 0                          # marker: feed zeros, do not advance
                else:
                    self.i += 1
            self.acc = ((self.acc << 8) | b) & 0xFFFFFFFFFF
            self.n += 8

    def peek16(self) -> int:
        if self.n < 16:
            self._fill()
        return (self.acc >> (self.n - 16)) & 0xFFFF

    def skip(self, k: int):
        self.n -= k

    def get(self, k: int) -> int:
        if self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def restart(self, expect: int):
        self.n = 0
        self.acc = 0
        if self.i + 1 >= len(self.d) or self.d[self.i] != 0xFF or self.d[self.i + 1] != 0xD0 + expect:
            raise JpegError("restart marker expected")
        self.i += 2


def _build_lookup(counts: List[int], symbols: List[int]):
    """code -> (length, symbol) on 16-bit lookahead (T.81 annex C code assignment)."""
    look = np.zeros(65536, np.int32)                    # (len << 8) | symbol, 0 = invalid
    code, k = 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            lo = code << (16 - ln)
            look[lo:lo + (1 << (16 - ln))] = (ln << 8) | symbols[k]
            code += 1
            k += 1
        code <<= 1
    return look


def decode_coefficients(frame: Dict) -> List[np.ndarray]:
    """-> per component int16 array (block_rows, block_cols, 64) in natural order, quantised (jdhuff.c decode_mcu)."""
    comps = frame["comps"]
    hmax = max(c["h"] for c in comps)
    vmax = max(c["v"] for c in comps)
    mcus_x = -(-frame["width"] // (8 * hmax))
    mcus_y = -(-frame["height"] // (8 * vmax))
    out = [np.zeros((mcus_y * c["v"], mcus_x * c["h"], 64), np.int16) for c in comps]
    looks = {k: _build_lookup(*v) for k, v in frame["huff"].items()}
    for c in comps:
        if (0, c["td"]) not in looks or (1, c["ta"]) not in looks:
            raise JpegError("missing Huffman table")
        if c["tq"] not in frame["qt"]:
            raise JpegError("missing quantisation table")
    bits = _Bits(frame["scan"])
    pred = [0] * len(comps)
    ri, rcount, rnext = frame["restart"], 0, 0
    zz = [int(z) for z in ZIGZAG]
    for my in range(mcus_y):
        for mx in range(mcus_x):
            if ri and rcount == ri:
                bits.restart(rnext)
                rnext = (rnext + 1) & 7
                rcount = 0
                pred = [0] * len(comps)
            rcount += 1
            for ci, c in enumerate(comps):
                dc, ac = looks[(0, c["td"])], looks[(1, c["ta"])]
                for by in range(c["v"]):
                    for bx in range(c["h"]):
                        blk = out[ci][my * c["v"] + by, mx * c["h"] + bx]
                        e = int(dc[bits.peek16()])
                        if e == 0:
                            raise JpegError("bad Huffman code")
                        bits.skip(e >> 8)
                        s = e & 255
                        if s:
                            r = bits.get(s)
                            pred[ci] += r if r >= (1 << (s - 1)) else r - (1 << s) + 1
                        blk[0] = ((pred[ci] + 32768) & 0xFFFF) - 32768          # (JCOEF) cast
                        k = 1
                        while k < 64:
                            e = int(ac[bits.peek16()])
                            if e == 0:
                                raise JpegError("bad Huffman code")
                            bits.skip(e >> 8)
                            r, s = (e & 255) >> 4, e & 15
                            if s:
                                k += r
                                if k > 63:
                                    raise JpegError("coefficient index out of range")
                                v = bits.get(s)
                                blk[zz[k]] = v if v >= (1 << (s - 1)) else v - (1 << s) + 1
                                k += 1
                            elif r == 15:
                                k += 16
                            else:
                                break
    return out


_C = dict(F0_298=2446, F0_390=3196, F0_541=4433, F0_765=6270, F0_899=7373, F1_175=9633, F1_501=12299, F1_847=15137,
          F1_961=16069, F2_053=16819, F2_562=20995, F3_072=25172)


def _idct_1d(x, shift):
    """jidctint.c jpeg_idct_islow, one pass over the LAST axis of x (.., 8) int64 -> (.., 8) descaled by `shift`."""
    C = _C
    z2, z3 = x[..., 2], x[..., 6]
    z1 = (z2 + z3) * C["F0_541"]
    tmp2 = z1 - z3 * C["F1_847"]
    tmp3 = z1 + z2 * C["F0_765"]
    z2, z3 = x[..., 0], x[..., 4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = x[..., 7], x[..., 5], x[..., 3], x[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * C["F1_175"]
    tmp0 = tmp0 * C["F0_298"]
    tmp1 = tmp1 * C["F2_053"]
    tmp2 = tmp2 * C["F3_072"]
    tmp3 = tmp3 * C["F1_501"]
    z1 = -z1 * C["F0_899"]
    z2 = -z2 * C["F2_562"]
    z3 = -z3 * C["F1_961"] + z5
    z4 = -z4 * C["F0_390"] + z5
    tmp0 = tmp0 + z1 + z3
    tmp1 = tmp1 + z2 + z4
    tmp2 = tmp2 + z2 + z3
    tmp3 = tmp3 + z1 + z4
    half = 1 << (shift - 1)
    return np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0,
                     tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], axis=-1) + half >> shift


def idct_islow(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """(R, C, 64) int16 quantised blocks -> (R*8, C*8) uint8 sample plane."""
    R, Cn, _ = coef.shape
    x = (coef.astype(np.int64) * qt.astype(np.int64)).reshape(R, Cn, 8, 8)
    ws = _idct_1d(x.swapaxes(-1, -2), 11).swapaxes(-1, -2)          # pass 1: columns
    y = _idct_1d(ws, 18) & 1023                                     # pass 2: rows; RANGE_MASK
    px = np.where(y < 128, y + 128, np.where(y < 512, 255, np.where(y < 896, 0, y - 896)))
    return px.transpose(0, 2, 1, 3).reshape(R * 8, Cn * 8).astype(np.uint8)


def _upsample_h2v2(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample on a (h, w) plane already cropped to the component's downsampled size."""
    h, w = p.shape
    p = p.astype(np.int64)
    up = np.concatenate([p[:1], p[:-1]], 0)                         # row above (top replicated)
    dn = np.concatenate([p[1:], p[-1:]], 0)                         # row below (bottom replicated)
    out = np.zeros((2 * h, 2 * w), np.int64)
    for v, other in ((0, up), (1, dn)):
        cs = 3 * p + other                                          # column sums
        prev = np.concatenate([cs[:, :1], cs[:, :-1]], 1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], 1)
        out[v::2, 0::2] = (3 * cs + prev + 8) >> 4
        out[v::2, 1::2] = (3 * cs + nxt + 7) >> 4
    return out


def _upsample_h2v1(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v1_fancy_upsample."""
    h, w = p.shape
    p = p.astype(np.int64)
    prev = np.concatenate([p[:, :1], p[:, :-1]], 1)
    nxt = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out = np.zeros((h, 2 * w), np.int64)
    out[:, 0::2] = (3 * p + prev + 1) >> 2
    out[:, 1::2] = (3 * p + nxt + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def _fix(x):
    return int(x * 65536 + 0.5)


def ycc_to_bgr(y, cb, cr) -> np.ndarray:
    """jdcolor.c ycc_rgb_convert with build_ycc_rgb_table's fixed-point tables."""
    y = y.astype(np.int64)
    xb, xr = cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((_fix(1.40200) * xr + 32768) >> 16)
    b = y + ((_fix(1.77200) * xb + 32768) >> 16)
    g = y + ((-_fix(0.34414) * xb + 32768 - _fix(0.71414) * xr) >> 16)
    return np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)


def decode_planes(frame: Dict) -> List[np.ndarray]:
    """Full-size (padded to whole blocks) uint8 sample planes per component."""
    coefs = decode_coefficients(frame)
    return [idct_islow(cf, frame["qt"][c["tq"]]) for cf, c in zip(coefs, frame["comps"])]


def decode_bgr(data: bytes) -> np.ndarray:
    """JPEG bytes -> (H, W, 3) uint8 BGR, the array cv2.imread(path, cv2.IMREAD_COLOR) returns."""
    frame = parse(data)
    H, W = frame["height"], frame["width"]
    planes = decode_planes(frame)
    comps = frame["comps"]
    if len(comps) == 1:
        yy = planes[0][:H, :W]
        return np.stack([yy, yy, yy], -1)
    hs, vs = comps[0]["h"], comps[0]["v"]
    ch, cw = -(-H // vs), -(-W // hs)                                # downsampled_height / downsampled_width
    chroma = []
    for p in planes[1:]:
        p = p[:ch, :cw]
        if (hs, vs) == (2, 2):
            # libjpeg-turbo falls back to plain replication when the chroma plane is only 1-2 columns wide
            p = _upsample_h2v2(p) if cw > 2 else np.repeat(np.repeat(p, 2, 0), 2, 1)
        elif (hs, vs) == (2, 1):
            p = _upsample_h2v1(p) if cw > 2 else np.repeat(p, 2, 1)
        chroma.append(p[:H, :W])
    return ycc_to_bgr(planes[0][:H, :W], chroma[0], chroma[1])
