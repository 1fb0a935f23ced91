"""Native PNG decode (csrc/ira_png.cpp behind ira_png_info / ira_png_decode, SURVEY 8 row f4: the cv2.imread / cv2.imdecode step of
AgenticIR/executor/.../Restormer/demo.py:40-44 and AgentApp model_serving.py:126-134).  Integer work: byte for byte against PIL (libpng)
on streams written by PIL with its adaptive row filters (all five filter types occur), by tools.encode_png (filter 0, striped deflate) and
hand-built ones (several IDAT chunks, every filter type forced).  Host-only code: runs without a GPU."""
import io
import os
import struct
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'image-restoration-agent_amd'))

from PIL import Image  # noqa: E402

from ira_amd._capi import IraError, decode_png  # noqa: E402
from ira_amd.tools import decode_image, encode_png  # noqa: E402


def _photo(h, w, c, seed):
    """smooth content + noise: PIL's filter heuristic then picks Sub / Up / Average / Paeth on different rows"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    base = [127 + 100 * np.sin(x / (7 + 3 * k) + k) * np.cos(y / (5 + 2 * k)) for k in range(c)]
    a = np.stack(base, -1) + rng.normal(0, 6, (h, w, c)) * (x[..., None] > w / 2)
    return np.clip(a, 0, 255).astype(np.uint8)


def _pil_png(arr, mode, **kw):
    buf = io.BytesIO()
    Image.fromarray(arr if arr.ndim == 3 and arr.shape[2] > 1 else arr.reshape(arr.shape[:2]), mode).save(buf, format='PNG', **kw)
    return buf.getvalue()


def _chunk(tag, data):
    return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xFFFFFFFF)


def _filter_rows(a, ftype):
    """PNG-filter an (h, w, c) raster with one filter type on every row (RFC 2083 section 6), in plain Python / numpy."""
    h, w, c = a.shape
    rows = a.reshape(h, w * c).astype(np.int32)
    out = np.zeros((h, 1 + w * c), np.uint8)
    for y in range(h):
        cur, up = rows[y], (rows[y - 1] if y else np.zeros(w * c, np.int32))
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        ul = np.concatenate([np.zeros(c, np.int32), up[:-c]])
        if ftype == 0: pred = 0
        elif ftype == 1: pred = left
        elif ftype == 2: pred = up
        elif ftype == 3: pred = (left + up) >> 1
        else:
            p = left + up - ul
            pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
        out[y, 0] = ftype
        out[y, 1:] = (cur - pred) & 0xFF
    return out.tobytes()


@pytest.mark.parametrize('h,w', [(1, 1), (5, 7), (33, 17), (240, 321)])
@pytest.mark.parametrize('mode,c', [('L', 1), ('LA', 2), ('RGB', 3), ('RGBA', 4)])
def test_decode_matches_pil(h, w, mode, c):
    a = _photo(h, w, c, h * 31 + w + c)
    png = _pil_png(a, mode)
    got = decode_png(png)
    with Image.open(io.BytesIO(png)) as im:
        ref = np.asarray(im).reshape(h, w, c)
    assert got.shape == (h, w, c) and got.dtype == np.uint8
    assert np.array_equal(got, ref) and np.array_equal(got, a)


def test_all_filter_types_occur_in_pil_output_and_forced():
    a = _photo(64, 96, 3, 9)
    raw = zlib.decompress(b''.join(_idat(_pil_png(a, 'RGB'))))
    used = {raw[y * (1 + 96 * 3)] for y in range(64)}
    assert len(used) >= 3, used                      # the PIL stream of test_decode_matches_pil really exercises several filters
    ihdr = struct.pack('>IIBBBBB', 96, 64, 8, 2, 0, 0, 0)
    for ft in range(5):
        z = zlib.compress(_filter_rows(a, ft), 6)
        third = len(z) // 3                          # three IDAT chunks, one of them cutting the stream at an arbitrary byte
        png = b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', ihdr) + _chunk(b'tEXt', b'k\0v') + _chunk(b'IDAT', z[:third]) + _chunk(b'IDAT', z[third:2 * third + 1]) \
            + _chunk(b'IDAT', z[2 * third + 1:]) + _chunk(b'IEND', b'')
        assert np.array_equal(decode_png(png), a), ft
        with Image.open(io.BytesIO(png)) as im:      # the hand-built stream is a valid PNG for libpng too
            assert np.array_equal(np.asarray(im), a)


def _idat(png):
    off, out = 8, []
    while off < len(png):
        n = struct.unpack('>I', png[off:off + 4])[0]
        if png[off + 4:off + 8] == b'IDAT':
            out.append(png[off + 8:off + 8 + n])
        off += 12 + n
    return out


def test_palette_and_own_encoder():
    a = _photo(40, 50, 3, 3)
    pal = Image.fromarray(a, 'RGB').quantize(colors=37)
    buf = io.BytesIO(); pal.save(buf, format='PNG')
    got = decode_png(buf.getvalue())
    assert np.array_equal(got, np.asarray(pal.convert('RGB')))
    big = _photo(700, 1100, 3, 4)                    # 2.3 MB of rows: three deflate stripes joined by sync flushes (tools._deflate_parallel)
    assert np.array_equal(decode_png(encode_png(big)), big)
    g = _photo(90, 70, 1, 5)
    assert np.array_equal(decode_png(encode_png(g)), g)


def test_decode_image_semantics():
    """what the harness gets: RGB (h,w,3) with alpha dropped and grey replicated, or (h,w,1) with PIL's 'L' weights (tools.decode_image)"""
    for mode, c in (('L', 1), ('LA', 2), ('RGB', 3), ('RGBA', 4)):
        png = _pil_png(_photo(21, 34, c, c), mode)
        with Image.open(io.BytesIO(png)) as im:
            ref_rgb, ref_l = np.asarray(im.convert('RGB')), np.asarray(im.convert('L'))[:, :, None]
        assert np.array_equal(decode_image(png), ref_rgb), mode
        assert np.array_equal(decode_image(png, gray=True), ref_l), mode
    a = _photo(16, 16, 3, 1)
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format='BMP')
    assert np.array_equal(decode_image(buf.getvalue()), a)              # not a PNG: PIL


def test_declined_and_broken_streams():
    a = _photo(12, 9, 3, 2)
    good = _pil_png(a, 'RGB')
    with pytest.raises(IraError, match='signature'):
        decode_png(b'not a png at all, but long enough to look at ........')
    with pytest.raises(IraError, match='CRC'):
        bad = bytearray(good); bad[len(good) // 2] ^= 0x40
        decode_png(bytes(bad))
    with pytest.raises(IraError, match='IEND|past the end'):
        decode_png(good[:-20])
    # 16-bit samples and Adam7 are left to PIL: decode_png says why, decode_image still returns PIL's pixels
    a16 = (np.arange(12 * 9, dtype=np.uint16) * 500).reshape(12, 9)
    buf = io.BytesIO(); Image.fromarray(a16).save(buf, format='PNG')
    with pytest.raises(IraError, match='16 bits'):
        decode_png(buf.getvalue())
    with Image.open(io.BytesIO(buf.getvalue())) as im:
        assert np.array_equal(decode_image(buf.getvalue()), np.asarray(im.convert('RGB')))
    ihdr = struct.pack('>IIBBBBB', 9, 12, 8, 2, 0, 0, 1)
    inter = b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', ihdr) + _chunk(b'IDAT', zlib.compress(b'\0' * 400)) + _chunk(b'IEND', b'')
    with pytest.raises(IraError, match='Adam7'):
        decode_png(inter)
    # a raster that inflates to the wrong size
    ihdr = struct.pack('>IIBBBBB', 9, 12, 8, 2, 0, 0, 0)
    short = b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', ihdr) + _chunk(b'IDAT', zlib.compress(b'\0' * 100)) + _chunk(b'IEND', b'')
    with pytest.raises(IraError, match='inflates to|inflate failed'):
        decode_png(short)
    long_ = b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', ihdr) + _chunk(b'IDAT', zlib.compress(b'\0' * 4000)) + _chunk(b'IEND', b'')
    with pytest.raises(IraError, match='inflates to|inflate failed'):
        decode_png(long_)
    huge = b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', struct.pack('>IIBBBBB', 1 << 19, 1 << 19, 8, 6, 0, 0, 0)) + _chunk(b'IDAT', zlib.compress(b'\0')) + _chunk(b'IEND', b'')
    with pytest.raises(IraError, match='raster limit'):
        decode_png(huge)
    badf = b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', ihdr) + _chunk(b'IDAT', zlib.compress(b'\7' * (12 * 28))) + _chunk(b'IEND', b'')
    with pytest.raises(IraError, match='filter type 7'):
        decode_png(badf)


def test_mutated_streams_never_crash():
    """a parser of untrusted bytes: 400 random mutations (byte flips with the CRC repaired so that the damage reaches the inflater and the
    filters, truncations, length-field edits) of valid streams either decode to a raster of the announced size or raise IraError"""
    rng = np.random.default_rng(7)
    seeds = [_pil_png(_photo(19, 23, c, c), m) for m, c in (('L', 1), ('LA', 2), ('RGB', 3), ('RGBA', 4))]
    pal = io.BytesIO(); Image.fromarray(_photo(19, 23, 3, 9), 'RGB').quantize(colors=11).save(pal, format='PNG'); seeds.append(pal.getvalue())

    def refix(b):
        out, off = bytearray(b[:8]), 8
        while off + 12 <= len(b):
            n = struct.unpack('>I', b[off:off + 4])[0]
            if off + 12 + n > len(b):
                out += b[off:]; return bytes(out)
            tag, data = b[off + 4:off + 8], b[off + 8:off + 8 + n]
            out += _chunk(bytes(tag), bytes(data)); off += 12 + n
        return bytes(out + b[off:])
    ok = bad = 0
    for i in range(400):
        b = bytearray(seeds[i % len(seeds)])
        kind = i % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(8, len(b)))] ^= int(rng.integers(1, 256))
            data = refix(bytes(b))
        elif kind == 1:
            data = bytes(b[:int(rng.integers(0, len(b)))])
        elif kind == 2:
            pos = int(rng.integers(8, len(b) - 4)); b[pos:pos + 4] = struct.pack('>I', int(rng.integers(0, 1 << 32)))
            data = bytes(b)
        else:
            b[int(rng.integers(8, len(b)))] ^= int(rng.integers(1, 256))
            data = bytes(b)
        try:
            a = decode_png(data)
            assert a.ndim == 3 and a.dtype == np.uint8
            ok += 1
        except IraError:
            bad += 1
    assert ok + bad == 400 and bad > 100
