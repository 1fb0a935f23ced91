// PNG decode on the host: the step in front of the hot path (cv2.imread in AgenticIR/executor/*/tools/Restormer/demo.py:40-44 and
// cv2.imdecode in AgentApp/.../model_serving.py:126-134).  Chunk walk + CRC check, one zlib inflate of the concatenated
// IDAT payloads (libz is the system's, as in libpng), then the five row filters of the PNG specification (RFC 2083 section 6) undone in
// place.  Rasters the agent's tools produce and consume: 8 bits per sample, grey / grey+alpha / RGB / RGBA / palette, not interlaced;
// anything else is reported as IRA_ERR_ARG with a message and the Python wrapper falls back to PIL for it.  Integer work: the test compares
// with PIL byte for byte (tests/test_png.py).
#include "ira_model.h"

#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <utility>
#include <algorithm>
#include <cstdio>
#include <new>
#include <vector>

using ira_rt::gfail;

namespace {

struct PngInfo {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    int src_ch = 0;      // samples per pixel in the filtered rows
    int out_ch = 0;      // channels of the decoded raster (palette -> 3)
};

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

// Walks the chunks.  Fills info from IHDR; when idat != nullptr also gathers the IDAT payloads and the palette.  Returns 0 or an error code.
typedef std::vector<std::pair<const uint8_t*, size_t>> Pieces;

int walk(const uint8_t* png, size_t len, PngInfo& info, Pieces* idat, uint8_t (*plte)[3], int* n_plte, bool check_crc) {
    if (!png || len < 8 + 25 || memcmp(png, kSig, 8) != 0) return gfail(IRA_ERR_ARG, "png: not a PNG stream (signature)");
    size_t off = 8;
    bool have_ihdr = false, have_iend = false;
    while (off + 12 <= len) {
        const uint32_t n = be32(png + off);
        const uint8_t* tag = png + off + 4;
        if ((size_t)n > len - off - 12) return gfail(IRA_ERR_ARG, "png: chunk '%.4s' runs past the end of the stream", (const char*)tag);
        const uint8_t* data = tag + 4;
        if (check_crc) {
            const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), tag, n + 4);
            if (crc != be32(data + n)) return gfail(IRA_ERR_ARG, "png: CRC mismatch in chunk '%.4s'", (const char*)tag);
        }
        if (!have_ihdr) {
            if (memcmp(tag, "IHDR", 4) != 0 || n != 13) return gfail(IRA_ERR_ARG, "png: first chunk is not IHDR");
            info.w = be32(data); info.h = be32(data + 4);
            info.depth = data[8]; info.ctype = data[9]; info.interlace = data[12];
            // a header is untrusted input: the caller allocates h * w * channels bytes on its word (2^31 bytes = a 23k x 23k RGBA frame)
            if (info.w == 0 || info.h == 0 || info.w > (1u << 20) || info.h > (1u << 20) || (uint64_t)info.w * info.h * 4 > (1ull << 31))
                return gfail(IRA_ERR_ARG, "png: size %u x %u is zero or beyond the 2 GiB raster limit", info.w, info.h);
            if (data[10] != 0 || data[11] != 0) return gfail(IRA_ERR_ARG, "png: unknown compression / filter method");
            if (info.depth != 8) return gfail(IRA_ERR_ARG, "png: %d bits per sample (only 8 is decoded natively)", info.depth);
            if (info.interlace != 0) return gfail(IRA_ERR_ARG, "png: Adam7 interlacing is not decoded natively");
            switch (info.ctype) {
                case 0: info.src_ch = 1; info.out_ch = 1; break;
                case 2: info.src_ch = 3; info.out_ch = 3; break;
                case 3: info.src_ch = 1; info.out_ch = 3; break;
                case 4: info.src_ch = 2; info.out_ch = 2; break;
                case 6: info.src_ch = 4; info.out_ch = 4; break;
                default: return gfail(IRA_ERR_ARG, "png: colour type %d", info.ctype);
            }
            have_ihdr = true;
            if (!idat) return IRA_OK;                        // header only
        } else if (memcmp(tag, "IDAT", 4) == 0) {
            if (n) idat->emplace_back(data, (size_t)n);
        } else if (memcmp(tag, "PLTE", 4) == 0) {
            if (n % 3 != 0 || n > 768) return gfail(IRA_ERR_ARG, "png: bad PLTE length %u", n);
            *n_plte = (int)(n / 3);
            memcpy(plte, data, n);
        } else if (memcmp(tag, "IEND", 4) == 0) {
            have_iend = true;
            break;
        }
        off += 12 + (size_t)n;
    }
    if (!have_ihdr) return gfail(IRA_ERR_ARG, "png: no IHDR");
    if (!have_iend) return gfail(IRA_ERR_ARG, "png: truncated (no IEND)");
    return IRA_OK;
}

inline uint8_t paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

// Undo the filter of one row: cur = filtered bytes (in / out), prev = the reconstructed row above (nullptr for the first row).
template <int BPP>
int unfilter_row(int ft, uint8_t* cur, const uint8_t* prev, size_t n) {
    switch (ft) {
        case 0: return 0;
        case 1:
            for (size_t i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - BPP]);
            return 0;
        case 2:
            if (prev) for (size_t i = 0; i < n; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
            return 0;
        case 3:
            for (size_t i = 0; i < n; ++i) {
                const int a = i >= BPP ? cur[i - BPP] : 0, b = prev ? prev[i] : 0;
                cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
            }
            return 0;
        case 4:
            for (size_t i = 0; i < n; ++i) {
                const int a = i >= BPP ? cur[i - BPP] : 0, b = prev ? prev[i] : 0, c = (prev && i >= BPP) ? prev[i - BPP] : 0;
                cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
            }
            return 0;
        default: return 1;
    }
}

}  // namespace

extern "C" int ira_png_info(const uint8_t* png, size_t len, int* h, int* w, int* channels) {
    if (!h || !w || !channels) return gfail(IRA_ERR_ARG, "ira_png_info: NULL output");
    PngInfo info;
    const int rc = walk(png, len, info, nullptr, nullptr, nullptr, false);
    if (rc != IRA_OK) return rc;
    *h = (int)info.h; *w = (int)info.w; *channels = info.out_ch;
    return IRA_OK;
}

extern "C" int ira_png_decode(const uint8_t* png, size_t len, uint8_t* out, size_t out_bytes) {
    if (!out) return gfail(IRA_ERR_ARG, "ira_png_decode: out is NULL");
    PngInfo info;
    Pieces idat;                                             // the IDAT payloads where they lie in the stream
    uint8_t plte[256][3];
    int n_plte = 0;
    int rc = walk(png, len, info, &idat, plte, &n_plte, true);
    if (rc != IRA_OK) return rc;
    const size_t row = (size_t)info.w * info.src_ch, need = (size_t)info.h * info.w * info.out_ch;
    if (out_bytes < need) return gfail(IRA_ERR_ARG, "ira_png_decode: output buffer holds %zu bytes, the raster needs %zu", out_bytes, need);
    if (info.ctype == 3 && n_plte == 0) return gfail(IRA_ERR_ARG, "png: palette image without PLTE");
    if (idat.empty()) return gfail(IRA_ERR_ARG, "png: no IDAT");

    // one inflate over the IDAT payloads, in stream order, into [h][1 + row]
    const size_t raw_bytes = (size_t)info.h * (row + 1);
    std::unique_ptr<uint8_t[]> raw_buf(new (std::nothrow) uint8_t[raw_bytes]);
    if (!raw_buf) return gfail(IRA_ERR_OOM, "png: cannot allocate %zu bytes for the filtered rows", raw_bytes);
    uint8_t* raw = raw_buf.get();
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) return gfail(IRA_ERR_OOM, "png: inflateInit failed");
    size_t out_off = 0;
    int zrc = Z_OK;
    for (size_t k = 0; k < idat.size() && zrc != Z_STREAM_END; ++k) {
        size_t in_off = 0;
        while (in_off < idat[k].second && zrc != Z_STREAM_END) {   // avail_in / avail_out are 32-bit: pieces of at most 1 GiB
            const size_t in_n = std::min<size_t>(idat[k].second - in_off, (size_t)1 << 30), out_n = std::min<size_t>(raw_bytes - out_off, (size_t)1 << 30);
            zs.next_in = const_cast<uint8_t*>(idat[k].first) + in_off; zs.avail_in = (uInt)in_n;
            zs.next_out = raw + out_off; zs.avail_out = (uInt)out_n;
            zrc = inflate(&zs, Z_NO_FLUSH);
            const size_t used = in_n - zs.avail_in, made = out_n - zs.avail_out;
            in_off += used; out_off += made;
            if (zrc == Z_STREAM_END) break;
            if (zrc != Z_OK || (used == 0 && made == 0)) {      // an error, or no progress: output full with input left
                char why[160];
                snprintf(why, sizeof why, "%s", zs.msg ? zs.msg : (out_off == raw_bytes ? "more data than the raster holds" : "corrupt IDAT"));
                inflateEnd(&zs);
                return gfail(IRA_ERR_ARG, "png: inflate failed (%s)", why);
            }
        }
    }
    inflateEnd(&zs);
    if (zrc != Z_STREAM_END || out_off != raw_bytes)
        return gfail(IRA_ERR_ARG, "png: IDAT inflates to %zu bytes%s, the raster needs %zu", out_off, zrc == Z_STREAM_END ? "" : " (stream not finished)", raw_bytes);

    // unfilter row by row; grey / RGB / RGBA rows land in `out` directly, palette rows are expanded
    const uint8_t* prev = nullptr;
    for (uint32_t y = 0; y < info.h; ++y) {
        uint8_t* cur = raw + (size_t)y * (row + 1);
        const int ft = cur[0];
        ++cur;
        int bad;
        switch (info.src_ch) {
            case 1: bad = unfilter_row<1>(ft, cur, prev, row); break;
            case 2: bad = unfilter_row<2>(ft, cur, prev, row); break;
            case 3: bad = unfilter_row<3>(ft, cur, prev, row); break;
            default: bad = unfilter_row<4>(ft, cur, prev, row); break;
        }
        if (bad) return gfail(IRA_ERR_ARG, "png: filter type %d in row %u", ft, y);
        prev = cur;
        uint8_t* dst = out + (size_t)y * info.w * info.out_ch;
        if (info.ctype != 3) {
            memcpy(dst, cur, row);
        } else {
            for (uint32_t x = 0; x < info.w; ++x) {
                const int i = cur[x];
                if (i >= n_plte) return gfail(IRA_ERR_ARG, "png: palette index %d of %d in row %u", i, n_plte, y);
                dst[3 * x] = plte[i][0]; dst[3 * x + 1] = plte[i][1]; dst[3 * x + 2] = plte[i][2];
            }
        }
    }
    return IRA_OK;
}
