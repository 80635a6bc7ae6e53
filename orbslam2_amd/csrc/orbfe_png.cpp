// orbfe_png.cpp -- input side of the drivers: cv::imread(path, cv::IMREAD_UNCHANGED) for PNG files, the call
// every replay driver of the reference makes per frame (Test/Replay/Stereo/stereo_kitti.cc:69-70, stereo_euroc.cc:119-120,
// RGBD/rgbd_tum.cc:80-81: 8-bit grey / 8-bit RGB images and 16-bit depth maps).
//
// Host code by design: a PNG is ONE DEFLATE stream (bit-serial Huffman decoding) followed by scanline filters whose bytes depend
// on their left / upper neighbours -- there is no data parallelism inside an image for a GPU to use.  What the MI355X path needs
// from this stage is that it does not stall the device: images decode on the host cores, many at a time
// (orbfe_png_decode_batch: one image per thread), straight into caller memory that can be the pinned staging block of the
// upload, while the GPU works on the previous batch.  DEFLATE itself is zlib's inflate (system library); chunk parsing, CRC
// checks, the five scanline filters, Adam7 de-interlacing, bit-depth / palette / alpha expansion and the channel order of
// cv::imread are written here against the PNG specification (ISO/IEC 15948) -- OpenCV and libpng headers are absent.
//
// Output convention = cv::imread(..., IMREAD_UNCHANGED) of OpenCV 4.x's PngDecoder (OPENCV-4.5.5-SEMANTICS, grfmt_png.cpp):
//   grey (1, 2, 4, 8 bit) -> 1 channel 8 bit (low depths scaled to 0..255);  grey 16 -> 1 channel 16 bit, host byte order;
//   RGB -> 3 channels in B, G, R order;  RGBA / grey+alpha -> 4 channels B, G, R, A (grey replicated);
//   palette -> B, G, R (B, G, R, A when a tRNS chunk is present);  RGB + tRNS -> B, G, R, A (alpha 0 for the key colour);
//   16-bit samples stay 16 bit, host byte order; rows tightly packed unless the caller gives a stride.
#include "../../include/orbfe.h"

#include <zlib.h>

#include <atomic>
#include <exception>
#include <new>
#include <system_error>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace
{

struct PngHeader {
    uint32_t w = 0, h = 0;
    int depth = 0, color = 0, interlace = 0;
    int src_channels = 0;  // samples per pixel in the file
    int out_channels = 0;  // channels cv::imread returns
    int out_depth = 0;     // 8 or 16
    bool has_trns = false;
};

thread_local char t_err[256] = "";
int png_fail(const char *msg)
{
    snprintf(t_err, sizeof(t_err), "%s", msg);
    return ORBFE_ERR_INVALID;
}

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

const uint8_t k_sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};

struct Chunks {
    PngHeader hd;
    std::vector<uint8_t> idat;      // concatenated IDAT payloads (the zlib stream)
    uint8_t palette[256][3];
    int n_palette = 0;
    uint8_t trns_alpha[256];        // palette alpha
    int n_trns = 0;
    uint16_t trns_key[3] = {0, 0, 0}; // colour key of grey / RGB images
};

// Walks the chunk list: IHDR first, PLTE / tRNS / IDAT..., IEND; every chunk's CRC-32 is checked (libpng rejects a critical
// chunk with a bad CRC, and cv::imread then returns an empty Mat).
int parse(const uint8_t *file, size_t size, Chunks &c, bool want_data)
{
    if (!file || size < 8 + 25 || memcmp(file, k_sig, 8) != 0) return png_fail("not a PNG file");
    size_t p = 8;
    bool seen_ihdr = false, seen_iend = false, seen_idat = false;
    memset(c.trns_alpha, 255, sizeof(c.trns_alpha));
    while (p + 12 <= size && !seen_iend) {
        const uint32_t len = be32(file + p);
        const uint8_t *type = file + p + 4, *data = file + p + 8;
        if ((size_t)len > size - p - 12) return png_fail("truncated chunk");
        const uint32_t crc = be32(data + len);
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != crc) return png_fail("chunk CRC mismatch");
        if (!seen_ihdr) {
            if (memcmp(type, "IHDR", 4) != 0 || len != 13) return png_fail("IHDR missing");
            PngHeader &h = c.hd;
            h.w = be32(data); h.h = be32(data + 4);
            h.depth = data[8]; h.color = data[9]; h.interlace = data[12];
            if (h.w == 0 || h.h == 0 || h.w > 65535u || h.h > 65535u) return png_fail("unsupported image size");
            if (data[10] != 0 || data[11] != 0 || h.interlace > 1) return png_fail("unsupported compression / filter / interlace method");
            const int d = h.depth;
            bool ok = false;
            switch (h.color) {
            case 0: ok = d == 1 || d == 2 || d == 4 || d == 8 || d == 16; h.src_channels = 1; break;
            case 2: ok = d == 8 || d == 16; h.src_channels = 3; break;
            case 3: ok = d == 1 || d == 2 || d == 4 || d == 8; h.src_channels = 1; break;
            case 4: ok = d == 8 || d == 16; h.src_channels = 2; break;
            case 6: ok = d == 8 || d == 16; h.src_channels = 4; break;
            default: break;
            }
            if (!ok) return png_fail("invalid colour type / bit depth");
            seen_ihdr = true;
        } else if (memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 || len > 768) return png_fail("bad PLTE");
            c.n_palette = (int)(len / 3);
            memcpy(c.palette, data, len);
        } else if (memcmp(type, "tRNS", 4) == 0) {
            if (c.hd.color == 3) { c.n_trns = (int)(len > 256 ? 256 : len); memcpy(c.trns_alpha, data, (size_t)c.n_trns); c.hd.has_trns = c.n_trns > 0; } // libpng: num_trans == 0 -> no alpha
            else if (c.hd.color == 2 && len == 6) { for (int k = 0; k < 3; k++) c.trns_key[k] = (uint16_t)((data[2 * k] << 8) | data[2 * k + 1]); c.hd.has_trns = true; }
            // grey + tRNS: cv::imread keeps one channel (readHeader only looks at tRNS for RGB / palette images)
        } else if (memcmp(type, "IDAT", 4) == 0) {
            seen_idat = true;
            if (want_data) c.idat.insert(c.idat.end(), data, data + len);
        } else if (memcmp(type, "IEND", 4) == 0) {
            seen_iend = true;
        } else if (!(type[0] & 0x20)) {
            return png_fail("unknown critical chunk");
        }
        p += 12 + (size_t)len;
    }
    if (!seen_ihdr || !seen_idat || !seen_iend) return png_fail("missing IDAT / IEND");
    PngHeader &h = c.hd;
    if (h.color == 3 && c.n_palette == 0) return png_fail("palette image without PLTE");
    h.out_depth = h.depth == 16 ? 16 : 8;
    switch (h.color) {
    case 0: h.out_channels = 1; break;
    case 2: h.out_channels = h.has_trns ? 4 : 3; break;
    case 3: h.out_channels = h.has_trns ? 4 : 3; break;
    default: h.out_channels = 4; break; // grey + alpha, RGBA
    }
    return ORBFE_OK;
}

// Reverses the scanline filter of one row in place (cur has the filter-type byte at cur[-1]; prev = reconstructed previous row
// of the same pass, or null for the first row); bpp = bytes per complete pixel, at least 1.
int unfilter_row(uint8_t *cur, const uint8_t *prev, size_t n, int bpp)
{
    const int ft = cur[-1];
    switch (ft) {
    case 0: break;
    case 1:
        for (size_t i = (size_t)bpp; i < n; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
        break;
    case 2:
        if (prev) for (size_t i = 0; i < n; i++) cur[i] = (uint8_t)(cur[i] + prev[i]);
        break;
    case 3:
        for (size_t i = 0; i < n; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
            cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
        }
        break;
    case 4:
        if (!prev) { // first row: b = c = 0, the predictor is a
            for (size_t i = (size_t)bpp; i < n; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
            break;
        }
        for (size_t i = 0; i < (size_t)bpp && i < n; i++) cur[i] = (uint8_t)(cur[i] + prev[i]); // a = c = 0: the predictor is b
        for (size_t i = (size_t)bpp; i < n; i++) { // the same decision as paeth(), with the shared differences formed once
            int a = cur[i - bpp];
            const int b = prev[i], c = prev[i - bpp];
            const int p = b - c, q = a - c;
            int pa = p < 0 ? -p : p;
            const int pb = q < 0 ? -q : q, pc = (p + q) < 0 ? -(p + q) : (p + q);
            if (pb < pa) { pa = pb; a = b; }
            if (pc < pa) a = c;
            cur[i] = (uint8_t)(cur[i] + a);
        }
        break;
    default: return png_fail("invalid filter type");
    }
    return ORBFE_OK;
}

// Sample s (0 .. samples - 1) of an unfiltered row, as an integer of `depth` bits
inline unsigned sample_at(const uint8_t *row, int depth, size_t s)
{
    switch (depth) {
    case 8: return row[s];
    case 16: return ((unsigned)row[2 * s] << 8) | row[2 * s + 1];
    case 4: return (row[s >> 1] >> (4 * (1 - (s & 1)))) & 15u;
    case 2: return (row[s >> 2] >> (2 * (3 - (s & 3)))) & 3u;
    default: return (row[s >> 3] >> (7 - (s & 7))) & 1u;
    }
}

// Writes pixel x of the output row from pixel px of an unfiltered source row (imread's channel order and depth)
inline void put_pixel(const Chunks &c, const uint8_t *src, size_t px, uint8_t *dst_row, size_t x)
{
    const PngHeader &h = c.hd;
    const int oc = h.out_channels;
    if (h.out_depth == 16) {
        uint16_t *d = (uint16_t *)dst_row + x * oc;
        switch (h.color) {
        case 0: d[0] = (uint16_t)sample_at(src, 16, px); break;
        case 2: {
            const unsigned r = sample_at(src, 16, 3 * px), g = sample_at(src, 16, 3 * px + 1), b = sample_at(src, 16, 3 * px + 2);
            d[0] = (uint16_t)b; d[1] = (uint16_t)g; d[2] = (uint16_t)r;
            if (oc == 4) d[3] = (r == c.trns_key[0] && g == c.trns_key[1] && b == c.trns_key[2]) ? 0 : 65535;
            break;
        }
        case 4: { const unsigned g = sample_at(src, 16, 2 * px); d[0] = d[1] = d[2] = (uint16_t)g; d[3] = (uint16_t)sample_at(src, 16, 2 * px + 1); break; }
        default: d[0] = (uint16_t)sample_at(src, 16, 4 * px + 2); d[1] = (uint16_t)sample_at(src, 16, 4 * px + 1); d[2] = (uint16_t)sample_at(src, 16, 4 * px);
                 d[3] = (uint16_t)sample_at(src, 16, 4 * px + 3); break;
        }
        return;
    }
    uint8_t *d = dst_row + x * oc;
    switch (h.color) {
    case 0: {
        const unsigned v = sample_at(src, h.depth, px);
        d[0] = (uint8_t)(h.depth == 8 ? v : h.depth == 4 ? v * 17u : h.depth == 2 ? v * 85u : v * 255u); // png_set_expand_gray_1_2_4_to_8
        break;
    }
    case 2: {
        const unsigned r = src[3 * px], g = src[3 * px + 1], b = src[3 * px + 2];
        d[0] = (uint8_t)b; d[1] = (uint8_t)g; d[2] = (uint8_t)r;
        if (oc == 4) d[3] = (r == c.trns_key[0] && g == c.trns_key[1] && b == c.trns_key[2]) ? 0 : 255;
        break;
    }
    case 3: {
        unsigned i = sample_at(src, h.depth, px);
        if ((int)i >= c.n_palette) i = 0; // libpng leaves out-of-range indices to the application; keep the output defined
        d[0] = c.palette[i][2]; d[1] = c.palette[i][1]; d[2] = c.palette[i][0];
        if (oc == 4) d[3] = c.trns_alpha[i];
        break;
    }
    case 4: d[0] = d[1] = d[2] = src[2 * px]; d[3] = src[2 * px + 1]; break;
    default: d[0] = src[4 * px + 2]; d[1] = src[4 * px + 1]; d[2] = src[4 * px]; d[3] = src[4 * px + 3]; break;
    }
}

int decode(const uint8_t *file, size_t size, uint8_t *dst, size_t dst_stride, PngHeader *out_hd)
{
    Chunks c;
    int rc = parse(file, size, c, true);
    if (rc != ORBFE_OK) return rc;
    const PngHeader &h = c.hd;
    if (out_hd) *out_hd = h;
    const size_t out_row = (size_t)h.w * h.out_channels * (h.out_depth / 8);
    if (dst_stride == 0) dst_stride = out_row;
    if (dst_stride < out_row) return png_fail("dst_stride smaller than a decoded row");
    const int bits_pp = h.depth * h.src_channels;
    const int bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    // pass geometry: one pass for a progressive image, seven for Adam7
    static const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    const int n_pass = h.interlace ? 7 : 1;
    size_t total = 0;
    size_t pass_w[7], pass_h[7], pass_rowbytes[7];
    for (int p = 0; p < n_pass; p++) {
        pass_w[p] = h.interlace ? ((size_t)h.w + dx[p] - 1 - x0[p]) / dx[p] : h.w;
        pass_h[p] = h.interlace ? ((size_t)h.h + dy[p] - 1 - y0[p]) / dy[p] : h.h;
        if (h.interlace && ((size_t)h.w <= (size_t)x0[p] || (size_t)h.h <= (size_t)y0[p])) pass_w[p] = pass_h[p] = 0;
        pass_rowbytes[p] = (pass_w[p] * bits_pp + 7) / 8;
        if (pass_w[p] && pass_h[p]) total += (pass_rowbytes[p] + 1) * pass_h[p];
    }
    std::vector<uint8_t> raw(total + 8);
    {   // one zlib stream over all IDAT chunks; it must deliver exactly the filtered scanlines
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) return png_fail("inflateInit failed");
        zs.next_in = c.idat.data(); zs.avail_in = (uInt)c.idat.size();
        zs.next_out = raw.data(); zs.avail_out = (uInt)total;
        const int zr = inflate(&zs, Z_FINISH);
        const size_t got = total - zs.avail_out;
        inflateEnd(&zs);
        if (!((zr == Z_STREAM_END || zr == Z_BUF_ERROR || zr == Z_OK) && got == total)) return png_fail("IDAT stream does not decode to the image size");
        if (zr != Z_STREAM_END && zr != Z_BUF_ERROR && zr != Z_OK) return png_fail("corrupt IDAT stream");
    }
    size_t off = 0;
    for (int p = 0; p < n_pass; p++) {
        if (!pass_w[p] || !pass_h[p]) continue;
        const uint8_t *prev = nullptr;
        for (size_t r = 0; r < pass_h[p]; r++) {
            uint8_t *cur = raw.data() + off + 1;
            rc = unfilter_row(cur, prev, pass_rowbytes[p], bpp);
            if (rc != ORBFE_OK) return rc;
            const size_t y = h.interlace ? (size_t)y0[p] + r * dy[p] : r;
            uint8_t *drow = dst + y * dst_stride;
            if (!h.interlace && h.color == 0 && h.depth == 8) memcpy(drow, cur, h.w); // the KITTI / EuRoC case: bytes as they are
            else
                for (size_t i = 0; i < pass_w[p]; i++) put_pixel(c, cur, i, drow, h.interlace ? (size_t)x0[p] + i * dx[p] : i);
            prev = cur;
            off += pass_rowbytes[p] + 1;
        }
    }
    return ORBFE_OK;
}

} // namespace

extern "C" const char *orbfe_png_last_error(void) { return t_err; }

extern "C" int orbfe_png_info(const uint8_t *file, size_t size, int *width, int *height, int *channels, int *bit_depth)
try {
    Chunks c;
    const int rc = parse(file, size, c, false);
    if (rc != ORBFE_OK) return rc;
    if (width) *width = (int)c.hd.w;
    if (height) *height = (int)c.hd.h;
    if (channels) *channels = c.hd.out_channels;
    if (bit_depth) *bit_depth = c.hd.out_depth;
    return ORBFE_OK;
} catch (const std::bad_alloc &) {
    return png_fail("out of host memory");
} catch (const std::exception &e) {
    snprintf(t_err, sizeof(t_err), "%s", e.what());
    return ORBFE_ERR_INVALID;
} catch (...) {
    return png_fail("unexpected exception");
}

extern "C" int orbfe_png_decode(const uint8_t *file, size_t size, uint8_t *dst, size_t dst_bytes, size_t dst_stride,
                                int *width, int *height, int *channels, int *bit_depth)
try {
    if (!dst) return png_fail("null destination");
    Chunks c;
    int rc = parse(file, size, c, false);
    if (rc != ORBFE_OK) return rc;
    const size_t row = (size_t)c.hd.w * c.hd.out_channels * (c.hd.out_depth / 8), stride = dst_stride ? dst_stride : row;
    if (stride < row || dst_bytes < stride * (c.hd.h - 1) + row) { snprintf(t_err, sizeof(t_err), "destination too small"); return ORBFE_ERR_CAPACITY; }
    PngHeader hd;
    rc = decode(file, size, dst, stride, &hd);
    if (rc != ORBFE_OK) return rc;
    if (width) *width = (int)hd.w;
    if (height) *height = (int)hd.h;
    if (channels) *channels = hd.out_channels;
    if (bit_depth) *bit_depth = hd.out_depth;
    return ORBFE_OK;
} catch (const std::bad_alloc &) {
    return png_fail("out of host memory");
} catch (const std::exception &e) {
    snprintf(t_err, sizeof(t_err), "%s", e.what());
    return ORBFE_ERR_INVALID;
} catch (...) {
    return png_fail("unexpected exception");
}

// n files of one common geometry (a camera stream), one image per worker thread, into dst[i * image_bytes ..]: what feeds a
// batched orbfe_enqueue_* call.  Returns the first error (the other images are still decoded).
extern "C" int orbfe_png_decode_batch(const uint8_t *const *files, const size_t *sizes, int n, uint8_t *dst, size_t image_bytes,
                                      int width, int height, int channels, int bit_depth, int threads)
try {
    if (!files || !sizes || !dst || n < 0 || width < 1 || height < 1) return png_fail("bad argument");
    const size_t need = (size_t)width * height * channels * (bit_depth / 8);
    if (image_bytes < need) { snprintf(t_err, sizeof(t_err), "image_bytes smaller than one decoded image"); return ORBFE_ERR_CAPACITY; }
    std::atomic<int> next(0), first_err(ORBFE_OK);
    char err_msg[256] = "";
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            int w = 0, h = 0, ch = 0, bd = 0;
            int rc = orbfe_png_info(files[i], sizes[i], &w, &h, &ch, &bd); // catches its own exceptions
            if (rc == ORBFE_OK && (w != width || h != height || ch != channels || bd != bit_depth)) {
                snprintf(t_err, sizeof(t_err), "image %d is %dx%d x%d channels x%d bit, the batch expects %dx%d x%d x%d", i, w, h, ch, bd, width, height, channels, bit_depth);
                rc = ORBFE_ERR_UNSUPPORTED;
            }
            if (rc == ORBFE_OK) {
                try { rc = decode(files[i], sizes[i], dst + (size_t)i * image_bytes, 0, nullptr); } // a worker must not throw: std::terminate
                catch (const std::bad_alloc &) { rc = png_fail("out of host memory"); }
                catch (...) { rc = png_fail("unexpected exception"); }
            }
            int expected = ORBFE_OK;
            if (rc != ORBFE_OK && first_err.compare_exchange_strong(expected, rc)) snprintf(err_msg, sizeof(err_msg), "%s", t_err);
        }
    };
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > n) nt = n > 0 ? n : 1;
    if (nt > 64) nt = 64; // threads = 0 on a large host: one image per worker is all the parallelism there is
    if (nt == 1) work();
    else {
        std::vector<std::thread> pool;
        pool.reserve(nt);
        try { for (int t = 0; t < nt; t++) pool.emplace_back(work); }
        catch (const std::system_error &) {} // fewer workers than asked for: the ones that started (or this thread) finish the queue
        if (pool.empty()) work();
        for (auto &t : pool) t.join();
    }
    if (first_err.load() != ORBFE_OK) snprintf(t_err, sizeof(t_err), "%s", err_msg);
    return first_err.load();
} catch (const std::bad_alloc &) {
    return png_fail("out of host memory");
} catch (const std::exception &e) {
    snprintf(t_err, sizeof(t_err), "%s", e.what());
    return ORBFE_ERR_INVALID;
} catch (...) {
    return png_fail("unexpected exception");
}
