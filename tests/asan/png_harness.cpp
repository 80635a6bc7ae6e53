// tests/asan/png_harness.cpp -- TEST-ONLY: orbslam2_amd/csrc/orbfe_png.cpp (the same translation unit liborbfe.so links) under
// AddressSanitizer + UBSan: every fixture, every truncation of it, and seeded random byte corruptions (with and without a
// repaired chunk CRC, so the corruption reaches the inflate / unfilter / expansion code) must either decode or be refused --
// never read or write out of bounds.  Prints a checksum of the successful decodes for the pytest to compare.
//   usage: png_harness file.png [file.png ...]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include <zlib.h>

#include "../../include/orbfe.h"

static std::vector<uint8_t> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
    std::vector<uint8_t> v((size_t)f.tellg());
    f.seekg(0);
    f.read((char *)v.data(), (std::streamsize)v.size());
    return v;
}

// decode into an exactly sized heap buffer (ASan guards both ends); returns 1 on success
static int try_decode(const std::vector<uint8_t> &file, uint64_t *sum)
{
    int w = 0, h = 0, ch = 0, bd = 0;
    if (orbfe_png_info(file.data(), file.size(), &w, &h, &ch, &bd) != ORBFE_OK) return 0;
    const size_t bytes = (size_t)w * h * ch * (bd / 8);
    if (bytes > (64u << 20)) return 0; // a corrupted IHDR may claim a huge image: refuse to allocate, as a caller would
    std::vector<uint8_t> out(bytes);
    if (orbfe_png_decode(file.data(), file.size(), out.data(), out.size(), 0, &w, &h, &ch, &bd) != ORBFE_OK) return 0;
    if (sum) for (size_t i = 0; i < out.size(); i++) *sum = *sum * 1099511628211ull + out[i];
    return 1;
}

static void fix_crcs(std::vector<uint8_t> &f)
{
    size_t p = 8;
    while (p + 12 <= f.size()) {
        const uint32_t len = ((uint32_t)f[p] << 24) | ((uint32_t)f[p + 1] << 16) | ((uint32_t)f[p + 2] << 8) | f[p + 3];
        if ((size_t)len > f.size() - p - 12) return;
        const uint32_t c = (uint32_t)crc32(crc32(0L, Z_NULL, 0), &f[p + 4], len + 4);
        f[p + 8 + len] = (uint8_t)(c >> 24); f[p + 9 + len] = (uint8_t)(c >> 16); f[p + 10 + len] = (uint8_t)(c >> 8); f[p + 11 + len] = (uint8_t)c;
        p += 12 + (size_t)len;
    }
}

int main(int argc, char **argv)
{
    uint64_t sum = 1469598103934665603ull;
    unsigned long ok = 0, refused = 0;
    uint32_t lcg = 12345u;
    for (int a = 1; a < argc; a++) {
        const std::vector<uint8_t> good = slurp(argv[a]);
        if (!try_decode(good, &sum)) { std::fprintf(stderr, "fixture %s does not decode\n", argv[a]); return 1; }
        ok++;
        for (size_t n = 0; n < good.size(); n += (good.size() > 400 ? 7 : 1)) { // truncations
            std::vector<uint8_t> t(good.begin(), good.begin() + n);
            try_decode(t, nullptr) ? ok++ : refused++;
        }
        for (int k = 0; k < 300; k++) { // corruptions
            std::vector<uint8_t> t(good);
            const int flips = 1 + (int)(lcg % 3);
            for (int j = 0; j < flips; j++) {
                lcg = lcg * 1664525u + 1013904223u;
                t[8 + (lcg >> 8) % (t.size() - 8)] ^= (uint8_t)(1u << ((lcg >> 3) & 7));
            }
            if (k & 1) fix_crcs(t);
            try_decode(t, nullptr) ? ok++ : refused++;
        }
    }
    std::printf("png harness ok: %lu decoded, %lu refused, checksum %016llx\n", ok, refused, (unsigned long long)sum);
    return 0;
}
