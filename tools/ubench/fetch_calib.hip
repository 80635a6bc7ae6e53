// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for the access shapes of the stereo / describe kernels?
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE is exactly 1/2 of the bytes of a wide coalesced streaming read (16 B per lane); other
// shapes are uncalibrated ("calibrate on a known byte count in your own access pattern before trusting an absolute").  Four kernels,
// each reading a KNOWN number of distinct bytes exactly once from a buffer far larger than the caches (1 GiB, so no reuse):
//   stream16   : 16 B per lane, consecutive lanes consecutive (the guide's calibrated case)
//   gather32   : 32-byte records at random 32-byte-aligned places, two 16-B loads per lane (descriptor fetches of stereo_match_kernel)
//   list8      : 8 B per lane, 16 consecutive lanes read 128 contiguous bytes of a random row (row-list entries)
//   rows16     : 16 B per lane, lane pairs read 32 contiguous bytes at an arbitrary byte offset, pair after pair `pitch` bytes apart
//                (raw patches of describe_kernel, SAD windows)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/ubench/fetch_calib   (prints the bytes each kernel read)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
struct __attribute__((packed, aligned(4))) u4u { uint32_t x, y, z, w; };

__global__ void stream16(const uint4 *src, uint32_t *sink, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) { const uint4 v = src[i]; if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = 1; }
}
__global__ void gather32(const uint8_t *src, uint32_t *sink, size_t n_rec, size_t total_rec)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    // a permutation-like scatter: record i of n_rec reads slot (i * stride) % total_rec with an odd stride -> every slot at most once
    const size_t slot = (i * 2654435761ull) % total_rec;
    const uint4 *p = (const uint4 *)(src + slot * 32);
    const uint4 a = p[0], b = p[1];
    if ((a.x ^ b.w) == 0x12345678u) sink[0] = 1;
}
__global__ void list8(const uint8_t *src, uint32_t *sink, size_t n_groups, size_t total_rows)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, g = t >> 4;
    if (g >= n_groups) return;
    const size_t row = (g * 2654435761ull) % total_rows; // rows of 128 B
    const uint2 v = *(const uint2 *)(src + row * 128 + (t & 15) * 8);
    if ((v.x ^ v.y) == 0x12345678u) sink[0] = 1;
}
__global__ void rows16(const uint8_t *src, uint32_t *sink, size_t n_pairs, size_t total_rows, int pitch)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, pr = t >> 1;
    if (pr >= n_pairs) return;
    const size_t row = (pr * 2654435761ull) % total_rows;
    const uint32_t off = mix((uint32_t)pr) % (uint32_t)(pitch - 32);
    const u4u v = *(const u4u *)(src + row * (size_t)pitch + (off & ~3u) + 16 * (t & 1)); // 4-byte aligned start, as in the kernels
    if ((v.x ^ v.w) == 0x12345678u) sink[0] = 1;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    uint8_t *d; uint32_t *sink;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes); hipMemset(sink, 0, 4);
    hipDeviceSynchronize();
    const size_t n16 = bytes / 16 / 4;             // 256 MiB streamed
    const size_t n_rec = (size_t)4 << 20, total_rec = bytes / 32;   // 4 M records = 128 MiB
    const size_t n_groups = (size_t)1 << 20, total_rows = bytes / 128; // 1 M rows x 128 B = 128 MiB
    const int pitch = 1280;
    const size_t n_pairs = (size_t)2 << 20, rows = bytes / pitch;   // 2 M x 32 B = 64 MiB of requested bytes
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(stream16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const uint4 *)d + rep * n16, sink, n16);
        hipLaunchKernelGGL(gather32, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, 0, d, sink, n_rec, total_rec);
        hipLaunchKernelGGL(list8, dim3((unsigned)((n_groups * 16 + 255) / 256)), dim3(256), 0, 0, d, sink, n_groups, total_rows);
        hipLaunchKernelGGL(rows16, dim3((unsigned)((n_pairs * 2 + 255) / 256)), dim3(256), 0, 0, d, sink, n_pairs, rows, pitch);
        hipDeviceSynchronize();
    }
    printf("{\"stream16_bytes\": %zu, \"gather32_bytes\": %zu, \"list8_bytes\": %zu, \"rows16_bytes\": %zu, \"rows16_note\": \"32 requested bytes per pair at a 4-byte-aligned offset: 1 or 2 64-byte sectors\"}\n",
           n16 * 16, n_rec * 32, n_groups * 128, n_pairs * 32);
    return 0;
}
