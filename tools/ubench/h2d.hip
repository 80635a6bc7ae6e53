// Host->device staging costs on the box: memcpy into pinned memory, DMA of the pinned block, both, and a pageable copy.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s; hipStreamCreate(&s);
    for (size_t bytes : {(size_t)466616, (size_t)921600, (size_t)1843200, (size_t)3686400}) {
        std::vector<uint8_t> src(bytes, 7);
        uint8_t *pin, *dev;
        hipHostMalloc((void **)&pin, bytes, hipHostMallocDefault);
        hipMalloc((void **)&dev, bytes);
        const int reps = 50;
        double t0, t_pack = 0, t_dma = 0, t_both = 0, t_page = 0;
        for (int i = 0; i < reps + 5; i++) {
            if (i == 5) t_pack = t_dma = t_both = t_page = 0;
            t0 = now(); memcpy(pin, src.data(), bytes); t_pack += now() - t0;
            t0 = now(); hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t_dma += now() - t0;
            t0 = now(); memcpy(pin, src.data(), bytes); hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t_both += now() - t0;
            t0 = now(); hipMemcpyAsync(dev, src.data(), bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t_page += now() - t0;
        }
        printf("%8zu B: pack %.3f ms  dma %.3f ms  pack+dma %.3f ms  pageable %.3f ms\n", bytes, t_pack / reps, t_dma / reps, t_both / reps, t_page / reps);
        hipHostFree(pin); hipFree(dev);
    }
    return 0;
}
