// tools/pmc_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the
// access widths the deflate kernels use (MI355X_MICROARCH.md: only 16 B/lane streams are
// calibrated; "calibrate on a known byte count in your own access pattern").
//   hipcc -O3 --offload-arch=gfx950 -o build_variants/pmc_calib tools/pmc_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o calib --output-format csv -- ./pmc_calib
// Each kernel streams the same 2 GiB buffer (8x the Infinity Cache) exactly once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read_4B_per_lane(const uint32_t *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc ^= p[i];
    if (acc == 0x12345678u)
        *sink = acc;
}
__global__ void read_16B_per_lane(const uint4 *p, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        *sink = acc;
}
// one wave reads 64 consecutive dwords at a random 256-byte aligned place, like a chain batch
__global__ void read_256B_batches_random(const uint32_t *p, size_t nbatch, uint32_t *sink)
{
    uint32_t acc = 0;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t b = wave; b < nbatch; b += nw) {
        size_t j = (b * 0x9E3779B97F4A7C15ull) % nbatch; // a permutation-ish scatter over the buffer
        acc ^= p[j * 64 + (threadIdx.x & 63)];
    }
    if (acc == 0x12345678u)
        *sink = acc;
}
__global__ void write_4B_per_lane(uint32_t *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (uint32_t)i;
}
__global__ void write_2B_scattered(uint16_t *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[(i * 0x9E3779B1ull) % n] = (uint16_t)i;
}

int main()
{
    const size_t bytes = 2ull << 30;
    void *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess)
        return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    read_4B_per_lane<<<4096, 256>>>((const uint32_t *)buf, bytes / 4, (uint32_t *)sink);
    read_16B_per_lane<<<4096, 256>>>((const uint4 *)buf, bytes / 16, (uint32_t *)sink);
    read_256B_batches_random<<<4096, 256>>>((const uint32_t *)buf, bytes / 256, (uint32_t *)sink);
    write_4B_per_lane<<<4096, 256>>>((uint32_t *)buf, bytes / 4);
    write_2B_scattered<<<4096, 256>>>((uint16_t *)buf, (256ull << 20) / 2);
    hipDeviceSynchronize();
    printf("each read kernel: %zu bytes; write_4B: %zu bytes; write_2B_scattered: %zu bytes\n", bytes, bytes,
           (size_t)(256ull << 20));
    return 0;
}
