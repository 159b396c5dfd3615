// Microbenchmark: cost of vector-memory load instructions by width / alignment (cycles per instruction per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int BYTES, int OFF, int LSTRIDE, int ACTIVE = 64>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ base, uint32_t* out, int iters, int pitch)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint8_t* p = base + (size_t)(blockIdx.x & 63) * 65536 + wv * 16384 + lane * LSTRIDE + OFF;
    uint32_t acc = 0;
    if (lane >= ACTIVE) { out[blockIdx.x * 256 + threadIdx.x] = 0; return; }
    for (int it = 0; it < iters; it++) {
        const uint8_t* q = p + (uint32_t)((it & 31) * pitch);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (BYTES == 1) { acc += q[u * 1024]; }
            if (BYTES == 2) { uint16_t v; __builtin_memcpy(&v, q + u * 1024, 2); acc += v; }
            if (BYTES == 4) { uint32_t v; __builtin_memcpy(&v, q + u * 1024, 4); acc += v; }
            if (BYTES == 8) { uint2 v; __builtin_memcpy(&v, q + u * 1024, 8); acc += v.x ^ v.y; }
            if (BYTES == 16) { uint4 v; __builtin_memcpy(&v, q + u * 1024, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int BYTES, int OFF, int LSTRIDE, int ACTIVE = 64>
void run(const char* name, const uint8_t* buf, uint32_t* out)
{
    const int blocks = 2048, iters = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<BYTES, OFF, LSTRIDE, ACTIVE>), dim3(blocks), dim3(256), 0, 0, buf, out, 8, 260);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<BYTES, OFF, LSTRIDE, ACTIVE>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, 260);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts = (double)blocks * 4 * iters * 8; // wave-level load instructions
    double per_cu_ns = ms * 1e6 / (insts / 256.0);
    printf("%-28s %7.3f ms  %6.2f ns per wave-load per CU (~%5.1f cycles @2.1GHz)  %7.1f GB/s useful\n", name, ms, per_cu_ns, per_cu_ns * 2.1,
           insts * 64 * BYTES / (ms * 1e-3) / 1e9);
}

int main()
{
    uint8_t* buf; uint32_t* out;
    hipMalloc(&buf, 64 * 65536 + 65536); hipMemset(buf, 1, 64 * 65536 + 65536);
    hipMalloc(&out, 2048 * 256 * 4);
    run<1, 0, 4>("u8  stride4", buf, out);
    run<2, 0, 4>("u16 stride4 off0", buf, out);
    run<2, 1, 4>("u16 stride4 off1", buf, out);
    run<2, 3, 4>("u16 stride4 off3 (splits dword)", buf, out);
    run<4, 0, 4>("u32 stride4 off0", buf, out);
    run<4, 1, 4>("u32 stride4 off1", buf, out);
    run<8, 0, 4>("u64 stride4 off0", buf, out);
    run<8, 1, 4>("u64 stride4 off1", buf, out);
    run<8, 3, 4>("u64 stride4 off3", buf, out);
    run<8, 0, 8>("u64 stride8 off0", buf, out);
    run<16, 0, 16>("u128 stride16 off0", buf, out);
    run<16, 0, 4>("u128 stride4 off0", buf, out);
    run<16, 1, 4>("u128 stride4 off1", buf, out);
    run<2, 1, 4, 32>("u16 off1, 32 lanes", buf, out);
    run<2, 1, 4, 16>("u16 off1, 16 lanes", buf, out);
    run<2, 1, 4, 8>("u16 off1, 8 lanes", buf, out);
    run<16, 0, 4, 16>("u128 stride4 16 lanes", buf, out);
    return 0;
}
