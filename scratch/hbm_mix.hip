// hbm_mix.hip -- what a streaming read of 6.4 GB (the scan's access pattern: every byte once, 16-byte non-temporal loads, 8 in
// flight per lane) loses when another stream's kernel touches HBM at the same time in small scattered pieces, as the filter and contour
// kernels do -- and what it loses to a kernel that only occupies wave slots.  Round 3: the pipelined step is the scan plus ~0.6 of the
// other kernels' stand-alone time whatever the placement; this probe separates wave slots, registers and memory-side requests.
// Parts: scattered reads / writes / no memory beside the stream; fat VALU kernels launched after the stream (82 / 202 / 493 registers);
// who runs when with the stream as a plain or persistent launch; VALU / LDS kernels of 2048 .. 512 fat waves; the stream's load policies
// against a small re-read set.  Results: profiles/history/r3_hbm_mix.log.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/hbm_mix scratch/hbm_mix.hip && scratch/hbm_mix
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY> __device__ __forceinline__ u32x4 load16(const uint8_t* p)
{ // 0 plain, 1 nt, 2 sc1 nt, 3 sc0 sc1 nt, 4 sc1, 5 sc0 sc1
    u32x4 v;
    if (POLICY == 0) return *(const u32x4*)p;
    if (POLICY == 1) return __builtin_nontemporal_load((const u32x4*)p);
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POLICY>
__global__ __launch_bounds__(256) void stream_read(const uint8_t* __restrict__ src, uint32_t* sink)
{ // one block = 32 KB: 8 rows of 4096 bytes, lane = 16 bytes of a row
    const uint8_t* p = src + (size_t)blockIdx.x * 32768 + threadIdx.x * 16;
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = load16<POLICY>(p + j * 4096);
    if (POLICY >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += __builtin_amdgcn_sad_u8(v[j].x, 0x3f3f3f3fu, 0u) + __builtin_amdgcn_sad_u8(v[j].y, 0x3f3f3f3fu, 0u) +
                                     __builtin_amdgcn_sad_u8(v[j].z, 0x3f3f3f3fu, 0u) + __builtin_amdgcn_sad_u8(v[j].w, 0x3f3f3f3fu, 0u);
    if (s == 0xffffffffu) *sink = s;
}

// mode 0: scattered 4-byte reads, each in a 64-byte line of its own (dependent on nothing: 8 in flight per lane)
// mode 1: scattered 4-byte writes; mode 2: no memory at all (the same number of waves busy for about as long)
__global__ __launch_bounds__(64) void scatter(uint32_t* __restrict__ buf, size_t lines, int per_lane, int mode, uint32_t* sink)
{
    uint32_t h = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < per_lane; i += 8) {
        uint32_t idx[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { h = h * 1664525u + 1013904223u; idx[j] = (uint32_t)(((uint64_t)h * lines) >> 32); }
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc += buf[(size_t)idx[j] * 16];
        } else if (mode == 1) {
#pragma unroll
            for (int j = 0; j < 8; j++) buf[(size_t)idx[j] * 16] = h + j;
        } else {
            for (int k = 0; k < 24; k++) {
#pragma unroll
                for (int j = 0; j < 8; j++) acc = acc * 1664525u + idx[j];
            }
        }
    }
    if (acc == 0xdeadbeefu) *sink = acc;
}

// a kernel shaped like box_filter_kernel: `waves` one-wave workgroups, ~240 registers each (two per SIMD), 20 KB of LDS, busy with
// VALU work (mode 0) or with LDS reads and writes (mode 1) for `iters` rounds, no global memory
template <int NR>
__global__ __launch_bounds__(64) void busy_n(int iters, uint32_t* sink)
{ // NR live registers per lane, VALU only, 20 KB of LDS like the box kernel
    __shared__ uint32_t lds[5120];
    uint32_t r[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) r[i] = threadIdx.x * 2654435761u + i;
    lds[threadIdx.x] = r[0];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NR; i++) r[i] = r[i] * 1664525u + r[(i + 7) % NR];
    }
    uint32_t acc = lds[(threadIdx.x * 7) & 63];
#pragma unroll
    for (int i = 0; i < NR; i++) acc ^= r[i];
    if (acc == 0xdeadbeefu) *sink = acc;
}

__global__ __launch_bounds__(64) void busy(int iters, int mode, uint32_t* sink)
{
    __shared__ uint32_t lds[5120];
    uint32_t r[200];
#pragma unroll
    for (int i = 0; i < 200; i++) r[i] = threadIdx.x * 2654435761u + i;
    for (int i = threadIdx.x; i < 5120; i += 64) lds[i] = i;
    __syncthreads();
    for (int it = 0; it < iters; it++) {
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 200; i++) r[i] = r[i] * 1664525u + r[(i + 7) % 200];
        } else {
#pragma unroll
            for (int i = 0; i < 200; i += 4) {
                const uint32_t a = lds[(r[i] + threadIdx.x) % 5120];
                lds[(a + threadIdx.x * 17) % 5120] = r[i + 1];
                r[i] += a;
            }
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 200; i++) acc ^= r[i];
    if (acc == 0xdeadbeefu) *sink = acc;
}

// the same streaming read as a persistent launch: `gridDim.x` workgroups take 32 KB blocks from a shared counter, 16 at a time
__global__ __launch_bounds__(256) void stream_read_persistent(const uint8_t* __restrict__ src, int nblk, uint32_t* ctr, uint32_t* sink)
{
    __shared__ uint32_t base;
    uint32_t s = 0;
    for (;;) {
        if (threadIdx.x == 0) base = atomicAdd(ctr, 16u);
        __syncthreads();
        const uint32_t b0 = base;
        __syncthreads();
        if (b0 >= (uint32_t)nblk) break;
        for (uint32_t b = b0; b < b0 + 16 && b < (uint32_t)nblk; b++) {
            const uint8_t* p = src + (size_t)b * 32768 + threadIdx.x * 16;
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load((const u32x4*)(p + j * 4096));
#pragma unroll
            for (int j = 0; j < 8; j++) s += __builtin_amdgcn_sad_u8(v[j].x, 0x3f3f3f3fu, 0u) + __builtin_amdgcn_sad_u8(v[j].y, 0x3f3f3f3fu, 0u) +
                                             __builtin_amdgcn_sad_u8(v[j].z, 0x3f3f3f3fu, 0u) + __builtin_amdgcn_sad_u8(v[j].w, 0x3f3f3f3fu, 0u);
        }
    }
    if (s == 0xffffffffu) *sink = s;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t big = 6370099200ull / 32768 * 32768; // the bench's 3072 frames
    const size_t lines = (1ull << 30) / 64;          // 1 GB of lines for the scattered kernel
    uint8_t* src; uint32_t* buf; uint32_t* sink;
    CK(hipMalloc(&src, big)); CK(hipMalloc(&buf, lines * 64)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, big)); CK(hipMemset(buf, 0, lines * 64));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int nblk = (int)(big / 32768);
    int policy = 1;
    size_t set_lines = lines;
    auto run = [&](int mode, int waves, int per_lane, bool with_stream, bool with_scatter) -> double {
        (void)hipDeviceSynchronize();
        const double t0 = now_ms();
        if (with_stream) {
            if (policy == 0) stream_read<0><<<nblk, 256, 0, s1>>>(src, sink);
            if (policy == 1) stream_read<1><<<nblk, 256, 0, s1>>>(src, sink);
            if (policy == 2) stream_read<2><<<nblk, 256, 0, s1>>>(src, sink);
            if (policy == 3) stream_read<3><<<nblk, 256, 0, s1>>>(src, sink);
            if (policy == 4) stream_read<4><<<nblk, 256, 0, s1>>>(src, sink);
            if (policy == 5) stream_read<5><<<nblk, 256, 0, s1>>>(src, sink);
        }
        if (with_scatter) scatter<<<waves, 64, 0, s2>>>(buf, set_lines, per_lane, mode, sink);
        (void)hipDeviceSynchronize();
        return now_ms() - t0;
    };
    run(0, 256, 64, true, true); // warm-up
    const char* names[3] = {"scattered reads ", "scattered writes", "no memory       "};
    for (int rep = 0; rep < 1; rep++) {
        const double alone = run(0, 0, 0, true, false);
        printf("stream alone: %.3f ms (%.2f TB/s)\n", alone, big / alone / 1e9);
        for (int mode = 0; mode < 3; mode++)
            for (int waves = 1024; waves <= 4096; waves *= 4)
                for (int per_lane = 32; per_lane <= 128; per_lane *= 2) {
                    const double mb = (double)waves * 64 * per_lane * 64 / 1e6; // bytes of DRAM lines touched
                    const double a = run(mode, waves, per_lane, false, true), both = run(mode, waves, per_lane, true, true);
                    printf("%s %5d waves x %3d per lane (%6.0f MB of lines): alone %.3f ms, with the stream %.3f ms\n", names[mode], waves, per_lane,
                           mode == 2 ? 0.0 : mb, a, both);
                }
    }
    {   // a VALU kernel of 2048 one-wave workgroups with 40 / 100 / 200 live registers launched 20 us AFTER the stream has started
        hipEvent_t e0, a1, b0, b1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
        for (int rep = 0; rep < 2; rep++)
            for (int nr = 0; nr < 3; nr++) {
                float alone = 0;
                for (int with_stream = 0; with_stream < 2; with_stream++) {
                    (void)hipDeviceSynchronize();
                    CK(hipEventRecord(e0, s1));
                    if (with_stream) { stream_read<1><<<nblk, 256, 0, s1>>>(src, sink); CK(hipEventRecord(a1, s1)); }
                    const double t0 = now_ms();
                    while (now_ms() - t0 < 0.02) { }
                    CK(hipEventRecord(b0, s2));
                    if (nr == 0) busy_n<40><<<2048, 64, 0, s2>>>(1500, sink);
                    if (nr == 1) busy_n<100><<<2048, 64, 0, s2>>>(600, sink);
                    if (nr == 2) busy_n<200><<<2048, 64, 0, s2>>>(300, sink);
                    CK(hipEventRecord(b1, s2));
                    (void)hipDeviceSynchronize();
                    float ta1 = 0, tb0, tb1;
                    if (with_stream) (void)hipEventElapsedTime(&ta1, e0, a1);
                    (void)hipEventElapsedTime(&tb0, e0, b0); (void)hipEventElapsedTime(&tb1, e0, b1);
                    if (!with_stream) alone = tb1 - tb0;
                    else printf("VALU kernel with %3d live registers, launched after the stream: alone %.3f ms; beside the stream %.3f .. %.3f ms, stream ends at %.3f ms\n",
                                nr == 0 ? 40 : nr == 1 ? 100 : 200, alone, tb0, tb1, ta1);
                }
            }
    }
    {   // who runs when: events around each kernel on its own stream, both referred to one start event
        hipEvent_t e0, a0, a1, b0, b1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
        uint32_t* ctr; CK(hipMalloc(&ctr, 64));
        for (int rep = 0; rep < 1; rep++)
            for (int form = 0; form < 4; form++)      // 0: one workgroup per block; 1 / 2 / 3: persistent, 8 / 4 / 2 workgroups per CU
                for (int order = 0; order < 2; order++) { // which kernel is launched first
                    (void)hipDeviceSynchronize();
                    CK(hipMemsetAsync(ctr, 0, 64, s1));
                    (void)hipDeviceSynchronize();
                    CK(hipEventRecord(e0, s1));
                    auto launch_a = [&]() {
                        (void)hipEventRecord(a0, s1);
                        if (form == 0) stream_read<1><<<nblk, 256, 0, s1>>>(src, sink);
                        else stream_read_persistent<<<256 * (16 >> form), 256, 0, s1>>>(src, nblk, ctr, sink);
                        (void)hipEventRecord(a1, s1);
                    };
                    auto launch_b = [&]() {
                        (void)hipEventRecord(b0, s2);
                        busy<<<1024, 64, 0, s2>>>(400, 0, sink);
                        (void)hipEventRecord(b1, s2);
                    };
                    if (order == 0) { launch_a(); launch_b(); } else { launch_b(); launch_a(); }
                    (void)hipDeviceSynchronize();
                    float ta0, ta1, tb0, tb1;
                    (void)hipEventElapsedTime(&ta0, e0, a0); (void)hipEventElapsedTime(&ta1, e0, a1);
                    (void)hipEventElapsedTime(&tb0, e0, b0); (void)hipEventElapsedTime(&tb1, e0, b1);
                    printf("stream %s, %s first: stream %.3f .. %.3f ms, VALU kernel (1024 fat waves, 0.39 ms alone) %.3f .. %.3f ms\n",
                           form == 0 ? "one workgroup per block" : form == 1 ? "persistent 8 per CU    " : form == 2 ? "persistent 4 per CU    " : "persistent 2 per CU    ",
                           order == 0 ? "stream" : "VALU  ", ta0, ta1, tb0, tb1);
                }
    }
    for (int rep = 0; rep < 1; rep++)
        for (int mode = 0; mode < 2; mode++)
          for (int nwaves = 2048; nwaves >= 512; nwaves /= 2)
            for (int iters = 100; iters <= 200; iters *= 2) {
                policy = 1;
                auto runb = [&](bool with_stream) {
                    (void)hipDeviceSynchronize();
                    const double t0 = now_ms();
                    if (with_stream) stream_read<1><<<nblk, 256, 0, s1>>>(src, sink);
                    busy<<<nwaves, 64, 0, s2>>>(iters * (2048 / nwaves), mode, sink);
                    (void)hipDeviceSynchronize();
                    return now_ms() - t0;
                };
                runb(false);
                const double a = runb(false), both = runb(true);
                printf("%s kernel, %d waves of ~240 registers, %3d rounds: alone %.3f ms, with the stream %.3f ms\n", mode ? "LDS " : "VALU", nwaves, iters * (2048 / nwaves), a, both);
            }
    // a SMALL set (32 / 128 MB of lines) read over and over beside the stream: does it stay in the memory-side cache (256 MB) under
    // each load policy of the stream?  4096 waves x 128 per lane = 33.5 M reads = 64 x the 32 MB set
    const char* pol[6] = {"plain", "nt", "sc1 nt", "sc0 sc1 nt", "sc1", "sc0 sc1"};
    for (int rep = 0; rep < 1; rep++)
        for (policy = 0; policy < 6; policy++) {
            set_lines = lines;
            const double alone = run(0, 0, 0, true, false);
            printf("stream [%-10s] alone %.3f ms |", pol[policy], alone);
            for (size_t mbs : {32, 128, 1024}) {
                set_lines = mbs * (1ull << 20) / 64;
                run(0, 4096, 128, false, true);
                const double a = run(0, 4096, 128, false, true), both = run(0, 4096, 128, true, true);
                printf("  %4zu MB set: reads alone %.3f, with the stream %.3f (+%.3f)", mbs, a, both, both - alone);
            }
            printf("\n");
        }
    return 0;
}
