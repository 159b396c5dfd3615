// xstream_dep.hip -- what a dependency between two HIP streams costs on this runtime (round 3: the scan forked onto a stream of its
// own serialised the three-batch pipeline; this probe separates the event / barrier latency from everything else).
//   hipcc --offload-arch=gfx950 -O3 -o scratch/xstream_dep scratch/xstream_dep.hip && scratch/xstream_dep
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(long long ticks, int* sink)
{ // one wave busy for `ticks` of the 100 MHz constant clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink && threadIdx.x == 1000) *sink = 1;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int N = 40;
    const long long T = 5000; // 50 us
    hipStream_t s[8];
    for (int i = 0; i < 8; i++) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(4 * N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    uint32_t* flag;
    CK(hipMalloc(&flag, 4096));
    CK(hipMemset(flag, 0, 4096));
    spin<<<1, 64, 0, s[0]>>>(100, nullptr);
    spin<<<1, 64, 0, s[1]>>>(100, nullptr);
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
        // 1: one stream
        double t0 = now_us();
        for (int i = 0; i < N; i++) spin<<<1, 64, 0, s[0]>>>(T, nullptr);
        double t1 = now_us();
        CK(hipDeviceSynchronize());
        double t2 = now_us();
        printf("one stream:            %7.1f us per kernel (host enqueue %5.1f us each)\n", (t2 - t0) / N, (t1 - t0) / N);
        // 2: alternating streams, event record / wait
        t0 = now_us();
        for (int i = 0; i < N; i++) {
            hipStream_t a = s[i & 1], b = s[(i & 1) ^ 1];
            spin<<<1, 64, 0, a>>>(T, nullptr);
            CK(hipEventRecord(ev[i], a));
            CK(hipStreamWaitEvent(b, ev[i], 0));
        }
        t1 = now_us();
        CK(hipDeviceSynchronize());
        t2 = now_us();
        printf("two streams, events:   %7.1f us per kernel (host enqueue %5.1f us each)\n", (t2 - t0) / N, (t1 - t0) / N);
        // 3: alternating streams, stream memory operations
        CK(hipMemset(flag, 0, 4096));
        t0 = now_us();
        bool ok = true;
        for (int i = 0; i < N && ok; i++) {
            hipStream_t a = s[i & 1], b = s[(i & 1) ^ 1];
            spin<<<1, 64, 0, a>>>(T, nullptr);
            ok = hipStreamWriteValue32(a, flag, (uint32_t)(i + 1), 0) == hipSuccess &&
                 hipStreamWaitValue32(b, flag, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
        }
        t1 = now_us();
        CK(hipDeviceSynchronize());
        t2 = now_us();
        printf("two streams, values:   %7.1f us per kernel (host enqueue %5.1f us each)%s\n", (t2 - t0) / N, (t1 - t0) / N, ok ? "" : "  [not supported]");
        // 4: three lanes, each a chain on its own stream (no dependencies between lanes): the lanes overlap
        for (int L = 1; L <= 3; L += 2) {
            t0 = now_us();
            for (int i = 0; i < N; i++)
                for (int l = 0; l < L; l++) spin<<<1, 64, 0, s[l]>>>(T, nullptr);
            CK(hipDeviceSynchronize());
            t2 = now_us();
            printf("%d lanes, plain:        %7.1f us per step\n", L, (t2 - t0) / N);
        }
        // 5: three lanes, each forking every second kernel onto a side stream (lane l: s[l] and s[3 + l])
        for (int L = 1; L <= 3; L += 2) {
            t0 = now_us();
            for (int i = 0; i < N; i++)
                for (int l = 0; l < L; l++) {
                    hipStream_t m = s[l], side = s[3 + l];
                    hipEvent_t e0 = ev[(i * 3 + l) % (2 * N)], e1 = ev[2 * N + (i * 3 + l) % (2 * N)];
                    if (i & 1) {
                        CK(hipEventRecord(e0, m));
                        CK(hipStreamWaitEvent(side, e0, 0));
                        spin<<<1, 64, 0, side>>>(T, nullptr);
                        CK(hipEventRecord(e1, side));
                        CK(hipStreamWaitEvent(m, e1, 0));
                    } else spin<<<1, 64, 0, m>>>(T, nullptr);
                }
            t1 = now_us();
            CK(hipDeviceSynchronize());
            t2 = now_us();
            printf("%d lanes, forked:       %7.1f us per step (host enqueue %5.1f us)\n", L, (t2 - t0) / N, (t1 - t0) / N);
        }
    }
    return 0;
}
