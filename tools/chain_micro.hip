// What does a dependent kernel boundary cost on this box, and what do the step's cross-stream events add to it?
// (the replayed DyCON step has ~226 dependent launches on its main stream and ~6 us between them, against
// the 1.45-1.9 us MI355X_MICROARCH.md quotes; ~50 fork / join events per step; 9.4 us of host time per launch.)
//
//   hipcc -O2 --offload-arch=gfx950 tools/chain_micro.hip -o /tmp/chain_micro && /tmp/chain_micro
//
// Every case: a chain of N dependent launches on stream A (each kernel reads what its predecessor wrote), timed on the GPU with
// timing events around the whole chain and on the host around the enqueue loop.  Variants add, after every k-th launch, what the
// step's backward adds: an event record on A + a wait on stream B + one launch on B (the weight-gradient fork), with the event
// created (a) as torch creates it (hipEventDisableTiming: system-scope release at record), (b) with hipEventDisableSystemFence,
// (c) with hipEventReleaseToDevice.  Two kernel sizes: trivial (256 workgroups, 4 KB) and streaming (28 MB in, 28 MB out = what a
// 48^3 x 32-channel bf16 normalisation pass of the step moves).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stream_kernel(const float4* __restrict__ in, float4* __restrict__ out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float4 v = in[i];
        v.x += 1.f;
        out[i] = v;
    }
}

__global__ void spin_kernel(long long ticks, float* out) {          // keeps the stream busy while the host pre-queues a chain
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out && threadIdx.x == 0) out[0] = 1.f;
}

struct Case { const char* name; long n16; int every; unsigned flags; bool other_kernel; int streams; };

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 240, REPS = 20;
    const long BIG = 28l << 20;
    float4 *a, *b, *c, *d;
    CK(hipMalloc(&a, BIG)); CK(hipMalloc(&b, BIG)); CK(hipMalloc(&c, BIG)); CK(hipMalloc(&d, BIG));
    CK(hipMemset(a, 0, BIG)); CK(hipMemset(c, 0, BIG));
    hipStream_t A, B, S[4];
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    for (auto& s : S) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const unsigned DT = hipEventDisableTiming, NF = hipEventDisableSystemFence, RD = hipEventReleaseToDevice;
    const long SMALL = 256, LARGE = BIG / 16;
    const Case cases[] = {
        {"trivial kernels, one stream, no events", SMALL, 0, 0, false, 1},
        {"trivial, record (torch flags) after every launch, no waiter", SMALL, 1, DT, false, 1},
        {"trivial, record (DisableSystemFence) after every launch", SMALL, 1, DT | NF, false, 1},
        {"trivial, record (ReleaseToDevice) after every launch", SMALL, 1, DT | RD, false, 1},
        {"trivial, fork to stream B after every launch (torch flags)", SMALL, 1, DT, true, 1},
        {"trivial, fork to stream B after every launch (DisableSystemFence)", SMALL, 1, DT | NF, true, 1},
        {"trivial, fork to stream B after every launch (ReleaseToDevice)", SMALL, 1, DT | RD, true, 1},
        {"trivial, fork to B after every 4th launch (torch flags)", SMALL, 4, DT, true, 1},
        {"trivial, fork to B after every 4th launch (ReleaseToDevice)", SMALL, 4, DT | RD, true, 1},
        {"streaming 28 MB kernels, one stream, no events", LARGE, 0, 0, false, 1},
        {"streaming, fork to B after every launch (torch flags)", LARGE, 1, DT, true, 1},
        {"streaming, fork to B after every launch (DisableSystemFence)", LARGE, 1, DT | NF, true, 1},
        {"streaming, fork to B after every launch (ReleaseToDevice)", LARGE, 1, DT | RD, true, 1},
        {"streaming, fork to B after every 4th launch (torch flags)", LARGE, 4, DT, true, 1},
        {"streaming, fork to B after every 4th launch (ReleaseToDevice)", LARGE, 4, DT | RD, true, 1},
        {"trivial, 4 independent chains on 4 streams, no events (per-stream N/4)", SMALL, 0, 0, false, 4},
        {"streaming, 4 independent chains on 4 streams, no events (per-stream N/4)", LARGE / 4, 0, 0, false, 4},
    };
    printf("%d launches per chain, %d repetitions; us per launch of the chain\n", N, REPS);
    printf("%-78s %9s %9s\n", "case", "GPU", "host");
    for (const Case& cs : cases) {
        std::vector<hipEvent_t> evs(N);
        if (cs.every) for (auto& e : evs) CK(hipEventCreateWithFlags(&e, cs.flags));
        const int grid = cs.n16 == SMALL ? 256 : 2048;
        double gpu = 0, host = 0;
        for (int r = -2; r < REPS; ++r) {
            CK(hipDeviceSynchronize());
            auto h0 = std::chrono::steady_clock::now();
            if (cs.streams == 1) {
                CK(hipEventRecord(t0, A));
                for (int i = 0; i < N; ++i) {
                    hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, A, (i & 1) ? b : a, (i & 1) ? a : b, cs.n16);
                    if (cs.every && i % cs.every == cs.every - 1) {
                        CK(hipEventRecord(evs[i], A));
                        if (cs.other_kernel) {
                            CK(hipStreamWaitEvent(B, evs[i], 0));
                            hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(256), 0, B, c, d, SMALL);
                        }
                    }
                }
                CK(hipEventRecord(t1, A));
            } else {
                CK(hipEventRecord(t0, S[0]));
                for (int i = 0; i < N / 4; ++i)
                    for (int q = 0; q < 4; ++q)
                        hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, S[q], (float4*)((char*)a + q * (BIG / 4)),
                                           (float4*)((char*)b + q * (BIG / 4)), cs.n16);
                for (int q = 1; q < 4; ++q) { CK(hipEventRecord(t1, S[q])); CK(hipStreamWaitEvent(S[0], t1, 0)); }
                CK(hipEventRecord(t1, S[0]));
            }
            auto h1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (r >= 0) { gpu += ms * 1e3; host += std::chrono::duration<double, std::micro>(h1 - h0).count(); }
        }
        const int per = cs.streams == 1 ? N : N / 4;
        printf("%-78s %9.2f %9.2f\n", cs.name, gpu / REPS / per, host / REPS / per);
        if (cs.every) for (auto& e : evs) CK(hipEventDestroy(e));
    }
    // ---- the same chains PRE-QUEUED behind a 4 ms blocker kernel (100 MHz wall clock): the host has enqueued everything before the
    // first chain kernel may start, so the figure is the GPU's own cost per dependent launch, not the host's enqueue rate
    for (int big = 0; big < 2; ++big) {
        double gpu = 0;
        for (int r = -1; r < 5; ++r) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, A, 400000ll, (float*)c);
            CK(hipEventRecord(t0, A));
            for (int i = 0; i < N; ++i)
                hipLaunchKernelGGL(stream_kernel, dim3(big ? 2048 : 256), dim3(256), 0, A, (i & 1) ? b : a, (i & 1) ? a : b, big ? LARGE : SMALL);
            CK(hipEventRecord(t1, A));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (r >= 0) gpu += ms * 1e3;
        }
        printf("%-78s %9.2f %9s\n", big ? "streaming 28 MB kernels, one stream, PRE-QUEUED behind a blocker" : "trivial kernels, one stream, PRE-QUEUED behind a blocker",
               gpu / 5 / N, "-");
    }
    // ---- the replayed DyCON step's launch pattern with trivial kernels: 456 launches on 4 streams (226 / 117 / 82 / 31), 55 forks
    // (event record on the producer + wait on the consumer): what the HIP runtime alone costs the host per step
    {
        const int per[4] = {226, 117, 82, 31};
        std::vector<int> order;
        int left[4] = {per[0], per[1], per[2], per[3]};
        for (int i = 0; i < 456; ++i) {                      // proportional interleave
            int q = 0;
            double best = -1;
            for (int k = 0; k < 4; ++k) { const double f = (double)left[k] / per[k]; if (left[k] > 0 && f > best) { best = f; q = k; } }
            order.push_back(q);
            --left[q];
        }
        for (unsigned flags : {DT, DT | NF}) {
            std::vector<hipEvent_t> evs(55);
            for (auto& e : evs) CK(hipEventCreateWithFlags(&e, flags));
            double gpu = 0, host = 0;
            for (int r = -2; r < REPS; ++r) {
                CK(hipDeviceSynchronize());
                auto h0 = std::chrono::steady_clock::now();
                CK(hipEventRecord(t0, S[0]));
                int nf = 0;
                for (int i = 0; i < 456; ++i) {
                    const int q = order[i];
                    hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(256), 0, S[q], (float4*)((char*)a + q * (BIG / 4)),
                                       (float4*)((char*)b + q * (BIG / 4)), SMALL);
                    if (i % 8 == 7 && nf < 55) { const int to = 1 + nf % 3; CK(hipEventRecord(evs[nf], S[0])); CK(hipStreamWaitEvent(S[to], evs[nf], 0)); ++nf; }
                }
                for (int q = 1; q < 4; ++q) { CK(hipEventRecord(t1, S[q])); CK(hipStreamWaitEvent(S[0], t1, 0)); }
                CK(hipEventRecord(t1, S[0]));
                auto h1 = std::chrono::steady_clock::now();
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, t0, t1));
                if (r >= 0) { gpu += ms * 1e3; host += std::chrono::duration<double, std::micro>(h1 - h0).count(); }
            }
            printf("step pattern: 456 trivial launches on 4 streams + 55 forks (%s): GPU %.0f us, host %.0f us per step\n",
                   flags == DT ? "torch event flags" : "hipEventDisableSystemFence", gpu / REPS, host / REPS);
            for (auto& e : evs) CK(hipEventDestroy(e));
        }
    }
    return 0;
}
