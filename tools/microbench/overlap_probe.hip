// Overlap probe (developer tool, round 3): can two kernels that follow each other in one stream / one graph be
// in flight at the same time on this stack, and what does a dependency carried by a memory counter cost against
// the hardware kernel boundary?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 overlap_probe.hip -o overlap_probe.bin
// Legs (each prints "B starts X us after A ends" - negative = overlapped - and us per launch of a 20-chain):
//   plain      : A, B on one stream (barrier bit set: the reference point)
//   anyorder   : hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch) on one stream
//   anyorder-g : the same captured into a hipGraph
//   fork-g     : B on a forked capture stream (parallel graph branches)
//   2streams   : eager on two streams
// and the counter chain: 20 kernels, kernel i's workgroups poll a device counter until every workgroup of kernel
// i-1 has arrived (bounded spin), launched plain / any-order / alternating on two streams.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

struct Stamp { unsigned long long start, end; };
constexpr int MAXB = 2048;          // per-block slots: plain stores, reduced on the host (contended atomics cost microseconds)

__device__ inline unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

// every workgroup: stamp start (min), spin `ticks` of the 100 MHz clock, stamp end (max)
__global__ __launch_bounds__(256) void work_kernel(Stamp* st, int idx, int ticks) {
    unsigned long long t0 = rt();
    if (threadIdx.x == 0) st[idx * MAXB + blockIdx.x].start = t0;
    while (rt() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) st[idx * MAXB + blockIdx.x].end = rt();
}

// counter chain: wait until cnt[i-1] == want (all workgroups of the predecessor arrived in this epoch), spin
// `ticks`, arrive on cnt[i].  Bounded: gives up after 20 ms and raises err.
__global__ __launch_bounds__(256) void chain_kernel(unsigned* cnt, int i, unsigned want, int ticks, Stamp* st, unsigned* err) {
    unsigned long long t0 = rt();
    if (threadIdx.x == 0) {
        st[i * MAXB + blockIdx.x].start = t0;
        if (i > 0) {
            while (__hip_atomic_load(&cnt[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (rt() - t0 > 2000000ull) { atomicAdd(err, 1u); break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    }
    __syncthreads();
    unsigned long long t1 = rt();
    while (rt() - t1 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        atomicAdd(&cnt[i], 1u);
        st[i * MAXB + blockIdx.x].end = rt();
    }
}

static Stamp* d_st; static Stamp h_st[64]; static std::vector<Stamp> h_raw(64 * MAXB);
static void reset_stamps(hipStream_t s) {
    for (auto& x : h_raw) { x.start = ~0ull; x.end = 0; }
    CK(hipMemcpyAsync(d_st, h_raw.data(), sizeof(Stamp) * h_raw.size(), hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
}
static void read_stamps() {
    CK(hipDeviceSynchronize()); CK(hipMemcpy(h_raw.data(), d_st, sizeof(Stamp) * h_raw.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < 64; ++i) {
        h_st[i].start = ~0ull; h_st[i].end = 0;
        for (int b = 0; b < MAXB; ++b) { h_st[i].start = std::min(h_st[i].start, h_raw[i * MAXB + b].start); h_st[i].end = std::max(h_st[i].end, h_raw[i * MAXB + b].end); }
    }
}
static double us(unsigned long long a, unsigned long long b) { return ((double)a - (double)b) * 0.01; }

static void launch_work(hipStream_t s, int idx, int ticks, int blocks, bool anyorder) {
    if (anyorder) hipExtLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d_st, idx, ticks);
    else work_kernel<<<blocks, 256, 0, s>>>(d_st, idx, ticks);
    CK(hipGetLastError());
}

static void report_pair(const char* name) {
    read_stamps();
    printf("%-11s: A ran %.2f us, B ran %.2f us, B starts %+.2f us after A ends (A start -> B end %.2f us)\n", name,
           us(h_st[0].end, h_st[0].start), us(h_st[1].end, h_st[1].start), us(h_st[1].start, h_st[0].end), us(h_st[1].end, h_st[0].start));
}

int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 1024;
    int ticks = argc > 2 ? atoi(argv[2]) : 500;      // 5 us
    hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipMalloc(&d_st, sizeof(Stamp) * 64 * MAXB));
    unsigned *cnt, *err; CK(hipMalloc(&cnt, 64 * 64)); CK(hipMalloc(&err, 4)); CK(hipMemset(cnt, 0, 64 * 64)); CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1, ef; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    printf("blocks %d, body %.1f us\n", blocks, ticks * 0.01);
    // warm
    launch_work(s, 2, 10, blocks, false); launch_work(s, 2, 10, blocks, true); CK(hipDeviceSynchronize());

    for (int rep = 0; rep < 2; ++rep) {
        reset_stamps(s); launch_work(s, 0, ticks, blocks, false); launch_work(s, 1, ticks, blocks, false); report_pair("plain");
        reset_stamps(s); launch_work(s, 0, ticks, blocks, true); launch_work(s, 1, ticks, blocks, true); report_pair("anyorder");
        {
            reset_stamps(s);
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            launch_work(s, 0, ticks, blocks, true); launch_work(s, 1, ticks, blocks, true);
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s)); report_pair("anyorder-g");
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        {
            reset_stamps(s);
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0));
            launch_work(s, 0, ticks, blocks, false); launch_work(s2, 1, ticks, blocks, false);
            CK(hipEventRecord(ef, s2)); CK(hipStreamWaitEvent(s, ef, 0));
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, s)); report_pair("fork-g");
            reset_stamps(s); CK(hipGraphLaunch(ge, s)); report_pair("fork-g (2)");
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        reset_stamps(s); launch_work(s, 0, ticks, blocks, false); launch_work(s2, 1, ticks, blocks, false); report_pair("2streams");
    }

    // ---- chains of 20 ----
    const int N = 20;
    for (int body : {0, 300}) {
        for (int mode = 0; mode < 7; ++mode) {
            // 0 plain dependent launches (hardware boundary), 1 counter chain plain launches (both costs),
            // 2 counter chain any-order, 3 counter chain alternating over two streams (eager), 4 the same inside one graph
            const char* names[] = {"hw boundary", "counter+hw", "counter anyorder", "counter 2 streams", "counter 2-branch graph", "hw boundary (graph)", "counter+hw (graph)"};
            unsigned epoch = 0;
            auto enqueue = [&](unsigned ep) {
                for (int i = 0; i < N; ++i) {
                    hipStream_t q = ((mode == 3 || mode == 4) && (i & 1)) ? s2 : s;
                    if (mode == 0 || mode == 5) work_kernel<<<blocks, 256, 0, q>>>(d_st, 8 + i, body);
                    else if (mode == 2) hipExtLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(256), 0, q, nullptr, nullptr, hipExtAnyOrderLaunch, cnt + 16 * 0, i, ep * (unsigned)blocks, body, d_st + 8 * MAXB, err);
                    else chain_kernel<<<blocks, 256, 0, q>>>(cnt, i, ep * (unsigned)blocks, body, d_st + 8 * MAXB, err);
                }
            };
            float best = 1e9f; double span = 0;
            if (mode >= 4) {
                // the epoch is a launch argument: one graph per epoch would defeat the purpose, so the graph leg uses
                // a fresh counter block per replay instead (memset node in front)
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                CK(hipMemsetAsync(cnt, 0, 64 * 64, s));
                if (mode == 4) { CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0)); }
                enqueue(1);
                if (mode == 4) { CK(hipEventRecord(ef, s2)); CK(hipStreamWaitEvent(s, ef, 0)); }
                CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int r = 0; r < 6; ++r) {
                    reset_stamps(s);
                    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
                    read_stamps(); span = us(h_st[8 + N - 1].end, h_st[8].start);
                    if (r == 1 && body == 0 && mode == 4) { printf("  graph timeline (us from kernel 0's start):"); for (int i = 0; i < N; ++i) printf(" [%.1f %.1f]", us(h_st[8 + i].start, h_st[8].start), us(h_st[8 + i].end, h_st[8].start)); printf("\n"); }
                }
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            } else {
                for (int r = 0; r < 6; ++r) {
                    CK(hipMemset(cnt, 0, 64 * 64)); reset_stamps(s); CK(hipDeviceSynchronize());
                    epoch = 1;
                    CK(hipEventRecord(e0, s));
                    if (mode == 3) { CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0)); }
                    enqueue(epoch);
                    if (mode == 3) { CK(hipEventRecord(ef, s2)); CK(hipStreamWaitEvent(s, ef, 0)); }
                    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s)); CK(hipDeviceSynchronize());
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
                    read_stamps(); span = us(h_st[8 + N - 1].end, h_st[8].start);
                }
            }
            unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            printf("chain body %.1f us  %-22s: events %.2f us per launch, device span %.2f us per launch  (timeouts %u)\n",
                   body * 0.01, names[mode], best * 1e3f / N, span / N, herr);
        }
    }
    return 0;
}
