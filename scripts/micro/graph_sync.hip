// Isolating experiment for the HIP-graph replay ordering question (VERDICT r2 task 7).
// No torch, no product kernels: ONE long-running kernel that sets a host-visible flag as its LAST action is captured into a
// graph; after hipGraphLaunch the host waits in one of several ways and reads the flag.  If a wait returns while the flag is
// still 0, that wait does not cover a replayed graph on this runtime.  A second probe launches an ordinary kernel into the same
// stream right behind the replay and lets IT record whether the flag was already set when it started (stream ordering of later
// launches).  Streams: the null stream, a blocking and a non-blocking created stream; capture stream == launch stream and
// capture on a side stream / launch on another (what torch.cuda.CUDAGraph does).
//
//   hipcc --offload-arch=gfx950 -O2 -o scripts/micro/graph_sync scripts/micro/graph_sync.hip && scripts/micro/graph_sync
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                     \
        }                                                                                \
    } while (0)

// spins `ticks` of the 100 MHz wall clock on one lane, then publishes flag = value (system scope)
__global__ void slow_then_flag(volatile int* flag, unsigned long long ticks, int value) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) {
        }
        __threadfence_system();
        *flag = value;
        __threadfence_system();
    }
}
// an ordinary launch behind the replay: what did the flag read when this kernel STARTED?
__global__ void probe(volatile int* flag, int* seen) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *seen = *flag;
}

enum Wait { STREAM_SYNC, DEVICE_SYNC, EVENT_SYNC, NEXT_KERNEL, N_WAIT };
static const char* wait_name[] = {"hipStreamSynchronize", "hipDeviceSynchronize", "event record+sync", "next kernel on the stream"};

int main(int argc, char** argv) {
    const double ms = argc > 1 ? atof(argv[1]) : 40.0;   // duration of the captured kernel
    const int nodes = argc > 2 ? atoi(argv[2]) : 3;      // kernel nodes in the graph (a chain); the LAST sets the flag
    const unsigned long long ticks = (unsigned long long)(ms * 1e5);
    int* flag = nullptr;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    int* seen_d = nullptr;
    CK(hipMalloc((void**)&seen_d, 64));
    hipStream_t blocking, nonblocking, side;
    CK(hipStreamCreate(&blocking));
    CK(hipStreamCreateWithFlags(&nonblocking, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    struct Case { const char* name; hipStream_t capture, launch; } cases[] = {
        {"capture+launch on a blocking stream", blocking, blocking},
        {"capture+launch on a non-blocking stream", nonblocking, nonblocking},
        {"capture on a side stream, launch on the NULL stream (torch.cuda.CUDAGraph + default stream)", side, nullptr},
        {"capture on a side stream, launch on a blocking stream", side, blocking},
        {"capture on a side stream, launch on a non-blocking stream", side, nonblocking},
    };
    int failures = 0;
    printf("graph of %d kernel node(s), the last one spins %.0f ms and then sets the flag\n", nodes, ms);
    for (const Case& c : cases) {
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(c.capture, hipStreamCaptureModeThreadLocal));
        for (int n = 0; n < nodes; ++n)  // the earlier nodes write 0 at once, only the last one publishes 1 after the spin
            hipLaunchKernelGGL(slow_then_flag, dim3(1), dim3(64), 0, c.capture, flag, n + 1 == nodes ? ticks : 0ull, n + 1 == nodes ? 1 : 0);
        CK(hipStreamEndCapture(c.capture, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        printf("%s\n", c.name);
        for (int w = 0; w < N_WAIT; ++w) {
            for (int rep = 0; rep < 3; ++rep) {
                *flag = 0;
                CK(hipMemset(seen_d, 0xff, 4));
                CK(hipDeviceSynchronize());
                CK(hipGraphLaunch(exec, c.launch));
                int observed = -1;
                if (w == STREAM_SYNC) {
                    CK(hipStreamSynchronize(c.launch));
                    observed = *flag;
                } else if (w == DEVICE_SYNC) {
                    CK(hipDeviceSynchronize());
                    observed = *flag;
                } else if (w == EVENT_SYNC) {
                    hipEvent_t ev;
                    CK(hipEventCreate(&ev));
                    CK(hipEventRecord(ev, c.launch));
                    CK(hipEventSynchronize(ev));
                    observed = *flag;
                    CK(hipEventDestroy(ev));
                } else {
                    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, c.launch, flag, seen_d);
                    hipEvent_t ev;   // (an event wait is used only to collect the probe's answer)
                    CK(hipEventCreate(&ev));
                    CK(hipEventRecord(ev, c.launch));
                    CK(hipEventSynchronize(ev));
                    CK(hipEventDestroy(ev));
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(&observed, seen_d, 4, hipMemcpyDeviceToHost));
                }
                const bool ok = observed == 1;
                failures += !ok;
                printf("    %-28s rep %d: flag %s\n", wait_name[w], rep, ok ? "SET (the wait covered the replay)" : "NOT SET (returned / started before the graph's last node finished)");
                CK(hipDeviceSynchronize());
            }
        }
        CK(hipGraphExecDestroy(exec));
        CK(hipGraphDestroy(graph));
    }
    printf("%s: %d of %d observations saw the flag unset\n", failures ? "ORDERING VIOLATIONS" : "all waits cover a replayed graph", failures,
           (int)(sizeof(cases) / sizeof(cases[0])) * N_WAIT * 3);
    return 0;
}
