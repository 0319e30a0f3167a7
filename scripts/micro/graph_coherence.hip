// Second isolating experiment for the HIP-graph replay question: graph_sync.hip showed that every wait
// COVERS a replayed graph in time.  This one asks whether the graph's WRITES are visible to an ordinary kernel launched behind
// the replay: MI355X has 8 XCDs with one L2 each, coherent with each other only through the release / acquire fences the
// runtime puts into its dispatch packets.  No torch, no product kernels.
//   warm : eager fill(buf, OLD) + an eager reader that pulls every line of buf into the L2 of every XCD
//   graph: fill(buf, NEW), captured once, replayed
//   probe: eager check(buf) launched right behind the replay on the same stream: every workgroup reads ALL of buf and counts the
//          words that are not NEW
// in four variants of what lies between replay and probe: nothing, hipStreamSynchronize, hipDeviceSynchronize, an event recorded on
// the stream and waited for by the same stream (the fence utils.graph_replay uses) -- and an all-eager control.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/micro/graph_coherence scripts/micro/graph_coherence.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

__global__ void fill(int* buf, int n, int v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = v;
}
// every workgroup reads the whole buffer (so every XCD's L2 holds / is asked for every line); words != want are counted
__global__ void check(const int* buf, int n, int want, unsigned* bad) {
    unsigned local = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) local += buf[i] != want;
    if (local) atomicAdd(bad, local);
}

enum Between { NOTHING, STREAM_SYNC, DEVICE_SYNC, EVENT_FENCE, N_BETWEEN };
static const char* between_name[] = {"nothing", "hipStreamSynchronize", "hipDeviceSynchronize", "event record + stream wait"};

int main(int argc, char** argv) {
    const int n = (argc > 1 ? atoi(argv[1]) : 256) * 1024;  // words (default 1 MiB: stays resident in a 4 MiB L2)
    const int reps = argc > 2 ? atoi(argv[2]) : 8;
    int* buf;
    unsigned* bad;
    CK(hipMalloc((void**)&buf, (size_t)n * 4));
    CK(hipMalloc((void**)&bad, 4));
    hipStream_t launch[2] = {nullptr, nullptr};
    hipStream_t side;
    CK(hipStreamCreateWithFlags(&launch[1], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    const char* sname[2] = {"NULL stream", "non-blocking stream"};
    int total_bad_runs = 0;
    for (int mode = 0; mode < 2; ++mode) {        // 0: graph replay, 1: all-eager control
        for (int si = 0; si < 2; ++si) {
            hipStream_t st = launch[si];
            for (int bw = 0; bw < N_BETWEEN; ++bw) {
                int bad_runs = 0;
                unsigned worst = 0;
                for (int rep = 0; rep < reps; ++rep) {
                    const int OLD = 1000 + rep, NEW = 5000 + rep;
                    hipGraph_t graph = nullptr;
                    hipGraphExec_t exec = nullptr;
                    if (mode == 0) {   // captured on a side stream, as torch.cuda.CUDAGraph does
                        CK(hipStreamBeginCapture(side, hipStreamCaptureModeThreadLocal));
                        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, side, buf, n, NEW);
                        CK(hipStreamEndCapture(side, &graph));
                        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                    }
                    CK(hipMemsetAsync(bad, 0, 4, st));
                    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, st, buf, n, OLD);
                    hipLaunchKernelGGL(check, dim3(512), dim3(256), 0, st, buf, n, OLD, bad);   // warm every L2 with the OLD lines
                    CK(hipMemsetAsync(bad, 0, 4, st));
                    if (mode == 0)
                        CK(hipGraphLaunch(exec, st));
                    else
                        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, st, buf, n, NEW);
                    if (bw == STREAM_SYNC) CK(hipStreamSynchronize(st));
                    if (bw == DEVICE_SYNC) CK(hipDeviceSynchronize());
                    if (bw == EVENT_FENCE) {
                        hipEvent_t ev;
                        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                        CK(hipEventRecord(ev, st));
                        CK(hipStreamWaitEvent(st, ev, 0));
                        CK(hipEventDestroy(ev));
                    }
                    hipLaunchKernelGGL(check, dim3(512), dim3(256), 0, st, buf, n, NEW, bad);
                    unsigned h = 0;
                    CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
                    CK(hipDeviceSynchronize());
                    bad_runs += h != 0;
                    worst = h > worst ? h : worst;
                    if (exec) CK(hipGraphExecDestroy(exec));
                    if (graph) CK(hipGraphDestroy(graph));
                }
                total_bad_runs += mode == 0 ? bad_runs : 0;
                printf("%-12s on the %-20s between replay and probe: %-28s -> %d of %d runs read stale words (worst %u word-reads)\n",
                       mode == 0 ? "graph replay" : "eager fill", sname[si], between_name[bw], bad_runs, reps, worst);
            }
        }
    }
    printf("%s\n", total_bad_runs ? "STALE READS behind a replayed graph" : "a kernel launched behind a replayed graph sees all of its writes");
    return 0;
}
