// Third isolating experiment for the HIP-graph replay question.  graph_sync.hip: every wait covers a
// replayed graph in time.  graph_coherence.hip: an eager kernel behind a replay sees the graph's writes.  Neither covers the two
// directions the product test actually exercises between replays:
//   IN    an EAGER write to a graph input (torch: static_x.copy_(x), a device-to-device copy on the launch stream) followed by a
//         replay whose first node reads that input -- while the L2 of every XCD still holds the lines the PREVIOUS replay read;
//   MID   a buffer written by graph node P and read by graph node C (the plan's arena) that an eager step in between has
//         rewritten with other data and pulled into every L2 (the eager step of the product test uses the same arena offsets).
// One graph [P: buf <- src ; C: out[wg] <- (min, max) over ALL of buf], captured once on a side stream (as torch.cuda.CUDAGraph
// does), replayed `iters` times on the launch stream with a new value in src each time; per iteration
//   eager : stage <- v (kernel); src <- stage (hipMemcpyAsync D2D, what Tensor.copy_ issues)
//   replay
//   [between: nothing | hipStreamSynchronize | hipDeviceSynchronize | event record + stream wait]
//   eager : verify(out == (v, v)) accumulates mismatching workgroups in a device counter (no device-to-host copy inside the loop:
//           a blocking copy is itself one of the "waits" under test)
//   eager : the in-between step: buf <- OTHER, then a reader that pulls every line of buf and of src into every XCD's L2
// No torch, no product kernels.   hipcc --offload-arch=gfx950 -O2 -o scripts/micro/graph_input_coherence scripts/micro/graph_input_coherence.hip
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
__global__ void copyk(const int* src, int* dst, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
// every workgroup reads the whole buffer: out[2 wg] = min, out[2 wg + 1] = max
__global__ void minmax(const int* buf, int n, int* out) {
    __shared__ int smin[256], smax[256];
    int lo = 0x7fffffff, hi = -0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int v = buf[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smin[threadIdx.x] = min(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = max(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = smin[0];
        out[2 * blockIdx.x + 1] = smax[0];
    }
}
// bad[0] += workgroups whose (min, max) != (v, v); bad[1] += those that saw the PREVIOUS value of src, bad[2] += those that saw OTHER
__global__ void verify(const int* out, int nwg, int v, int prev, int other, unsigned* bad) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwg) return;
    const int lo = out[2 * w], hi = out[2 * w + 1];
    if (lo != v || hi != v) {
        atomicAdd(bad, 1u);
        if (lo == prev || hi == prev) atomicAdd(bad + 1, 1u);
        if (lo == other || hi == other) atomicAdd(bad + 2, 1u);
    }
}
__global__ void sink(const int* a, const int* b, int n, int* out) {   // reads every line of a and b from every workgroup
    int acc = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += a[i] ^ b[i];
    if (acc == 0x12345678) out[0] = acc;
}

enum Between { NOTHING, STREAM_SYNC, DEVICE_SYNC, EVENT_FENCE, N_BETWEEN };
static const char* between_name[] = {"nothing", "hipStreamSynchronize", "hipDeviceSynchronize", "event record + stream wait"};

int main(int argc, char** argv) {
    const int n = (argc > 1 ? atoi(argv[1]) : 384) * 1024;  // words (default 1.5 MiB = the product test's static input)
    const int iters = argc > 2 ? atoi(argv[2]) : 8;
    const int NWG = 512;
    int *stage, *src, *buf, *out, *dump;
    unsigned* bad;
    CK(hipMalloc((void**)&stage, (size_t)n * 4));
    CK(hipMalloc((void**)&src, (size_t)n * 4));
    CK(hipMalloc((void**)&buf, (size_t)n * 4));
    CK(hipMalloc((void**)&out, NWG * 2 * 4));
    CK(hipMalloc((void**)&dump, 64));
    CK(hipMalloc((void**)&bad, 16));
    hipStream_t launch[2] = {nullptr, nullptr};
    hipStream_t side;
    CK(hipStreamCreateWithFlags(&launch[1], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    const char* sname[2] = {"NULL stream", "non-blocking stream"};
    int total_bad = 0;
    for (int mode = 0; mode < 2; ++mode) {   // 0: graph replay, 1: the same two kernels launched eagerly (control)
        for (int si = 0; si < 2; ++si) {
            hipStream_t st = launch[si];
            for (int bw = 0; bw < N_BETWEEN; ++bw) {
                hipGraph_t graph = nullptr;
                hipGraphExec_t exec = nullptr;
                CK(hipDeviceSynchronize());
                CK(hipStreamBeginCapture(side, hipStreamCaptureModeThreadLocal));
                hipLaunchKernelGGL(copyk, dim3(256), dim3(256), 0, side, src, buf, n);
                hipLaunchKernelGGL(minmax, dim3(NWG), dim3(256), 0, side, buf, n, out);
                CK(hipStreamEndCapture(side, &graph));
                CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                CK(hipMemsetAsync(bad, 0, 16, st));
                const int OTHER = 77;
                int prev = -1;
                for (int it = 0; it < iters; ++it) {
                    const int v = 1000 * (bw + 1) + it;
                    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, st, stage, n, v);
                    CK(hipMemcpyAsync(src, stage, (size_t)n * 4, hipMemcpyDeviceToDevice, st));   // Tensor.copy_
                    if (mode == 0) {
                        CK(hipGraphLaunch(exec, st));
                    } else {
                        hipLaunchKernelGGL(copyk, dim3(256), dim3(256), 0, st, src, buf, n);
                        hipLaunchKernelGGL(minmax, dim3(NWG), dim3(256), 0, st, buf, n, out);
                    }
                    if (bw == STREAM_SYNC) CK(hipStreamSynchronize(st));
                    if (bw == DEVICE_SYNC) CK(hipDeviceSynchronize());
                    if (bw == EVENT_FENCE) {
                        hipEvent_t ev;
                        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                        CK(hipEventRecord(ev, st));
                        CK(hipStreamWaitEvent(st, ev, 0));
                        CK(hipEventDestroy(ev));
                    }
                    hipLaunchKernelGGL(verify, dim3((NWG + 255) / 256), dim3(256), 0, st, out, NWG, v, prev, OTHER, bad);
                    // the eager step in between: same intermediate buffer, other data; every L2 ends up holding buf (OTHER) and src (v)
                    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, st, buf, n, OTHER);
                    hipLaunchKernelGGL(sink, dim3(NWG), dim3(256), 0, st, buf, src, n, dump);
                    prev = v;
                }
                unsigned h[4] = {0, 0, 0, 0};
                CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
                CK(hipDeviceSynchronize());
                total_bad += mode == 0 ? (int)h[0] : 0;
                printf("%-12s on the %-20s between replay and verify: %-28s -> %u of %d workgroup-results wrong (%u saw the previous input, %u the in-between step's data)\n",
                       mode == 0 ? "graph replay" : "eager pair", sname[si], between_name[bw], h[0], iters * NWG, h[1], h[2]);
                CK(hipGraphExecDestroy(exec));
                CK(hipGraphDestroy(graph));
            }
        }
    }
    printf("%s\n", total_bad ? "STALE READS inside / in front of a replayed graph" : "a replayed graph reads what eager work wrote in front of it, and its nodes read each other's writes");
    return 0;
}
