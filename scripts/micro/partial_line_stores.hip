// How fast are plain stores whose wave-instructions cover only PART of each 128-byte line?  (the stride-2 data gradient writes one
// pixel-parity class per workgroup: 32-byte segments at a 64-byte stride, the other class's workgroup fills the gaps later)
//   mode 0: 16 B per lane, 1 KiB contiguous per wave-instruction (the forward kernels' stores)
//   mode 1: 32-byte segments at a 64-byte stride per wave-instruction; the two parities written by DIFFERENT workgroups that are
//           adjacent in the grid (roughly concurrent, same XCD range)
//   mode 2: the same segments, both parities written by the SAME wave in two consecutive instructions
//   mode 3: as mode 1 but the second parity only in a second launch (never merges in L2)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void __launch_bounds__(256) st(uint4* out, size_t n_units, int mode, int parity_launch) {
    // unit = 16 bytes; a "pixel" = 2 units (32 B); pixels alternate between parity 0 and 1
    const size_t wg = blockIdx.x, tid = threadIdx.x;
    const uint4 v = make_uint4(1, 2, 3, (unsigned)tid);
    if (mode == 0) {
        for (size_t u = wg * 256 + tid; u < n_units; u += (size_t)gridDim.x * 256) out[u] = v;
    } else if (mode == 2) {
        // thread -> (pixel pair q, half hh): writes pixel 2q (parity 0) then pixel 2q+1 (parity 1)
        for (size_t t = wg * 256 + tid; t < n_units / 2; t += (size_t)gridDim.x * 256) {
            const size_t q = t >> 1, hh = t & 1;
            out[(2 * q) * 2 + hh] = v;
            out[(2 * q + 1) * 2 + hh] = v;
        }
    } else {
        // workgroup pairs (2k, 2k+1) write parity 0 / 1 of the same pixel range
        const int par = mode == 3 ? parity_launch : (int)(wg & 1);
        const size_t pairs = mode == 3 ? gridDim.x : gridDim.x / 2, k = mode == 3 ? wg : wg >> 1;
        for (size_t t = k * 256 + tid; t < n_units / 2; t += pairs * 256) {
            const size_t q = t >> 1, hh = t & 1;
            out[(2 * q + par) * 2 + hh] = v;
        }
    }
}

int main() {
    const size_t bytes = (size_t)2 << 30, n_units = bytes / 16;
    uint4* buf;
    CK(hipMalloc((void**)&buf, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[] = {"contiguous 1 KiB per wave-instruction", "32-B segments, parities by neighbouring workgroups", "32-B segments, both parities by the same wave",
                           "32-B segments, parities in two launches"};
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (mode == 3) {
                hipLaunchKernelGGL(st, dim3(4096), dim3(256), 0, 0, buf, n_units, mode, 0);
                hipLaunchKernelGGL(st, dim3(4096), dim3(256), 0, 0, buf, n_units, mode, 1);
            } else {
                hipLaunchKernelGGL(st, dim3(8192), dim3(256), 0, 0, buf, n_units, mode, 0);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-56s %7.3f ms for 2 GiB = %5.2f TB/s\n", names[mode], best, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
