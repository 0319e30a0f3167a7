// Microbenchmark: the weight-gradient kernel's inner loop without any global traffic -- what do the LDS transposing reads cost?
// 8 waves, per "segment" every wave reads 4 dz fragments + 9 x fragments (two ds_read_b64_tr_b16 each) and issues 36
// v_mfma_f32_16x16x32_bf16.  MODE 0: as the kernel; 1: plain ds_read_b64 at the same addresses (wrong data, same bytes);
// 2: no LDS reads at all (MFMA only); 3: reads only (no MFMA).
// build: hipcc -O3 --offload-arch=gfx950 -o scripts/micro/wgrad_inner scripts/micro/wgrad_inner.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int BUF = 21 * 1024;

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * BUF / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u + i;
    __syncthreads();
    const int cw = wave / 4, iw = wave % 4;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    int aoff[2], boff[3][2];
    for (int s = 0; s < 2; ++s) {
        const int r = 8 * g + 4 * s + q;
        aoff[s] = ((cw * 4) * 32 + r) * 32 + p * 8;
        for (int kw = 0; kw < 3; ++kw) boff[kw][s] = 8192 + ((iw * 3) * 34 + r + kw) * 32 + p * 8;
    }
    f32x4 acc[4][9];
    for (int j = 0; j < 4; ++j)
        for (int t = 0; t < 9; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto rd = [&](const unsigned char* a0, const unsigned char* a1) __attribute__((always_inline)) {
        s16x4 lo, hi;
        if constexpr (MODE == 0 || MODE == 3) {
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
        } else {
            lo = *reinterpret_cast<const s16x4*>(a0);
            hi = *reinterpret_cast<const s16x4*>(a1);
        }
        const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bf16x8 f;
        __builtin_memcpy(&f, v, 16);
        return f;
    };
    bf16x8 keep[4] = {}, kb = {};
    for (int s = 0; s < iters; ++s) {
        const unsigned char* L = lds + (s & 1) * BUF;
        bf16x8 af[4];
        if constexpr (MODE != 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = rd(L + aoff[0] + j * 1024, L + aoff[1] + j * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = keep[j];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            bf16x8 bfr;
            if constexpr (MODE != 2)
                bfr = rd(L + boff[t % 3][0] + (t / 3) * 34 * 32, L + boff[t % 3][1] + (t / 3) * 34 * 32);
            else
                bfr = kb;
            if constexpr (MODE != 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bfr, acc[j][t], 0, 0, 0);
            } else {
                asm volatile("" ::"v"(bfr), "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3]));
            }
        }
    }
    float sum = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int t = 0; t < 9; ++t) sum += acc[j][t][0] + acc[j][t][3];
    out[blockIdx.x * 512 + tid] = sum;
}

template <int MODE>
static void run(const char* name, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 8 * iters * 36 * 16384.0;
    printf("%-34s %8.3f ms  %7.1f ns/segment  %7.1f TFLOP/s-equivalent\n", name, ms, ms * 1e6 / iters, flops / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    run<0>("tr reads + MFMA (kernel)", out, iters);
    run<1>("plain b64 reads + MFMA", out, iters);
    run<2>("MFMA only", out, iters);
    run<3>("tr reads only", out, iters);
    return 0;
}
