// Microbenchmark: MFMA + LDS fragment reads only (no DMA, no global traffic), two wave-tile shapes on one CU-filling workgroup:
//   A: 8 waves (2 per SIMD), wave tile 64 ch x 128 px (MT=2, NT=4): per tap 6 ds_read_b128, 8 MFMA   (the ring kernel's shape)
//   B: 4 waves (1 per SIMD), wave tile 128 ch x 128 px (MT=4, NT=4): per tap 8 ds_read_b128, 16 MFMA (0.5 KB of LDS per MFMA)
// build: hipcc -O3 --offload-arch=gfx950 -o scripts/micro/wave_tile scripts/micro/wave_tile.hip ; run: scripts/micro/wave_tile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ int g_random;

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_addr))
                 : "memory");
}

// DMA: 0 none; 1 = 8 one-KiB LDS-DMA pieces per wave and stage (the 16x32 ring kernel's 57 KiB per stage) from an L2-resident
// buffer, issued one behind each MFMA group, `s_waitcnt vmcnt(0)` before the stage barrier; 2 = the same from a buffer too
// large for L2 (HBM stream)
template <int WAVES, int MT, int NT, int DMA = 0>
__global__ void __launch_bounds__(WAVES * 64, 1) k(float* out, int stages, const unsigned char* gbuf = nullptr, size_t gmask = 0) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[120 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // RANDOM_DATA: operands with random mantissas and signs (bf16 in [-2, 2)) instead of near-constant ones: the MFMA clock the
    // chip sustains depends on how much the datapath toggles
    for (int i = tid; i < 120 * 1024 / 4; i += WAVES * 64) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        reinterpret_cast<unsigned*>(lds)[i] = g_random ? ((h & 0x80ff80ffu) | 0x3f003f00u) : 0x3c003c00u + (i & 7);
    }
    __syncthreads();
    f32x16 acc[MT][NT];
    for (int m = 0; m < MT; ++m)
        for (int n = 0; n < NT; ++n)
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const unsigned char* pa = lds + (wave * MT * 32 + (lane & 31)) * 16 + (lane >> 5) * 2048;   // filter-like image
    const unsigned char* pb = lds + 64 * 1024 + ((wave * NT) % 16 * 32 + (lane & 31)) * 16 + (lane >> 5) * 10240;
    const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds;
    size_t goff = ((size_t)blockIdx.x * 57 * 1024 * 7 + (size_t)wave * 1024 + lane * 16);
    for (int s = 0; s < stages; ++s) {
        bf16x8 af[2][MT], bf[2][NT];
        auto load = [&](int t, bf16x8 (&fa)[MT], bf16x8 (&fb)[NT]) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const bf16x8*>(pa + t * 4096 + m * 512);
#pragma unroll
            for (int n = 0; n < NT; ++n) fb[n] = *reinterpret_cast<const bf16x8*>(pb + (t / 3 * 34 + t % 3) * 16 + n * 512);
        };
        load(0, af[0], bf[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) load(t + 1, af[(t + 1) & 1], bf[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t & 1][m], bf[t & 1][n], acc[m][n], 0, 0, 0);
            __builtin_amdgcn_s_setprio(2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DMA != 0) {
                if (t < 8 * 8 / WAVES) {  // 64 pieces per stage over the waves
                    // destination: the half of LDS the MFMAs do not read in this stage would be the ring's other slot; here a
                    // 56-KiB window above the fragment images (the data is never read)
                    // DMA 3/4: pieces 20.. of a stage (the filter slab) come from the SAME 36 KiB in every workgroup, as in the
                    // real kernel (all CUs stream the same filters at the same time)
                    const int q = t * WAVES + wave;
                    const size_t src = (DMA >= 3 && q >= 20) ? (size_t)((s & 15) * 36 + (q - 20)) * 1024 + lane * 16 : (goff & gmask);
                    dma16(gbuf + src, lds_base + 62 * 1024 + (q % 56) * 1024);
                    goff += WAVES * 1024;
                }
            }
        }
        if constexpr (DMA == 4) {  // every 8th stage: an "epilogue" of 16 one-KiB stores per wave (128 KiB per workgroup)
            if ((s & 7) == 7) {
                uint4* o = reinterpret_cast<uint4*>(const_cast<unsigned char*>(gbuf) + ((size_t)1 << 30) +
                                                    (((size_t)blockIdx.x * 4000 + s) * 128 * 1024 & (((size_t)2 << 30) - 1))) + wave * 16 * 64 + lane;
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i * 64] = make_uint4(s, i, lane, wave);
            }
        }
        if constexpr (DMA != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float sum = 0.f;
    for (int m = 0; m < MT; ++m)
        for (int n = 0; n < NT; ++n)
            for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
    if (sum == 123.456f) out[tid] = sum;
}

template <int WAVES, int MT, int NT, int DMA = 0>
void run(const char* name, float* out, const unsigned char* gbuf = nullptr, size_t gmask = 0) {
    const int stages = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<WAVES, MT, NT, DMA>), dim3(256), dim3(WAVES * 64), 0, 0, out, stages, gbuf, gmask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 256.0 * WAVES * stages * 9 * MT * NT * 32768.0;
        printf("%s: %.2f ms  %.0f TFLOP/s  (%.2f us per stage of %d MFMAs/wave)\n", name, ms, flops / ms / 1e9, ms * 1e3 / stages, 9 * MT * NT);
    }
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 20);
    run<8, 2, 4>("A 8 waves 64x128 ", out);
    int one = 1;
    hipMemcpyToSymbol(HIP_SYMBOL(g_random), &one, sizeof(int));
    run<8, 2, 4>("A' = A with random operands", out);
    run<8, 2, 2>("C' = C with random operands", out);
    run<4, 4, 4>("B' = B with random operands", out);
    int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_random), &zero, sizeof(int));
    run<4, 4, 4>("B 4 waves 128x128", out);
    run<8, 2, 2>("C 8 waves 64x64  ", out);
    unsigned char* g;
    const size_t big = (size_t)4 << 30;
    hipMalloc(&g, big);
    hipMemset(g, 0, big);
    run<8, 2, 4, 1>("D = A + 57 KiB/stage LDS-DMA, L2-hot (2 MiB)", out, g, ((size_t)2 << 20) - 1);
    run<8, 2, 4, 2>("E = A + 57 KiB/stage LDS-DMA, HBM (4 GiB)  ", out, g, big - 1);
    run<4, 4, 4, 1>("F = B + 57 KiB/stage LDS-DMA, L2-hot       ", out, g, ((size_t)2 << 20) - 1);
    run<8, 2, 4, 3>("G = A + DMA: 20 KiB HBM + 36 KiB shared filters", out, g, ((size_t)1 << 30) - 1);
    run<8, 2, 4, 4>("H = G + 128 KiB of stores every 8 stages    ", out, g, ((size_t)1 << 30) - 1);
    hipMemcpyToSymbol(HIP_SYMBOL(g_random), &one, sizeof(int));
    run<8, 2, 4, 4>("H' = H with random operands                 ", out, g, ((size_t)1 << 30) - 1);
    return 0;
}
