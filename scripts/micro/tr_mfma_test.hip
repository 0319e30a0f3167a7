// micro-test: ds_read_b64_tr_b16 + v_mfma_f32_16x16x32_bf16 with K = pixels on [pixel][16 ch] LDS tiles
// (the operand fetch of the wgrad kernel).  Exact small-integer data; asymmetric operands.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* a_tile, const uint16_t* b_tile, float* out) {
  __shared__ __attribute__((aligned(16))) uint16_t la[32 * 16], lb[32 * 16];
  const int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) { la[i] = a_tile[i]; lb[i] = b_tile[i]; }
  __syncthreads();
  const int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  s4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(la + (8 * g + q) * 16 + p * 4));
  s4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(la + (8 * g + 4 + q) * 16 + p * 4));
  s4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lb + (8 * g + q) * 16 + p * 4));
  s4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lb + (8 * g + 4 + q) * 16 + p * 4));
  short av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
  short bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  bf16x8 A, B;
  __builtin_memcpy(&A, av, 16); __builtin_memcpy(&B, bv, 16);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + (l & 15)] = c[r];
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
int main() {
  uint16_t ha[512], hb[512]; float fa[512], fb[512];
  for (int p = 0; p < 32; ++p) for (int c = 0; c < 16; ++c) {
    fa[p * 16 + c] = (float)((p * 3 + c * 5) % 7 - 3);       // a[pixel][co]
    fb[p * 16 + c] = (float)((p * 2 + c * c + 1) % 5 - 2);   // b[pixel][ci]
    ha[p * 16 + c] = f2bf(fa[p * 16 + c]); hb[p * 16 + c] = f2bf(fb[p * 16 + c]);
  }
  uint16_t *da, *db; float* dout;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 1024);
  hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
  float ho[256]; hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int co = 0; co < 16; ++co) for (int ci = 0; ci < 16; ++ci) {
    float ref = 0; for (int p = 0; p < 32; ++p) ref += fa[p * 16 + co] * fb[p * 16 + ci];
    if (ho[co * 16 + ci] != ref) { if (bad < 5) printf("mismatch co %d ci %d got %g want %g\n", co, ci, ho[co * 16 + ci], ref); ++bad; }
  }
  printf("tr_mfma_test: %d mismatches of 256 (D[co][ci] = sum_p a[p][co]*b[p][ci])\n", bad);
  return bad != 0;
}
