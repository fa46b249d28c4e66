// Verifies the lane layout assumed for v_mfma_i32_32x32x32_i8 on gfx950 with exact integers:
//   A: lane l holds A[m = l & 31][k = 16 * (l >> 5) + j], j = 0..15 (16 x i8 in 4 VGPRs)
//   B: lane l holds B[k = 16 * (l >> 5) + j][n = l & 31]
//   D: lane l, reg r: D[m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][n = l & 31]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const signed char* A, const signed char* B, int* D)
{
    const int l = threadIdx.x, m = l & 31, h = l >> 5;
    union { i32x4 v; signed char b[16]; } a, b;
    for (int j = 0; j < 16; ++j) { a.b[j] = A[m * 32 + 16 * h + j]; b.b[j] = B[(16 * h + j) * 32 + m]; }
    i32x16 c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + m] = c[r];
}

int main()
{
    signed char hA[32 * 32], hB[32 * 32];
    int hD[32 * 32], ref[32 * 32];
    srand(5);
    for (int i = 0; i < 1024; ++i) { hA[i] = (signed char)(rand() % 256 - 128); hB[i] = (signed char)(rand() % 256 - 128); }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { int s = 0; for (int k = 0; k < 32; ++k) s += hA[m * 32 + k] * hB[k * 32 + n]; ref[m * 32 + n] = s; }
    signed char *dA, *dB; int* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("mfma_i32_32x32x32_i8 layout check: %d mismatches of 1024 (D[0][0]=%d ref=%d, D[5][7]=%d ref=%d)\n", bad, hD[0], ref[0], hD[5*32+7], ref[5*32+7]);
    return bad != 0;
}
