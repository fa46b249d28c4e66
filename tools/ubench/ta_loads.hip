// Microbenchmark (not part of the library): what do k_dials' window loads cost the texture-address path?
// 4096 waves (4 per SIMD, all resident, as the kernel), each fetching a 48 x 48 pixel window of packed 3-byte pixels out of a
// 640 x 480 frame:
//   A  one UNALIGNED DWORD per pixel and lane (lane = column, 48 rows -> 48 load instructions per wave)       [the kernel today]
//   B  DWORDX3 per lane = four pixels, 12 lanes per row, 5 rows per instruction -> 10 load instructions per wave
// Both sum what they load (so that the loads are kept) and write one dword per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u3 __attribute__((ext_vector_type(3)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const unsigned char* __restrict__ frames, size_t frame_stride, int row_stride, unsigned* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned char* org = frames + (size_t)blockIdx.x * frame_stride + (size_t)(200 + 7 * wv) * row_stride + (size_t)(60 + 37 * wv + (blockIdx.x & 7)) * 3;
    unsigned acc = 0;
    if (MODE == 0) {
        unsigned v[48];
#pragma unroll
        for (int r = 0; r < 48; ++r) { unsigned t; __builtin_memcpy(&t, org + (size_t)r * row_stride + (lane < 48 ? lane : 47) * 3, 4); v[r] = t; }
#pragma unroll
        for (int r = 0; r < 48; ++r) acc += v[r] & 0xffffffu;
    } else {
        const int rs = lane / 12 < 5 ? lane / 12 : 4, q = lane - 12 * (lane / 12);
        u3 v[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int row = 5 * j + rs < 48 ? 5 * j + rs : 47;
            u3 t;
            __builtin_memcpy(&t, org + (size_t)row * row_stride + q * 12, 12);
            v[j] = t;
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) acc += v[j].x + v[j].y + v[j].z;
    }
    acc += __shfl_xor(acc, 32); acc += __shfl_xor(acc, 16); acc += __shfl_xor(acc, 8); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 1);
    if (lane == 0) out[blockIdx.x * 4 + wv] = acc;
}

int main()
{
    const int n = 1024, H = 640, W = 480;
    const size_t fs = (size_t)H * W * 3;
    unsigned char* frames;
    unsigned* out;
    CK(hipMalloc(&frames, 4 * n * fs)); CK(hipMalloc(&out, n * 16));
    CK(hipMemset(frames, 7, 4 * n * fs));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            float sum = 0;
            for (int it = 0; it < 24; ++it) {
                const unsigned char* f = frames + (size_t)(it % 4) * n * fs;   // rotate over 3.7 GB: nothing stays in the caches
                CK(hipEventRecord(e0));
                if (mode == 0) k<0><<<n, 256>>>(f, fs, W * 3, out); else k<1><<<n, 256>>>(f, fs, W * 3, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 4) sum += ms;
            }
            printf("%s: %.2f us per launch of 4096 waves\n", mode == 0 ? "A 48 unaligned dword loads per wave (lane = column)" : "B 10 dwordx3 loads per wave (12 lanes per row)  ", sum / 20 * 1e3);
        }
    return 0;
}
