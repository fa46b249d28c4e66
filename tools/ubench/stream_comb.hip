// Microbenchmark (not part of the library): does the ORDER in which persistent workgroups walk memory matter at HBM
// sizes?  Same traffic as the fused-mask kernel (48 B in, 16 B out per thread and step, 512-thread workgroups,
// two per CU), 1024 frames of 640x480 (944 MB in, 315 MB out: beyond the Infinity Cache).
//   sweep: workgroup b handles chunks b, b + G, b + 2G, ...  (all workgroups read one contiguous window at a time)
//   comb : workgroup b streams its own contiguous region start to end (the fused kernel's segments: G regions
//          walked in parallel, 1/G of the buffer apart)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int COMB, int T>
__global__ __launch_bounds__(T) void k(const u4* __restrict__ in, u4* __restrict__ out, size_t nchunks)
{
    // chunk = T groups of 16 px
    const size_t per = (nchunks + gridDim.x - 1) / gridDim.x;
    for (size_t it = 0; it < per; ++it) {
        const size_t c = COMB ? (size_t)blockIdx.x * per + it : it * gridDim.x + blockIdx.x;
        if (c >= nchunks) break;
        const size_t g = c * T + threadIdx.x;
        const u4* p = in + g * 3;
        u4 a = p[0], b = p[1], d = p[2];
        u4 o;
        o.x = a.x ^ b.x ^ d.x; o.y = a.y ^ b.y ^ d.y; o.z = a.z ^ b.z ^ d.z; o.w = a.w ^ b.w ^ d.w;
        __builtin_nontemporal_store(o, out + g);
    }
}

int main(int argc, char** argv)
{
    const int nfr = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t px = (size_t)nfr * 640 * 480, ngroups = px / 16;
    printf("frames %d  (%.0f MB in, %.0f MB out)\n", nfr, px * 3 / 1e6, px / 1e6);
    u4 *in, *out;
    CK(hipMalloc(&in, px * 3)); CK(hipMalloc(&out, px));
    CK(hipMemset(in, 1, px * 3));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {256, 512, 1024, 2048}) {
        for (int v = 0; v < 4; ++v) {
            float best = 1e9, sum = 0;
            int cnt = 0;
            for (int it = 0; it < 12; ++it) {
                CK(hipEventRecord(e0));
                if (v == 0) k<0, 512><<<blocks, 512>>>(in, out, ngroups / 512);
                if (v == 1) k<1, 512><<<blocks, 512>>>(in, out, ngroups / 512);
                if (v == 2) k<0, 1024><<<blocks, 1024>>>(in, out, ngroups / 1024);
                if (v == 3) k<1, 1024><<<blocks, 1024>>>(in, out, ngroups / 1024);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2) { if (ms < best) best = ms; sum += ms; ++cnt; }
            }
            printf("blocks %5d %s T=%4d: best %.4f ms %.0f GB/s   avg %.4f ms %.0f GB/s\n", blocks, (v & 1) ? "comb " : "sweep", v < 2 ? 512 : 1024,
                   best, px * 4.0 / best / 1e6, sum / cnt, px * 4.0 / (sum / cnt) / 1e6);
        }
    }
    return 0;
}
