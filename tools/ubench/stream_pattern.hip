// Microbenchmark (not part of the library): does the LANE ACCESS PATTERN of a 3:1 read/write stream matter at HBM sizes?
// The fused-mask kernel reads 48 B per thread as three dwordx4 at a 48-byte lane stride (each 128-B line is touched by
// three instructions of the wave).  Variants, same bytes, same persistent "sweep" order:
//   strided : p[0], p[1], p[2] of a 48-byte-per-thread record                       (the kernel's pattern)
//   packed  : three lane-contiguous dwordx4 loads (each instruction reads 1 KiB contiguous per wave)
//   packed-nt: the same with non-temporal loads
//   copy    : float4 copy 1:1 (the guide's 6.3 TB/s reference)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int T>
__global__ __launch_bounds__(T) void k(const u4* __restrict__ in, u4* __restrict__ out, size_t nchunks)
{
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t g = c * T + threadIdx.x;
        u4 a, b, d;
        if (MODE == 0) { const u4* p = in + g * 3; a = p[0]; b = p[1]; d = p[2]; }
        else if (MODE == 1) { const u4* p = in + c * T * 3 + threadIdx.x; a = p[0]; b = p[T]; d = p[2 * T]; }
        else if (MODE == 2) { const u4* p = in + c * T * 3 + threadIdx.x; a = __builtin_nontemporal_load(p); b = __builtin_nontemporal_load(p + T); d = __builtin_nontemporal_load(p + 2 * T); }
        else { const u4* p = in + c * T * 2 + threadIdx.x; a = p[0]; b = p[T]; d = a; }
        u4 o;
        o.x = a.x ^ b.x ^ d.x; o.y = a.y ^ b.y ^ d.y; o.z = a.z ^ b.z ^ d.z; o.w = a.w ^ b.w ^ d.w;
        if (MODE == 3) { out[c * T * 2 + threadIdx.x] = a; out[c * T * 2 + T + threadIdx.x] = b; }
        else __builtin_nontemporal_store(o, out + g);
    }
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 3200;           // input megabytes
    const size_t ngroups = mb * 1000000 / 48, T = 1024, nchunks = ngroups / T;
    const size_t in_bytes = nchunks * T * 48, out_bytes = nchunks * T * 32;   // out sized for the copy variant
    u4 *in, *out;
    CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, out_bytes));
    CK(hipMemset(in, 1, in_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"strided  3:1", "packed   3:1", "packed-nt 3:1", "copy     1:1"};
    for (int blocks : {512, 1024, 2048}) {
        for (int v = 0; v < 4; ++v) {
            float best = 1e9, sum = 0;
            int cnt = 0;
            for (int it = 0; it < 8; ++it) {
                CK(hipEventRecord(e0));
                if (v == 0) k<0, 1024><<<blocks, 1024>>>(in, out, nchunks);
                if (v == 1) k<1, 1024><<<blocks, 1024>>>(in, out, nchunks);
                if (v == 2) k<2, 1024><<<blocks, 1024>>>(in, out, nchunks);
                if (v == 3) k<3, 1024><<<blocks, 1024>>>(in, out, nchunks);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2) { if (ms < best) best = ms; sum += ms; ++cnt; }
            }
            const double bytes = v == 3 ? (double)nchunks * T * 64 : (double)nchunks * T * 64;   // 48 in + 16 out, or 32 + 32
            printf("%4zu MB in, blocks %5d %-14s: best %.4f ms %.0f GB/s   avg %.4f ms %.0f GB/s\n", mb, blocks, names[v], best, bytes / best / 1e6,
                   sum / cnt, bytes / (sum / cnt) / 1e6);
        }
    }
    return 0;
}
