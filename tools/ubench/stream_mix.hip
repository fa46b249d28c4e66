// Microbenchmark (not part of the library): ceilings for the fused-mask kernel's traffic
// shape: read 3 B/px, write 1 B/px, 78.6 Mpx per launch.
//   v0: thread loads 48 contiguous bytes (3 x dwordx4 at 48-byte lane stride), stores 16 bytes
//   v1: wave-coalesced loads (lane i reads 16 B at 16*i in three 1 KiB chunks), stores 16 bytes
//   v2: v0 with non-temporal loads/stores
//   v3: v1 with non-temporal
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(const u4* __restrict__ in, u4* __restrict__ out, size_t ngroups)
{
    // one "group" = 16 px = 48 B in, 16 B out
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += stride) {
        u4 a, b, c;
        if (V == 0 || V == 2) {
            const u4* p = in + g * 3;
            if (V == 2) { a = __builtin_nontemporal_load(p); b = __builtin_nontemporal_load(p + 1); c = __builtin_nontemporal_load(p + 2); }
            else { a = p[0]; b = p[1]; c = p[2]; }
        } else {
            const size_t wave0 = (g & ~(size_t)63) * 3;  // first u4 of this wave's 3 KiB
            const int lane = threadIdx.x & 63;
            const u4* p = in + wave0 + lane;
            if (V == 3) { a = __builtin_nontemporal_load(p); b = __builtin_nontemporal_load(p + 64); c = __builtin_nontemporal_load(p + 128); }
            else { a = p[0]; b = p[64]; c = p[128]; }
        }
        u4 o;
        o.x = a.x ^ b.x ^ c.x; o.y = a.y ^ b.y ^ c.y; o.z = a.z ^ b.z ^ c.z; o.w = a.w ^ b.w ^ c.w;
        if (V >= 2) __builtin_nontemporal_store(o, out + g); else out[g] = o;
    }
}

int main(int argc, char** argv)
{
    const int nfr = argc > 1 ? atoi(argv[1]) : 256;
    const size_t px = (size_t)nfr * 640 * 480, ngroups = px / 16;
    printf("frames %d  (%.0f MB in, %.0f MB out)\n", nfr, px * 3 / 1e6, px / 1e6);
    u4 *in, *out;
    CK(hipMalloc(&in, px * 3)); CK(hipMalloc(&out, px));
    CK(hipMemset(in, 1, px * 3));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {2048, 8192, 19200}) {
        for (int v = 0; v < 4; ++v) {
            float best = 1e9;
            for (int it = 0; it < 12; ++it) {
                CK(hipEventRecord(e0));
                if (v == 0) k<0><<<blocks, 256>>>(in, out, ngroups);
                if (v == 1) k<1><<<blocks, 256>>>(in, out, ngroups);
                if (v == 2) k<2><<<blocks, 256>>>(in, out, ngroups);
                if (v == 3) k<3><<<blocks, 256>>>(in, out, ngroups);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2 && ms < best) best = ms;
            }
            printf("blocks %5d v%d: %.4f ms  %.0f GB/s\n", blocks, v, best, px * 4.0 / best / 1e6);
        }
    }
    return 0;
}
