// Synthetic partners for a co-residency experiment (tools/corun_partner.py): small kernels (<= 32 registers, 256 threads) that can
// live on a SIMD beside one k_match_mfma wave and do ONE kind of work, so that what the match waves lose to each kind can be told apart.
//   mode 0: vector ALU only (dependent fma chains)
//   mode 1: 16-byte coalesced loads over a 256 KiB buffer (stays in L2: texture-addresser / L2 bandwidth, no HBM traffic)
//   mode 2: 16-byte coalesced loads streaming over a large buffer (HBM traffic, L2 fills)
//   mode 3: the same with non-temporal loads
//   mode 4: 16-byte coalesced streaming stores
//   mode 5: the same, non-temporal
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC partner.hip -o libpartner.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 8) void k_partner(u32x4* __restrict__ buf, size_t n16, int iters, int gap, unsigned* sink)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    float x = (float)tid, y = 1.0001f;
    u32x4 acc = {0u, 0u, 0u, 0u};
    size_t p = tid;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 64; ++k) { x = x * y + 0.5f; y = y * 0.99999f + 1e-6f; }
        } else if (MODE <= 3) {
            const size_t lim = MODE == 1 ? (size_t)16384 : n16;
            const u32x4 v = MODE == 3 ? __builtin_nontemporal_load(buf + (p % lim)) : buf[p % lim];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            p += stride;
        } else {
            const u32x4 v = {(unsigned)it, (unsigned)tid, 3u, 4u};
            if (MODE == 5) __builtin_nontemporal_store(v, buf + (p % n16)); else buf[p % n16] = v;
            p += stride;
        }
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(8);   // throttle: ~64 x 8 cycles per unit
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u || x == 12345.f) *sink = 1u;
}

extern "C" __attribute__((visibility("default"))) int partner_launch(int mode, int blocks, int iters, int gap, void* buf, size_t bytes, void* stream)
{
    static unsigned* sink = nullptr;
    if (!sink && hipMalloc((void**)&sink, 4) != hipSuccess) return -1;
    const size_t n16 = bytes / 16;
#define L(M) hipLaunchKernelGGL(k_partner<M>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (u32x4*)buf, n16, iters, gap, sink)
    switch (mode) { case 0: L(0); break; case 1: L(1); break; case 2: L(2); break; case 3: L(3); break; case 4: L(4); break; default: L(5); break; }
#undef L
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
