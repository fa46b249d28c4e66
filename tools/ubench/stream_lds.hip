// Microbenchmark (not part of the library): does reading the pixel stream through LDS-DMA (global_load_lds_dwordx4, the
// loads that write LDS directly, with and without the non-temporal policy) beat plain register loads for the fused-mask
// kernel's traffic -- 48 bytes in, 16 bytes out per thread and step, persistent workgroups -- at HBM sizes?
// MI355X_MICROARCH.md quotes 6.4-6.8 TB/s for an LDS-DMA READ stream and 6.29 TB/s for a float4 copy on its box; round 3's
// bare streams of this mix reached 5.3-5.5 TB/s on this pool's boxes.  Variants:
//   read-only  plain        : three dwordx4 loads per thread and step, summed (one dword written per workgroup at the end)
//   read-only  lds-dma      : the same bytes through global_load_lds_dwordx4 into a ring of LDS buffers, read back with ds_read_b128
//   read-only  lds-dma nt   : the same with the non-temporal cache policy
//   3:1        plain        : + one 16-byte non-temporal store per thread and step   (round 3's "packed" stream)
//   3:1        lds-dma      : loads through LDS-DMA, stores as above
//   3:1        lds-dma nt
// One 1024-thread workgroup per launch slot, 2 per CU (as the kernel), every step a workgroup's own 48 KiB + 16 KiB.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int T = 1024;
constexpr int DEPTH = 2;   // LDS buffers of 48 KiB in flight per workgroup (2 x 48 = 96 KiB: one workgroup per CU at DEPTH 2 with 1024 threads ... two fit at 64 KiB)

// MODE 0 plain loads, 1 LDS-DMA, 2 LDS-DMA non-temporal; WRITE: also stream 16 B per thread out
template <int MODE, bool WRITE>
__global__ __launch_bounds__(T) void k(const u4* __restrict__ in, u4* __restrict__ out, size_t nchunks, unsigned* __restrict__ sink)
{
    extern __shared__ u4 lds[];   // [DEPTH][3][T]
    u4 acc = {0, 0, 0, 0};
    const int tid = threadIdx.x;
    if (MODE == 0) {
        for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
            const u4* p = in + c * T * 3 + tid;
            const u4 a = p[0], b = p[T], d = p[2 * T];
            u4 o;
            o.x = a.x ^ b.x ^ d.x; o.y = a.y ^ b.y ^ d.y; o.z = a.z ^ b.z ^ d.z; o.w = a.w ^ b.w ^ d.w;
            if (WRITE) __builtin_nontemporal_store(o, out + c * T + tid);
            else { acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        }
    } else {
        // ring of DEPTH buffers: the DMA of step s + DEPTH - 1 is issued before step s is consumed
        constexpr int AUX = MODE == 2 ? 2 : 0;   // bit 1 = nt on gfx94x / gfx950 (sc0 = 1, sc1 = 16 are the other cache-policy bits)
        auto issue = [&](size_t c, int slot) {
            const u4* p = in + c * T * 3 + tid;
            u4* l = lds + (size_t)slot * 3 * T;
            // the LDS destination of a wave-instruction is its base + lane * 16: wave w of the workgroup writes its own 1 KiB piece
            __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void*)(l + (tid & ~63)), 16, 0, AUX);
            __builtin_amdgcn_global_load_lds(p + T, (__attribute__((address_space(3))) void*)(l + T + (tid & ~63)), 16, 0, AUX);
            __builtin_amdgcn_global_load_lds(p + 2 * T, (__attribute__((address_space(3))) void*)(l + 2 * T + (tid & ~63)), 16, 0, AUX);
        };
        size_t c = blockIdx.x;
        int s = 0;
        for (int q = 0; q < DEPTH - 1 && c + (size_t)q * gridDim.x < nchunks; ++q) issue(c + (size_t)q * gridDim.x, q);
        for (; c < nchunks; c += gridDim.x, ++s) {
            const size_t cn = c + (size_t)(DEPTH - 1) * gridDim.x;
            if (cn < nchunks) {
                issue(cn, (s + DEPTH - 1) % DEPTH);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (DEPTH - 1)) : "memory");   // this step's three pieces have landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // every wave reads back exactly the pieces it wrote: no barrier needed
            const u4* l = lds + (size_t)(s % DEPTH) * 3 * T + tid;
            const u4 a = l[0], b = l[T], d = l[2 * T];
            u4 o;
            o.x = a.x ^ b.x ^ d.x; o.y = a.y ^ b.y ^ d.y; o.z = a.z ^ b.z ^ d.z; o.w = a.w ^ b.w ^ d.w;
            if (WRITE) __builtin_nontemporal_store(o, out + c * T + tid);
            else { acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the slot is refilled next step
        }
    }
    if (!WRITE && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[blockIdx.x] = acc.x;   // keeps the loads alive
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 3200;           // input megabytes
    const size_t nchunks = mb * 1000000 / 48 / T;
    const size_t in_bytes = nchunks * T * 48, out_bytes = nchunks * T * 16;
    u4 *in, *out;
    unsigned* sink;
    CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, out_bytes)); CK(hipMalloc(&sink, 1 << 16));
    CK(hipMemset(in, 1, in_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t shmem = (size_t)DEPTH * 3 * T * sizeof(u4);
    CK(hipFuncSetAttribute((const void*)k<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    CK(hipFuncSetAttribute((const void*)k<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    CK(hipFuncSetAttribute((const void*)k<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    CK(hipFuncSetAttribute((const void*)k<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const char* names[6] = {"read-only plain", "read-only lds-dma", "read-only lds-dma nt", "3:1 plain", "3:1 lds-dma", "3:1 lds-dma nt"};
    for (int blocks : {256, 512}) {
        for (int v = 0; v < 6; ++v) {
            float best = 1e9, sum = 0;
            int cnt = 0;
            for (int it = 0; it < 8; ++it) {
                CK(hipEventRecord(e0));
                switch (v) {
                    case 0: k<0, false><<<blocks, T, 0>>>(in, out, nchunks, sink); break;
                    case 1: k<1, false><<<blocks, T, shmem>>>(in, out, nchunks, sink); break;
                    case 2: k<2, false><<<blocks, T, shmem>>>(in, out, nchunks, sink); break;
                    case 3: k<0, true><<<blocks, T, 0>>>(in, out, nchunks, sink); break;
                    case 4: k<1, true><<<blocks, T, shmem>>>(in, out, nchunks, sink); break;
                    default: k<2, true><<<blocks, T, shmem>>>(in, out, nchunks, sink); break;
                }
                CK(hipGetLastError());
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2) { if (ms < best) best = ms; sum += ms; ++cnt; }
            }
            const double bytes = (double)nchunks * T * (v >= 3 ? 64 : 48);
            printf("%4zu MB in, blocks %4d %-22s: best %.4f ms %5.0f GB/s   avg %.4f ms %5.0f GB/s\n", mb, blocks, names[v], best, bytes / best / 1e6,
                   sum / cnt, bytes / (sum / cnt) / 1e6);
        }
    }
    return 0;
}
