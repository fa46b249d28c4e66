// Microbenchmark (not part of the library): does a DYNAMIC work queue beat the static grid-stride split for the fused-mask
// kernel's traffic (48 B in, 16 B out per thread and step, persistent 1024-thread workgroups, two per CU)?
// Round 4's per-workgroup clocks (profiles/r04/fused_workgroup_clock.txt) show the CU's two workgroups finishing 25 % apart under
// the static split (oldest-first arbitration): the launch's last quarter runs with one workgroup per CU.
//   static      : chunk c = blockIdx + k * grid                       (the "3:1 plain" stream of round 4's stream_lds microbenchmark, profiles/r04/fused_split_and_lds_stream.txt)
//   dynamic/CH  : blocks of CH consecutive chunks from an atomic counter, the next block's index fetched one block ahead
// Also prints the spread of the workgroups' end times (100 MHz real-time clock) for both.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int T = 1024;

__device__ __forceinline__ void step(const u4* __restrict__ in, u4* __restrict__ out, size_t c, int tid)
{
    const u4* p = in + c * T * 3 + tid;
    const u4 a = p[0], b = p[T], d = p[2 * T];
    u4 o;
    o.x = a.x ^ b.x ^ d.x; o.y = a.y ^ b.y ^ d.y; o.z = a.z ^ b.z ^ d.z; o.w = a.w ^ b.w ^ d.w;
    __builtin_nontemporal_store(o, out + c * T + tid);
}

__global__ __launch_bounds__(T) void k_static(const u4* __restrict__ in, u4* __restrict__ out, size_t nchunks, unsigned long long* __restrict__ ends)
{
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) step(in, out, c, threadIdx.x);
    if (threadIdx.x == 0) ends[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

__global__ __launch_bounds__(T) void k_dynamic(const u4* __restrict__ in, u4* __restrict__ out, size_t nchunks, int CH, unsigned* __restrict__ counter,
                                               unsigned long long* __restrict__ ends)
{
    __shared__ unsigned nxt[2];
    const size_t nblocks = (nchunks + CH - 1) / CH;
    if (threadIdx.x == 0) { nxt[0] = atomicAdd(counter, 1u); nxt[1] = atomicAdd(counter, 1u); }
    __syncthreads();
    for (int it = 0;; ++it) {
        const unsigned b = nxt[it & 1];
        if (b >= nblocks) break;
        __syncthreads();   // everybody has read nxt[it & 1]
        if (threadIdx.x == 0) nxt[it & 1] = atomicAdd(counter, 1u);   // the block after next
        const size_t c0 = (size_t)b * CH, c1 = c0 + CH < nchunks ? c0 + CH : nchunks;
        for (size_t c = c0; c < c1; ++c) step(in, out, c, threadIdx.x);
        __syncthreads();
    }
    if (threadIdx.x == 0) ends[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? atoi(argv[1]) : 3200;           // input megabytes
    const size_t nchunks = mb * 1000000 / 48 / T;
    const size_t in_bytes = nchunks * T * 48, out_bytes = nchunks * T * 16;
    const int NBUF = mb < 1000 ? 4 : 1;   // small launches rotate over buffers beyond the Infinity Cache
    u4 *in, *out;
    unsigned* counter;
    unsigned long long* ends;
    CK(hipMalloc(&in, in_bytes * NBUF)); CK(hipMalloc(&out, out_bytes * NBUF)); CK(hipMalloc(&counter, 4096)); CK(hipMalloc(&ends, 8 * 1024));
    CK(hipMemset(in, 1, in_bytes * NBUF));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(512);
    const int blocks = 512;
    for (int v : {0, 1, 2, 4, 8, 16, 32, 0}) {
        float best = 1e9, sum = 0;
        int cnt = 0;
        double spread = 0;
        for (int it = 0; it < 10; ++it) {
            const u4* pin = in + (size_t)(it % NBUF) * (in_bytes / 16);
            u4* pout = out + (size_t)(it % NBUF) * (out_bytes / 16);
            CK(hipMemsetAsync(counter, 0, 4));
            CK(hipEventRecord(e0));
            if (v == 0) k_static<<<blocks, T>>>(pin, pout, nchunks, ends);
            else k_dynamic<<<blocks, T>>>(pin, pout, nchunks, v, counter, ends);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), ends, 8 * blocks, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            if (it >= 2) { if (ms < best) best = ms; sum += ms; ++cnt; spread += (double)(h[blocks - 1] - h[0]) / 100.0; }
        }
        const double bytes = (double)nchunks * T * 64;
        printf("%4zu MB in, %-10s CH %2d: best %.4f ms %5.0f GB/s   avg %.4f ms %5.0f GB/s   first-to-last workgroup end %.1f us\n", mb, v ? "dynamic" : "static", v,
               best, bytes / best / 1e6, sum / cnt, bytes / (sum / cnt) / 1e6, spread / cnt);
    }
    return 0;
}
