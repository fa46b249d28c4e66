// Cycles per v_mfma_i32_32x32x32_i8 / v_mfma_i32_16x16x64_i8 issued back to back by one wave per SIMD
// (independent accumulators, operands in registers), and the clock the chip holds while doing so on zero
// and on random operands.  hipcc --offload-arch=gfx950 -O3 mfma_i8_rate.hip -o mfma_i8_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(64, 1) void k_rate(const i32x4* __restrict__ src, int iters, uint64_t* __restrict__ stamps, int* sink)
{
    const i32x4 a = src[threadIdx.x], b = src[64 + threadIdx.x];
    i32x16 acc32[SHAPE == 32 ? NACC : 1];
    i32x4 acc16[SHAPE == 16 ? NACC : 1];
    for (int q = 0; q < NACC; ++q) {
        if (SHAPE == 32) for (int e = 0; e < 16; ++e) acc32[q][e] = 0;
        else for (int e = 0; e < 4; ++e) acc16[q][e] = 0;
    }
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            if (SHAPE == 32) acc32[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc32[q], 0, 0, 0);
            else acc16[q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc16[q], 0, 0, 0);
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int q = 0; q < NACC; ++q) s += SHAPE == 32 ? acc32[q][0] + acc32[q][15] : acc16[q][0] + acc16[q][3];
    if (s == 0x12345678) *sink = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int NACC>
static void run(const char* name, const i32x4* d_src, uint64_t* d_st, int* d_sink, int nblk, int iters)
{
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_rate<SHAPE, NACC>), dim3(nblk), dim3(64), 0, 0, d_src, iters, d_st, d_sink);
    hipDeviceSynchronize();
    std::vector<uint64_t> st(2 * nblk);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int i = 0; i < nblk; ++i) { cyc.push_back((double)st[2 * i] / ((double)iters * NACC)); clk.push_back((double)st[2 * i] / ((double)st[2 * i + 1] / 100e6) / 1e9); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double ops = SHAPE == 32 ? 2.0 * 32 * 32 * 32 : 2.0 * 16 * 16 * 64;
    printf("%-34s waves %4d  cycles/MFMA median %.2f  clock median %.3f GHz  -> %.2f POP/s on %d SIMDs\n", name, nblk,
           cyc[nblk / 2], clk[nblk / 2], ops / cyc[nblk / 2] * clk[nblk / 2] * 1e9 * nblk / 1e15, nblk);
}


// The match kernel's sub-block shape: ten 32x32x32 MFMAs on ten accumulators, NLD 16-byte loads per lane
// issued between the groups and consumed (as B operands) a whole group later.  LDK: 0 = global (L1/L2
// resident buffer), 1 = LDS.
template <int NLD, int LDK>
__global__ __launch_bounds__(64, 1) void k_rate_loads(const i32x4* __restrict__ src, int iters, uint64_t* __restrict__ stamps, int* sink)
{
    __shared__ i32x4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = src[i & 127];
    __syncthreads();
    const i32x4 a = src[threadIdx.x];
    i32x4 b[2][4];
    for (int q = 0; q < 4; ++q) { b[0][q] = src[64 + threadIdx.x]; b[1][q] = src[threadIdx.x]; }
    i32x16 acc[10];
    for (int q = 0; q < 10; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int idx = ((it + h) * 4 + q) * 64 % 1024 + threadIdx.x;
                b[h ^ 1][q] = LDK ? lds[idx] : src[idx & 127];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 10; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[h][q & 3], acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int q = 0; q < 10; ++q) s += acc[q][0] + acc[q][15];
    if (s == 0x12345678) *sink = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NLD, int LDK>
static void run_loads(const char* name, const i32x4* d_src, uint64_t* d_st, int* d_sink, int nblk, int iters)
{
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_rate_loads<NLD, LDK>), dim3(nblk), dim3(64), 0, 0, d_src, iters, d_st, d_sink);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> st(2 * nblk);
    (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int i = 0; i < nblk; ++i) { cyc.push_back((double)st[2 * i] / ((double)iters * 10)); clk.push_back((double)st[2 * i] / ((double)st[2 * i + 1] / 100e6) / 1e9); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    printf("%-34s waves %4d  cycles/MFMA median %.2f  clock median %.3f GHz\n", name, nblk, cyc[nblk / 2], clk[nblk / 2]);
}

int main()
{
    const int nblk = 1024, iters = 20000;
    std::vector<int> h(2 * 64 * 4);  // 128 i32x4
    i32x4* d_src; uint64_t* d_st; int* d_sink;
    hipMalloc(&d_src, h.size() * 4); hipMalloc(&d_st, 2 * nblk * 8); hipMalloc(&d_sink, 4);
    for (int pass = 0; pass < 2; ++pass) {
        srand(7);
        for (auto& v : h) v = pass ? (int)((unsigned)rand() * 2654435761u) : 0;
        hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        printf("== %s operands\n", pass ? "random" : "zero");
        run<32, 10>("32x32x32 i8, 10 accumulators", d_src, d_st, d_sink, nblk, iters / 10);
        run<32, 2>("32x32x32 i8, 2 accumulators", d_src, d_st, d_sink, nblk, iters / 2);
        run<16, 10>("16x16x64 i8, 10 accumulators", d_src, d_st, d_sink, nblk, iters / 10 * 2);
        run<16, 2>("16x16x64 i8, 2 accumulators", d_src, d_st, d_sink, nblk, iters);
        run_loads<0, 0>("10 MFMA groups, no loads", d_src, d_st, d_sink, nblk, iters / 10);
        run_loads<2, 0>("10 MFMA + 2 global dwordx4", d_src, d_st, d_sink, nblk, iters / 10);
        run_loads<4, 0>("10 MFMA + 4 global dwordx4", d_src, d_st, d_sink, nblk, iters / 10);
        run_loads<2, 1>("10 MFMA + 2 ds_read_b128", d_src, d_st, d_sink, nblk, iters / 10);
        run_loads<4, 1>("10 MFMA + 4 ds_read_b128", d_src, d_st, d_sink, nblk, iters / 10);
    }
    return 0;
}
