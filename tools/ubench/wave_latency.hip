// What ONE wave pays per instruction when nothing else runs on its SIMD: dependent chains of the instruction kinds the
// latency-bound kernels here are made of (k_jpeg_huff's synchronisation rounds, k_dials' angle phase).
// One workgroup of 64 threads on an idle GPU; s_memtime around `iters` repetitions of a 16-fold pattern.
// hipcc --offload-arch=gfx950 -O3 wave_latency.hip -o wave_latency
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define TEST(NAME, PATTERN)                                                                                       \
    __global__ __launch_bounds__(64) void NAME(int iters, uint64_t* out, uint32_t* buf)                           \
    {                                                                                                             \
        __shared__ uint32_t lds[1024];                                                                            \
        for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = ((i * 37 + 11) & 1023) * 4;                         \
        __syncthreads();                                                                                          \
        uint32_t a = threadIdx.x * 4, b = 3, c = 5, d = 7;                                                        \
        uint64_t q = threadIdx.x;                                                                                 \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                         \
        for (int it = 0; it < iters; ++it)                                                                        \
            asm volatile(REP16(PATTERN) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(q)::"s20", "s21", "s22", "s23", \
                         "s24", "s25", "vcc", "scc", "memory");                                                    \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                         \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                                   \
        buf[threadIdx.x] = a + b + c + d + (uint32_t)q;                                                           \
    }

TEST(k_add_dep, "v_add_u32 %0, %0, %1\n")
TEST(k_add_indep, "v_add_u32 %0, %1, %2\n v_add_u32 %3, %1, %2\n")
TEST(k_bfe_dep, "v_bfe_u32 %0, %0, 1, 31\n")
TEST(k_alignbit_dep, "v_alignbit_b32 %0, %0, %1, 7\n")
TEST(k_perm_dep, "v_perm_b32 %0, %0, %1, %2\n")
TEST(k_lshl64_dep, "v_lshlrev_b64 %4, 1, %4\n")
TEST(k_lshladd64_dep, "v_lshl_add_u64 %4, %4, 1, %4\n")
TEST(k_cmp_cndmask_vcc, "v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %2, %0, vcc\n")
TEST(k_cmp_cndmask_sgpr, "v_cmp_lt_u32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %2, %0, s[20:21]\n")
TEST(k_cmp_sand_cndmask, "v_cmp_lt_u32 s[20:21], %0, %1\n s_and_b64 s[22:23], s[20:21], exec\n s_nop 1\n v_cndmask_b32 %0, %2, %0, s[22:23]\n")
TEST(k_cmp_2sand_cndmask,
     "v_cmp_lt_u32 s[20:21], %0, %1\n s_and_b64 s[22:23], s[20:21], exec\n s_or_b64 s[22:23], s[22:23], s[20:21]\n s_nop 1\n v_cndmask_b32 %0, %2, %0, s[22:23]\n")
TEST(k_saveexec_pair, "v_cmp_lt_u32 vcc, %1, %2\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]\n")
TEST(k_salu_dep, "s_add_u32 s20, s20, 1\n")
TEST(k_readfirstlane_roundtrip, "v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 1\n v_add_u32 %0, s20, %0\n")
TEST(k_lds_chase, "ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n")
TEST(k_lds_chase_plus4, "ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %1, %1, %0\n v_add_u32 %2, %2, %1\n v_add_u32 %3, %3, %2\n v_and_b32 %0, 0xffc, %0\n")
TEST(k_branch_skip, "v_cmp_lt_u32 vcc, %1, %2\n s_cbranch_vccz 1f\n v_add_u32 %0, %0, %1\n1:\n")
TEST(k_branch_taken, "v_cmp_gt_u32 vcc, %1, %2\n s_cbranch_vccz 1f\n v_add_u32 %0, %0, %1\n1:\n")
TEST(k_execz_skip, "v_cmp_gt_u32 vcc, %1, %2\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, %1\n1:\n s_or_b64 exec, exec, s[20:21]\n")

// global pointer chase: every lane walks its own ring inside one 128-byte line set (L1 hits after the first lap)
__global__ __launch_bounds__(64) void k_global_chase(int iters, uint64_t* out, uint32_t* buf, const uint32_t* __restrict__ ring)
{
    uint32_t i = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters * 16; ++it) i = __builtin_nontemporal_load(ring + i) & 1023u;
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    buf[threadIdx.x] = i;
}
__global__ __launch_bounds__(64) void k_global_chase_plain(int iters, uint64_t* out, uint32_t* buf, const uint32_t* ring)
{
    uint32_t i = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters * 16; ++it) i = ((const volatile uint32_t*)ring)[i] & 1023u;
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    buf[threadIdx.x] = i;
}

#define RUN(NAME, NINSTR)                                                                                      \
    do {                                                                                                        \
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(NAME, dim3(1), dim3(64), 0, 0, iters, d_out, d_buf);      \
        (void)hipDeviceSynchronize();                                                                           \
        uint64_t c = 0;                                                                                         \
        (void)hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);                                                   \
        printf("%-28s %7.2f cycles per pattern (%d instructions) = %6.2f per instruction\n", #NAME,            \
               (double)c / (iters * 16.0), NINSTR, (double)c / (iters * 16.0) / NINSTR);                        \
    } while (0)

int main()
{
    const int iters = 4000;
    uint64_t* d_out;
    uint32_t *d_buf, *d_ring;
    (void)hipMalloc(&d_out, 8);
    (void)hipMalloc(&d_buf, 4096);
    (void)hipMalloc(&d_ring, 4096);
    uint32_t ring[1024];
    for (int i = 0; i < 1024; ++i) ring[i] = (i * 37 + 11) & 1023;
    (void)hipMemcpy(d_ring, ring, sizeof(ring), hipMemcpyHostToDevice);
    RUN(k_add_dep, 1); RUN(k_add_indep, 2); RUN(k_bfe_dep, 1); RUN(k_alignbit_dep, 1); RUN(k_perm_dep, 1);
    RUN(k_lshl64_dep, 1); RUN(k_lshladd64_dep, 1);
    RUN(k_cmp_cndmask_vcc, 3); RUN(k_cmp_cndmask_sgpr, 3); RUN(k_cmp_sand_cndmask, 4); RUN(k_cmp_2sand_cndmask, 5);
    RUN(k_saveexec_pair, 4); RUN(k_salu_dep, 1); RUN(k_readfirstlane_roundtrip, 3);
    RUN(k_lds_chase, 1); RUN(k_lds_chase_plus4, 5);
    RUN(k_branch_skip, 3); RUN(k_branch_taken, 2); RUN(k_execz_skip, 4);
    for (int v = 0; v < 2; ++v) {
        for (int w = 0; w < 2; ++w) {
            if (v == 0) hipLaunchKernelGGL(k_global_chase, dim3(1), dim3(64), 0, 0, iters, d_out, d_buf, d_ring);
            else hipLaunchKernelGGL(k_global_chase_plain, dim3(1), dim3(64), 0, 0, iters, d_out, d_buf, d_ring);
        }
        (void)hipDeviceSynchronize();
        uint64_t c = 0;
        (void)hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
        printf("%-28s %7.2f cycles per dependent load (4 KB ring: %s)\n", v == 0 ? "k_global_chase_nt" : "k_global_chase", (double)c / (iters * 16.0),
               "address arithmetic included");
    }
    return 0;
}
