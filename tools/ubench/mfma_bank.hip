// Does the VGPR alignment of the A / B operands of v_mfma_i32_32x32x32_i8 change its issue rate?
// Ten accumulators in AGPRs (as in k_match_mfma), A at v[8:11], B at v[12:15] or v[14:17]
// (register tuples must be even-aligned on gfx950, so those are the two possible alignments modulo 4).
// hipcc --offload-arch=gfx950 -O3 mfma_bank.hip -o mfma_bank
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define MFMA10(B)                                                     \
    "v_mfma_i32_32x32x32_i8 a[0:15], v[8:11], " B ", a[0:15]\n"       \
    "v_mfma_i32_32x32x32_i8 a[16:31], v[8:11], " B ", a[16:31]\n"     \
    "v_mfma_i32_32x32x32_i8 a[32:47], v[8:11], " B ", a[32:47]\n"     \
    "v_mfma_i32_32x32x32_i8 a[48:63], v[8:11], " B ", a[48:63]\n"     \
    "v_mfma_i32_32x32x32_i8 a[64:79], v[8:11], " B ", a[64:79]\n"     \
    "v_mfma_i32_32x32x32_i8 a[80:95], v[8:11], " B ", a[80:95]\n"     \
    "v_mfma_i32_32x32x32_i8 a[96:111], v[8:11], " B ", a[96:111]\n"   \
    "v_mfma_i32_32x32x32_i8 a[112:127], v[8:11], " B ", a[112:127]\n" \
    "v_mfma_i32_32x32x32_i8 a[128:143], v[8:11], " B ", a[128:143]\n" \
    "v_mfma_i32_32x32x32_i8 a[144:159], v[8:11], " B ", a[144:159]\n"

#define CLOBBERS                                                                                                      \
    "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", \
        "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a31", "a32", "a47", "a48", "a63", "a64", "a79", "a80", "a95", "a96", \
        "a111", "a112", "a127", "a128", "a143", "a144", "a159"

template <int OFF>
__global__ __launch_bounds__(64, 1) void k_bank(int iters, uint64_t* __restrict__ stamps)
{
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OFF == 0) asm volatile(MFMA10("v[12:15]") ::: CLOBBERS);
        if (OFF == 2) asm volatile(MFMA10("v[14:17]") ::: CLOBBERS);
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) stamps[blockIdx.x] = c1 - c0;
}

template <int OFF>
static void run(uint64_t* d_st, int nblk, int iters)
{
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_bank<OFF>, dim3(nblk), dim3(64), 0, 0, iters, d_st);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> st(nblk);
    (void)hipMemcpy(st.data(), d_st, nblk * 8, hipMemcpyDeviceToHost);
    std::sort(st.begin(), st.end());
    printf("B at v[%d:%d] (A at v[8:11]): %.2f cycles per MFMA\n", 12 + OFF, 15 + OFF, (double)st[nblk / 2] / (iters * 10.0));
}

int main()
{
    const int nblk = 1024, iters = 2000;
    uint64_t* d_st;
    (void)hipMalloc(&d_st, nblk * 8);
    run<0>(d_st, nblk, iters); run<2>(d_st, nblk, iters);
    return 0;
}
