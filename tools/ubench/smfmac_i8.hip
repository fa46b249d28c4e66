// v_smfmac_i32_32x32x64_i8 on gfx950 (2:4 structured-sparse A, 64 dense K columns per instruction):
//   1. lane maps found by PROBING, not assumed: one compressed A byte set to 1 (lane la, byte ja, its two index bits = v),
//      B filled with position codes; the non-zero D entries tell which dense (m, k) that compressed byte stands for and
//      which (lane, byte) of B it meets;
//   2. a random exact-integer check of the map that came out;
//   3. cycles per instruction and the clock the chip holds (random operands), alone, mixed 5 dense : 1 sparse (the match
//      kernel's mix after the change) and with the 8 v_perm_b32 per sparse B operand between the matrix instructions.
// hipcc --offload-arch=gfx950 -O3 smfmac_i8.hip -o smfmac_i8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// one wave: D = smfmac(A, B, 0, idx) with operands as given per lane
__global__ void k_one(const i32x4* A, const i32x8* B, const int* idx, i32x16* D, int nprobe)
{
    const int p = blockIdx.x, l = threadIdx.x;
    i32x16 c = {0};
    c = __builtin_amdgcn_smfmac_i32_32x32x64_i8(A[p * 64 + l], B[p * 64 + l], c, idx[p * 64 + l], 0, 0);
    D[p * 64 + l] = c;
}


// ---- the match kernel's scheme (k_match_mfma, round 6): one template row x one image row of 32 frames, two 32-column output blocks,
// six matrix instructions each: the Toeplitz blocks d = 0 and d = 6 of a block pair up in ONE sparse instruction.
// Row slots (1 KiB each = 64 lanes x 16 B): P0 = image blocks (0, 6) interleaved by 16-bit pairs (dword t of lane (n, h) =
// {L'[16 h + 2 t], L'[16 h + 2 t + 1]} of block 0, then of block 6), slots 0-1; P1 = blocks (1, 7), slots 2-3; blocks 2..5 plain, slots 4-7.
// A fragments: 0 = pair (d0, d6) compressed, 1..5 = d1..d5 dense, 6 = d1 in the low positions of P1, 7 = d5 in the high positions of P0.
__global__ void k_scheme(const i32x4* slots, const i32x4* afr, const int* idxp, i32x16* D)
{
    const int l = threadIdx.x;
    auto S8 = [&](int s) { i32x8 v; const i32x4 lo = slots[s * 64 + l], hi = slots[(s + 1) * 64 + l]; for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; } return v; };
    const i32x8 p0 = S8(0), p1 = S8(2);
    const i32x4 k2 = slots[4 * 64 + l], k3 = slots[5 * 64 + l], k4 = slots[6 * 64 + l], k5 = slots[7 * 64 + l];
    i32x4 a[8];
    for (int f = 0; f < 8; ++f) a[f] = afr[f * 64 + l];
    const int idx = idxp[l];
    i32x16 c0 = {0}, c1 = {0};
    c0 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a[0], p0, c0, idx, 0, 0);
    c0 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a[6], p1, c0, 0x44444444, 0, 0);
    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2], k2, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[3], k3, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[4], k4, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[5], k5, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a[0], p1, c1, idx, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1], k2, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2], k3, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[3], k4, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[4], k5, c1, 0, 0, 0);
    c1 = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a[7], p0, c1, (int)0xEEEEEEEE, 0, 0);
    D[l] = c0; D[64 + l] = c1;
}

static int scheme_check(int tw, unsigned seed)
{
    srand(seed);
    std::vector<int> T(tw), Lr(32 * 256);
    for (auto& v : T) v = rand() % 256 - 128;
    for (auto& v : Lr) v = rand() % 256 - 128;
    auto Tq = [&](int col) { return col >= 0 && col < tw ? T[col] : 0; };
    std::vector<int8_t> slots(8 * 64 * 16), afr(8 * 64 * 16, 0);
    std::vector<int> idx(64, 0);
    for (int l = 0; l < 64; ++l) {
        const int n = l & 31, h = l >> 5;
        for (int p = 0; p < 2; ++p)      // P0 = blocks (0, 6), P1 = blocks (1, 7)
            for (int t = 0; t < 8; ++t)
                for (int c = 0; c < 4; ++c) {
                    const int kb = (c < 2 ? p : p + 6), k = 16 * h + 2 * t + (c & 1);
                    slots[(((2 * p + (t >> 2)) * 64 + l) * 16) + 4 * (t & 3) + c] = (int8_t)Lr[n * 256 + 32 * kb + k];
                }
        for (int kb = 2; kb < 6; ++kb)
            for (int j = 0; j < 16; ++j) slots[((kb + 2) * 64 + l) * 16 + j] = (int8_t)Lr[n * 256 + 32 * kb + 16 * h + j];
        const int m = l & 31, hA = l >> 5;
        for (int d = 1; d <= 5; ++d)
            for (int j = 0; j < 16; ++j) afr[(d * 64 + l) * 16 + j] = (int8_t)Tq(32 * d + 16 * hA + j - m);
        for (int ja = 0; ja < 16; ja += 2) {
            const int hB = ja >> 3, t = 4 * hA + ((ja >> 1) & 3), k0 = 16 * hB + 2 * t;
            afr[(6 * 64 + l) * 16 + ja] = (int8_t)Tq(32 + k0 - m); afr[(6 * 64 + l) * 16 + ja + 1] = (int8_t)Tq(32 + k0 + 1 - m);
            afr[(7 * 64 + l) * 16 + ja] = (int8_t)Tq(160 + k0 - m); afr[(7 * 64 + l) * 16 + ja + 1] = (int8_t)Tq(160 + k0 + 1 - m);
            // the pair: positions 0, 1 = block d = 0 at k0, k0 + 1; positions 2, 3 = block d = 6
            int cand[4], ncand = 0;
            const int cols[4] = {k0 - m, k0 + 1 - m, 192 + k0 - m, 192 + k0 + 1 - m};
            for (int v = 0; v < 4; ++v) if (cols[v] >= 0 && cols[v] < tw) cand[ncand++] = v;
            if (ncand > 2) return -1;
            int v0, v1;
            if (ncand == 2) { v0 = cand[0]; v1 = cand[1]; }
            else if (ncand == 1) { if (cand[0] < 3) { v0 = cand[0]; v1 = 3; } else { v0 = 0; v1 = 3; } }
            else { v0 = 0; v1 = 1; }
            afr[(0 * 64 + l) * 16 + ja] = (int8_t)Tq(cols[v0]); afr[(0 * 64 + l) * 16 + ja + 1] = (int8_t)Tq(cols[v1]);
            idx[l] |= (v0 << (2 * ja)) | (v1 << (2 * ja + 2));
        }
    }
    i32x4 *dS, *dF; int* dI; i32x16* dD;
    (void)hipMalloc(&dS, slots.size()); (void)hipMalloc(&dF, afr.size()); (void)hipMalloc(&dI, 256); (void)hipMalloc(&dD, 2 * 64 * 64);
    (void)hipMemcpy(dS, slots.data(), slots.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dF, afr.data(), afr.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dI, idx.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_scheme, dim3(1), dim3(64), 0, 0, dS, dF, dI, dD);
    std::vector<int> out(2 * 64 * 16);
    (void)hipMemcpy(out.data(), dD, out.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int xb = 0; xb < 2; ++xb) for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int x = 32 * xb + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = l & 31;
        int ref = 0;
        for (int j = 0; j < tw; ++j) if (x + j < 256) ref += T[j] * Lr[n * 256 + x + j];
        bad += out[(xb * 64 + l) * 16 + r] != ref;
    }
    (void)hipFree(dS); (void)hipFree(dF); (void)hipFree(dI); (void)hipFree(dD);
    return bad;
}

template <int MODE, int NACC>
__global__ __launch_bounds__(64, 1) void k_rate(const i32x4* __restrict__ src, int iters, uint64_t* __restrict__ stamps, int* sink)
{
    const i32x4 a = src[threadIdx.x];
    i32x4 b0 = src[64 + threadIdx.x], b1 = src[128 + threadIdx.x];
    i32x8 bb;
    for (int e = 0; e < 4; ++e) { bb[e] = b0[e]; bb[4 + e] = b1[e]; }
    const int idx = src[192 + threadIdx.x][0] & 0x77777777 | 0x44444444;   // index pairs (0..3, 1..3): any bits are legal
    i32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0;
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // sparse only
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, bb, acc[q], idx, 0, 0);
        } else if (MODE == 1) {   // dense only (reference, same loop shape)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0, acc[q], 0, 0, 0);
        } else if (MODE == 2) {   // 5 dense : 1 sparse per accumulator (six sub-blocks of NACC)
#pragma unroll
            for (int d = 0; d < 6; ++d) {
#pragma unroll
                for (int q = 0; q < NACC; ++q) {
                    if (d == 0) acc[q] = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, bb, acc[q], idx, 0, 0);
                    else acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, d & 1 ? b0 : b1, acc[q], 0, 0, 0);
                }
            }
        } else if (MODE == 3) {   // the same with the sparse B operand built by 8 v_perm_b32 in front of every sparse instruction
            asm volatile("" : "+v"(b0), "+v"(b1));   // the operands "change" every iteration: the perms cannot be hoisted
#pragma unroll
            for (int d = 0; d < 6; ++d) {
#pragma unroll
                for (int q = 0; q < NACC; ++q) {
                    if (d == 0) {
                        i32x8 pb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            pb[2 * e] = (int)__builtin_amdgcn_perm((uint32_t)b1[e], (uint32_t)(b0[e] + q), 0x05040100u);
                            pb[2 * e + 1] = (int)__builtin_amdgcn_perm((uint32_t)b1[e], (uint32_t)(b0[e] + q), 0x07060302u);
                        }
                        acc[q] = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, pb, acc[q], idx, 0, 0);
                    } else acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, d & 1 ? b0 : b1, acc[q], 0, 0, 0);
                }
            }
        } else {   // MODE 4: 7 dense per accumulator (today's kernel mix)
#pragma unroll
            for (int d = 0; d < 7; ++d)
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, d & 1 ? b0 : b1, acc[q], 0, 0, 0);
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][15];
    if (s == 0x12345678) *sink = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, int NACC>
static void run(const char* name, int per_iter, const i32x4* d_src, uint64_t* d_st, int* d_sink, int nblk, int iters)
{
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_rate<MODE, NACC>), dim3(nblk), dim3(64), 0, 0, d_src, iters, d_st, d_sink);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> st(2 * nblk);
    (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk, ns;
    for (int i = 0; i < nblk; ++i) {
        cyc.push_back((double)st[2 * i] / ((double)iters * per_iter));
        clk.push_back((double)st[2 * i] / ((double)st[2 * i + 1] / 100e6) / 1e9);
        ns.push_back((double)st[2 * i + 1] * 10.0 / ((double)iters * per_iter));
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end()); std::sort(ns.begin(), ns.end());
    printf("%-58s cycles/instr median %.2f  clock %.3f GHz  ns/instr %.2f\n", name, cyc[nblk / 2], clk[nblk / 2], ns[nblk / 2]);
}

int main()
{
    // ---------------- 1. probe the lane maps ----------------
    // probes: (la, ja, v) for every lane / compressed byte / index value; four B patterns each
    const int NPAT = 4;
    const int nprobe = 64 * 16 * 4 * NPAT;
    std::vector<int> hA((size_t)nprobe * 64 * 4, 0), hB((size_t)nprobe * 64 * 8), hI((size_t)nprobe * 64, 0);
    std::vector<int> hD((size_t)nprobe * 64 * 16);
    for (int la = 0; la < 64; ++la) for (int ja = 0; ja < 16; ++ja) for (int v = 0; v < 4; ++v) for (int pat = 0; pat < NPAT; ++pat) {
        const int p = ((la * 16 + ja) * 4 + v) * NPAT + pat;
        ((int8_t*)&hA[((size_t)p * 64 + la) * 4])[ja] = 1;
        // the partner byte of the pair (ja ^ 1) gets a different index so that the pair is a legal 2:4 pattern
        const int vp = (v + 1) & 3;
        uint32_t id = 0;
        id |= (uint32_t)v << (2 * ja);
        id |= (uint32_t)vp << (2 * (ja ^ 1));
        hI[(size_t)p * 64 + la] = (int)id;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
            int code = 1;
            if (pat == 1) code = j + 1;            // byte inside the lane's 32
            if (pat == 2) code = (l >> 5) + 1;     // lane half
            if (pat == 3) code = (l & 31) + 1;     // lane inside the half
            ((int8_t*)&hB[((size_t)p * 64 + l) * 8])[j] = (int8_t)code;
        }
    }
    i32x4* dA; i32x8* dB; int* dI; i32x16* dD;
    (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dB, hB.size() * 4); (void)hipMalloc(&dI, hI.size() * 4); (void)hipMalloc(&dD, hD.size() * 4);
    (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dI, hI.data(), hI.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_one, dim3(nprobe), dim3(64), 0, 0, dA, dB, dI, dD, nprobe);
    (void)hipMemcpy(hD.data(), dD, hD.size() * 4, hipMemcpyDeviceToHost);
    // analysis: D rows as in the dense instruction: lane l reg r = D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31]
    // map[la][ja][v] = (m, B byte, B half); n checked separately
    std::vector<int> mp(64 * 16 * 4 * 3, -1);
    int irregular = 0;
    for (int la = 0; la < 64; ++la) for (int ja = 0; ja < 16; ++ja) for (int v = 0; v < 4; ++v) {
        int res[NPAT]; int mrow = -1; bool ok = true;
        for (int pat = 0; pat < NPAT; ++pat) {
            const int p = ((la * 16 + ja) * 4 + v) * NPAT + pat;
            int rowseen = -1, val = -1;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
                const int d = hD[((size_t)p * 64 + l) * 16 + r];
                if (!d) continue;
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = l & 31;
                if (rowseen < 0) rowseen = m; else if (rowseen != m) ok = false;
                if (pat == 3) { if (d != n + 1) ok = false; }
                else { if (val < 0) val = d; else if (val != d) ok = false; }
            }
            if (rowseen < 0) ok = false;
            if (pat == 0) mrow = rowseen; else if (mrow != rowseen) ok = false;
            res[pat] = val;
        }
        if (!ok) { ++irregular; continue; }
        int* o = &mp[((la * 16 + ja) * 4 + v) * 3];
        o[0] = mrow; o[1] = res[1] - 1; o[2] = res[2] - 1;
    }
    printf("probe: %d irregular of %d (la, ja, v) probes\n", irregular, 64 * 16 * 4);
    const int show[] = {0, 1, 5, 31, 32, 33, 63};
    for (int la : show) {
        printf("A lane %2d:", la);
        for (int ja = 0; ja < 16; ++ja) {
            printf(" [b%d:", ja);
            for (int v = 0; v < 4; ++v) { const int* o = &mp[((la * 16 + ja) * 4 + v) * 3]; printf("%sm%d/h%d/j%d", v ? "," : "", o[0], o[2], o[1]); }
            printf("]");
        }
        printf("\n");
    }
    // the map (found by the probe on MI355X, round 6): compressed byte ja of A lane (m = la & 31, hA = la >> 5) with index v meets
    // B lane half hB = ja >> 3, byte 16 hA + 4 ((ja >> 1) & 3) + v, i.e. dword t = 4 hA + ((ja >> 1) & 3) of that lane's eight, byte v
    int hyp_bad = 0;
    for (int la = 0; la < 64; ++la) for (int ja = 0; ja < 16; ++ja) for (int v = 0; v < 4; ++v) {
        const int* o = &mp[((la * 16 + ja) * 4 + v) * 3];
        if (o[0] != (la & 31) || o[2] != (ja >> 3) || o[1] != 16 * (la >> 5) + 4 * ((ja >> 1) & 3) + v) ++hyp_bad;
    }
    printf("map {m = la & 31, B half = ja >> 3, B byte = 16 (la >> 5) + 4 ((ja >> 1) & 3) + v, n = lane & 31}: %d of %d probes disagree\n", hyp_bad, 64 * 16 * 4);

    // ---------------- 2. random exact check of the hypothesis ----------------
    {
        srand(11);
        std::vector<int8_t> Ad(32 * 64, 0), Bd(64 * 32);
        std::vector<int> rA(64 * 4, 0), rB(64 * 8), rI(64, 0), ref(32 * 32, 0), out(64 * 16);
        for (auto& b : Bd) b = (int8_t)(rand() % 256 - 128);
        for (int m = 0; m < 32; ++m) for (int g = 0; g < 16; ++g) {
            int i0 = rand() % 4, i1 = rand() % 4;
            if (i0 == i1) i1 = (i0 + 1) & 3;
            if (i0 > i1) std::swap(i0, i1);
            const int8_t v0 = (int8_t)(rand() % 256 - 128), v1 = (int8_t)(rand() % 256 - 128);
            Ad[m * 64 + 4 * g + i0] = v0; Ad[m * 64 + 4 * g + i1] = v1;
            // group g covers dense K = 4 g .. 4 g + 3 = B lane half hB = g >> 3, dword t = g & 7 -> A lane half t >> 2, bytes 8 hB + 2 (t & 3) + {0, 1}
            const int hB = g >> 3, t = g & 7, lane = m + 32 * (t >> 2), ja = 8 * hB + 2 * (t & 3);
            ((int8_t*)&rA[lane * 4])[ja] = v0; ((int8_t*)&rA[lane * 4])[ja + 1] = v1;
            rI[lane] |= (i0 << (2 * ja)) | (i1 << (2 * ja + 2));
        }
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) ((int8_t*)&rB[l * 8])[j] = Bd[(32 * (l >> 5) + j) * 32 + (l & 31)];
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { int s = 0; for (int k = 0; k < 64; ++k) s += Ad[m * 64 + k] * Bd[k * 32 + n]; ref[m * 32 + n] = s; }
        (void)hipMemcpy(dA, rA.data(), rA.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, rB.data(), rB.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dI, rI.data(), rI.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, dA, dB, dI, dD, 1);
        (void)hipMemcpy(out.data(), dD, out.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) bad += out[l * 16 + r] != ref[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
        printf("random 2:4 check (sorted index pairs): %d mismatches of 1024\n", bad);
    }

    // ---------------- 2b. the match kernel's 6-instruction scheme against the direct correlation ----------------
    for (int tw : {188, 162, 163, 175, 190, 192, 193})
        printf("scheme check, template row of %d columns: %d mismatches of 2048%s\n", tw, scheme_check(tw, 100 + tw), tw == 193 ? " (-1 = not 2:4, expected for 193)" : "");

    // ---------------- 3. rates ----------------
    const int nblk = 1024, iters = 2000;
    std::vector<int> h(4 * 64 * 4);
    i32x4* d_src; uint64_t* d_st; int* d_sink;
    (void)hipMalloc(&d_src, h.size() * 4); (void)hipMalloc(&d_st, 2 * nblk * 8); (void)hipMalloc(&d_sink, 4);
    for (int pass = 0; pass < 2; ++pass) {
        srand(7);
        for (auto& v : h) v = pass ? (int)((unsigned)rand() * 2654435761u) : 0;
        (void)hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        printf("== %s operands, 1024 waves (one per SIMD)\n", pass ? "random" : "zero");
        run<1, 10>("dense 32x32x32 i8, 10 accumulators", 10, d_src, d_st, d_sink, nblk, iters);
        run<0, 10>("sparse 32x32x64 i8, 10 accumulators", 10, d_src, d_st, d_sink, nblk, iters);
        run<4, 10>("7 dense per accumulator (today's mix), per 7-group", 10, d_src, d_st, d_sink, nblk, iters / 4);
        run<2, 10>("5 dense + 1 sparse per accumulator, per 6-group", 10, d_src, d_st, d_sink, nblk, iters / 4);
        run<3, 10>("5 dense + 1 sparse + 8 v_perm per sparse, per 6-group", 10, d_src, d_st, d_sink, nblk, iters / 4);
    }
    return 0;
}
