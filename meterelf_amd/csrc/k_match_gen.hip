// K2, general form -- cv2.matchTemplate(L, template, TM_CCOEFF) + cv2.minMaxLoc (reference: meterelf/_utils.py:91-97)
// on the matrix cores for ANY template up to 256 columns and ANY image size.  k_match_mfma.hip is the kernel tuned
// for one shape class (7 Toeplitz blocks per template row, maps up to 64 columns, enough rows to fill the chip);
// this one covers the rest with the same exact arithmetic, the same operand layouts (Lg, Atab, ws) and the same
// result records, so that no shape falls off a cliff onto the VALU kernel.
//
// Same GEMM, different loop nest.  D[x][frame] += A[x][k] B[k][frame] with A = Toeplitz block d of template row i and
// B = columns 32 (xb + d) .. + 31 of image row y + i (see k_match_mfma.hip).  There every wave keeps ALL column blocks
// of R + 1 image rows in registers (192 VGPRs for 8 blocks), which fixes the number of blocks at compile time.  Here
// the Toeplitz block d is the OUTER loop: for one d a wave slides down the template rows holding only the NXB image
// blocks d + xb of R + PD rows, so the number of blocks, the strip of the map and the template-row range are all runtime
// values, and R can be 8 rows (16 accumulator tiles): 3 loads per 16 MFMAs.
//
// Work is cut into TILES (R <= 8 map rows x NXB <= 2 column blocks x 32 frames); ONE WORKGROUP PER TILE AND FRAME GROUP,
// and the tile's K range (d, i) is cut into SLICES, one per wave of the workgroup, when the map is too small to fill
// the chip otherwise (BASELINE config 4: 17 x 33 positions).  Round 4: the slices add up in LDS -- every wave adds its
// accumulator tile into the workgroup's (ds_add_u32, conflict-free: element e of lane l at dword (16 block + e) 64 + l), one
// barrier, then the tile's row blocks are dealt out to the waves for the epilogue (window sums, OpenCV's double-precision
// post-pass, first maximum), and a second small LDS pass folds the waves' maxima.  Rounds 2-3 ran every slice as a
// workgroup of its own and added them up through global partial tiles (write-through stores, an arrival counter per
// tile, the last wave pulling the others' tiles at ~220 cycles per KiB: 4-5 us of a 33 us launch at config 4, 40 MB of
// the launch's 100 MB of traffic, and one wave doing the whole epilogue).  What the workgroup form costs: all slices of a
// tile live on one CU, so a tile has at most 8 slices (2 waves per SIMD; 4 for the tile shapes that need more than 256
// registers), and the planner counts workgroups per CU instead of waves per chip.
// All fragment loads are buffer loads with SCALAR offsets (one resource per operand array, lane * 16 as the only vector
// offset): a K step issues no vector instruction but its loads and its MFMAs.
//
// A map whose width is a few columns past a multiple of 32 (config 4: 33) would spend a whole column block on them.
// Those columns use the transposed ("V") form instead: for ONE map column x, D[y][frame] += A[y][k] B[k][frame] with
// A = template column-block kbv as a vertical Toeplitz matrix over image row rho (AtabV, built by the host) and B the
// very same Lg fragment: (rows + th - 1) * blocks MFMAs for 32 map rows of one column.
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "melf_device.h"
#include <hip/hip_ext.h>

#include "melf_internal.h"

namespace melf {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

struct GenGeom {
    int rh, rw;        // correlation map
    int rows;          // searched image rows (= rows of R per group)
    int rwp;           // row length of R in map columns (multiple of 32)
    int rows_pad;      // image rows per group in Lg (zero rows beyond the image)
    int nkb;           // 32-column blocks per image row in Lg
    int th, nd;        // template rows, Toeplitz blocks per template row
    int ndv, ndelta;   // V form: blocks per image row, rows of AtabV per column
    int vx0, vkb0;     // V form: first remainder column, its image block
    int nframes, ntiles;
    int rc, nxb_tile, nxb_h, nstrips, nhtiles, nvy;   // tile grid: rows computed per tile, column blocks per tile / of the H form, strips, H tiles, V tiles per column
    int atab_bytes, atabv_bytes;
    int k1;            // 128 * (sum T - 128 th tw)
    double tmean;
};

#ifdef MELF_GEN_STAMP
// Diagnostic build only (make stamp; never the shipped library): shader-clock stamps at the phase boundaries of each wave
// (tools/gen_clock.py).  Per wave: 8 stamps + {tile kind, wave index, workgroup id}.
__device__ uint64_t g_gen_stamps[12 * 16384];
extern "C" __attribute__((visibility("default"))) int melf_debug_gen_stamps(uint64_t* out, int nwaves)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gen_stamps), sizeof(uint64_t) * 12 * (size_t)(nwaves < 16384 ? nwaves : 16384)) == hipSuccess ? 0 : -1;
}
#define GSTAMP(k) do { const int gw_ = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); if ((threadIdx.x & 63) == 0 && gw_ < 16384) g_gen_stamps[12 * gw_ + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define GSTAMP_ID(kind) do { const int gw_ = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); if ((threadIdx.x & 63) == 0 && gw_ < 16384) { g_gen_stamps[12 * gw_ + 8] = (kind); g_gen_stamps[12 * gw_ + 9] = threadIdx.x >> 6; g_gen_stamps[12 * gw_ + 10] = blockIdx.x; g_gen_stamps[12 * gw_ + 11] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define GSTAMP(k) do { } while (0)
#define GSTAMP_ID(kind) do { } while (0)
#endif

__device__ inline bool better_g(float v, int i, float bv, int bi)
{
    return i != INT_MAX && (bi == INT_MAX || v > bv || (v == bv && i < bi));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t frag_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x27000);
}
// one 1 KiB operand fragment: lane * 16 bytes at a wave-uniform byte offset (an SGPR: no address arithmetic on the vector unit)
__device__ __forceinline__ i32x4 ldfrag(__amdgpu_buffer_rsrc_t rs, unsigned lane16, unsigned soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, soff, 0);
}

// ---- epilogue of one accumulator tile: exact u8 correlation, OpenCV's float post-pass, first maximum ----------
// elem(e) -> (y, x) of register e in this lane; wsv[e] = the window sum of L at that position.
template <class ELEM>
__device__ __forceinline__ void tile_epilogue(const i32x16& acc, const uint32_t* wsv, const GenGeom& g, int f, bool lane_ok, ELEM elem,
                                              float* __restrict__ result_map, float& bestv, int& besti)
{
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int y, x;
        elem(e, y, x);
        const bool valid = lane_ok && y < g.rh && x < g.rw;
        // cc = sum T L < 2^32 (checked by the host), so the three terms add modulo 2^32
        const uint32_t cc = (uint32_t)acc[e] + 128u * wsv[e] + (uint32_t)g.k1;
        double num = (double)cc;
        num -= (double)wsv[e] * g.tmean;
        const float v = valid ? (float)num : -INFINITY;
        const int idx = y * g.rw + x;
        if (result_map && valid) result_map[(size_t)f * g.rh * g.rw + idx] = v;
        // elements are NOT visited in raster order here: first maximum = greater value, or equal value at a smaller index
        if (valid && (besti == INT_MAX || v > bestv || (v == bestv && idx < besti))) { bestv = v; besti = idx; }
    }
}

// A workgroup's LDS (more than one wave): the tile's accumulators [block][16][64 lanes] i32 (block = row * nxb + column
// block; V form: one block), the tile's window sums in the same layout (u32), then one (max, arg-max) per wave and frame.
__device__ __forceinline__ int* lds_ws(int* s, int nblocks_max) { return s + nblocks_max * 1024; }
__device__ __forceinline__ MatchPartial* lds_best(int* s, int nblocks_max) { return (MatchPartial*)(s + 2 * nblocks_max * 1024); }

// 16 values of this lane added into a block of the workgroup's (LDS atomic adds without return: ds_add_u32)
template <class V>
__device__ __forceinline__ void lds_accumulate(int* s, int block, const V& v, int lane)
{
    int* p = s + block * 1024 + lane;
#pragma unroll
    for (int e = 0; e < 16; ++e) (void)__hip_atomic_fetch_add(p + e * 64, (int)v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ i32x16 lds_block(const int* s, int block, int lane)
{
    i32x16 a;
    const int* p = s + block * 1024 + lane;
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = p[e * 64];
    return a;
}

// the waves' (max, first arg-max) per frame -> the tile's partial
__device__ __forceinline__ void fold_and_write(int* s, int nblocks_max, float bestv, int besti, int grp, const GenGeom& g, int tile, int w, int ns,
                                               MatchPartial* __restrict__ partials)
{
    const int lane = threadIdx.x & 63;
    const float ov = __shfl_xor(bestv, 32, 64);
    const int oi = __shfl_xor(besti, 32, 64);
    if (better_g(ov, oi, bestv, besti)) { bestv = ov; besti = oi; }
    const int f = grp * 32 + (lane & 31);
    if (ns > 1) {
        MatchPartial* sb = lds_best(s, nblocks_max);
        if (lane < 32) { sb[w * 32 + lane].val = bestv; sb[w * 32 + lane].idx = besti; }
        __syncthreads();
        if (w != 0) return;
        for (int k = 1; k < ns; ++k) {
            const MatchPartial q = sb[k * 32 + (lane & 31)];
            if (better_g(q.val, q.idx, bestv, besti)) { bestv = q.val; besti = q.idx; }
        }
    }
    if (lane < 32 && f < g.nframes) {
        MatchPartial p;
        p.val = bestv;
        p.idx = besti;
        partials[(size_t)f * g.ntiles + tile] = p;
    }
}

// ---- window sums of TM_CCOEFF: sum over the template rows of the row-window sums R (k_prep_lplane) -------------------------
// R[group][row][piece = 2 xb + half][lane][8] u16: one 16-byte piece holds, for lane (n, hh), the sums of its accumulator
// elements e = 8 half .. + 7 of column block xb -- so a wave that adds pieces up has its window sums in EPILOGUE order.
__device__ __forceinline__ void piece_add(uint32_t* s8, const i32x4& v)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) { s8[2 * j] += (uint32_t)v[j] & 0xffffu; s8[2 * j + 1] += (uint32_t)v[j] >> 16; }
}
__device__ __forceinline__ void piece_sub(uint32_t* s8, const i32x4& v)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) { s8[2 * j] -= (uint32_t)v[j] & 0xffffu; s8[2 * j + 1] -= (uint32_t)v[j] >> 16; }
}

// H form, template rows [a, b) (this wave's share): emit(r, s) for the tile's rows r = 0 .. R - 1 with s[xb][e] = sum over
// i in [a, b) of R[y0 + r + i] -- the first row summed up, the others by sliding (minus the row that leaves, plus the one
// that enters).  Rows beyond the image are clamped (their map rows do not exist).
// FREE: the wave holds nothing else in registers (a slice before its K loop, or after its accumulators went to LDS): the
// rows of the sliding phase are requested together with the first sum's, so the wave waits for memory once or twice.
// Otherwise (a wave that owns a whole tile: its accumulators are live, emit() is a row's epilogue) the next row's two rows
// are requested before the current row's epilogue runs.
template <int R, int NXB, bool FREE, int PB /* 1 KiB pieces in flight at most */, class EMIT>
__device__ __forceinline__ void hform_winsums(__amdgpu_buffer_rsrc_t rsR, unsigned lane16, const GenGeom& g, const GenTile& t, int a, int b, EMIT emit)
{
    const unsigned rowbR = (unsigned)g.rwp * 64u;                 // bytes per row of R: rwp x 32 frames x 2 B
    const unsigned colb = (unsigned)t.xb0 * 2048u;
    auto row_off = [&](int row) { return (unsigned)min(row, g.rows - 1) * rowbR + colb; };
    constexpr int NP = 2 * NXB;                                    // 1 KiB pieces per row
    auto load_row = [&](i32x4* v, int row) {
        const unsigned o = row_off(row);
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = ldfrag(rsR, lane16, o + (unsigned)p * 1024u);
    };
    uint32_t s[NXB][16];
#pragma unroll
    for (int xb = 0; xb < NXB; ++xb)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[xb][e] = 0;
    // the sliding phase's rows: in[r - 1] = y0 + r - 1 + b enters, out[r - 1] = y0 + r - 1 + a leaves, r = 1 .. R - 1
    constexpr int NSL = FREE ? (R - 1) : 1;
    i32x4 vin[NSL > 0 ? NSL : 1][NP], vout[NSL > 0 ? NSL : 1][NP];
    if (FREE) {
#pragma unroll
        for (int r = 1; r < R; ++r) { load_row(vin[r - 1], t.y0 + r - 1 + b); load_row(vout[r - 1], t.y0 + r - 1 + a); }
    }
    constexpr int UBF = (PB - 2 * (R - 1) * NP) / NP;
    constexpr int UB = FREE ? (UBF < 2 ? 2 : (UBF > 16 ? 16 : UBF)) : 8 / NP;   // rows of the first sum in flight
    for (int i = a; i < b; i += UB) {
        i32x4 v[UB][NP];
#pragma unroll
        for (int u = 0; u < UB; ++u) load_row(v[u], t.y0 + min(i + u, b - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const bool in = i + u < b;    // wave-uniform: the last pass repeats row b - 1, not added
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                i32x4 q = v[u][p];
                if (!in) q = i32x4{0, 0, 0, 0};
                piece_add(&s[p >> 1][8 * (p & 1)], q);
            }
        }
    }
    if (!FREE && R > 1) { load_row(vin[0], t.y0 + b); load_row(vout[0], t.y0 + a); }
    emit(0, s);
#pragma unroll
    for (int r = 1; r < R; ++r) {
        constexpr int dummy = 0;
        (void)dummy;
        const int k = FREE ? r - 1 : 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            piece_add(&s[p >> 1][8 * (p & 1)], vin[k][p]);
            piece_sub(&s[p >> 1][8 * (p & 1)], vout[k][p]);
        }
        if (!FREE && r + 1 < R) { load_row(vin[0], t.y0 + r + b); load_row(vout[0], t.y0 + r + a); }
        emit(r, s);
    }
}

// V form, template rows [a, b): emit(r, S) for the 32 map rows r of the tile, S = this lane's FRAME's sum over i in [a, b) of
// R[yb + r + i] at the tile's map column (element c of the column block's first piece, lanes (n, 0)).  All the sliding
// phase's values (62 dwords) are requested up front, the first sum's in passes of 16.
template <class EMIT>
__device__ __forceinline__ void vform_winsums(__amdgpu_buffer_rsrc_t rsR, int lane, const GenGeom& g, const GenTile& t, int a, int b, EMIT emit)
{
    const unsigned rowbR = (unsigned)g.rwp * 64u;
    const int c = t.xb0;
    const unsigned voff = (unsigned)(lane & 31) * 16u + (unsigned)(c >> 1) * 4u;
    const unsigned sh = (unsigned)(c & 1) * 16u;
    const unsigned colb = (unsigned)g.vkb0 * 2048u;
    auto ld = [&](int row) -> uint32_t {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsR, voff, (unsigned)min(row, g.rows - 1) * rowbR + colb, 0);
    };
    uint32_t vin[31], vout[31];
#pragma unroll
    for (int r = 0; r < 31; ++r) { vin[r] = ld(t.y0 + r + b); vout[r] = ld(t.y0 + r + a); }
    uint32_t S = 0;
    for (int i = a; i < b; i += 16) {
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = ld(t.y0 + min(i + u, b - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) S += i + u < b ? ((v[u] >> sh) & 0xffffu) : 0u;
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        emit(r, S);
        if (r < 31) S += ((vin[r] >> sh) & 0xffffu) - ((vout[r] >> sh) & 0xffffu);
    }
}

// ---- H form: R map rows x NXB column blocks; this wave's slice [k_lo, k_hi) of the linearised (d, i) space ---------
template <int R, int NXB, int NBMAX, int PB>
__device__ __forceinline__ void gen_hform(const int8_t* __restrict__ Lg, const int8_t* __restrict__ Atab,
                                          const uint16_t* __restrict__ Rs, const GenGeom& g, const GenTile& t, int tile, int grp, int w, int ns,
                                          int* s_dyn, float* __restrict__ result_map, MatchPartial* __restrict__ partials)
{
    // Requests run PD steps ahead of their use; a step is R * NXB MFMAs (32 cycles each) and an L2 round trip under
    // load is ~1000 cycles, so small tiles need a longer lead.  Image rows and template fragments rotate through rings
    // of the same length NBUF = R + PD, which is the unroll period.
    // Round 4 (tools/gen_clock.py): at config 4 a 4 x 1 tile's K step took 213 cycles with 6 steps of lead -- the L2 round trip
    // under load, ~1300 cycles, divided by the lead -- against 128 cycles of MFMAs: small tiles are bound by the bytes they
    // keep in flight, so their lead is as long as the 256-register budget allows.
#ifndef MELF_GEN_PD_SMALL
#define MELF_GEN_PD_SMALL 12
#endif
    constexpr int PD = R * NXB >= 16 ? 2 : (R * NXB >= 12 ? 4 : (R * NXB >= 6 ? 6 : (NXB == 2 ? 8 : MELF_GEN_PD_SMALL)));
    constexpr int NBUF = R + PD;
    constexpr int PERIOD = NBUF;
    const int lane = threadIdx.x & 63;
    const unsigned lane16 = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t rsL = frag_rsrc(Lg + (size_t)grp * g.rows_pad * g.nkb * 1024, (unsigned)g.rows_pad * (unsigned)g.nkb * 1024u);
    const __amdgpu_buffer_rsrc_t rsA = frag_rsrc(Atab, (unsigned)g.atab_bytes);
    const __amdgpu_buffer_rsrc_t rsR = frag_rsrc(Rs + (size_t)grp * g.rows * g.rwp * 32, (unsigned)g.rows * (unsigned)g.rwp * 64u);
    const unsigned rowb = (unsigned)g.nkb * 1024u;  // bytes per image row of the group
    // this wave's share of the template rows for the window sums
    const int wa = (int)((long)w * g.th / ns), wb = (int)((long)(w + 1) * g.th / ns);
    // With several waves per SIMD, half of them add up their window sums BEFORE their K loop and half AFTER it, so that the
    // matrix pipes are not idle while every wave of the CU waits for row-window sums at the same time.
    // (waves w and w + 4 of a workgroup share a SIMD)
    const bool ws_first = ns > 1 && (ns > 4 ? ((w >> 2) & 1) == 0 : (w & 1) == 0);
    GSTAMP(1);
    if (ws_first)
        hform_winsums<R, NXB, true, PB>(rsR, lane16, g, t, wa, wb, [&](int r, uint32_t (*s)[16]) {
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) lds_accumulate(lds_ws(s_dyn, NBMAX), r * NXB + xb, s[xb], lane);
        });

    GSTAMP(2);
    i32x16 acc[R][NXB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int xb = 0; xb < NXB; ++xb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][xb][e] = 0;

    const int k_lo = (int)((long)w * t.klen / ns), k_hi = (int)((long)(w + 1) * t.klen / ns);
    i32x4 buf[NBUF][NXB];
    i32x4 a[NBUF];
    int k = k_lo;
    while (k < k_hi) {
        const int d = k / g.th, i_lo = k - d * g.th;
        const int i_hi = min(g.th, i_lo + (k_hi - k));
        const unsigned Lrow = (unsigned)(t.y0 + i_lo) * rowb + (unsigned)(d + t.xb0) * 1024u;
        // prime: image rows y0 + i_lo .. + NBUF - 2, template rows i_lo .. + PD - 1 (row th of Atab is all zero)
#pragma unroll
        for (int r = 0; r < NBUF - 1; ++r)
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) buf[r][xb] = ldfrag(rsL, lane16, Lrow + (unsigned)r * rowb + (unsigned)xb * 1024u);
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            const int irow = i_lo + q < i_hi ? i_lo + q : g.th;
            a[q] = ldfrag(rsA, lane16, (unsigned)(irow * g.nd + d) * 1024u);
        }
        unsigned Lnext = Lrow + (unsigned)(NBUF - 1) * rowb;  // image row of the next request
        // Whole periods only, and no branch inside one: hipcc sinks loads across block boundaries to their first use
        // (request -> wait a full L2 round trip -> use), whatever sched_barrier says, but keeps them pinned inside a
        // block.  Steps past i_hi multiply the zero template row.
        for (int i0 = i_lo; i0 < i_hi; i0 += PERIOD) {
#pragma unroll
            for (int s = 0; s < PERIOD; ++s) {
                const int i = i0 + s;
                // requests for PD steps ahead, pinned in front of this step's MFMAs (Lg has zero rows past the image, and
                // a buffer load beyond the group's rows returns zero)
#pragma unroll
                for (int xb = 0; xb < NXB; ++xb) buf[(s + NBUF - 1) % NBUF][xb] = ldfrag(rsL, lane16, Lnext + (unsigned)xb * 1024u);
                Lnext += rowb;
                const int irow = i + PD < i_hi ? i + PD : g.th;  // scalar select, no branch
                a[(s + PD) % NBUF] = ldfrag(rsA, lane16, (unsigned)(irow * g.nd + d) * 1024u);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int xb = 0; xb < NXB; ++xb)
                        acc[r][xb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s % NBUF], buf[(s + r) % NBUF][xb], acc[r][xb], 0, 0, 0);
                // the step's NXB + 1 requests go out one after each of its first matrix instructions (a 1 KiB load takes the issue port
                // for half a matrix instruction's time: one per instruction is free, a cluster in front of the step is not)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if constexpr (NXB == 2) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        k += i_hi - i_lo;
    }

    GSTAMP(3);
    const int n = lane & 31, hh = lane >> 5;
    const int f = grp * 32 + n;
    const bool lane_ok = f < g.nframes;
    float bestv = -INFINITY;
    int besti = INT_MAX;
    if (ns == 1) {
        // the wave owns the whole tile: its window sums slide down the tile's rows in registers, the epilogue of a row runs
        // as soon as its sums are complete; no LDS
        hform_winsums<R, NXB, false, PB>(rsR, lane16, g, t, 0, g.th, [&](int r, uint32_t (*s)[16]) {
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) {
                const int y = t.y0 + r, xbase = 32 * (t.xb0 + xb) + 4 * hh;
                tile_epilogue(acc[r][xb], s[xb], g, f, lane_ok && r < t.R, [&](int e, int& yy, int& xx) { yy = y; xx = xbase + (e & 3) + 8 * (e >> 2); },
                              result_map, bestv, besti);
            }
        });
    } else {
        // the slices add up in LDS (zeroed by the workgroup before the K loops); then the tile's row blocks are dealt out
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) lds_accumulate(s_dyn, r * NXB + xb, acc[r][xb], lane);
        if (!ws_first)
            hform_winsums<R, NXB, true, PB>(rsR, lane16, g, t, wa, wb, [&](int r, uint32_t (*s)[16]) {
#pragma unroll
                for (int xb = 0; xb < NXB; ++xb) lds_accumulate(lds_ws(s_dyn, NBMAX), r * NXB + xb, s[xb], lane);
            });
        GSTAMP(4);
        __syncthreads();
        GSTAMP(5);
        const int nblk = t.R * NXB;   // rows past the map are not looked at
        for (int b = w; b < nblk; b += ns) {
            const int r = b / NXB, xb = b - r * NXB;
            const i32x16 sum = lds_block(s_dyn, b, lane);
            const i32x16 wsum = lds_block(lds_ws(s_dyn, NBMAX), b, lane);
            uint32_t wsv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) wsv[e] = (uint32_t)wsum[e];
            const int y = t.y0 + r, xbase = 32 * (t.xb0 + xb) + 4 * hh;
            tile_epilogue(sum, wsv, g, f, lane_ok, [&](int e, int& yy, int& xx) { yy = y; xx = xbase + (e & 3) + 8 * (e >> 2); }, result_map, bestv,
                          besti);
        }
    }
    GSTAMP(6);
    fold_and_write(s_dyn, NBMAX, bestv, besti, grp, g, tile, w, ns, partials);
    GSTAMP(7);
}

// ---- V form: one map column, 32 map rows; this wave's slice [k_lo, k_hi) of the linearised (rho - 32 yb, kbv) space -------
template <int NBMAX>
__device__ __forceinline__ void gen_vform(const int8_t* __restrict__ Lg, const int8_t* __restrict__ AtabV,
                                          const uint16_t* __restrict__ Rs, const GenGeom& g, const GenTile& t, int tile, int grp, int w, int ns,
                                          int* s_dyn, float* __restrict__ result_map, MatchPartial* __restrict__ partials)
{
    const int lane = threadIdx.x & 63;
    const unsigned lane16 = (unsigned)lane * 16u;
    const int c = t.xb0;                 // remainder column index
    const int yb = t.y0;                 // first of the 32 map rows
    const __amdgpu_buffer_rsrc_t rsL = frag_rsrc(Lg + (size_t)grp * g.rows_pad * g.nkb * 1024, (unsigned)g.rows_pad * (unsigned)g.nkb * 1024u);
    const __amdgpu_buffer_rsrc_t rsV = frag_rsrc(AtabV, (unsigned)g.atabv_bytes);
    const __amdgpu_buffer_rsrc_t rsR = frag_rsrc(Rs + (size_t)grp * g.rows * g.rwp * 32, (unsigned)g.rows * (unsigned)g.rwp * 64u);
    const unsigned rowb = (unsigned)g.nkb * 1024u;
    const unsigned vbase = (unsigned)(c * g.ndelta * g.ndv) * 1024u;
    const unsigned lbase = (unsigned)yb * rowb + (unsigned)g.vkb0 * 1024u;
    const int hh = lane >> 5;
    // row r of the tile is element (r & 3) + 4 (r >> 3) of the lanes with hh = (r >> 2) & 1
    GSTAMP(1);
    const bool ws_first = ns > 1 && (ns > 4 ? ((w >> 2) & 1) == 0 : (w & 1) == 0);   // as in gen_hform
    auto ws_share = [&]() {
        const int wa = (int)((long)w * g.th / ns), wb = (int)((long)(w + 1) * g.th / ns);
        int* wl = lds_ws(s_dyn, NBMAX) + lane;
        vform_winsums(rsR, lane, g, t, wa, wb, [&](int r, uint32_t S) {
            if (hh == ((r >> 2) & 1)) (void)__hip_atomic_fetch_add(wl + ((r & 3) + 4 * (r >> 3)) * 64, (int)S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        });
    };
    if (ws_first) ws_share();
    GSTAMP(2);
    i32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0;
    const int k_lo = (int)((long)w * t.klen / ns), k_hi = (int)((long)(w + 1) * t.klen / ns);
    constexpr int U = 8;  // MFMAs per group; two groups of operands in flight (two loads per MFMA: the loop lives on load latency)
    i32x4 av[2][U], bv[2][U];
    // position of the next request: q = delta * ndv + kbv, kept as (q, delta, kbv) and advanced without divisions
    int rq = k_lo, rdelta = k_lo / g.ndv, rkbv = k_lo - rdelta * g.ndv;
#define MELF_V_REQUEST(SET)                                                                                    \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                            \
        /* past the slice's end the operands are whatever lies there: the A fragment is zeroed when consumed */ \
        av[SET][u] = ldfrag(rsV, lane16, vbase + (unsigned)rq * 1024u);                                         \
        bv[SET][u] = ldfrag(rsL, lane16, lbase + (unsigned)rdelta * rowb + (unsigned)rkbv * 1024u);             \
        ++rq; ++rkbv;                                                                                          \
        if (rkbv == g.ndv) { rkbv = 0; ++rdelta; }                                                             \
    }
#define MELF_V_CONSUME(KK, SET)                                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                        \
        i32x4 aa = av[SET][u];                                                                             \
        if ((KK) + u >= k_hi) aa = i32x4{0, 0, 0, 0};                                                      \
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bv[SET][u], acc, 0, 0, 0);                         \
    }
    MELF_V_REQUEST(0)
    for (int kk = k_lo; kk < k_hi; kk += 2 * U) {  // one block per iteration: no branch inside (see gen_hform)
        MELF_V_REQUEST(1)
        __builtin_amdgcn_sched_barrier(0);
        MELF_V_CONSUME(kk, 0)
        __builtin_amdgcn_sched_barrier(0);
        MELF_V_REQUEST(0)
        __builtin_amdgcn_sched_barrier(0);
        MELF_V_CONSUME(kk + U, 1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef MELF_V_REQUEST
#undef MELF_V_CONSUME
    GSTAMP(3);
    const int f = grp * 32 + (lane & 31);
    float bestv = -INFINITY;
    int besti = INT_MAX;
    const int x = g.vx0 + c;
    auto elem = [&](int e, int& yy, int& xx) { yy = yb + (e & 3) + 8 * (e >> 2) + 4 * hh; xx = x; };
    if (ns == 1) {
        uint32_t wsv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) wsv[e] = 0;
        vform_winsums(rsR, lane, g, t, 0, g.th, [&](int r, uint32_t S) {
            if (hh == ((r >> 2) & 1)) wsv[(r & 3) + 4 * (r >> 3)] = S;
        });
        tile_epilogue(acc, wsv, g, f, f < g.nframes, elem, result_map, bestv, besti);
    } else {
        lds_accumulate(s_dyn, 0, acc, lane);
        if (!ws_first) ws_share();
        GSTAMP(4);
        __syncthreads();
        GSTAMP(5);
        if (w == 0) {
            const i32x16 sum = lds_block(s_dyn, 0, lane);
            const i32x16 wsum = lds_block(lds_ws(s_dyn, NBMAX), 0, lane);
            uint32_t wsv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) wsv[e] = (uint32_t)wsum[e];
            tile_epilogue(sum, wsv, g, f, f < g.nframes, elem, result_map, bestv, besti);
        }
    }
    GSTAMP(6);
    fold_and_write(s_dyn, NBMAX, bestv, besti, grp, g, tile, w, ns, partials);
    GSTAMP(7);
}

// One kernel per tile shape class: RC rows computed (2 / 4 / 6 / 8), at most NXBMAX column blocks per tile, at most NSMAX
// waves (K slices) per workgroup -- the launch bound is what lets the small shapes run two waves per SIMD (<= 256 registers)
// while the large ones keep all 512.
template <int RC, int NXBMAX, int NSMAX>
__global__ __launch_bounds__(64 * NSMAX) void k_match_gen(const int8_t* __restrict__ Lg, const int8_t* __restrict__ Atab,
                                                          const int8_t* __restrict__ AtabV, const uint16_t* __restrict__ Rs,
                                                          const GenTile* __restrict__ tiles, GenGeom g, float* __restrict__ result_map,
                                                          MatchPartial* __restrict__ partials)
{
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    constexpr int NBMAX = RC * NXBMAX;
    // XCD-aware order (as k_match_mfma): consecutive virtual ids share an XCD, so a frame group's tiles share an L2
    const int nblk = gridDim.x, id = blockIdx.x;
    const int per = nblk / 8, rem = nblk % 8, xcd = id & 7, sub = id >> 3;
    const int vid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + sub;
    const int grp = vid / g.ntiles, ti = vid - grp * g.ntiles;
    // the tile from its index (the same arithmetic as gen_plan's list, which the host keeps for queries: no descriptor load
    // in front of the kernel's first request)
    GenTile t;
    if (ti < g.nhtiles) {
        const int tr = ti / g.nstrips, ts = ti - tr * g.nstrips;
        t.y0 = (int16_t)(tr * g.rc); t.Rc = (int8_t)g.rc; t.R = (int8_t)min(g.rc, g.rh - tr * g.rc);
        t.xb0 = (int16_t)(ts * g.nxb_tile); t.nxb = (int8_t)min(g.nxb_tile, g.nxb_h - ts * g.nxb_tile);
        t.klen = g.nd * g.th;
    } else {
        const int tv = ti - g.nhtiles, c = tv / g.nvy, yb = (tv - c * g.nvy) * 32;
        t.y0 = (int16_t)yb; t.R = 0; t.Rc = 0; t.nxb = 0; t.xb0 = (int16_t)c;
        t.klen = (min(32, g.rh - yb) + g.th - 1) * g.ndv;
    }
    t.pad0 = 0;
    (void)tiles;
    const int ns = (int)(blockDim.x >> 6);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    GSTAMP(0);
    GSTAMP_ID(t.R == 0 ? 0 : 1);
    if (ns > 1) {   // the workgroup's accumulator and window-sum tiles start at zero (the waves ADD into them)
        const int nb = t.R ? (int)t.Rc * (int)t.nxb : 1;
        i32x4* z = (i32x4*)s_dyn;
        i32x4* zw = (i32x4*)lds_ws(s_dyn, NBMAX);
        for (int i = threadIdx.x; i < nb * 256; i += blockDim.x) { z[i] = i32x4{0, 0, 0, 0}; zw[i] = i32x4{0, 0, 0, 0}; }
        __syncthreads();
    }
    if (t.R == 0) gen_vform<NBMAX>(Lg, AtabV, Rs, g, t, ti, grp, w, ns, s_dyn, result_map, partials);
    else if (NXBMAX == 2 && t.nxb == 2) gen_hform<RC, NXBMAX, NBMAX, (NSMAX == 8 ? 36 : 64)>(Lg, Atab, Rs, g, t, ti, grp, w, ns, s_dyn, result_map, partials);
    else gen_hform<RC, 1, NBMAX, (NSMAX == 8 ? 36 : 64)>(Lg, Atab, Rs, g, t, ti, grp, w, ns, s_dyn, result_map, partials);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static const int GEN_VREM_MAX = 4;    // remainder columns that go through the V form
static const int GEN_SLICE_MIN = 24;  // template rows per slice at least (priming a slice costs ~3 steps)

// waves a workgroup of this tile shape may have: the small shapes are compiled for two waves per SIMD (8 per CU)
static int gen_nsmax(int rc, int nxb) { return rc * nxb <= 4 ? 8 : 4; }

bool gen_match_ok(int th, int tw, int rows, int cols)
{
    const int rh = rows - th + 1, rw = cols - tw + 1;
    // u16 row-window sums (tw * 255 < 2^16) and the epilogue's modulo-2^32 sum (th * tw * 255^2 < 2^32)
    if (!(rh >= 1 && rw >= 1 && tw <= 256 && (long)th * tw * 65025L < (1L << 32) && rows < 32768 && cols < 32768)) return false;
    // What the launches can hold (shapes beyond go to the VALU kernel, which has no such limits):
    //  * k_prep_lplane keeps the row prefix of 32 frames in LDS, 64 * (32 nkb + 8) bytes of the 128 KiB it may ask for;
    //  * GenTile::y0 is 16-bit, a group's L plane is addressed with 32-bit byte offsets, and the number of tiles is bounded
    //    by its worst case (2-row tiles of one column block).
    const int nd = (tw + 62) / 32, nxb_full = rw / 32, rem = rw % 32;
    const int vcols = (rem > 0 && rem <= GEN_VREM_MAX && nxb_full >= 1) ? rem : 0;
    const int nxb_h = vcols ? nxb_full : nxb_full + (rem ? 1 : 0);
    const int ndv = vcols ? (vcols - 1 + tw - 1) / 32 + 1 : 0;
    const int nkb = std::max(std::max(nxb_h + nd - 1, vcols ? nxb_full + ndv : 0), (cols + 31) / 32);
    if (64L * (32L * nkb + 8) > 128L * 1024) return false;
    if ((long)(rows + th + 64) * nkb * 1024L >= (1L << 32)) return false;
    const long ntiles_max = (long)((rh + 1) / 2) * std::max(nxb_h, 1) + (long)vcols * ((rh + 31) / 32);
    return ntiles_max <= 8000;
}

GenPlan gen_plan(int th, int tw, int rows, int cols, int nframes)
{
    GenPlan p;
    p.rh = rows - th + 1;
    p.rw = cols - tw + 1;
    p.nd = (tw + 62) / 32;
    p.groups = (nframes + 31) / 32;
    int nxb_full = p.rw / 32;
    const int rem = p.rw % 32;
    p.vcols = (rem > 0 && rem <= GEN_VREM_MAX && nxb_full >= 1) ? rem : 0;
    const int nxb_h = p.vcols ? nxb_full : nxb_full + (rem ? 1 : 0);  // column blocks of the H form
    p.vx0 = 32 * nxb_full;
    p.vkb0 = nxb_full;
    p.ndv = p.vcols ? (p.vcols - 1 + tw - 1) / 32 + 1 : 0;
    p.ndelta = 32 + th - 1;
    p.rwp = 32 * (nxb_full + (rem ? 1 : 0));
    p.nkb = std::max(nxb_h + p.nd - 1, p.vcols ? p.vkb0 + p.ndv : 0);
    p.nkb = std::max(p.nkb, (cols + 31) / 32);

    // ---- tile shape and K slices: a small search over (rows per tile, column blocks per tile, slices = waves per workgroup) ----
    // Cost model in shader cycles, fitted to launches on MI355X (tools/gen_shape_sweep.py, tools/gen_clock.py,
    // profiles/r04/gen_shape_sweep_*.txt, gen_clock_config4.txt).  A workgroup = one tile of one frame group, its waves = the
    // K slices.  A CU holds 8 waves of the small tile shapes (<= 256 registers) and 4 of the others, and whole workgroups only;
    // the launch runs in rounds of what the 256 CUs hold.  What the stamps say about a K step: it costs a wave ~35 cycles per
    // 1 KiB fragment load PLUS 32 per MFMA (the two do not overlap within a wave), and a CU's waves together get one fragment
    // per ~27 cycles out of its vector memory path (38 bytes per cycle: 4 x 1 tiles at config 4 run exactly there, whatever
    // the prefetch distance and with 4 or 8 waves per CU), and four SIMDs' worth of MFMAs.  Around the K loop: LDS zeroing,
    // the wave's share of the window sums (passes of <= 36 pieces, ~3 500 cycles each: R comes from beyond L2), the LDS
    // reduction, its share of the tile's epilogue (~4 200 cycles per row block), the fold.
    const int nvt = p.vcols * ((p.rh + 31) / 32);          // V-form tiles per group
    const long ksteps = (long)p.nd * th;
    const long vsteps = (long)(std::min(32, p.rh) + th - 1) * std::max(p.ndv, 1);
    double best_cost = 1e30;
    int best_rc = 8, best_nxb = 1, best_ns = 1;
    for (int rc = 8; rc >= 2; rc -= 2)
        for (int nx = 2; nx >= 1; --nx) {
            if (nx > nxb_h && nx > 1) continue;
            if (rc == 8 && nx == 2) continue;   // 256 accumulator registers + the operand rings do not fit 512 without spilling
            const int ntr = (p.rh + rc - 1) / rc, nstr = (nxb_h + nx - 1) / nx;
            const long ntile = (long)ntr * nstr + nvt;
            const int wpc = gen_nsmax(rc, nx) == 8 ? 8 : 4;      // waves per CU of this register class
            const int pd = rc * nx >= 16 ? 2 : (rc * nx >= 12 ? 4 : (rc * nx >= 6 ? 6 : (nx == 2 ? 8 : 12)));
            for (int ns = 1; ns <= gen_nsmax(rc, nx); ++ns) {
                if (ns > 1 && ksteps / ns < GEN_SLICE_MIN) break;
                const long wgs = ntile * p.groups;
                const int lds = ns > 1 ? 2 * rc * nx * 4096 + ns * 256 : 0;
                const int wg_per_cu = std::max(1, std::min(wpc / ns, lds ? (160 * 1024) / lds : 16));
                const long rounds = (wgs + 256L * wg_per_cu - 1) / (256L * wg_per_cu);
                const long on_cu = std::min<long>(wg_per_cu, (wgs + 255) / 256);     // workgroups sharing a CU in a round
                const int dpasses = (ns >= p.nd) ? 2 : (p.nd + ns - 1) / ns + (ns > 1 ? 1 : 0);
                const double wave_k = (double)((ksteps + ns - 1) / ns + (rc + pd) / 2 * dpasses) * (35.0 * (nx + 1) + 32.0 * rc * nx) + 300.0 * dpasses;
                const double cu_loads = (double)on_cu * ksteps * (nx + 1) * 27.0;
                const double cu_mfma = (double)on_cu * ksteps * rc * nx * 32.0 / 4.0;
                double tk = std::max(wave_k, std::max(cu_loads, cu_mfma));
                if (nvt) tk = std::max(tk, (double)((vsteps + ns - 1) / ns) * 220.0);   // a V-form step: two loads per MFMA
                const int ws_pieces = (int)((th + ns - 1) / ns + 2 * (rc - 1)) * 2 * nx;
                const double ws_cyc = 1500.0 + 3500.0 * ((ws_pieces + 35) / 36) + 9.0 * ws_pieces;
                const double tail = ns > 1 ? 3000.0 + 12.0 * rc * nx * 16 + 4200.0 * ((rc * nx + ns - 1) / ns) + 2500.0
                                           : 2600.0 * rc * nx;   // a lone wave: the rows' epilogues run under the sliding window sums
                const double cost = rounds * (tk + ws_cyc + tail + 4000.0 + 900.0 * on_cu * ns);   // + what every further wave of a CU adds (LDS atomics, barrier skew)
                if (cost < best_cost * 0.98) { best_cost = cost; best_rc = rc; best_nxb = nx; best_ns = ns; }
            }
        }
    if (const char* e = getenv("MELF_GEN_SHAPE")) {  // experiments: "rc,nxb,ns"
        int a = 0, b = 0, c2 = 0;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c2) == 3 && a >= 2 && a <= 8 && a % 2 == 0 && b >= 1 && b <= 2 && c2 >= 1 && !(a == 8 && b == 2)) {
            best_rc = a; best_nxb = std::min(b, std::max(nxb_h, 1)); best_ns = std::min(c2, gen_nsmax(best_rc, best_nxb));
        }
    }
    const int Rc = best_rc;
    p.rc = Rc;
    p.nxb_tile = best_nxb;
    p.nslices = best_ns;
    p.nxb_h = nxb_h;
    int max_row_used = 0;
    for (int y0 = 0; y0 < p.rh; y0 += Rc) {
        const int R = std::min(Rc, p.rh - y0);
        for (int xb = 0; xb < nxb_h; xb += best_nxb) {
            const int nxb = std::min(best_nxb, nxb_h - xb);
            GenTile t;
            t.y0 = (int16_t)y0; t.R = (int8_t)R; t.Rc = (int8_t)Rc; t.nxb = (int8_t)nxb; t.pad0 = 0; t.xb0 = (int16_t)xb;
            t.klen = p.nd * th;
            p.tiles.push_back(t);
            max_row_used = std::max(max_row_used, y0 + Rc);
        }
    }
    for (int c = 0; c < p.vcols; ++c)
        for (int yb = 0; yb < p.rh; yb += 32) {
            const int nrow = std::min(32, p.rh - yb);
            GenTile t;
            t.y0 = (int16_t)yb; t.R = 0; t.Rc = 0; t.nxb = 0; t.pad0 = 0; t.xb0 = (int16_t)c;
            t.klen = (nrow + th - 1) * p.ndv;
            p.tiles.push_back(t);
        }
    p.ntiles = (int)p.tiles.size();
    p.ntasks = p.ntiles * p.nslices;
    p.lds_bytes = p.nslices > 1 ? 2 * (size_t)Rc * best_nxb * 4096 + (size_t)gen_nsmax(Rc, best_nxb) * 32 * sizeof(MatchPartial) : 0;
    // last image row a wave asks for: the K loop runs whole periods of NBUF = Rc + PD steps and requests NBUF - 1 rows ahead
    // (a request beyond the group's rows returns zero: the buffer resource ends there)
    p.rows_pad = std::max(rows, max_row_used + th + 2 * (Rc + 8)) + 1;
    if (p.vcols) p.rows_pad = std::max(p.rows_pad, ((p.rh + 31) / 32) * 32 + th);
    p.lg_bytes = (size_t)p.groups * p.rows_pad * p.nkb * 1024;
    p.r_bytes = (size_t)p.groups * rows * p.rwp * 32 * sizeof(uint16_t);
    p.atab_bytes = (size_t)(th + 1) * p.nd * 1024;
    p.atabv_bytes = (size_t)p.vcols * p.ndelta * p.ndv * 1024;
    return p;
}

// Atab[i][d][lane][j] = T'[i][32 d + 16 (lane >> 5) + j - (lane & 31)], zero outside the template; one zero row appended
void gen_build_atab(const uint8_t* templ, int th, int tw, const GenPlan& p, int8_t* atab)
{
    for (int i = 0; i < th + 1; ++i)
        for (int d = 0; d < p.nd; ++d)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 16; ++j) {
                    const int col = 32 * d + 16 * (l >> 5) + j - (l & 31);
                    int8_t v = 0;
                    if (i < th && col >= 0 && col < tw) v = (int8_t)((int)templ[(size_t)i * tw + col] - 128);
                    atab[(((size_t)i * p.nd + d) * 64 + l) * 16 + j] = v;
                }
}

// AtabV[c][delta][kbv][lane][j] = T'[delta - (lane & 31)][32 kbv + 16 (lane >> 5) + j - c]: map column x = vx0 + c, map row
// 32 yb + (lane & 31), image row 32 yb + delta, image columns 32 (vkb0 + kbv) ..
void gen_build_atabv(const uint8_t* templ, int th, int tw, const GenPlan& p, int8_t* atabv)
{
    for (int c = 0; c < p.vcols; ++c)
        for (int dl = 0; dl < p.ndelta; ++dl)
            for (int kb = 0; kb < p.ndv; ++kb)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 16; ++j) {
                        const int row = dl - (l & 31), col = 32 * kb + 16 * (l >> 5) + j - c;
                        int8_t v = 0;
                        if (row >= 0 && row < th && col >= 0 && col < tw) v = (int8_t)((int)templ[(size_t)row * tw + col] - 128);
                        atabv[((((size_t)c * p.ndelta + dl) * p.ndv + kb) * 64 + l) * 16 + j] = v;
                    }
}

template <int RC, int NXBMAX, int NSMAX>
static void launch_gen(dim3 grid, dim3 block, size_t lds, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, const int8_t* d_lg,
                       const GenDev& dev, const uint16_t* d_r, const GenGeom& g, float* d_result_map, MatchPartial* d_partials)
{
    if (lds > 48 * 1024) {   // once per device and instantiation: dynamic LDS beyond the default limit
        static bool attr_set[64] = {false};
        int devid = 0;
        (void)hipGetDevice(&devid);
        if (devid >= 0 && devid < 64 && !attr_set[devid]) {
            (void)hipFuncSetAttribute((const void*)k_match_gen<RC, NXBMAX, NSMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            attr_set[devid] = true;
        }
    }
    hipExtLaunchKernelGGL((k_match_gen<RC, NXBMAX, NSMAX>), grid, block, (unsigned)lds, stream, ev_start, ev_stop, 0, d_lg, dev.atab, dev.atabv, d_r,
                          dev.tiles, g, d_result_map, d_partials);
}

void launch_gen_match(int n, const GenPlan& p, int rows, int th, int tw, long tsum, double tmean, const GenDev& dev, const int8_t* d_lg,
                      const uint16_t* d_r, float* d_result_map, MatchPartial* d_partials, hipStream_t stream, hipEvent_t ev_start,
                      hipEvent_t ev_stop)
{
    GenGeom g;
    g.rh = p.rh; g.rw = p.rw; g.rows = rows; g.rwp = p.rwp; g.rows_pad = p.rows_pad; g.nkb = p.nkb; g.th = th; g.nd = p.nd;
    g.ndv = p.ndv; g.ndelta = p.ndelta; g.vx0 = p.vx0; g.vkb0 = p.vkb0;
    g.nframes = n; g.ntiles = p.ntiles;
    g.rc = p.rc; g.nxb_tile = p.nxb_tile; g.nxb_h = p.nxb_h; g.nstrips = (p.nxb_h + p.nxb_tile - 1) / p.nxb_tile;
    g.nvy = (p.rh + 31) / 32; g.nhtiles = p.ntiles - p.vcols * g.nvy;
    g.atab_bytes = (int)p.atab_bytes; g.atabv_bytes = (int)p.atabv_bytes;
    g.k1 = (int)(128 * (tsum - 128L * th * tw));
    g.tmean = tmean;
    dim3 grid(p.ntiles * p.groups), block(64 * p.nslices);
#define MELF_GEN_CASE(RC, NX, NS) \
    case (RC) * 4 + (NX): launch_gen<RC, NX, NS>(grid, block, p.lds_bytes, stream, ev_start, ev_stop, d_lg, dev, d_r, g, d_result_map, d_partials); break;
    switch (p.rc * 4 + p.nxb_tile) {
        MELF_GEN_CASE(2, 1, 8) MELF_GEN_CASE(2, 2, 8) MELF_GEN_CASE(4, 1, 8)
        MELF_GEN_CASE(4, 2, 4) MELF_GEN_CASE(6, 1, 4) MELF_GEN_CASE(6, 2, 4) MELF_GEN_CASE(8, 1, 4)
        default:
            fprintf(stderr, "[melf] k_match_gen: no instantiation for %d-row tiles of %d column blocks\n", p.rc, p.nxb_tile);
            abort();
    }
#undef MELF_GEN_CASE
}

}  // namespace melf
