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
// blocks d + xb of R + 2 rows (<= 80 VGPRs), so the number of blocks, the strip of the map and the template-row
// range are all runtime values, and R can be 8 rows (16 accumulator tiles): 3 loads per 16 MFMAs.
//
// Work is cut into TILES (R <= 8 map rows x NXB <= 2 column blocks x 32 frames); a tile's K range (d, i) may be cut
// into SLICES handled by different waves when the map is too small to fill the chip otherwise (BASELINE config 4:
// 17 x 33 positions).  Slices add up through global partial tiles; the wave whose arrival completes a tile (a
// returning atomic on the tile's counter, release / acquire fences at agent scope) sums them and runs the epilogue.
// Nobody spins, so no placement or residency assumption is made.
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
    int rwp;           // row length of R / ws in map columns (multiple of 32)
    int rows_pad;      // image rows per group in Lg (zero rows beyond the image)
    int nkb;           // 32-column blocks per image row in Lg
    int th, nd;        // template rows, Toeplitz blocks per template row
    int ndv, ndelta;   // V form: blocks per image row, rows of AtabV per column
    int vx0, vkb0;     // V form: first remainder column, its image block
    int nframes, ntasks, ntiles;
    int k1;            // 128 * (sum T - 128 th tw)
    double tmean;
};

__device__ inline bool better_g(float v, int i, float bv, int bi)
{
    return i != INT_MAX && (bi == INT_MAX || v > bv || (v == bv && i < bi));
}

// ---- slices: partial tiles through global memory ------------------------------------------------------------
// part[(slice * NQ + q) * 64 + lane] = 4 consecutive accumulator registers (i32x4), q = tile * 4 + quarter.
// Hand-off without L2 write-back / invalidate fences (each costs microseconds once a wave has tens of KiB dirty, and
// every slice would pay them): the partial tiles are written with write-through (sc1) 16-byte stores, drained with
// vmcnt(0), then ONE lane counts the wave's arrival with an agent-scope atomic add on the tile's counter; the wave
// whose add returns nslices - 1 is the last one and reads the other slices with sc1 loads (served past its CU's L1),
// only after its add has returned.  MI355X_MICROARCH.md, "Workgroup dispatch ... visibility", hand-off table row 1.
// Nobody polls.  The buffer intrinsics keep the loads under the compiler's own vmcnt bookkeeping.
#define MELF_SC1 16  // cache-policy bit of the raw buffer intrinsics on gfx94x / gfx950
__device__ __forceinline__ __amdgpu_buffer_rsrc_t part_rsrc(i32x4* part, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)bytes, 0x27000);
}
__device__ __forceinline__ void part_store(const i32x16& acc, __amdgpu_buffer_rsrc_t rs, unsigned slice_base /* bytes, lane included */, int tile_q)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, slice_base + (unsigned)(tile_q * 4 + q) * 1024u, 0, MELF_SC1);
    }
}
__device__ __forceinline__ bool part_arrive(int* __restrict__ counter, int nslices)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every lane's write-through stores have been acknowledged
    int old = 0;
    if (threadIdx.x == 0) old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != nslices - 1) return false;
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    return true;
}
// The completing wave adds the other slices' partial tiles: TB tiles (4 TB KiB per slice) from up to four slices per
// batch, i.e. up to 16 TB loads in flight -- one load at a time is a memory round trip per KiB.
template <int NT, int TB>
__device__ __forceinline__ void part_add_all(i32x16* acc /* NT tiles */, __amdgpu_buffer_rsrc_t rs, unsigned tile_base /* bytes, lane included */,
                                             int slice, int nslices)
{
    constexpr int NQ = NT * 4;
    for (int s0 = 0; s0 < nslices; s0 += 4) {
#pragma unroll
        for (int tb = 0; tb < NT; tb += TB) {
            i32x4 v[4][TB * 4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sj = s0 + j;
                const int src = (sj < nslices && sj != slice) ? sj : slice;  // surplus: this wave's own tile, not added below
                const unsigned other = tile_base + (unsigned)(src * NQ + tb * 4) * 1024u;
#pragma unroll
                for (int q = 0; q < TB * 4; ++q)
                    if (tb * 4 + q < NQ) v[j][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, other + (unsigned)q * 1024u, 0, MELF_SC1);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sj = s0 + j;
                const int keep = (sj < nslices && sj != slice) ? -1 : 0;
#pragma unroll
                for (int q = 0; q < TB * 4; ++q)
                    if (tb * 4 + q < NQ) {
                        i32x16& A = acc[tb + q / 4];
                        A[4 * (q & 3)] += v[j][q].x & keep; A[4 * (q & 3) + 1] += v[j][q].y & keep;
                        A[4 * (q & 3) + 2] += v[j][q].z & keep; A[4 * (q & 3) + 3] += v[j][q].w & keep;
                    }
            }
        }
    }
}

// ---- epilogue of one accumulator tile: exact u8 correlation, OpenCV's float post-pass, first maximum ----------
// elem(e) -> (y, x) of register e in this lane.
template <class ELEM>
__device__ __forceinline__ void tile_epilogue(const i32x16& acc, const uint32_t* __restrict__ ws, const GenGeom& g, int grp, int f,
                                              bool lane_ok, ELEM elem, float* __restrict__ result_map, float& bestv, int& besti)
{
    uint32_t wsv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int y, x;
        elem(e, y, x);
        const int yc = min(y, g.rh - 1), xc = min(x, g.rwp - 1);
        wsv[e] = ws[(((size_t)grp * g.rh + yc) * g.rwp + xc) * 32 + (f & 31)];
    }
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

__device__ __forceinline__ void write_partial(float bestv, int besti, int f, const GenGeom& g, int tile, MatchPartial* __restrict__ partials)
{
    const int lane = threadIdx.x;
    const float ov = __shfl_xor(bestv, 32, 64);
    const int oi = __shfl_xor(besti, 32, 64);
    if (better_g(ov, oi, bestv, besti)) { bestv = ov; besti = oi; }
    if (lane < 32 && f < g.nframes) {
        MatchPartial p;
        p.val = bestv;
        p.idx = besti;
        partials[(size_t)f * g.ntiles + tile] = p;
    }
}

// ---- H form: R map rows x NXB column blocks, K slice [k_lo, k_hi) of the linearised (d, i) space ---------------
template <int R, int NXB>
__device__ __forceinline__ void gen_hform(const int8_t* __restrict__ Lg, const int8_t* __restrict__ Atab,
                                          const uint32_t* __restrict__ ws, const GenGeom& g, const GenTask& t, int grp,
                                          i32x4* __restrict__ part, int* __restrict__ counters,
                                          float* __restrict__ result_map, MatchPartial* __restrict__ partials)
{
    // Requests run PD steps ahead of their use; a step is R * NXB MFMAs (32 cycles each) and an L2 round trip under
    // load is ~1000 cycles, so small tiles need a longer lead.  Image rows and template fragments rotate through rings
    // of the same length NBUF = R + PD, which is the unroll period.
    constexpr int PD = R * NXB >= 16 ? 2 : (R * NXB >= 8 ? 4 : (R * NXB >= 4 ? 6 : 8));
    constexpr int NBUF = R + PD;
    constexpr int PERIOD = NBUF;
    const int lane = threadIdx.x;
    const size_t rowv = (size_t)g.nkb * 64;  // i32x4 per image row
    const i32x4* Lgrp = (const i32x4*)Lg + (size_t)grp * g.rows_pad * rowv + lane;
    const i32x4* Ap = (const i32x4*)Atab + lane;

    i32x16 acc[R][NXB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int xb = 0; xb < NXB; ++xb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][xb][e] = 0;

    i32x4 buf[NBUF][NXB];
    i32x4 a[NBUF];
    int k = t.k_lo;
    while (k < t.k_hi) {
        const int d = k / g.th, i_lo = k - d * g.th;
        const int i_hi = min(g.th, i_lo + (t.k_hi - k));
        const i32x4* Lrow = Lgrp + (size_t)(t.y0 + i_lo) * rowv + (size_t)(d + t.xb0) * 64;
        // prime: image rows y0 + i_lo .. + NBUF - 2, template rows i_lo .. + PD - 1 (row th of Atab is all zero)
#pragma unroll
        for (int r = 0; r < NBUF - 1; ++r)
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) buf[r][xb] = Lrow[(size_t)r * rowv + xb * 64];
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            const int irow = i_lo + q < i_hi ? i_lo + q : g.th;
            a[q] = Ap[((size_t)irow * g.nd + d) * 64];
        }
        const i32x4* Lnext = Lrow + (size_t)(NBUF - 1) * rowv;  // image row of the next request
        // Whole periods only, and no branch inside one: hipcc sinks loads across block boundaries to their first use
        // (request -> wait a full L2 round trip -> use), whatever sched_barrier says, but keeps them pinned inside a
        // block.  Steps past i_hi multiply the zero template row.
        for (int i0 = i_lo; i0 < i_hi; i0 += PERIOD) {
#pragma unroll
            for (int s = 0; s < PERIOD; ++s) {
                const int i = i0 + s;
                // requests for PD steps ahead, pinned in front of this step's MFMAs (Lg has zero rows past the image)
#pragma unroll
                for (int xb = 0; xb < NXB; ++xb) buf[(s + NBUF - 1) % NBUF][xb] = Lnext[xb * 64];
                Lnext += rowv;
                const int irow = i + PD < i_hi ? i + PD : g.th;  // scalar select, no branch
                a[(s + PD) % NBUF] = Ap[((size_t)irow * g.nd + d) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int xb = 0; xb < NXB; ++xb)
                        acc[r][xb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s % NBUF], buf[(s + r) % NBUF][xb], acc[r][xb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        k += i_hi - i_lo;
    }

    const int n = lane & 31, hh = lane >> 5;
    const int f = grp * 32 + n;
    if (t.nslices > 1) {
        constexpr int NQ = R * NXB * 4;
        // this group's partial tiles (below 4 GiB per group: checked by the host), 32-bit byte offsets inside
        const __amdgpu_buffer_rsrc_t rs = part_rsrc(part + (size_t)grp * t.part_stride * 64, (unsigned)t.part_stride * 1024u);
        const unsigned tile_base = (unsigned)t.part_off * 1024u + (unsigned)lane * 16u;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) part_store(acc[r][xb], rs, tile_base + (unsigned)(t.slice * NQ) * 1024u, r * NXB + xb);
        if (!part_arrive(counters + (size_t)grp * g.ntiles + t.tile, t.nslices)) return;
        part_add_all<R * NXB, (R * NXB >= 12 ? 1 : 2)>(&acc[0][0], rs, tile_base, t.slice, t.nslices);
    }
    const bool lane_ok = f < g.nframes;
    float bestv = -INFINITY;
    int besti = INT_MAX;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int xb = 0; xb < NXB; ++xb) {
            const int y = t.y0 + r, xbase = 32 * (t.xb0 + xb) + 4 * hh;
            tile_epilogue(acc[r][xb], ws, g, grp, f, lane_ok && r < t.R,
                          [&](int e, int& yy, int& xx) { yy = y; xx = xbase + (e & 3) + 8 * (e >> 2); }, result_map, bestv, besti);
            __builtin_amdgcn_sched_barrier(0);
        }
    write_partial(bestv, besti, f, g, t.tile, partials);
}

// ---- V form: one map column, 32 map rows; K slice [k_lo, k_hi) of the linearised (rho - 32 yb, kbv) space -------
__device__ __forceinline__ void gen_vform(const int8_t* __restrict__ Lg, const int8_t* __restrict__ AtabV,
                                          const uint32_t* __restrict__ ws, const GenGeom& g, const GenTask& t, int grp,
                                          i32x4* __restrict__ part, int* __restrict__ counters,
                                          float* __restrict__ result_map, MatchPartial* __restrict__ partials)
{
    const int lane = threadIdx.x;
    const size_t rowv = (size_t)g.nkb * 64;
    const int c = t.xb0;                 // remainder column index
    const int yb = t.y0;                 // first of the 32 map rows
    const i32x4* Lgrp = (const i32x4*)Lg + ((size_t)grp * g.rows_pad + yb) * rowv + (size_t)g.vkb0 * 64 + lane;
    const i32x4* Av = (const i32x4*)AtabV + ((size_t)c * g.ndelta * g.ndv) * 64 + lane;
    i32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0;
    constexpr int U = 8;  // MFMAs per group; two groups of operands in flight (two loads per MFMA: the loop lives on load latency)
    i32x4 av[2][U], bv[2][U];
#define MELF_V_REQUEST(KK, SET)                                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                        \
        const int q = min((KK) + u, t.k_hi - 1); /* clamped: the surplus of the last groups is zeroed */   \
        const int delta = q / g.ndv, kbv = q - delta * g.ndv;                                              \
        av[SET][u] = Av[(size_t)q * 64];                                                                   \
        bv[SET][u] = Lgrp[(size_t)delta * rowv + (size_t)kbv * 64];                                        \
    }
#define MELF_V_CONSUME(KK, SET)                                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                        \
        i32x4 aa = av[SET][u];                                                                             \
        if ((KK) + u >= t.k_hi) aa = i32x4{0, 0, 0, 0};                                                    \
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bv[SET][u], acc, 0, 0, 0);                         \
    }
    MELF_V_REQUEST(t.k_lo, 0)
    for (int kk = t.k_lo; kk < t.k_hi; kk += 2 * U) {  // one block per iteration: no branch inside (see gen_hform)
        MELF_V_REQUEST(kk + U, 1)
        __builtin_amdgcn_sched_barrier(0);
        MELF_V_CONSUME(kk, 0)
        __builtin_amdgcn_sched_barrier(0);
        MELF_V_REQUEST(kk + 2 * U, 0)
        __builtin_amdgcn_sched_barrier(0);
        MELF_V_CONSUME(kk + U, 1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef MELF_V_REQUEST
#undef MELF_V_CONSUME
    const int n = lane & 31, hh = lane >> 5;
    const int f = grp * 32 + n;
    if (t.nslices > 1) {
        const __amdgpu_buffer_rsrc_t rs = part_rsrc(part + (size_t)grp * t.part_stride * 64, (unsigned)t.part_stride * 1024u);
        const unsigned tile_base = (unsigned)t.part_off * 1024u + (unsigned)lane * 16u;
        part_store(acc, rs, tile_base + (unsigned)(t.slice * 4) * 1024u, 0);
        if (!part_arrive(counters + (size_t)grp * g.ntiles + t.tile, t.nslices)) return;
        part_add_all<1, 1>(&acc, rs, tile_base, t.slice, t.nslices);
    }
    float bestv = -INFINITY;
    int besti = INT_MAX;
    const int x = g.vx0 + c;
    tile_epilogue(acc, ws, g, grp, f, f < g.nframes,
                  [&](int e, int& yy, int& xx) { yy = yb + (e & 3) + 8 * (e >> 2) + 4 * hh; xx = x; }, result_map, bestv, besti);
    write_partial(bestv, besti, f, g, t.tile, partials);
}

// One kernel per tile height RC (2 / 4 / 6 / 8 rows computed): register allocation follows the tile, not the largest one.
template <int RC>
__global__ __launch_bounds__(64, 1) void k_match_gen(const int8_t* __restrict__ Lg, const int8_t* __restrict__ Atab,
                                                     const int8_t* __restrict__ AtabV, const uint32_t* __restrict__ ws,
                                                     const GenTask* __restrict__ tasks, GenGeom g, i32x4* __restrict__ part,
                                                     int* __restrict__ counters, float* __restrict__ result_map,
                                                     MatchPartial* __restrict__ partials)
{
    // XCD-aware order (as k_match_mfma): consecutive virtual ids share an XCD, so a frame group's tasks share an L2
    const int nblk = gridDim.x, id = blockIdx.x;
    const int per = nblk / 8, rem = nblk % 8, xcd = id & 7, sub = id >> 3;
    const int vid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + sub;
    const int grp = vid / g.ntasks, ti = vid - grp * g.ntasks;
    const GenTask t = tasks[ti];
    if (t.R == 0) gen_vform(Lg, AtabV, ws, g, t, grp, part, counters, result_map, partials);
    else if (t.nxb == 2) gen_hform<RC, 2>(Lg, Atab, ws, g, t, grp, part, counters, result_map, partials);
    else gen_hform<RC, 1>(Lg, Atab, ws, g, t, grp, part, counters, result_map, partials);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static const int GEN_VREM_MAX = 4;    // remainder columns that go through the V form
static const int GEN_SLICE_MIN = 24;  // template rows per slice at least (priming a slice costs ~3 steps)

bool gen_match_ok(int th, int tw, int rows, int cols)
{
    const int rh = rows - th + 1, rw = cols - tw + 1;
    // u16 row-window sums (tw * 255 < 2^16) and the epilogue's modulo-2^32 sum (th * tw * 255^2 < 2^32)
    if (!(rh >= 1 && rw >= 1 && tw <= 256 && (long)th * tw * 65025L < (1L << 32) && rows < 32768 && cols < 32768)) return false;
    // What the launches can hold (shapes beyond go to the VALU kernel, which has no such limits):
    //  * k_prep_lplane keeps the row prefix of 32 frames in LDS, 64 * (32 nkb + 8) bytes of the 128 KiB it may ask for;
    //  * GenTask::tile / y0 are 16-bit and a group's partial tiles are addressed with 32-bit byte offsets: bound the
    //    number of tiles by its worst case (2-row tiles of one column block, 16 slices of up to 32 KiB each).
    const int nd = (tw + 62) / 32, nxb_full = rw / 32, rem = rw % 32;
    const int vcols = (rem > 0 && rem <= GEN_VREM_MAX && nxb_full >= 1) ? rem : 0;
    const int nxb_h = vcols ? nxb_full : nxb_full + (rem ? 1 : 0);
    const int ndv = vcols ? (vcols - 1 + tw - 1) / 32 + 1 : 0;
    const int nkb = std::max(std::max(nxb_h + nd - 1, vcols ? nxb_full + ndv : 0), (cols + 31) / 32);
    if (64L * (32L * nkb + 8) > 128L * 1024) return false;
    const long ntiles_max = (long)((rh + 1) / 2) * std::max(nxb_h, 1) + (long)vcols * ((rh + 31) / 32);
    return ntiles_max <= 8000;   // 8000 tiles x 512 KiB of partial tiles < 4 GiB, and < 2^15 for the 16-bit fields
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

    // ---- tile shape and K slices: a small search over (rows per tile, column blocks per tile, slices per tile) ----
    // cost model, in cycles of one SIMD, fitted to launches on MI355X (tools/gpu_gen_shapes.sh): an MFMA takes ~40 cycles
    // all told, or 90 cycles per KiB it needs from L2 when every SIMD loads (NXB + 1 requests per R * NXB MFMAs); waves
    // run in rounds of 1024 (one wave per SIMD); every K pass primes its rings, every wave has an epilogue; a sliced
    // tile pays the hand-off and the completing wave's pass over the other slices' partial tiles.
    struct Tile { int y0, R, Rc, xb0, nxb; long work; bool v; };
    const int nvt = p.vcols * ((p.rh + 31) / 32);          // V-form tiles per group
    const long slots = std::max(64L, 1024L - 4L * nvt * p.groups);   // wave slots for the H form: >= 4 slices per V tile reserved
    double best_cost = 1e30;
    int best_rc = 8, best_nxb = 2, best_ns = 1;
    for (int rc = 8; rc >= 2; rc -= 2)
        for (int nx = 2; nx >= 1; --nx) {
            if (nx > nxb_h && nx > 1) continue;
            const int ntr = (p.rh + rc - 1) / rc, nstr = (nxb_h + nx - 1) / nx;
            const long ntile = (long)ntr * nstr;
            // a K step = nx + 1 fragment loads + rc * nx MFMAs: ~42 cycles per load (issue, address arithmetic, the wait it
            // eventually causes) + ~29 per MFMA (66 / 93 / 117 / 145 ns per step for 2 / 4 / 6 / 8-row tiles of one block); a sliced tile pays its hand-off: ~5 500 cycles + ~220 per KiB of partial
            // tile the completing wave has to pull (write-through stores drop the lines from L2: those reads come from
            // beyond it).  Refitted in round 3 to a sweep of 36 shapes at config 4 (tools/gen_shape_sweep.py,
            // profiles/r03/gen_shape_sweep_config4.txt): round 2's model priced the hand-off six times too cheap and picked
            // 6-row tiles in 8 slices (0.0405 ms) where 2-row tiles in 3 slices take 0.0322.
            const double step_cyc = 42.5 * (nx + 1) + 29.0 * rc * nx;
            const long ksteps = (long)p.nd * th;
            const int nq = rc * nx * 4;
            for (int ns = 1; ns <= 16; ++ns) {
                if (ns > 1 && ksteps / ns < GEN_SLICE_MIN) break;
                if (ns > 4 && nq > 32) break;  // a completing wave would pull > 256 KiB of partial tiles
                const long waves = ntile * ns * p.groups;
                const long rounds = (waves + slots - 1) / slots;
                const int dpasses = (ns >= p.nd) ? 2 : (p.nd + ns - 1) / ns + (ns > 1 ? 1 : 0);
                const double per_wave = (double)((ksteps + ns - 1) / ns + rc + 8) * step_cyc + 2500.0 * dpasses + 6000.0;
                const double reduce = ns > 1 ? 5500.0 + (double)ns * nq * 220.0 : 0.0;
                const double cost = rounds * per_wave + reduce;
                if (cost < best_cost * 0.97) { best_cost = cost; best_rc = rc; best_nxb = nx; best_ns = ns; }
            }
        }
    if (const char* e = getenv("MELF_GEN_SHAPE")) {  // experiments: "rc,nxb,ns"
        int a = 0, b = 0, c2 = 0;
        if (sscanf(e, "%d,%d,%d", &a, &b, &c2) == 3 && a >= 2 && a <= 8 && a % 2 == 0 && b >= 1 && b <= 2 && c2 >= 1 && c2 <= 16) {
            best_rc = a; best_nxb = b; best_ns = c2;
        }
    }
    std::vector<Tile> tiles;
    const int Rc = best_rc;
    p.rc = Rc;
    p.nxb_tile = best_nxb;
    p.nslices = best_ns;
    for (int y0 = 0; y0 < p.rh; y0 += Rc) {
        const int R = std::min(Rc, p.rh - y0);
        for (int xb = 0; xb < nxb_h; xb += best_nxb) {
            const int nxb = std::min(best_nxb, nxb_h - xb);
            tiles.push_back({y0, R, Rc, xb, nxb, (long)Rc * nxb * p.nd * th, false});
        }
    }
    const long h_slice_work = tiles.empty() ? 0 : tiles[0].work / best_ns;
    for (int c = 0; c < p.vcols; ++c)
        for (int yb = 0; yb < p.rh; yb += 32) {
            const int nrow = std::min(32, p.rh - yb);
            // two loads per MFMA: priced at 4 MFMA slots each when balancing
            tiles.push_back({yb, 0, 0, c, 0, 4L * (nrow + th - 1) * p.ndv, true});
        }
    p.ntiles = (int)tiles.size();
    p.part_stride = 0;
    int max_row_used = 0;
    for (int ti = 0; ti < p.ntiles; ++ti) {
        const Tile& t = tiles[ti];
        const int klen = t.v ? (std::min(32, p.rh - t.y0) + th - 1) * p.ndv : p.nd * th;
        int ns = best_ns;
        if (t.v) {
            // V-form tiles: two loads per MFMA make their K loop latency-bound (~0.1 us per MFMA), so they take every wave
            // slot the H-form tiles leave in the first round (never a second round), in slices of >= 48 MFMAs, <= 16 of them
            (void)h_slice_work;
            const long h_waves = (long)(p.ntiles - nvt) * best_ns;
            const long left = std::max(1L, (1024L / p.groups - h_waves) / std::max(1, nvt));
            ns = (int)std::max(1L, std::min<long>(std::min<long>(16, left), klen / 48));
        }
        const int nq = t.v ? 4 : t.Rc * t.nxb * 4;
        for (int sl = 0; sl < ns; ++sl) {
            GenTask k;
            k.y0 = (int16_t)t.y0; k.R = (int8_t)t.R; k.Rc = (int8_t)t.Rc; k.nxb = (int8_t)t.nxb; k.pad0 = 0; k.xb0 = (int16_t)t.xb0;
            k.tile = (int16_t)ti;
            k.k_lo = (int)((long)sl * klen / ns);
            k.k_hi = (int)((long)(sl + 1) * klen / ns);
            k.slice = (int16_t)sl; k.nslices = (int16_t)ns;
            k.part_off = ns > 1 ? p.part_stride : 0;
            k.part_stride = 0;
            p.tasks.push_back(k);
        }
        if (ns > 1) p.part_stride += ns * nq;
        if (!t.v) max_row_used = std::max(max_row_used, t.y0 + t.Rc);
    }
    for (auto& k : p.tasks) k.part_stride = p.part_stride;
    // heavy tasks first: the tail of the launch is made of the light ones
    std::stable_sort(p.tasks.begin(), p.tasks.end(), [&](const GenTask& a, const GenTask& b) {
        const long wa = (a.R ? (long)a.Rc * a.nxb : 4L) * (a.k_hi - a.k_lo), wb = (b.R ? (long)b.Rc * b.nxb : 4L) * (b.k_hi - b.k_lo);
        return wa > wb;
    });
    p.ntasks = (int)p.tasks.size();
    // last image row a wave asks for: the K loop runs whole periods of NBUF = Rc + 2 steps and requests NBUF - 1 rows ahead
    p.rows_pad = std::max(rows, max_row_used + th + 2 * (Rc + 8)) + 1;
    if (p.vcols) p.rows_pad = std::max(p.rows_pad, ((p.rh + 31) / 32) * 32 + th);
    p.lg_bytes = (size_t)p.groups * p.rows_pad * p.nkb * 1024;
    p.r_bytes = (size_t)p.groups * rows * p.rwp * 32 * sizeof(uint16_t);
    p.ws_bytes = (size_t)p.groups * p.rh * p.rwp * 32 * sizeof(uint32_t);
    p.part_bytes = (size_t)p.groups * p.part_stride * 64 * sizeof(i32x4);
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

void launch_gen_match(int n, const GenPlan& p, int th, int tw, long tsum, double tmean, const GenDev& dev, const int8_t* d_lg,
                      const uint32_t* d_ws, float* d_result_map, MatchPartial* d_partials, hipStream_t stream, hipEvent_t ev_start,
                      hipEvent_t ev_stop)
{
    GenGeom g;
    g.rh = p.rh; g.rw = p.rw; g.rwp = p.rwp; g.rows_pad = p.rows_pad; g.nkb = p.nkb; g.th = th; g.nd = p.nd;
    g.ndv = p.ndv; g.ndelta = p.ndelta; g.vx0 = p.vx0; g.vkb0 = p.vkb0;
    g.nframes = n; g.ntasks = p.ntasks; g.ntiles = p.ntiles;
    g.k1 = (int)(128 * (tsum - 128L * th * tw));
    g.tmean = tmean;
    dim3 grid(p.ntasks * p.groups), block(64);
#define MELF_GEN_LAUNCH(RC) \
    hipExtLaunchKernelGGL((k_match_gen<RC>), grid, block, 0, stream, ev_start, ev_stop, 0, d_lg, dev.atab, dev.atabv, d_ws, dev.tasks, g, \
                          (i32x4*)dev.part, dev.counters, d_result_map, d_partials)
    switch (p.rc) {
        case 2: MELF_GEN_LAUNCH(2); break;
        case 4: MELF_GEN_LAUNCH(4); break;
        case 6: MELF_GEN_LAUNCH(6); break;
        default: MELF_GEN_LAUNCH(8); break;
    }
#undef MELF_GEN_LAUNCH
}

}  // namespace melf
