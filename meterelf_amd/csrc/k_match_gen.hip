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
    int rwp;           // row length of R / ws in map columns (multiple of 32)
    int rows_pad;      // image rows per group in Lg (zero rows beyond the image)
    int nkb;           // 32-column blocks per image row in Lg
    int th, nd;        // template rows, Toeplitz blocks per template row
    int ndv, ndelta;   // V form: blocks per image row, rows of AtabV per column
    int vx0, vkb0;     // V form: first remainder column, its image block
    int nframes, ntiles;
    int atab_bytes, atabv_bytes;
    int k1;            // 128 * (sum T - 128 th tw)
    double tmean;
};

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

// A workgroup's LDS: the tile's accumulators [block][16][64 lanes] i32 (block = row * nxb + column block; V form: one
// block), then one (max, arg-max) per wave and frame.
__device__ __forceinline__ MatchPartial* lds_best(int* s, int nblocks_max) { return (MatchPartial*)(s + nblocks_max * 1024); }

// this wave's accumulator block added into the workgroup's (LDS atomic adds without return: ds_add_u32)
__device__ __forceinline__ void lds_accumulate(int* s, int block, const i32x16& acc, int lane)
{
    int* p = s + block * 1024 + lane;
#pragma unroll
    for (int e = 0; e < 16; ++e) (void)__hip_atomic_fetch_add(p + e * 64, acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

// ---- H form: R map rows x NXB column blocks; this wave's slice [k_lo, k_hi) of the linearised (d, i) space ---------
template <int R, int NXB, int NBMAX>
__device__ __forceinline__ void gen_hform(const int8_t* __restrict__ Lg, const int8_t* __restrict__ Atab,
                                          const uint32_t* __restrict__ ws, const GenGeom& g, const GenTile& t, int tile, int grp, int w, int ns,
                                          int* s_dyn, float* __restrict__ result_map, MatchPartial* __restrict__ partials)
{
    // Requests run PD steps ahead of their use; a step is R * NXB MFMAs (32 cycles each) and an L2 round trip under
    // load is ~1000 cycles, so small tiles need a longer lead.  Image rows and template fragments rotate through rings
    // of the same length NBUF = R + PD, which is the unroll period.
    constexpr int PD = R * NXB >= 16 ? 2 : (R * NXB >= 8 ? 4 : (R * NXB >= 4 ? 6 : 8));
    constexpr int NBUF = R + PD;
    constexpr int PERIOD = NBUF;
    const int lane = threadIdx.x & 63;
    const unsigned lane16 = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t rsL = frag_rsrc(Lg + (size_t)grp * g.rows_pad * g.nkb * 1024, (unsigned)g.rows_pad * (unsigned)g.nkb * 1024u);
    const __amdgpu_buffer_rsrc_t rsA = frag_rsrc(Atab, (unsigned)g.atab_bytes);
    const unsigned rowb = (unsigned)g.nkb * 1024u;  // bytes per image row of the group

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
    const bool lane_ok = f < g.nframes;
    float bestv = -INFINITY;
    int besti = INT_MAX;
    if (ns == 1) {   // the wave owns the whole tile: epilogue straight from its registers, no LDS
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) {
                const int y = t.y0 + r, xbase = 32 * (t.xb0 + xb) + 4 * hh;
                tile_epilogue(acc[r][xb], ws, g, grp, f, lane_ok && r < t.R,
                              [&](int e, int& yy, int& xx) { yy = y; xx = xbase + (e & 3) + 8 * (e >> 2); }, result_map, bestv, besti);
                __builtin_amdgcn_sched_barrier(0);
            }
    } else {
        // the slices add up in LDS (zeroed by the workgroup before the K loops); then the tile's row blocks are dealt out
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) lds_accumulate(s_dyn, r * NXB + xb, acc[r][xb], lane);
        __syncthreads();
        const int nblk = t.R * NXB;   // rows past the map are not looked at
        for (int b = w; b < nblk; b += ns) {
            const int r = b / NXB, xb = b - r * NXB;
            const i32x16 sum = lds_block(s_dyn, b, lane);
            const int y = t.y0 + r, xbase = 32 * (t.xb0 + xb) + 4 * hh;
            tile_epilogue(sum, ws, g, grp, f, lane_ok, [&](int e, int& yy, int& xx) { yy = y; xx = xbase + (e & 3) + 8 * (e >> 2); },
                          result_map, bestv, besti);
        }
    }
    fold_and_write(s_dyn, NBMAX, bestv, besti, grp, g, tile, w, ns, partials);
}

// ---- V form: one map column, 32 map rows; this wave's slice [k_lo, k_hi) of the linearised (rho - 32 yb, kbv) space -------
template <int NBMAX>
__device__ __forceinline__ void gen_vform(const int8_t* __restrict__ Lg, const int8_t* __restrict__ AtabV,
                                          const uint32_t* __restrict__ ws, const GenGeom& g, const GenTile& t, int tile, int grp, int w, int ns,
                                          int* s_dyn, float* __restrict__ result_map, MatchPartial* __restrict__ partials)
{
    const int lane = threadIdx.x & 63;
    const unsigned lane16 = (unsigned)lane * 16u;
    const int c = t.xb0;                 // remainder column index
    const int yb = t.y0;                 // first of the 32 map rows
    const __amdgpu_buffer_rsrc_t rsL = frag_rsrc(Lg + (size_t)grp * g.rows_pad * g.nkb * 1024, (unsigned)g.rows_pad * (unsigned)g.nkb * 1024u);
    const __amdgpu_buffer_rsrc_t rsV = frag_rsrc(AtabV, (unsigned)g.atabv_bytes);
    const unsigned rowb = (unsigned)g.nkb * 1024u;
    const unsigned vbase = (unsigned)(c * g.ndelta * g.ndv) * 1024u;
    const unsigned lbase = (unsigned)yb * rowb + (unsigned)g.vkb0 * 1024u;
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
    const int n = lane & 31, hh = lane >> 5;
    const int f = grp * 32 + n;
    float bestv = -INFINITY;
    int besti = INT_MAX;
    const int x = g.vx0 + c;
    auto elem = [&](int e, int& yy, int& xx) { yy = yb + (e & 3) + 8 * (e >> 2) + 4 * hh; xx = x; };
    if (ns == 1) {
        tile_epilogue(acc, ws, g, grp, f, f < g.nframes, elem, result_map, bestv, besti);
    } else {
        lds_accumulate(s_dyn, 0, acc, lane);
        __syncthreads();
        if (w == 0) {
            const i32x16 sum = lds_block(s_dyn, 0, lane);
            tile_epilogue(sum, ws, g, grp, f, f < g.nframes, elem, result_map, bestv, besti);
        }
    }
    fold_and_write(s_dyn, NBMAX, bestv, besti, grp, g, tile, w, ns, partials);
}

// One kernel per tile shape class: RC rows computed (2 / 4 / 6 / 8), at most NXBMAX column blocks per tile, at most NSMAX
// waves (K slices) per workgroup -- the launch bound is what lets the small shapes run two waves per SIMD (<= 256 registers)
// while the large ones keep all 512.
template <int RC, int NXBMAX, int NSMAX>
__global__ __launch_bounds__(64 * NSMAX) void k_match_gen(const int8_t* __restrict__ Lg, const int8_t* __restrict__ Atab,
                                                          const int8_t* __restrict__ AtabV, const uint32_t* __restrict__ ws,
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
    const GenTile t = tiles[ti];
    const int ns = (int)(blockDim.x >> 6);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (ns > 1) {   // the workgroup's accumulator tile starts at zero (the slices ADD into it)
        const int nb = t.R ? (int)t.Rc * (int)t.nxb : 1;
        i32x4* z = (i32x4*)s_dyn;
        for (int i = threadIdx.x; i < nb * 256; i += blockDim.x) z[i] = i32x4{0, 0, 0, 0};
        __syncthreads();
    }
    if (t.R == 0) gen_vform<NBMAX>(Lg, AtabV, ws, g, t, ti, grp, w, ns, s_dyn, result_map, partials);
    else if (NXBMAX == 2 && t.nxb == 2) gen_hform<RC, NXBMAX, NBMAX>(Lg, Atab, ws, g, t, ti, grp, w, ns, s_dyn, result_map, partials);
    else gen_hform<RC, 1, NBMAX>(Lg, Atab, ws, g, t, ti, grp, w, ns, s_dyn, result_map, partials);
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
    // Cost model in shader cycles, fitted to launches on MI355X (tools/gen_shape_sweep.py, profiles/r04/gen_shape_sweep_*.txt).
    // A workgroup = one tile of one frame group, its waves = the K slices.  A CU holds 8 waves of the small tile shapes
    // (<= 256 registers) and 4 of the others, and whole workgroups only; the launch runs in rounds of what the 256 CUs hold.
    // A wave's time: K steps of (nx + 1) fragment loads + rc * nx MFMAs, a priming phase per Toeplitz block it touches, the
    // LDS reduction and its share of the tile's epilogue; a CU that holds several waves per SIMD is bound by the sum of their
    // MFMAs (32 cycles each) if that is longer.
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
            const double step_cyc = std::max(32.0 * rc * nx + 12.0, 30.0 * (nx + 1) + 29.0 * rc * nx);
            const int wpc = gen_nsmax(rc, nx) == 8 ? 8 : 4;      // waves per CU of this register class
            for (int ns = 1; ns <= gen_nsmax(rc, nx); ++ns) {
                if (ns > 1 && ksteps / ns < GEN_SLICE_MIN) break;
                const long wgs = ntile * p.groups;
                const int lds = ns > 1 ? rc * nx * 4096 + ns * 256 : 0;
                const int wg_per_cu = std::max(1, std::min(wpc / ns, lds ? (160 * 1024) / lds : 16));
                const long rounds = (wgs + 256L * wg_per_cu - 1) / (256L * wg_per_cu);
                const long on_cu = std::min<long>(wg_per_cu, (wgs + 255) / 256);     // workgroups sharing a CU in a round
                const int dpasses = (ns >= p.nd) ? 2 : (p.nd + ns - 1) / ns + (ns > 1 ? 1 : 0);
                double per_wave = (double)((ksteps + ns - 1) / ns + (rc + 8) * dpasses) * step_cyc + 1500.0 * dpasses + 5000.0;
                per_wave += ns > 1 ? rc * nx * 16 * 10.0 + 1500.0 + ((rc * nx + ns - 1) / ns) * 1800.0 : rc * nx * 1800.0;
                if (nvt) per_wave = std::max(per_wave, (double)((vsteps + ns - 1) / ns) * 75.0 + 7000.0);
                // matrix-pipe bound of a CU's share (4 SIMDs), and the price of sharing a SIMD between waves
                const double mfma_cu = (double)on_cu * ksteps * rc * nx * 32.0 / 4.0;
                const double share = on_cu * ns > 4 ? 1.0 + 0.15 * ((double)on_cu * ns / 4.0 - 1.0) : 1.0;
                const double cost = rounds * std::max(per_wave * share, mfma_cu);
                if (cost < best_cost * 0.97) { best_cost = cost; best_rc = rc; best_nxb = nx; best_ns = ns; }
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
    p.lds_bytes = p.nslices > 1 ? (size_t)Rc * best_nxb * 4096 + (size_t)gen_nsmax(Rc, best_nxb) * 32 * sizeof(MatchPartial) : 0;
    // last image row a wave asks for: the K loop runs whole periods of NBUF = Rc + PD steps and requests NBUF - 1 rows ahead
    // (a request beyond the group's rows returns zero: the buffer resource ends there)
    p.rows_pad = std::max(rows, max_row_used + th + 2 * (Rc + 8)) + 1;
    if (p.vcols) p.rows_pad = std::max(p.rows_pad, ((p.rh + 31) / 32) * 32 + th);
    p.lg_bytes = (size_t)p.groups * p.rows_pad * p.nkb * 1024;
    p.r_bytes = (size_t)p.groups * rows * p.rwp * 32 * sizeof(uint16_t);
    p.ws_bytes = (size_t)p.groups * p.rh * p.rwp * 32 * sizeof(uint32_t);
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
                       const GenDev& dev, const uint32_t* d_ws, const GenGeom& g, float* d_result_map, MatchPartial* d_partials)
{
    if (lds > 48 * 1024) {   // once per device and instantiation: dynamic LDS beyond the default limit
        static bool attr_set[64] = {false};
        int devid = 0;
        (void)hipGetDevice(&devid);
        if (devid >= 0 && devid < 64 && !attr_set[devid]) {
            (void)hipFuncSetAttribute((const void*)k_match_gen<RC, NXBMAX, NSMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr_set[devid] = true;
        }
    }
    hipExtLaunchKernelGGL((k_match_gen<RC, NXBMAX, NSMAX>), grid, block, (unsigned)lds, stream, ev_start, ev_stop, 0, d_lg, dev.atab, dev.atabv, d_ws,
                          dev.tiles, g, d_result_map, d_partials);
}

void launch_gen_match(int n, const GenPlan& p, int th, int tw, long tsum, double tmean, const GenDev& dev, const int8_t* d_lg,
                      const uint32_t* d_ws, float* d_result_map, MatchPartial* d_partials, hipStream_t stream, hipEvent_t ev_start,
                      hipEvent_t ev_stop)
{
    GenGeom g;
    g.rh = p.rh; g.rw = p.rw; g.rwp = p.rwp; g.rows_pad = p.rows_pad; g.nkb = p.nkb; g.th = th; g.nd = p.nd;
    g.ndv = p.ndv; g.ndelta = p.ndelta; g.vx0 = p.vx0; g.vkb0 = p.vkb0;
    g.nframes = n; g.ntiles = p.ntiles;
    g.atab_bytes = (int)p.atab_bytes; g.atabv_bytes = (int)p.atabv_bytes;
    g.k1 = (int)(128 * (tsum - 128L * th * tw));
    g.tmean = tmean;
    dim3 grid(p.ntiles * p.groups), block(64 * p.nslices);
#define MELF_GEN_CASE(RC, NX, NS) \
    case (RC) * 4 + (NX): launch_gen<RC, NX, NS>(grid, block, p.lds_bytes, stream, ev_start, ev_stop, d_lg, dev, d_ws, g, d_result_map, d_partials); break;
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
