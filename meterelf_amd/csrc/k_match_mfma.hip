// K2 on the matrix cores -- cv2.matchTemplate(L, template, TM_CCOEFF) + cv2.minMaxLoc
// (reference: meterelf/_utils.py:91-97, called from meterelf/_image.py:57-66) for whole
// batches: the exact integer cross-correlation as an i8 MFMA GEMM with FRAMES on the N
// dimension.
//
//   D[m][n] += A[m][k] * B[k][n]      v_mfma_i32_32x32x32_i8
//   m = output column x inside a 32-wide block xb, n = frame inside a group of 32 frames,
//   k = image column inside a 32-wide block kb of image row y' = y + i
//   A[m][k] = T'[i][32 d + k - m]   (Toeplitz expansion of template row i, d = kb - xb, zero outside)
//   B[k][n] = L'_n[y'][32 kb + k]
// with T' = T - 128 and L' = L - 128 as signed bytes.  The exact u8 correlation follows from
//   sum T L = sum T' L' + 128 * winsum(L) + 128 * (sum T - 128 N)
// and the window sums are exact too: row-window sums from the prep pass, added up over the template rows by the match waves.
// 188 useful of every 224 K columns are non-zero (84 % dense); accumulators stay in registers
// over the whole 119 x 224 K loop, so there is no scatter and no partial-sum traffic.
//
// One wave owns R consecutive output rows x 64 output columns x 32 frames (16 R x 2 accumulator registers).  At template
// row i it needs image rows y0+i .. y0+i+R-1: a sliding window of R + 1 register row-buffers (R live + 1 incoming, rotated
// by an (R + 1)-way unroll), so every image row is fetched once per wave, straight from L2 in B-fragment order (lane * 16
// bytes).  R and the number of waves per frame group follow the batch size (mfma_plan: RB = 2..5 full rows per wave, plus
// pairs of (RB + 1)-row waves that share a map row), so that one round of ~1024 waves fills the chip from 481 frames up.
// The window sums of TM_CCOEFF are added up by the same waves from the row-window sums k_prep_lplane leaves (round 3).
//
// The operand / result lane maps were verified with exact integer data
// (tools/ubench/mfma_i8_layout.hip): A lane l = A[l & 31][16 (l >> 5) + j],
// B lane l = B[16 (l >> 5) + j][l & 31], D lane l reg r = D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "melf_device.h"
#include <hip/hip_ext.h>

#include "melf_internal.h"

namespace melf {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// k_prep_lplane: searched image ->
//   Lg[group][row][kb][h][n][16] : L' = L - 128 (int8) in MFMA B-fragment order (a row is
//                                  NKB KiB: 32 frames x 32*NKB image columns), zero beyond the
//                                  image / beyond the batch;
//   R[group][row][x][n]          : row-window sums sum_{j < tw} L[row][x + j] (u16), x < 64.
// One workgroup (256 threads) per (group, row); thread = (frame n = t / 8, block kb = t % 8)
// converts 32 pixels, keeps a 32-element running sum, and an 8-lane scan turns the per-block
// totals into the row's inclusive prefix sums (int16, in LDS); the window sums are then two
// LDS reads each.  The 16-byte fragment pieces go through LDS so that the row leaves with
// coalesced 16-byte stores.
// ---------------------------------------------------------------------------
typedef unsigned int u32x4m __attribute__((ext_vector_type(4)));

// R is written in the match waves' EPILOGUE order: [row][piece (rwp / 16)][lane (64)][q (8)] u16, piece = 2 xb + half: lane (n, hh)
// of a match wave finds the row-window sums of its accumulator elements e = 8 half + q of column block xb, i.e. of map columns
// x = 32 xb + (e & 3) + 8 (e >> 2) + 4 hh, in one lane-contiguous 16-byte piece.  The match waves add them up over the template
// rows themselves (k_match_mfma since round 3, k_match_gen since round 4): no column-sum kernel, no window-sum array.
// pairs (k_match_mfma, round 6; 0 for the general kernel): the first `pairs` image blocks share their registers with blocks
// 6 .. 6 + pairs - 1 as the B operand of a 2:4-sparse matrix instruction: P_p = blocks (p, p + 6), dword t of lane (n, h) =
// {L'[32 p + 16 h + 2 t], L'[.. + 1], L'[32 (p + 6) + 16 h + 2 t], L'[.. + 1]}, dwords 0-3 in 1 KiB slot 2 p, dwords 4-7 in slot
// 2 p + 1; the blocks in between follow in plain fragment order (slots 2 pairs ..).  The interleave happens on the row's way out of LDS
// (two 8-byte reads and four v_perm_b32 per 16-byte piece of a paired slot); the row still leaves in coalesced 16-byte stores.
template <bool FROM_BGR>
__global__ __launch_bounds__(256) void k_prep_lplane(MatchSrc src, int nframes, int nkb, int rows_pad, int tw, int rwp, int pairs,
                                                     int8_t* __restrict__ Lg, uint16_t* __restrict__ R)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[8 * 2 * 32 * 4];  // [kb][h][n][16 B], one chunk of 8 blocks
    extern __shared__ int16_t pre_dyn[];                                    // [32][nkb * 32 + 8]: inclusive prefix of L' per frame (mod 2^16)
    const int pstride = nkb * 32 + 8;
    const int y = blockIdx.x, grp = blockIdx.y;
    const int t = threadIdx.x, n = t >> 3, kl = t & 7;
    const int f = grp * 32 + n;
    int carry = 0;  // prefix of the blocks before this chunk (per frame, same in its 8 lanes)
    // (wave-uniform, once) every 100-byte gather window of this row, in every frame of the group, ends inside the caller's buffer -- all
    // rows but the last few of the last frame; the per-lane pointer test below then never runs (as the branch condition of every
    // thread it cost 10 % of the kernel: profiles/r06/prep_bisect.txt)
    const bool rows_safe = (size_t)min(grp * 32 + 31, nframes - 1) * src.frame_stride + (size_t)(src.y0 + y) * src.row_stride +
                           (size_t)(src.x0 + 32 * (nkb - 1)) * 3 + 100 <= src.readable;
    u32x4m* out = (u32x4m*)(Lg + ((size_t)grp * rows_pad + y) * (size_t)nkb * 1024);
    for (int kc = 0; kc < nkb; kc += 8) {
        const int kb = kc + kl;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // 32 output bytes (L' = 0 <=> pad)
        const bool live = kb < nkb && f < nframes && y < src.rows && kb * 32 < src.cols;  // uniform per wave except ragged tails
        if (live) {
            const uint8_t* prow = src.base + (size_t)f * src.frame_stride + (size_t)(src.y0 + y) * src.row_stride;
            const int xbeg = kb * 32;
            const int npx = min(32, src.cols - xbeg);  // < 32 only in the last block: masked below, not branched on
            if (FROM_BGR) {
                const size_t o = (size_t)(src.x0 + xbeg) * 3;
                const uint8_t* p = prow + o;
                const int mis = (int)((size_t)p & 3);
                // 25 aligned dwords cover the 96 bytes of 32 pixels at any byte alignment; the window may
                // reach past the crop (never used: masked) but must stay inside the caller's buffer
                if (rows_safe || (size_t)f * src.frame_stride + (size_t)(src.y0 + y) * src.row_stride + o + 100 <= src.readable) {
                    const uint32_t* q = (const uint32_t*)(p - mis);
                    uint32_t d[25];
#pragma unroll
                    for (int i = 0; i < 25; ++i) d[i] = q[i];
                    uint32_t a[24];
#pragma unroll
                    for (int i = 0; i < 24; ++i) a[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], (uint32_t)mis * 8u);
#pragma unroll
                    for (int k = 0; k < 32; ++k) {
                        const int j = (3 * k) >> 2, sh = ((3 * k) & 3) * 8;
                        const uint32_t px = sh <= 8 ? (a[j] >> sh) : __builtin_amdgcn_alignbit(a[j + 1 < 24 ? j + 1 : 23], a[j], sh);
                        const int L = hls_lightness_fast(px & 255, (px >> 8) & 255, (px >> 16) & 255);
                        w[k >> 2] |= (uint32_t)((L - 128) & 255) << ((k & 3) * 8);
                    }
                    // (The last block's columns beyond the image keep whatever the gather found there -- pixels of the same frame.  No map
                    // position inside the map reaches them: x + j <= cols - 1 for x < rw, and the window sums stop at column cols - 1 too;
                    // the positions that do are thrown away by the match kernels.  Masking them cost 40 issue slots per wave and row.)
                } else {  // last bytes of the frame buffer: byte loads
                    for (int k = 0; k < npx; ++k) {
                        const int L = hls_lightness(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
                        w[k >> 2] |= (uint32_t)((L - 128) & 255) << ((k & 3) * 8);
                    }
                }
            } else {
                const uint8_t* p = prow + src.x0 + xbeg;
                for (int k = 0; k < npx; ++k) w[k >> 2] |= (uint32_t)(((int)p[k] - 128) & 255) << ((k & 3) * 8);
            }
        }
        if (kc) __syncthreads();  // the previous chunk's tile has been written out
        // fragment-order image of the row (plain: the paired operands are put together when the row leaves, below)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4m v = {w[4 * h], w[4 * h + 1], w[4 * h + 2], w[4 * h + 3]};
            *(u32x4m*)(tile + ((kl * 2 + h) * 32 + n) * 4) = v;
        }
        // inclusive prefix sums of L' along the row: the block's total (four signed bytes per v_dot4), a scan of the totals over the
        // frame's 8 lanes, then the 32 running sums are produced and stored pair by pair (nothing but the running sum stays live)
        int run = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) run = __builtin_amdgcn_sdot4((int)w[j], 0x01010101, run, false);
        int off = run;  // inclusive scan of block totals over lanes with equal t >> 3
#pragma unroll
        for (int dlt = 1; dlt < 8; dlt <<= 1) {
            const int o2 = __shfl_up(off, dlt, 8);
            if (kl >= dlt) off += o2;
        }
        const int chunk_total = __shfl(off, 7, 8);
        off += carry - run;  // exclusive, including the earlier chunks
        carry += chunk_total;
        if (kb < nkb) {
            int acc = off;
#pragma unroll
            for (int k = 0; k < 32; k += 2) {
                acc += (int)(int8_t)((w[k >> 2] >> ((k & 3) * 8)) & 255u);
                const uint32_t lo16 = (uint32_t)acc & 0xffffu;
                acc += (int)(int8_t)((w[k >> 2] >> (((k + 1) & 3) * 8)) & 255u);
                *(uint32_t*)&pre_dyn[n * pstride + kb * 32 + k] = lo16 | ((uint32_t)acc << 16);
            }
        }
        __syncthreads();
        const int nb = min(8, nkb - kc);
        if (pairs == 0) {
            for (int i = t; i < nb * 64; i += 256) out[kc * 64 + i] = *(const u32x4m*)(tile + i * 4);
        } else {
            // (nkb <= 8: one chunk)  output piece i = 1 KiB slot i >> 6, lane i & 63.  Slots 2 p, 2 p + 1: dwords 0-3 / 4-7 of P_p -- two
            // dwords of block p's fragment and two of block p + 6's, interleaved by 16-bit pairs; slots 2 pairs ..: blocks pairs .. 5
            for (int i = t; i < nb * 64; i += 256) {
                const int slot = i >> 6, li = i & 63;
                u32x4m v;
                if (slot < 2 * pairs) {   // (uniform per wave)
                    const int p = slot >> 1, hl = slot & 1;
                    const uint2 lo2 = *(const uint2*)(tile + (p * 64 + li) * 4 + 2 * hl), hi2 = *(const uint2*)(tile + ((p + 6) * 64 + li) * 4 + 2 * hl);
                    v.x = __builtin_amdgcn_perm(hi2.x, lo2.x, 0x05040100u); v.y = __builtin_amdgcn_perm(hi2.x, lo2.x, 0x07060302u);
                    v.z = __builtin_amdgcn_perm(hi2.y, lo2.y, 0x05040100u); v.w = __builtin_amdgcn_perm(hi2.y, lo2.y, 0x07060302u);
                } else {
                    v = *(const u32x4m*)(tile + ((slot - pairs) * 64 + li) * 4);
                }
                out[i] = v;
            }
        }
    }
    // window sums: R[x] = P[x + tw - 1] - P[x - 1] + 128 tw   (P = inclusive prefix of L - 128, modulo 2^16:
    // the window sum itself is below 2^16 for tw <= 257)
    if (y < src.rows) {
        uint16_t* ro = R + (((size_t)grp * src.rows + y) * rwp) * 32;
        const int bias = tw * 128;
        const int pmax = nkb * 32 - 1;
        const int ln = t & 63, nn = ln & 31, hh = ln >> 5;
        for (int kk = t >> 6; kk < rwp / 16; kk += 4) {   // 1 KiB pieces, four per pass of the workgroup
            const int xb = kk >> 1, e0 = 8 * (kk & 1);
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + q;
                const int x = 32 * xb + (e & 3) + 8 * (e >> 2) + 4 * hh;
                const int hi = (int)pre_dyn[nn * pstride + min(x + tw - 1, pmax)], lo = x > 0 ? (int)pre_dyn[nn * pstride + min(x - 1, pmax)] : 0;
                const uint32_t v = (uint32_t)(hi - lo + bias) & 0xffffu;
                if (q & 1) o[q >> 1] |= v << 16; else o[q >> 1] = v;
            }
            u32x4m ov = {o[0], o[1], o[2], o[3]};
            *(u32x4m*)(ro + ((size_t)kk * 64 + ln) * 8) = ov;
        }
    }
}

// ---------------------------------------------------------------------------
// k_match_mfma
// ---------------------------------------------------------------------------
struct MfmaGeom {
    int rh, rw;          // correlation map size
    int rows_pad;        // rows per group in Lg
    int th_pad;          // template rows padded to a multiple of 6 (zero rows)
    int nframes;
    int nparts;          // row blocks per frame = partials per frame = na + 2 * (pairs)
    int na;              // blocks 0..na-1 own RB full rows each; then pairs of (RB + 1)-row blocks that share their
                         // middle row (one 32-column block of it each): 2 RB + 1 map rows per pair
    int k1;              // 128 * (sum T - 128 * th * tw)
    int rows, th;        // searched image rows, template rows (fused window sums)
    int ntiles;          // row blocks per frame group (nparts = ntiles x K slices)
    double tmean;
};

#ifdef MELF_MATCH_STAMP
// Diagnostic build only (make stamp; never the shipped library): per-wave shader-clock and 100 MHz
// real-time stamps around the whole wave, read back with melf_debug_match_stamps -- the in-kernel clock
// check of MI355X_MICROARCH.md ("DVFS give-back", item 6).
__device__ uint64_t g_match_stamps[4 * 8192];
__device__ uint64_t g_match_loop_end[8192];  // shader clock when the MFMA loops are done (epilogue starts)
extern "C" __attribute__((visibility("default"))) int melf_debug_match_stamps(uint64_t* out, int nwaves)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_match_stamps), sizeof(uint64_t) * 4 * (size_t)(nwaves < 8192 ? nwaves : 8192)) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int melf_debug_match_loop_end(uint64_t* out, int nwaves)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_match_loop_end), sizeof(uint64_t) * (size_t)(nwaves < 8192 ? nwaves : 8192)) == hipSuccess ? 0 : -1;
}
#endif

__device__ inline bool better_m(float v, int i, float bv, int bi)
{
    return i != INT_MAX && (bi == INT_MAX || v > bv || (v == bv && i < bi));
}

// One wave's whole job.  MFIRST / MLAST: which 32-column blocks (bit xb) the first / last of the R
// rows computes -- the balanced layout gives two neighbouring waves one column block each of a
// shared row, so that every wave carries 8 or 9 half-row units instead of 10.
// Window sums: `ws` points at the row-window sums R in epilogue order (k_prep_lplane) and the wave adds them
// up itself -- one 4 KiB row per template row, requested a step ahead like every other operand, the 64
// additions spread over the step's MFMA sub-blocks (the vector ALU is idle there) -- into the window sums of its first
// map row; the following rows slide (minus the row that leaves, plus the row that enters) in the epilogue.
// K slices (round 5, KS = 2 or 4): the workgroup's KS waves each run the whole tile over 1 / KS of the template rows; the slices'
// accumulators and window sums add up in the workgroup's LDS (ds_add_u32, lane-contiguous: conflict-free) and the tile's map
// rows are dealt out to the waves for the epilogue.  A 512-frame launch then runs 1024 waves of 8 or 9 half-row units x 60
// template rows (the 1024-frame layout's operand reuse) instead of 4 or 5 units x 120 rows, a 256-frame launch 8 or 9 x 30.
template <int R, int NXB, int KS>
struct SliceLds {
    uint32_t acc[R * NXB][16][64];   // tile (r, xb), accumulator register e, lane
    uint32_t wsa[32][64];            // window sums of the tile's first map row
};

constexpr int MM_NF_DEV = 8;   // template fragments per row in Atab (= MM_NF of the host side)
template <int ND, int NXB, int R, int PD /* prefetch distance in template rows */, int MFIRST, int MLAST, int KS = 1>
__device__ __forceinline__ void match_wave(const int8_t* __restrict__ Lg, const int8_t* __restrict__ Atab,
                                           const uint32_t* __restrict__ ws, const MfmaGeom& g,
                                           float* __restrict__ result_map, MatchPartial* __restrict__ partials,
                                           int grp, int rblk, int y0, void* lds_raw = nullptr)
{
    auto on = [](int r, int xb) -> bool {
        return r == 0 ? ((MFIRST >> xb) & 1) : (r == R - 1 ? ((MLAST >> xb) & 1) : true);
    };
    constexpr int NKB = ND + NXB - 1;
    constexpr int NBUF = R + PD;   // image-row register buffers: R live + PD in flight
    static_assert(PD == 1, "the template fragments are single-buffered: the next row's fragment d is requested right after this row's last use of fragment d");
    const int lane = threadIdx.x & 63;
    const int ks = KS > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;   // this wave's K slice
    const int SL = g.th_pad / KS, i_lo = ks * SL, i_hi = i_lo + SL;                           // its template rows
    // the map row r of the tile whose epilogue this wave runs (KS > 1: rows dealt out evenly)
    auto mine = [&](int r) -> bool { return KS == 1 || (r * KS) / R == ks; };
    SliceLds<R, NXB, KS>* const X = (SliceLds<R, NXB, KS>*)lds_raw;
    if (KS > 1) {   // zero the workgroup's tile; the barrier is long past when the first wave adds to it
        uint32_t* z = (uint32_t*)X;
        for (int i = threadIdx.x; i < (int)(sizeof(SliceLds<R, NXB, KS>) / 4); i += 64 * KS) z[i] = 0;
        __syncthreads();
    }

    // Every operand is fetched with a buffer load: resource in scalar registers, ONE vector register (16 lane) as the offset of
    // every load of the kernel, the row / slot part as a 32-bit scalar offset -- no address arithmetic on the vector ALU and no
    // 64-bit address pairs in the register file the accumulators and row buffers fill.
    const uint32_t l16 = (uint32_t)lane * 16u;
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(Lg + (size_t)grp * g.rows_pad * (size_t)NKB * 1024), 0,
                                                                         (int)((unsigned)g.rows_pad * (unsigned)NKB * 1024u), 0x27000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(Atab + 1024), 0, (int)((unsigned)(g.th_pad + PD) * MM_NF_DEV * 1024u), 0x27000);
    // row-window sums of the group; an offset at or beyond the end reads zeros (the hardware's range check): that is the "row" of a padding template row
    const unsigned rsR_bytes = (unsigned)g.rows * 4096u;
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(ws + (size_t)grp * g.rows * 1024), 0, (int)rsR_bytes, 0x27000);
    auto LD = [&](__amdgpu_buffer_rsrc_t rs, unsigned rowoff, int slot) -> i32x4 {
        return __builtin_amdgcn_raw_buffer_load_b128(rs, l16, rowoff + (unsigned)slot * 1024u, 0);
    };
    const unsigned Lrow = (unsigned)y0 * (unsigned)NKB * 1024u;   // image row y0 + r at + r ROWB
    const int idxP = ((const int*)Atab)[lane];               // 2:4 positions of the (d = 0, d = ND - 1) pair: the same for every template row
    const int idxLo = 0x44444444, idxHi = (int)0xEEEEEEEEu;  // "positions 0, 1" / "positions 2, 3" of every group of four
    constexpr unsigned ROWB = (unsigned)NKB * 1024u;  // bytes per image row
    constexpr unsigned AROWB = (unsigned)MM_NF_DEV * 1024u;   // bytes per template row in Atab
    constexpr int NP = NXB;         // paired operands per image row: P0 = blocks (0, ND - 1) [, P1 = blocks (1, ND)]
    constexpr int NKD = NKB - 2 * NP;   // plain fragments per image row: blocks NP .. NP + NKD - 1 (slots 2 NP ..)
    static_assert(ND == 7 && (NXB == 1 || NXB == 2), "the sub-block schedule below is written out for 7 Toeplitz blocks");

    i32x16 acc[R][NXB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int xb = 0; xb < NXB; ++xb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][xb][e] = 0;

    // The image-row rotation has period NBUF; the loop body covers one period, so every register index is a
    // constant.  th_pad is a multiple of the period (zero template rows).
    constexpr int PERIOD = NBUF;
    // The sum over template rows may run in any order: every wave walks the rows cyclically from
    // its own start so that, at any moment, all row blocks of a frame group are reading (nearly) the
    // same image rows -- one L2 miss serves the whole group instead of every wave streaming its own
    // 1 MiB through L2 at a different offset.  Block rblk is PERIOD template rows "late" per block
    // while its output rows are only R lower, so neighbouring blocks are PERIOD - R image rows apart.
    // start so that image row y0 + i is roughly the same for every block at any moment (rounded to the
    // rotation period)
    const int istart = i_lo + (((SL - y0 % SL) % SL) / PERIOD) * PERIOD;   // inside the wave's slice [i_lo, i_hi)
    // Image-row registers: the paired operands (8 registers each, B operand of the sparse instruction) and the plain fragments.
    // The incoming row's pieces are requested right after the oldest row's last use of the same piece: the two never live at
    // once, so the ring costs R rows + a fragment or two instead of R + 1 rows.
    i32x8 pp[NBUF][NP];
    i32x4 kd[NBUF][NKD];
    // ONE set of template fragments: the request for the next template row's fragment is placed behind this row's last
    // matrix instruction that reads it -- a full step ahead of its use.  a06 = the compressed pair, ad[0..4] = d = 1..5 dense,
    // a1s / a5s (two column blocks only) = d = 1 / d = 5 laid out for the low / high positions of a paired operand.
    i32x4 a06, ad[5], a1s, a5s;
    // fused window sums: wsa[16 xb + e] of map row y0 for this lane's (frame, half); rwv = the row in flight
    uint32_t wsa[32];
    i32x4 rwv[4];
    auto rw_row = [&](int i) -> unsigned {   // row-window sums of image row y0 + i; template rows >= th are padding (zeros)
        return i < g.th ? (unsigned)min(y0 + i, g.rows - 1) * 4096u : rsR_bytes;
    };
    auto load_pp = [&](i32x8& d, unsigned row, int slot) {
        const i32x4 lo = LD(rsL, row, slot), hi = LD(rsL, row, slot + 1);
        d = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto SP = [&](int r, int xb, const i32x4& a, const i32x8& b, int idx) {
        if (on(r, xb)) acc[r][xb] = __builtin_amdgcn_smfmac_i32_32x32x64_i8(a, b, acc[r][xb], idx, 0, 0);
    };
    auto DN = [&](int r, int xb, const i32x4& a, const i32x4& b) {
        if (on(r, xb)) acc[r][xb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[r][xb], 0, 0, 0);
    };
    // matrix instructions of a sub-block that covers column block xb only / both
    constexpr int NM1_0 = R - ((MFIRST & 1) == 0) - ((MLAST & 1) == 0);                       // xb = 0 (a row's block may belong to the neighbour wave)
    constexpr int NM1_1 = NXB == 2 ? R - ((MFIRST & 2) == 0) - ((MLAST & 2) == 0) : 0;        // xb = 1
    constexpr int NM2 = NM1_0 + NM1_1;
#pragma unroll
    for (int j = 0; j < 32; ++j) wsa[j] = 0;
    // piece p of this template row's R row (requested a step ago) joins the window sums, its additions issued BETWEEN the
    // sub-block's matrix instructions (a vector instruction that issues while a matrix instruction runs is free; a cluster
    // of them in front of the sub-block holds the matrix pipe up)
    auto ws_add = [&](int p) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t v = (uint32_t)rwv[p][c];
            wsa[8 * p + 2 * c] += v & 0xffffu;
            wsa[8 * p + 2 * c + 1] += v >> 16;
            // an empty, zero-instruction asm keeps the optimiser from reasoning through the chain of several
            // hundred additions per accumulator (InstCombine took minutes per instantiation without it)
            asm("" : "+v"(wsa[8 * p + 2 * c]));
            asm("" : "+v"(wsa[8 * p + 2 * c + 1]));
        }
    };
#define MELF_SGB(MFMAS, VALUS) __builtin_amdgcn_sched_group_barrier(0x008, MFMAS, 0); __builtin_amdgcn_sched_group_barrier(0x002, VALUS, 0);
#define MELF_SPREAD(NM) \
    if constexpr ((NM) >= 8) { MELF_SGB(1, 1) MELF_SGB(1, 1) MELF_SGB(1, 1) MELF_SGB(1, 1) MELF_SGB(1, 1) MELF_SGB(1, 1) MELF_SGB(1, 1) MELF_SGB(1, 1) } \
    else if constexpr ((NM) >= 4) { MELF_SGB(1, 2) MELF_SGB(1, 2) MELF_SGB(1, 2) MELF_SGB(1, 2) } \
    else { MELF_SGB(1, 4) MELF_SGB(1, 4) }
#define MELF_FENCE __builtin_amdgcn_sched_barrier(0);
#define MELF_LOAD1 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one matrix instruction, one vector-memory read
#define MELF_LOADS1 MELF_LOAD1
#define MELF_LOADS2 MELF_LOAD1 MELF_LOAD1
#define MELF_LOADS3 MELF_LOAD1 MELF_LOAD1 MELF_LOAD1
    for (int phase = 0; phase < 2; ++phase) {
        const int ibeg = phase == 0 ? istart : i_lo, iend = phase == 0 ? i_hi : istart;
        if (ibeg >= iend) continue;
        // (re-)prime: image rows y0+ibeg .. y0+ibeg+R+PD-2 and template row ibeg
#pragma unroll
        for (int r = 0; r < NBUF - 1; ++r) {
            const unsigned row = Lrow + (unsigned)(ibeg + r) * ROWB;
#pragma unroll
            for (int p = 0; p < NP; ++p) load_pp(pp[r][p], row, 2 * p);
#pragma unroll
            for (int c = 0; c < NKD; ++c) kd[r][c] = LD(rsL, row, 2 * NP + c);
        }
        {
            const unsigned an = (unsigned)ibeg * AROWB;
            a06 = LD(rsA, an, 0);
#pragma unroll
            for (int d = 0; d < 5; ++d) ad[d] = LD(rsA, an, 1 + d);
            if constexpr (NXB == 2) { a1s = LD(rsA, an, 6); a5s = LD(rsA, an, 7); }
            const unsigned rp = rw_row(ibeg);
#pragma unroll
            for (int k = 0; k < 4; ++k) rwv[k] = LD(rsR, rp, k);
        }
        for (int i0 = ibeg; i0 < iend; i0 += PERIOD) {
#pragma unroll
            for (int s = 0; s < PERIOD; ++s) {
                const int i = i0 + s;
                // Loads for later steps are spread over this step's sub-blocks of matrix instructions and pinned there with
                // sched_barrier: they stay in flight for a whole step while the matrix pipe never waits for a burst of load
                // issue.  (Left to itself hipcc sinks each load next to its first use and waits vmcnt(0) every few MFMAs.)
                const unsigned rowin = Lrow + (unsigned)(i + NBUF - 1) * ROWB;   // image row y0 + i + R + PD - 1: first needed at step i + PD
                const unsigned an = (unsigned)(i + 1) * AROWB;             // template row i + 1 (the table carries one extra all-zero row)
                const unsigned rwn = rw_row(i + 1);
                const int inc = (s + NBUF - 1) % NBUF;   // (constant after unrolling)
#define CUR(r) ((s + (r)) % NBUF)
                if constexpr (NXB == 2) {
                    // Six matrix instructions per (row, column block): block xb meets image blocks xb .. xb + 6; its first and last
                    // Toeplitz blocks (d = 0: columns k >= m only; d = 6: k <= m - 5 only) share ONE 2:4-sparse instruction over the
                    // paired operand P_xb = image blocks (xb, xb + 6); d = 1 of block 0 sits in the low positions of P1 and d = 5 of
                    // block 1 in the high positions of P0 (sparse instructions with fixed positions); the rest is dense.
                    // Eight regions (one template fragment each).  A region's loads -- the fragment the region before has finished with,
                    // for the next template row; the incoming image row's pieces whose registers the oldest row has just released; the
                    // next piece of row-window sums -- are issued ONE AFTER EACH of the region's first matrix instructions
                    // (sched_group_barrier): a 1 KiB load takes the wave's issue port for about half a matrix instruction's 32 cycles,
                    // so one per instruction is free while two or three in a row let the pipe run dry (round 5 issued them in
                    // clusters between the regions: ~300 idle cycles per template row, tools/match_clock.py).
                    const unsigned rowinp = rowin - ROWB, anc = (unsigned)i * AROWB;
                    const int incp = (s + NBUF - 2) % NBUF;   // the row that came in during the step before (its last piece is fetched here)
                    // -- q0: the pair, both column blocks
                    MELF_FENCE
                    ad[4] = LD(rsA, anc, 5);                   // d = 5 of THIS template row (its registers were last read at the end of the step before)
                    kd[incp][3] = LD(rsL, rowinp, 7);
#pragma unroll
                    for (int r = 0; r < R; ++r) { SP(r, 0, a06, pp[CUR(r)][0], idxP); SP(r, 1, a06, pp[CUR(r)][1], idxP); }
                    ws_add(0);
                    MELF_LOADS2 MELF_SPREAD(NM2 - 2)
                    // -- q1: d = 5 of column block 1 (image block 6 = high half of P0)
                    MELF_FENCE
                    a06 = LD(rsA, an, 0);
                    rwv[0] = LD(rsR, rwn, 0);
#pragma unroll
                    for (int r = 0; r < R; ++r) SP(r, 1, a5s, pp[CUR(r)][0], idxHi);
                    MELF_LOADS2
                    // -- q2: d = 1 of column block 0 (image block 1 = low half of P1); the oldest row's P0 is free
                    MELF_FENCE
                    a5s = LD(rsA, an, 7);
                    load_pp(pp[inc][0], rowin, 0);
#pragma unroll
                    for (int r = 0; r < R; ++r) SP(r, 0, a1s, pp[CUR(r)][1], idxLo);
                    MELF_LOADS3
                    // -- q3: d = 1 of column block 1 (image block 2)
                    MELF_FENCE
                    a1s = LD(rsA, an, 6);
                    load_pp(pp[inc][1], rowin, 2);
#pragma unroll
                    for (int r = 0; r < R; ++r) DN(r, 1, ad[0], kd[CUR(r)][0]);
                    MELF_LOADS3
                    // -- q4 .. q6: d = 2, 3, 4 of block 0 with d = 2, 3, 4 of block 1
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        MELF_FENCE
                        ad[q] = LD(rsA, an, 1 + q);
                        if (q >= 1) { kd[inc][q - 1] = LD(rsL, rowin, 4 + q - 1); rwv[q] = LD(rsR, rwn, q); }
#pragma unroll
                        for (int r = 0; r < R; ++r) { DN(r, 0, ad[1 + q], kd[CUR(r)][q]); DN(r, 1, ad[1 + q], kd[CUR(r)][q + 1]); }
                        ws_add(1 + q);
                        if (q >= 1) { MELF_LOADS3 MELF_SPREAD(NM2 - 3) } else { MELF_LOADS1 MELF_SPREAD(NM2 - 1) }
                    }
                    // -- q7: d = 5 of column block 0 (image block 5)
                    MELF_FENCE
                    ad[3] = LD(rsA, an, 4);
                    kd[inc][2] = LD(rsL, rowin, 6);
                    rwv[3] = LD(rsR, rwn, 3);
#pragma unroll
                    for (int r = 0; r < R; ++r) DN(r, 0, ad[4], kd[CUR(r)][3]);
                    MELF_LOADS3
                    MELF_FENCE
                } else {
                    // one column block: the pair over P0 = image blocks (0, 6), then d = 1 .. 5 dense over blocks 1 .. 5
                    MELF_FENCE
#pragma unroll
                    for (int r = 0; r < R; ++r) SP(r, 0, a06, pp[CUR(r)][0], idxP);
                    ws_add(0);
                    MELF_SPREAD(NM1_0)
                    MELF_FENCE
                    a06 = LD(rsA, an, 0);
#pragma unroll
                    for (int d = 0; d < 5; ++d) {
                        if (d == 0) load_pp(pp[inc][0], rowin, 0);
                        else kd[inc][d - 1] = LD(rsL, rowin, 2 + d - 1);
                        if (d < 4) rwv[d] = LD(rsR, rwn, d);
                        MELF_FENCE
#pragma unroll
                        for (int r = 0; r < R; ++r) DN(r, 0, ad[d], kd[CUR(r)][d]);
                        if (d < 3) { ws_add(1 + d); MELF_SPREAD(NM1_0) }
                        MELF_FENCE
                        ad[d] = LD(rsA, an, 1 + d);
                    }
                    kd[inc][4] = LD(rsL, rowin, 6);
                    MELF_FENCE
                }
#undef CUR
            }
        }
    }
#undef MELF_SGB
#undef MELF_SPREAD
#undef MELF_FENCE
#undef MELF_LOAD1
#undef MELF_LOADS1
#undef MELF_LOADS2
#undef MELF_LOADS3

#ifdef MELF_MATCH_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_match_loop_end[blockIdx.x] = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (KS > 1) {
        // the slices add up: window sums of the first map row (every wave needs the total: the following rows slide from it),
        // and every (row, column block) tile into the LDS copy its epilogue wave reads back
#pragma unroll
        for (int j = 0; j < 32; ++j) __hip_atomic_fetch_add(&X->wsa[j][lane], wsa[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) {
                if (!on(r, xb)) continue;
                if (!mine(r)) {   // wave-uniform
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        __hip_atomic_fetch_add(&X->acc[r * NXB + xb][e][lane], (uint32_t)acc[r][xb][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) wsa[j] = X->wsa[j][lane];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb) {
                if (!on(r, xb)) continue;
                if (mine(r)) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[r][xb][e] += (int)X->acc[r * NXB + xb][e][lane];
                }
            }
    }
    // ---- epilogue: exact u8 correlation, OpenCV's float post-pass, first-max reduction ----
    const int n = lane & 31, hh = lane >> 5;
    const int f = grp * 32 + n;
    // cc = sum T*L is below 2^31 (119*188*255^2 at most for templates this kernel takes), so the three terms
    // can be added modulo 2^32 and converted with one cvt_f64_u32 (an int64 -> double conversion is four
    // instructions, two of them quarter rate).  A lane visits its elements in increasing raster index, so the
    // first maximum is "strictly greater wins".  One map row at a time: a row's window sums and its accumulators
    // (which leave the accumulator file for the vector ALU) stay within the registers the main loop needs anyway.
    const bool lane_ok = f < g.nframes;
    // The running maximum carries the element's CODE (row, column block, accumulator element: a literal in the select), not its
    // map index: the index is put together once at the end.  Lanes beyond the batch compute on zero rows and never store.
    float bestv = -INFINITY;
    int bestc = -1;
    // fused window sums: the two R rows that leave / enter the window between map rows r - 1 and r are requested one row
    // ahead (while row r - 1's arithmetic runs)
    i32x4 slide[2][4];
    auto slide_load = [&](int r) {
        const unsigned rout = (unsigned)min(y0 + r - 1, g.rows - 1) * 4096u;
        const unsigned rin = (unsigned)min(y0 + r - 1 + g.th, g.rows - 1) * 4096u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { slide[0][k] = LD(rsR, rout, k); slide[1][k] = LD(rsR, rin, k); }
    };
#pragma unroll
    for (int r = 0; r < R; ++r) {
        // window sums of row y0 + r: those of the row above minus the image row that left, plus the one that entered
        if (r > 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t a0 = (uint32_t)slide[1][k][c], b0 = (uint32_t)slide[0][k][c];
                    wsa[8 * k + 2 * c] += (a0 & 0xffffu) - (b0 & 0xffffu);
                    wsa[8 * k + 2 * c + 1] += (a0 >> 16) - (b0 >> 16);
                }
        }
        if (r + 1 < R) slide_load(r + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!mine(r)) continue;   // (wave-uniform) another slice's wave runs this row; the window sums above slide on regardless
        const int y = y0 + r;
        if (y >= g.rh) continue;  // (wave-uniform) rows below the map: the last tile's padding
        // The row's values first (ten instructions per element, nothing lane-dependent in the control flow) ...
        float v[NXB][16];
#pragma unroll
        for (int xb = 0; xb < NXB; ++xb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (!on(r, xb)) continue;
                const uint32_t wsv = wsa[16 * xb + e];
                const uint32_t cc = (uint32_t)acc[r][xb][e] + 128u * wsv + (uint32_t)g.k1;
                double num = (double)cc;
                num -= (double)wsv * g.tmean;
                v[xb][e] = (float)num;
            }
        // ... columns beyond the map (only a column block that sticks out of it has any) ...
#pragma unroll
        for (int xb = 0; xb < NXB; ++xb) {
            if (!on(r, xb) || 32 * xb + 32 <= g.rw) continue;   // (wave-uniform)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (32 * xb + (e & 3) + 8 * (e >> 2) + 4 * hh >= g.rw) v[xb][e] = -INFINITY;
        }
        // ... the maximum: a lane visits its elements in increasing raster index, so the first maximum is "strictly greater wins" ...
#pragma unroll
        for (int xb = 0; xb < NXB; ++xb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (!on(r, xb)) continue;
                if (v[xb][e] > bestv) { bestv = v[xb][e]; bestc = (r * NXB + xb) * 16 + e; }
            }
        // ... and the whole map for the callers that ask for it (tests, melf_match_ccoeff with a result map): one uniform branch per row
        if (result_map) {
            float* mrow = result_map + (size_t)f * g.rh * g.rw + (size_t)y * g.rw;
#pragma unroll
            for (int xb = 0; xb < NXB; ++xb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (!on(r, xb)) continue;
                    const int x = 32 * xb + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (lane_ok && x < g.rw) mrow[x] = v[xb][e];
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    int besti = INT_MAX;
    if (bestc >= 0) {
        const int e = bestc & 15, xb = (bestc >> 4) % NXB, r = (bestc >> 4) / NXB;
        besti = (y0 + r) * g.rw + 32 * xb + (e & 3) + 8 * (e >> 2) + 4 * hh;
    }
    {
        const float ov = __shfl_xor(bestv, 32, 64);
        const int oi = __shfl_xor(besti, 32, 64);
        if (better_m(ov, oi, bestv, besti)) { bestv = ov; besti = oi; }
    }
    if (lane < 32 && f < g.nframes) {
        MatchPartial p;
        p.val = bestv;
        p.idx = besti;
        partials[(size_t)f * g.nparts + rblk * KS + ks] = p;
    }
}


// Registers: the 4-row / 5-row-pair instantiation (1024 frames) takes 477 of the SIMD's 512 (accumulators 8-9 x 16, six
// image rows, the template fragments, 32 window-sum accumulators + the R row in flight), no scratch.  Round 2 capped the
// kernel at 408 ("amdgpu_num_vgpr(204)", spilling 400 bytes per lane) so that a wave of the other caller stream's dials /
// prep kernels fits beside a match wave; with this round's layouts the capped build was never faster than the plain
// one, with one caller stream or with two, and 3x slower for 5-row waves (profiles/r03/match_vgpr_cap_ab.txt): removed.
// The launch's layout (MfmaGeom::na, template RB): every wave carries 2 RB half-row units (RB full map rows x two
// 32-column blocks) or, in a pair, 2 RB + 1 (RB + 1 rows of which the shared middle row counts half) -- so that
// na + 2 pairs waves per frame group fill the chip's 1024 SIMDs in ONE round whatever the batch size: RB = 4 with pairs
// at 1024 frames (8 or 9 units instead of 10), RB = 2 with pairs at 512 (4 or 5 instead of 8), RB = 3 at 640-900 ...
template <int ND, int NXB, int RB, int PD, int KS>
__device__ __forceinline__ void match_block(const int8_t* __restrict__ Lg, const int8_t* __restrict__ Atab,
                                            const uint32_t* __restrict__ ws, const MfmaGeom& g,
                                            float* __restrict__ result_map, MatchPartial* __restrict__ partials, void* lds)
{
    // XCD-aware order: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so
    // ids with equal (id % 8) share an L2.  Give each XCD whole frame groups (they share Lg rows).
    const int nblk = gridDim.x;
    const int id = blockIdx.x;
#ifdef MELF_MATCH_STAMP
    const uint64_t st_clk = __builtin_amdgcn_s_memtime(), st_rt = __builtin_amdgcn_s_memrealtime();
#endif
    const int per = nblk / 8, rem = nblk % 8, xcd = id & 7, sub = id >> 3;
    const int vid = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + sub;
    const int grp = vid / g.ntiles, rblk = vid - grp * g.ntiles;   // one workgroup (KS waves) per tile
    if (NXB == 2 && RB < 5 && rblk >= g.na) {
        // a pair: two (RB + 1)-row waves, the first owns column block 0 of the shared middle row, the second block 1
        const int q = rblk - g.na, base = RB * g.na + (2 * RB + 1) * (q >> 1);
        if ((q & 1) == 0) match_wave<ND, 2, (RB < 5 ? RB + 1 : RB), PD, 3, 1, KS>(Lg, Atab, ws, g, result_map, partials, grp, rblk, base, lds);
        else match_wave<ND, 2, (RB < 5 ? RB + 1 : RB), PD, 2, 3, KS>(Lg, Atab, ws, g, result_map, partials, grp, rblk, base + RB, lds);
    } else {
        match_wave<ND, NXB, RB, PD, 3, 3, KS>(Lg, Atab, ws, g, result_map, partials, grp, rblk, rblk * RB, lds);
    }
#ifdef MELF_MATCH_STAMP
    if (threadIdx.x == 0 && id < 8192) {
        g_match_stamps[4 * id + 0] = st_clk; g_match_stamps[4 * id + 1] = __builtin_amdgcn_s_memtime();
        g_match_stamps[4 * id + 2] = st_rt;  g_match_stamps[4 * id + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int ND, int NXB, int RB, int PD, int KS = 1>
__global__ __launch_bounds__(64 * KS, 1) void k_match_mfma(const int8_t* __restrict__ Lg, const int8_t* __restrict__ Atab,
                                                           const uint32_t* __restrict__ ws, MfmaGeom g,
                                                           float* __restrict__ result_map, MatchPartial* __restrict__ partials)
{
    // The large tiles keep their SIMD to themselves.  They use 440-450 of its 512 registers; whether the 64 or 72 left over admit a
    // wave of the other lane's prep kernel (65 registers, allocated as 72) was an accident of the allocator -- and when it does, both
    // kernels lose: the prep kernel streams the next batch through the L2 this kernel keeps its image rows in (config 5, 512 frames:
    // 0.174 ms per step on two lanes against 0.154 on one; profiles/r06/overlap_*.txt).  Naming a high accumulator register makes
    // the allocation 456: nothing else fits, the other lane's kernels fill the SIMDs as the waves of this launch retire.
#ifndef MELF_NO_SIMD_OWNER   // (experiments only: tools/corun_partner.py builds the kernel without it)
    if constexpr (NXB == 2 && RB >= 4) asm volatile("" ::: "a199");
#endif
    if constexpr (KS > 1) {
        // room for the largest tile of the launch: (RB + 1)-row pair waves
        __shared__ __attribute__((aligned(16))) SliceLds<(NXB == 2 && RB < 5 ? RB + 1 : RB), NXB, KS> lds;
        match_block<ND, NXB, RB, PD, KS>(Lg, Atab, ws, g, result_map, partials, &lds);
    } else {
        match_block<ND, NXB, RB, PD, 1>(Lg, Atab, ws, g, result_map, partials, nullptr);   // ws = row-window sums R in epilogue order
    }
}
// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
constexpr int MM_ND = 7, MM_PD = 1;
constexpr int MM_NF = 8;   // template fragments per row in Atab (mfma_build_atab)
// Template rows are padded with zero rows to a multiple of every wave type's rotation period (R + PD image-row
// buffers): a launch with RB-row waves and pairs of (RB + 1)-row waves needs lcm(RB + 1, RB + 2), without pairs
// RB + 1.  The fragment table carries the largest padding any layout can ask for (60 = lcm(3, 4, 5, 6)).
static int round_up(int v, int m) { return (v + m - 1) / m * m; }
// K slices (ks = 2, 4; rb = 4 only): every slice is a multiple of the period(s)
static int mm_th_pad(int th, int rb, bool pairs, int ks = 1)
{
    static const int lcm2[6] = {0, 0, 12, 20, 30, 6};  // lcm(rb + 1, rb + 2); rb = 5 has no pairs
    return ks * round_up((th + ks - 1) / ks, pairs ? lcm2[rb] : rb + 1);
}
static int mm_th_pad_max(int th) { return std::max(round_up(th, 60), std::max(mm_th_pad(th, 4, true, 2), mm_th_pad(th, 4, true, 4))); }

bool mfma_match_ok(int th, int tw, int rows, int cols)
{
    const int rh = rows - th + 1, rw = cols - tw + 1;
    const int nd = (tw + 31 + 31) / 32;  // Toeplitz blocks per template row
    // th * tw * 255^2 < 2^32: the epilogue adds the correlation's three terms modulo 2^32
    // tw <= 32 (nd - 1): the first and last Toeplitz blocks of a template row never meet in a group of four (mfma_build_atab)
    return rh >= 1 && rw >= 1 && rw <= 64 && nd == MM_ND && tw <= 32 * (MM_ND - 1) && cols <= 32 * (MM_ND + (rw > 32 ? 2 : 1) - 1) &&
           (long)th * tw * 65025L < (1L << 32);
}

// Cost model of one wave, in shader cycles (round 6; fitted to tools/match_clock.py and tools/match_sweep.py, profiles/r06/
// match_sweep_round6.txt): 32 cycles per matrix instruction; what a 1 KiB operand load adds on top depends on how many matrix
// instructions a region has to hide it behind -- 7 cycles for waves of four rows or more (one load per instruction is free), 17 for
// three rows, 30 for two; the epilogue (double-precision post-pass, arg-max) ~1 300 cycles per half-row unit + ~300 per map row (its
// window sums slide in), priming ~4 000.
// K slices (ks > 1): a wave runs th_pad / ks template rows of the whole tile and the epilogue of 1 / ks of its units; the exchange
// through LDS (zeroing, ~150 ds_add + as many reads, two barriers -- one of them waits for the tile's slowest slice) ~6 000.
static double mm_wave_cycles(int R, int units, int nxb, int th_pad, int ks = 1)
{
    const int nkb = MM_ND + nxb - 1;   // image-row fragments per step; + the template fragments + 4 pieces of the R row
    const int nf = nxb == 2 ? 8 : 6;   // template fragments per step; MM_ND - 1 matrix instructions per unit (the first and last Toeplitz blocks pair up)
    const double load = R >= 4 ? 7.0 : (R == 3 ? 17.0 : 30.0);
    return (double)(th_pad / ks) * ((double)units * (MM_ND - 1) * 32.0 + (double)(nkb + nf + 4) * load) + (double)((units + ks - 1) / ks) * 1300.0 + 4000.0 + R * 300.0 +
           (ks > 1 ? 6000.0 : 0.0);
}

MfmaPlan mfma_plan(int th, int tw, int rows, int cols, int nframes)
{
    MfmaPlan p = {};
    p.rh = rows - th + 1;
    p.rw = cols - tw + 1;
    p.nxb = p.rw > 32 ? 2 : 1;
    p.nkb = MM_ND + p.nxb - 1;
    p.groups = (nframes + 31) / 32;
    // Layout search: RB full rows per wave (2..5), np pairs of (RB + 1)-row waves sharing their middle row.  The chip is
    // power-limited in this kernel: a launch takes the LONGER of (rounds of waves over the 1024 SIMDs) x (its longest
    // wave at the clock a part-filled chip reaches) and (the cycles of ALL its waves) / 1024 at the clock the full chip holds -- 1.5
    // times lower (2.2 against 1.46 GHz-equivalents in tools/match_sweep.py).  So fewer, larger tiles beat a layout that fills every
    // SIMD with smaller ones: 4-row waves on 594-990 SIMDs for 576-992 frames (rounds 3-5 ran 2- and 3-row layouts there: 5-7 %
    // slower).  Among equals the fewest pairs.
    const int simds = 1024;
    int force_rb = 0, force_np = -1, force_ks = 0;
    if (const char* e = getenv("MELF_MATCH_LAYOUT")) {  // experiments / tests: "rb,np[,ks]"; anything outside the family is ignored
        if (sscanf(e, "%d,%d,%d", &force_rb, &force_np, &force_ks) < 1 || force_rb < 2 || force_rb > 5) { force_rb = 0; force_np = -1; force_ks = 0; }
        if (force_ks != 1 && force_ks != 2 && force_ks != 4) force_ks = force_rb ? 1 : 0;   // "rb,np" alone: no slices (as before round 5)
        if (force_ks > 1 && force_rb != 4) force_ks = 1;
    }
    double best = 0;
    p.rb = 0;
    p.ks = 1;
    for (int ks = 1; ks <= 4; ks *= 2) {
        if (force_ks && ks != force_ks) continue;
        for (int rb = 2; rb <= 5; ++rb) {
            if (force_rb && rb != force_rb) continue;
            if (ks > 1 && rb != 4) continue;   // K slices are instantiated for the 4-row tiles (the 1024-frame layout's operand reuse)
            const int np_max = (p.nxb == 2 && rb < 5) ? (p.rh + 2 * rb) / (2 * rb + 1) : 0;
            auto waves_of = [&](int np) -> long {
                const int rest = p.rh - (2 * rb + 1) * np;
                return (long)((rest > 0 ? (rest + rb - 1) / rb : 0) + 2 * np) * p.groups * ks;
            };
            // candidates: no pairs, and the FEWEST pairs that reach the fewest rounds any number of pairs reaches
            long rounds_min = LONG_MAX;
            for (int np = 0; np <= np_max; ++np) rounds_min = std::min(rounds_min, (waves_of(np) + simds - 1) / simds);
            int np_few = 0;
            while (np_few < np_max && (waves_of(np_few) + simds - 1) / simds > rounds_min) ++np_few;
            for (int np = 0; np <= np_max; ++np) {
                if (force_np >= 0) { if (np != std::min(force_np, np_max)) continue; }
                else if (np != 0 && np != np_few) continue;
                const int rest = p.rh - (2 * rb + 1) * np;
                const int na = rest > 0 ? (rest + rb - 1) / rb : 0;
                const long waves = (long)(na + 2 * np) * p.groups * ks;
                const long rounds = (waves + simds - 1) / simds;
                const int th_pad = mm_th_pad(th, rb, np > 0, ks);
                const int upr = p.nxb;  // units per full row
                const double w_full = mm_wave_cycles(rb, upr * rb, p.nxb, th_pad, ks), w_pair = mm_wave_cycles(rb + 1, upr * rb + 1, p.nxb, th_pad, ks);
                const double longest = np > 0 ? w_pair : w_full;
                const double total = (double)p.groups * ks * ((double)na * w_full + 2.0 * np * w_pair);
                // full rounds run at the full chip's (power-limited) rate, a last part-filled round at the faster of its longest wave and its share
                const double w_avg = total / (double)std::max<long>(waves, 1);
                const long full = waves / simds, part = waves - full * simds;
                const double cost = (double)full * 1.5 * w_avg + (part > 0 ? std::max(longest, 1.5 * (double)part * w_avg / simds) : 0.0);
                (void)rounds;
                if (!p.rb || cost < best * 0.98) {
                    best = cost;
                    p.rb = rb; p.na = na; p.np = np; p.th_pad = th_pad; p.ks = ks;
                }
            }
        }
    }
    p.ntiles = p.na + 2 * p.np;
    p.nparts = p.ntiles * p.ks;
    const int rows_cov = p.rb * p.na + (2 * p.rb + 1) * p.np;
    p.rows_pad = rows_cov + p.th_pad + MM_PD + 1;   // last row touched: y0 + (th_pad - 1) + R + PD - 1 (prefetched, unused)
    p.lg_bytes = (size_t)p.groups * p.rows_pad * p.nkb * 1024;
    p.r_bytes = (size_t)p.groups * rows * 64 * 32 * sizeof(uint16_t);
    return p;
}

// Atab: a 1 KiB header (the pair's index dword of every lane), then MM_NF fragments of 1 KiB per template row
size_t mfma_atab_bytes(int th)
{
    return 1024 + (size_t)(mm_th_pad_max(th) + MM_PD) * MM_NF * 1024;
}

// The template as matrix-instruction A operands.  With T'[i][c] = templ - 128 inside the template and 0 outside, Toeplitz block d of
// template row i is A_d[m][k] = T'[i][32 d + k - m] (m = map column inside a 32-column block, k = image column inside a 32-column
// image block).  Per template row, fragments of 64 lanes x 16 bytes:
//   f = 1 .. 5: A_d, d = f, dense: lane l byte j = A_d[l & 31][16 (l >> 5) + j]               (v_mfma_i32_32x32x32_i8)
//   f = 0:      A_0 and A_6 in ONE 2:4-sparse operand over the paired image operand P (dword t of P's lane (n, h) = image block
//               kb, columns 16 h + 2 t, + 1, then block kb + 6, same columns): A_0 is non-zero only for k >= m, A_6 only for
//               k <= m - (193 - tw), so of the four candidates of a group {A_0[k0], A_0[k0 + 1], A_6[k0], A_6[k0 + 1]} at most two
//               are ever inside the template -- for tw <= 192.  Compressed byte ja of lane (m, hA) belongs to the group of P's
//               lane half hB = ja >> 3, dword t = 4 hA + ((ja >> 1) & 3), i.e. k0 = 16 hB + 2 t (v_smfmac_i32_32x32x64_i8; the map was
//               found by probing the instruction: tools/ubench/smfmac_i8.hip); its 2-bit position goes to bits 2 ja of the lane's
//               index dword, which depends on (m, hA, ja) only and is stored once, in the header
//   f = 6:      A_1 for the LOW positions of a paired operand (image block 1 lives in P1 = blocks (1, 7)): bytes A_1[m][k0], A_1[m][k0 + 1], index 0x4 per group
//   f = 7:      A_5 for the HIGH positions (image block 6 lives in P0 = blocks (0, 6)): index 0xE per group
void mfma_build_atab(const uint8_t* templ, int th, int tw, int8_t* atab)
{
    const int th_pad = mm_th_pad_max(th);
    uint32_t* idx = (uint32_t*)atab;
    memset(atab, 0, 1024);
    int8_t* fr = atab + 1024;
    auto cand_cols = [&](int m, int hA, int ja, int cols[4]) {
        const int hB = ja >> 3, t = 4 * hA + ((ja >> 1) & 3), k0 = 16 * hB + 2 * t;
        cols[0] = k0 - m; cols[1] = k0 + 1 - m; cols[2] = 32 * (MM_ND - 1) + k0 - m; cols[3] = cols[2] + 1;
        return k0;
    };
    // positions of the pair's two compressed bytes per group: the candidates inside the template, in increasing position,
    // filled up with unused positions (their bytes are zero)
    int pos[64][16];
    for (int l = 0; l < 64; ++l)
        for (int ja = 0; ja < 16; ja += 2) {
            int cols[4], cand[4], nc = 0;
            cand_cols(l & 31, l >> 5, ja, cols);
            for (int v = 0; v < 4; ++v) if (cols[v] >= 0 && cols[v] < tw) cand[nc++] = v;
            if (nc > 2) { fprintf(stderr, "[melf] mfma_build_atab: template of %d columns does not pair up\n", tw); abort(); }
            int v0 = 0, v1 = 1;
            if (nc == 2) { v0 = cand[0]; v1 = cand[1]; }
            else if (nc == 1) { v0 = cand[0] < 3 ? cand[0] : 0; v1 = 3; }
            pos[l][ja] = v0; pos[l][ja + 1] = v1;
            idx[l] |= ((uint32_t)v0 << (2 * ja)) | ((uint32_t)v1 << (2 * ja + 2));
        }
    for (int i = 0; i < th_pad + MM_PD; ++i) {
        auto Tq = [&](int col) -> int8_t { return (i < th && col >= 0 && col < tw) ? (int8_t)((int)templ[(size_t)i * tw + col] - 128) : (int8_t)0; };
        int8_t* row = fr + (size_t)i * MM_NF * 1024;
        for (int l = 0; l < 64; ++l) {
            const int m = l & 31, hA = l >> 5;
            for (int d = 1; d <= 5; ++d)
                for (int j = 0; j < 16; ++j) row[((size_t)d * 64 + l) * 16 + j] = Tq(32 * d + 16 * hA + j - m);
            for (int ja = 0; ja < 16; ++ja) {
                int cols[4];
                const int k0 = cand_cols(m, hA, ja & ~1, cols);
                row[((size_t)0 * 64 + l) * 16 + ja] = Tq(cols[pos[l][ja]]);
                row[((size_t)6 * 64 + l) * 16 + ja] = Tq(32 * 1 + k0 + (ja & 1) - m);
                row[((size_t)7 * 64 + l) * 16 + ja] = Tq(32 * 5 + k0 + (ja & 1) - m);
            }
        }
    }
}

void launch_match_prep(const MatchSrc& src, bool from_bgr, int n, int groups, int rows_pad, int nkb, int rwp, int tw, int8_t* d_lg,
                       uint16_t* d_r, hipStream_t stream, int pairs)
{
    dim3 grid(rows_pad, groups), block(256);
    const size_t pre_bytes = (size_t)32 * (nkb * 32 + 8) * sizeof(int16_t);
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_set[64] = {false};
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {  // once per device: dynamic LDS beyond the 64 KiB default
        (void)hipFuncSetAttribute((const void*)k_prep_lplane<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute((const void*)k_prep_lplane<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set[dev] = true;
    }
    if (from_bgr) hipLaunchKernelGGL((k_prep_lplane<true>), grid, block, pre_bytes, stream, src, n, nkb, rows_pad, tw, rwp, pairs, d_lg, d_r);
    else hipLaunchKernelGGL((k_prep_lplane<false>), grid, block, pre_bytes, stream, src, n, nkb, rows_pad, tw, rwp, pairs, d_lg, d_r);
}

void launch_mfma_prep(const MatchSrc& src, bool from_bgr, int n, const MfmaPlan& p, int th, int tw, int8_t* d_lg,
                      uint16_t* d_r, hipStream_t stream)
{
    (void)th;
    launch_match_prep(src, from_bgr, n, p.groups, p.rows_pad, p.nkb, 64, tw, d_lg, d_r, stream, p.nxb);   // one paired operand per column block
}

template <int NXB, int RB, int KS>
static void launch_mm(dim3 grid, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, const int8_t* d_lg,
                      const int8_t* d_atab, const uint32_t* d_ws, const MfmaGeom& g, float* d_result_map, MatchPartial* d_partials)
{
    // ev_start / ev_stop (optional): time stamps taken by the dispatch itself (hipExtLaunchKernelGGL) -- no
    // hipEventRecord barrier packets in the queue around the kernel
    hipExtLaunchKernelGGL((k_match_mfma<MM_ND, NXB, RB, MM_PD, KS>), grid, dim3(64 * KS), 0, stream, ev_start, ev_stop, 0, d_lg, d_atab, d_ws, g,
                              d_result_map, d_partials);
}

void launch_mfma_match(int n, const MfmaPlan& p, int th, int tw, long tsum, double tmean, const int8_t* d_atab,
                       const int8_t* d_lg, const uint32_t* d_ws, float* d_result_map, MatchPartial* d_partials,
                       hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    MfmaGeom g;
    g.rh = p.rh; g.rw = p.rw; g.rows_pad = p.rows_pad; g.th_pad = p.th_pad; g.nframes = n; g.nparts = p.nparts;
    g.na = p.na; g.ntiles = p.ntiles;
    g.k1 = (int)(128 * (tsum - 128L * th * tw));
    g.rows = p.rh + th - 1; g.th = th;
    g.tmean = tmean;
    dim3 grid(p.ntiles * p.groups);
#define MM_CASE(NXB_, RB_) \
    case NXB_ * 8 + RB_: launch_mm<NXB_, RB_, 1>(grid, stream, ev_start, ev_stop, d_lg, d_atab, d_ws, g, d_result_map, d_partials); break;
#define MM_CASE_KS(NXB_, KS_) \
    case 64 * KS_ + NXB_ * 8 + 4: launch_mm<NXB_, 4, KS_>(grid, stream, ev_start, ev_stop, d_lg, d_atab, d_ws, g, d_result_map, d_partials); break;
    switch ((p.ks > 1 ? 64 * p.ks : 0) + p.nxb * 8 + p.rb) {
#ifdef MELF_MATCH_ONLY_RB4   // experiments: one instantiation (fast compile)
        MM_CASE(2, 4)
#else
        MM_CASE(1, 2) MM_CASE(1, 3) MM_CASE(1, 4) MM_CASE(1, 5)
        MM_CASE(2, 2) MM_CASE(2, 3) MM_CASE(2, 4) MM_CASE(2, 5)
        MM_CASE_KS(1, 2) MM_CASE_KS(2, 2) MM_CASE_KS(1, 4) MM_CASE_KS(2, 4)
#endif
        default:   // a plan outside the instantiated family must never pass silently: the records would come from stale partials
            fprintf(stderr, "[melf] k_match_mfma: no instantiation for %d column blocks x %d rows per wave x %d K slices\n", p.nxb, p.rb, p.ks);
            abort();
    }
#undef MM_CASE
#undef MM_CASE_KS
}

}  // namespace melf
