// K1  -- convert_to_hls: cv2.cvtColor(BGR2HLS_FULL) + uint8 hue shift
//        (reference: meterelf/_utils.py:100-102)
// K1b -- the fused full-frame stage of BASELINE config 2:
//        HLS(+shift) -> inRange(fixed needle bounds) -> dilate 3x3 -> erode 3x3
//        (reference: meterelf/_utils.py:113-119 get_mask_by_color with the bounds
//        of meterelf/_calibration.py:82-84; closing as meterelf/_reading.py:128-130)
//
// K1b is HBM-bound by design: 3 B/px read (NHWC BGR) + 1 B/px written.  A
// workgroup owns a strip of FM_ROWS full-width rows; pass 1 turns 4-pixel groups
// (three aligned dword loads) into in-range nibbles that are OR-ed into a
// bit-packed LDS image (1 bit per pixel, 2 halo rows above and below); passes 2
// and 3 do the closing on 32-pixel words with carries between neighbouring words;
// pass 4 expands bits to bytes, one dword store per 4 pixels.
#include <limits.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include "melf_device.h"
#include <hip/hip_ext.h>

#include "melf_internal.h"

namespace melf {

__global__ __launch_bounds__(256) void k_bgr2hls(const uint8_t* __restrict__ src, int rows, int cols,
                                                 size_t row_stride, int hue_shift, uint8_t* __restrict__ dst)
{
    const size_t total = (size_t)rows * cols;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(p / cols), x = (int)(p - (size_t)y * cols);
        const uint8_t* s = src + (size_t)y * row_stride + (size_t)x * 3;
        int H, L, S;
        hls_pixel(s[0], s[1], s[2], hls_scalar_tail(x, cols), hue_shift, H, L, S);
        uint8_t* d = dst + p * 3;
        d[0] = (uint8_t)H; d[1] = (uint8_t)L; d[2] = (uint8_t)S;
    }
}

void launch_bgr2hls(const uint8_t* d_src, int rows, int cols, size_t row_stride, int hue_shift, uint8_t* d_dst,
                    hipStream_t stream)
{
    const size_t total = (size_t)rows * cols;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_bgr2hls, dim3(blocks), dim3(256), 0, stream, d_src, rows, cols, row_stride, hue_shift, d_dst);
}

struct Bounds {
    int lo[3], hi[3];
};

constexpr int FM_ROWS = 32;  // output rows per workgroup

__device__ inline bool in_bounds(int b, int g, int r, bool tail, int hue_shift, const Bounds& B)
{
    int H, L, S;
    hls_pixel(b, g, r, tail, hue_shift, H, L, S);
    return H >= B.lo[0] && H <= B.hi[0] && L >= B.lo[1] && L <= B.hi[1] && S >= B.lo[2] && S <= B.hi[2];
}

// grid: (ceil(H / FM_ROWS), n).  LDS: two bit images of (FM_ROWS + 4) x wpr dwords.
__global__ __launch_bounds__(256) void k_fused_mask(const uint8_t* __restrict__ frames, int H, int W, int hue_shift,
                                                    Bounds B, uint8_t* __restrict__ masks)
{
    extern __shared__ uint32_t lds[];
    const int wpr = (W + 31) >> 5;      // dwords per bit row
    const int nrow = FM_ROWS + 4;       // rows y0-2 .. y0+FM_ROWS+1
    uint32_t* raw = lds;                // in-range bits
    uint32_t* dil = lds + nrow * wpr;   // dilated bits (neutral = 1 outside the image)
    const int y0 = blockIdx.x * FM_ROWS;
    const uint8_t* frame = frames + (size_t)blockIdx.y * H * W * 3;
    uint8_t* out = masks + (size_t)blockIdx.y * H * W;
    const int tid = threadIdx.x;

    for (int i = tid; i < nrow * wpr; i += 256) raw[i] = 0;
    __syncthreads();

    // pass 1: in-range bits, 4 pixels per work item
    const int G = (W + 3) >> 2;
    const bool aligned = (W & 3) == 0 && ((size_t)frame & 3) == 0;
    for (int it = tid; it < nrow * G; it += 256) {
        const int row = it / G, g = it - row * G;
        const int y = y0 - 2 + row;
        if (y < 0 || y >= H) continue;
        const int x = g * 4;
        uint32_t nib = 0;
        const uint8_t* p = frame + ((size_t)y * W + x) * 3;
        if (aligned) {
            const uint32_t* q = (const uint32_t*)p;
            const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
            nib |= in_bounds(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255, hls_scalar_tail(x, W), hue_shift, B) ? 1u : 0u;
            nib |= in_bounds(w0 >> 24, w1 & 255, (w1 >> 8) & 255, hls_scalar_tail(x + 1, W), hue_shift, B) ? 2u : 0u;
            nib |= in_bounds((w1 >> 16) & 255, w1 >> 24, w2 & 255, hls_scalar_tail(x + 2, W), hue_shift, B) ? 4u : 0u;
            nib |= in_bounds((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24, hls_scalar_tail(x + 3, W), hue_shift, B) ? 8u : 0u;
        } else {
            for (int k = 0; k < 4 && x + k < W; ++k)
                nib |= in_bounds(p[3 * k], p[3 * k + 1], p[3 * k + 2], hls_scalar_tail(x + k, W), hue_shift, B) ? (1u << k) : 0u;
        }
        if (nib) atomicOr(&raw[row * wpr + (x >> 5)], nib << (x & 31));
    }
    __syncthreads();

    // pass 2: dilate rows 1 .. nrow-2 (image rows y0-1 .. y0+FM_ROWS)
    const uint32_t lastmask = (W & 31) ? ((1u << (W & 31)) - 1u) : 0xffffffffu;
    for (int i = tid; i < (nrow - 2) * wpr; i += 256) {
        const int row = 1 + i / wpr, k = i % wpr;
        const int y = y0 - 2 + row;
        uint32_t v = 0xffffffffu;
        if (y >= 0 && y < H) {
            v = 0;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const uint32_t* r = raw + (row + dy) * wpr;  // rows outside the image hold zeros
                const uint32_t c = r[k], l = k > 0 ? r[k - 1] : 0u, rr = k + 1 < wpr ? r[k + 1] : 0u;
                v |= c | (c << 1) | (l >> 31) | (c >> 1) | (rr << 31);
            }
            if (k == wpr - 1) v |= ~lastmask;  // columns >= W are neutral for the erosion
        }
        dil[row * wpr + k] = v;
    }
    __syncthreads();

    // pass 3 + 4: erode rows 2 .. nrow-3 (image rows y0 .. y0+FM_ROWS-1) and store bytes
    const int Gw = wpr * 8;  // 4-pixel groups per padded bit row
    for (int it = tid; it < FM_ROWS * Gw; it += 256) {
        const int rr = it / Gw, g = it - rr * Gw;
        const int row = 2 + rr, y = y0 + rr;
        const int x = g * 4;
        if (y >= H || x >= W) continue;
        const int k = x >> 5;
        uint32_t v = 0xffffffffu;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const uint32_t* r = dil + (row + dy) * wpr;
            const uint32_t c = r[k], l = k > 0 ? r[k - 1] : 0xffffffffu, rn = k + 1 < wpr ? r[k + 1] : 0xffffffffu;
            v &= c & ((c << 1) | (l >> 31)) & ((c >> 1) | (rn << 31));
        }
        const uint32_t nib = (v >> (x & 31)) & 15u;
        uint8_t* o = out + (size_t)y * W + x;
        if (aligned && x + 3 < W) {
            const uint32_t bytes = ((nib & 1u) * 0xffu) | ((nib & 2u) * (0xff00u >> 1)) | ((nib & 4u) * (0xff0000u >> 2)) |
                                   ((nib & 8u) * (0xff000000u >> 3));
            *(uint32_t*)o = bytes;
        } else {
            for (int j = 0; j < 4 && x + j < W; ++j) o[j] = (nib >> j & 1u) ? 255 : 0;
        }
    }
}

// ---------------------------------------------------------------------------
// K1b fast path: table-driven in-range test, exact by construction.
//
// The float32 HLS formulas are piecewise functions of small integers:
//   * L and S depend only on the (vmax, vmin) pair               -> 64 Ki entries, 1 bit
//   * H depends on which channel is the maximum, on diff = vmax - vmin and on the
//     signed numerator num (g-b, b-r or r-g), EXCEPT that at exact rounding ties
//     the float32 error of the particular (vmax, vmin, mid) triple decides.
// k_build_fused_tables evaluates the exact float path (hls_pixel) for all 2^24
// BGR triples and records, per (case, diff, num) entry, whether an in-range and/or
// an out-of-range H was seen (2 bits).  01 = in, 10 = out, 11 = depends on the
// triple: the main kernel then re-evaluates that pixel with the exact float path.
// Entry index = case * 65536 + diff^2 + diff + num  (sum_{d<diff}(2d+1) = diff^2).
// ---------------------------------------------------------------------------
constexpr int HUE2_DWORDS = 3 * 65536 * 2 / 32;  // 12288 (48 KiB): 2 bits per entry (seen in / seen out)
constexpr int LS_DWORDS = 65536 / 32;            // 2048  (8 KiB)
constexpr int HUE1_DWORDS = 3 * 65536 / 32;      // 6144  (24 KiB): 1 bit per entry (in), valid when nothing is ambiguous
constexpr int HUES_DWORDS = 512 * 512 / 32;      // 8192  (32 KiB): one hue sector as [diff + 255][num + 256] bits;
                                                 // rows with diff < |num| (pixel not in this sector) stay all-zero
// interval form of the single-sector tables: one (lo : i16, count : u16) entry per row
//   lsI[sector][P]        : vmin in [lo, lo + count) passes the L and S bounds when vmax = P
//   hI[sector][diff + 256]: num  in [lo, lo + count) passes the H bounds
// valid only if every row's set bits are one contiguous run (checked at build time).
constexpr int LSI_ROWS = 256, HI_ROWS = 512;
// global table buffer: [hue2][ls][hue1][hueS x 3][lsI x 3][hI x 3][ties, active sectors, non-interval rows x 3]
constexpr int OFF_LS = HUE2_DWORDS, OFF_HUE1 = OFF_LS + LS_DWORDS, OFF_HUES = OFF_HUE1 + HUE1_DWORDS,
              OFF_LSI = OFF_HUES + 3 * HUES_DWORDS, OFF_HI = OFF_LSI + 3 * LSI_ROWS,
              OFF_COUNT = OFF_HI + 3 * HI_ROWS, OFF_ACTIVE = OFF_COUNT + 1, OFF_NONIV = OFF_ACTIVE + 1;
static_assert(OFF_NONIV + 3 <= FUSED_TABLE_DWORDS, "table buffer too small");

// e = case * 65536 + diff * (diff + 1) + num.  All three arms are computed and
// then selected, so that the compiler emits v_cndmask instead of branches.
__device__ __forceinline__ uint32_t hue_entry(int b, int g, int r, int vmax, int vmin)
{
    const int diff = vmax - vmin;
    const int ep_r = g - b, ep_g = b - r + 65536, ep_b = r - g + 131072;
    int ep = ep_b;
    ep = (g == vmax) ? ep_g : ep;
    ep = (r == vmax) ? ep_r : ep;
    return (uint32_t)(diff * (diff + 1) + ep);
}

__global__ __launch_bounds__(256) void k_build_fused_tables(int hue_shift, Bounds B, uint32_t* __restrict__ tables)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;  // 2^24 threads: one BGR triple each
    const int b = t & 255, g = (t >> 8) & 255, r = (t >> 16) & 255;
    int H, L, S;
    hls_pixel(b, g, r, false, hue_shift, H, L, S);
    const int vmax = max(max(b, g), r), vmin = min(min(b, g), r);
    const uint32_t e = hue_entry(b, g, r, vmax, vmin);
    const bool hin = H >= B.lo[0] && H <= B.hi[0];
    const uint32_t hm = (hin ? 1u : 2u) << ((e & 15u) * 2u);
    uint32_t* hw = tables + (e >> 4);
    if ((*(volatile uint32_t*)hw & hm) == 0) atomicOr(hw, hm);
    const bool lsin = L >= B.lo[1] && L <= B.hi[1] && S >= B.lo[2] && S <= B.hi[2];
    if (lsin) {
        const uint32_t li = (uint32_t)(vmax << 8 | vmin);
        uint32_t* lw = tables + OFF_LS + (li >> 5);
        const uint32_t lm = 1u << (li & 31u);
        if ((*(volatile uint32_t*)lw & lm) == 0) atomicOr(lw, lm);
    }
}

// hue1[e] = (hue2[e] == 01); counts entries with hue2[e] == 11
__global__ __launch_bounds__(256) void k_finish_fused_tables(uint32_t* __restrict__ tables)
{
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;  // one hue1 dword = 32 entries = 2 hue2 dwords
    if (w >= (uint32_t)HUE1_DWORDS) return;
    uint32_t out = 0, amb = 0;
    for (int h = 0; h < 2; ++h) {
        const uint32_t v = tables[2 * w + h];
        for (int k = 0; k < 16; ++k) {
            const uint32_t two = (v >> (2 * k)) & 3u;
            out |= (two == 1u ? 1u : 0u) << (16 * h + k);
            amb += two == 3u ? 1u : 0u;
        }
    }
    tables[OFF_HUE1 + w] = out;
    if (amb) atomicAdd(tables + OFF_COUNT, amb);
    // the same bits again in the per-sector [diff][num + 256] layout of the single-sector kernels
    for (int k = 0; k < 32; ++k) {
        if (!((out >> k) & 1u)) continue;
        const uint32_t e = w * 32u + (uint32_t)k, c = e >> 16, local = e & 65535u;
        int diff = (int)sqrtf((float)local);
        while (diff * diff > (int)local) --diff;
        while ((diff + 1) * (diff + 1) <= (int)local) ++diff;
        const int num = (int)local - diff * (diff + 1);  // diff*diff <= local <= diff*diff + 2*diff
        const uint32_t se = (uint32_t)((diff + 255) * 512 + num + 256);
        atomicOr(tables + OFF_HUES + c * HUES_DWORDS + (se >> 5), 1u << (se & 31u));
        atomicOr(tables + OFF_ACTIVE, 1u << c);
    }
}

// One thread per table row: turns the row's bits into (lo, count) and counts rows whose set bits
// are not one contiguous run.  Rows 0..255 of each sector: LS rows (vmax = P = row; the LS bit of
// (P, vmin) is sector-independent); rows 256..767: hue rows (diff = row - 512).
__global__ __launch_bounds__(256) void k_interval_tables(uint32_t* __restrict__ tables)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 3 * (LSI_ROWS + HI_ROWS)) return;
    const int c = t / (LSI_ROWS + HI_ROWS), row = t - c * (LSI_ROWS + HI_ROWS);
    int lo = 0, count = 0, runs = 0;
    bool prev = false;
    if (row < LSI_ROWS) {
        for (int m = 0; m < 256; ++m) {
            const uint32_t li = (uint32_t)(row << 8 | m);
            const bool b = (tables[OFF_LS + (li >> 5)] >> (li & 31u)) & 1u;
            if (b && !prev) { ++runs; if (runs == 1) lo = m; }
            if (b && runs == 1) ++count;
            prev = b;
        }
        tables[OFF_LSI + c * LSI_ROWS + row] = ((uint32_t)count << 16) | (uint32_t)(lo & 0xffff);
    } else {
        const int hr = row - LSI_ROWS;  // 0..511 = diff + 256; the bit table has rows diff + 255 in 0..510
        if (hr >= 1) {
            for (int nn = 0; nn < 512; ++nn) {
                const uint32_t se = (uint32_t)((hr - 1) * 512 + nn);
                const bool b = (tables[OFF_HUES + c * HUES_DWORDS + (se >> 5)] >> (se & 31u)) & 1u;
                if (b && !prev) { ++runs; if (runs == 1) lo = nn - 256; }
                if (b && runs == 1) ++count;
                prev = b;
            }
        }
        tables[OFF_HI + c * HI_ROWS + hr] = ((uint32_t)count << 16) | (uint32_t)(lo & 0xffff);
    }
    if (runs > 1) atomicAdd(tables + OFF_NONIV + c, 1u);
}

// The work queue of a launch is one of the FUSED_QUEUE_SLOTS counters behind the context's tables, and the launch's last workgroup
// leaves it zeroed.  A slot belongs to ONE stream of ONE context: launches of a stream run one after the other, so the next launch
// finds its counter at zero, while launches on different streams -- which may run together -- never share one, however many are
// enqueued.  (Round 5 rotated the slots per launch: two launches 64 apart on different streams could have met in one counter.)
// A context with more streams than slots: -1, and that launch takes the static split.  Building a context's tables (which zeroes
// the counters) forgets the streams an earlier context at the same address had.
static std::mutex g_fused_slot_m;
static std::map<std::pair<const void*, hipStream_t>, int> g_fused_slot_of;
static std::map<const void*, int> g_fused_slots_used;
static int fused_queue_slot(const uint32_t* d_tables, hipStream_t stream)
{
    std::lock_guard<std::mutex> g(g_fused_slot_m);
    auto it = g_fused_slot_of.find({d_tables, stream});
    if (it != g_fused_slot_of.end()) return it->second;
    int& n = g_fused_slots_used[d_tables];
    if (n >= FUSED_QUEUE_SLOTS) return -1;
    g_fused_slot_of[{d_tables, stream}] = n;
    return n++;
}
static void fused_queue_forget(const uint32_t* d_tables)
{
    std::lock_guard<std::mutex> g(g_fused_slot_m);
    for (auto it = g_fused_slot_of.begin(); it != g_fused_slot_of.end();) it = it->first.first == d_tables ? g_fused_slot_of.erase(it) : std::next(it);
    g_fused_slots_used.erase(d_tables);
}

void launch_build_fused_tables(int hue_shift, const int lo[3], const int hi[3], uint32_t* d_tables, hipStream_t stream)
{
    fused_queue_forget(d_tables);
    Bounds B;
    for (int c = 0; c < 3; ++c) { B.lo[c] = lo[c]; B.hi[c] = hi[c]; }
    (void)hipMemsetAsync(d_tables, 0, FUSED_BUF_DWORDS * sizeof(uint32_t), stream);   // tables and work queues
    hipLaunchKernelGGL(k_build_fused_tables, dim3(65536), dim3(256), 0, stream, hue_shift, B, d_tables);
    hipLaunchKernelGGL(k_finish_fused_tables, dim3(HUE1_DWORDS / 256), dim3(256), 0, stream, d_tables);
    hipLaunchKernelGGL(k_interval_tables, dim3((3 * (LSI_ROWS + HI_ROWS) + 255) / 256), dim3(256), 0, stream, d_tables);
}
int fused_tables_noniv_offset() { return OFF_NONIV; }
int fused_tables_count_offset() { return OFF_COUNT; }
int fused_tables_active_offset() { return OFF_ACTIVE; }


constexpr int FUSED_MAX_RC = 256;  // rows per pass (bounds the LDS rings of very narrow images)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Px16 {
    u32x4 q0, q1, q2;  // 16 BGR pixels = 48 bytes
};

// 16 pixels -> 16 in-range bits (bit k = pixel k)
// REP (interval variants): copies of every table row.  32: one per lane of a 32-lane group, conflict-free, 64 KiB -- two
// 1024-thread workgroups per CU.  8 / 16 (round 5): lanes l and l + REP share a copy (rows r and r' of a copy fall on the same
// bank when r = r' mod 32 / REP: about two-way conflicts on the two table reads per pixel), 16 / 32 KiB -- eight 256-thread or
// four 512-thread workgroups per CU: four times / twice as many, shorter passes per workgroup, i.e. less pipeline fill and drain
// in a launch that gives every workgroup only ten passes (BASELINE config 2).
template <int VAR, int REP = 32>
__device__ __forceinline__ uint32_t inrange16(const Px16& in, const uint32_t* __restrict__ hue,
                                              const uint32_t* __restrict__ ls, int hue_shift, const Bounds& B)
{
    constexpr bool AMB = VAR == 4;
    constexpr bool SINGLE = VAR < 3;
    constexpr bool IV = VAR >= 6;  // interval tables, sector VAR - 6
    const uint32_t d[12] = {in.q0.x, in.q0.y, in.q0.z, in.q0.w, in.q1.x, in.q1.y, in.q1.z, in.q1.w,
                            in.q2.x, in.q2.y, in.q2.z, in.q2.w};
    if (VAR == 5)  // timing-only build: consume every loaded dword, no pixel math
        return (d[0] ^ d[1] ^ d[2] ^ d[3] ^ d[4] ^ d[5] ^ d[6] ^ d[7] ^ d[8] ^ d[9] ^ d[10] ^ d[11]) & 0xffffu;
    uint32_t bits = 0, amb = 0;
    // Interval variant: one 64 KiB-aligned block of 256 rows x 256 B; bytes 0..127 of row r hold hI row r
    // once per lane of a 32-lane group, bytes 128..255 lsI row r likewise, so every lane reads its own
    // bank (no LDS conflicts).  A table address is the byte vector {lane * 4 (+128), row, base >> 16, 0}:
    // one v_perm / one SDWA subtract into byte 1 builds it.
    typedef const __attribute__((address_space(3))) uint32_t* lds_u32;
    const uint32_t ls_lane = (uint32_t)(uintptr_t)ls + (threadIdx.x & (uint32_t)(REP - 1)) * 4u;
    const uint32_t hue_lane = (uint32_t)(uintptr_t)hue + (threadIdx.x & (uint32_t)(REP - 1)) * 4u;
    constexpr int ROWSH = REP == 32 ? 8 : (REP == 16 ? 6 : 5);   // log2(bytes per table row): REP < 32 keeps hI and lsI in arrays of their own
    uint32_t haddr[4];  // four hue addresses in flight
#pragma unroll
    for (int q = 0; q < 4; ++q) haddr[q] = hue_lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = (3 * k) >> 2, sh = ((3 * k) & 3) * 8;
        const uint32_t px = sh <= 8 ? (d[j] >> sh) : __builtin_amdgcn_alignbit(d[j + 1 < 12 ? j + 1 : 11], d[j], sh);
        const int b = px & 255, g = (px >> 8) & 255, r = (px >> 16) & 255;
        if (IV) {
            // Bytes are taken straight from the loaded dwords (SDWA byte selects), no per-pixel extraction.
            constexpr int SEC = VAR - 6;
            const int iP = 3 * k + (SEC == 0 ? 2 : (SEC == 1 ? 1 : 0));
            const int iQ = 3 * k + (SEC == 0 ? 1 : (SEC == 1 ? 0 : 2));
            const int iS = 3 * k + (SEC == 0 ? 0 : (SEC == 1 ? 2 : 1));
            const uint32_t Q = (d[iQ >> 2] >> ((iQ & 3) * 8)) & 255u;
            const uint32_t S = (d[iS >> 2] >> ((iS & 3) * 8)) & 255u;
            const uint32_t mn = min(Q, S);
            uint32_t lse, hie;
            if constexpr (REP != 32) {
                const uint32_t P = (d[iP >> 2] >> ((iP & 3) * 8)) & 255u;
                lse = *(lds_u32)(uintptr_t)(ls_lane + (P << ROWSH));
                hie = *(lds_u32)(uintptr_t)(hue_lane + (((P - mn) & 255u) << ROWSH));   // negative differences alias harmlessly (below)
            } else {
            // lsI row P: address bytes {lane*4, P, base, 0}
            const uint32_t lso = __builtin_amdgcn_perm(d[iP >> 2], ls_lane, 0x0c020000u | ((4u + (iP & 3)) << 8));
            lse = *(lds_u32)(uintptr_t)lso;
            // hI row (P - mn) & 255, written into byte 1 of the address register.  Negative differences
            // alias onto real rows, which is harmless: then mn > P and no lsI row P holds a minimum above P.
            switch (iP & 3) {
                case 0: asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(haddr[k & 3]) : "v"(d[iP >> 2]), "v"(mn)); break;
                case 1: asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(haddr[k & 3]) : "v"(d[iP >> 2]), "v"(mn)); break;
                case 2: asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:DWORD" : "+v"(haddr[k & 3]) : "v"(d[iP >> 2]), "v"(mn)); break;
                default: asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(haddr[k & 3]) : "v"(d[iP >> 2]), "v"(mn)); break;
            }
            hie = *(lds_u32)(uintptr_t)haddr[k & 3];
            }
            // compares straight into lane masks (v_cmp -> SGPR pair), combined on the scalar unit
            const uint64_t m_ls = __builtin_amdgcn_uicmp((uint32_t)((int)mn - (int)(int16_t)(lse & 0xffffu)), lse >> 16, 36 /* ult */);
            const uint64_t m_h = __builtin_amdgcn_uicmp((uint32_t)((int)Q - ((int)S + (int)(int16_t)(hie & 0xffffu))), hie >> 16, 36);
            // bits = 2 * bits + in (one add-with-carry on the lane mask): pixel 0 ends up at bit 15
            const uint64_t m = m_ls & m_h;
            uint64_t cout;
            asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(bits), "=s"(cout) : "s"(m));
        } else if (SINGLE) {
            // Sector of maximum P: num = Q - S, vmin = min(Q, S).  A pixel belongs to the sector iff
            // |num| <= diff (ties resolved r > g > b by which entries the table builder ever sets), so
            // pixels of other sectors index rows/columns of the [diff + 255][num + 256] table that are
            // all-zero: no compare or select is needed here.
            const int P = VAR == 0 ? r : (VAR == 1 ? g : b);
            const int Q = VAR == 0 ? g : (VAR == 1 ? b : r);
            const int S = VAR == 0 ? b : (VAR == 1 ? r : g);
            const int mn = min(Q, S);
            const int diff = P - mn, num = Q - S;
            const uint32_t li = (uint32_t)(P << 8 | mn);
            const uint32_t lv = ls[li >> 5] >> (li & 31u);
            const int e = (diff << 9) + num + (255 * 512 + 256);  // >= 0; low 5 bits = (num + 256) & 31
            const uint32_t hv = hue[e >> 5] >> ((uint32_t)e & 31u);
            bits = __builtin_amdgcn_alignbit(lv & hv, bits, 1);  // bit 0 enters at bit 31
        } else {
            const int vmax = max(max(b, g), r), vmin = min(min(b, g), r);
            const uint32_t li = (uint32_t)(vmax << 8 | vmin);
            const uint32_t lv = ls[li >> 5] >> (li & 31u);
            const uint32_t e = hue_entry(b, g, r, vmax, vmin);
            if (AMB) {
                const uint32_t hv = hue[e >> 4] >> ((e & 15u) * 2u);
                bits = __builtin_amdgcn_alignbit(lv & hv & ~(hv >> 1), bits, 1);
                amb |= (lv & hv & (hv >> 1) & 1u) << k;
            } else {
                const uint32_t hv = hue[e >> 5] >> (e & 31u);
                bits = __builtin_amdgcn_alignbit(lv & hv, bits, 1);
            }
        }
        // keep at most four pixels' lookups in flight (register pressure)
        if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (IV) bits = __builtin_bitreverse32(bits);  // pixel k from bit 15 - k to bit 16 + k
    bits >>= 16;  // pixel k sits at bit 16 + k
    if (AMB) {
        while (amb) {  // rounding ties that depend on the triple: exact float path
            const int k = __builtin_ctz(amb);
            amb &= amb - 1;
            const uint32_t w0 = d[(3 * k) >> 2], w1 = d[((3 * k) >> 2) + 1 < 12 ? ((3 * k) >> 2) + 1 : 11];
            const uint32_t px = (uint32_t)((((uint64_t)w1 << 32) | w0) >> (((3 * k) & 3) * 8));
            int Hh, Ll, Ss;
            hls_pixel(px & 255, (px >> 8) & 255, (px >> 16) & 255, false, hue_shift, Hh, Ll, Ss);
            bits |= (Hh >= B.lo[0] && Hh <= B.hi[0]) ? (1u << k) : 0u;
        }
    }
    return bits;
}

// Persistent workgroups; each takes segments (frame, row range) and streams them top to
// bottom in passes of RC rows:
//  (1) 16 px per thread -> in-range bits into a ring of bit rows (48 B loaded, coalesced per row),
//  (2) per 32-px word: 3x3 dilation and the horizontal half of the erosion -> ring of "he" rows,
//  (3) per 16-px strip: vertical AND of three he rows, expand to bytes, one 16-byte store.
// PREFETCH: the next pass's 48 bytes per thread are requested before the current pass is processed
// (ping-pong register sets) and stay in flight across its two barriers.
// Requires W % 16 == 0 (no scalar-tail pixels, 16-byte aligned rows).
// VAR 0 / 1 / 2: only the hue sector whose maximum is r / g / b can be in range (decided
// from the table at context creation) -> no 3-way select, one 32 KiB table.
// VAR 3: any sectors, no ties.  VAR 4: any sectors, ties re-evaluated exactly.
// VAR 5: timing-only (memory traffic and barriers, no pixel math; output is garbage).
#ifdef MELF_FUSED_STAMP
// Diagnostic build only (make stamp; tools/fused_clock.py): per workgroup, the 100 MHz real-time clock at its start, when its
// tables are in LDS, when its first pass has been stored and at its end; XCC id for the placement.
__device__ uint64_t g_fused_stamps[8 * 4096];
extern "C" __attribute__((visibility("default"))) int melf_debug_fused_stamps(uint64_t* out, int nwg)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_stamps), sizeof(uint64_t) * 8 * (size_t)(nwg < 4096 ? nwg : 4096)) == hipSuccess ? 0 : -1;
}
#define FSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_fused_stamps[8 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FSTAMP(k) do { } while (0)
#endif

template <int VAR, int THREADS, int PD /* passes prefetched ahead in registers, 0 = none */, int WPS /* waves per SIMD the register budget must allow */,
          int REP /* copies of an interval-table row (inrange16) */, int MODE = 0 /* PD == 1 only.  1: segments from the launch's work queue instead of the static split (experiment, no gain);
                          2: static split, EARLY refill: the rows of pass p + 2 are requested as soon as pass p's in-range test has
                          consumed its register set (into that set), not at the start of pass p + 1 -- 1.7 passes of lead for the
                          loads instead of 1.0 with the same two register sets, two passes requested before anything is computed,
                          and nothing fetched beyond a segment's end */>
__global__ __launch_bounds__(THREADS, WPS) void k_fused_mask_lut(
    const uint8_t* __restrict__ frames, int n, int H, int W, int hue_shift, Bounds B,
    const uint32_t* __restrict__ g_tables, uint8_t* __restrict__ masks, int segs_per_frame, int seg_rows, int NB, int plain_store, int rc_dma,
    uint32_t* __restrict__ wq /* work queue of this launch {next segment, workgroups done}, or NULL: static split */,
    int big_segs, int big_rows /* DYN: a frame's first big_segs x big_rows rows are "big" segments (ids 0 .. n big_segs - 1: every workgroup's
                                  first one), the rest segs_per_frame small ones of seg_rows rows, handed out behind them */)
{
    constexpr bool DYN = MODE == 1, EARLY = MODE == 2;
    static_assert(MODE == 0 || PD == 1, "the queue loop and the early refill are written for two register sets");
    constexpr bool PREFETCH = PD > 0;
    // PD < 0 (round 4, experiment MELF_FUSED_CONFIG=6): the pixel rows of a pass arrive by LDS-DMA with the non-temporal policy
    // (global_load_lds_dwordx4 ... nt: 1 KiB lane-contiguous pieces straight into a staging buffer in LDS, two buffers: the
    // next pass's rows land while this pass is processed), and every thread reads its 48 bytes back with three ds_read_b128.
    // A bare stream of this traffic mix runs 5 % faster that way (tools/ubench/stream_lds.hip).  LDS: 64 KiB of tables + two
    // staging buffers + the rings = one 1024-thread workgroup per CU, rows per pass cut to what fits (rc_dma).
    constexpr bool DMA = PD < 0;
    constexpr bool AMB = VAR == 4;
    constexpr bool IV = VAR >= 6;
    constexpr bool SINGLE = VAR < 3 || VAR == 5;
    static_assert(REP == 32 || (IV && (REP == 8 || REP == 16)), "row copies: 32, or 8 / 16 for the interval tables");
    constexpr int HDW = IV ? 256 * 2 * REP /* both interval tables: interleaved by row (REP 32), one behind the other (REP < 32) */
                           : (AMB ? HUE2_DWORDS : (SINGLE ? HUES_DWORDS : HUE1_DWORDS));
    constexpr int LSDW = IV ? 4 /* lives inside hue[] */ : LS_DWORDS;
    // tables in static LDS (their addresses fold into the ds_read offset field), rings in dynamic LDS
    __shared__ __attribute__((aligned(IV && REP == 32 ? 65536 : 16))) uint32_t hue[HDW];
    __shared__ __attribute__((aligned(16))) uint32_t ls_own[LSDW];
    uint32_t* const ls = IV ? hue + (REP == 32 ? 32 : 256 * REP) : ls_own;
    __shared__ uint32_t expand4[16];  // 4 mask bits -> 4 mask bytes
    extern __shared__ uint32_t ring[];
    const int wpr = (W + 31) >> 5;
    uint32_t* raw = ring;
    uint32_t* he = raw + NB * wpr;
    const int tid = threadIdx.x;
    FSTAMP(0);
#ifdef MELF_FUSED_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 4096) { uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); g_fused_stamps[8 * blockIdx.x + 4] = xcc & 15u; uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_fused_stamps[8 * blockIdx.x + 5] = hw; }
    int fstamp_passes = 0;
#endif
    // Table fill, run once per workgroup AFTER the first frame loads have been issued (see below).  Interval tables: the two
    // table words of a thread's row are requested here, BEFORE the first frame loads (round 5): loads return in order, so
    // behind 48 bytes per thread of pixel rows the tables reached LDS only when the whole chip's first pass had landed.
    static_assert(!IV || THREADS >= 256, "one interval-table row per thread");
    uint32_t pre_vh = 0, pre_vl = 0;
    if (IV && DYN && tid < 256) {   // (the static launch has no two registers to spare at 8 waves per SIMD: it loads them in fill_tables)
        pre_vh = g_tables[OFF_HI + (VAR - 6) * HI_ROWS + 256 + tid];   // hI rows diff = 0..255 (table rows diff + 256)
        pre_vl = g_tables[OFF_LSI + (VAR - 6) * LSI_ROWS + tid];       // lsI row P
    }
    auto fill_tables = [&]() {
        if (IV) {
            // each (lo, count) row replicated so that lane L of a 32-lane group always reads bank L: 16-byte LDS stores
            if (tid < 256) {
                const int r = tid;
                const uint32_t vh = DYN ? pre_vh : g_tables[OFF_HI + (VAR - 6) * HI_ROWS + 256 + r];
                const uint32_t vl = DYN ? pre_vl : g_tables[OFF_LSI + (VAR - 6) * LSI_ROWS + r];
                const u32x4 h4 = {vh, vh, vh, vh}, l4 = {vl, vl, vl, vl};
                if constexpr (REP == 32) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        *(u32x4*)(hue + r * 64 + q * 4) = h4;
                        *(u32x4*)(hue + r * 64 + 32 + q * 4) = l4;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < REP / 4; ++q) {
                        *(u32x4*)(hue + r * REP + q * 4) = h4;
                        *(u32x4*)(hue + 256 * REP + r * REP + q * 4) = l4;
                    }
                }
            }
        } else {
            for (int i = tid; i < HDW; i += THREADS)
                hue[i] = g_tables[(AMB ? 0 : (SINGLE ? OFF_HUES + (VAR % 3) * HUES_DWORDS : OFF_HUE1)) + i];
            for (int i = tid; i < LSDW; i += THREADS) ls[i] = g_tables[OFF_LS + i];
        }
        for (int i = tid; i < 2 * NB * wpr; i += THREADS) raw[i] = 0;  // unused half-words must read as 0
        if (tid < 16) expand4[tid] = ((uint32_t)tid * 0x00204081u & 0x01010101u) * 255u;
    };
    bool tables_ready = false;

    const int G16 = W >> 4;
    const int RC = DMA ? rc_dma : min(THREADS / G16, FUSED_MAX_RC);  // rows per pass
    // DMA staging: two buffers of whole 1 KiB pieces behind the rings
    const int pass_bytes = RC * W * 3, npieces = (pass_bytes + 1023) >> 10;
    uint8_t* const stage_base = (uint8_t*)ring + (((size_t)2 * NB * wpr * 4 + 1023) & ~(size_t)1023);
    const long frame_bytes = (long)H * W * 3;
    auto lds_barrier = [&]() {   // a barrier that does NOT drain the vector-memory counter (an LDS-DMA in flight stays in flight)
        if (DMA) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
    };
    const int trow = tid / G16, tg = tid - trow * G16;
    const bool active = trow < RC;
    const int tgc = active ? tg : 0;
    const int drow = tid / wpr, dk = tid - drow * wpr;  // word work item of step 2
    const uint32_t lastmask = (W & 31) ? ((1u << (W & 31)) - 1u) : 0xffffffffu;
    const int nbm = NB - 1;

    // the segment being processed: (frame, rows [r0, r1)); the lambdas below see it by reference
    __shared__ int q_ids[2];         // work queue: the workgroup's next two segments
    bool dq_first = false;           // this pass is the first of a queue-fed segment
    int dq_slot = 0;
    int r0 = 0, r1 = 0;
    const uint8_t* frame = frames;
    uint8_t* out = masks;
    auto seg_decode = [&](int seg, int& f, int& q0, int& q1) {
        if (DYN && seg < n * big_segs) {
            f = seg / big_segs;
            q0 = (seg - f * big_segs) * big_rows;
            q1 = q0 + big_rows;
        } else {
            if (DYN) seg -= n * big_segs;
            f = seg / segs_per_frame;
            q0 = (DYN ? big_segs * big_rows : 0) + (seg - f * segs_per_frame) * seg_rows;
            q1 = min(H, q0 + seg_rows);
        }
    };
    auto set_segment = [&](int seg) {
        int f;
        seg_decode(seg, f, r0, r1);
        frame = frames + (size_t)f * H * W * 3;
        out = masks + (size_t)f * H * W;
    };
    {
        // Unconditional, address-clamped loads: keeping them out of divergent control flow lets the
        // compiler wait with an exact vmcnt(N) instead of vmcnt(0).
        auto load_from = [&](const uint8_t* fr, int a, int ylast /* last row worth fetching */, Px16& dst) {
            int y = a + trow;
            y = y < 0 ? 0 : (y > ylast ? ylast : y);
            // plain (cached) loads: each 128-byte line is touched by three dwordx4 instructions of the
            // wave (48-byte lane stride); non-temporal loads refetch it and measured 25 % slower
            // 32-bit offsets from a uniform base (full-rate 24-bit multiplies; a frame is < 4 GiB)
            const u32x4* p = (const u32x4*)(fr + (__umul24((uint32_t)y, (uint32_t)W) + 16u * (uint32_t)tgc) * 3u);
            dst.q0 = p[0]; dst.q1 = p[1]; dst.q2 = p[2];
        };
        auto load = [&](int a, Px16& dst) { load_from(frame, a, H - 1, dst); };
        // DMA mode: request the rows [a, a + RC) of the frame into staging buffer `slot` (wave w takes pieces w, w + 16, ...)
        auto issue = [&](int a, int slot) {
            const long row0 = (long)a * W * 3;
            for (int p = tid >> 6; p < npieces; p += THREADS / 64) {
                long o = row0 + (long)p * 1024 + (long)(tid & 63) * 16;
                o = o < 0 ? 0 : (o > frame_bytes - 16 ? frame_bytes - 16 : o);   // rows outside the frame: anything valid (their bits are zeroed)
                __builtin_amdgcn_global_load_lds((const u32x4*)(frame + o),
                                                 (__attribute__((address_space(3))) void*)(stage_base + (size_t)slot * npieces * 1024 + (size_t)p * 1024), 16, 0, 2 /* nt */);
            }
        };
        int dma_slot = 0;
        auto pass = [&](int a, Px16& cur, int refill_a = INT_MIN /* EARLY: first row of the pass whose rows go into `cur` next */) {
            // ---- (1) in-range bits of input rows [a, a+RC) ----
            if (DMA) {
                const int y = a + trow;
                uint32_t bits = 0;
                if (active) {
                    const u32x4* sp = (const u32x4*)(stage_base + (size_t)dma_slot * npieces * 1024 + (size_t)tid * 48);
                    Px16 px;
                    px.q0 = sp[0]; px.q1 = sp[1]; px.q2 = sp[2];
                    bits = inrange16<VAR, REP>(px, hue, ls, hue_shift, B);
                }
                bits = (y >= 0 && y < H) ? bits : 0u;
                if (active && y < r1 + 2) ((uint16_t*)raw)[__umul24((y + 4 * NB) & nbm, wpr * 2) + tg] = (uint16_t)bits;
            } else if (PREFETCH) {
                const int y = a + trow;
                uint32_t bits = inrange16<VAR, REP>(cur, hue, ls, hue_shift, B);
                if (EARLY && refill_a != INT_MIN) load_from(frame, refill_a, min(r1 + 1, H - 1), cur);   // (uniform) the set is free: pass p + 2's rows
                bits = (y >= 0 && y < H) ? bits : 0u;
                if (active && y < r1 + 2) ((uint16_t*)raw)[__umul24((y + 4 * NB) & nbm, wpr * 2) + tg] = (uint16_t)bits;
            } else {
                // load next to its use: shortest live ranges (this path runs at 64..80 VGPRs)
                const int y = a + trow;
                if (active && y < r1 + 2) {
                    uint32_t bits = 0;
                    if (y >= 0 && y < H) {
                        const u32x4* p = (const u32x4*)(frame + ((size_t)y * W + 16 * tg) * 3);
                        Px16 px;
                        px.q0 = p[0]; px.q1 = p[1]; px.q2 = p[2];
                        bits = inrange16<VAR, REP>(px, hue, ls, hue_shift, B);
                    }
                    ((uint16_t*)raw)[((y + 4 * NB) & nbm) * wpr * 2 + tg] = (uint16_t)bits;
                }
            }
            // work queue (round 5, see below): in a segment's first pass thread 0 requests the segment after next HERE -- behind
            // the in-range test, where the register pressure peaks -- and hands it on behind the wait in front of the store,
            // two barriers later: by then the atomic has long returned, and its register is live only over the two light steps
            uint32_t dq_pend = 0;
            if (DYN && dq_first && tid == 0) dq_pend = __hip_atomic_fetch_add(wq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_barrier();
            // ---- (2) rows [a-1, a+RC-2]: 3x3 dilation, then the horizontal part of the erosion.
            //      Pixels outside the image are neutral (never win): 0 for the dilation, 1 for the erosion.
            if (drow < RC) {
                const int y = a - 1 + drow;
                if (y < r1 + 1 && y >= r0 - 1) {
                    uint32_t v = 0xffffffffu;
                    if (y >= 0 && y < H) {
                        uint32_t C = 0, L = 0, R = 0;  // vertical OR of the three raw rows: this word and its neighbours
#pragma unroll
                        for (int dy = -1; dy <= 1; ++dy) {
                            const int yy = y + dy;
                            if (yy < 0 || yy >= H) continue;
                            const uint32_t* rr = raw + __umul24((yy + 4 * NB) & nbm, wpr);
                            C |= rr[dk];
                            L |= dk > 0 ? rr[dk - 1] : 0u;
                            R |= dk + 1 < wpr ? rr[dk + 1] : 0u;
                        }
                        uint32_t dil = C | (C << 1) | (L >> 31) | (C >> 1) | (R << 31);
                        // dilated bit just left / right of this word (bit 31 of word k-1, bit 0 of word k+1)
                        uint32_t dl = ((L >> 31) | (L >> 30) | C) & 1u;
                        uint32_t dr = (R | (R >> 1) | (C >> 31)) & 1u;
                        if (dk == 0) dl = 1u;
                        if (dk == wpr - 1) { dil |= ~lastmask; dr = 1u; }
                        v = dil & ((dil << 1) | dl) & ((dil >> 1) | (dr << 31));
                    }
                    he[__umul24((y + 4 * NB) & nbm, wpr) + dk] = v;
                }
            }
            lds_barrier();
            // ---- (3) rows [a-2, a+RC-3] of the segment: vertical AND, expand, store ----
            // gfx9 counts loads and stores in the same vmcnt and the compiler treats them as completing
            // out of order, so a wait for the prefetched pixels issued AFTER this pass's store would also
            // wait for the store's write-ack (a full memory round trip per pass).  Waiting here, just
            // before the store is issued, costs nothing: the only store in flight is one pass old.
            if (EARLY) {
                // the next pass's rows (requested one pass ago) must be in; the three loads just issued for the pass after it may stay in
                // flight.  Loads return in order, so "at most three outstanding" cannot leave one of the older three out, whenever the
                // previous pass's store is acknowledged.
                if (refill_a != INT_MIN) __builtin_amdgcn_s_waitcnt(0x0F70 | 3);
                else __builtin_amdgcn_s_waitcnt(0x0F70);
            } else if (PD > 0) __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * (PD - 1)));  // vmcnt(3*(PD-1)), others untouched
            if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next pass's rows have landed (this wave's pieces); the store below is younger
            if (DYN && dq_first && tid == 0) q_ids[dq_slot] = (int)(dq_pend + gridDim.x);   // every thread read this slot at least two barriers ago
            {
                const int y = a - 2 + trow;
                if (active && y >= r0 && y < r1) {
                    const uint16_t* h16p = (const uint16_t*)he;
                    const uint32_t h16 = (uint32_t)h16p[__umul24((y - 1 + 4 * NB) & nbm, wpr * 2) + tg] &
                                         (uint32_t)h16p[__umul24((y + 4 * NB) & nbm, wpr * 2) + tg] &
                                         (uint32_t)h16p[__umul24((y + 1 + 4 * NB) & nbm, wpr * 2) + tg];
                    u32x4 o;  // 16 bits -> 16 bytes through the 16-entry LDS table (broadcast reads)
                    o.x = *(const uint32_t*)((const char*)expand4 + ((h16 << 2) & 0x3cu));
                    o.y = *(const uint32_t*)((const char*)expand4 + ((h16 >> 2) & 0x3cu));
                    o.z = *(const uint32_t*)((const char*)expand4 + ((h16 >> 6) & 0x3cu));
                    o.w = *(const uint32_t*)((const char*)expand4 + ((h16 >> 10) & 0x3cu));
                    u32x4* dstp = (u32x4*)(out + (__umul24((uint32_t)y, (uint32_t)W) + 16u * (uint32_t)tg));
                    if (plain_store & 1) *dstp = o;
                    else __builtin_nontemporal_store(o, dstp);
                }
            }
            // no barrier needed here: the rings (NB >= 2*RC + 4 rows) keep this pass's rows apart
            // from the rows the next pass writes.  DMA mode: every wave's pieces of the next pass must have landed before
            // any wave reads the staging buffer
            if (DMA) { asm volatile("s_barrier" ::: "memory"); dma_slot ^= 1; }
        };
        // ---- round 5 (experiment, MELF_FUSED_DYN=N; off by default): the launch's segments come from a work queue (wq) ----
        // Under the static split the two workgroups of a CU finish 25 % apart (oldest-first arbitration; profiles/r04/
        // fused_workgroup_clock.txt) and the launch's last quarter runs with half its workgroups; a bare stream of this traffic
        // mix WITHOUT prefetch gains 9 % from a dynamic split at 1080p sizes (tools/ubench/stream_dyn.hip) -- this kernel does
        // not (see launch_lut_t): its lone workgroups keep two passes in flight.  Here: the first segment of a workgroup is its
        // block index (no round trip before the first loads), every further one comes from an atomic counter, requested a
        // whole segment before it is needed (thread 0, result handed on through LDS behind the passes' barriers), and the
        // register prefetch runs ACROSS segment boundaries: the first rows of the next segment are in flight while the last
        // pass of this one is processed (more, smaller segments under the static split lost exactly there: every segment start
        // refilled the pipeline).  PD == 1: the load cursor is one pass ahead, i.e. in this segment or the next.
        static_assert(!DYN || PD == 1, "the work-queue loop is written for one pass of register prefetch");
        if constexpr (DYN) {
            {
                const int total = n * (big_segs + segs_per_frame);
                uint32_t pend = 0;
                if (tid == 0) pend = __hip_atomic_fetch_add(wq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                set_segment(blockIdx.x);
                int a = r0 - 2, nxt = total;
                // The load cursor is exactly one pass ahead of (frame, a): the same segment's next rows, or the first rows of
                // segment `nxt` (no state of its own: scalar registers are what this launch shape is short of)
                auto load_ahead = [&](Px16& dst) {
                    const int na = a + RC;
                    // rows beyond the segment's last halo row are not fetched again (the clamp makes them repeats of that row)
                    if (na < r1 + 2) {
                        load_from(frame, na, min(r1 + 1, H - 1), dst);
                    } else if (nxt < total) {   // uniform
                        int f2, q0, q1;
                        seg_decode(nxt, f2, q0, q1);
                        load_from(frames + (size_t)f2 * H * W * 3, q0 - 2, min(q1 + 1, H - 1), dst);
                    }
                };
                Px16 pbuf[2];
                load_from(frame, a, min(r1 + 1, H - 1), pbuf[0]);
                fill_tables();
                if (tid == 0) q_ids[1] = (int)(pend + gridDim.x);
                __syncthreads();
                FSTAMP(1);
                nxt = q_ids[1];
                dq_first = true;     // segment k's first pass requests segment k + 2 into q_ids[k & 1] (pass())
                dq_slot = 0;
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the first pass's rows (see pass())
                for (;;) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        load_ahead(pbuf[q ^ 1]);
                        pass(a, pbuf[q]);
#ifdef MELF_FUSED_STAMP
                        if (fstamp_passes++ == 0) FSTAMP(2);
#endif
                        dq_first = false;
                        a += RC;
                        if (a >= r1 + 2) {   // next segment
                            lds_barrier();
                            if (nxt >= total) goto queue_empty;
                            set_segment(nxt);
                            dq_slot ^= 1;
                            nxt = q_ids[dq_slot ^ 1];
                            a = r0 - 2;
                            dq_first = true;
                        }
                    }
                }
            queue_empty:
                // the launch's last workgroup leaves the queue zeroed for the launch that gets this slot next
                if (tid == 0) {
                    const uint32_t done = __hip_atomic_fetch_add(wq + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (done == gridDim.x - 1) {
                        __hip_atomic_store(wq, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(wq + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                FSTAMP(3);
                return;
            }
        }
    if constexpr (!DYN)
    for (int seg = blockIdx.x; seg < n * segs_per_frame; seg += gridDim.x) {
        set_segment(seg);
        int a = r0 - 2;
        const int aend = r1 + 2;
        if constexpr (DMA) {
            dma_slot = 0;
            issue(a, 0);
            if (!tables_ready) {  // the first rows are in flight while LDS is filled
                fill_tables();
                tables_ready = true;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            Px16 unused = {};
            for (; a < aend; a += RC) {
                if (a + RC < aend) issue(a + RC, dma_slot ^ 1);
                pass(a, unused);
            }
        } else if constexpr (EARLY) {
            Px16 pbuf[2];
            const int ylast = min(r1 + 1, H - 1);
            const bool two = a + RC < aend;          // uniform
            load_from(frame, a, ylast, pbuf[0]);
            if (two) load_from(frame, a + RC, ylast, pbuf[1]);
            if (!tables_ready) {  // the frame loads above are already in flight while LDS is filled
                fill_tables();
                tables_ready = true;
                __syncthreads();
                FSTAMP(1);
            }
            if (two) __builtin_amdgcn_s_waitcnt(0x0F70 | 3);   // the first pass's rows
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            for (bool more = true; more;) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ra = a + 2 * RC;
                    pass(a, pbuf[q], ra < aend ? ra : INT_MIN);
#ifdef MELF_FUSED_STAMP
                    if (fstamp_passes++ == 0) FSTAMP(2);
#endif
                    a += RC;
                    if (a >= aend) { more = false; break; }
                }
            }
        } else if constexpr (PREFETCH) {
            // PD + 1 rotating register sets: the loads of the next PD passes are in flight while the
            // current pass is processed (all indices are compile-time after unrolling)
            Px16 pbuf[PD + 1];
#ifdef MELF_FUSED_STAMP
            // Experiment of the diagnostic build (MELF_FUSED_PRIO, tools/fused_clock.py): wave priorities that favour the CU's
            // second workgroup (2), or each of the two half the time by the clock both read (5..9: half-periods of 2.5 .. 41 us).
            // Outcome (profiles/r04/fused_workgroup_clock.txt): the priority decides WHICH workgroup of a CU finishes first
            // (0.56 vs 0.74 ms at 1080p, either way round) and the complementary schemes make them finish together -- at the
            // time the slower one needed before: the launch is as long as it was.  What the workgroups share is the memory
            // system's throughput for this mix, not the CU's issue slots.
            const int prio_mode = plain_store >> 8;
            const bool young = blockIdx.x >= 256;   // the second workgroup a CU received
#endif
#pragma unroll
            for (int q = 0; q < PD; ++q) load(a + q * RC, pbuf[q]);
            if (!tables_ready) {  // the frame loads above are already in flight while LDS is filled
                fill_tables();
                tables_ready = true;
                __syncthreads();
                FSTAMP(1);
            }
            // same explicit wait as before each store (see pass()): the loop body then never waits on
            // a load that is younger than a store
            __builtin_amdgcn_s_waitcnt(0x0F70 | (3 * (PD - 1)));
            for (bool more = true; more;) {
#pragma unroll
                for (int q = 0; q <= PD; ++q) {
#ifdef MELF_FUSED_STAMP
                    if (prio_mode == 2) { if (young) __builtin_amdgcn_s_setprio(1); }
                    else if (prio_mode >= 5) {
                        const bool hi = (((uint32_t)__builtin_amdgcn_s_memrealtime() >> (prio_mode + 3)) & 1u) != (young ? 1u : 0u);
                        if (hi) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                    }
#endif
                    load(a + PD * RC, pbuf[(q + PD) % (PD + 1)]);
                    pass(a, pbuf[q]);
#ifdef MELF_FUSED_STAMP
                    if (fstamp_passes++ == 0) FSTAMP(2);
#endif
                    a += RC;
                    if (a >= aend) { more = false; break; }
                }
            }
        } else {
            if (!tables_ready) {
                fill_tables();
                tables_ready = true;
                __syncthreads();
            }
            Px16 unused = {};
            for (; a < aend; a += RC) pass(a, unused);
        }
        lds_barrier();
    }
    }
    FSTAMP(3);
}

bool fused_mask_lut_ok(const void* d_frames, const void* d_masks, int H, int W)
{
    return W >= 16 && W <= 512 * 16 && (W & 15) == 0 && (((size_t)d_frames | (size_t)d_masks) & 15) == 0 && H >= 1;
}

// launch configuration (the product compiles ONE per kernel variant, see launch_lut_v; MELF_FUSED_CONFIG of the diagnostic build selects the others):
//   0: 512 threads, 3 workgroups/CU (6 waves/SIMD, <= 80 VGPRs), no prefetch   [default of the bit-table variants until round 4]
//   4: 1024 threads, 2 workgroups/CU (8 waves/SIMD, <= 64 VGPRs), register prefetch [default of the interval-table
//      variants: with the launches rotating over buffers beyond the Infinity Cache it is 3-10 % faster than 2 x 512
//      threads x 4 waves/SIMD, which was the fastest while the working set sat in the cache]
//   1: 512 threads, 2 workgroups/CU (4 waves/SIMD), register prefetch
//   2: 1024 threads, 1-2 workgroups/CU (4 waves/SIMD), register prefetch   [default of the bit-table / generic variants]
//   3: 1024 threads, 2 workgroups/CU (8 waves/SIMD, <= 64 VGPRs), no prefetch
static int g_fused_config = -1;  // -1: per-variant default
static thread_local hipEvent_t g_fused_ev_start = nullptr, g_fused_ev_stop = nullptr;
void fused_mask_timing_events(hipEvent_t start, hipEvent_t stop) { g_fused_ev_start = start; g_fused_ev_stop = stop; }

template <int V, int T, int PF, int WPS, int REP = 32>
static void launch_lut_t(const uint8_t* d_frames, int n, int H, int W, int hue_shift, const Bounds& B,
                         uint32_t* d_tables, uint8_t* d_masks, hipStream_t stream)
{
    const int G16 = W >> 4, wpr = (W + 31) >> 5;
    int RC = T / G16 < FUSED_MAX_RC ? T / G16 : FUSED_MAX_RC;
    int NB = 8;
    while (NB < 2 * RC + 4) NB <<= 1;
    size_t dma_bytes = 0;   // PF < 0: two staging buffers behind the rings; rows per pass cut to what the CU's 160 KiB hold
    if (PF < 0) {
        for (;; --RC) {
            NB = 8;
            while (NB < 2 * RC + 4) NB <<= 1;
            const size_t ringb = (((size_t)2 * NB * wpr * 4 + 1023) & ~(size_t)1023), sb = (((size_t)RC * W * 3 + 1023) >> 10) << 10;
            dma_bytes = ringb + 2 * sb - (size_t)2 * NB * wpr * 4;
            if (RC <= 1 || 65664 + ringb + 2 * sb <= 160 * 1024) break;   // 65 600 bytes of static LDS (tables, expand table)
        }
    }
    int per_cu = PF < 0 ? 1 : WPS * 256 / T;                // resident workgroups per CU
    if (REP != 32) {   // small tables: LDS (tables + this shape's rings) may admit one workgroup fewer than the wave count does
        const size_t lds_wg = (size_t)256 * 2 * REP * 4 + 128 + (size_t)(2 * NB * wpr) * sizeof(uint32_t);
        per_cu = std::min<int>(per_cu, (int)(160 * 1024 / lds_wg));
    }
    const int target = 256 * (per_cu < 1 ? 1 : per_cu);
    static const int seg_mult = diag_env("MELF_FUSED_SEGMULT") ? atoi(diag_env("MELF_FUSED_SEGMULT")) : 1;  // experiments
    static const int plain_store = diag_env("MELF_FUSED_PLAINSTORE") ? atoi(diag_env("MELF_FUSED_PLAINSTORE")) : 0;
    int segs = (target * seg_mult + n - 1) / n;
    int seg_rows = (H + segs - 1) / segs;
    if (seg_rows < 32) seg_rows = H < 32 ? H : 32;
    segs = (H + seg_rows - 1) / seg_rows;
    // Work queue (the default launch shape: PD == 1, 1024 threads, full tables).  Round 6: the launch's time depends on WHERE its
    // buffers lie -- +-6 % between allocations of one process, stable within one (tools/fused_alloc_probe.py), and the bare stream
    // shows the same: each workgroup walking its own long run ("comb") is what the placement hurts, all workgroups taking small
    // consecutive pieces from one counter (a compact front moving through memory) is not.  So a launch whose workgroups would each
    // stream a long run hands out SMALL segments instead: two or three passes including the 4 halo rows (20 rows of a 1080p frame: the halo
    // rows are hits in the memory-side cache, their neighbours are being read at the same time), at least 64 KB of pixels.  On the
    // same buffers (tools/fused_queue_ab.py, profiles/r06/fused_queue_ab_*.txt): 1080p B = 512 0.749-0.783 ms on every placement
    // against 0.738-0.833 for the static split (mean -3.3 %, worst case -6 %, best case +2 %); coarser segments lose (the round-5
    // A/Bs used 135-540 rows and compared separate processes, i.e. placements).  Short runs (B = 256 640 x 480: 240 rows per
    // workgroup) measure the same either way and keep the static split.
    // Diagnostic build: MELF_FUSED_DYN = 0 forces the static split, N > 0 aims at N segments per workgroup (MELF_FUSED_BIG = percent
    // of a frame's rows dealt as one big static first segment per workgroup, MELF_FUSED_GRID = workgroups).
    constexpr bool queue_shape = PF == 1 && REP == 32 && T == 1024;
    int dyn = queue_shape ? -1 : 0;   // -1: the rule below
    int grid_cap = target, big_pct = 0;
#ifdef MELF_DIAG
    if (queue_shape && diag_env("MELF_FUSED_DYN")) dyn = atoi(diag_env("MELF_FUSED_DYN"));
    if (diag_env("MELF_FUSED_GRID")) grid_cap = std::max(1, atoi(diag_env("MELF_FUSED_GRID")));
    if (diag_env("MELF_FUSED_BIG")) big_pct = std::min(95, std::max(0, atoi(diag_env("MELF_FUSED_BIG"))));
#endif
    const int wgs = std::min(target, grid_cap);
    uint32_t* wq = nullptr;
    int big_segs = 0, big_rows = 0;
    if (dyn != 0) {
        int Hs = H;   // rows dealt as small segments
        if (dyn > 0 && big_pct > 0 && n > 0) {
            big_segs = (wgs + n - 1) / n;                                      // every workgroup's first segment is a big one
            int Pb = (int)(((double)H * big_pct / 100.0 / big_segs + 4.0) / RC + 0.5);
            big_rows = Pb * RC - 4;
            if (big_rows < RC || big_segs * big_rows > H - RC) { big_segs = 0; big_rows = 0; }
            else Hs = H - big_segs * big_rows;
        }
        int P;
        bool use = true;
        if (dyn > 0) {
            const double rows_aimed = (double)n * Hs / ((double)wgs * dyn);
            P = (int)((rows_aimed + 4.0) / RC + 0.5);
        } else {
            // the fewest passes that make a segment of >= 64 KB of pixels with the 4 halo rows at most a quarter of its own rows
            // (1080p: 3 passes = 20 rows, 115 KB; 640 x 480: 2 passes = 46 rows, 88 KB -- measured best of P = 2 .. 5 for both);
            // where a workgroup gets six or more such segments ...
            P = std::max(2, (int)((64.0 * 1024 / ((double)W * 3) + 4.0) / RC + 0.999));
            while (P * RC - 4 < 16 && P < 64) ++P;
            const int own = P * RC - 4;
            // ... or where the static split comes out badly: its segments do not go into the workgroups evenly (57 frames of 1080p:
            // 513 segments for 512 workgroups, a second round for one of them -- the queue is 20 % faster there), measured by the rows
            // the busiest workgroup gets either way (the queue's: its share with the halo rows, plus one segment of imbalance)
            const long st_total = (long)n * segs, st_grid = std::min<long>(st_total, wgs);
            const double st_rows = (double)((st_total + st_grid - 1) / st_grid) * seg_rows;
            const double q_rows = (double)n * H * (1.0 + 4.0 / own) / wgs + own;
            use = (double)n * H / wgs >= 6.0 * own || q_rows < 0.9 * st_rows;
        }
        if (P * RC - 4 < RC) P = (2 * RC + 3) / RC;                           // at least RC rows of its own
        const int dr0 = std::min(P * RC - 4, Hs), ds = (Hs + dr0 - 1) / dr0;
        const int dr = (Hs + ds - 1) / ds;                                     // evenly: no short last segment
        const int qslot = (use && (long)n * (ds + big_segs) > wgs) ? fused_queue_slot(d_tables, stream) : -1;
        if (qslot >= 0) {
            seg_rows = dr;
            segs = ds;
            wq = d_tables + FUSED_TABLE_DWORDS + 16 * qslot;
        } else {
            big_segs = big_rows = 0;
        }
    }
    const long total = (long)n * (segs + big_segs);
    const int grid = (int)std::min<long>(total, wgs);
    const size_t shmem = (size_t)(2 * NB * wpr) * sizeof(uint32_t) + dma_bytes;
    static bool attr_set[64] = {false};  // per device (several contexts on several GPUs may live in one process)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)k_fused_mask_lut<V, T, PF, WPS, REP>, hipFuncAttributeMaxDynamicSharedMemorySize, PF < 0 ? 160 * 1024 - 65664 : 28 * 1024);
        if (diag_env("MELF_FUSED_TRACE")) {
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_fused_mask_lut<V, T, PF, WPS, REP>, T, shmem);
            fprintf(stderr, "[melf fused] V=%d T=%d PD=%d WPS=%d grid=%d shmem=%zu resident blocks/CU=%d\n", V, T, PF, WPS, grid, shmem, nb);
        }
        attr_set[dev] = true;
    }
    // timing events (optional, set by the caller through fused_mask_timing_events): the dispatch's own start / stop stamps,
    // no event-record packets in the queue around the kernel
    const int ps = plain_store | ((diag_env("MELF_FUSED_PRIO") ? atoi(diag_env("MELF_FUSED_PRIO")) : 0) << 8);
    if constexpr (queue_shape) {
        if (wq) {
            static bool dyn_attr_set[64] = {false};
            if (dev >= 0 && dev < 64 && !dyn_attr_set[dev]) {
                (void)hipFuncSetAttribute((const void*)k_fused_mask_lut<V, T, PF, WPS, REP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 28 * 1024);
                dyn_attr_set[dev] = true;
            }
            hipExtLaunchKernelGGL((k_fused_mask_lut<V, T, PF, WPS, REP, 1>), dim3(grid), dim3(T), shmem, stream, g_fused_ev_start, g_fused_ev_stop, 0, d_frames,
                                  n, H, W, hue_shift, B, d_tables, d_masks, segs, seg_rows, NB, ps, RC, wq, big_segs, big_rows);
            g_fused_ev_start = g_fused_ev_stop = nullptr;
            return;
        }
#ifdef MELF_DIAG
        // early refill (round 5 experiment, MELF_FUSED_EARLY=1): measured no different from the refill at the start of the next
        // pass (config 2 0.0668 / 0.0681 against 0.0664 ms, config 5 0.782 / 0.824 against 0.781 / 0.820: profiles/r05/
        // fused_early_refill_ab.txt) -- the launch is not short of requests in flight.  Diagnostic build only.
        if (diag_env("MELF_FUSED_EARLY") && atoi(diag_env("MELF_FUSED_EARLY")) == 1) {
            static bool early_attr_set[64] = {false};
            if (dev >= 0 && dev < 64 && !early_attr_set[dev]) {
                (void)hipFuncSetAttribute((const void*)k_fused_mask_lut<V, T, PF, WPS, REP, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 28 * 1024);
                early_attr_set[dev] = true;
            }
            hipExtLaunchKernelGGL((k_fused_mask_lut<V, T, PF, WPS, REP, 2>), dim3(grid), dim3(T), shmem, stream, g_fused_ev_start, g_fused_ev_stop, 0, d_frames,
                                  n, H, W, hue_shift, B, d_tables, d_masks, segs, seg_rows, NB, ps, RC, (uint32_t*)nullptr, 0, 0);
            g_fused_ev_start = g_fused_ev_stop = nullptr;
            return;
        }
#endif
    }
    hipExtLaunchKernelGGL((k_fused_mask_lut<V, T, PF, WPS, REP>), dim3(grid), dim3(T), shmem, stream, g_fused_ev_start, g_fused_ev_stop, 0, d_frames,
                          n, H, W, hue_shift, B, d_tables, d_masks, segs, seg_rows, NB, ps, RC, (uint32_t*)nullptr, 0, 0);
    g_fused_ev_start = g_fused_ev_stop = nullptr;
}

template <int V>
static void launch_lut_v(const uint8_t* d_frames, int n, int H, int W, int hue_shift, const Bounds& B,
                         uint32_t* d_tables, uint8_t* d_masks, hipStream_t stream)
{
#ifdef MELF_DIAG   // the launch shapes of rounds 2-5 (MELF_FUSED_CONFIG), for A/B runs with the diagnostic build
    if constexpr (V >= 5) {  // interval tables (64 KiB per workgroup); 5: timing-only twin of the same launch shapes
        switch (g_fused_config) {
            case 1: launch_lut_t<V, 512, 2, 4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return;
            case 2: launch_lut_t<V, 1024, 1, 4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return;
            case 3: launch_lut_t<V, 1024, 2, 4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return;
            case 5: launch_lut_t<V, 1024, 0, 8>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return;
            case 6: launch_lut_t<V, 1024, -1, 4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return;   // LDS-DMA staging
            case 0: launch_lut_t<V, 512, 1, 4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return;
            // round 5: smaller tables (8 / 16 copies of a row), more and smaller workgroups per CU
            case 8: if constexpr (V >= 6) { launch_lut_t<V, 256, 1, 8, 8>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return; } break;
            case 9: if constexpr (V >= 6) { launch_lut_t<V, 512, 1, 8, 16>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return; } break;
            case 10: if constexpr (V >= 6) { launch_lut_t<V, 512, 1, 8, 8>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return; } break;
            default: break;
        }
    } else {
        switch (g_fused_config) {
            case 0: if constexpr (V != 4) { launch_lut_t<V, 512, 0, 6>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return; } break;
            case 1: launch_lut_t<V, 512, 1, 4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return;
            case 2: if constexpr (V == 4) { launch_lut_t<V, 1024, 1, 4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return; } break;
            case 3: launch_lut_t<V, 1024, 0, 8>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); return;
            default: break;
        }
    }
#endif
    // ONE launch shape per variant (what production runs and the parity tests cover):
    //   interval tables (6 / 7 / 8; single hue sector, every table row one run): 1024 threads, 2 workgroups per CU (8 waves per SIMD,
    //     <= 64 registers), register prefetch one pass ahead -- with the launches rotating over buffers beyond the Infinity Cache 3-10 %
    //     faster than 2 x 512 threads;
    //   bit tables / generic (0 / 1 / 2 / 3): 1024 threads, 4 waves per SIMD, register prefetch (round 4: 0.0705 / 0.093 ms per B = 256
    //     launch against 0.100 / 0.110 for 512 threads x 6 waves per SIMD, which spilled);
    //   generic with tie re-evaluation (4): 512 threads, no prefetch (the tie path needs more than 80 registers).
    if constexpr (V >= 5) launch_lut_t<V, 1024, 1, 8>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream);
    else if constexpr (V == 4) launch_lut_t<V, 512, 0, 4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream);
    else launch_lut_t<V, 1024, 1, 4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream);
}

// variant: 0/1/2 single sector r/g/b (bit tables), 3 generic, 4 generic with tie re-evaluation, 5 timing-only,
//          6/7/8 single sector r/g/b with interval tables
void launch_fused_mask_lut(const uint8_t* d_frames, int n, int H, int W, int hue_shift, const int lo[3],
                           const int hi[3], uint32_t* d_tables, int variant, uint8_t* d_masks,
                           hipStream_t stream)
{
    Bounds B;
    for (int c = 0; c < 3; ++c) { B.lo[c] = lo[c]; B.hi[c] = hi[c]; }
    {   // read at every launch (a getenv is nothing beside a launch): tests and A/B scripts switch it inside one process
        const char* e = diag_env("MELF_FUSED_CONFIG");
        g_fused_config = e ? atoi(e) & 15 : -1;
    }
    switch (variant) {
        case 0: launch_lut_v<0>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); break;
        case 1: launch_lut_v<1>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); break;
        case 2: launch_lut_v<2>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); break;
        case 3: launch_lut_v<3>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); break;
#ifdef MELF_DIAG
        case 5: launch_lut_v<5>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); break;   // timing only (garbage output)
#endif
        case 6: launch_lut_v<6>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); break;
        case 7: launch_lut_v<7>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); break;
        case 8: launch_lut_v<8>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); break;
        default: launch_lut_v<4>(d_frames, n, H, W, hue_shift, B, d_tables, d_masks, stream); break;
    }
}

#ifdef MELF_DIAG
// ---------------------------------------------------------------------------
// Measurement aid (bench.py's `stream_ceiling`; no pixel arithmetic, no caller in the product path): a BARE persistent stream
// of the fused kernel's traffic mix over the caller's buffers -- 48 bytes in, 16 bytes out per thread and step, lane-contiguous
// 16-byte loads, non-temporal stores, 1024-thread workgroups, two per CU like the kernel -- so that "what this part streams
// for this mix" is measured in the same run, on the same buffers, as the kernel's own figure.  Two ways of dealing the
// chunks (a chunk = one step of a workgroup = 48 KiB in, 16 KiB out): the static grid-stride split, and blocks of CH chunks
// from a work queue (the first two blocks of a workgroup are static, every further one is requested a block ahead).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void stream_step(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t c, int tid)
{
    const u32x4* p = in + c * 3072 + tid;
    const u32x4 a = p[0], b = p[1024], d = p[2048];
    u32x4 o;
    o.x = a.x ^ b.x ^ d.x; o.y = a.y ^ b.y ^ d.y; o.z = a.z ^ b.z ^ d.z; o.w = a.w ^ b.w ^ d.w;
    __builtin_nontemporal_store(o, out + c * 1024 + tid);
}

// static split with the kernel's register prefetch: the next chunk's 48 bytes are requested before this chunk is stored
__global__ __launch_bounds__(1024) void k_stream_probe_prefetch(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t nchunks)
{
    const int tid = threadIdx.x;
    size_t c = blockIdx.x;
    if (c >= nchunks) return;
    const u32x4* p = in + c * 3072 + tid;
    u32x4 a = p[0], b = p[1024], d = p[2048];
    for (;;) {
        const size_t cn = c + gridDim.x;
        const bool more = cn < nchunks;
        const u32x4* pn = in + (more ? cn : c) * 3072 + tid;   // unconditional (address-clamped) loads: exact vmcnt waits
        const u32x4 a2 = pn[0], b2 = pn[1024], d2 = pn[2048];
        u32x4 o;
        o.x = a.x ^ b.x ^ d.x; o.y = a.y ^ b.y ^ d.y; o.z = a.z ^ b.z ^ d.z; o.w = a.w ^ b.w ^ d.w;
        __builtin_nontemporal_store(o, out + c * 1024 + tid);
        if (!more) break;
        a = a2; b = b2; d = d2;
        c = cn;
    }
}

// "comb" variants (what the fused kernel's static split does): every workgroup walks its OWN contiguous run of chunks, top to
// bottom, instead of all workgroups sweeping one moving window; STRIDE48: with the kernel's lane pattern on top (a thread's 48
// bytes are contiguous: three 16-byte loads at a 48-byte lane stride); BARRIERS: two workgroup barriers per step and the
// 16-bit-per-thread LDS hand-over of the kernel's passes; TABLES: the kernel's 64 KiB of static LDS, filled before the first step.
template <bool STRIDE48, bool BARRIERS, bool TABLES>
__global__ __launch_bounds__(1024) void k_stream_probe_comb(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t nchunks)
{
    __shared__ uint32_t ring[BARRIERS ? 2048 : 1];
    __shared__ __attribute__((aligned(16))) uint32_t tables[TABLES ? 16384 : 4];
    const int tid = threadIdx.x;
    const uint32_t per = (nchunks + gridDim.x - 1) / gridDim.x;
    const size_t c0 = (size_t)blockIdx.x * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
    if (c0 >= c1) return;
    auto src = [&](size_t c, int k) -> const u32x4* { return STRIDE48 ? in + c * 3072 + 3 * tid + k : in + c * 3072 + 1024 * k + tid; };
    u32x4 a = *src(c0, 0), b = *src(c0, 1), d = *src(c0, 2);
    if (TABLES) {
        for (int i = tid; i < 4096; i += 1024) *(u32x4*)(tables + 4 * i) = u32x4{(uint32_t)i, 0u, 0u, 0u};
        __syncthreads();
    }
    uint32_t acc = 0;
    for (size_t c = c0; c < c1; ++c) {
        const size_t cn = c + 1 < c1 ? c + 1 : c;   // register prefetch of the next chunk, as the kernel
        const u32x4 a2 = *src(cn, 0), b2 = *src(cn, 1), d2 = *src(cn, 2);
        u32x4 o;
        o.x = a.x ^ b.x ^ d.x; o.y = a.y ^ b.y ^ d.y; o.z = a.z ^ b.z ^ d.z; o.w = a.w ^ b.w ^ d.w;
        if (TABLES) o.x ^= tables[(o.y & 255u) * 64 + (tid & 31)];
        if (BARRIERS) {
            ((uint16_t*)ring)[tid] = (uint16_t)o.x;
            __syncthreads();
            acc = ring[(tid >> 1) ^ 1];
            __syncthreads();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            o.w ^= acc;
        }
        __builtin_nontemporal_store(o, out + c * 1024 + tid);
        a = a2; b = b2; d = d2;
    }
}

template <bool DYN>
__global__ __launch_bounds__(1024) void k_stream_probe(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t nchunks, uint32_t CH,
                                                       uint32_t* __restrict__ wq)
{
    const int tid = threadIdx.x;
    if (!DYN) {
        for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) stream_step(in, out, c, tid);
        return;
    }
    __shared__ uint32_t nxt[2];
    const uint32_t nblocks = (nchunks + CH - 1) / CH;
    if (tid == 0) { nxt[0] = blockIdx.x; nxt[1] = blockIdx.x + gridDim.x; }
    __syncthreads();
    for (uint32_t it = 0;; ++it) {
        const uint32_t b = nxt[it & 1];
        if (b >= nblocks) break;
        __syncthreads();   // every thread has read nxt[it & 1]
        uint32_t v = 0;
        if (tid == 0) v = __hip_atomic_fetch_add(wq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the block after next
        const size_t c0 = (size_t)b * CH, c1 = c0 + CH < nchunks ? c0 + CH : nchunks;
        for (size_t c = c0; c < c1; ++c) stream_step(in, out, c, tid);
        if (tid == 0) nxt[it & 1] = v + 2u * gridDim.x;   // read two iterations on, behind the next iteration's barrier
    }
    if (tid == 0) {
        const uint32_t done = __hip_atomic_fetch_add(wq + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(wq, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(wq + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// chunks_per_block = 0: static split; -1: static split with one chunk of register prefetch; -2 .. -5: the "comb" variants (own
// contiguous run per workgroup; + 48-byte lane stride; + barriers and an LDS hand-over; + 64 KiB of LDS tables); > 0: work queue.  Streams floor(in_bytes / 48 KiB) chunks; returns the bytes moved (in + out).
size_t launch_stream_probe(const void* d_in, size_t in_bytes, void* d_out, int chunks_per_block, uint32_t* d_tables, hipStream_t stream,
                           hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const uint32_t nchunks = (uint32_t)std::min<size_t>(in_bytes / (48 * 1024), 0x7fffffffu);
    if (!nchunks) return 0;
    const int grid = (int)std::min<uint32_t>(512u, nchunks);
    if (chunks_per_block <= -2) {
#define MELF_COMB(S, B, T) hipExtLaunchKernelGGL((k_stream_probe_comb<S, B, T>), dim3(grid), dim3(1024), 0, stream, ev_start, ev_stop, 0, (const u32x4*)d_in, (u32x4*)d_out, nchunks)
        switch (chunks_per_block) {
            case -2: MELF_COMB(false, false, false); break;
            case -3: MELF_COMB(true, false, false); break;
            case -4: MELF_COMB(true, true, false); break;
            default: MELF_COMB(true, true, true); break;
        }
#undef MELF_COMB
    } else if (chunks_per_block < 0) {
        hipExtLaunchKernelGGL(k_stream_probe_prefetch, dim3(grid), dim3(1024), 0, stream, ev_start, ev_stop, 0, (const u32x4*)d_in, (u32x4*)d_out, nchunks);
    } else if (chunks_per_block > 0) {
        const int qslot = fused_queue_slot(d_tables, stream);
        if (qslot < 0) return 0;
        uint32_t* wq = d_tables + FUSED_TABLE_DWORDS + 16 * qslot;
        hipExtLaunchKernelGGL((k_stream_probe<true>), dim3(grid), dim3(1024), 0, stream, ev_start, ev_stop, 0, (const u32x4*)d_in, (u32x4*)d_out, nchunks,
                              (uint32_t)chunks_per_block, wq);
    } else {
        hipExtLaunchKernelGGL((k_stream_probe<false>), dim3(grid), dim3(1024), 0, stream, ev_start, ev_stop, 0, (const u32x4*)d_in, (u32x4*)d_out, nchunks, 0u,
                              (uint32_t*)nullptr);
    }
    return (size_t)nchunks * 64 * 1024;
}
#endif  // MELF_DIAG

void launch_fused_mask(const uint8_t* d_frames, int n, int H, int W, int hue_shift, const int lo[3], const int hi[3],
                       uint8_t* d_masks, hipStream_t stream)
{
    Bounds B;
    for (int c = 0; c < 3; ++c) { B.lo[c] = lo[c]; B.hi[c] = hi[c]; }
    const int wpr = (W + 31) >> 5;
    const size_t shmem = (size_t)2 * (FM_ROWS + 4) * wpr * sizeof(uint32_t);
    dim3 grid((H + FM_ROWS - 1) / FM_ROWS, n), block(256);
    hipLaunchKernelGGL(k_fused_mask, grid, block, shmem, stream, d_frames, H, W, hue_shift, B, d_masks);
}

}  // namespace melf

// ---------------------------------------------------------------------------
// Calibration stage kernels (reference: meterelf/_calibration.py:60-84, _image.py:34-44,
// _utils.py:64-88, 113-119).  Offline path: correctness first.
// ---------------------------------------------------------------------------
namespace melf {

// get_average_meter_image: every frame's meter crop is translated so that its dial match lands
// at (ax, ay) (cv2.warpAffine with an integer translation = exact shift, zero border), the
// float64 running mean is updated in the reference's operation order
//     p = p * ((k - 1) / k) + (img / 255.0) / k,  k = 2, 3, ...
// and the result is denormalised with (p * 255.0 + 0.5).astype(uint8).
__global__ __launch_bounds__(256) void k_aligned_average(const uint8_t* __restrict__ frames, int n, size_t frame_stride,
                                                         int row_stride, int x0, int y0, int rows, int cols,
                                                         const int32_t* __restrict__ mx, const int32_t* __restrict__ my,
                                                         int ax, int ay, uint8_t* __restrict__ out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * cols * 3) return;
    const int c = t % 3, x = (t / 3) % cols, y = t / (3 * cols);
    double p = 0.0;
    for (int i = 0; i < n; ++i) {
        const int sx = x - (ax - mx[i]), sy = y - (ay - my[i]);
        double v = 0.0;
        if (sx >= 0 && sx < cols && sy >= 0 && sy < rows)
            v = (double)frames[(size_t)i * frame_stride + (size_t)(y0 + sy) * row_stride + (size_t)(x0 + sx) * 3 + c];
        const double nv = v / 255.0;
        if (i == 0) {
            p = nv;
        } else {
            const double k = (double)(i + 1);
            p = p * ((double)i / k) + nv / k;
        }
    }
    out[t] = (uint8_t)(int)(p * 255.0 + 0.5);
}

void launch_aligned_average(const uint8_t* d_frames, int n, size_t frame_stride, int row_stride, int x0, int y0, int rows,
                            int cols, const int32_t* d_mx, const int32_t* d_my, int ax, int ay, uint8_t* d_out,
                            hipStream_t stream)
{
    const int total = rows * cols * 3;
    hipLaunchKernelGGL(k_aligned_average, dim3((total + 255) / 256), dim3(256), 0, stream, d_frames, n, frame_stride,
                       row_stride, x0, y0, rows, cols, d_mx, d_my, ax, ay, d_out);
}

// cv2.inRange on a packed 3-channel u8 image (get_mask_by_color, meterelf/_utils.py:113-119)
__global__ __launch_bounds__(256) void k_inrange3(const uint8_t* __restrict__ img, int npx, Bounds B, uint8_t* __restrict__ out)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= npx) return;
    const uint8_t* v = img + (size_t)p * 3;
    out[p] = (v[0] >= B.lo[0] && v[0] <= B.hi[0] && v[1] >= B.lo[1] && v[1] <= B.hi[1] && v[2] >= B.lo[2] && v[2] <= B.hi[2]) ? 255 : 0;
}

void launch_inrange3(const uint8_t* d_img, int npx, const int lo[3], const int hi[3], uint8_t* d_out, hipStream_t stream)
{
    Bounds B;
    for (int c = 0; c < 3; ++c) { B.lo[c] = lo[c]; B.hi[c] = hi[c]; }
    hipLaunchKernelGGL(k_inrange3, dim3((npx + 255) / 256), dim3(256), 0, stream, d_img, npx, B, d_out);
}

}  // namespace melf
