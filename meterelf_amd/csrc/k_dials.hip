// K3 -- per-dial needle reading + digit combine
// (reference: meterelf/_reading.py:19-115 get_meter_value, :118-151
// get_needle_points, :154-160 get_dial_color, :163-182
// determine_value_by_dial_positions; meterelf/_utils.py:18-42
// get_angle_by_vector; meterelf/_colors.py:38-50 get_range).
//
// Mapping (gfx950, wave64): one workgroup per frame, one wave per dial.
// Everything a dial needs lives inside its disk mask (radius R <= 29) plus the
// 2-px halo of the 3x3 closing, i.e. a window of at most 64 x 64 pixels, so a
// binary image is ONE 64-bit register per lane (lane = window row, bit = window
// column).  inRange masks come out of __ballot, the closing / flood fills / 8-
// connected labelling are shifts within the register plus neighbour-lane reads,
// contour areas are popcounts.  No pixel of the HLS image is ever written to
// memory: the window's BGR pixels are converted on the fly.
//
// cv2 semantics restated here:
//  * external contours of M = closed_mask & disk: the 8-connected components of
//    G = M u {pixels not 4-connected to the outside through ~M}; a component
//    inside another one's hole is not external and is swallowed by it;
//  * contourArea of an outer border = Q4(F) + Q3(F)/2 over the hole-filled
//    component F (2x2 blocks with 4 resp. 3 pixels set) -- tests/ check this
//    identity against a traced shoelace area;
//  * drawContours(thickness=-1) paints F.
#include <limits.h>
#include <stdlib.h>

#include "melf_device.h"
#include "melf_internal.h"

namespace melf {

__device__ inline double py_fmod(double a, double b)
{
    // CPython float_rem
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0) != (m < 0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

// get_angle_by_vector, meterelf/_utils.py:18-42; false = None
__device__ inline bool angle_by_vector(double x, double y, double& out)
{
    if (y == 0) {
        if (x > 0) { out = 0.25; return true; }
        if (x < 0) { out = 0.75; return true; }
        return false;
    }
    const double at = atan(x / y) / (2 * 3.141592653589793);
    // CPython's float % 1.0 without the library's fmod: v - trunc(v) IS fmod(v, 1.0) (the subtraction is exact), sign included
    // except at the zeros, which float_rem replaces by +0.0 anyway
    const double v = -at + (y > 0 ? 0.5 : 0.0);
    double m = v - trunc(v);
    if (m != 0.0) { if (m < 0) m += 1.0; } else { m = 0.0; }
    out = m;
    return true;
}

// determine_value_by_dial_positions, meterelf/_reading.py:163-182
__device__ inline double value_by_positions(double r4, double r3, double r2, double r1)
{
    int d3 = (int)r3 + ((py_fmod(r3, 1.0) > 0.55 && r4 <= 2) ? 1 : 0) - ((py_fmod(r3, 1.0) < 0.45 && r4 >= 8) ? 1 : 0);
    d3 = ((d3 % 10) + 10) % 10;
    int d2 = (int)r2 + ((py_fmod(r2, 1.0) > 0.55 && d3 <= 2) ? 1 : 0) - ((py_fmod(r2, 1.0) < 0.45 && d3 >= 8) ? 1 : 0);
    d2 = ((d2 % 10) + 10) % 10;
    int d1 = (int)r1 + ((py_fmod(r1, 1.0) > 0.55 && d2 <= 2) ? 1 : 0) - ((py_fmod(r1, 1.0) < 0.45 && d2 >= 8) ? 1 : 0);
    d1 = ((d1 % 10) + 10) % 10;
    return (d1 * 100.0) + (d2 * 10.0) + (d3 * 1.0) + r4 / 10.0;
}

struct Key {
    double a, d;
};
__device__ inline bool key_lt(const Key& p, const Key& q) { return p.a < q.a || (p.a == q.a && p.d < q.d); }
// One trimming round: the smallest of the lanes' `lo` keys and the largest of their `hi` keys.  The angle's extreme first (the two
// scans interleave); the lane that holds it is nearly always alone, and its distance is then a v_readlane away.
__device__ inline void wave_min_max_key(const Key& lo, const Key& hi, Key& klo, Key& khi)
{
    klo.a = wave_min_f64(lo.a);
    khi.a = wave_max_f64(hi.a);
    const uint64_t blo = __builtin_amdgcn_ballot_w64(lo.a == klo.a), bhi = __builtin_amdgcn_ballot_w64(hi.a == khi.a);
    if (__popcll(blo) == 1 && __popcll(bhi) == 1) {
        const int jl = __builtin_ctzll(blo), jh = __builtin_ctzll(bhi);
        const uint64_t dl = __double_as_longlong(lo.d), dh = __double_as_longlong(hi.d);
        klo.d = __longlong_as_double((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dl, jl) |
                                     ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dl >> 32), jl) << 32));
        khi.d = __longlong_as_double((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dh, jh) |
                                     ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dh >> 32), jh) << 32));
    } else {
        klo.d = wave_min_f64(lo.a == klo.a ? lo.d : 1e300);
        khi.d = wave_max_f64(hi.a == khi.a ? hi.d : -1e300);
    }
}

// Visits the ring points that the reference keeps (meterelf/_reading.py:53-69):
// set bits of `outer` (lane = row), angle within 0.25 turn of the momentum angle.
template <class F>
__device__ inline void for_each_kept(uint64_t outer, int lane, int wx0, int wy0, double cx, double cy,
                                     bool have_mom, double mom, F&& fn)
{
    const double dy = (double)(wy0 + lane) - cy;
    uint64_t bits = outer;
    while (bits) {
        const int x = __builtin_ctzll(bits);
        bits &= bits - 1;
        const double dx = (double)(wx0 + x) - cx;
        double a;
        if (angle_by_vector(dx, dy, a) && have_mom) {
            double dist = fabs(a - mom);
            const double dist2 = fabs(fabs(a - mom) - 1);
            if (dist2 < dist) dist = dist2;
            if (dist < 0.25) fn(a, dx * dx + dy * dy);
        }
    }
}

constexpr int DIAL_LIST_CAP = 768;   // candidate pixels of a dial window that take the exact float path
// Ring points (needle pixels inside the annulus) whose angles the angle phase caches.  512 since round 5: with 256 the timed
// workloads had a wave or two per batch just above it (257, 262 points), and the uncached path (an arctangent per point and pass,
// serially along the rows) cost such a wave 60 000 cycles instead of 14 000 -- and the launch, which ends with its slowest wave,
// 58 us instead of 38.  The squared distances are recomputed from the positions now (six instructions) instead of cached.
constexpr int RING_CAP = 512;
constexpr int RING_CAP_REGS = 1024;   // ... beyond the LDS cache: angles in registers, 16 per lane (positions: 2 KiB of the same LDS)
// LDS of one dial (one wave), carved from the kernel's dynamic block: the ring arrays of the angle phase re-use the
// candidate list and the in-range bits of the pixel phase (all of them dead by then).
//   [0, 4096) ra double[RING_CAP]          | pixel phase: [0, 3072) list_px u32[CAP], [3072, 4608) list_pos u16[CAP]
//   [4096, 5120) ring list u16[RING_CAP]   |              [4608, 5120) exact in-range bits, two dwords per window row
constexpr int DIAL_LDS_BYTES = 5120;
static_assert(DIAL_LIST_CAP * 4 + DIAL_LIST_CAP * 2 <= 4608 && RING_CAP * 8 + RING_CAP * 2 <= DIAL_LDS_BYTES, "dial LDS layout");

// Three bytes of a packed 3-channel pixel with ONE (unaligned) dword load instead of three byte loads: the
// dword starts one byte early (so it never runs past the buffer's end) except at the buffer's very first pixel.
// The top byte of the result is unspecified (every user looks at bytes 0..2 only).
__device__ __forceinline__ uint32_t load_px3(const uint8_t* p, const uint8_t* buffer_start)
{
    const uint32_t back = p == buffer_start ? 0u : 1u;
    uint32_t v;
    __builtin_memcpy(&v, p - back, 4);
    return v >> (8u * back);
}

// The same for a lane's COLUMN of pixels, col + row_off for wave-uniform row offsets: the address arithmetic of
// load_px3 (a 64-bit multiply-add, a 64-bit compare against the buffer's start, a select) cost ten issue slots per window
// row and lane, a tenth of the kernel's vector instructions.  Here the direction is fixed per LANE: the dword starts one
// byte early, except in the lane whose column begins at the buffer's first byte, which reads forward in every row (one
// byte into its right-hand neighbour: inside the buffer, rows being at least two pixels wide -- melf_ctx_create refuses
// a one-pixel-wide template).  Per row: one 64-bit add, the load, one shift.
struct PxColumn {
    const uint8_t* first;  // col - 1, or col in the lane at the buffer's start
    uint32_t shift;        // 8, or 0 there
};
__device__ __forceinline__ PxColumn px_column(const uint8_t* col, const uint8_t* buffer_start)
{
    const bool at_start = col == buffer_start;
    return PxColumn{at_start ? col : col - 1, at_start ? 0u : 8u};
}
__device__ __forceinline__ uint32_t load_px3_row(const PxColumn& c, size_t row_off)
{
    asm("" : "+s"(row_off));  // the offset stays a scalar product: otherwise the compiler folds it into one 64-bit vector multiply-add per row
    uint32_t v;
    __builtin_memcpy(&v, c.first + row_off, 4);
    return v >> c.shift;
}

// packed 16-bit arithmetic on two values per register (v_pk_*_u16)
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2v, a), __builtin_bit_cast(u16x2v, b)));
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2v, a), __builtin_bit_cast(u16x2v, b)));
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2v)(__builtin_bit_cast(u16x2v, a) + __builtin_bit_cast(u16x2v, b)));
}
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2v)(__builtin_bit_cast(u16x2v, a) - __builtin_bit_cast(u16x2v, b)));
}
__device__ __forceinline__ uint32_t pk_mul_u16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (u16x2v)(__builtin_bit_cast(u16x2v, a) * __builtin_bit_cast(u16x2v, b)));
}

// A frame's record, written by one wave: lanes 0 .. MELF_MAX_DIALS - 1 store pos[lane] / angle[lane] (zero beyond the
// context's dials), lane 0 the scalars.  Every byte of the record is written (no padding in melf_result).
__device__ __forceinline__ void write_record(melf_result* __restrict__ out, int lane, int status, int mx, int my, float mv, int failed_dial,
                                             uint32_t unreadable, double value, double pos_lane, double angle_lane)
{
    if (lane < MELF_MAX_DIALS) {
        out->pos[lane] = pos_lane;
        out->angle[lane] = angle_lane;
    }
    if (lane == 0) {
        out->status = status;
        out->match_x = mx; out->match_y = my;
        out->failed_dial = failed_dial;
        out->unreadable_mask = unreadable;
        out->match_val = mv;
        out->value = value;
    }
}
static_assert(sizeof(melf_result) == 24 + 16 * MELF_MAX_DIALS + 8, "melf_result has padding: write_record must fill it");

#ifdef MELF_DIALS_STAMP
// Diagnostic build only: shader-clock stamps at the phase boundaries of each wave (tools/dials_clock.py).
__device__ uint64_t g_dials_stamps[8 * 8192];
extern "C" __attribute__((visibility("default"))) int melf_debug_dials_stamps(uint64_t* out, int nwaves)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dials_stamps), sizeof(uint64_t) * 8 * (size_t)(nwaves < 8192 ? nwaves : 8192)) == hipSuccess ? 0 : -1;
}
// ... and the 100 MHz real-time counter at four of them (the shader-clock counters of different CUs cannot be compared) + HW_ID
__device__ uint64_t g_dials_real[8 * 8192];
extern "C" __attribute__((visibility("default"))) int melf_debug_dials_real(uint64_t* out, int nwaves)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dials_real), sizeof(uint64_t) * 8 * (size_t)(nwaves < 8192 ? nwaves : 8192)) == hipSuccess ? 0 : -1;
}
// ... and shader-clock stamps inside the closing / labelling and the momentum / angle phases
__device__ uint64_t g_dials_fine[16 * 8192];
extern "C" __attribute__((visibility("default"))) int melf_debug_dials_fine(uint64_t* out, int nwaves)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dials_fine), sizeof(uint64_t) * 16 * (size_t)(nwaves < 8192 ? nwaves : 8192)) == hipSuccess ? 0 : -1;
}
#define FSTAMPD(k) do { if (lane == 0 && (int)(blockIdx.x * (blockDim.x >> 6) + wv) < 8192) g_dials_fine[16 * (blockIdx.x * (blockDim.x >> 6) + wv) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DSTAMP(k) do { if (lane == 0 && (int)(blockIdx.x * (blockDim.x >> 6) + wv) < 8192) { \
        g_dials_stamps[8 * (blockIdx.x * (blockDim.x >> 6) + wv) + (k)] = __builtin_amdgcn_s_memtime(); \
        if ((k) < 6) g_dials_real[8 * (blockIdx.x * (blockDim.x >> 6) + wv) + (k)] = __builtin_amdgcn_s_memrealtime(); \
        if ((k) == 0) { uint32_t hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); uint32_t xc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc_)); \
                        g_dials_real[8 * (blockIdx.x * (blockDim.x >> 6) + wv) + 6] = hw_; g_dials_real[8 * (blockIdx.x * (blockDim.x >> 6) + wv) + 7] = xc_ & 15u; } } } while (0)
#else
#define DSTAMP(k) do { } while (0)
#define FSTAMPD(k) do { } while (0)
#endif

// Register budget: 128 of the SIMD's 512 ("amdgpu-num-vgpr" is doubled by the backend for gfx90a+'s unified file): four waves
// per SIMD, i.e. all 4 096 waves of a 1024-frame batch resident at once.  (Round 2 held it at 104 so that a wave fitted beside
// a register-capped match wave of the other caller stream; that variant is gone, and at 128 nothing spills and -- since round 4,
// tests/test_host_logic.py reads the code object's notes -- the kernel has no private segment at all.)
#ifndef MELF_DIALS_VGPRS
#define MELF_DIALS_VGPRS 64
#endif
// NR: window rows whose pixels a lane requests up front (the largest dial window of the context, rounded up to 8)
template <bool FROM_HLS, int NR>
__global__ __launch_bounds__(64 * MELF_MAX_DIALS, 4) __attribute__((amdgpu_num_vgpr(MELF_DIALS_VGPRS))) void k_dials(DialsSrc src, melf_params P,
                                                              const DialGeom* __restrict__ geom,
                                                              const uint64_t* __restrict__ rowmasks,
                                                              const MatchPartial* __restrict__ partials,
                                                              int nparts, int rw, melf_result* __restrict__ results)
{
    __shared__ int s_status[MELF_MAX_DIALS];
    __shared__ double s_pos[MELF_MAX_DIALS], s_angle[MELF_MAX_DIALS];
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];  // DIAL_LDS_BYTES per dial

    const int f = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // one dial per wave.  (Round 4 tried rotating the dial <-> wave assignment by the frame index, in case wave k of every
    // workgroup landed on SIMD k and one SIMD got all the small dials: no difference, 0.0631 / 0.0702 ms against 0.0639 / 0.0701
    // with event brackets, profiles/r04/dials_rotate_ab.txt -- the dispatcher already mixes them.)
    const int d = wv;
    DSTAMP(0);
    const uint8_t* frame = src.base + (size_t)f * src.frame_stride;
    uint8_t* const lds = s_dyn + (size_t)wv * DIAL_LDS_BYTES;
    uint32_t* const list_px = (uint32_t*)lds;                       // pixel phase
    uint16_t* const list_pos = (uint16_t*)(lds + 3072);             // pixel phase
    double* const s_ra_d = (double*)lds;                            // angle phase
    uint16_t* const ring_list = (uint16_t*)(lds + 4096);
    uint32_t* const mask_d = (uint32_t*)(lds + 4608);

    // ---- minMaxLoc over the K2 partials; DialsNotFoundError check (_image.py:62-64) ----
    int mx = 0, my = 0;
    float mv = 0.f;
    if (!FROM_HLS) {
        // every wave folds the partials itself (a few dozen entries): no LDS hand-off, no workgroup barrier
        float bv = 0.f;
        int bi = INT_MAX;
        for (int k = lane; k < nparts; k += 64) {
            const MatchPartial p = partials[(size_t)f * nparts + k];
            if (p.idx != INT_MAX && (bi == INT_MAX || p.val > bv || (p.val == bv && p.idx < bi))) { bv = p.val; bi = p.idx; }
        }
        {   // (value, first index) maximum over the lanes, DPP scan: lane 63 ends with the result
            auto fold = [&](float ov, int oi) {
                if (oi != INT_MAX && (bi == INT_MAX || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            };
#define MELF_PFOLD(CTRL, RM, BM) \
            fold(__int_as_float(dpp_i32<CTRL, RM, BM>(0, __float_as_int(bv))), dpp_i32<CTRL, RM, BM>(INT_MAX, bi));
            MELF_PFOLD(0x111, 0xf, 0xf) MELF_PFOLD(0x112, 0xf, 0xf) MELF_PFOLD(0x114, 0xf, 0xe) MELF_PFOLD(0x118, 0xf, 0xc)
            MELF_PFOLD(0x142, 0xa, 0xf) MELF_PFOLD(0x143, 0xc, 0xf)
#undef MELF_PFOLD
            bv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bv), 63));
            bi = __builtin_amdgcn_readlane(bi, 63);
        }
        mv = bv;
        const int mi = bi;
        mx = mi % rw;
        my = mi / rw;
        if ((double)mv < P.match_threshold) {
            if (wv == 0) write_record(results + f, lane, MELF_FRAME_DIALS_NOT_FOUND, mx, my, mv, -1, 0u, 0.0, 0.0, 0.0);
            return;
        }
    }

    DSTAMP(1);
    // ---- one wave per dial ----
    const DialGeom G = geom[d];
    const melf_dial D = P.dial[d];
    // one dial per wave: its geometry is wave-uniform (tell the compiler, so that row / window tests are scalar)
    const int wx0 = __builtin_amdgcn_readfirstlane(G.wx0), wy0 = __builtin_amdgcn_readfirstlane(G.wy0), ws = __builtin_amdgcn_readfirstlane(G.ws);
    FSTAMPD(8);    // the dial's geometry has arrived

    // Every pixel this wave needs is requested here, before anything waits: the 5x5 colour core (one pixel per lane)
    // and the window (four pixels of a row per lane and load, below).  Unconditional loads at
    // clamped (always valid) coordinates: a load inside a bounds check makes the compiler wait for each one separately.
    // The colour phase below then runs while the window's rows are still in flight -- one memory round trip for the
    // wave instead of one for the core and one per 16 rows (half of the wave's life was such waits).
    // Lanes and rows beyond the window request its LAST column / row again (the same cache lines: nothing more leaves HBM).
    // Round 5: they used to request the 64 x NR pixels around the window's corner whatever its size -- a 49 x 49 window
    // fetched 12.3 KB of which it used 7.2 (the 1.6x of the traffic counters).
    const int Xl = wx0 + lane;
    const bool colvalid = lane < ws && Xl >= 0 && Xl < P.tw;
    const int Xc = min(max(wx0 + min(lane, ws - 1), 0), P.tw - 1);
    const int ylast = ws - 1;   // wave-uniform
    const bool tail = !FROM_HLS && hls_scalar_tail(mx + Xl, src.crop_cols);
    const size_t rstride = FROM_HLS ? (size_t)P.tw * 3 : (size_t)src.row_stride;
    const uint8_t* const origin = FROM_HLS ? frame : frame + (size_t)(src.y0 + my) * src.row_stride + (size_t)(src.x0 + mx) * 3;
    const int coreX = G.core_x - 2 + lane % 5, coreY = G.core_y - 2 + (lane < 25 ? lane / 5 : 0);
    const bool corevalid = lane < 25 && coreX >= 0 && coreX < P.tw && coreY >= 0 && coreY < P.th;
    const uint32_t corepx = load_px3(origin + (size_t)min(max(coreY, 0), P.th - 1) * rstride + (size_t)min(max(coreX, 0), P.tw - 1) * 3, src.base);
    const PxColumn pcol = px_column(origin + (size_t)Xc * 3, src.base);   // (the exact path's loads: one pixel per lane and row)
    const int th1 = __builtin_amdgcn_readfirstlane(P.th - 1);
    const int rs_u = __builtin_amdgcn_readfirstlane((int)rstride);  // uniform: the row offsets below are scalar products
    // The window for the integer test, round 5: a lane fetches FOUR pixels of a row as one aligned 16-byte load (the 12 bytes and
    // what the alignment adds), sixteen lanes a row, four rows per instruction -- NR / 4 loads per wave instead of NR.  Until then
    // a lane fetched its column's pixel of every row as an unaligned dword: the texture addresser took 12 cycles per such
    // instruction, and the 784 of a CU's sixteen waves were issued over the launch's first 4.6 us with nothing else to do
    // (tools/dials_clock.py: "pixels requested" 9 700 cycles; 4 200 for a wave alone on its SIMD).
    constexpr int NG = NR / 4;
    static_assert(NR % 4 == 0, "window rows come in groups of four");
    const int rg = lane >> 4, pc = lane & 15;
    const int npiece = (ws + 3) >> 2;   // wave-uniform: 12-byte pieces of a window row
    // wave-uniform: every piece lies inside the crop's rows (no column clamping: a lane's four pixels stay four neighbours) and
    // the last load ends inside the frames' buffer; otherwise every pixel takes the exact path below
    const uint8_t* const buf_end = src.base + src.readable;   // (not frames x stride: the last frame of a padded-stride buffer may end earlier)
    const bool quads = !FROM_HLS && ((uintptr_t)src.base & 3) == 0 && wx0 >= 0 && wx0 + 4 * npiece <= P.tw &&
                       origin + (size_t)th1 * rstride + (size_t)(wx0 + 4 * npiece) * 3 + 4 <= buf_end;
    u32x4v raw[NG];
    uint32_t mshift = 0;   // bytes between a load's aligned address and its first pixel (0..3), two bits per load
    if (quads) {
        const uint8_t* const lane0 = origin + (size_t)(wx0 + 4 * min(pc, npiece - 1)) * 3;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int Y = min(max(wy0 + min(4 * g + rg, ylast), 0), th1);
            const uint8_t* const a = lane0 + (size_t)Y * (size_t)rs_u;
            mshift |= ((uint32_t)(uintptr_t)a & 3u) << (2 * g);
            raw[g] = *(const u32x4v*)((uintptr_t)a & ~(uintptr_t)3);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    FSTAMPD(9);    // every pixel requested
#ifdef MELF_DIALS_STAMP
    asm volatile("; colour core here" :: "v"(corepx) : "memory");
    FSTAMPD(10);   // the colour core has arrived
#endif

    // get_dial_color (_reading.py:154-160): mean of the 5x5 core, Python round()
    int sh = 0, sl = 0, ss = 0, cnt = 0;
    if (corevalid) {
        if (FROM_HLS) { sh = corepx & 255; sl = (corepx >> 8) & 255; ss = (corepx >> 16) & 255; }
        else hls_pixel(corepx & 255, (corepx >> 8) & 255, (corepx >> 16) & 255, hls_scalar_tail(mx + coreX, src.crop_cols), P.hue_shift, sh, sl, ss);
        cnt = 1;
    }
    sh = wave_sum_i32(sh); sl = wave_sum_i32(sl); ss = wave_sum_i32(ss); cnt = wave_sum_i32(cnt);
    const double inv = cnt ? 1.0 / (double)cnt : 0.0;  // cv::mean: sum * (1./N)
    const int ch = (int)rint((double)sh * inv), cl = (int)rint((double)sl * inv), cs = (int)rint((double)ss * inv);
    // HlsColor.get_range (_colors.py:38-50): plain clamped ints, hue does not wrap
    const int loh = max(ch - D.range_h, 0), hih = min(ch + D.range_h, 255);
    const int lol = max(cl - D.range_l, 0), hil = min(cl + D.range_l, 255);
    const int los = max(cs - D.range_s, 0), his = min(cs + D.range_s, 255);

    DSTAMP(2);
    // inRange over the window (get_mask_by_color, _utils.py:113-119): row masks via ballot.
    uint64_t m0 = 0, V = 0;
    auto exact_rows = [&]() {  // every window pixel through the exact float path (its own loads: the rare path must not
                               // keep the prefilter's pixel registers alive)
        for (int yc = 0; yc < ws; yc += 16) {
            uint32_t pxe[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int Y = min(max(wy0 + min(yc + k, ylast), 0), th1);
                pxe[k] = load_px3_row(pcol, (size_t)((int64_t)Y * rs_u));
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int y = yc + k, Y = wy0 + y;
                const bool valid = y < ws && colvalid && Y >= 0 && Y < P.th;
                bool in = false;
                if (valid) {
                    int H, L, S;
                    const uint32_t px = pxe[k];
                    if (FROM_HLS) { H = px & 255; L = (px >> 8) & 255; S = (px >> 16) & 255; }
                    else hls_pixel(px & 255, (px >> 8) & 255, (px >> 16) & 255, tail, P.hue_shift, H, L, S);
                    in = H >= loh && H <= hih && L >= lol && L <= hil && S >= los && S <= his;
                }
                const uint64_t b = __builtin_amdgcn_ballot_w64(in), vb = __builtin_amdgcn_ballot_w64(valid);
                if (lane == y) { m0 = b; V = vb; }
            }
        }
    };
    if (FROM_HLS) {
        exact_rows();
    } else {
        // Two steps.  (1) An integer test that can only err towards "maybe" picks candidates: with
        // sum = max + min and diff = max - min, L is sum/2 rounded either way and S is 255*diff/den rounded
        // (den = sum below mid-grey, 510 - sum above), both float paths within 1e-4 of the real value, so a
        // pixel whose L or S misses the bounds by a whole unit cannot be in range.  (2) The candidates --
        // typically the needle, a tenth of the window -- take the exact float path, 64 at a time: the candidates
        // are appended (position and pixel, by the lane that holds it) to a list in LDS as they are found (the list's
        // order does not matter: the exact test ORs bits into the window's row masks).  More candidates than the list
        // holds: every pixel takes the exact path.
        // The test runs on TWO pixels of the lane's four per instruction, as packed 16-bit halves (v_pk_*_u16): 255 * diff and
        // den * (bound) stay below 2^16, so the compares  255 diff >= (los - 1) den  and  255 diff <= (his + 1) den  (the
        // inequalities above halved) are exact in 16 bits; "x outside [lo, hi]" is  x != min(max(x, lo), hi).  Grey pixels
        // (diff = 0) pass here when los <= 1 although only los = 0 admits them: a candidate more for the exact test.
        int total = 0;  // wave-uniform
        const uint64_t colb = __builtin_amdgcn_ballot_w64(colvalid);
        const uint32_t LO2 = (uint32_t)max(2 * lol - 1, 0) * 0x00010001u, HI2 = (uint32_t)(2 * hil + 1) * 0x00010001u;
        const uint32_t SLO = (uint32_t)max(los - 1, 0) * 0x00010001u, SHI = (uint32_t)(his + 1) * 0x00010001u;
        const int xleft = ws - 4 * pc;   // pixels j < xleft of this lane's four are window columns
        // the lanes whose pixel j is a window column, as wave masks: a pixel's candidacy is then ballot(test) & masks, scalar work
        const uint64_t xm[4] = {__builtin_amdgcn_ballot_w64(0 < xleft), __builtin_amdgcn_ballot_w64(1 < xleft),
                                __builtin_amdgcn_ballot_w64(2 < xleft), __builtin_amdgcn_ballot_w64(3 < xleft)};
        if (quads) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                // the lane's 12 bytes: B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
                const uint32_t ms = (mshift >> (2 * g)) & 3u;
                const uint32_t e0 = __builtin_amdgcn_alignbyte(raw[g].y, raw[g].x, ms);
                const uint32_t e1 = __builtin_amdgcn_alignbyte(raw[g].z, raw[g].y, ms);
                const uint32_t e2 = __builtin_amdgcn_alignbyte(raw[g].w, raw[g].z, ms);
                const int y = 4 * g + rg, Y = wy0 + y;
                const uint64_t rowb = __builtin_amdgcn_ballot_w64((y < ws) & (Y >= 0) & (Y < P.th));
#pragma unroll
                for (int h = 0; h < 2; ++h) {   // pixels (0, 1), then (2, 3): two per instruction, as packed 16-bit halves
                    const uint32_t B2 = h ? __builtin_amdgcn_perm(e2, e1, 0x0c050c02u) : __builtin_amdgcn_perm(e1, e0, 0x0c030c00u);
                    const uint32_t G2 = h ? __builtin_amdgcn_perm(e2, e1, 0x0c060c03u) : __builtin_amdgcn_perm(e1, e0, 0x0c040c01u);
                    const uint32_t R2 = h ? __builtin_amdgcn_perm(e2, e1, 0x0c070c04u) : __builtin_amdgcn_perm(e1, e0, 0x0c050c02u);
                    const uint32_t vmax = pk_max_u16(pk_max_u16(B2, G2), R2), vmin = pk_min_u16(pk_min_u16(B2, G2), R2);
                    const uint32_t sum = pk_add_u16(vmax, vmin), diff = pk_sub_u16(vmax, vmin);
                    const uint32_t den = pk_min_u16(sum, pk_sub_u16(0x01fe01feu, sum));
                    const uint32_t lbad = pk_min_u16(pk_max_u16(sum, LO2), HI2) ^ sum;
                    const uint32_t a = pk_mul_u16(diff, 0x00ff00ffu);
                    const uint32_t sbad = pk_min_u16(pk_max_u16(a, pk_mul_u16(den, SLO)), pk_mul_u16(den, SHI)) ^ a;
                    const uint32_t bad = lbad | sbad;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int j = 2 * h + q;
                        const uint64_t cb = __builtin_amdgcn_ballot_w64(q ? bad < 0x10000u : (bad & 0xffffu) == 0u) & rowb & xm[j];
                        if (cb) {   // wave-uniform: most rows above and below the needle have no candidate at all
                            const int slot = total + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cb, 0u));
                            const bool cand = (cb >> lane) & 1ull;
                            if (cand && slot < DIAL_LIST_CAP) {
                                list_pos[slot] = (uint16_t)(y << 6 | (4 * pc + j));
                                list_px[slot] = j == 0 ? e0 : (j == 1 ? __builtin_amdgcn_alignbyte(e1, e0, 3) : (j == 2 ? __builtin_amdgcn_alignbyte(e2, e1, 2) : e2 >> 8));
                            }
                            total += __popcll(cb);
                        }
                    }
                }
            }
        } else {
            total = DIAL_LIST_CAP + 1;   // a window that leaves the crop: the exact path for every pixel
        }
        // lane y's row of the window's valid-pixel mask (what the per-row ballots of `valid` used to deliver)
        V = (lane < ws && wy0 + lane >= 0 && wy0 + lane < P.th && lane < NR) ? colb : 0ull;
        DSTAMP(6);
#ifdef MELF_DIALS_STAMP
        if (lane == 0 && (int)(blockIdx.x * (blockDim.x >> 6) + wv) < 8192) g_dials_real[8 * (blockIdx.x * (blockDim.x >> 6) + wv) + 6] |= (uint64_t)total << 32;
#endif
        if (total > DIAL_LIST_CAP) {
            exact_rows();
        } else {
            mask_d[lane] = 0;
            mask_d[64 + lane] = 0;
            DSTAMP(7);
            for (int t = lane; t < total; t += 64) {
                const int e = list_pos[t], y = e >> 6, x = e & 63;
                const uint32_t px = list_px[t];
                int H, L, S;
                hls_pixel(px & 255, (px >> 8) & 255, (px >> 16) & 255, hls_scalar_tail(mx + wx0 + x, src.crop_cols), P.hue_shift, H, L, S);
                if (H >= loh && H <= hih && L >= lol && L <= hil && S >= los && S <= his)
                    atomicOr(&mask_d[2 * y + (x >> 5)], 1u << (x & 31));
            }
            m0 = (uint64_t)mask_d[2 * lane] | ((uint64_t)mask_d[2 * lane + 1] << 32);
        }
    }

    // dilate then erode, 3x3, pixels outside the dials crop never win (_reading.py:128-130)
    DSTAMP(3);
    const uint64_t hz = m0 | (m0 << 1) | (m0 >> 1);
    const uint64_t dil = (hz | row_up(hz, lane, 0) | row_down(hz, lane, 0)) | ~V;
    const uint64_t he = dil & ((dil << 1) | 1ull) & ((dil >> 1) | (1ull << 63));
    const uint64_t mde = he & row_up(he, lane, ~0ull) & row_down(he, lane, ~0ull) & V;

    const uint64_t disk = rowmasks[((size_t)d * 3 + 0) * 64 + lane];
    const uint64_t annulus = rowmasks[((size_t)d * 3 + 1) * 64 + lane];
    const uint64_t outside0 = rowmasks[((size_t)d * 3 + 2) * 64 + lane];  // truly outside the disk (no pockets)
    const uint64_t M = mde & disk;
    FSTAMPD(0);   // closing done, row masks loaded

    int status = 0;  // 0 ok, 1 no contours, 2 unreadable
    double pos = 0.0, angle = 0.0;
    if (__builtin_amdgcn_ballot_w64(M != 0) == 0) {
        status = 1;  // NeedleContoursNotFoundError (_reading.py:137-138)
    } else {
        // 8-connected components of a set in raster order of their first pixel = external contours in cv2's discovery order; keep
        // the largest by contourArea (stable sort + [-1] with cv2's reversed list => earliest wins ties).  Two propagation steps
        // per convergence test: the test (compare, ballot, branch) is a third of a lone wave's trip.
        uint64_t bestF = 0;
        int best2 = -1, ncomp = 0;
        auto label = [&](const uint64_t Gs) {
            uint64_t rem = Gs;
            bestF = 0; best2 = -1; ncomp = 0;
            for (;;) {
                const uint64_t rowsb = __builtin_amdgcn_ballot_w64(rem != 0);
                if (rowsb == 0) break;
                const int r0 = __builtin_ctzll(rowsb);
                // row r0 of `rem` for every lane: r0 is wave-uniform, so this is two v_readlane (no LDS permute)
                const uint64_t rv = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rem, r0) |
                                    ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rem >> 32), r0) << 32);
                const int b0 = __builtin_ctzll(rv);
                uint64_t s = (lane == r0) ? (1ull << b0) : 0ull;
                for (;;) {
                    const uint64_t h3 = s | (s << 1) | (s >> 1);
                    const uint64_t n1 = (h3 | row_up(h3, lane, 0) | row_down(h3, lane, 0)) & Gs;
                    const uint64_t g3 = n1 | (n1 << 1) | (n1 >> 1);
                    const uint64_t n2 = (g3 | row_up(g3, lane, 0) | row_down(g3, lane, 0)) & Gs;
                    const bool ch2 = n2 != n1;
                    s = n2;
                    if (__builtin_amdgcn_ballot_w64(ch2) == 0) break;
                }
                const uint64_t a = s, b = row_down(s, lane, 0), a1 = a >> 1, b1 = b >> 1;
                const int c = 2 * __builtin_popcountll(a & a1 & b & b1) + __builtin_popcountll(a & a1 & b & ~b1) +
                              __builtin_popcountll(a & a1 & ~b & b1) + __builtin_popcountll(a & ~a1 & b & b1) +
                              __builtin_popcountll(~a & a1 & b & b1);
                const int area2 = wave_sum_i32(c);  // 2 * cv2.contourArea
                if (area2 > best2) { best2 = area2; bestF = s; }
                rem &= ~s;
                ++ncomp;
            }
        };
        // The reference fills each EXTERNAL contour: a component's holes count as its pixels.  Needle masks rarely have holes, and
        // whether this one has any follows from its Euler number without a flood (Gray's bit quads: 4 E = Q1 - Q3 - 2 QD for
        // 8-connected foreground / 4-connected background, E = components - holes; the window's 2-pixel margin makes the all-zero
        // quads beyond its edges irrelevant): no holes -> the components of M are the filled contours.  Otherwise, as until round 5:
        // the pixels of ~M that are 4-connected to the outside of the disk (the seed is host-computed: unfilled pockets inside the
        // reference's disk mask -- thin rings -- are not outside), everything else is M plus what its outer borders enclose.
        int euler4, isolated;
        {
            const uint64_t a = M, b = row_down(M, lane, 0), a1 = a >> 1, b1 = b >> 1;
            const uint64_t odd = a ^ a1 ^ b ^ b1;                          // one or three pixels of the quad
            const uint64_t three = odd & ((a & a1) | (b & b1));
            const uint64_t diag = (a & b1 & ~a1 & ~b) | (a1 & b & ~a & ~b1);
            euler4 = wave_sum_i32(__builtin_popcountll(odd) - 2 * __builtin_popcountll(three) - 2 * __builtin_popcountll(diag));
            // isolated pixels (no 8-neighbour) are components without holes: the rest of the mask has the Euler number E - their count
            const uint64_t up = row_up(M, lane, 0);
            const uint64_t nb = (a << 1) | a1 | up | (up << 1) | (up >> 1) | b | (b << 1) | b1;
            isolated = wave_sum_i32(__builtin_popcountll(a & ~nb));
        }
        // E <= 0 -- here for the mask without its isolated pixels -- means at least as many holes as components, i.e. at least one
        // (or nothing but isolated pixels: the flood path is right for any mask, only slower): no need to label M first to find out
        bool holes = euler4 - 4 * isolated <= 0;
        if (!holes) {
            label(M);
            holes = 4 * ncomp != euler4;
        }
        FSTAMPD(1);
#ifdef MELF_DIALS_STAMP
        if (lane == 0 && (int)(blockIdx.x * (blockDim.x >> 6) + wv) < 8192) {
            g_dials_fine[16 * (blockIdx.x * (blockDim.x >> 6) + wv) + 12] = (uint64_t)(uint32_t)euler4;
            g_dials_fine[16 * (blockIdx.x * (blockDim.x >> 6) + wv) + 13] = (uint64_t)(uint32_t)ncomp;
            g_dials_fine[16 * (blockIdx.x * (blockDim.x >> 6) + wv) + 14] = (uint64_t)(uint32_t)isolated;
        }
#endif
        if (holes) {
            const uint64_t freeb = ~M;
            uint64_t o = outside0;
            for (;;) {   // two propagation steps per convergence test, as in the labelling
                const uint64_t n1 = o | ((((o << 1) | (o >> 1)) | row_up(o, lane, ~0ull) | row_down(o, lane, ~0ull)) & freeb);
                const uint64_t n2 = n1 | ((((n1 << 1) | (n1 >> 1)) | row_up(n1, lane, ~0ull) | row_down(n1, lane, ~0ull)) & freeb);
                const bool ch2 = n2 != n1;
                o = n2;
                if (__builtin_amdgcn_ballot_w64(ch2) == 0) break;
            }
            label(~o);   // M plus everything its outer borders enclose
        }
        // contourArea > 100: filled contour, else the whole closed mask (_reading.py:141-148); both are
        // used only through `& dial.mask` / `& dial.circle_mask` (:150, :51) -- the filled contour can
        // cover pocket pixels that the disk mask lacks.
        DSTAMP(4);
        const uint64_t N = (best2 > 200 ? bestF : M) & disk;
        const uint64_t outer = N & annulus;

        // momentum vector (_reading.py:32-41)
        const double cx = D.cx, cy = D.cy;
        double sx = 0.0, sy = 0.0;
        FSTAMPD(2);
        {
            // a row's x term from eight table entries (melf_ctx_create: momx[d][byte of the row mask][its value] = the sum of
            // sign(dx) dx^2 over the byte's set bits), its y term times its pixel count: no loop over the pixels (a lone wave
            // spent 3 000-4 000 cycles in the longest row's)
            const double dy = (double)(wy0 + lane) - cy;
            const double ty = (dy < 0 ? -1.0 : 1.0) * (dy * dy);
            const double* const mt = (const double*)(rowmasks + (size_t)P.ndials * 3 * 64) + (size_t)d * 2048;
            double part[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) part[b] = mt[b * 256 + (int)((N >> (8 * b)) & 255)];
            sx = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
            sy = ty * (double)__builtin_popcountll(N);
        }
        sx = wave_sum_f64(sx);
        sy = wave_sum_f64(sy);
        FSTAMPD(3);   // momentum sums reduced (loop + two reductions)
        const double msign = D.negative_momentum ? -1.0 : 1.0;
        double mom = 0.0;
        const bool have_mom = angle_by_vector(msign * sx, msign * sy, mom);
        FSTAMPD(4);   // momentum angle

        // The ring points (needle pixels inside the annulus, a few dozen) are compacted into a list first: the
        // angle of each -- a double-precision atan -- is then computed once, 64 points at a time, and the three
        // passes of the reference (count / minimum, trimming, weighted mean) run over the cached values.  Walking
        // each row's bits per lane instead costs an atan per point and pass, times the longest row.
        const int rmine = __popcll(outer);
        const int rincl = wave_scan_i32(rmine);
        const int rtotal = __builtin_amdgcn_readlane(rincl, 63);
#ifdef MELF_DIALS_STAMP
        if (lane == 0 && (int)(blockIdx.x * (blockDim.x >> 6) + wv) < 8192) g_dials_real[8 * (blockIdx.x * (blockDim.x >> 6) + wv) + 7] |= (uint64_t)rtotal << 32;
#endif
        if (rtotal <= RING_CAP) {
            uint16_t* list = ring_list;
            double* ra = s_ra_d;
            {
                int at = rincl - rmine;
                uint64_t bits = outer;
                while (bits) {
                    const int x = __builtin_ctzll(bits);
                    bits &= bits - 1;
                    list[at++] = (uint16_t)(lane << 6 | x);
                }
            }
            FSTAMPD(5);   // ring list written
            auto ring_d2 = [&](int e) {   // squared distance from the dial's centre of list entry e (row << 6 | column)
                const double dx = (double)(wx0 + (e & 63)) - cx, dy = (double)(wy0 + (e >> 6)) - cy;
                return dx * dx + dy * dy;
            };
            int nk = 0;
            double mina = 1e300;
            for (int t = lane; t < rtotal; t += 64) {
                const int e = list[t];
                const double dx = (double)(wx0 + (e & 63)) - cx, dy = (double)(wy0 + (e >> 6)) - cy;
                double a, keep = __builtin_nan("");
                if (angle_by_vector(dx, dy, a) && have_mom) {
                    double dist = fabs(a - mom);
                    const double dist2 = fabs(fabs(a - mom) - 1);
                    if (dist2 < dist) dist = dist2;
                    if (dist < 0.25) { keep = a; ++nk; if (a < mina) mina = a; }
                }
                ra[t] = keep;
            }
            nk = wave_sum_i32(nk);
            mina = wave_min_f64(mina);
            if (nk == 0) {
                status = 2;  // unreadable dial (_reading.py:79-81)
            } else {
                FSTAMPD(6);   // ring angles cached, count and minimum known
                const int cut = nk >= 5 ? min(2, (nk - 3) / 2) : 0;
                const Key PINF = {1e300, 1e300}, NINF = {-1e300, -1e300};
                Key klo = NINF, khi = PINF;
                if (cut > 0) {
                    Key l1 = PINF, l2 = PINF, h1 = NINF, h2 = NINF;
                    for (int t = lane; t < rtotal; t += 64) {
                        const double a = ra[t];
                        if (a != a) continue;
                        Key k;
                        k.a = fabs(a - mina) < 0.75 ? a : a - 1;
                        k.d = ring_d2(list[t]);
                        if (key_lt(k, l1)) { l2 = l1; l1 = k; } else if (key_lt(k, l2)) { l2 = k; }
                        if (key_lt(h1, k)) { h2 = h1; h1 = k; } else if (key_lt(h2, k)) { h2 = k; }
                    }
                    for (int c2 = 0; c2 < cut; ++c2) {
                        wave_min_max_key(l1, h1, klo, khi);
                        if (l1.a == klo.a && l1.d == klo.d) { l1 = l2; l2 = PINF; }
                        if (h1.a == khi.a && h1.d == khi.d) { h1 = h2; h2 = NINF; }
                    }
                }
                FSTAMPD(7);   // trimming keys known
                double sad = 0.0, sd = 0.0;
                for (int t = lane; t < rtotal; t += 64) {
                    const double a = ra[t];
                    if (a != a) continue;
                    Key k;
                    k.a = fabs(a - mina) < 0.75 ? a : a - 1;
                    k.d = ring_d2(list[t]);
                    if (cut == 0 || (key_lt(klo, k) && key_lt(k, khi))) {
                        sad += k.a * k.d;
                        sd += k.d;
                    }
                }
                sad = wave_sum_f64(sad);
                sd = wave_sum_f64(sd);
                angle = sad / sd;
                const double fixed = angle - (D.angle_of_zero / 360.0);
                pos = py_fmod(10.0 * fixed, 10.0);  // _reading.py:95-96
            }
        } else if (rtotal <= RING_CAP_REGS) {
            // More ring points than the LDS cache holds, up to 1024 (a dial whose disk is mostly "needle": a flare, a wrong match):
            // the positions go to the list (2 bytes each), the angles stay in REGISTERS, sixteen per lane (point t = lane + 64 j
            // in ang[j]; the window's pixel registers are dead by now).  One arctangent per point, as in the cached path -- the
            // passes below this branch compute it once per point AND pass, serially along the rows (60 000 cycles for 260 points).
            uint16_t* list = (uint16_t*)lds;   // [0, 2048)
            {
                int at = rincl - rmine;
                uint64_t bits = outer;
                while (bits) {
                    const int x = __builtin_ctzll(bits);
                    bits &= bits - 1;
                    list[at++] = (uint16_t)(lane << 6 | x);
                }
            }
            auto ring_d2 = [&](int e) {
                const double dx = (double)(wx0 + (e & 63)) - cx, dy = (double)(wy0 + (e >> 6)) - cy;
                return dx * dx + dy * dy;
            };
            constexpr int NJ = RING_CAP_REGS / 64;
            double ang[NJ];
            int nk = 0;
            double mina = 1e300;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                double keep = __builtin_nan("");
                if (64 * j < rtotal) {   // wave-uniform
                    const int t = lane + 64 * j;
                    if (t < rtotal) {
                        const int e = list[t];
                        const double dx = (double)(wx0 + (e & 63)) - cx, dy = (double)(wy0 + (e >> 6)) - cy;
                        double a;
                        if (angle_by_vector(dx, dy, a) && have_mom) {
                            double dist = fabs(a - mom);
                            const double dist2 = fabs(fabs(a - mom) - 1);
                            if (dist2 < dist) dist = dist2;
                            if (dist < 0.25) { keep = a; ++nk; if (a < mina) mina = a; }
                        }
                    }
                }
                ang[j] = keep;
            }
            nk = wave_sum_i32(nk);
            mina = wave_min_f64(mina);
            if (nk == 0) {
                status = 2;  // unreadable dial (_reading.py:79-81)
            } else {
                const int cut = nk >= 5 ? min(2, (nk - 3) / 2) : 0;
                const Key PINF = {1e300, 1e300}, NINF = {-1e300, -1e300};
                Key klo = NINF, khi = PINF;
                if (cut > 0) {
                    Key l1 = PINF, l2 = PINF, h1 = NINF, h2 = NINF;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const double a = ang[j];
                        if (a != a) continue;
                        Key k;
                        k.a = fabs(a - mina) < 0.75 ? a : a - 1;
                        k.d = ring_d2(list[lane + 64 * j]);
                        if (key_lt(k, l1)) { l2 = l1; l1 = k; } else if (key_lt(k, l2)) { l2 = k; }
                        if (key_lt(h1, k)) { h2 = h1; h1 = k; } else if (key_lt(h2, k)) { h2 = k; }
                    }
                    for (int c2 = 0; c2 < cut; ++c2) {
                        wave_min_max_key(l1, h1, klo, khi);
                        if (l1.a == klo.a && l1.d == klo.d) { l1 = l2; l2 = PINF; }
                        if (h1.a == khi.a && h1.d == khi.d) { h1 = h2; h2 = NINF; }
                    }
                }
                double sad = 0.0, sd = 0.0;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const double a = ang[j];
                    if (a != a) continue;
                    Key k;
                    k.a = fabs(a - mina) < 0.75 ? a : a - 1;
                    k.d = ring_d2(list[lane + 64 * j]);
                    if (cut == 0 || (key_lt(klo, k) && key_lt(k, khi))) {
                        sad += k.a * k.d;
                        sd += k.d;
                    }
                }
                sad = wave_sum_f64(sad);
                sd = wave_sum_f64(sd);
                angle = sad / sd;
                const double fixed = angle - (D.angle_of_zero / 360.0);
                pos = py_fmod(10.0 * fixed, 10.0);  // _reading.py:95-96
            }
        } else {
            // pass 1: count and minimum of the kept angles (_reading.py:79-82)
            int nk = 0;
            double mina = 1e300;
            for_each_kept(outer, lane, wx0, wy0, cx, cy, have_mom, mom, [&](double a, double) {
                ++nk;
                if (a < mina) mina = a;
            });
            nk = wave_sum_i32(nk);
                mina = wave_min_f64(mina);
            if (nk == 0) {
                status = 2;  // unreadable dial (_reading.py:79-81)
            } else {
                // pass 2: the `cut` smallest / largest (angle, sqdist) tuples to drop (_reading.py:86-91)
                const int cut = nk >= 5 ? min(2, (nk - 3) / 2) : 0;
                const Key PINF = {1e300, 1e300}, NINF = {-1e300, -1e300};
                Key klo = NINF, khi = PINF;
                if (cut > 0) {
                    Key l1 = PINF, l2 = PINF, h1 = NINF, h2 = NINF;
                    for_each_kept(outer, lane, wx0, wy0, cx, cy, have_mom, mom, [&](double a, double dd) {
                        Key k;
                        k.a = fabs(a - mina) < 0.75 ? a : a - 1;
                        k.d = dd;
                        if (key_lt(k, l1)) { l2 = l1; l1 = k; } else if (key_lt(k, l2)) { l2 = k; }
                        if (key_lt(h1, k)) { h2 = h1; h1 = k; } else if (key_lt(h2, k)) { h2 = k; }
                    });
                    for (int c2 = 0; c2 < cut; ++c2) {
                        wave_min_max_key(l1, h1, klo, khi);
                        if (l1.a == klo.a && l1.d == klo.d) { l1 = l2; l2 = PINF; }
                        if (h1.a == khi.a && h1.d == khi.d) { h1 = h2; h2 = NINF; }
                    }
                }
                // pass 3: distance^2-weighted mean angle of the rest (_reading.py:92-94)
                double sad = 0.0, sd = 0.0;
                for_each_kept(outer, lane, wx0, wy0, cx, cy, have_mom, mom, [&](double a, double dd) {
                    Key k;
                    k.a = fabs(a - mina) < 0.75 ? a : a - 1;
                    k.d = dd;
                    if (cut == 0 || (key_lt(klo, k) && key_lt(k, khi))) {
                        sad += k.a * dd;
                        sd += dd;
                    }
                });
                sad = wave_sum_f64(sad);
                sd = wave_sum_f64(sd);
                angle = sad / sd;
                const double fixed = angle - (D.angle_of_zero / 360.0);
                pos = py_fmod(10.0 * fixed, 10.0);  // _reading.py:95-96
            }
        }
    }
    DSTAMP(5);
    if (lane == 0) { s_status[d] = status; s_pos[d] = pos; s_angle[d] = angle; }
    __syncthreads();

    // ---- error aggregation + digit combine (_reading.py:98-111): wave 0, the record written field by field from LDS (a local
    // melf_result indexed by name_order lived in scratch: 168 bytes of private segment per lane of every wave, round 3) ----
    if (wv == 0) {
        int st = MELF_FRAME_OK, failed = -1;
        uint32_t unread = 0;
        double value = 0.0;
        for (int k = 0; k < P.ndials; ++k)
            if (s_status[k] == 2) unread |= 1u << k;
        for (int k = 0; k < P.ndials; ++k)
            if (s_status[k] == 1) { st = MELF_FRAME_NEEDLE_CONTOURS_NOT_FOUND; failed = k; break; }
        if (st == MELF_FRAME_NEEDLE_CONTOURS_NOT_FOUND) {
            unread &= (1u << failed) - 1u;   // the reference raises at this dial: later dials are never looked at
        } else if (unread) {
            st = MELF_FRAME_ANGLE_UNDETERMINED;
        } else if (P.ndials == 4) {
            value = value_by_positions(s_pos[P.name_order[0]], s_pos[P.name_order[1]], s_pos[P.name_order[2]], s_pos[P.name_order[3]]);
        }
        const bool has = lane < P.ndials;
        write_record(results + f, lane, st, mx, my, mv, failed, unread, value, has ? s_pos[has ? lane : 0] : 0.0, has ? s_angle[has ? lane : 0] : 0.0);
    }
}

void launch_dials(const DialsSrc& src, bool from_hls, int n, const melf_params& P, const DialGeom* d_geom,
                  const uint64_t* d_rowmasks, const MatchPartial* d_partials, int nparts, int rw,
                  melf_result* d_results, hipStream_t stream, int ws_max)
{
    dim3 grid(n), block(64 * P.ndials);
    const size_t shmem = (size_t)P.ndials * DIAL_LDS_BYTES;
    const int nr = ws_max <= 32 ? 32 : (ws_max <= 40 ? 40 : (ws_max <= 48 ? 48 : (ws_max <= 52 ? 52 : (ws_max <= 56 ? 56 : 64))));
#define MELF_DIALS_LAUNCH(HLS, NRV) \
    hipLaunchKernelGGL((k_dials<HLS, NRV>), grid, block, shmem, stream, src, P, d_geom, d_rowmasks, d_partials, nparts, rw, d_results)
#define MELF_DIALS_NR(HLS)                                   \
    switch (nr) {                                            \
        case 32: MELF_DIALS_LAUNCH(HLS, 32); break;          \
        case 40: MELF_DIALS_LAUNCH(HLS, 40); break;          \
        case 48: MELF_DIALS_LAUNCH(HLS, 48); break;          \
        case 52: MELF_DIALS_LAUNCH(HLS, 52); break;          \
        case 56: MELF_DIALS_LAUNCH(HLS, 56); break;          \
        default: MELF_DIALS_LAUNCH(HLS, 64); break;          \
    }
    if (from_hls) { MELF_DIALS_NR(true) } else { MELF_DIALS_NR(false) }
#undef MELF_DIALS_NR
#undef MELF_DIALS_LAUNCH
}

}  // namespace melf
