// Shared device/host helpers of libmeterelf_hip: the bit-exact float32
// restatement of cv2.cvtColor(COLOR_BGR2HLS_FULL) on u8 data and small
// wave-level utilities.  gfx950 only (wave64).
//
// Build with -ffp-contract=off: OpenCV's SSE2 baseline never fuses a*b+c and
// every intermediate rounding below is part of the result.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace melf {

// OpenCV converts each row in blocks of 256 px; inside a block the first
// 4*floor(n/4) pixels go through the 4-wide SIMD body and the rest through the
// scalar tail loop, whose saturation formula has a different operation order
// (RGB2HLS_b / RGB2HLS_f of OpenCV 3.4 color_hsv.cpp).  `cols` is the width of
// the image handed to cvtColor (the meter_rect crop, meterelf/_image.py:29-32).
__host__ __device__ inline bool hls_scalar_tail(int x, int cols)
{
    int x0 = x & ~255;
    int dn = cols - x0 < 256 ? cols - x0 : 256;
    return (x - x0) >= (dn & ~3);
}

__host__ __device__ inline int sat_u8_rne(float v)
{
    float r = rintf(v);  // round half to even (cvRound / cvtps2dq)
    return r < 0.f ? 0 : (r > 255.f ? 255 : (int)r);
}

// L only (what template matching needs): cv2.split(hls)[1], meterelf/_image.py:59
__host__ __device__ inline int hls_lightness(int b8, int g8, int r8)
{
    const float inv255 = 1.f / 255.f;
    float b = (float)b8 * inv255, g = (float)g8 * inv255, r = (float)r8 * inv255;
    float vmax = fmaxf(fmaxf(r, g), b);
    float vmin = fminf(fminf(r, g), b);
    float l = (vmax + vmin) * 0.5f;
    return sat_u8_rne(l * 255.f);
}

// The same value with fewer operations (k_prep_lplane converts every pixel of every crop): x -> fl(x * (1/255)) is
// monotonic, so the channel maximum / minimum can be taken on the integers before the scaling (2 multiplications
// instead of 3); (s * 0.5f) * 255.f == s * 127.5f exactly (the halving is exact, so both forms round the same real
// number once); and the result lies in [0, 255] for every input, so the saturation never acts.  Equal to
// hls_lightness for all 2^16 (max, min) pairs (tests/test_gpu_parity.py: the L plane of random and structured crops).
__host__ __device__ inline int hls_lightness_fast(int b8, int g8, int r8)
{
    const float inv255 = 1.f / 255.f;
    const float vmax = fmaxf(fmaxf((float)r8, (float)g8), (float)b8) * inv255;
    const float vmin = fminf(fminf((float)r8, (float)g8), (float)b8) * inv255;
    return (int)rintf((vmax + vmin) * 127.5f);
}

// convert_to_hls for one pixel (meterelf/_utils.py:100-102): returns H (with the
// uint8 wrap-around hue shift applied), L, S.
__host__ __device__ inline void hls_pixel(int b8, int g8, int r8, bool scalar_tail, int hue_shift,
                                          int& H, int& L, int& S)
{
    const float inv255 = 1.f / 255.f;
    const float hscale = 256.f / 360.f;
    float b = (float)b8 * inv255, g = (float)g8 * inv255, r = (float)r8 * inv255;
    float vmax = fmaxf(fmaxf(r, g), b);
    float vmin = fminf(fminf(r, g), b);
    float diff = vmax - vmin;
    float sum = vmax + vmin;
    float l = sum * 0.5f;
    float h = 0.f, s = 0.f;
    if (diff > 1.1920928955078125e-07f /* FLT_EPSILON */) {
        float den;
        if (scalar_tail)
            den = l < 0.5f ? sum : (2.f - vmax) - vmin;
        else
            den = l < 0.5f ? sum : 2.0f - sum;
        s = diff / den;
        float k = 60.f / diff;
        if (vmax == r)
            h = (g - b) * k + (g < b ? 360.f : 0.f);
        else if (vmax == g)
            h = (b - r) * k + 120.f;
        else
            h = (r - g) * k + 240.f;
    }
    H = (sat_u8_rne(h * hscale) + hue_shift) & 255;
    L = sat_u8_rne(l * 255.f);
    S = sat_u8_rne(s * 255.f);
}

#ifdef __HIPCC__
// ---- wave64 helpers -------------------------------------------------------
// Wave-wide reductions and scans on the DPP path (GFX9 row shifts + row broadcasts): six VALU steps and no LDS round
// trips, where a __shfl_xor butterfly is six dependent ds_bpermute / ds_swizzle round trips.  A lane without a DPP
// source (or masked off) keeps `ident`, the operation's neutral element.
template <int CTRL, int RM, int BM>
__device__ __forceinline__ int dpp_i32(int ident, int v)
{
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, RM, BM, false);
}
template <int CTRL, int RM, int BM>
__device__ __forceinline__ double dpp_f64(double ident, double v)
{
    const uint64_t iu = __double_as_longlong(ident), vu = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)iu, (int)(uint32_t)vu, CTRL, RM, BM, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(iu >> 32), (int)(uint32_t)(vu >> 32), CTRL, RM, BM, false);
    return __longlong_as_double(((uint64_t)hi << 32) | lo);
}
// inclusive scan over the 64 lanes (lane 63 ends with the total)
#define MELF_DPP_SCAN(T, DPP, v, ident, OP)                  \
    v = OP(v, DPP<0x111, 0xf, 0xf>(ident, v)); /* row_shr:1 */  \
    v = OP(v, DPP<0x112, 0xf, 0xf>(ident, v)); /* row_shr:2 */  \
    v = OP(v, DPP<0x114, 0xf, 0xe>(ident, v)); /* row_shr:4 */  \
    v = OP(v, DPP<0x118, 0xf, 0xc>(ident, v)); /* row_shr:8 */  \
    v = OP(v, DPP<0x142, 0xa, 0xf>(ident, v)); /* row_bcast:15 */ \
    v = OP(v, DPP<0x143, 0xc, 0xf>(ident, v)); /* row_bcast:31 */
__device__ __forceinline__ int op_add_i(int a, int b) { return a + b; }
__device__ __forceinline__ double op_add_d(double a, double b) { return a + b; }
__device__ __forceinline__ double op_min_d(double a, double b) { return fmin(a, b); }
__device__ inline int wave_scan_i32(int v)
{
    MELF_DPP_SCAN(int, dpp_i32, v, 0, op_add_i)
    return v;
}
__device__ inline int wave_sum_i32(int v)
{
    MELF_DPP_SCAN(int, dpp_i32, v, 0, op_add_i)
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ inline double readlane63_f64(double v)
{
    const uint64_t u = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 63);
    return __longlong_as_double(((uint64_t)hi << 32) | lo);
}
// 64-bit floating-point reductions, DPP scans too (two moves per step): lane 63 ends with the result, two v_readlane hand it to
// every lane.  (Until round 5 these were __shfl_xor butterflies -- twelve ds_bpermute round trips each; with four waves per SIMD the
// LDS latency hides behind the other waves and the DPP form's extra moves had measured slower, but k_dials ends with ONE wave per
// SIMD running its angle phase alone, and there the round trips are the critical path: tools/dials_clock.py.)
// Sums: zero fill (bound_ctrl: a lane without a source adds 0.0).  Minimum / maximum: a lane without a source keeps its own value,
// which an idempotent operation ignores.
template <int CTRL>
__device__ __forceinline__ double dpp0_f64(double v)
{
    const uint64_t vu = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)vu, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(vu >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((uint64_t)hi << 32) | lo);
}
__device__ inline double wave_sum_f64(double v)
{
    v += dpp0_f64<0x111>(v); v += dpp0_f64<0x112>(v); v += dpp0_f64<0x114>(v); v += dpp0_f64<0x118>(v);
    v += dpp0_f64<0x142>(v); v += dpp0_f64<0x143>(v);
    return readlane63_f64(v);
}
__device__ __forceinline__ double vmin_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double vmax_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define MELF_DPP_FOLD_F64(OP, v)                       \
    v = OP(v, dpp_f64<0x111, 0xf, 0xf>(v, v));         \
    v = OP(v, dpp_f64<0x112, 0xf, 0xf>(v, v));         \
    v = OP(v, dpp_f64<0x114, 0xf, 0xe>(v, v));         \
    v = OP(v, dpp_f64<0x118, 0xf, 0xc>(v, v));         \
    v = OP(v, dpp_f64<0x142, 0xa, 0xf>(v, v));         \
    v = OP(v, dpp_f64<0x143, 0xc, 0xf>(v, v));
__device__ inline double wave_min_f64(double v) { MELF_DPP_FOLD_F64(vmin_f64, v) return readlane63_f64(v); }
__device__ inline double wave_max_f64(double v) { MELF_DPP_FOLD_F64(vmax_f64, v) return readlane63_f64(v); }
__device__ inline uint64_t shfl_u64(uint64_t v, int src)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}
// value of the lane above (lane-1) / below (lane+1); `edge` for lanes 0 / 63.
// DPP whole-wave shifts (wave_shr:1 / wave_shl:1, GFX9 encodings 0x138 / 0x130): two VALU moves per 64-bit value, the
// lane without a source keeps `old` = the edge value.  (The ds_bpermute route costs an LDS round trip per half.)
__device__ inline uint64_t row_up(uint64_t v, int lane, uint64_t edge)
{
    (void)lane;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)edge, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(edge >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ inline uint64_t row_down(uint64_t v, int lane, uint64_t edge)
{
    (void)lane;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)edge, (int)(uint32_t)v, 0x130, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(edge >> 32), (int)(uint32_t)(v >> 32), 0x130, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
#endif

}  // namespace melf
