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
__device__ inline int wave_sum_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline uint64_t shfl_u64(uint64_t v, int src)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}
// value of the lane above (lane-1) / below (lane+1); `edge` for lanes 0 / 63
__device__ inline uint64_t row_up(uint64_t v, int lane, uint64_t edge)
{
    uint64_t r = shfl_u64(v, (lane + 63) & 63);
    return lane == 0 ? edge : r;
}
__device__ inline uint64_t row_down(uint64_t v, int lane, uint64_t edge)
{
    uint64_t r = shfl_u64(v, (lane + 1) & 63);
    return lane == 63 ? edge : r;
}
#endif

}  // namespace melf
