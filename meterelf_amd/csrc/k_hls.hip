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
#include "melf_device.h"
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
