// K2 -- cv2.matchTemplate(L, template, TM_CCOEFF) + cv2.minMaxLoc
// (reference: meterelf/_utils.py:91-97, called from meterelf/_image.py:57-66).
//
// Exact integer formulation: cc[y,x] = sum_ij T[i,j] * I[y+i,x+j] in u32 (max
// 119*188*255^2 < 2^31), window sum in u32, then OpenCV's own post-pass
//   R = float32( double(cc) - double(winsum) * mean(T) ).
// OpenCV reaches cc through a float32 DFT; the exact value is what that
// approximates.
//
// Mapping (gfx950, wave64): one workgroup = 4 waves = a 44-row x 64-col tile of
// the correlation map.  The needed image rows (L plane, computed on the fly from
// the BGR frame) are staged once into LDS.  Lane = output column; each lane
// keeps MATCH_R = 11 consecutive output rows in registers, so one 4-byte image
// window (one LDS dword read + one v_alignbit) feeds 11 v_dot4_u32_u8 against
// wave-uniform template dwords that arrive through the scalar cache.  The window
// sum rides along as one extra dot4 with 0x01010101 per window.
#include <limits.h>

#include "melf_device.h"
#include "melf_internal.h"

namespace melf {

__device__ inline bool better(float v, int i, float bv, int bi)
{
    // minMaxLoc: first maximum in raster order
    return i != INT_MAX && (bi == INT_MAX || v > bv || (v == bv && i < bi));
}

template <bool FROM_BGR>
__global__ __launch_bounds__(256) void k_match(MatchSrc src, MatchGeom g, const uint32_t* __restrict__ tplT,
                                               int rh, int rw, int nrb, float* __restrict__ result_map,
                                               MatchPartial* __restrict__ partials, int nparts)
{
    constexpr int R = MATCH_R;
    extern __shared__ uint32_t lds[];
    __shared__ float s_v[MATCH_WAVES];
    __shared__ int s_i[MATCH_WAVES];

    const int tile = blockIdx.x, f = blockIdx.y;
    const int rb = tile % nrb, cb = tile / nrb;
    const int yb = rb * MATCH_RBLK, xb = cb * MATCH_CBLK;
    const uint8_t* img = src.base + (size_t)f * src.frame_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // ---- stage the tile's image rows into LDS as packed u8 (zero outside) ----
    for (int row = wave; row < g.lds_rows; row += MATCH_WAVES) {
        const int y = yb + row;
        const uint8_t* prow = img + (size_t)(src.y0 + y) * src.row_stride;
        for (int c4 = lane; c4 < g.ldsw; c4 += 64) {
            uint32_t packed = 0;
            if (y < src.rows) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int x = xb + c4 * 4 + k;
                    uint32_t v = 0;
                    if (x < src.cols) {
                        if (FROM_BGR) {
                            const uint8_t* p = prow + (size_t)(src.x0 + x) * 3;
                            v = (uint32_t)hls_lightness(p[0], p[1], p[2]);
                        } else {
                            v = prow[src.x0 + x];
                        }
                    }
                    packed |= v << (8 * k);
                }
            }
            lds[row * g.ldsw + c4] = packed;
        }
    }
    __syncthreads();

    // ---- sliding-window correlation ----
    const int q = lane >> 2;
    const uint32_t sh = (uint32_t)(lane & 3) * 8u;
    const int yr0 = wave * R;
    uint32_t acc[R], ws[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = ws[r] = 0;

    const int T = g.th + R - 1;
    const int last = g.tw4 - 1;
    for (int t = 0; t < T; ++t) {
        const uint32_t* row = lds + (yr0 + t) * g.ldsw + q;
        const uint32_t* tp = tplT + (t + R - 1);  // tplT[jj][padded_row], padded_row = (t - r) + R - 1
        uint32_t d0 = row[0];
        uint32_t rsum = 0;
        for (int jj = 0; jj < last; ++jj) {
            const uint32_t d1 = row[jj + 1];
            const uint32_t w = __builtin_amdgcn_alignbit(d1, d0, sh);
            d0 = d1;
            rsum = __builtin_amdgcn_udot4(w, 0x01010101u, rsum, false);
            const uint32_t* tq = tp + jj * g.trows;
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_udot4(w, tq[-r], acc[r], false);
        }
        {
            const uint32_t d1 = row[last + 1];
            const uint32_t w = __builtin_amdgcn_alignbit(d1, d0, sh);
            rsum = __builtin_amdgcn_udot4(w, g.last_ones, rsum, false);
            const uint32_t* tq = tp + last * g.trows;
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_udot4(w, tq[-r], acc[r], false);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if ((unsigned)(t - r) < (unsigned)g.th) ws[r] += rsum;
    }

    // ---- OpenCV post-pass + first-max reduction ----
    float bestv = 0.f;
    int besti = INT_MAX;
    const int x = xb + lane;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = yb + yr0 + r;
        if (y < rh && x < rw) {
            double num = (double)acc[r];
            num -= (double)ws[r] * g.tmean;
            const float v = (float)num;
            const int idx = y * rw + x;
            if (result_map) result_map[(size_t)f * rh * rw + idx] = v;
            if (better(v, idx, bestv, besti)) { bestv = v; besti = idx; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bestv, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (better(ov, oi, bestv, besti)) { bestv = ov; besti = oi; }
    }
    if (lane == 0) { s_v[wave] = bestv; s_i[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < MATCH_WAVES; ++w)
            if (better(s_v[w], s_i[w], bestv, besti)) { bestv = s_v[w]; besti = s_i[w]; }
        MatchPartial p;
        p.val = bestv;
        p.idx = besti;
        partials[(size_t)f * nparts + tile] = p;
    }
}

int match_parts(const MatchGeom& g, int rows, int cols)
{
    const int rh = rows - g.th + 1, rw = cols - g.tw + 1;
    if (rh <= 0 || rw <= 0) return 0;
    const int nrb = (rh + MATCH_RBLK - 1) / MATCH_RBLK, ncb = (rw + MATCH_CBLK - 1) / MATCH_CBLK;
    return nrb * ncb;
}

void launch_match(const MatchSrc& src, bool from_bgr, int n, const MatchGeom& g, const uint32_t* d_tplT,
                  float* d_result_map, MatchPartial* d_partials, int* nparts_out, hipStream_t stream)
{
    const int rh = src.rows - g.th + 1, rw = src.cols - g.tw + 1;
    const int nrb = (rh + MATCH_RBLK - 1) / MATCH_RBLK, ncb = (rw + MATCH_CBLK - 1) / MATCH_CBLK;
    const int nparts = nrb * ncb;
    if (nparts_out) *nparts_out = nparts;
    const size_t shmem = (size_t)g.lds_rows * g.ldsw * sizeof(uint32_t);
    dim3 grid(nparts, n), block(256);
    if (from_bgr)
        hipLaunchKernelGGL(k_match<true>, grid, block, shmem, stream, src, g, d_tplT, rh, rw, nrb, d_result_map,
                           d_partials, nparts);
    else
        hipLaunchKernelGGL(k_match<false>, grid, block, shmem, stream, src, g, d_tplT, rh, rw, nrb, d_result_map,
                           d_partials, nparts);
}

}  // namespace melf
