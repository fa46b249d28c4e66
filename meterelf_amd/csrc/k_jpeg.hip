// Baseline JPEG decode on the GPU: replaces the host decode behind
// cv2.imread (meterelf/_image.py:49) for the files the meter cameras write -- baseline sequential
// DCT, 8 bit, Huffman, one interleaved scan, YCbCr 4:2:0 / 4:2:2 / 4:4:4 or greyscale.
//
// The result must be the bytes libjpeg(-turbo) produces with its defaults (what cv2.imread and
// Pillow both use): JDCT_ISLOW inverse DCT, "fancy" triangle-filter chroma upsampling, the
// 16-bit fixed-point YCbCr->RGB tables.  All of that is integer arithmetic, restated here from
// the algorithm descriptions in the libjpeg documentation (jidctint / jdsample / jdcolor) and
// pinned bit-for-bit against Pillow on the reference's 304 fixture files plus synthetic files.
//
//   host   parse markers, canonical Huffman data, strip byte stuffing / RSTn markers (recording where the
//          restart intervals begin)
//   J1     k_jpeg_huff      one workgroup per image, one lane per bit-stream segment (speculative decode until
//                           the segments' exit states reach a fixed point) -> int16 coefficient blocks in decode order
//          k_jpeg_huff_rst  streams with restart intervals: one lane per interval, no speculation needed
//   J2     k_jpeg_idct      one thread per 8x8 block: dequantise + ISLOW IDCT -> u8 planes
//   J3     k_jpeg_color420  8 pixels per thread: fancy upsample + YCC->BGR -> NHWC frame (k_jpeg_color: other modes)
#include "melf_internal.h"
#include "melf_threads.h"

#include <emmintrin.h>

#include <algorithm>
#include <climits>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace melf {

// ------------------------------------------------------------------ host: parsing ----
static const uint8_t ZIGZAG_TO_NATURAL[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffSpec {
    uint8_t bits[17];  // bits[l] = number of codes of length l
    uint8_t vals[256];
    int nvals;
    bool set;
};

struct JpegHeader {
    int H = 0, W = 0, ncomp = 0;
    int id[3] = {0, 0, 0}, hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1}, tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    int restart_interval = 0;
    uint16_t qt[4][64];
    bool qt_set[4] = {false, false, false, false};
    HuffSpec dc[2], ac[2];
    size_t scan_begin = 0;
    bool saw_jfif = false, saw_adobe = false;
    int adobe_transform = 0;
    const char* why = nullptr;  // non-NULL: a valid-looking file this decoder does not handle
};

static inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Returns 0 = parsed (check h.why for "unsupported"), -1 = not a JPEG / truncated headers.
// The orientation tag (0x0112) of IFD0 of an EXIF block (TIFF header at `t`): 1..8, or 0 when absent / unreadable.
static int exif_orientation(const uint8_t* t, int n)
{
    if (n < 8) return 0;
    const bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return 0;
    auto u16 = [&](int o) { return le ? (t[o] | t[o + 1] << 8) : (t[o] << 8 | t[o + 1]); };
    auto u32 = [&](int o) { return le ? ((uint32_t)t[o] | (uint32_t)t[o + 1] << 8 | (uint32_t)t[o + 2] << 16 | (uint32_t)t[o + 3] << 24)
                                      : ((uint32_t)t[o] << 24 | (uint32_t)t[o + 1] << 16 | (uint32_t)t[o + 2] << 8 | (uint32_t)t[o + 3]); };
    if (u16(2) != 42) return 0;
    const uint32_t ifd = u32(4);
    if (ifd > (uint32_t)n - 2) return 0;
    const int cnt = u16((int)ifd);
    for (int k = 0; k < cnt; ++k) {
        const long e = (long)ifd + 2 + 12L * k;
        if (e + 12 > n) return 0;
        if (u16((int)e) == 0x0112) {
            const int v = u16((int)e + 8);   // type SHORT, count 1: the value sits in the first two bytes of the value field
            return (v >= 1 && v <= 8) ? v : 0;
        }
    }
    return 0;
}

static int parse_headers(const uint8_t* d, size_t n, JpegHeader& h)
{
    for (int i = 0; i < 2; ++i) { h.dc[i].set = false; h.ac[i].set = false; }
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return -1;
    size_t i = 2;
    bool have_sof = false;
    for (;;) {
        while (i < n && d[i] != 0xFF) ++i;  // tolerate garbage between segments like libjpeg's next_marker
        while (i < n && d[i] == 0xFF) ++i;
        if (i >= n) return -1;
        const int m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;  // parameterless
        if (m == 0xD9) return -1;                                           // EOI before SOS
        if (i + 2 > n) return -1;
        const int L = be16(d + i);
        if (L < 2 || i + L > n) return -1;
        const uint8_t* s = d + i + 2;
        const int len = L - 2;
        switch (m) {
            case 0xC0: case 0xC1: {  // baseline / extended sequential, Huffman
                if (len < 6) return -1;
                if (s[0] != 8) { h.why = "sample precision is not 8 bits"; }
                h.H = be16(s + 1); h.W = be16(s + 3); h.ncomp = s[5];
                if (h.ncomp != 1 && h.ncomp != 3) { h.why = "neither greyscale nor three components"; h.ncomp = h.ncomp > 3 ? 3 : h.ncomp; }
                if (len < 6 + 3 * (int)s[5]) return -1;
                for (int c = 0; c < h.ncomp; ++c) {
                    h.id[c] = s[6 + 3 * c]; h.hs[c] = s[7 + 3 * c] >> 4; h.vs[c] = s[7 + 3 * c] & 15; h.tq[c] = s[8 + 3 * c] & 3;
                }
                have_sof = true;
                break;
            }
            case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                h.why = "not a baseline sequential Huffman JPEG (progressive / lossless / arithmetic)";
                if (len >= 6) { h.H = be16(s + 1); h.W = be16(s + 3); }
                return 0;
            case 0xDB: {  // DQT
                int o = 0;
                while (o < len) {
                    const int pq = s[o] >> 4, tq = s[o] & 15;
                    ++o;
                    if (tq > 3 || o + (pq ? 128 : 64) > len) return -1;
                    for (int k = 0; k < 64; ++k) {
                        const int v = pq ? be16(s + o + 2 * k) : s[o + k];
                        h.qt[tq][ZIGZAG_TO_NATURAL[k]] = (uint16_t)v;
                    }
                    h.qt_set[tq] = true;
                    o += pq ? 128 : 64;
                }
                break;
            }
            case 0xC4: {  // DHT
                int o = 0;
                while (o < len) {
                    if (o + 17 > len) return -1;
                    const int tc = s[o] >> 4, th = s[o] & 15;
                    int total = 0;
                    for (int l = 1; l <= 16; ++l) total += s[o + l];
                    if (total > 256 || o + 17 + total > len || tc > 1) return -1;
                    if (th > 1) { h.why = "Huffman table id above 1"; o += 17 + total; continue; }
                    HuffSpec& t = tc ? h.ac[th] : h.dc[th];
                    t.bits[0] = 0;
                    for (int l = 1; l <= 16; ++l) t.bits[l] = s[o + l];
                    memcpy(t.vals, s + o + 17, total);
                    t.nvals = total;
                    t.set = true;
                    o += 17 + total;
                }
                break;
            }
            case 0xDD:
                if (len < 2) return -1;
                h.restart_interval = be16(s);
                break;
            case 0xE0:
                if (len >= 5 && !memcmp(s, "JFIF\0", 5)) h.saw_jfif = true;
                break;
            case 0xE1:  // APP1: cv2.imread (3.4) applies the EXIF orientation tag; the kernels do not rotate, so a file
                        // that asks for it goes to the caller's host decoder (which does: meterelf_amd/_image.py)
                if (len >= 14 && !memcmp(s, "Exif\0\0", 6) && exif_orientation(s + 6, len - 6) > 1)
                    h.why = "EXIF orientation other than top-left (decoded and rotated on the host)";
                break;
            case 0xEE:
                if (len >= 12 && !memcmp(s, "Adobe", 5)) { h.saw_adobe = true; h.adobe_transform = s[11]; }
                break;
            case 0xDA: {  // SOS
                if (!have_sof) return -1;
                if (len < 1) return -1;
                const int ns = s[0];
                if (len < 1 + 2 * ns + 3) return -1;
                if (ns != h.ncomp) { h.why = "more than one scan (non-interleaved components)"; return 0; }
                for (int k = 0; k < ns; ++k) {
                    const int cid = s[1 + 2 * k];
                    int c = -1;
                    for (int q = 0; q < h.ncomp; ++q) if (h.id[q] == cid) c = q;
                    if (c != k) { h.why = "scan component order differs from the frame header"; return 0; }
                    h.td[c] = s[2 + 2 * k] >> 4; h.ta[c] = s[2 + 2 * k] & 15;
                    if (h.td[c] > 1 || h.ta[c] > 1) h.why = "Huffman table id above 1";
                }
                const uint8_t* e = s + 1 + 2 * ns;
                if (e[0] != 0 || e[1] != 63 || e[2] != 0) { h.why = "not a full sequential scan"; return 0; }
                h.scan_begin = i + L;
                goto done;
            }
            default: break;
        }
        i += L;
    }
done:
    if (h.why) return 0;
    if (h.H <= 0 || h.W <= 0) { h.why = "empty image"; return 0; }
    if (h.ncomp == 3) {
        if (h.saw_adobe && h.adobe_transform != 1) h.why = "Adobe marker says the data are not YCbCr";
        else if (!h.saw_jfif && !h.saw_adobe && h.id[0] == 'R' && h.id[1] == 'G' && h.id[2] == 'B') h.why = "RGB component ids";
        else if (h.hs[1] != 1 || h.vs[1] != 1 || h.hs[2] != 1 || h.vs[2] != 1) h.why = "subsampled chroma with sampling factors above 1";
        else if (!((h.hs[0] == 1 && h.vs[0] == 1) || (h.hs[0] == 2 && h.vs[0] == 1) || (h.hs[0] == 2 && h.vs[0] == 2)))
            h.why = "luma sampling other than 1x1, 2x1, 2x2";
    } else {
        h.hs[0] = h.vs[0] = 1;  // a single-component scan is never interleaved: one block per MCU
    }
    for (int c = 0; c < h.ncomp && !h.why; ++c) {
        if (!h.qt_set[h.tq[c]]) h.why = "missing quantisation table";
        else if (!h.dc[h.td[c]].set || !h.ac[h.ta[c]].set) h.why = "missing Huffman table";
    }
    return 0;
}

int jpeg_probe(const uint8_t* data, size_t size, int* H, int* W, int* supported, std::string* why)
{
    JpegHeader h;
    if (parse_headers(data, size, h) != 0) {
        *H = *W = 0; *supported = 0;
        if (why) *why = "not a JPEG file or truncated headers";
        return 0;
    }
    *H = h.H; *W = h.W; *supported = h.why ? 0 : 1;
    if (why) *why = h.why ? h.why : "";
    return 0;
}

// ------------------------------------------------------------ device-side records ----
struct JpegImageDev {
    uint32_t scan_off, scan_len;  // clean entropy-coded bytes (no stuffing, no markers) in the scan area
    uint32_t coef_blk[3];         // first coefficient block of each component (units of 64 int16)
    uint32_t plane_off[3];        // byte offset of each component's sample plane
    uint16_t blocks_x[3], blocks_y[3];
    uint16_t mcus_x, mcus_y;
    uint16_t restart_interval;
    uint8_t ncomp, hs0, vs0, ok;
    uint8_t tq[3], td[3], ta[3];
    uint8_t pad[3];
    uint32_t rst_off, rst_cnt;    // restart intervals: table of their byte offsets in the clean scan (relative to the
                                  // scan area), number of intervals
    // Filled by the host; scan_len and rst_cnt (and the bytes they describe) by k_jpeg_clean on the GPU (round 4):
    uint32_t scan_cap;            // bytes of the scan area reserved for this file's clean scan (zero-filled behind scan_len)
    uint32_t raw_off, raw_len;    // the file's entropy-coded segment as it is in the file (stuffing, fill bytes, RSTn, EOI and
                                  // whatever follows), in the raw area
};
static_assert(sizeof(JpegImageDev) % 4 == 0, "record must stay dword aligned");

struct HuffSlow {  // canonical decode data (JPEG spec F.2.2.3) for code lengths 1..16
    // limit[l - 1]: the canonical code counter after the codes of length l (one past the largest code of
    // length l, carried over from shorter lengths if there is none).  A bit window's l-bit prefix is
    // below limit[l - 1] exactly for l >= the code's length, so the length is 1 + the number of l with
    // prefix >= limit[l - 1]: independent compares, no data-dependent loop.
    uint32_t limit[16];
    int32_t valoff[16];  // huffval index of the first code of length l, minus that code
    uint8_t huffval[256];
};
constexpr int SLOW_DW = 96;
static_assert(sizeof(HuffSlow) == SLOW_DW * 4, "HuffSlow is copied to LDS as dwords");

static void build_huff(const HuffSpec& t, HuffSlow* slow)
{
    memset(slow, 0, sizeof(*slow));
    memcpy(slow->huffval, t.vals, t.nvals);
    int code = 0, p = 0;
    for (int l = 1; l <= 16; ++l) {
        slow->valoff[l - 1] = p - code;
        p += t.bits[l];
        code += t.bits[l];
        slow->limit[l - 1] = (uint32_t)code;
        code <<= 1;
    }
}

// ------------------------------------------------------------ J0: scan cleaning ----
// The entropy-coded segment without byte stuffing (FF 00 -> FF), fill bytes and RSTn markers, up to the first other marker
// (normally EOI) -- on the GPU since round 4 (rounds 1-3: a host pass, clean_scan): the host no longer touches the entropy-coded bytes except to copy them (or, for
// the file-name entry points, not at all: the files are read straight into the pinned buffer the upload starts from).
// One workgroup per file, 16 KiB per round, 16 consecutive bytes per thread.  In entropy-coded data every FF is special and
// the byte behind it says how: 00 = a stuffed FF (keep the FF, drop the 00), FF = fill (drop this one, look at the next),
// D0..D7 = RSTn (drop both, the next restart interval begins at the clean offset reached), anything else -- or the end of
// the data -- ends the scan.  Per round: (1) the first scan-ending FF of the round (minimum over the workgroup), (2) keep
// flags and restart markers below it, one packed prefix scan of both counts, (3) byte stores of the kept bytes at their
// clean offsets, restart offsets into the table behind the scan.  Leaves scan_len / rst_cnt in the file's record and the
// rest of the file's scan region zeroed, exactly what the host pass left.
constexpr int CLEAN_T = 1024;   // threads of k_jpeg_clean: 16 KiB per round, a camera frame's scan in two or three
// 0x80 in every byte of x that is zero (exact: no borrow between bytes), and such flags gathered into a nibble
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }
__device__ __forceinline__ uint32_t flag_nibble(uint32_t f)
{
    const uint32_t y = f >> 7;   // bits 0, 8, 16, 24
    return (y | (y >> 7) | (y >> 14) | (y >> 21)) & 15u;
}
__global__ __launch_bounds__(CLEAN_T) void k_jpeg_clean(JpegImageDev* __restrict__ imgs, const uint8_t* __restrict__ raw,
                                                        uint8_t* __restrict__ scan)
{
    __shared__ int s_end;
    __shared__ uint32_t s_wsum[CLEAN_T / 64];
    JpegImageDev* R = imgs + blockIdx.x;
    if (!R->ok) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint8_t* __restrict__ s = raw + R->raw_off;
    const int n = (int)R->raw_len;
    uint8_t* __restrict__ out = scan + R->scan_off;
    const bool has_rst = R->restart_interval != 0;
    uint32_t* __restrict__ rst = (uint32_t*)(scan + R->rst_off);
    const int rst_cap = has_rst ? (int)R->rst_cnt + 1 : 0;   // the host left the EXPECTED number of intervals here
    int o_base = 0, r_base = 0;   // clean bytes written, restart markers seen, before this round
    bool ended = false;
    for (int base = 0; base < n && !ended; base += CLEAN_T * 16) {
        if (tid == 0) s_end = INT_MAX;
        const int my = base + tid * 16;
        // bytes my - 1 .. my + 16 (the raw area is padded: reads past n stay inside it; their values are masked below)
        uint32_t w[5] = {0, 0, 0, 0, 0};
        uint32_t prev = 0;
        if (my < n) {
#pragma unroll
            for (int q = 0; q < 5; ++q) __builtin_memcpy(&w[q], s + my + 4 * q, 4);
            if (my > 0) prev = s[my - 1];
        }
        // One bit per byte 0 .. 16 of the thread's window: is it FF (F), 00 (Z), D0..D7 (D)?  The whole classification then
        // is a handful of 17-bit mask operations (the per-byte form of this loop was 2 000 instructions per round).
        uint32_t F = 0, Z = 0, D = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            F |= flag_nibble(zero_bytes(~w[q])) << (4 * q);
            Z |= flag_nibble(zero_bytes(w[q])) << (4 * q);
            D |= flag_nibble(zero_bytes((w[q] & 0xF8F8F8F8u) ^ 0xD0D0D0D0u)) << (4 * q);
        }
        {
            const uint32_t b16 = w[4] & 255u;
            F |= (uint32_t)(b16 == 0xFFu) << 16;
            Z |= (uint32_t)(b16 == 0x00u) << 16;
            D |= (uint32_t)((b16 & 0xF8u) == 0xD0u) << 16;
        }
        const int left = n - my;                                                         // bytes of the data from `my` on (may be <= 0)
        const uint32_t Vm = left >= 16 ? 0xFFFFu : (left > 0 ? (1u << left) - 1u : 0u);   // byte j exists
        const uint32_t NV = left >= 17 ? 0xFFFFu : (left > 1 ? (1u << (left - 1)) - 1u : 0u);   // byte j + 1 exists
        const uint32_t Fn = F >> 1, Zn = Z >> 1, Dn = D >> 1;
        const uint32_t Fp = ((F << 1) | (uint32_t)(prev == 0xFFu)) & 0xFFFFu;
        const uint32_t F16 = F & 0xFFFFu;
        // an FF that ends the scan: no byte behind it, or one that is neither 00 nor FF nor D0..D7
        const uint32_t endmask = F16 & Vm & ~((Zn | Fn | Dn) & NV);
        __syncthreads();
        if (endmask) atomicMin(&s_end, my + (int)__builtin_ctz(endmask));
        __syncthreads();
        const int end = min(s_end, n);
        const int live = end - my;
        const uint32_t Lm = live >= 16 ? 0xFFFFu : (live > 0 ? (1u << live) - 1u : 0u);
        // kept: a stuffed FF (FF 00), and every other byte that is not the second byte of a stuffed pair or of a marker
        const uint32_t keep = Lm & ((F16 & Zn & NV) | (~F16 & ~(Fp & (Z | D)) & 0xFFFFu));
        const uint32_t mark = Lm & F16 & Dn & NV;
        // exclusive prefix of (kept bytes | markers << 16) over the workgroup
        const uint32_t cnt = (uint32_t)__popc(keep);
        const uint32_t mine = cnt | ((uint32_t)__popc(mark) << 16);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        uint32_t before = incl - mine, total = 0;
#pragma unroll
        for (int q = 0; q < CLEAN_T / 64; ++q) {
            const uint32_t t = s_wsum[q];
            if (q < wv) before += t;
            total += t;
        }
        const int ko = o_base + (int)(before & 0xffffu);
        // restart markers (rare: only files with restart intervals have any): the interval behind marker k begins at the
        // clean offset reached where the marker stood
        if (__builtin_amdgcn_ballot_w64(mark != 0u) != 0ull) {
            int ro = r_base + (int)(before >> 16);
            uint32_t m = mark;
            while (m) {
                const int j = __builtin_ctz(m);
                m &= m - 1u;
                ++ro;
                if (ro < rst_cap) rst[ro] = (uint32_t)(ko + __popc(keep & ((1u << j) - 1u)));
            }
        }
        // the kept bytes moved together: a dropped byte below the highest kept one is squeezed out of the 16-byte value, highest
        // first (typically none or one per thread; the loop runs as often as the wave's worst lane needs)
        uint32_t v0 = w[0], v1 = w[1], v2 = w[2], v3 = w[3];
        uint32_t drop = keep ? (~keep & ((2u << (31 - __builtin_clz(keep))) - 1u)) : 0u;
        while (__builtin_amdgcn_ballot_w64(drop != 0u) != 0ull) {
            if (drop) {
                const int q = 31 - __builtin_clz(drop);   // the byte to squeeze out
                drop &= ~(1u << q);
                const int dq = q >> 2;
                const uint32_t lm = (1u << (8 * (q & 3))) - 1u;   // bytes of dword dq below it
                const uint32_t s0 = __builtin_amdgcn_alignbit(v1, v0, 8), s1 = __builtin_amdgcn_alignbit(v2, v1, 8),
                               s2 = __builtin_amdgcn_alignbit(v3, v2, 8), s3 = v3 >> 8;
                v0 = dq == 0 ? ((v0 & lm) | (s0 & ~lm)) : v0;
                v1 = dq == 1 ? ((v1 & lm) | (s1 & ~lm)) : (dq < 1 ? s1 : v1);
                v2 = dq == 2 ? ((v2 & lm) | (s2 & ~lm)) : (dq < 2 ? s2 : v2);
                v3 = dq == 3 ? ((v3 & lm) | (s3 & ~lm)) : s3;
            }
        }
        {   // cnt bytes at out + ko: whole dwords (any alignment), then up to three single bytes
            uint8_t* o = out + ko;
            if (cnt >= 4u) __builtin_memcpy(o, &v0, 4);
            if (cnt >= 8u) __builtin_memcpy(o + 4, &v1, 4);
            if (cnt >= 12u) __builtin_memcpy(o + 8, &v2, 4);
            if (cnt >= 16u) __builtin_memcpy(o + 12, &v3, 4);
            const uint32_t full = cnt >> 2;
            const uint32_t tailw = full == 0u ? v0 : (full == 1u ? v1 : (full == 2u ? v2 : v3));
            const uint32_t r = cnt & 3u;
            uint8_t* t = o + 4 * full;
            if (r >= 1u) t[0] = (uint8_t)tailw;
            if (r >= 2u) t[1] = (uint8_t)(tailw >> 8);
            if (r >= 3u) t[2] = (uint8_t)(tailw >> 16);
        }
        o_base += (int)(total & 0xffffu);
        r_base += (int)(total >> 16);
        ended = end < base + CLEAN_T * 16;   // uniform: the scan ended inside this round (or the data did)
        __syncthreads();                      // s_end and s_wsum are rewritten by the next round
    }
    const int cap = (int)R->scan_cap;
    for (int i = o_base + tid; i < cap; i += CLEAN_T) out[i] = 0;
    if (tid == 0) {
        R->scan_len = (uint32_t)o_base;
        if (has_rst) {
            rst[0] = 0;
            R->rst_cnt = (uint32_t)(r_base + 1);
        }
    }
}

// Stage entry point (include/meterelf_hip.h, melf_jpeg_clean_segment; tests/test_jpeg.py): k_jpeg_clean on ONE byte string as if it were a file's
// entropy-coded segment -- arbitrary bytes, so that the tests can feed it every FF pattern, legal or not, and compare with the
// sequential rule.  restart_expected > 0: a file with restart intervals, table of restart_expected + 1 entries.
extern "C" __attribute__((visibility("default"))) int melf_jpeg_clean_segment(const uint8_t* raw, int n, int restart_expected, uint8_t* out,
                                                                            int32_t* out_len, uint32_t* rst, int32_t* rst_cnt)
{
    if (n < 0 || !out || !out_len || (n > 0 && !raw)) return -1;
    const size_t scan_cap = ((size_t)n + 128 + 63) / 64 * 64, table = restart_expected > 0 ? ((size_t)restart_expected + 1) * 4 + 64 : 0;
    JpegImageDev rec;
    memset(&rec, 0, sizeof(rec));
    rec.ok = restart_expected > 0 ? 2 : 1;
    rec.restart_interval = restart_expected > 0 ? 1 : 0;
    rec.rst_cnt = (uint32_t)std::max(restart_expected, 0);
    rec.rst_off = (uint32_t)scan_cap;
    rec.scan_cap = (uint32_t)scan_cap;
    rec.raw_len = (uint32_t)n;
    uint8_t *d_rec = nullptr, *d_raw = nullptr, *d_scan = nullptr;
    int rc = -1;
    if (hipMalloc((void**)&d_rec, sizeof(rec)) == hipSuccess && hipMalloc((void**)&d_raw, (size_t)n + 64) == hipSuccess &&
        hipMalloc((void**)&d_scan, scan_cap + table + 64) == hipSuccess &&
        hipMemset(d_raw, 0xFF, (size_t)n + 64) == hipSuccess &&   // what lies behind the data must not matter: make it hostile
        hipMemset(d_scan, 0xEE, scan_cap + table + 64) == hipSuccess &&
        hipMemcpy(d_rec, &rec, sizeof(rec), hipMemcpyHostToDevice) == hipSuccess &&
        (n == 0 || hipMemcpy(d_raw, raw, (size_t)n, hipMemcpyHostToDevice) == hipSuccess)) {
        hipLaunchKernelGGL(k_jpeg_clean, dim3(1), dim3(CLEAN_T), 0, 0, (JpegImageDev*)d_rec, d_raw, d_scan);
        std::vector<uint8_t> back(scan_cap + table);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(&rec, d_rec, sizeof(rec), hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(back.data(), d_scan, back.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            *out_len = (int32_t)rec.scan_len;
            memcpy(out, back.data(), scan_cap);   // the caller's buffer holds n + 192 bytes: clean bytes, then the zero fill
            if (restart_expected > 0 && rst && rst_cnt) {
                *rst_cnt = (int32_t)rec.rst_cnt;
                memcpy(rst, back.data() + scan_cap, ((size_t)restart_expected + 1) * 4);
            }
            rc = 0;
        }
    }
    (void)hipFree(d_rec); (void)hipFree(d_raw); (void)hipFree(d_scan);
    return rc;
}

// ------------------------------------------------------------------ J1: Huffman ----
__device__ __constant__ uint8_t c_zz2nat[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------- J1: Huffman, segment-parallel ----
// One workgroup per image, one lane per bit-stream segment.  A Huffman stream can only be decoded from
// a known state (bit position, block within the MCU, coefficient index), but decoders started from a
// wrong state fall into step with the true one by themselves (self-synchronisation), so:
//   round 0   every lane walks its segment from a guessed state (state only: no values) and publishes its exit state;
//   round r   a lane whose predecessor's exit state changed walks again from that state (the changed segments
//             compacted onto the first lanes); the true state of lane 0 propagates at least one lane per round; the
//             fixtures need 8-19 rounds (fixed point: exit[i] = F_i(exit[i-1]) for all i);
//   scan      blocks completed per segment -> exclusive prefix over the lanes = each segment's first block;
//   output    one more decode that writes the coefficients, DC predictors starting at zero in every segment;
//   fix-up    prefix over the lanes of the segments' DC-difference sums -> the predictors at each segment's entry,
//             added to the DC coefficients the segment stored.
// The decode tables of the image sit in LDS, the stream is read through L1 a dword ahead.  What bounds the kernel is
// the number of INSTRUCTIONS per decode step (tools/ubench/wave_latency.hip: a lone wave issues one per ~6 cycles
// whatever it is, four waves on a SIMD one per ~4 between them), so both decoders are written to be short.
struct SegState {
    uint32_t p;      // bit position
    int blk, k;      // block within the MCU, coefficient index (0 = DC symbol comes next)
};
// First-level lookup of the Huffman kernels, built in LDS by the kernels themselves: four tables (DC 0, DC 1, AC 0,
// AC 1) of 2^TAB_BITS dwords.  An entry says what the next TAB_BITS bits of the stream start with -- ONE symbol, or
// TWO when the second code also lies wholly inside the window.  A symbol is the pair (bits it occupies: code +
// magnitude bits, the symbol alone fixes their number; coefficients it advances: DC 1, AC run + 1, ZRL 16, end of block
// 64 = k leaves the block), packed so that ONE addition advances the state-only decoder's packed position
// (bit position in bits 0..15, coefficient index from bit 16):
//   bits  0.. 4  bits1        bits 16..22  adv1       the first symbol
//   bits  7..11  bits2        bits 23..29  adv2       both symbols together (adv2 = 64 when the second ends the block);
//                                                      an entry WITHOUT a second symbol repeats the first here, so that
//                                                      the decoder need not ask whether there is one
//   bits 12..15  sbits        magnitude bits of the first symbol (the last sbits of its bits1)
// Bit 31 set: the first code is longer than the window.  For the AC tables the entry then carries (bits 0..11) where
// the window's 2^6 continuations start in `longtab`, a direct table over the 16-bit windows that begin with a long code
// (LONG_N of them per AC table; their Kraft sum is small: 319 for the standard tables); bit 30: not covered (a DC table,
// or a code set with more long-code space) -- canonical compares from `slow`.  A full wave meets a long code in most of
// its steps, so what it costs is paid by every step: one more lookup instead of ~40 instructions.
// The second symbol of a DC entry is the block's first AC symbol, decoded with the AC table that goes with that DC
// table; a DC table whose blocks do not all use that AC table gets no second symbols (McuLayout::pair_dc).  The
// coefficient-writing decoder reads the first symbol only; the state-only decoder of the speculative pass and of the
// synchronisation rounds takes both: a flat block (DC difference + end of block, 4-6 bits) is one step, most short AC
// symbols go two at a time.
constexpr uint32_t SYM_MASK = 0x007f001fu;  // bits | adv << 16 of the first symbol; the second sits 7 bits higher
constexpr int TAB_BITS = 10;
constexpr int LONG_N = 512;  // entries of the long-code table per AC table
// k_jpeg_huff keeps bit positions inside a segment in 16 bits: segments of at most 32 000 bits, i.e. scans of at most
// 4 MB with 1024 lanes (2 MB with the usual 512); larger baseline files without restart markers go to the host decoder
constexpr size_t JPEG_MAX_PAR_SCAN = 4000000;
constexpr uint32_t TAB_MASK = (1u << TAB_BITS) - 1u;

// Per-MCU-position tables packed into registers: 1 bit of DC table id, 1 bit of AC table id and 2 bits
// of component per block position (an MCU has at most 6 blocks here).
// The pixel window the decoder produces (x0, y0 multiples of 16): the whole frame, or -- when the caller only
// reads the meter_rect crop -- that crop plus one MCU of context for the chroma filter.  Coefficients outside it
// are neither zeroed nor stored, J2 / J3 do not run there.
struct JpegWindow {
    int x0, y0, x1, y1;
};
struct McuWindow {  // the same in MCUs of one image
    int mx0, mx1, my0, my1;
};

struct McuLayout {
    uint32_t dc_bits, ac_bits, comp_bits;
    uint32_t tsel5;      // per block, 5 bits: DC table id | (2 + AC table id) << 2 (the state-only decoder's table choice)
    uint32_t nxt5;       // per block, 5 bits: 5 * index of the block that follows it in the MCU
    uint32_t oh0, oh1, oh2;  // per block, at bit 5 * block: the block belongs to component 0 / 1 / 2
    uint32_t pair_dc;    // bit t: every block with DC table t uses the AC table its entries' second symbols come from
    int ac_of_dc0, ac_of_dc1;  // that AC table, per DC table
    int bpm, yblocks;
};

__device__ __forceinline__ McuLayout jpeg_mcu_layout(const JpegImageDev* R)
{
    const int ncomp = R->ncomp;
    McuLayout L;
    L.yblocks = ncomp == 1 ? 1 : R->hs0 * R->vs0;
    L.bpm = ncomp == 1 ? 1 : L.yblocks + 2;
    L.dc_bits = L.ac_bits = L.comp_bits = L.tsel5 = L.nxt5 = 0;
    L.oh0 = L.oh1 = L.oh2 = 0;
    L.pair_dc = 3;
    int a0 = -1, a1 = -1;
    for (int b = 0; b < L.bpm; ++b) {
        const int c = b < L.yblocks ? 0 : 1 + b - L.yblocks;
        const int td = R->td[c] & 1, ta = R->ta[c] & 1;
        L.dc_bits |= (uint32_t)td << b;
        L.ac_bits |= (uint32_t)ta << b;
        L.comp_bits |= (uint32_t)c << (2 * b);
        L.tsel5 |= ((uint32_t)td | (2u + (uint32_t)ta) << 2) << (5 * b);
        L.nxt5 |= (uint32_t)(5 * (b + 1 == L.bpm ? 0 : b + 1)) << (5 * b);
        L.oh0 |= (c == 0 ? 1u : 0u) << (5 * b);
        L.oh1 |= (c == 1 ? 1u : 0u) << (5 * b);
        L.oh2 |= (c == 2 ? 1u : 0u) << (5 * b);
        if (td == 0 && a0 < 0) a0 = ta;
        if (td == 1 && a1 < 0) a1 = ta;
        if ((td == 0 ? a0 : a1) != ta) L.pair_dc &= ~(1u << td);
    }
    L.ac_of_dc0 = a0 < 0 ? 0 : a0;
    L.ac_of_dc1 = a1 < 0 ? 0 : a1;
    return L;
}

// the entry of a single symbol (the second-symbol fields repeat it)
__device__ __forceinline__ uint32_t huff_step(const bool dc, const int len, const int sym)
{
    const int sbits = sym & 15, run = sym >> 4;
    const int adv = dc ? 1 : (sbits ? run + 1 : (run == 15 ? 16 : 64));
    const uint32_t one = (uint32_t)(len + sbits) | (uint32_t)adv << 16;
    return one | one << 7 | (uint32_t)sbits << 12;
}

// Canonical decode of the code at the top of the nb-bit window x, looking at its first `avail` bits only: the symbol's
// step, or 0 when the code is longer than that.  sl: the table's HuffSlow in LDS.
template <int NB>
__device__ __forceinline__ uint32_t huff_window_step(const uint32_t* __restrict__ sl, const bool dc, const uint32_t x, const int avail)
{
    int len = 1;
#pragma unroll
    for (int l = 1; l <= NB; ++l) len += (l <= avail && (x >> (NB - l)) >= sl[l - 1]) ? 1 : 0;
    if (len > avail) return 0;
    const int idx = (int)sl[16 + len - 1] + (int)(x >> (NB - len));
    return huff_step(dc, len, ((const uint8_t*)(sl + 32))[idx & 255]);
}

__device__ __forceinline__ uint32_t huff_long_step(const uint32_t* __restrict__ sl, const bool dc, const uint32_t w, bool& invalid);

// The four tables and the long-code table, by all T threads of the workgroup (slow[] complete; a barrier must follow).
template <int T>
__device__ __forceinline__ void jpeg_build_tables(uint32_t* __restrict__ tab, uint32_t* __restrict__ longtab, const uint32_t* __restrict__ slow,
                                                  const McuLayout L, const int tid)
{
    for (int i = tid; i < (4 << TAB_BITS); i += T) {
        const int t = i >> TAB_BITS;
        const bool dc = t < 2;
        const uint32_t x = (uint32_t)i & TAB_MASK;
        uint32_t e = huff_window_step<TAB_BITS>(slow + t * SLOW_DW, dc, x, TAB_BITS);
        const int bits1 = (int)(e & 31u), adv1 = (int)((e >> 16) & 127u);
        if (e && adv1 != 64 && bits1 < TAB_BITS && (!dc || ((L.pair_dc >> t) & 1u))) {  // a second code may lie wholly inside the window
            const int t2 = dc ? 2 + (t == 0 ? L.ac_of_dc0 : L.ac_of_dc1) : t;
            const uint32_t e2 = huff_window_step<TAB_BITS>(slow + t2 * SLOW_DW, false, (x << bits1) & TAB_MASK, TAB_BITS - bits1);
            if (e2) {
                const int a2 = (int)((e2 >> 16) & 127u);
                const uint32_t both = (uint32_t)(bits1 + (int)(e2 & 31u)) | (uint32_t)(a2 == 64 ? 64 : adv1 + a2) << 16;
                e = (e & ~(SYM_MASK << 7)) | both << 7;
            }
        }
        if (!e) {  // a long code: where its 16-bit windows start in longtab, if they are there
            const uint32_t first = slow[t * SLOW_DW + TAB_BITS - 1] << (16 - TAB_BITS);  // the first 16-bit window with a long code
            const uint32_t off = (x << (16 - TAB_BITS)) - first;
            e = (!dc && off + (1u << (16 - TAB_BITS)) <= (uint32_t)LONG_N) ? 0x80000000u | ((uint32_t)(t - 2) * LONG_N + off) : 0xC0000000u;
        }
        tab[i] = e;
    }
    for (int i = tid; i < 2 * LONG_N; i += T) {
        const uint32_t* sl = slow + (2 + i / LONG_N) * SLOW_DW;
        const uint32_t w16 = (sl[TAB_BITS - 1] << (16 - TAB_BITS)) + (uint32_t)(i % LONG_N);
        uint32_t e = 0;
        if (w16 <= 0xffffu) {
            bool invalid = false;
            e = huff_long_step(sl, false, w16 << 16, invalid);
            if (invalid) e |= 0x80000000u;
        }
        longtab[i] = e;
    }
}

// The first symbol when its code is longer than the window: the canonical compares for the remaining lengths.
// A window that starts with no code at all (only a speculative decode or a corrupt file gets there) counts as a
// 16-bit end of block / zero DC difference and sets `invalid`.
__device__ __forceinline__ uint32_t huff_long_step(const uint32_t* __restrict__ sl, const bool dc, const uint32_t w, bool& invalid)
{
    const uint32_t code16 = w >> 16;
    int len = TAB_BITS + 1, sym = 0;
#pragma unroll
    for (int l = TAB_BITS + 1; l <= 16; ++l) len += (code16 >> (16 - l)) >= sl[l - 1] ? 1 : 0;
    if (len > 16) {
        invalid = true;
        len = 16;
    } else {
        const int idx = (int)sl[16 + len - 1] + (int)(code16 >> (16 - len));
        sym = ((const uint8_t*)(sl + 32))[idx & 255];
    }
    return huff_step(dc, len, sym);
}

// An entry with bit 31 set: the first symbol's code is longer than the window (w: the next 32 bits).  Returns the
// symbol's entry; bit 31 of the result: no code at all (see huff_long_step).
__device__ __forceinline__ uint32_t huff_long_entry(const uint32_t e, const uint32_t* __restrict__ longtab, const uint32_t* __restrict__ slow,
                                                    const uint32_t t, const bool dc, const uint32_t w)
{
    if (e & 0x40000000u) {
        bool invalid = false;
        const uint32_t r = huff_long_step(slow + t * SLOW_DW, dc, w, invalid);
        return invalid ? r | 0x80000000u : r;
    }
    return longtab[(e & 0xfffu) + ((w >> 16) & ((1u << (16 - TAB_BITS)) - 1u))];
}

// The coefficient blocks of an image are stored in DECODE ORDER (block n of the scan at n * 64 coefficients: MCU after
// MCU, the MCU's blocks in scan order), so that the decoders' "next block" is the next 128 bytes and the IDCT kernel does
// the little arithmetic of finding a plane position's block.  Where block n of MCU (mx, my) lives, or NULL outside the
// window (such blocks are decoded -- the DC predictors need them -- but not stored):
struct CoefBlocks {
    int16_t* base;
    McuWindow mw;
};
__device__ __forceinline__ int16_t* coef_block_ptr(const CoefBlocks& cp, const int mx, const int my, const int n)
{
    const bool inside = (unsigned)(mx - cp.mw.mx0) < (unsigned)(cp.mw.mx1 - cp.mw.mx0) && (unsigned)(my - cp.mw.my0) < (unsigned)(cp.mw.my1 - cp.mw.my0);
    return inside ? cp.base + (size_t)n * 64 : nullptr;
}

// One segment, symbol by symbol, writing the coefficients of the blocks inside the window.  pred0..2: the DC predictors
// at the segment's entry (in), at its exit (out); ndc: DC symbols decoded.  Every lane of a full wave is at another
// place of its block, so whatever any symbol kind needs is executed in every step: the loop is written for the UNION --
// one value extraction, one store (DC: predictor sum at index 0; AC: the value at its natural position), predictor sums
// by multiply-add with a one-hot component vector that changes with the block, the stream window of the state-only
// decoder -- and only the block change (next block's pointer, tables, component) sits behind a branch.
__device__ __forceinline__ void jpeg_decode_segment(
    const uint32_t* __restrict__ W, const uint32_t* __restrict__ tab, const uint32_t* __restrict__ longtab, const uint32_t* __restrict__ slow,
    const uint8_t* __restrict__ nat, const McuLayout L, SegState& s, const uint32_t p_end, int& nblk,
    int nb, const int total_blocks, int& pred0, int& pred1, int& pred2, int& ndc, const int mcus_x, const CoefBlocks& cp, int& bad)
{
    uint32_t p = s.p;
    int blk = s.blk, k = s.k;
    int mx, my;
    {
        const int mcu = nb / L.bpm;
        my = mcu / mcus_x;
        mx = mcu - my * mcus_x;
    }
    int16_t* cb = coef_block_ptr(cp, mx, my, nb);
    int nbl = 0, nd = 0;
    int a0 = pred0, a1 = pred1, a2 = pred2;
    uint32_t bs = 5u * (uint32_t)blk;
    int c0 = (int)__builtin_amdgcn_ubfe(L.oh0, bs, 1), c1 = (int)__builtin_amdgcn_ubfe(L.oh1, bs, 1), c2 = (int)__builtin_amdgcn_ubfe(L.oh2, bs, 1);
    // the stream: three byte-swapped dwords that rotate when the position crosses a dword, the next one requested a step ahead
    const uint32_t* __restrict__ Wp = W + (p >> 5);
    uint32_t Xs = p << 27;
    uint32_t d0 = __builtin_bswap32(Wp[0]), d1 = __builtin_bswap32(Wp[1]), d2 = __builtin_bswap32(Wp[2]);
    uint32_t off = 3;
    uint32_t nraw = Wp[3];
    uint32_t t = __builtin_amdgcn_ubfe(L.tsel5, bs + (k == 0 ? 0u : 2u), 2);
    uint32_t b0 = (uint32_t)(((((uint64_t)d0 << 32) | d1) << (Xs >> 27)) >> 32);
    while (p < p_end && nb < total_blocks) {
        uint32_t e = tab[__builtin_amdgcn_alignbit(t, b0, 32 - TAB_BITS)];
        if (__builtin_amdgcn_sicmp((int32_t)e, 0, 40 /* < */) != 0ull) {  // some lane met a long code
            asm volatile("");
            if ((int32_t)e < 0) {
                e = huff_long_entry(e, longtab, slow, t, k == 0, b0);
                if ((int32_t)e < 0) bad = 1;
            }
        }
        const int used = (int)(e & 31u), adv = (int)((e >> 16) & 127u), sbits = (int)((e >> 12) & 15u);
        const int raw = (int)__builtin_amdgcn_ubfe(b0, 32 - used, sbits);  // width 0 -> 0
        const int m = 1 << sbits;
        const int v = 2 * raw < m ? raw - m + 1 : raw;  // EXTEND (F.2.2.1); sbits = 0 -> 0
        const bool isdc = k == 0;
        const int vd = isdc ? v : 0;
        a0 = __mul24(vd, c0) + a0;
        a1 = __mul24(vd, c1) + a1;
        a2 = __mul24(vd, c2) + a2;
        const int pr = c0 ? a0 : c1 ? a1 : a2;
        nd += isdc ? 1 : 0;
        k += adv;  // DC: 0 -> 1
        const int pos = k - 1;  // run zeros, then this coefficient
        const int zz = nat[pos & 63];
        const bool st = cb != nullptr && (isdc || (sbits != 0 && pos < 64));
        if (st) cb[isdc ? 0 : zz] = (int16_t)(isdc ? pr : v);
        p += (uint32_t)used;
        const bool rot = __builtin_uadd_overflow(Xs, (uint32_t)used << 27, &Xs);
        d0 = rot ? d1 : d0;
        d1 = rot ? d2 : d1;
        d2 = rot ? __builtin_bswap32(nraw) : d2;
        off += rot ? 1u : 0u;
        // only the lanes that moved on ask for their next dword: with eight full waves per image a load in EVERY step kept
        // the stream's lines the most recently used ones of the L2 and pushed the coefficient lines out between two
        // stores to the same block (WRITE_SIZE 449 -> 290 MB per 512 files, same kernel time)
        if (rot) nraw = Wp[off];
        __builtin_amdgcn_sched_barrier(0);
        if (k >= 64) {
            k = 0;
            ++nbl;
            ++nb;
            if (++blk == L.bpm) {
                blk = 0;
                if (++mx == mcus_x) { mx = 0; ++my; }
            }
            bs = 5u * (uint32_t)blk;
            cb = coef_block_ptr(cp, mx, my, nb);
            c0 = (int)__builtin_amdgcn_ubfe(L.oh0, bs, 1);
            c1 = (int)__builtin_amdgcn_ubfe(L.oh1, bs, 1);
            c2 = (int)__builtin_amdgcn_ubfe(L.oh2, bs, 1);
        }
        t = __builtin_amdgcn_ubfe(L.tsel5, bs + (k == 0 ? 0u : 2u), 2);
        b0 = (uint32_t)(((((uint64_t)d0 << 32) | d1) << (Xs >> 27)) >> 32);
    }
    s.p = p; s.blk = blk; s.k = k;
    pred0 = a0; pred1 = a1; pred2 = a2;
    nblk = nbl;
    ndc = nd;
}

// What a segment's decode does to the decoder STATE, without coefficients or DC values: exit state and blocks completed.
// The same walk as jpeg_decode_segment (same tables, same treatment of impossible codes, same stop at p_end; the second
// symbol of an entry is taken only where the one-at-a-time decoder would decode it next: same block, before p_end), so a
// segment entered in the true state leaves in the true state.
// The synchronisation rounds are a chain of dependent steps of ONE wave, and a lone wave pays ~6 cycles per instruction
// whatever it is, ~20 more per VALU -> SALU -> VALU hop, 20-55 per branch (tools/ubench/wave_latency.hip): what counts
// is the number of instructions per step.  Hence: no data-dependent branches but the rare long code; bit position
// (relative to `base`, 16 bits) and coefficient index packed into one register X that an entry's symbol advances with one
// addition; both conditions for the second symbol from one packed 16-bit subtraction; the stream as three byte-swapped
// dwords d0 d1 d2 that rotate when the position crosses a dword, the dword after them requested a step ahead; table
// id and next block from 5-bit-per-block lookup words in scalar registers.  No DC values here: the output pass sums
// them per segment and a fix-up adds the predictors (k_jpeg_huff).
typedef short melf_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void jpeg_state_segment(
    const uint32_t* __restrict__ W, const uint32_t* __restrict__ tab, const uint32_t* __restrict__ longtab, const uint32_t* __restrict__ slow, const McuLayout L,
    SegState& s, const uint32_t p_end, int& nblk, int* diag = nullptr)
{
    const uint32_t base = s.p & ~31u;
    const uint32_t* __restrict__ Wp = W + (base >> 5);
    uint32_t X = (s.p - base) | (uint32_t)s.k << 16;           // position | coefficient index << 16
    uint32_t Xs = X << 27;                                      // the position's low five bits on top: their carry = "next dword"
    const uint32_t C = (p_end - base) | 64u << 16;             // their limits (p_end - base < 32768: checked by the kernel)
    uint32_t bs = 5u * (uint32_t)s.blk;
    int nb = 0;
    uint32_t d0 = __builtin_bswap32(Wp[0]), d1 = __builtin_bswap32(Wp[1]), d2 = __builtin_bswap32(Wp[2]);
    uint32_t off = 3;
    uint32_t nraw = Wp[3];
    uint32_t t = __builtin_amdgcn_ubfe(L.tsel5, bs + (s.k == 0 ? 0u : 2u), 2);
    uint32_t b0 = (uint32_t)(((((uint64_t)d0 << 32) | d1) << (Xs >> 27)) >> 32);
    while ((X & 0xffffu) < (C & 0xffffu)) {
        uint32_t e = tab[__builtin_amdgcn_alignbit(t, b0, 32 - TAB_BITS)];  // (t << TAB_BITS) | (b0 >> (32 - TAB_BITS))
#ifdef MELF_JPEG_ROUNDS
        if (diag) ++diag[0];
#endif
        if (__builtin_amdgcn_sicmp((int32_t)e, 0, 40 /* < */) != 0ull) {  // some lane met a long code (one compare + scalar branch)
            asm volatile("");  // keeps the compiler from folding the two conditions into one divergent branch
            if ((int32_t)e < 0) e = huff_long_entry(e, longtab, slow, t, (X >> 16) == 0u, b0);
        }
        const uint32_t sym1 = e & SYM_MASK, sym2 = (e >> 7) & SYM_MASK;
        const uint32_t Y = X + sym1;  // after the first symbol
        const melf_s16x2 z = __builtin_bit_cast(melf_s16x2, Y) - __builtin_bit_cast(melf_s16x2, C);
        // both still inside: position < p_end and coefficient index < 64 -> the second symbol is the decoder's next
        const bool two = (__builtin_bit_cast(uint32_t, z) & 0x80008000u) == 0x80008000u;
#ifdef MELF_JPEG_ROUNDS
        if (diag && two && sym1 != sym2) ++diag[1];
#endif
        const uint32_t sel = two ? sym2 : sym1;
        uint32_t Xn = X + sel;
        const bool rot = __builtin_uadd_overflow(Xs, sel << 27, &Xs);  // crossed into the next dword (a step consumes < 32 bits)
        d0 = rot ? d1 : d0;
        d1 = rot ? d2 : d1;
        d2 = rot ? __builtin_bswap32(nraw) : d2;
        off += rot ? 1u : 0u;
        nraw = Wp[off];  // requested here, consumed a whole step later (an L1 hit takes most of a step)
        __builtin_amdgcn_sched_barrier(0);
        const bool end = Xn >= (64u << 16);       // the block is complete
        Xn = end ? (Xn & 0xffffu) : Xn;
        nb += end ? 1 : 0;
        const uint32_t nxt = __builtin_amdgcn_ubfe(L.nxt5, bs, 5);
        bs = end ? nxt : bs;
        t = __builtin_amdgcn_ubfe(L.tsel5, bs + (end ? 0u : 2u), 2);
        X = Xn;
        b0 = (uint32_t)(((((uint64_t)d0 << 32) | d1) << (Xs >> 27)) >> 32);
    }
    s.p = base + (X & 0xffffu);
    s.k = (int)(X >> 16);
    s.blk = (int)((bs * 13u) >> 6);  // bs / 5 for bs <= 25
    nblk = nb;
}

// the three 16-bit lanes of a packed DC-difference sum (each true sum fits: it is a difference of two DC values)
__device__ __forceinline__ void unpack_dsum(int64_t d, int& a, int& b, int& c)
{
    a = (int)(int16_t)(d & 0xffff);
    d = (d - a) >> 16;
    b = (int)(int16_t)(d & 0xffff);
    d = (d - b) >> 16;
    c = (int)(int16_t)(d & 0xffff);
}

#ifdef MELF_JPEG_ROUNDS
__device__ uint64_t g_jpeg_stamps[8 * 8192];  // per image: start, tables built, round 0 done, rounds done, scan done, output pass done, end
extern "C" __attribute__((visibility("default"))) int melf_debug_jpeg_stamps(uint64_t* out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_jpeg_stamps), sizeof(uint64_t) * 8 * (size_t)(n < 8192 ? n : 8192)) == hipSuccess ? 0 : -1;
}
#define JSTAMP(k) do { if (tid == 0 && img < 8192) g_jpeg_stamps[8 * img + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ uint32_t g_jpeg_rounds[8192];
// images 0..7, rounds 1..32: segments decoded again, cycles, loop iterations of the slowest lane, of all lanes, two-symbol steps
__device__ uint32_t g_jpeg_round_log[8 * 32 * 5];
extern "C" __attribute__((visibility("default"))) int melf_debug_jpeg_round_log(uint32_t* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_jpeg_round_log), sizeof(g_jpeg_round_log)) == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int melf_debug_jpeg_rounds(uint32_t* out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_jpeg_rounds), sizeof(uint32_t) * (size_t)(n < 8192 ? n : 8192)) == hipSuccess ? 0 : -1;
}
#endif

// Zeroes the coefficient blocks of the image's window (the rest of the buffer is never read) and returns the
// window in MCUs.  Called by all threads of the image's workgroup; a __syncthreads() must follow before stores.
template <int T>
__device__ __forceinline__ McuWindow jpeg_zero_window(const JpegImageDev* R, const JpegWindow win, int16_t* __restrict__ coefs, int tid)
{
    const int hs0 = R->hs0, vs0 = R->vs0, ncomp = R->ncomp;
    const int mwid = 8 * hs0, mhei = 8 * vs0;
    McuWindow m;
    m.mx0 = win.x0 / mwid; m.mx1 = min((win.x1 + mwid - 1) / mwid, (int)R->mcus_x);
    m.my0 = win.y0 / mhei; m.my1 = min((win.y1 + mhei - 1) / mhei, (int)R->mcus_y);
    // decode order: the window's MCUs of one MCU row are one run of bpm * 128 bytes each
    const int bpm = ncomp == 1 ? 1 : hs0 * vs0 + 2;
    const int run16 = (m.mx1 - m.mx0) * bpm * 8, rows = m.my1 - m.my0;  // uint4s per row of MCUs
    uint4* base = (uint4*)(coefs + (size_t)R->coef_blk[0] * 64);
    for (int i = tid; i < rows * run16; i += T) {
        const int r = i / run16, q = i - r * run16;
        base[((size_t)(m.my0 + r) * R->mcus_x + m.mx0) * bpm * 8 + q] = make_uint4(0u, 0u, 0u, 0u);
    }
    return m;
}

#ifndef MELF_JPEG_ROUNDS
#define JSTAMP(k) do { } while (0)
#endif

// At most 80 SGPRs per wave, VCC and the other implicit ones included: one more 16-register granule and a SIMD holds
// 7 waves instead of 8, i.e. three workgroups of 512 per CU instead of four and a 1024-image batch in two passes
// (measured in round 2: 73 -> 75 numbered SGPRs took the kernel from 1.07 to 1.47 ms with the same cycles per workgroup).
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_num_sgpr(80))) void k_jpeg_huff(const JpegImageDev* __restrict__ imgs, const HuffSlow* __restrict__ g_slow,
                                                 const uint8_t* __restrict__ scan, int16_t* __restrict__ coefs,
                                                 int32_t* __restrict__ status, JpegWindow win, int limit_to_window)
{
    __shared__ uint32_t tab[4 << TAB_BITS];
    __shared__ uint32_t longtab[2 * LONG_N];
    __shared__ uint32_t slow[4 * SLOW_DW];
    __shared__ uint8_t nat[64];
    __shared__ uint32_t e_p[T], e_s[T];  // exit state of each segment: bit position, blk << 8 | k
    __shared__ uint32_t n_p[T], n_s[T];  // entry state its last decode started from
    __shared__ int sc_n[T];              // blocks completed in the segment (then: prefix sums)
    __shared__ int64_t sc_d[T];          // packed DC-difference sums of the output pass (then: prefix sums)
    __shared__ uint16_t todo[T];         // segments to decode again this round, compacted
    __shared__ int wcount[T / 64];
#ifdef MELF_JPEG_ROUNDS
    __shared__ int s_diag[3];
#endif
    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const JpegImageDev* R = imgs + img;
    if (R->ok != 1) return;
    const uint32_t* W = (const uint32_t*)(scan + R->scan_off);  // zero-padded by 128 bytes; read through L1/L2
    JSTAMP(0);
    const McuWindow mwin = jpeg_zero_window<T>(R, win, coefs, tid);  // visible to the output pass: barriers in between
    const uint32_t scan_len = R->scan_len;
    const uint32_t bits = scan_len * 8u;
    // segment length: a multiple of 32 bits with an odd dword count, so that the lanes' stream reads fall into
    // different LDS banks
    const uint32_t S = 32u * (max(8u, (bits + 32u * T - 1) / (32u * T)) | 1u);
    const int nseg = (int)((bits + S - 1) / S);
    if (S > 32000u) {  // the state-only decoder keeps positions inside a segment in 16 bits (host: a larger T, or its own decoder)
        if (tid == 0) status[img] = 2;
        return;
    }
    {
        const uint32_t* ssrc = (const uint32_t*)(g_slow + (size_t)img * 4);
        for (int i = tid; i < 4 * SLOW_DW; i += T) slow[i] = ssrc[i];
        for (int i = tid; i < 64; i += T) nat[i] = c_zz2nat[i];
    }
    __syncthreads();
    const McuLayout L = jpeg_mcu_layout(R);
    jpeg_build_tables<T>(tab, longtab, slow, L, tid);
    const int mcus_x = R->mcus_x;
    const int total_blocks = mcus_x * (int)R->mcus_y * L.bpm;
    const CoefBlocks cp = {coefs + (size_t)R->coef_blk[0] * 64, mwin};
    __syncthreads();

    JSTAMP(1);
    // Per-segment state lives in LDS (entry used by the segment's last decode, exit state, blocks completed, DC
    // sums): in the synchronisation rounds the few segments whose entry changed are COMPACTED onto the first lanes,
    // so that a round occupies one or two waves instead of a lane here and there in all of them (a wave64
    // instruction costs the same issue slot however few lanes are active, and five workgroups share the SIMDs).
    const bool mine = tid < nseg;
    int bad = 0;
    {
        SegState ex = {(uint32_t)tid * S, 0, 0};
        int nblk = 0;
        if (mine) jpeg_state_segment(W, tab, longtab, slow, L, ex, min((uint32_t)(tid + 1) * S, bits + 32u), nblk);
        n_p[tid] = (uint32_t)tid * S;  // entry of the last decode
        n_s[tid] = 0;
        e_p[tid] = ex.p;
        e_s[tid] = (uint32_t)(ex.blk << 8 | ex.k);
        sc_n[tid] = mine ? nblk : 0;
    }
    __syncthreads();
    JSTAMP(2);
    // Window-limited convergence (round 4).  When only a window of the frame is wanted (melf_jpeg_process_batch: the
    // meter_rect crop, MCU rows 9-27 of 40 on the fixtures), nothing behind the window's last MCU row is ever read: not
    // the coefficients, not the DC predictors.  The synchronisation rounds then only have to settle the segments up to the
    // one that holds the window's end, and the output pass skips the segments behind it.  Where that segment lies is
    // estimated from the speculative pass's block counts (a wrongly-phased decoder still counts blocks almost right: a
    // block too many or too few per segment at worst, i.e. a random walk of a few dozen blocks over 512 segments) with a
    // margin of one and a half MCU rows, and CHECKED against the exact counts once the rounds are over: if the settled
    // prefix does not reach the window's end after all, the rounds go on over every segment.  What is given up: a stream
    // damaged only BEHIND the window is no longer reported corrupt (libjpeg would decode the window just the same).
    const int wend = limit_to_window ? min(total_blocks, mwin.my1 * mcus_x * L.bpm) : total_blocks;   // first block behind the window
    int cut = nseg - 1;                                               // last segment the rounds have to settle
    if (wend < total_blocks) {
        const int mine_n = sc_n[tid];
        __syncthreads();
        for (int off = 1; off < T; off <<= 1) {
            const int a = tid >= off ? sc_n[tid - off] : 0;
            __syncthreads();
            sc_n[tid] += a;
            __syncthreads();
        }
        // segments whose (estimated) first block lies behind the window's end + margin: not needed
        const int first_blk = sc_n[tid] - mine_n;
        const int need = mine && first_blk < wend + (3 * mcus_x * L.bpm) / 2 ? 1 : 0;
        const int nneed = __syncthreads_count(need);                  // the needed segments are a prefix: their number
        cut = max(0, min(nseg, nneed) - 1);
        sc_n[tid] = mine_n;                                            // back to per-segment counts for the rounds
        __syncthreads();
    }
    int rounds = 0, redone = 0;  // reported by the diagnostic build only
    (void)rounds;
    (void)redone;
    int nblk = 0, done_blocks = 0;
    for (;;) {   // rounds over the segments up to `cut`, then the exact block counts; once more over all segments if they fall short
    for (;;) {
        // which segments see a new entry state?
        uint32_t np = 0, ns = 0;
        if (tid > 0) { np = e_p[tid - 1]; ns = e_s[tid - 1]; }
        const bool ch = mine && tid <= cut && (np != n_p[tid] || ns != n_s[tid]);
        // compact them: position = number of changed segments before this one
        const uint64_t bal = __ballot(ch);
        const int wave = tid >> 6, lane = tid & 63;
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < T / 64; ++w) {
            const int c = wcount[w];
            before += w < wave ? c : 0;
            total += c;
        }
        if (total == 0) break;
        ++rounds;
#ifdef MELF_JPEG_ROUNDS
        const uint64_t round_t0 = __builtin_amdgcn_s_memtime();
        if (tid < 3) s_diag[tid] = 0;
#endif
        if (ch) {
            const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
            todo[pos] = (uint16_t)tid;
            n_p[tid] = np;  // the entry this segment is about to be decoded from
            n_s[tid] = ns;
        }
        __syncthreads();
        for (int q = tid; q < total; q += T) {
            const int i = todo[q];
            ++redone;
            SegState ex = {n_p[i], (int)(n_s[i] >> 8), (int)(n_s[i] & 255u)};
            int nblk;
#ifdef MELF_JPEG_ROUNDS
            int diag[2] = {0, 0};
            jpeg_state_segment(W, tab, longtab, slow, L, ex, min((uint32_t)(i + 1) * S, bits + 32u), nblk, diag);
            atomicMax(&s_diag[0], diag[0]);
            atomicAdd(&s_diag[1], diag[0]);
            atomicAdd(&s_diag[2], diag[1]);
#else
            jpeg_state_segment(W, tab, longtab, slow, L, ex, min((uint32_t)(i + 1) * S, bits + 32u), nblk);
#endif
            e_p[i] = ex.p;  // nobody reads exit states before the next barrier
            e_s[i] = (uint32_t)(ex.blk << 8 | ex.k);
            sc_n[i] = nblk;
        }
        __syncthreads();
#ifdef MELF_JPEG_ROUNDS
        if (tid == 0 && img < 8 && rounds <= 32) {
            uint32_t* lg = g_jpeg_round_log + (img * 32 + rounds - 1) * 5;
            lg[0] = (uint32_t)total;
            lg[1] = (uint32_t)(__builtin_amdgcn_s_memtime() - round_t0);
            lg[2] = (uint32_t)s_diag[0];
            lg[3] = (uint32_t)s_diag[1];
            lg[4] = (uint32_t)s_diag[2];
        }
#endif
    }
    JSTAMP(3);
    // exclusive prefix over the segments: blocks completed
    nblk = sc_n[tid];
    __syncthreads();
    for (int off = 1; off < T; off <<= 1) {
        const int a = tid >= off ? sc_n[tid - off] : 0;
        __syncthreads();
        sc_n[tid] += a;
        __syncthreads();
    }
    done_blocks = sc_n[T - 1];
    // the settled prefix must reach the window's end (it always does unless the estimate above was off by more than its margin)
    if (cut >= nseg - 1 || sc_n[cut] >= wend) break;
    __syncthreads();
    sc_n[tid] = nblk;
    cut = nseg - 1;
    __syncthreads();
    }
    const bool limited = cut < nseg - 1;
    const SegState entry = {n_p[tid], (int)(n_s[tid] >> 8), (int)(n_s[tid] & 255u)};
    const uint32_t p_end = min((uint32_t)(tid + 1) * S, bits + 32u);
    JSTAMP(4);
    const int nb_in = sc_n[tid] - (mine ? nblk : 0);
    // output pass: every segment from its true entry state and first block, DC predictors starting at ZERO -- the DC
    // coefficients it stores are sums of the segment's own differences; what they lack is known only after a prefix sum
    // over the segments of those sums (three 16-bit lanes of one 64-bit word), and a fix-up pass adds it
    const bool ran = mine && tid <= cut && nb_in < wend;   // (wend = total_blocks when the whole frame is wanted)
    int q0 = 0, q1 = 0, q2 = 0, ndc = 0;
    if (ran) {
        if (entry.blk != nb_in % L.bpm) bad = 1;  // the propagated state and the block count disagree: corrupt stream
        SegState st = entry;
        int n2;
        jpeg_decode_segment(W, tab, longtab, slow, nat, L, st, p_end, n2, nb_in, total_blocks, q0, q1, q2, ndc, mcus_x, cp, bad);
    }
    const int64_t dsum = (int64_t)q0 + ((int64_t)q1 << 16) + ((int64_t)q2 << 32);
    sc_d[tid] = dsum;
    __syncthreads();
    JSTAMP(5);
    for (int off = 1; off < T; off <<= 1) {
        const int64_t b = tid >= off ? sc_d[tid - off] : 0;
        __syncthreads();
        sc_d[tid] += b;
        __syncthreads();
    }
    if (ran && ndc > 0) {
        int pr0, pr1, pr2;
        unpack_dsum(sc_d[tid] - dsum, pr0, pr1, pr2);  // the predictors at the segment's entry
        if ((pr0 | pr1 | pr2) != 0) {
            // the blocks whose DC symbol this segment decoded: from the first block that BEGINS here
            const int b0 = nb_in + (entry.k != 0 ? 1 : 0);
            const int mcu = b0 / L.bpm;
            int blk = b0 - mcu * L.bpm, my = mcu / mcus_x, mx = mcu - (mcu / mcus_x) * mcus_x;
            // four blocks at a time: their loads are in flight together (a load that follows the lane's own store misses L1)
            for (int i = 0; i < ndc; i += 4) {
                int16_t* cb[4];
                int add[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    cb[u] = i + u < ndc ? coef_block_ptr(cp, mx, my, b0 + i + u) : nullptr;
                    const int comp = (int)((L.comp_bits >> (2 * blk)) & 3u);
                    add[u] = comp == 0 ? pr0 : comp == 1 ? pr1 : pr2;
                    if (++blk == L.bpm) {
                        blk = 0;
                        if (++mx == mcus_x) { mx = 0; ++my; }
                    }
                }
                int16_t v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = cb[u] ? cb[u][0] : (int16_t)0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (cb[u]) cb[u][0] = (int16_t)(v[u] + add[u]);
            }
        }
    }
    const int anybad = __syncthreads_or(bad);
    JSTAMP(6);
#ifdef MELF_JPEG_ROUNDS  // diagnostic build: rounds and re-decoded segments per image (tools/jpeg_rounds.py)
    {
        __shared__ int s_redone;
        if (tid == 0) s_redone = 0;
        __syncthreads();
        atomicAdd(&s_redone, redone);
        __syncthreads();
        if (tid == 0 && img < 8192) g_jpeg_rounds[img] = ((uint32_t)rounds << 16) | (uint32_t)min(s_redone * 100 / max(nseg, 1), 65535);
    }
#endif
    // the whole scan settled: it must hold every block of the frame; a window-limited decode: the settled prefix reached the
    // window's end (checked above) and the segments it decoded were consistent
    if (tid == 0) status[img] = (anybad || (!limited && done_blocks < total_blocks)) ? 2 : 0;
}

// Streams with restart intervals need no speculation: every interval starts byte aligned with the DC
// predictors at zero.  One workgroup per image, lanes stride over the intervals (the host recorded where
// each begins while stripping the markers) and write their coefficients directly.
template <int T>
__global__ __launch_bounds__(T) void k_jpeg_huff_rst(const JpegImageDev* __restrict__ imgs, const HuffSlow* __restrict__ g_slow,
                                                     const uint8_t* __restrict__ scan, int16_t* __restrict__ coefs,
                                                     int32_t* __restrict__ status, JpegWindow win)
{
    __shared__ uint32_t tab[4 << TAB_BITS];
    __shared__ uint32_t longtab[2 * LONG_N];
    __shared__ uint32_t slow[4 * SLOW_DW];
    __shared__ uint8_t nat[64];
    const int tid = threadIdx.x;
    const int img = blockIdx.x;
    const JpegImageDev* R = imgs + img;
    if (R->ok != 2) return;
    const uint32_t* W = (const uint32_t*)(scan + R->scan_off);
    const uint32_t* rst = (const uint32_t*)(scan + R->rst_off);
    const McuWindow mwin = jpeg_zero_window<T>(R, win, coefs, tid);
    {
        const uint32_t* ssrc = (const uint32_t*)(g_slow + (size_t)img * 4);
        for (int i = tid; i < 4 * SLOW_DW; i += T) slow[i] = ssrc[i];
        for (int i = tid; i < 64; i += T) nat[i] = c_zz2nat[i];
    }
    __syncthreads();
    const McuLayout L = jpeg_mcu_layout(R);
    jpeg_build_tables<T>(tab, longtab, slow, L, tid);
    const int mcus_x = R->mcus_x;
    const int total_blocks = mcus_x * (int)R->mcus_y * L.bpm;
    const int per_interval = (int)R->restart_interval * L.bpm;
    const int nint = (int)R->rst_cnt;
    const int expected = (total_blocks + per_interval - 1) / per_interval;
    const uint32_t bits = R->scan_len * 8u;
    const CoefBlocks cp = {coefs + (size_t)R->coef_blk[0] * 64, mwin};
    __syncthreads();
    int bad = nint != expected ? 1 : 0;  // markers missing or surplus: corrupt stream
    for (int it = tid; it < min(nint, expected); it += T) {
        SegState st = {rst[it] * 8u, 0, 0};
        const int nb0 = it * per_interval, nb1 = min(nb0 + per_interval, total_blocks);
        int n2, ndc, q0 = 0, q1 = 0, q2 = 0;  // every interval starts with zero predictors
        jpeg_decode_segment(W, tab, longtab, slow, nat, L, st, bits + 32u, n2, nb0, nb1, q0, q1, q2, ndc, mcus_x, cp, bad);
        if (n2 < nb1 - nb0) bad = 1;  // ran out of data before the interval's last block
    }
    const int anybad = __syncthreads_or(bad);
    if (tid == 0) status[img] = anybad ? 2 : 0;
}

// ------------------------------------------------------------------ J2: IDCT ----
// The "accurate integer" inverse DCT (libjpeg jidctint: Loeffler-Ligtenberg-Moschytz, 13-bit
// constants, 2 extra bits kept between the passes).  Every shift below is part of the result.
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ __forceinline__ void idct_1d(const int in[8], int out[8], const int shift, const bool pass1)
{
    constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633,
                  F1_501 = 12299, F1_847 = 15137, F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    (void)pass1;
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * F0_541;
    int tmp2 = z1 + z3 * (-F1_847);
    int tmp3 = z1 + z2 * F0_765;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) << 13;
    int tmp1 = (z2 - z3) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = descale(tmp10 + tmp3, shift); out[7] = descale(tmp10 - tmp3, shift);
    out[1] = descale(tmp11 + tmp2, shift); out[6] = descale(tmp11 - tmp2, shift);
    out[2] = descale(tmp12 + tmp1, shift); out[5] = descale(tmp12 - tmp1, shift);
    out[3] = descale(tmp13 + tmp0, shift); out[4] = descale(tmp13 - tmp0, shift);
}

__global__ __launch_bounds__(256) void k_jpeg_idct(const JpegImageDev* __restrict__ imgs, const uint16_t* __restrict__ g_qt,
                                                   const int16_t* __restrict__ coefs, const int32_t* __restrict__ status,
                                                   uint8_t* __restrict__ planes, JpegWindow win)
{
    const int img = blockIdx.y;
    const JpegImageDev& I = imgs[img];   // a reference: a local copy indexed by the component lives in scratch
    if (!I.ok || status[img] != 0) return;
    // blocks of the MCUs that overlap the pixel window (16-aligned, so whole MCUs in every sampling mode)
    const int mw = 8 * I.hs0, mh = 8 * I.vs0;
    const int mx0 = win.x0 / mw, mx1 = min((win.x1 + mw - 1) / mw, (int)I.mcus_x);
    const int my0 = win.y0 / mh, my1 = min((win.y1 + mh - 1) / mh, (int)I.mcus_y);
    const int w0 = (mx1 - mx0) * I.hs0, h0 = (my1 - my0) * I.vs0, wc = mx1 - mx0, hc = my1 - my0;
    const int nb0 = w0 * h0;
    const int nbc = I.ncomp == 3 ? wc * hc : 0;
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nb0 + 2 * nbc) return;
    int c = 0;
    if (j >= nb0) { j -= nb0; c = 1; if (j >= nbc) { j -= nbc; c = 2; } }
    const int ww = c == 0 ? w0 : wc;
    const int by = (c == 0 ? my0 * I.vs0 : my0) + j / ww, bx = (c == 0 ? mx0 * I.hs0 : mx0) + j % ww;
    // the block's place in decode order (CoefBlocks): its MCU, then its index among the MCU's blocks
    const int fx = c == 0 ? I.hs0 : 1, fy = c == 0 ? I.vs0 : 1, bpm = I.ncomp == 1 ? 1 : I.hs0 * I.vs0 + 2;
    const int mcu = (by / fy) * I.mcus_x + bx / fx;
    const int inb = c == 0 ? (by % fy) * I.hs0 + bx % fx : I.hs0 * I.vs0 + c - 1;
    const int16_t* src = coefs + ((size_t)I.coef_blk[0] + (size_t)mcu * bpm + inb) * 64;
    const uint16_t* q = g_qt + ((size_t)img * 4 + I.tq[c]) * 64;
    int ws[64];
    // pass 1: columns
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        int in[8], out[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) in[y] = (int)src[y * 8 + x] * (int)q[y * 8 + x];
        idct_1d(in, out, 13 - 2, true);
#pragma unroll
        for (int y = 0; y < 8; ++y) ws[y * 8 + x] = out[y];
    }
    // pass 2: rows, descale by 2^(13+2+3), level shift and clamp
    const int stride = I.blocks_x[c] * 8;
    uint8_t* dst = planes + I.plane_off[c] + (size_t)(by * 8) * stride + bx * 8;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        int out[8];
        idct_1d(ws + y * 8, out, 13 + 2 + 3, false);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            lo |= (uint32_t)min(max(out[x] + 128, 0), 255) << (8 * x);
            hi |= (uint32_t)min(max(out[x + 4] + 128, 0), 255) << (8 * x);
        }
        *(uint2*)(dst + (size_t)y * stride) = make_uint2(lo, hi);
    }
}

// ------------------------------------------------------ J3: upsample + colour ----
// "Fancy" upsampling (libjpeg jdsample): a triangle filter, 3/4 nearer sample + 1/4 further one in
// each direction, with the rounding constant alternating 8, 7 (or 1, 2) so that the bias cancels;
// at the image border the missing neighbour is the edge sample itself.
__device__ __forceinline__ int chroma_h2v2(const uint8_t* P, int stride, int cw, int ch, int x, int y)
{
    const int cy = y >> 1, cx = x >> 1;
    const int fy = (y & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0);
    const uint8_t* near = P + (size_t)cy * stride;
    const uint8_t* far = P + (size_t)fy * stride;
    if (cw <= 2) return near[cx];  // libjpeg falls back to plain replication for very narrow components
    const int t = 3 * near[cx] + far[cx];
    if (x & 1) return cx == cw - 1 ? (4 * t + 7) >> 4 : (3 * t + 3 * near[cx + 1] + far[cx + 1] + 7) >> 4;
    return cx == 0 ? (4 * t + 8) >> 4 : (3 * t + 3 * near[cx - 1] + far[cx - 1] + 8) >> 4;
}
__device__ __forceinline__ int chroma_h2v1(const uint8_t* P, int stride, int cw, int x, int y)
{
    const uint8_t* row = P + (size_t)y * stride;
    const int cx = x >> 1, v = row[cx];
    if (cw <= 2) return v;
    if (x & 1) return cx == cw - 1 ? v : (3 * v + row[cx + 1] + 2) >> 2;
    return cx == 0 ? v : (3 * v + row[cx - 1] + 1) >> 2;
}

// YCbCr -> RGB with libjpeg's 16-bit fixed-point tables (jdcolor): the rounding term sits in the
// Cb part of the green sum, right shifts are arithmetic.
__device__ __forceinline__ uint32_t ycc_to_bgr(int y, int cb, int cr)
{
    cb -= 128; cr -= 128;
    const int r = y + ((91881 * cr + 32768) >> 16);
    const int g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    const int b = y + ((116130 * cb + 32768) >> 16);
    return (uint32_t)min(max(b, 0), 255) | ((uint32_t)min(max(g, 0), 255) << 8) | ((uint32_t)min(max(r, 0), 255) << 16);
}

// 4:2:0 fast path (W % 8 == 0, even window rows): one thread per 8 output pixels of TWO rows (2 cy, 2 cy + 1) = 4 chroma
// columns of chroma row cy, which both rows weigh 3/4, plus the row above for the even and the row below for the odd
// output row: three chroma rows fetched for two output rows instead of four, and the work items of a window are packed
// into the workgroups (a 272-pixel window has 34 items per row: one row per 256-thread workgroup left 87 % of the lanes
// idle).  The border cases of the triangle filter are the general formula with the missing neighbour replaced by the
// sample itself ((4t + 8) >> 4 == (3t + t + 8) >> 4), i.e. clamped indices.
__global__ __launch_bounds__(256) void k_jpeg_color420(const JpegImageDev* __restrict__ imgs, const int32_t* __restrict__ status,
                                                       const uint8_t* __restrict__ planes, int H, int W,
                                                       uint8_t* __restrict__ frames, JpegWindow win)
{
    const int img = blockIdx.y;
    const int gx = (win.x1 - win.x0 + 7) / 8;                 // groups of 8 pixels per window row
    const int item = blockIdx.x * 256 + threadIdx.x;
    const int pr = item / gx, g = win.x0 / 8 + (item - pr * gx);
    const int y = win.y0 + 2 * pr;                            // even (win.y0 is a multiple of 16)
    if (y >= win.y1 || g * 8 >= win.x1) return;
    const bool two = y + 1 < win.y1;
    // the record and the status in one round trip (field by field, each && waited for its own load)
    const JpegImageDev rec = imgs[img];
    const int32_t st = status[img];
    const JpegImageDev* R = &rec;
    if (!(rec.ok && rec.ncomp == 3 && rec.hs0 == 2 && rec.vs0 == 2)) return;  // the generic kernel's image
    uint8_t* const orow = frames + ((size_t)img * H + y) * W * 3 + (size_t)g * 24;
    if (st != 0) {  // failed in the entropy decoder: zero frame
        uint2* z = (uint2*)orow;
        z[0] = z[1] = z[2] = make_uint2(0u, 0u);
        if (two) {
            uint2* z2 = (uint2*)(orow + (size_t)W * 3);
            z2[0] = z2[1] = z2[2] = make_uint2(0u, 0u);
        }
        return;
    }
    const int ys = R->blocks_x[0] * 8, cs = R->blocks_x[1] * 8;
    const int cw = (W + 1) >> 1, ch = (H + 1) >> 1;
    const int cy = y >> 1, fu = max(cy - 1, 0), fd = min(cy + 1, ch - 1);
    const int cx0 = g * 4;
    const int cl = max(cx0 - 1, 0), cr = min(cx0 + 4, cw - 1);
    int te[2][6], to[2][6];   // 3 * near + far for the even / the odd output row
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const uint8_t* P = planes + R->plane_off[1 + c];
        const uint8_t* near = P + (size_t)cy * cs;
        const uint8_t* up = P + (size_t)fu * cs;
        const uint8_t* dn = P + (size_t)fd * cs;
        const uint32_t n4 = *(const uint32_t*)(near + cx0), u4 = *(const uint32_t*)(up + cx0), d4 = *(const uint32_t*)(dn + cx0);
        const int nl = near[cl], nr = near[cr];
        te[c][0] = 3 * nl + up[cl]; te[c][5] = 3 * nr + up[cr];
        to[c][0] = 3 * nl + dn[cl]; to[c][5] = 3 * nr + dn[cr];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nj = 3 * (int)((n4 >> (8 * j)) & 255u);
            te[c][1 + j] = nj + (int)((u4 >> (8 * j)) & 255u);
            to[c][1 + j] = nj + (int)((d4 >> (8 * j)) & 255u);
        }
    }
    const uint8_t* yrow = planes + R->plane_off[0] + (size_t)y * ys + g * 8;
    const uint2 yy0 = *(const uint2*)yrow;
    const uint2 yy1 = two ? *(const uint2*)(yrow + ys) : yy0;
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        if (row == 1 && !two) break;
        const int (*t)[6] = row ? to : te;
        const uint2 yy = row ? yy1 : yy0;
        uint32_t px[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cbe = (3 * t[0][1 + j] + t[0][j] + 8) >> 4, cbo = (3 * t[0][1 + j] + t[0][2 + j] + 7) >> 4;
            const int cre = (3 * t[1][1 + j] + t[1][j] + 8) >> 4, cro = (3 * t[1][1 + j] + t[1][2 + j] + 7) >> 4;
            const uint32_t ysrc = j < 2 ? yy.x : yy.y;
            px[2 * j] = ycc_to_bgr((int)((ysrc >> (16 * (j & 1))) & 255u), cbe, cre);
            px[2 * j + 1] = ycc_to_bgr((int)((ysrc >> (16 * (j & 1) + 8)) & 255u), cbo, cro);
        }
        uint32_t* o = (uint32_t*)(orow + (size_t)row * W * 3);
        uint32_t d[6];
        d[0] = px[0] | (px[1] << 24); d[1] = (px[1] >> 8) | (px[2] << 16); d[2] = (px[2] >> 16) | (px[3] << 8);
        d[3] = px[4] | (px[5] << 24); d[4] = (px[5] >> 8) | (px[6] << 16); d[5] = (px[6] >> 16) | (px[7] << 8);
        *(uint2*)(o) = make_uint2(d[0], d[1]);
        *(uint2*)(o + 2) = make_uint2(d[2], d[3]);
        *(uint2*)(o + 4) = make_uint2(d[4], d[5]);
    }
}

__global__ __launch_bounds__(256) void k_jpeg_color(const JpegImageDev* __restrict__ imgs, const int32_t* __restrict__ status,
                                                    const uint8_t* __restrict__ planes, int H, int W,
                                                    uint8_t* __restrict__ frames, int fast420, JpegWindow win)
{
    const int img = blockIdx.z, y = win.y0 + blockIdx.y;
    const int x0 = win.x0 + (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= win.x1) return;
    const JpegImageDev& I = imgs[img];   // a reference: a local copy indexed by the component lives in scratch
    if (fast420 && I.ok && I.ncomp == 3 && I.hs0 == 2 && I.vs0 == 2) return;  // k_jpeg_color420 did it
    uint8_t* out = frames + ((size_t)img * H + y) * W * 3 + (size_t)x0 * 3;
    const int npx = min(4, W - x0);
    if (!I.ok || status[img] != 0) {
        for (int k = 0; k < npx * 3; ++k) out[k] = 0;
        return;
    }
    const uint8_t* Y = planes + I.plane_off[0];
    const int ys = I.blocks_x[0] * 8;
    uint32_t px[4] = {0, 0, 0, 0};
    if (I.ncomp == 1) {
        for (int k = 0; k < npx; ++k) { const uint32_t v = Y[(size_t)y * ys + x0 + k]; px[k] = v | (v << 8) | (v << 16); }
    } else {
        const uint8_t* Cb = planes + I.plane_off[1];
        const uint8_t* Cr = planes + I.plane_off[2];
        const int cs = I.blocks_x[1] * 8;
        const int cw = (W + I.hs0 - 1) / I.hs0, ch = (H + I.vs0 - 1) / I.vs0;
        for (int k = 0; k < npx; ++k) {
            const int x = x0 + k;
            int cb, cr;
            if (I.hs0 == 2 && I.vs0 == 2) { cb = chroma_h2v2(Cb, cs, cw, ch, x, y); cr = chroma_h2v2(Cr, cs, cw, ch, x, y); }
            else if (I.hs0 == 2) { cb = chroma_h2v1(Cb, cs, cw, x, y); cr = chroma_h2v1(Cr, cs, cw, x, y); }
            else { cb = Cb[(size_t)y * cs + x]; cr = Cr[(size_t)y * cs + x]; }
            px[k] = ycc_to_bgr(Y[(size_t)y * ys + x], cb, cr);
        }
    }
    if (npx == 4 && ((W * 3) & 3) == 0) {  // 12 bytes, dword aligned
        uint32_t* o = (uint32_t*)out;
        o[0] = px[0] | (px[1] << 24);
        o[1] = (px[1] >> 8) | (px[2] << 16);
        o[2] = (px[2] >> 16) | (px[3] << 8);
    } else {
        for (int k = 0; k < npx; ++k) { out[3 * k] = px[k] & 255; out[3 * k + 1] = (px[k] >> 8) & 255; out[3 * k + 2] = (px[k] >> 16) & 255; }
    }
}

// ------------------------------------------------------------------ workspace ----
struct JpegWorkspace {
    uint8_t* h_stage = nullptr;  // pinned: records and tables of one batch
    size_t h_cap = 0;
    uint8_t* d_stage = nullptr;  // the same on the device, and behind them the clean scans (written by k_jpeg_clean)
    size_t d_cap = 0;
    uint8_t* h_raw = nullptr;    // pinned: the files' entropy-coded segments as they are in the files
    size_t h_raw_cap = 0;
    uint8_t* d_raw = nullptr;
    size_t d_raw_cap = 0;
    size_t raw_total = 0;        // bytes of the raw area in use
    const uint8_t* raw_src = nullptr;  // where the upload of the raw area starts: h_raw, or the caller's own pinned buffer
    int16_t* d_coefs = nullptr;
    size_t coef_cap = 0;  // int16 elements
    uint8_t* d_planes = nullptr;
    size_t plane_cap = 0;
    int32_t* d_status = nullptr;
    size_t status_cap = 0;
    // layout of the current batch inside the stage buffers
    size_t off_imgs = 0, off_qt = 0, off_slow = 0, off_scan = 0, total = 0;   // total: what is uploaded (records + tables)
    size_t scan_bytes = 0;      // device-only scan area behind them
    size_t coef_elems = 0, plane_bytes = 0;
    int max_blocks = 0;
    int n_par = 0, n_seq = 0;   // images for the segment-parallel / the restart-interval Huffman kernel
    int n_420 = 0;              // three-component 4:2:0 images (fast colour kernel)
    size_t max_par_scan = 0;    // longest clean scan among the former (bytes)
};

void jpeg_workspace_free(JpegWorkspace* w)
{
    if (!w) return;
    if (w->h_stage) (void)hipHostFree(w->h_stage);
    if (w->d_stage) (void)hipFree(w->d_stage);
    if (w->h_raw) (void)hipHostFree(w->h_raw);
    if (w->d_raw) (void)hipFree(w->d_raw);
    if (w->d_coefs) (void)hipFree(w->d_coefs);
    if (w->d_planes) (void)hipFree(w->d_planes);
    if (w->d_status) (void)hipFree(w->d_status);
    delete w;
}

template <class T>
static hipError_t grow_dev(T** p, size_t* cap, size_t need)
{
    if (need <= *cap) return hipSuccess;
    if (*p) {
        // With several chunks (and up to three calls) in flight the kernels of an older chunk may still read this workspace:
        // the caller has only waited for the slot's UPLOAD event.  hipFree happens to synchronise, but nothing promises it
        // (melf_api.hip's grow() does the same); growth is rare.
        (void)hipDeviceSynchronize();
        (void)hipFree(*p);
    }
    *p = nullptr; *cap = 0;
    const size_t want = need + need / 4;
    hipError_t e = hipMalloc((void**)p, want * sizeof(T));
    if (e == hipSuccess) *cap = want;
    return e;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Header parse of n files in one parallel pass (a chunked caller parses the whole call's files once and prepares chunk
// by chunk): status[i] 0 = for the GPU, 1 = valid but unsupported, 2 = unreadable, 3 = other size.
struct JpegParsed {
    std::vector<JpegHeader> hdr;
    std::vector<HuffSlow> slow;   // 4 per file (DC 0, DC 1, AC 0, AC 1), when built with the headers; else empty
};
static void build_slow4(const JpegHeader& h, HuffSlow* slow)
{
    for (int t = 0; t < 2; ++t) {
        if (h.dc[t].set) build_huff(h.dc[t], slow + t); else memset(slow + t, 0, sizeof(HuffSlow));
        if (h.ac[t].set) build_huff(h.ac[t], slow + 2 + t); else memset(slow + 2 + t, 0, sizeof(HuffSlow));
    }
}
JpegParsed* jpeg_parsed_new(int n, bool with_tables)
{
    JpegParsed* p = new JpegParsed();
    p->hdr.resize(n);
    if (with_tables) p->slow.resize((size_t)n * 4);
    return p;
}
void jpeg_parsed_resize(JpegParsed* p, int n)   // keeps what it has allocated: entries are reset by jpeg_parse_one
{
    if ((int)p->hdr.size() < n) p->hdr.resize(n);
    if (!p->slow.empty() && p->slow.size() < (size_t)n * 4) p->slow.resize((size_t)n * 4);
}
void jpeg_parse_one(JpegParsed* p, int i, const uint8_t* data, size_t size, int* H, int* W, int* supported)
{
    JpegHeader& h = p->hdr[i];
    h = JpegHeader();   // the object may be a previous call's
    if (!data || parse_headers(data, size, h) != 0) { *H = *W = 0; *supported = 0; return; }
    *H = h.H; *W = h.W; *supported = h.why ? 0 : 1;
    if (!h.why && !p->slow.empty()) build_slow4(h, p->slow.data() + (size_t)i * 4);
}
JpegParsed* jpeg_parse_files(const uint8_t* const* data, const size_t* sizes, int n, int H, int W, int32_t* host_status)
{
    JpegParsed* p = new JpegParsed();
    p->hdr.resize(n);
    host_pool().run(n, [&](int i) {
        JpegHeader& h = p->hdr[i];
        if (!data[i] || parse_headers(data[i], sizes[i], h) != 0) { host_status[i] = 2; return; }
        if (h.why) { host_status[i] = 1; return; }
        if (h.H != H || h.W != W) { host_status[i] = 3; return; }
        host_status[i] = 0;
    });
    return p;
}
void jpeg_parsed_free(JpegParsed* p) { delete p; }

// Host half of a batch: parse every file (unless `parsed` holds the headers already: files first .. first + n - 1 of that
// pass, host_status filled), lay the batch out, fill the pinned stage buffer.
// host_status[i]: 0 = handed to the GPU, 1 = valid but unsupported, 2 = unreadable, 3 = other size.
int jpeg_prepare_batch(JpegWorkspace** pws, const uint8_t* const* data, const size_t* sizes, int n, int H, int W,
                       int32_t* host_status, std::string* err, const JpegParsed* parsed, int first, const int* pidx,
                       const uint8_t* pin_base, size_t pin_len)
{
    if (!*pws) *pws = new JpegWorkspace();
    JpegWorkspace* w = *pws;
    static const bool trace = diag_env("MELF_JPEG_TRACE") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    std::vector<JpegHeader> own;
    std::vector<size_t> scan_off(n + 1, 0), raw_off(n + 1, 0);
    const int nthreads = host_pool().size() + 1;
    auto par_for = [&](const std::function<void(int)>& fn) { host_pool().run(n, fn); };
    if (!parsed) {
        own.resize(n);
        par_for([&](int i) {
            JpegHeader& h = own[i];
            if (!data[i] || parse_headers(data[i], sizes[i], h) != 0) { host_status[i] = 2; return; }
            if (h.why) { host_status[i] = 1; return; }
            if (h.H != H || h.W != W) { host_status[i] = 3; return; }
            host_status[i] = 0;
        });
    }
    auto hidx = [&](int i) { return parsed ? (pidx ? pidx[first + i] : first + i) : i; };
    const JpegHeader* const hbase = parsed ? parsed->hdr.data() : own.data();
    // Every file of the batch already in the caller's pinned buffer: the upload starts there (one span from the first file's
    // first byte to the last one's last; what lies between the files travels along), no byte is copied here.
    bool direct = pin_base != nullptr && n > 0;
    const uint8_t* span_lo = nullptr;
    const uint8_t* span_hi = nullptr;
    size_t bytes_sum = 0;
    if (direct) {
        for (int i = 0; i < n && direct; ++i) {
            if (host_status[i] != 0) continue;
            if (data[i] < pin_base || data[i] + sizes[i] + 64 > pin_base + pin_len) { direct = false; break; }
            if (!span_lo || data[i] < span_lo) span_lo = data[i];
            if (!span_hi || data[i] + sizes[i] > span_hi) span_hi = data[i] + sizes[i];
            bytes_sum += sizes[i];
        }
        // a batch picked from all over the buffer (the flat files that go first, a second frame size): copy after all
        if (direct && (!span_lo || (size_t)(span_hi - span_lo) > 2 * bytes_sum + (1u << 20))) direct = false;
    }
    const auto tp1 = std::chrono::steady_clock::now();
    // layout
    size_t coef_blocks = 0, plane_bytes = 0;
    int max_blocks = 0;
    std::vector<JpegImageDev> rec(n);
    for (int i = 0; i < n; ++i) {
        JpegImageDev& r = rec[i];
        memset(&r, 0, sizeof(r));
        scan_off[i + 1] = scan_off[i];
        raw_off[i + 1] = raw_off[i];
        if (host_status[i] != 0) continue;
        const JpegHeader& h = hbase[hidx(i)];
        r.ok = 1;
        r.ncomp = (uint8_t)h.ncomp; r.hs0 = (uint8_t)h.hs[0]; r.vs0 = (uint8_t)h.vs[0];
        r.restart_interval = (uint16_t)h.restart_interval;
        r.mcus_x = (uint16_t)((W + 8 * h.hs[0] - 1) / (8 * h.hs[0]));
        r.mcus_y = (uint16_t)((H + 8 * h.vs[0] - 1) / (8 * h.vs[0]));
        int blocks = 0;
        for (int c = 0; c < h.ncomp; ++c) {
            r.tq[c] = (uint8_t)h.tq[c]; r.td[c] = (uint8_t)h.td[c]; r.ta[c] = (uint8_t)h.ta[c];
            r.blocks_x[c] = (uint16_t)(r.mcus_x * (c == 0 ? h.hs[0] : 1));
            r.blocks_y[c] = (uint16_t)(r.mcus_y * (c == 0 ? h.vs[0] : 1));
            r.coef_blk[c] = (uint32_t)coef_blocks;
            r.plane_off[c] = (uint32_t)plane_bytes;
            const size_t nb = (size_t)r.blocks_x[c] * r.blocks_y[c];
            coef_blocks += nb;
            plane_bytes += nb * 64;
            blocks += (int)nb;
        }
        max_blocks = std::max(max_blocks, blocks);
        const size_t raw_len = sizes[i] - h.scan_begin;
        r.raw_off = direct ? (uint32_t)(data[i] + h.scan_begin - span_lo) : (uint32_t)raw_off[i];
        r.raw_len = (uint32_t)raw_len;
        raw_off[i + 1] = raw_off[i] + align_up(raw_len + 32, 64);   // k_jpeg_clean reads up to 20 bytes past the end
        size_t region = align_up(raw_len + 128, 64);
        if (h.restart_interval) {  // room for the table of interval offsets behind the scan
            const size_t mcus = (size_t)r.mcus_x * r.mcus_y;
            r.rst_cnt = (uint32_t)((mcus + h.restart_interval - 1) / h.restart_interval);  // expected; replaced by the number found
            r.rst_off = (uint32_t)region;                                                   // relative for now
            region += align_up(((size_t)r.rst_cnt + 1) * sizeof(uint32_t), 64);
        }
        scan_off[i + 1] = scan_off[i] + region;
    }
    if (coef_blocks >= (1ull << 32) / 64 || plane_bytes >= (1ull << 32) || scan_off[n] >= (1ull << 32) || raw_off[n] >= (1ull << 32)) {
        if (err) *err = "JPEG batch too large for 32-bit offsets; decode in smaller batches";
        return MELF_ERR_TOO_LARGE;
    }
    w->off_imgs = 0;
    w->off_qt = align_up(w->off_imgs + (size_t)n * sizeof(JpegImageDev), 256);
    w->off_slow = align_up(w->off_qt + (size_t)n * 4 * 64 * sizeof(uint16_t), 256);
    w->off_scan = align_up(w->off_slow + (size_t)n * 4 * sizeof(HuffSlow), 256);
    w->total = w->off_scan;                 // uploaded: records + tables
    w->scan_bytes = scan_off[n] + 64;       // device only
    w->raw_total = direct ? (size_t)(span_hi - span_lo) + 64 : raw_off[n] + 64;
    if (w->raw_total >= (1ull << 32)) {
        if (err) *err = "JPEG batch too large for 32-bit offsets; decode in smaller batches";
        return MELF_ERR_TOO_LARGE;
    }
    if (!direct && w->raw_total > w->h_raw_cap) {
        if (w->h_raw) (void)hipHostFree(w->h_raw);
        w->h_raw = nullptr; w->h_raw_cap = 0;
        const size_t want = w->raw_total + w->raw_total / 4;
        if (hipHostMalloc((void**)&w->h_raw, want, hipHostMallocDefault) != hipSuccess) {
            if (err) *err = "hipHostMalloc failed for the JPEG raw-scan buffer";
            return MELF_ERR_HIP;
        }
        w->h_raw_cap = want;
    }
    w->raw_src = direct ? span_lo : w->h_raw;
    w->coef_elems = coef_blocks * 64;
    w->plane_bytes = plane_bytes;
    w->max_blocks = max_blocks;
    if (w->total > w->h_cap) {
        if (w->h_stage) (void)hipHostFree(w->h_stage);
        w->h_stage = nullptr; w->h_cap = 0;
        const size_t want = w->total + w->total / 4;
        if (hipHostMalloc((void**)&w->h_stage, want, hipHostMallocDefault) != hipSuccess) {
            if (err) *err = "hipHostMalloc failed for the JPEG staging buffer";
            return MELF_ERR_HIP;
        }
        w->h_cap = want;
    }
    const auto tp2 = std::chrono::steady_clock::now();
    uint8_t* base = w->h_stage;
    const bool tables_ready = parsed && !parsed->slow.empty();
    auto fill = [&](int i) {
        JpegImageDev& r = rec[i];
        uint16_t* qt = (uint16_t*)(base + w->off_qt) + (size_t)i * 256;
        HuffSlow* slow = (HuffSlow*)(base + w->off_slow) + (size_t)i * 4;
        if (host_status[i] != 0) return;
        const JpegHeader& h = hbase[hidx(i)];
        for (int t = 0; t < 4; ++t) {
            if (h.qt_set[t]) memcpy(qt + t * 64, h.qt[t], 128); else memset(qt + t * 64, 0, 128);
        }
        if (tables_ready) memcpy(slow, parsed->slow.data() + (size_t)hidx(i) * 4, 4 * sizeof(HuffSlow));
        else build_slow4(h, slow);
        // the entropy-coded segment as it is: stuffing, fill bytes and restart markers are taken out on the GPU (k_jpeg_clean),
        // which also writes the clean length and the restart table (r.rst_cnt: the EXPECTED number of intervals until then)
        if (!direct) {
            // streaming (non-temporal) stores: the pinned buffer is written once and read by the DMA engine, never by this core
            // (a plain memcpy reads every destination line before it writes it: a third of the copy's memory traffic)
            uint8_t* dst = w->h_raw + r.raw_off;   // 64-byte aligned
            const uint8_t* src = data[i] + h.scan_begin;
            const size_t whole = (size_t)r.raw_len & ~(size_t)15;
            for (size_t o = 0; o < whole; o += 16) _mm_stream_si128((__m128i*)(dst + o), _mm_loadu_si128((const __m128i*)(src + o)));
            memcpy(dst + whole, src + whole, r.raw_len - whole);
            _mm_sfence();
        }
        r.scan_cap = (uint32_t)(h.restart_interval ? r.rst_off : scan_off[i + 1] - scan_off[i]);
        r.scan_off = (uint32_t)scan_off[i];
        r.scan_len = 0;
        if (h.restart_interval) r.rst_off = (uint32_t)(scan_off[i] + r.rst_off);
        r.ok = h.restart_interval != 0 ? 2 : 1;
        // longer than 1024 lanes x the segment length k_jpeg_huff can address (the raw length: the clean one is only known on
        // the GPU and at most 1/128 shorter)
        if (!h.restart_interval && r.raw_len > JPEG_MAX_PAR_SCAN) {
            r.ok = 0;
            host_status[i] = 1;  // valid, but for the host decoder
        }
    };
    // nothing but two small copies per file left to do: not worth waking the pool (whose threads the file reads of the next
    // call are using)
    if (direct && tables_ready) { for (int i = 0; i < n; ++i) fill(i); }
    else par_for(fill);
    w->n_par = w->n_seq = w->n_420 = 0;
    w->max_par_scan = 0;
    for (int i = 0; i < n; ++i) {
        if (rec[i].ok == 1) { ++w->n_par; w->max_par_scan = std::max(w->max_par_scan, (size_t)rec[i].raw_len); }
        else if (rec[i].ok == 2) ++w->n_seq;
        if (rec[i].ok && rec[i].ncomp == 3 && rec[i].hs0 == 2 && rec[i].vs0 == 2) ++w->n_420;
    }
    memcpy(base + w->off_imgs, rec.data(), (size_t)n * sizeof(JpegImageDev));
    if (trace) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[melf jpeg] prepare: parse %.2f ms, layout %.2f ms, tables + scan copies %.2f ms (%d threads)\n", ms(tp0, tp1), ms(tp1, tp2),
                ms(tp2, tp3), nthreads);
    }
    return MELF_SUCCESS;
}

// Device half, in two steps so that a caller can pipeline batches: jpeg_upload_batch grows the device buffers and copies
// the stage buffer (on `copy_stream`), jpeg_decode_batch_kernels runs J1..J3 (on `stream`) and leaves the per-file status
// in w->d_status.  `timer(k)` brackets kernel k (0..2) when profiling.
#define JTRY(expr)                                                               \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) {                                                  \
            if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_);   \
            return MELF_ERR_HIP;                                                 \
        }                                                                        \
    } while (0)
int jpeg_upload_batch(JpegWorkspace* w, int n, hipStream_t copy_stream, std::string* err)
{
    JTRY(grow_dev(&w->d_stage, &w->d_cap, w->total + w->scan_bytes));
    JTRY(grow_dev(&w->d_raw, &w->d_raw_cap, w->raw_total));
    JTRY(grow_dev(&w->d_coefs, &w->coef_cap, w->coef_elems + 64));
    JTRY(grow_dev(&w->d_planes, &w->plane_cap, w->plane_bytes + 64));
    JTRY(grow_dev(&w->d_status, &w->status_cap, (size_t)n));
    JTRY(hipMemcpyAsync(w->d_stage, w->h_stage, w->total, hipMemcpyHostToDevice, copy_stream));
    JTRY(hipMemcpyAsync(w->d_raw, w->raw_src, w->raw_total, hipMemcpyHostToDevice, copy_stream));
    return MELF_SUCCESS;
}

int jpeg_decode_batch_kernels(JpegWorkspace* w, int n, int H, int W, uint8_t* d_frames, hipStream_t stream, std::string* err,
                              void (*timer)(void*, int, int), void* timer_arg, const int* rect)
{
    // window: the caller's rectangle grown by one MCU (chroma filter context) and aligned to 16, or the frame
    JpegWindow win = {0, 0, W, H};
    if (rect) {
        win.x0 = std::max(0, (rect[0] - 16) & ~15); win.y0 = std::max(0, (rect[1] - 16) & ~15);
        win.x1 = std::min(W, (rect[2] + 16 + 15) & ~15); win.y1 = std::min(H, (rect[3] + 16 + 15) & ~15);
        if (win.x1 <= win.x0 || win.y1 <= win.y0) win = {0, 0, W, H};
    }
    const int win_w = win.x1 - win.x0, win_h = win.y1 - win.y0;
    JTRY(hipMemsetAsync(w->d_status, 0, (size_t)n * sizeof(int32_t), stream));
    const JpegImageDev* imgs = (const JpegImageDev*)(w->d_stage + w->off_imgs);
    const uint16_t* qt = (const uint16_t*)(w->d_stage + w->off_qt);
    const HuffSlow* slow = (const HuffSlow*)(w->d_stage + w->off_slow);
    const uint8_t* scan = w->d_stage + w->off_scan;
    if (timer) timer(timer_arg, 0, 0);
    if (w->n_par + w->n_seq > 0)   // J0: stuffing, fill bytes and restart markers out of the uploaded segments
        hipLaunchKernelGGL(k_jpeg_clean, dim3(n), dim3(CLEAN_T), 0, stream, (JpegImageDev*)(w->d_stage + w->off_imgs), w->d_raw, w->d_stage + w->off_scan);
    if (w->n_par > 0) {  // one workgroup per image, one lane per stream segment
        static int tenv = -1;
        if (tenv < 0) { const char* e = diag_env("MELF_JPEG_T"); tenv = e ? atoi(e) : 0; }
        // 512 lanes per image; a batch with a scan too long for 512 segments of the length the kernel can address takes 1024
        const int tsel = tenv ? tenv : (w->max_par_scan > JPEG_MAX_PAR_SCAN / 2 ? 1024 : 512);
        static const int limit = diag_env("MELF_JPEG_WINDOW_LIMIT") ? atoi(diag_env("MELF_JPEG_WINDOW_LIMIT")) : 1;   // A/B switch
#define LAUNCH_HUFF(TT) \
    hipLaunchKernelGGL(k_jpeg_huff<TT>, dim3(n), dim3(TT), 0, stream, imgs, slow, scan, w->d_coefs, w->d_status, win, limit)
        if (tsel == 128) LAUNCH_HUFF(128);
        else if (tsel == 256) LAUNCH_HUFF(256);
        else if (tsel == 1024) LAUNCH_HUFF(1024);
        else LAUNCH_HUFF(512);
#undef LAUNCH_HUFF
    }
    if (w->n_seq > 0) {  // streams with restart intervals: one lane per interval
        hipLaunchKernelGGL(k_jpeg_huff_rst<256>, dim3(n), dim3(256), 0, stream, imgs, slow, scan, w->d_coefs, w->d_status, win);
    }
    if (timer) timer(timer_arg, 0, 1);
    JTRY(hipGetLastError());
    if (w->max_blocks > 0) {
        if (timer) timer(timer_arg, 1, 0);
        // at most (window / 8)^2 luma blocks + two chroma planes of up to the same count (4:4:4)
        const int wblocks = 3 * ((win_w + 15) / 8 + 2) * ((win_h + 15) / 8 + 2);
        const int nblk = std::min(w->max_blocks, wblocks);
        hipLaunchKernelGGL(k_jpeg_idct, dim3((nblk + 255) / 256, n), dim3(256), 0, stream, imgs, qt, w->d_coefs, w->d_status, w->d_planes, win);
        if (timer) timer(timer_arg, 1, 1);
        JTRY(hipGetLastError());
    }
    if (timer) timer(timer_arg, 2, 0);
    const int fast420 = (W % 8 == 0 && w->n_420 > 0) ? 1 : 0;
    if (fast420) {
        const int items = ((win_w + 7) / 8) * ((win_h + 1) / 2);   // 8 pixels x 2 rows each
        hipLaunchKernelGGL(k_jpeg_color420, dim3((items + 255) / 256, n), dim3(256), 0, stream, imgs, w->d_status, w->d_planes, H, W, d_frames, win);
    }
    if (!fast420 || w->n_420 < n)
        hipLaunchKernelGGL(k_jpeg_color, dim3((win_w + 1023) / 1024, win_h, n), dim3(256), 0, stream, imgs, w->d_status, w->d_planes, H, W, d_frames, fast420, win);
    if (timer) timer(timer_arg, 2, 1);
    JTRY(hipGetLastError());
    return MELF_SUCCESS;
}

const int32_t* jpeg_device_status(const JpegWorkspace* w) { return w->d_status; }

// One batch start to end on one stream (upload, J1..J3, status back, synchronised when status_out_host is given).
int jpeg_launch_batch(JpegWorkspace* w, int n, int H, int W, uint8_t* d_frames, int32_t* status_out_host,
                      hipStream_t stream, std::string* err, void (*timer)(void*, int, int), void* timer_arg, const int* rect)
{
    if (int rc = jpeg_upload_batch(w, n, stream, err)) return rc;
    if (int rc = jpeg_decode_batch_kernels(w, n, H, W, d_frames, stream, err, timer, timer_arg, rect)) return rc;
    if (status_out_host) {
        JTRY(hipMemcpyAsync(status_out_host, w->d_status, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        JTRY(hipStreamSynchronize(stream));
    }
    return MELF_SUCCESS;
}
#undef JTRY

}  // namespace melf
