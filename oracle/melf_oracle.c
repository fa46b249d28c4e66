/*
 * melf_oracle.c -- CPU restatement of meterelf's per-image hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the
 * "restated CPU baseline".  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (meterelf_amd/)
 * never links, imports or calls anything in oracle/.
 *
 * Parity pinning: the reference's arithmetic lives in the un-vendored
 * third-party dependency opencv-python==3.4.5.20 (reference
 * requirements.txt:7), which is absent here, so this is a restatement of
 * OpenCV 3.4's published algorithms (cvtColor BGR2HLS_FULL, matchTemplate
 * TM_CCOEFF, inRange, dilate/erode, findContours/contourArea/drawContours,
 * circle/floodFill) driven exactly as the reference drives them.  It is
 * pinned by the reference's own goldens: all 81 + 223 stdout lines of
 * tests/sample-images{1,2}_stdout.txt, the e136 intermediate golden of
 * tests/test_meterelf.py:170-188 and the get_angle_by_vector doctest
 * (meterelf/_utils.py:32-36); see tests/test_oracle_golden.py.
 *
 * Every function cites the reference file:line it follows (paths relative
 * to the reference checkout).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * -ffp-contract=off matters: OpenCV's SSE2 baseline never fuses a*b+c.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))
#define ORC_MAX_DIALS 8

/* status codes of a frame (mirror the reference's exception classes,
 * meterelf/exceptions.py:35-52) */
enum {
    ORC_OK = 0,
    ORC_DIALS_NOT_FOUND = 1,           /* DialsNotFoundError            */
    ORC_NEEDLE_CONTOURS_NOT_FOUND = 2, /* NeedleContoursNotFoundError   */
    ORC_ANGLE_UNDETERMINED = 3         /* DialAngleDeterminingError     */
};

typedef struct {
    double cx, cy;        /* dial centre in dials-crop coordinates (params.yml center) */
    int32_t range_h, range_l, range_s; /* color_range */
    int32_t negative_momentum;
    double angle_of_zero; /* degrees */
} orc_dial;

typedef struct {
    int32_t th, tw;       /* template rows, cols (dials_template_size swapped to (h,w),
                             meterelf/_params.py:136-138) */
    int32_t hue_shift;
    int32_t ndials;
    double match_threshold;
    orc_dial dial[ORC_MAX_DIALS];
} orc_params;

typedef struct {
    int32_t status;
    int32_t match_x, match_y;
    int32_t failed_dial;        /* NEEDLE_CONTOURS_NOT_FOUND: index of the dial */
    uint32_t unreadable_mask;   /* ANGLE_UNDETERMINED: bit d = dial d unreadable */
    float match_val;
    double pos[ORC_MAX_DIALS];    /* dial positions 0..10 */
    double angle[ORC_MAX_DIALS];  /* needle angle in turns before zero fix */
    double value;                 /* meter value (only when status OK and ndials == 4) */
    /* intermediate facts for finer-grained parity checks */
    int32_t dial_color[ORC_MAX_DIALS][3];
    int32_t n_needle[ORC_MAX_DIALS];
    int32_t n_outer[ORC_MAX_DIALS];
    int32_t n_kept[ORC_MAX_DIALS];
    double contour_area[ORC_MAX_DIALS];
} orc_result;

/* ------------------------------------------------------------------ */
/* A.2  cvtColor(COLOR_BGR2HLS_FULL) on u8  + uint8 hue shift           */
/* reference: meterelf/_utils.py:100-102                                */
/* ------------------------------------------------------------------ */

/* Sensitivity audit (tests/test_oracle_sensitivity.py): every OpenCV semantic this restatement had to
 * BELIEVE (OpenCV is absent) can be flipped to its plausible alternative, so that a test can
 * count how many of the reference's 304 golden lines notice.  Value 0 is always the restatement proper. */
enum {
    ORC_OPT_HLS_VARIANT = 0,   /* 1: every pixel through the scalar-tail S formula, 2: every pixel through the SIMD form */
    ORC_OPT_CONTOUR_TIE = 1,   /* 1: among equal contour areas the LAST discovered wins                              */
    ORC_OPT_MEAN_FORM = 2,     /* 1: cv::mean as sum / N instead of sum * (1. / N) (dial colour and template mean)   */
    ORC_OPT_HLS_ROUND = 3,     /* 1: round half away from zero instead of cvRound's half-to-even                     */
    ORC_OPT_AREA_RULE = 4,     /* 1: contourArea = pixel count of the filled contour instead of the polygon area     */
    ORC_OPT_NO_HOLE_FILL = 5,  /* 1: drawContours(-1) paints the component only, holes stay open                     */
    ORC_OPT_ERODE_BORDER = 6,  /* 1: erode sees zeros outside the image instead of the neutral default border        */
    ORC_OPT_L_INTEGER = 7,     /* 1: L = (max + min + 1) >> 1 instead of the float32 path                            */
    ORC_OPT_MINMAX_LAST = 8,   /* 1: minMaxLoc returns the last maximum in raster order instead of the first         */
    ORC_OPT_HUE_G_FIRST = 9,   /* 1: the hue sector test tries vmax == g before vmax == r                            */
    ORC_OPT_COUNT = 10
};
static int g_opt[ORC_OPT_COUNT];
#define g_hls_variant (g_opt[ORC_OPT_HLS_VARIANT])
ORC_API void orc_set_hls_variant(int v) { g_opt[ORC_OPT_HLS_VARIANT] = v; }
ORC_API int orc_set_option(int key, int value)
{
    if (key < 0 || key >= ORC_OPT_COUNT) return -1;
    g_opt[key] = value;
    return 0;
}
ORC_API int orc_option_count(void) { return ORC_OPT_COUNT; }

static inline uint8_t sat_u8_rne(float v)
{
    /* cvRound (SSE2 cvtss2si, round-half-even) then saturate_cast<uchar> */
    long r = g_opt[ORC_OPT_HLS_ROUND] ? (long)floorf(v + 0.5f) : lrintf(v);
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

/* One pixel.  scalar_tail selects the operation order of OpenCV's scalar
 * remainder loop (S uses 2 - vmax - vmin), otherwise the 4-wide SIMD body
 * (S uses 2 - (vmax + vmin)). */
static inline void hls_pixel(uint8_t b8, uint8_t g8, uint8_t r8, int scalar_tail,
                             uint8_t* H, uint8_t* L, uint8_t* S)
{
    const float inv255 = 1.f / 255.f;
    const float hscale = 256.f / 360.f;
    float b = b8 * inv255, g = g8 * inv255, r = r8 * inv255;
    float vmax = r, vmin = r;
    if (vmax < g) vmax = g;
    if (vmax < b) vmax = b;
    if (vmin > g) vmin = g;
    if (vmin > b) vmin = b;
    float diff = vmax - vmin;
    float sum = vmax + vmin;
    float l = sum * 0.5f;
    float h = 0.f, s = 0.f;
    if (diff > FLT_EPSILON) {
        if (scalar_tail)
            s = l < 0.5f ? diff / (vmax + vmin) : diff / (2 - vmax - vmin);
        else
            s = diff / (l < 0.5f ? sum : 2.0f - sum);
        float k = 60.f / diff;
        if (g_opt[ORC_OPT_HUE_G_FIRST] && vmax == g)
            h = (b - r) * k + 120.f;
        else if (vmax == r)
            h = (g - b) * k + (g < b ? 360.f : 0.f);
        else if (vmax == g)
            h = (b - r) * k + 120.f;
        else
            h = (r - g) * k + 240.f;
        if (h < 0.f) h += 360.f; /* cannot trigger after the lines above; kept for the scalar form */
    }
    *H = sat_u8_rne(h * hscale);
    *L = sat_u8_rne(l * 255.f);
    *S = sat_u8_rne(s * 255.f);
    if (g_opt[ORC_OPT_L_INTEGER]) {
        int mx8 = r8 > g8 ? (r8 > b8 ? r8 : b8) : (g8 > b8 ? g8 : b8);
        int mn8 = r8 < g8 ? (r8 < b8 ? r8 : b8) : (g8 < b8 ? g8 : b8);
        *L = (uint8_t)((mx8 + mn8 + 1) >> 1);
    }
}

/* OpenCV converts row by row, each row in blocks of 256 px; within a block the
 * first 4*floor(n/4) pixels go through the SIMD body and the rest through the
 * scalar tail (RGB2HLS_b / RGB2HLS_f in color_hsv.cpp of 3.4). */
ORC_API void orc_bgr2hls_full(const uint8_t* bgr, int rows, int cols, long row_stride,
                              int hue_shift, uint8_t* hls /* rows*cols*3 packed */)
{
    for (int y = 0; y < rows; ++y) {
        const uint8_t* s = bgr + (long)y * row_stride;
        uint8_t* d = hls + (long)y * cols * 3;
        for (int x0 = 0; x0 < cols; x0 += 256) {
            int dn = cols - x0 < 256 ? cols - x0 : 256;
            int simd_n = dn & ~3;
            for (int j = 0; j < dn; ++j) {
                int x = x0 + j;
                uint8_t H, L, S;
                int tail = (j >= simd_n);
                if (g_hls_variant == 1) tail = 1;
                if (g_hls_variant == 2) tail = 0;
                hls_pixel(s[3 * x], s[3 * x + 1], s[3 * x + 2], tail, &H, &L, &S);
                d[3 * x] = (uint8_t)(H + hue_shift); /* numpy uint8 wrap-around add */
                d[3 * x + 1] = L;
                d[3 * x + 2] = S;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* A.3  matchTemplate(TM_CCOEFF) + minMaxLoc                            */
/* reference: meterelf/_utils.py:91-97, meterelf/_image.py:57-66        */
/* ------------------------------------------------------------------ */

/* Exact integer cross-correlation (OpenCV computes it by float32 DFT; the
 * exact value is what that approximates), then OpenCV's own post-pass:
 * num = (double)cc - winsum * mean(T); result = (float)num.
 * mean(T) = sum * (1.0/N) as cv::mean computes it. */
ORC_API void orc_match_ccoeff(const uint8_t* img, int rows, int cols, long stride,
                              const uint8_t* tpl, int th, int tw,
                              float* result /* (rows-th+1)*(cols-tw+1) or NULL */,
                              float* max_val, int* max_x, int* max_y)
{
    int rh = rows - th + 1, rw = cols - tw + 1;
    long tsum = 0;
    for (int i = 0; i < th * tw; ++i) tsum += tpl[i];
    double tmean = g_opt[ORC_OPT_MEAN_FORM] ? (double)tsum / ((double)th * tw) : (double)tsum * (1.0 / ((double)th * tw));

    /* integral image for window sums (exact) */
    long* integ = (long*)calloc((size_t)(rows + 1) * (cols + 1), sizeof(long));
    for (int y = 0; y < rows; ++y) {
        long rs = 0;
        for (int x = 0; x < cols; ++x) {
            rs += img[(long)y * stride + x];
            integ[(long)(y + 1) * (cols + 1) + x + 1] = integ[(long)y * (cols + 1) + x + 1] + rs;
        }
    }
    int32_t* acc = (int32_t*)malloc(sizeof(int32_t) * rw);
    float best = 0.f;
    int bx = -1, by = -1;
    for (int y = 0; y < rh; ++y) {
        memset(acc, 0, sizeof(int32_t) * rw);
        for (int i = 0; i < th; ++i) {
            const uint8_t* irow = img + (long)(y + i) * stride;
            const uint8_t* trow = tpl + (long)i * tw;
            for (int j = 0; j < tw; ++j) {
                int32_t t = trow[j];
                const uint8_t* ip = irow + j;
                for (int x = 0; x < rw; ++x) acc[x] += t * (int32_t)ip[x];
            }
        }
        for (int x = 0; x < rw; ++x) {
            long ws = integ[(long)(y + th) * (cols + 1) + x + tw] - integ[(long)y * (cols + 1) + x + tw]
                    - integ[(long)(y + th) * (cols + 1) + x] + integ[(long)y * (cols + 1) + x];
            double num = (double)acc[x];
            num -= (double)ws * tmean;
            float r = (float)num;
            if (result) result[(long)y * rw + x] = r;
            /* minMaxLoc: strict > in raster order => first maximum */
            if (bx < 0 || r > best || (g_opt[ORC_OPT_MINMAX_LAST] && r == best)) { best = r; bx = x; by = y; }
        }
    }
    free(acc);
    free(integ);
    *max_val = best; *max_x = bx; *max_y = by;
}

/* ------------------------------------------------------------------ */
/* A.9  dial masks: cv2.circle + cv2.floodFill                          */
/* reference: meterelf/_dial_data.py:22-48                              */
/* ------------------------------------------------------------------ */

static double py_round(double v) { return nearbyint(v); } /* Python 3 round(): half to even */

static void put_px(uint8_t* img, int rows, int cols, int x, int y)
{
    if (x >= 0 && x < cols && y >= 0 && y < rows) img[(long)y * cols + x] = 255;
}

/* OpenCV Circle() (drawing.cpp), thickness 1, LINE_8, shift 0 */
static void draw_circle(uint8_t* img, int rows, int cols, int cx, int cy, int radius)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        int mask;
        int y11 = cy - dy, y12 = cy + dy, y21 = cy - dx, y22 = cy + dx;
        int x11 = cx - dx, x12 = cx + dx, x21 = cx - dy, x22 = cx + dy;
        put_px(img, rows, cols, x11, y11); put_px(img, rows, cols, x11, y12);
        put_px(img, rows, cols, x12, y11); put_px(img, rows, cols, x12, y12);
        put_px(img, rows, cols, x21, y21); put_px(img, rows, cols, x21, y22);
        put_px(img, rows, cols, x22, y21); put_px(img, rows, cols, x22, y22);
        dy++;
        err += plus;
        plus += 2;
        mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

/* cv2.floodFill(img, fill_mask, seed, 255): 4-connectivity, loDiff=upDiff=0;
 * pixels whose fill_mask entry is non-zero are barriers; filled pixels get
 * fill_mask = 1.  fill_mask is (rows+2)x(cols+2) as in the reference. */
static void flood_fill(uint8_t* img, uint8_t* fmask, int rows, int cols, int sx, int sy)
{
    if (sx < 0 || sx >= cols || sy < 0 || sy >= rows) return; /* cv2 would raise */
    uint8_t v0 = img[(long)sy * cols + sx];
    int* stack = (int*)malloc(sizeof(int) * (size_t)rows * cols * 4 + 16);
    int sp = 0;
    long mstep = cols + 2;
#define FM(x, y) fmask[(long)((y) + 1) * mstep + (x) + 1]
    if (FM(sx, sy)) { free(stack); return; }
    /* work on a visited map so that newVal == seedVal cannot loop */
    FM(sx, sy) = 1;
    stack[sp++] = sy * cols + sx;
    while (sp) {
        int p = stack[--sp];
        int x = p % cols, y = p / cols;
        img[p] = 255;
        static const int dxs[4] = {1, -1, 0, 0}, dys[4] = {0, 0, 1, -1};
        for (int k = 0; k < 4; ++k) {
            int nx = x + dxs[k], ny = y + dys[k];
            if (nx < 0 || nx >= cols || ny < 0 || ny >= rows) continue;
            if (FM(nx, ny)) continue;
            if (img[(long)ny * cols + nx] != v0) continue;
            FM(nx, ny) = 1;
            stack[sp++] = ny * cols + nx;
        }
    }
#undef FM
    free(stack);
}

/* masks: [ndials][2][th*tw], plane 0 = `mask` (disk), plane 1 = `circle_mask`
 * (annulus).  diameter/dist/thickness per dial from params.yml. */
ORC_API void orc_build_dial_masks(int th, int tw, int ndials, const double* centers_xy,
                                  const int* diameter, const int* dist_from_center,
                                  const int* circle_thickness, uint8_t* masks)
{
    long n = (long)th * tw;
    for (int d = 0; d < ndials; ++d) {
        uint8_t* mask = masks + (long)d * 2 * n;
        uint8_t* circle_mask = mask + n;
        memset(mask, 0, n);
        int dial_radius = (int)py_round(diameter[d] / 2.0);        /* _dial_data.py:28 */
        int cx = (int)py_round(centers_xy[2 * d]);                  /* float_point_to_int, _utils.py:14-15 */
        int cy = (int)py_round(centers_xy[2 * d + 1]);
        int start_radius = dial_radius + dist_from_center[d];
        int ii[2] = {0, circle_thickness[d] - 1};
        for (int k = 0; k < 2; ++k) draw_circle(mask, th, tw, cx, cy, start_radius + ii[k]);
        uint8_t* fmask = (uint8_t*)calloc((size_t)(th + 2) * (tw + 2), 1);
        flood_fill(mask, fmask, th, tw, cx + start_radius + 1, cy);
        memcpy(circle_mask, mask, n);
        flood_fill(mask, fmask, th, tw, cx, cy);
        free(fmask);
    }
}

/* ------------------------------------------------------------------ */
/* A.5/A.6  inRange + 3x3 dilate + 3x3 erode                            */
/* reference: meterelf/_utils.py:113-119, meterelf/_reading.py:124-130  */
/* ------------------------------------------------------------------ */

ORC_API void orc_inrange(const uint8_t* hls, int rows, int cols, const int* lo, const int* hi, uint8_t* out)
{
    for (long p = 0; p < (long)rows * cols; ++p) {
        const uint8_t* v = hls + 3 * p;
        out[p] = (v[0] >= lo[0] && v[0] <= hi[0] && v[1] >= lo[1] && v[1] <= hi[1] &&
                  v[2] >= lo[2] && v[2] <= hi[2]) ? 255 : 0;
    }
}

/* 3x3 rect, anchor centre, border pixels never win (cv2 default border value) */
static void morph3(const uint8_t* in, int rows, int cols, int is_dilate, uint8_t* out)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int v = is_dilate ? 0 : 255;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= rows || xx < 0 || xx >= cols) {
                        if (!is_dilate && g_opt[ORC_OPT_ERODE_BORDER]) v = 0;
                        continue;
                    }
                    int q = in[(long)yy * cols + xx];
                    if (is_dilate ? q > v : q < v) v = q;
                }
            out[(long)y * cols + x] = (uint8_t)v;
        }
}

ORC_API void orc_close3(const uint8_t* in, int rows, int cols, uint8_t* out)
{
    uint8_t* tmp = (uint8_t*)malloc((size_t)rows * cols);
    morph3(in, rows, cols, 1, tmp);
    morph3(tmp, rows, cols, 0, out);
    free(tmp);
}

/* The full-frame fused stage of BASELINE config 2: HLS(+shift) -> inRange with
 * fixed bounds -> closing; out = u8 mask {0,255}.  Optionally emits the L plane. */
ORC_API void orc_hls_inrange_close(const uint8_t* bgr, int rows, int cols, long stride, int hue_shift,
                                   const int* lo, const int* hi, uint8_t* mask_out, uint8_t* l_out)
{
    uint8_t* hls = (uint8_t*)malloc((size_t)rows * cols * 3);
    uint8_t* m = (uint8_t*)malloc((size_t)rows * cols);
    orc_bgr2hls_full(bgr, rows, cols, stride, hue_shift, hls);
    orc_inrange(hls, rows, cols, lo, hi, m);
    orc_close3(m, rows, cols, mask_out);
    if (l_out)
        for (long p = 0; p < (long)rows * cols; ++p) l_out[p] = hls[3 * p + 1];
    free(hls);
    free(m);
}

/* ------------------------------------------------------------------ */
/* A.7  findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) + contourArea    */
/* reference: meterelf/_reading.py:132-148                              */
/* ------------------------------------------------------------------ */

/* 8 directions in OpenCV chain-code order: 0 = E, 1 = NE, 2 = N ... 7 = SE */
static const int CDX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int CDY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

typedef struct {
    double area;    /* cv2.contourArea of the outer border */
    int start;      /* raster index of the border's start pixel (discovery order) */
} contour_rec;

/* Suzuki-Abe outer-border following as cvFindNextContour/icvFetchContour do it
 * in RETR_EXTERNAL mode: raster scan of a 1-px zero-padded copy; an outer
 * border starts at a 0->1 transition whose last marked border pixel on the row
 * (lnbd) is not a positively marked one; traced pixels are marked +2, or -2
 * when the border leaves them to the right.  Returns number of contours. */
static int find_external_contours_pts(const uint8_t* bin, int rows, int cols, contour_rec* out, int max_out,
                                      int32_t* pts, int max_pts, int32_t* counts);

static int find_external_contours(const uint8_t* bin, int rows, int cols, contour_rec* out, int max_out)
{
    return find_external_contours_pts(bin, rows, cols, out, max_out, NULL, 0, NULL);
}

/* pts (optional): x,y pairs of every contour in discovery order, coordinates of the original
 * image; counts[k] = number of points of contour k */
static int find_external_contours_pts(const uint8_t* bin, int rows, int cols, contour_rec* out, int max_out,
                                      int32_t* pts, int max_pts, int32_t* counts)
{
    int npts_total = 0;
    int W = cols + 2, Hh = rows + 2;
    int8_t* img = (int8_t*)calloc((size_t)W * Hh, 1);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) img[(y + 1) * W + x + 1] = bin[(long)y * cols + x] ? 1 : 0;
    int delt[16];
    for (int k = 0; k < 8; ++k) delt[k] = delt[k + 8] = CDY[k] * W + CDX[k];
    int n = 0;
    for (int y = 1; y < Hh - 1; ++y) {
        int lnbd_x = 0;
        int prev = 0;
        for (int x = 1; x < W - 1; ++x) {
            int p = img[y * W + x];
            if (p == prev) continue;
            if (prev == 0 && p == 1 && !(img[y * W + lnbd_x] > 0)) {
                /* follow the outer border starting at (x, y) */
                int i0 = y * W + x, i1, i3, i4 = 0;
                int s = 4, s_end = 4;
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + delt[s];
                } while (img[i1] == 0 && s != s_end);
                double a00 = 0;
                long npts = 0;
                int px = x, py = y, fx = x, fy = y, qx = x, qy = y;
                if (s == s_end) {
                    img[i0] = (int8_t)(2 | -128);
                    npts = 1;
                    if (pts && npts_total < max_pts) { pts[2 * npts_total] = x - 1; pts[2 * npts_total + 1] = y - 1; ++npts_total; }
                } else {
                    i3 = i0;
                    for (;;) {
                        s_end = s;
                        while (s < 15) {
                            i4 = i3 + delt[++s];
                            if (img[i4] != 0) break;
                        }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end)
                            img[i3] = (int8_t)(2 | -128);
                        else if (img[i3] == 1)
                            img[i3] = 2;
                        /* emit point (px,py), then step */
                        if (npts > 0) a00 += (double)qx * py - (double)qy * px;
                        qx = px; qy = py;
                        ++npts;
                        if (pts && npts_total < max_pts) { pts[2 * npts_total] = px - 1; pts[2 * npts_total + 1] = py - 1; ++npts_total; }
                        px += CDX[s]; py += CDY[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                    /* closing edge: last point -> first point */
                    a00 += (double)qx * fy - (double)qy * fx;
                }
                if (n < max_out) {
                    out[n].area = fabs(a00 * 0.5);
                    out[n].start = (y - 1) * cols + (x - 1);
                    if (counts) counts[n] = (int32_t)npts;
                }
                ++n;
                p = img[y * W + x];
            }
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
    free(img);
    return n;
}

/* Region painted by cv2.drawContours(zeros, [outer border], -1, 255, -1):
 * the 8-connected component containing `start` plus everything its outer
 * border encloses (complement pixels not 4-connected to the outside). */
static void fill_external_contour(const uint8_t* bin, int rows, int cols, int start, uint8_t* out)
{
    long n = (long)rows * cols;
    uint8_t* comp = (uint8_t*)calloc(n, 1);
    int* stack = (int*)malloc(sizeof(int) * (size_t)(rows + 2) * (cols + 2) + 16);
    int sp = 0;
    comp[start] = 1;
    stack[sp++] = start;
    while (sp) {
        int p = stack[--sp];
        int x = p % cols, y = p / cols;
        for (int k = 0; k < 8; ++k) {
            int nx = x + CDX[k], ny = y + CDY[k];
            if (nx < 0 || nx >= cols || ny < 0 || ny >= rows) continue;
            long q = (long)ny * cols + nx;
            if (!bin[q] || comp[q]) continue;
            comp[q] = 1;
            stack[sp++] = (int)q;
        }
    }
    /* flood the complement from a 1-px outside frame, 4-connected */
    int W = cols + 2, Hh = rows + 2;
    uint8_t* outside = (uint8_t*)calloc((size_t)W * Hh, 1);
    sp = 0;
    outside[0] = 1;
    stack[sp++] = 0;
    while (sp) {
        int p = stack[--sp];
        int x = p % W, y = p / W;
        static const int dxs[4] = {1, -1, 0, 0}, dys[4] = {0, 0, 1, -1};
        for (int k = 0; k < 4; ++k) {
            int nx = x + dxs[k], ny = y + dys[k];
            if (nx < 0 || nx >= W || ny < 0 || ny >= Hh) continue;
            int q = ny * W + nx;
            if (outside[q]) continue;
            int ix = nx - 1, iy = ny - 1;
            if (ix >= 0 && ix < cols && iy >= 0 && iy < rows && comp[(long)iy * cols + ix]) continue;
            outside[q] = 1;
            stack[sp++] = q;
        }
    }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            out[(long)y * cols + x] = g_opt[ORC_OPT_NO_HOLE_FILL] ? (comp[(long)y * cols + x] ? 255 : 0)
                                                                  : (outside[(y + 1) * W + x + 1] ? 0 : 255);
    free(outside);
    free(stack);
    free(comp);
}

/* cv2.findContours(img, RETR_EXTERNAL, CHAIN_APPROX_NONE): all contours with their points, in
 * discovery (raster) order -- cv2 returns the list reversed.  Used by the calibration restatement
 * (reference: meterelf/_calibration.py:47-48). */
ORC_API int orc_external_contours(const uint8_t* bin, int rows, int cols, int32_t* pts, int max_pts,
                                  int32_t* counts, int max_contours)
{
    contour_rec* recs = (contour_rec*)malloc(sizeof(contour_rec) * (size_t)(max_contours > 0 ? max_contours : 1));
    int n = find_external_contours_pts(bin, rows, cols, recs, max_contours, pts, max_pts, counts);
    free(recs);
    return n;
}

/* exported for unit tests of the contour semantics */
ORC_API int orc_largest_contour(const uint8_t* bin, int rows, int cols, double* area, uint8_t* filled)
{
    int cap = rows * cols / 2 + 4;
    contour_rec* recs = (contour_rec*)malloc(sizeof(contour_rec) * cap);
    int n = find_external_contours(bin, rows, cols, recs, cap);
    if (n == 0) { free(recs); *area = 0; return 0; }
    /* python: sorted(contours, key=contourArea)[-1]; cv2 lists contours in
     * reverse discovery order and sorted() is stable, so among equal areas the
     * earliest discovered (raster-first) wins */
    if (g_opt[ORC_OPT_AREA_RULE]) { /* audit: area = number of pixels the filled contour paints */
        uint8_t* tmp = (uint8_t*)malloc((size_t)rows * cols);
        for (int k = 0; k < n && k < cap; ++k) {
            fill_external_contour(bin, rows, cols, recs[k].start, tmp);
            long cnt = 0;
            for (long q = 0; q < (long)rows * cols; ++q) cnt += tmp[q] != 0;
            recs[k].area = (double)cnt;
        }
        free(tmp);
    }
    int best = 0;
    for (int k = 1; k < n; ++k)
        if (recs[k].area > recs[best].area || (g_opt[ORC_OPT_CONTOUR_TIE] && recs[k].area == recs[best].area)) best = k;
    *area = recs[best].area;
    if (filled) fill_external_contour(bin, rows, cols, recs[best].start, filled);
    free(recs);
    return n;
}

/* ------------------------------------------------------------------ */
/* angle helpers                                                        */
/* reference: meterelf/_utils.py:18-42                                  */
/* ------------------------------------------------------------------ */

static double py_fmod(double a, double b)
{
    /* CPython float_rem */
    double m = fmod(a, b);
    if (m) {
        if ((b < 0) != (m < 0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

/* returns 1 and writes *out, or 0 for None */
ORC_API int orc_angle_by_vector(double x, double y, double* out)
{
    if (y == 0) {
        if (x > 0) { *out = 0.25; return 1; }
        if (x < 0) { *out = 0.75; return 1; }
        return 0;
    }
    double at = atan(x / y) / (2 * M_PI);
    *out = py_fmod(-at + (y > 0 ? 0.5 : 0.0), 1.0);
    return 1;
}

/* reference: meterelf/_reading.py:163-182 ; r[] ordered (r4, r3, r2, r1) =
 * positions sorted by dial-name string */
ORC_API double orc_value_by_positions(const double* r)
{
    double r4 = r[0], r3 = r[1], r2 = r[2], r1 = r[3];
    int d3 = (int)r3 + ((py_fmod(r3, 1.0) > 0.55 && r4 <= 2) ? 1 : 0) - ((py_fmod(r3, 1.0) < 0.45 && r4 >= 8) ? 1 : 0);
    d3 = ((d3 % 10) + 10) % 10;
    int d2 = (int)r2 + ((py_fmod(r2, 1.0) > 0.55 && d3 <= 2) ? 1 : 0) - ((py_fmod(r2, 1.0) < 0.45 && d3 >= 8) ? 1 : 0);
    d2 = ((d2 % 10) + 10) % 10;
    int d1 = (int)r1 + ((py_fmod(r1, 1.0) > 0.55 && d2 <= 2) ? 1 : 0) - ((py_fmod(r1, 1.0) < 0.45 && d2 >= 8) ? 1 : 0);
    d1 = ((d1 % 10) + 10) % 10;
    return (d1 * 100.0) + (d2 * 10.0) + (d3 * 1.0) + r4 / 10.0;
}

typedef struct { double a, d; } ang_sq;

static int cmp_ang_sq(const void* pa, const void* pb)
{
    const ang_sq* a = (const ang_sq*)pa;
    const ang_sq* b = (const ang_sq*)pb;
    if (a->a < b->a) return -1;
    if (a->a > b->a) return 1;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* per-dial reading                                                     */
/* reference: meterelf/_reading.py:28-96 and :118-160                   */
/* ------------------------------------------------------------------ */

/* dials_hls: th x tw x 3 packed.  Returns 0 ok, 1 = no contours
 * (NeedleContoursNotFoundError), 2 = unreadable (no kept angles). */
static int read_one_dial(const uint8_t* dials_hls, int th, int tw, const orc_dial* dd,
                         const uint8_t* disk, const uint8_t* annulus, int d, orc_result* res)
{
    long n = (long)th * tw;
    /* get_dial_color, _reading.py:154-160 */
    int x0 = (int)dd->cx, y0 = (int)dd->cy; /* int() truncation */
    double sum[3] = {0, 0, 0};
    int cnt = 0;
    for (int y = y0 - 2; y < y0 + 3; ++y)
        for (int x = x0 - 2; x < x0 + 3; ++x) {
            if (x < 0 || x >= tw || y < 0 || y >= th) continue; /* numpy slice clamps (for x0-2 >= 0) */
            for (int c = 0; c < 3; ++c) sum[c] += dials_hls[((long)y * tw + x) * 3 + c];
            ++cnt;
        }
    int col[3];
    for (int c = 0; c < 3; ++c) {
        double mean = cnt ? (g_opt[ORC_OPT_MEAN_FORM] ? sum[c] / cnt : sum[c] * (1.0 / cnt)) : 0.0; /* cv::mean: sum * (1./N) */
        col[c] = (int)py_round(mean);
        res->dial_color[d][c] = col[c];
    }
    /* HlsColor.get_range, _colors.py:38-50 */
    int rng[3] = {dd->range_h, dd->range_l, dd->range_s};
    int lo[3], hi[3];
    for (int c = 0; c < 3; ++c) {
        lo[c] = col[c] - rng[c] > 0 ? col[c] - rng[c] : 0;
        hi[c] = col[c] + rng[c] < 255 ? col[c] + rng[c] : 255;
    }
    uint8_t* m0 = (uint8_t*)malloc(n);
    uint8_t* mde = (uint8_t*)malloc(n);
    uint8_t* md = (uint8_t*)malloc(n);
    uint8_t* needle = (uint8_t*)malloc(n);
    orc_inrange(dials_hls, th, tw, lo, hi, m0);
    orc_close3(m0, th, tw, mde);
    for (long p = 0; p < n; ++p) md[p] = mde[p] & disk[p];
    double area = 0;
    int ncont = orc_largest_contour(md, th, tw, &area, needle);
    res->contour_area[d] = area;
    int rc = 0;
    if (ncont == 0) {
        rc = 1;
        goto done;
    }
    if (!(area > 100)) memcpy(needle, mde, n); /* needle_mask = needle_mask_de, _reading.py:147-148 */

    /* momentum, _reading.py:32-41 ; find_non_zero is raster order */
    double mx = 0.0, my = 0.0;
    int nn = 0;
    for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x) {
            long p = (long)y * tw + x;
            if (!(needle[p] & disk[p])) continue;
            double dx = (double)x - dd->cx, dy = (double)y - dd->cy;
            mx += (dx < 0 ? -1 : 1) * (dx * dx);
            my += (dy < 0 ? -1 : 1) * (dy * dy);
            ++nn;
        }
    res->n_needle[d] = nn;
    int msign = dd->negative_momentum ? -1 : 1;
    double mom_angle = 0;
    int have_mom = orc_angle_by_vector(msign * mx, msign * my, &mom_angle);

    /* ring angles, _reading.py:51-69 */
    ang_sq* as = (ang_sq*)malloc(sizeof(ang_sq) * (size_t)n);
    int na = 0, nouter = 0;
    for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x) {
            long p = (long)y * tw + x;
            if (!(needle[p] & annulus[p])) continue;
            ++nouter;
            double dx = (double)x - dd->cx, dy = (double)y - dd->cy;
            double a;
            if (orc_angle_by_vector(dx, dy, &a) && have_mom) {
                double dist = fabs(a - mom_angle);
                double dist2 = fabs(fabs(a - mom_angle) - 1);
                if (dist2 < dist) dist = dist2;
                if (dist < 0.25) {
                    as[na].a = a;
                    as[na].d = dx * dx + dy * dy;
                    ++na;
                }
            }
        }
    res->n_outer[d] = nouter;
    res->n_kept[d] = na;
    if (na == 0) {
        rc = 2;
        free(as);
        goto done;
    }
    /* unwrap / trim / weighted mean, _reading.py:82-96 */
    double min_angle = as[0].a;
    for (int k = 1; k < na; ++k)
        if (as[k].a < min_angle) min_angle = as[k].a;
    for (int k = 0; k < na; ++k)
        if (!(fabs(as[k].a - min_angle) < 0.75)) as[k].a = as[k].a - 1;
    int b = 0, e = na;
    if (na >= 5) {
        int cut = (na - 3) / 2 < 2 ? (na - 3) / 2 : 2;
        qsort(as, na, sizeof(ang_sq), cmp_ang_sq);
        b = cut;
        e = na - cut;
    }
    double sad = 0, sd = 0; /* sum() starts from int 0 and adds left to right */
    for (int k = b; k < e; ++k) sad += as[k].a * as[k].d;
    for (int k = b; k < e; ++k) sd += as[k].d;
    double angle = sad / sd;
    double fixed = angle - (dd->angle_of_zero / 360.0);
    res->angle[d] = angle;
    res->pos[d] = py_fmod(10.0 * fixed, 10.0);
    free(as);
done:
    free(m0); free(mde); free(md); free(needle);
    return rc;
}

/* Stage entry: read all dials from an already-cropped dials HLS image. */
ORC_API int orc_read_dials(const uint8_t* dials_hls, const orc_params* P, const uint8_t* masks,
                           const int* name_order /* dial indices sorted by name string */, orc_result* res)
{
    long n = (long)P->th * P->tw;
    res->unreadable_mask = 0;
    res->failed_dial = -1;
    for (int d = 0; d < P->ndials; ++d) {
        const uint8_t* disk = masks + (long)d * 2 * n;
        int rc = read_one_dial(dials_hls, P->th, P->tw, &P->dial[d], disk, disk + n, d, res);
        if (rc == 1) {
            res->status = ORC_NEEDLE_CONTOURS_NOT_FOUND;
            res->failed_dial = d;
            return res->status;
        }
        if (rc == 2) res->unreadable_mask |= 1u << d;
    }
    if (res->unreadable_mask) {
        res->status = ORC_ANGLE_UNDETERMINED;
        return res->status;
    }
    res->status = ORC_OK;
    if (P->ndials == 4) { /* determine_value_by_dial_positions asserts len == 4 */
        double r[4];
        for (int k = 0; k < 4; ++k) r[k] = res->pos[name_order[k]];
        res->value = orc_value_by_positions(r);
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* whole frame: get_meter_value on a meter_rect crop                    */
/* reference: meterelf/_reading.py:19-115, meterelf/_image.py:23-66     */
/* ------------------------------------------------------------------ */

ORC_API int orc_process_crop(const uint8_t* crop_bgr, int rows, int cols, long stride,
                             const orc_params* P, const uint8_t* tpl, const uint8_t* masks,
                             const int* name_order, orc_result* res)
{
    memset(res, 0, sizeof(*res));
    res->failed_dial = -1;
    uint8_t* hls = (uint8_t*)malloc((size_t)rows * cols * 3);
    uint8_t* lplane = (uint8_t*)malloc((size_t)rows * cols);
    orc_bgr2hls_full(crop_bgr, rows, cols, stride, P->hue_shift, hls);
    for (long p = 0; p < (long)rows * cols; ++p) lplane[p] = hls[3 * p + 1];
    float mv;
    int mx, my;
    orc_match_ccoeff(lplane, rows, cols, cols, tpl, P->th, P->tw, NULL, &mv, &mx, &my);
    res->match_val = mv;
    res->match_x = mx;
    res->match_y = my;
    int rc;
    if ((double)mv < P->match_threshold) {
        res->status = ORC_DIALS_NOT_FOUND;
        rc = res->status;
    } else {
        uint8_t* dials = (uint8_t*)malloc((size_t)P->th * P->tw * 3);
        for (int y = 0; y < P->th; ++y)
            memcpy(dials + (long)y * P->tw * 3, hls + ((long)(my + y) * cols + mx) * 3, (size_t)P->tw * 3);
        rc = orc_read_dials(dials, P, masks, name_order, res);
        free(dials);
    }
    free(hls);
    free(lplane);
    return rc;
}

/* Batch of full frames (NHWC BGR u8): crop meter_rect then process.  Frames
 * are independent (meterelf/_api.py:22-33). */
ORC_API void orc_process_frames(const uint8_t* frames, int nframes, int H, int W, long frame_stride,
                                int rx0, int ry0, int rx1, int ry1, const orc_params* P,
                                const uint8_t* tpl, const uint8_t* masks, const int* name_order,
                                orc_result* res)
{
    /* numpy slicing img[y0:y1, x0:x1] clamps to the image */
    int x0 = rx0 < W ? rx0 : W, x1 = rx1 < W ? rx1 : W, y0 = ry0 < H ? ry0 : H, y1 = ry1 < H ? ry1 : H;
    for (int f = 0; f < nframes; ++f) {
        const uint8_t* fr = frames + (long)f * frame_stride;
        orc_process_crop(fr + ((long)y0 * W + x0) * 3, y1 - y0, x1 - x0, (long)W * 3, P, tpl, masks,
                         name_order, &res[f]);
    }
}

ORC_API int orc_sizeof_params(void) { return (int)sizeof(orc_params); }
ORC_API int orc_sizeof_result(void) { return (int)sizeof(orc_result); }
