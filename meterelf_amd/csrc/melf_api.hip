// Host side of libmeterelf_hip: calibration blob, per-GPU context, entry points.
// The C ABI is declared and documented in include/meterelf_hip.h.
#include <emmintrin.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "melf_device.h"
#include "melf_internal.h"
#include "melf_threads.h"

using namespace melf;

// ---------------------------------------------------------------- errors ----
static thread_local std::string g_err;

static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(MELF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

extern "C" const char* melf_last_error(void) { return g_err.c_str(); }
extern "C" int melf_abi_version(void) { return MELF_ABI_VERSION; }
extern "C" int melf_device_count(int* count)
{
    if (!count) return fail(MELF_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(MELF_ERR_NO_DEVICE, hipGetErrorString(e)); }
    *count = n;
    return MELF_SUCCESS;
}
extern "C" const char* melf_kernel_name(int k)
{
    static const char* names[MELF_K_COUNT] = {"k_lplane", "k_match", "k_dials", "k_fused_mask", "k_bgr2hls",
                                               "k_jpeg_huff", "k_jpeg_idct", "k_jpeg_color", "k_stream_probe"};
    return (k >= 0 && k < MELF_K_COUNT) ? names[k] : "?";
}

static int check_params(const melf_params* p)
{
    if (!p) return fail(MELF_ERR_INVALID, "params is NULL");
    if (p->abi_version != MELF_ABI_VERSION) return fail(MELF_ERR_INVALID, "melf_params.abi_version mismatch");
    if (p->ndials < 1 || p->ndials > MELF_MAX_DIALS) return fail(MELF_ERR_INVALID, "ndials must be 1..8");
    if (p->th < 1 || p->tw < 2) return fail(MELF_ERR_INVALID, "bad template size (at least 1 row of 2 pixels)");
    return MELF_SUCCESS;
}

// ------------------------------------------------- dial masks (host, a5) ----
// Restates cv2.circle (thickness 1) + cv2.floodFill (4-connected, exact value)
// as the reference combines them in meterelf/_dial_data.py:22-48.

static inline void plot(uint8_t* img, int rows, int cols, int x, int y)
{
    if ((unsigned)x < (unsigned)cols && (unsigned)y < (unsigned)rows) img[(size_t)y * cols + x] = 255;
}

// integer midpoint circle exactly as OpenCV's Circle() walks it
static void circle_outline(uint8_t* img, int rows, int cols, int cx, int cy, int radius)
{
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        plot(img, rows, cols, cx - dx, cy - dy); plot(img, rows, cols, cx - dx, cy + dy);
        plot(img, rows, cols, cx + dx, cy - dy); plot(img, rows, cols, cx + dx, cy + dy);
        plot(img, rows, cols, cx - dy, cy - dx); plot(img, rows, cols, cx - dy, cy + dx);
        plot(img, rows, cols, cx + dy, cy - dx); plot(img, rows, cols, cx + dy, cy + dx);
        ++dy;
        err += plus;
        plus += 2;
        const int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

// scanline flood fill: 4-connected region of pixels equal to the seed value that
// are not blocked by `blocked`; filled pixels become 255 and blocked.
static void flood_fill_scanline(uint8_t* img, std::vector<uint8_t>& blocked, int rows, int cols, int sx, int sy)
{
    if ((unsigned)sx >= (unsigned)cols || (unsigned)sy >= (unsigned)rows) return;
    const uint8_t v0 = img[(size_t)sy * cols + sx];
    auto open = [&](int x, int y) {
        return !blocked[(size_t)y * cols + x] && img[(size_t)y * cols + x] == v0;
    };
    if (!open(sx, sy)) return;
    std::vector<std::pair<int, int>> todo;
    todo.emplace_back(sx, sy);
    while (!todo.empty()) {
        auto [x, y] = todo.back();
        todo.pop_back();
        if (!open(x, y)) continue;
        int xl = x, xr = x;
        while (xl > 0 && open(xl - 1, y)) --xl;
        while (xr + 1 < cols && open(xr + 1, y)) ++xr;
        for (int i = xl; i <= xr; ++i) blocked[(size_t)y * cols + i] = 1;
        for (int ny = y - 1; ny <= y + 1; ny += 2) {
            if (ny < 0 || ny >= rows) continue;
            bool run = false;
            for (int i = xl; i <= xr; ++i) {
                const bool o = open(i, ny);
                if (o && !run) todo.emplace_back(i, ny);
                run = o;
            }
        }
        for (int i = xl; i <= xr; ++i) img[(size_t)y * cols + i] = 255;
    }
}

static int py_round_int(double v) { return (int)nearbyint(v); }  // Python round(): half to even

extern "C" int melf_build_dial_masks(const melf_params* p, uint8_t* masks)
{
    if (int rc = check_params(p)) return rc;
    if (!masks) return fail(MELF_ERR_INVALID, "masks is NULL");
    const int th = p->th, tw = p->tw;
    const size_t n = (size_t)th * tw;
    for (int d = 0; d < p->ndials; ++d) {
        const melf_dial& D = p->dial[d];
        uint8_t* mask = masks + (size_t)d * 2 * n;
        uint8_t* circle_mask = mask + n;
        memset(mask, 0, n);
        const int dial_radius = py_round_int(D.diameter / 2.0);
        const int cx = py_round_int(D.cx), cy = py_round_int(D.cy);
        const int start_radius = dial_radius + D.dist_from_center;
        circle_outline(mask, th, tw, cx, cy, start_radius);
        circle_outline(mask, th, tw, cx, cy, start_radius + D.circle_thickness - 1);
        std::vector<uint8_t> blocked(n, 0);
        if (cx + start_radius + 1 < 0 || cx + start_radius + 1 >= tw || cy < 0 || cy >= th)
            return fail(MELF_ERR_INVALID, "dial annulus seed point lies outside the dials template");
        flood_fill_scanline(mask, blocked, th, tw, cx + start_radius + 1, cy);
        memcpy(circle_mask, mask, n);
        if (cx < 0 || cx >= tw) return fail(MELF_ERR_INVALID, "dial centre lies outside the dials template");
        flood_fill_scanline(mask, blocked, th, tw, cx, cy);
    }
    return MELF_SUCCESS;
}

// ------------------------------------------------------------------ blob ----
struct BlobHeader {
    uint32_t magic;     // 'MELF'
    uint32_t version;
    uint64_t total;
    melf_params params;
};
static const uint32_t BLOB_MAGIC = 0x464c454du;

extern "C" size_t melf_blob_size(const melf_params* p)
{
    if (check_params(p)) return 0;
    const size_t n = (size_t)p->th * p->tw;
    return sizeof(BlobHeader) + n + (size_t)p->ndials * 2 * n;
}

extern "C" int melf_blob_pack(const melf_params* p, const uint8_t* templ, void* blob, size_t blob_bytes)
{
    if (int rc = check_params(p)) return rc;
    if (!templ || !blob) return fail(MELF_ERR_INVALID, "NULL argument");
    const size_t need = melf_blob_size(p);
    if (blob_bytes < need) return fail(MELF_ERR_INVALID, "blob buffer too small");
    BlobHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = BLOB_MAGIC;
    h.version = MELF_ABI_VERSION;
    h.total = need;
    h.params = *p;
    uint8_t* b = (uint8_t*)blob;
    memcpy(b, &h, sizeof(h));
    const size_t n = (size_t)p->th * p->tw;
    memcpy(b + sizeof(h), templ, n);
    return melf_build_dial_masks(p, b + sizeof(h) + n);
}

static int blob_check(const void* blob, size_t blob_bytes, BlobHeader* h)
{
    if (!blob || blob_bytes < sizeof(BlobHeader)) return fail(MELF_ERR_INVALID, "blob too small");
    memcpy(h, blob, sizeof(*h));
    if (h->magic != BLOB_MAGIC || h->version != MELF_ABI_VERSION) return fail(MELF_ERR_INVALID, "not a meterelf blob");
    if (int rc = check_params(&h->params)) return rc;
    if (h->total != melf_blob_size(&h->params) || blob_bytes < h->total) return fail(MELF_ERR_INVALID, "blob size mismatch");
    return MELF_SUCCESS;
}

extern "C" int melf_blob_params(const void* blob, size_t blob_bytes, melf_params* out)
{
    BlobHeader h;
    if (int rc = blob_check(blob, blob_bytes, &h)) return rc;
    if (out) *out = h.params;
    return MELF_SUCCESS;
}

enum { MK_DOT4 = MELF_MATCH_KERNEL_DOT4, MK_FAST = MELF_MATCH_KERNEL_MFMA, MK_GEN = MELF_MATCH_KERNEL_GEN };

// --------------------------------------------------------------- context ----
struct TimedEvent {
    int kernel;
    hipEvent_t start, stop;
};

struct melf_ctx {
    int device = 0;
    melf_params P;
    std::vector<uint8_t> h_masks;   // [ndials][2][th*tw]
    std::vector<uint8_t> h_templ;
    MatchGeom mg;
    uint32_t* d_tplT = nullptr;
    DialGeom* d_geom = nullptr;
    int ws_max = 0;                      // largest dial window (rows) of the context
    uint64_t* d_rowmasks = nullptr;
    int8_t* d_atab = nullptr;            // Toeplitz template fragments of the MFMA match (NULL: template shape unsupported)
    long tsum = 0;
    bool use_mfma = true;                // MELF_MATCH=dot4 forces the VALU kernel
    int force_kind = -1;                 // MELF_MATCH=fast / gen: force the tuned / the general matrix-core kernel where it can run
    // plans of the general matrix-core kernel, one per (crop shape, frame groups) seen (read-only once built: both pipeline
    // lanes launch from the same entry)
    struct GenEntry {
        int rows, cols, groups;
        GenPlan plan;
        GenDev dev;
    };
    std::vector<GenEntry*> gen_cache;
    // pipeline lanes: sets of work buffers that can be in flight at once (one per caller stream, up to NLANES), so that
    // one batch's VALU-bound kernels (prep, dials) overlap another batch's matrix-core-bound match
    static const int NLANES = 2;   // three or four lanes (and as many caller streams) measured no faster than two
    int active_lane = 0;                 // melf_process_stream_dev: which lane's work buffers the next batch uses
    int lanes = 1;                       // MELF_LANES=2 enables the split (measured slower on MI355X: the two
                                         // half-batch match kernels do not overlap usefully; kept for experiments)
    hipStream_t lane_stream[NLANES] = {};
    hipEvent_t ev_fork = nullptr, ev_join[NLANES] = {};
    int8_t* d_lg[NLANES] = {}; size_t lg_cap[NLANES] = {};
    uint16_t* d_rsum[NLANES] = {}; size_t rsum_cap[NLANES] = {};
    MatchPartial* d_lpart[NLANES] = {}; size_t lpart_cap[NLANES] = {};
    uint32_t* d_fused_tables = nullptr;  // K1b lookup tables (built on the GPU at creation)
    int fused_ambiguous = 0;             // hue-table entries whose answer depends on float32 rounding of the triple
    int fused_active_sectors = 0;        // bit c: hue sector c (max = r/g/b) has in-range entries
    int fused_variant = 3;
    hipStream_t stream = nullptr;
    // workspaces (grown on demand)
    melf_result* d_results = nullptr;
    size_t results_cap = 0;
    uint8_t* d_stage_in = nullptr;
    size_t stage_in_cap = 0;
    // host-fed path (melf_process_batch): two pinned staging buffers, a copy stream and the packed crops in HBM
    uint8_t* h_pin[2] = {nullptr, nullptr};
    size_t pin_cap[2] = {0, 0};
    hipEvent_t ev_h2d[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    uint8_t* d_crops = nullptr;
    size_t crops_cap = 0;
    uint8_t* d_stage_out = nullptr;
    size_t stage_out_cap = 0;
    JpegWorkspace* jpeg = nullptr;     // created by the first JPEG batch
    // pipelined JPEG path (melf_jpeg_process_batch): a second workspace and decode stream, so that the host prepares chunk
    // k + 1 and its upload runs while chunk k decodes; per-chunk events; the files' decode status in pinned memory
    static const int NJ = 4;           // workspaces in the ring (a 1024-file call in 256-file chunks never waits for one)
    JpegWorkspace* jpeg_ring[NJ] = {};  // [0] is unused: slot 0 is `jpeg` itself
    hipStream_t jpeg_stream[NJ] = {};   // [0] unused: slot 0 decodes on the context's stream
    hipEvent_t ev_jup[NJ] = {}, ev_jdec[NJ] = {};
    // per CALL slot (calls of the file-name API overlap: the next one's preparation and uploads run while the previous
    // one's kernels do): the decoded frames, the records and the files' decode status in pinned memory, "call done".
    // As many slots as calls in flight: call k + NJC can only be begun after call k's _end.
    static const int NJC = MELF_FILES_IN_FLIGHT_MAX;
    uint8_t* d_jframes[NJC] = {};
    size_t jframes_cap[NJC] = {};
    melf_result* d_jresults[NJC] = {};
    size_t jresults_cap[NJC] = {};
    int32_t* h_jstatus[NJC] = {};
    size_t jstatus_cap[NJC] = {};
    hipEvent_t ev_jcall[NJC] = {};
    uint64_t jpeg_chunk_seq = 0;       // chunks decoded since the context was created: chunk q uses ring slot q % NJ
    // melf_jpeg_process_files: the files' bytes, in PINNED host memory (round 4: the upload starts from here, the host copies
    // no byte of a file after read() has; grow-only: no per-file allocation, no zero fill, and after the first call no fresh
    // pages to fault in or to pin), and the call in flight of the begin / end pair
    // Up to NFJ begin / end calls in flight: each on its own thread, which READS its files at once (into the arena of its
    // slot) and then waits for its turn at the context (decode + reading path, in the order of the _begin calls).
    // Three stages, three calls: one reads its files, one prepares and enqueues, one waits for its kernels (with two, the
    // file reads of call k + 2 could only start when call k was over and the GPU sat idle meanwhile).
    static const int NFJ = MELF_FILES_IN_FLIGHT_MAX;
    struct FilesJob {
        std::thread th;
        int rc = 0;
        std::string err;
    };
    uint8_t* file_arena[NFJ] = {};
    size_t file_arena_cap[NFJ] = {};
    JpegParsed* files_parsed[NFJ] = {};     // the slot's parsed headers + decode data (re-used call after call)
    size_t file_arena_per_file[NFJ] = {};   // arena bytes per file of the slot's last call: sizes the arena for the next one
    int file_arena_oversized[NFJ] = {};     // consecutive calls that needed less than a quarter of the slot's arena
    std::deque<FilesJob*> files_jobs;      // oldest first
    uint64_t files_next_ticket = 0;        // of the next _begin
    uint64_t files_decode_turn = 0;        // the ticket whose decode stage may run
    std::mutex files_m;
    std::condition_variable files_cv;
    // where the file-name calls spend their host time (melf_ctx_files_stats; bench.py's jpeg_decode.get_meter_values.host):
    // sums over the _begin calls since the last reset, milliseconds, under files_m
    struct FilesStats {
        double calls = 0, files = 0, ms_read = 0, ms_turn_wait = 0, ms_enqueue = 0, ms_gpu_wait = 0;
    } files_stats;
    // profiling
    bool force_generic_mask = false;  // MELF_FORCE_GENERIC_MASK=1: float path for every shape (tests)
    int profiling = 0;                // 0 off, 1 every kernel, 2 only the dominant kernel (k_match)
    // The work buffers belong to the context's pipeline lanes, and a lane serves one caller stream at a time: calls that
    // arrive on two different streams run on the two lanes and overlap on the GPU (one batch's prep / dials kernels in
    // the shadow of the other's match kernel); a third stream, or a call that needs a particular lane, first waits for
    // what the lane's previous stream enqueued (claim_lane).
    // "Frames resident" (melf_ctx_set_frames_resident): the caller guarantees that the frames of a *_dev call are complete in
    // memory when the call is made.  Consecutive calls -- even on ONE caller stream -- then alternate between the context's
    // two lanes, each call's kernels on its lane's own stream: prep and match start at once (they read only the frames
    // and write only the lane's work buffers), the dials kernel (the one that writes the caller's records) first waits for
    // everything the caller's stream held at the time of the call, and the caller's stream continues when the call is done.
    // So a call's prep and match run under the previous call's match tail and dials kernel, as with two caller streams.
    bool frames_resident = false;
    int resident_next_lane = 0;
    hipEvent_t ev_call[NLANES] = {};     // caller-stream position at the time of a call (waited for by its dials kernel)
    hipStream_t order_stream = nullptr;  // during a resident-mode call: the caller's stream (NULL stream: see order_null)
    bool order_valid = false;
    hipStream_t lane_owner[NLANES] = {};
    bool lane_owned[NLANES] = {};
    uint64_t lane_used[NLANES] = {};
    uint64_t use_clock = 0;
    hipEvent_t ev_order = nullptr;
    std::vector<TimedEvent> events;
    melf_match_info last_match = {};     // what run_match launched last (melf_ctx_last_match)
    double acc_ms[MELF_K_COUNT] = {0};
    int64_t acc_n[MELF_K_COUNT] = {0};
};

// Hands lane l to stream st: if another stream used the lane last, st first waits for everything that stream had
// enqueued by now.  Costs nothing while the lane stays with one stream.
static int claim_lane(melf_ctx* c, int l, hipStream_t st)
{
    if (c->lane_owned[l] && c->lane_owner[l] != st) {
        if (!c->ev_order) HIP_TRY(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
        // a stream the caller has destroyed meanwhile has nothing pending: a failed record is not an error
        if (hipEventRecord(c->ev_order, c->lane_owner[l]) == hipSuccess) HIP_TRY(hipStreamWaitEvent(st, c->ev_order, 0));
        else (void)hipGetLastError();
    }
    c->lane_owner[l] = st;
    c->lane_owned[l] = true;
    c->lane_used[l] = ++c->use_clock;
    return MELF_SUCCESS;
}
// The lane st already owns, else a free one, else the least recently used one.
static int acquire_lane(melf_ctx* c, hipStream_t st, int* lane)
{
    int pick = -1;
    for (int l = 0; l < melf_ctx::NLANES; ++l)
        if (c->lane_owned[l] && c->lane_owner[l] == st) { pick = l; break; }
    if (pick < 0)
        for (int l = 0; l < melf_ctx::NLANES; ++l)
            if (!c->lane_owned[l]) { pick = l; break; }
    if (pick < 0) {
        pick = 0;
        for (int l = 1; l < melf_ctx::NLANES; ++l)
            if (c->lane_used[l] < c->lane_used[pick]) pick = l;
    }
    *lane = pick;
    return claim_lane(c, pick, st);
}
static int claim_all_lanes(melf_ctx* c, hipStream_t st)
{
    for (int l = 0; l < melf_ctx::NLANES; ++l)
        if (int rc = claim_lane(c, l, st)) return rc;
    return MELF_SUCCESS;
}

// Work buffers only ever grow.  A buffer that is replaced may still be read by kernels in flight -- another caller stream's
// batch on the other lane, or the previous melf_jpeg_process_files_begin call's kernels -- so the whole device is drained
// first (explicitly: hipFree happens to synchronise, but nothing promises it).  Growth is rare (first use of a size).
template <class T>
static int grow(T** ptr, size_t* cap, size_t need)
{
    if (need <= *cap) return MELF_SUCCESS;
    if (*ptr) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(*ptr));
    }
    *ptr = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc((void**)ptr, need * sizeof(T)));
    *cap = need;
    return MELF_SUCCESS;
}

struct KernelTimer {
    melf_ctx* c;
    hipStream_t s;
    TimedEvent ev;
    bool on;
    KernelTimer(melf_ctx* ctx, int k, hipStream_t st) : c(ctx), s(st), on(ctx->profiling == 1 || (ctx->profiling == 2 && k == MELF_K_MATCH))
    {
        if (!on) return;
        ev.kernel = k;
        if (hipEventCreateWithFlags(&ev.start, hipEventDisableSystemFence) != hipSuccess ||
            hipEventCreateWithFlags(&ev.stop, hipEventDisableSystemFence) != hipSuccess) { on = false; return; }
        hipEventRecord(ev.start, s);
    }
    ~KernelTimer()
    {
        if (!on) return;
        hipEventRecord(ev.stop, s);
        c->events.push_back(ev);
    }
};

static int setup_device_tables(melf_ctx* c)
{
    const melf_params& P = c->P;
    const int th = P.th, tw = P.tw;
    // --- K2 geometry + transposed, row-padded template dwords ---
    MatchGeom& g = c->mg;
    g.th = th;
    g.tw = tw;
    g.tw4 = (tw + 3) / 4;
    g.trows = th + 2 * (MATCH_R - 1);
    const int rem = tw & 3;
    g.last_ones = rem == 0 ? 0x01010101u : (rem == 1 ? 0x00000001u : (rem == 2 ? 0x00000101u : 0x00010101u));
    g.ldsw = MATCH_CBLK / 4 + g.tw4 + 1;
    g.lds_rows = MATCH_RBLK + th - 1;
    if ((size_t)g.ldsw * g.lds_rows * 4 > 64 * 1024)
        return fail(MELF_ERR_TOO_LARGE, "dials template does not fit the match kernel's 64 KiB LDS tile");
    long tsum = 0;
    for (size_t i = 0; i < (size_t)th * tw; ++i) tsum += c->h_templ[i];
    g.tmean = (double)tsum * (1.0 / ((double)th * tw));
    c->tsum = tsum;
    if ((tw + 62) / 32 == 7 && tw <= 192) {  // the MFMA kernel is instantiated for 7 Toeplitz blocks per template row whose first and last pair up (tw 162..192)
        std::vector<int8_t> atab(mfma_atab_bytes(th));
        mfma_build_atab(c->h_templ.data(), th, tw, atab.data());
        HIP_TRY(hipMalloc((void**)&c->d_atab, atab.size()));
        HIP_TRY(hipMemcpy(c->d_atab, atab.data(), atab.size(), hipMemcpyHostToDevice));
    }
    if (const char* e = getenv("MELF_MATCH")) {
        c->use_mfma = strcmp(e, "dot4") != 0;
        if (!strcmp(e, "fast")) c->force_kind = 1;
        if (!strcmp(e, "gen")) c->force_kind = 2;
    }
    std::vector<uint32_t> tplT((size_t)g.tw4 * g.trows, 0u);
    for (int i = 0; i < th; ++i)
        for (int jj = 0; jj < g.tw4; ++jj) {
            uint32_t w = 0;
            for (int k = 0; k < 4; ++k) {
                const int j = jj * 4 + k;
                if (j < tw) w |= (uint32_t)c->h_templ[(size_t)i * tw + j] << (8 * k);
            }
            tplT[(size_t)jj * g.trows + (i + MATCH_R - 1)] = w;
        }
    HIP_TRY(hipMalloc((void**)&c->d_tplT, tplT.size() * 4));
    HIP_TRY(hipMemcpy(c->d_tplT, tplT.data(), tplT.size() * 4, hipMemcpyHostToDevice));

    // --- K3 windows and row bit masks ---
    std::vector<DialGeom> geom(P.ndials);
    // planes per dial: 0 = `mask` (disk), 1 = `circle_mask` (annulus), 2 = window pixels that are
    // 4-connected to the window border through non-disk pixels (the disk mask of a thin ring can have
    // unfilled pockets: they are NOT outside)
    std::vector<uint64_t> rowmasks((size_t)P.ndials * 3 * 64, 0);
    const size_t n = (size_t)th * tw;
    for (int d = 0; d < P.ndials; ++d) {
        const melf_dial& D = P.dial[d];
        const int R = py_round_int(D.diameter / 2.0) + D.dist_from_center + D.circle_thickness - 1;
        const int mcx = py_round_int(D.cx), mcy = py_round_int(D.cy);
        DialGeom& G = geom[d];
        G.ws = 2 * R + 5;
        G.wx0 = mcx - R - 2;
        G.wy0 = mcy - R - 2;
        G.core_x = (int)D.cx;  // int() truncation, meterelf/_reading.py:156
        G.core_y = (int)D.cy;
        if (G.ws > 64) return fail(MELF_ERR_TOO_LARGE, "dial mask radius > 29 px does not fit the 64x64 dial window");
        if (G.ws > c->ws_max) c->ws_max = G.ws;
        const uint8_t* disk = c->h_masks.data() + (size_t)d * 2 * n;
        for (int pl = 0; pl < 2; ++pl)
            for (int y = 0; y < th; ++y)
                for (int x = 0; x < tw; ++x) {
                    if (!disk[pl * n + (size_t)y * tw + x]) continue;
                    const int wy = y - G.wy0, wx = x - G.wx0;
                    // the kernel needs a 2-px margin around the disk inside its window
                    if (wy < 2 || wy >= G.ws - 2 || wx < 2 || wx >= G.ws - 2)
                        return fail(MELF_ERR_TOO_LARGE, "dial mask leaks outside its window (clipped circle?)");
                    rowmasks[((size_t)d * 3 + pl) * 64 + wy] |= 1ull << wx;
                }
        {   // plane 2: flood the complement of the disk from the 64 x 64 window border
            const uint64_t* dk = &rowmasks[((size_t)d * 3 + 0) * 64];
            uint64_t* outp = &rowmasks[((size_t)d * 3 + 2) * 64];
            for (int y = 0; y < 64; ++y) outp[y] = (y == 0 || y == 63) ? ~dk[y] : ((1ull | (1ull << 63)) & ~dk[y]);
            for (bool changed = true; changed;) {
                changed = false;
                for (int y = 0; y < 64; ++y) {
                    const uint64_t o = outp[y];
                    uint64_t nb = (o << 1) | (o >> 1);
                    if (y > 0) nb |= outp[y - 1];
                    if (y < 63) nb |= outp[y + 1];
                    const uint64_t n2 = o | (nb & ~dk[y]);
                    if (n2 != o) { outp[y] = n2; changed = true; }
                }
            }
        }
        if (G.core_x - 2 < G.wx0 || G.core_x + 2 >= G.wx0 + G.ws || G.core_y - 2 < G.wy0 || G.core_y + 2 >= G.wy0 + G.ws)
            return fail(MELF_ERR_INVALID, "dial colour core outside the dial window");
    }
    HIP_TRY(hipMalloc((void**)&c->d_geom, geom.size() * sizeof(DialGeom)));
    HIP_TRY(hipMemcpy(c->d_geom, geom.data(), geom.size() * sizeof(DialGeom), hipMemcpyHostToDevice));
    // behind the masks: the momentum vector's x sums per byte of a row mask (k_dials), momx[d][byte][value] as doubles
    {
        const size_t at = rowmasks.size();
        rowmasks.resize(at + (size_t)P.ndials * 8 * 256, 0);
        for (int d = 0; d < P.ndials; ++d) {
            double f[64];
            for (int x = 0; x < 64; ++x) {
                const double dx = (double)(geom[d].wx0 + x) - P.dial[d].cx;
                f[x] = (dx < 0 ? -1.0 : 1.0) * (dx * dx);
            }
            for (int b = 0; b < 8; ++b)
                for (int v = 0; v < 256; ++v) {
                    double sum = 0.0;
                    for (int j = 0; j < 8; ++j)
                        if (v >> j & 1) sum += f[8 * b + j];
                    memcpy(&rowmasks[at + ((size_t)d * 8 + b) * 256 + v], &sum, 8);
                }
        }
    }
    HIP_TRY(hipMalloc((void**)&c->d_rowmasks, rowmasks.size() * 8));
    HIP_TRY(hipMemcpy(c->d_rowmasks, rowmasks.data(), rowmasks.size() * 8, hipMemcpyHostToDevice));

    return MELF_SUCCESS;
}

// K1b lookup tables for the context's fixed needle bounds: built on the GPU from the exact float path for all 2^24
// BGR triples -- 1.3 ms of kernel plus three synchronising read-backs, which only the fused full-frame stage
// (melf_hls_inrange_close*) needs: built on its first use, not at context creation (the reading path never looks at them).
static int ensure_fused_tables(melf_ctx* c)
{
    if (c->d_fused_tables) return MELF_SUCCESS;
    const melf_params& P = c->P;
    uint32_t* tables = nullptr;
    HIP_TRY(hipMalloc((void**)&tables, (size_t)FUSED_BUF_DWORDS * 4));
    launch_build_fused_tables(P.hue_shift, P.needle_lo, P.needle_hi, tables, c->stream);
    uint32_t namb = 0, active = 0, noniv[3] = {1, 1, 1};
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(&namb, tables + fused_tables_count_offset(), 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&active, tables + fused_tables_active_offset(), 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(noniv, tables + fused_tables_noniv_offset(), 12, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipFree(tables);
        return fail(MELF_ERR_HIP, std::string("building the fused-mask tables: ") + hipGetErrorString(e));
    }
    c->fused_ambiguous = (int)namb;
    c->fused_active_sectors = (int)active;
    // kernel variant: ties -> 4; exactly one hue sector can be in range -> 0/1/2; otherwise 3
    c->fused_variant = namb > 0 ? 4 : (active == 1 ? 0 : (active == 2 ? 1 : (active == 4 ? 2 : 3)));
    if (c->fused_variant < 3 && noniv[c->fused_variant] == 0) c->fused_variant += 6;  // single sector, every row one contiguous run: interval tables
    if (const char* ev = getenv("MELF_FUSED_VARIANT")) {  // tests: "generic" = 3, "ties" = 4
        if (!strcmp(ev, "generic") && namb == 0) c->fused_variant = 3;
        if (!strcmp(ev, "bits") && c->fused_variant >= 6) c->fused_variant -= 6;  // single-sector bit tables
        if (!strcmp(ev, "ties")) c->fused_variant = 4;
#ifdef MELF_DIAG
        if (!strcmp(ev, "memonly")) c->fused_variant = 5;  // timing experiments only: output is garbage
#endif
    }
    c->d_fused_tables = tables;
    return MELF_SUCCESS;
}

extern "C" int melf_ctx_create(int device, const void* blob, size_t blob_bytes, int blob_on_device, melf_ctx** out)
{
    if (!out) return fail(MELF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(MELF_ERR_NO_DEVICE, "no HIP device: libmeterelf_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MELF_ERR_INVALID, "bad device index");
    HIP_TRY(hipSetDevice(device));
    std::vector<uint8_t> host;
    if (blob_on_device) {
        if (!blob || blob_bytes < sizeof(BlobHeader)) return fail(MELF_ERR_INVALID, "blob too small");
        host.resize(blob_bytes);
        HIP_TRY(hipMemcpy(host.data(), blob, blob_bytes, hipMemcpyDeviceToHost));
        blob = host.data();
    }
    BlobHeader h;
    if (int rc = blob_check(blob, blob_bytes, &h)) return rc;
    melf_ctx* c = new melf_ctx();
    c->device = device;
    pool_note_device(device);   // the host pools share the cores out over the devices this process has LIVE contexts on (undone in melf_ctx_destroy)
    if (const char* e = getenv("MELF_FORCE_GENERIC_MASK")) c->force_generic_mask = e[0] == '1';
    c->P = h.params;
    const size_t n = (size_t)c->P.th * c->P.tw;
    const uint8_t* b = (const uint8_t*)blob + sizeof(BlobHeader);
    c->h_templ.assign(b, b + n);
    c->h_masks.assign(b + n, b + n + (size_t)c->P.ndials * 2 * n);
    int rc = MELF_SUCCESS;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(MELF_ERR_HIP, "hipStreamCreate failed");
    for (int l = 0; l < melf_ctx::NLANES && !rc; ++l)
        if (hipStreamCreateWithFlags(&c->lane_stream[l], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join[l], hipEventDisableTiming) != hipSuccess)
            rc = fail(MELF_ERR_HIP, "hipStreamCreate failed");
    if (!rc && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess)
        rc = fail(MELF_ERR_HIP, "hipEventCreate failed");
    if (const char* e = diag_env("MELF_LANES")) c->lanes = atoi(e) == 2 ? 2 : 1;
    if (!rc) rc = setup_device_tables(c);
    if (rc) {
        melf_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return MELF_SUCCESS;
}

// ---- one process, several GPUs: the calibration blob by RCCL broadcast (SURVEY 8b's melf_ctx_bcast, 8e) ----
// RCCL is bound at run time (dlopen: a process that already carries an RCCL -- torch's -- gets that one; a single-GPU user
// of the library never loads it): the five entry points used, with the prototypes of <rccl/rccl.h>.
namespace {
struct Rccl {
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int /* ncclDataType_t */, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};
Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        // The RCCL that belongs to THIS library's HIP runtime: the one next to the libamdhip64 our HIP calls resolve to (a process
        // may hold a second runtime and a second RCCL -- torch ships its own and loads them by path -- and device memory of one
        // runtime is foreign to the other).  RTLD_LOCAL: its symbols must not interpose on another copy loaded later (round 5:
        // with RTLD_GLOBAL a process that imported torch afterwards died in exit() with a double free).
        std::vector<std::string> names;
        Dl_info di;
        if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash);
                names.push_back(dir + "/librccl.so.1");
                names.push_back(dir + "/librccl.so");
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        void* h = nullptr;
        for (const std::string& name : names) {
            h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) { r.why = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "?"); return; }
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
        r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Broadcast;
        if (!r.ok) r.why = "RCCL lacks an entry point";
    });
    return r;
}
}  // namespace

extern "C" int melf_ctx_create_bcast(const int* devices, int n, const void* blob, size_t blob_bytes, melf_ctx** out)
{
    if (!devices || n < 1 || n > 64 || !blob || !out) return fail(MELF_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return fail(MELF_ERR_INVALID, "melf_ctx_create_bcast: a device is listed twice (one RCCL rank per GPU)");
    Rccl& R = rccl();
    if (!R.ok) return fail(MELF_ERR_HIP, "melf_ctx_create_bcast: " + R.why);
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);   // put back at the end: the call must not move the calling thread to another GPU
    std::vector<void*> d_blob(n, nullptr);
    std::vector<hipStream_t> st(n, nullptr);
    std::vector<Rccl::comm_t> comm(n, nullptr);
    int rc = MELF_SUCCESS;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == MELF_SUCCESS) rc = fail(MELF_ERR_HIP, std::string("melf_ctx_create_bcast: ") + what + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    auto nccl_ok = [&](int e, const char* what) {
        if (e != 0 && rc == MELF_SUCCESS) rc = fail(MELF_ERR_HIP, std::string("melf_ctx_create_bcast: ") + what + ": " + (R.GetErrorString ? R.GetErrorString(e) : "RCCL error"));
        return e == 0;
    };
    for (int i = 0; i < n && rc == MELF_SUCCESS; ++i) {
        if (!hip_ok(hipSetDevice(devices[i]), "hipSetDevice")) break;
        if (!hip_ok(hipMalloc(&d_blob[i], blob_bytes), "hipMalloc")) break;
        hip_ok(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking), "hipStreamCreate");
    }
    // rank 0 = devices[0] holds the blob; one ncclBroadcast (ncclUint8, root 0) over xGMI puts it on every other GPU
    if (rc == MELF_SUCCESS) {
        hip_ok(hipSetDevice(devices[0]), "hipSetDevice");
        hip_ok(hipMemcpyAsync(d_blob[0], blob, blob_bytes, hipMemcpyHostToDevice, st[0]), "upload to the root");
        hip_ok(hipStreamSynchronize(st[0]), "upload to the root");
    }
    if (rc == MELF_SUCCESS && nccl_ok(R.CommInitAll(comm.data(), n, devices), "ncclCommInitAll")) {
        nccl_ok(R.GroupStart(), "ncclGroupStart");
        for (int i = 0; i < n && rc == MELF_SUCCESS; ++i)
            nccl_ok(R.Broadcast(d_blob[i], d_blob[i], blob_bytes, 1 /* ncclUint8 */, 0, comm[i], st[i]), "ncclBroadcast");
        nccl_ok(R.GroupEnd(), "ncclGroupEnd");
        for (int i = 0; i < n; ++i) {
            (void)hipSetDevice(devices[i]);
            hip_ok(hipStreamSynchronize(st[i]), "broadcast");
        }
    }
    for (int i = 0; i < n; ++i)
        if (comm[i]) (void)R.CommDestroy(comm[i]);
    // every context from its own GPU's copy of the bytes
    for (int i = 0; i < n && rc == MELF_SUCCESS; ++i) rc = melf_ctx_create(devices[i], d_blob[i], blob_bytes, 1, &out[i]);
    const std::string keep = g_err;
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(devices[i]);
        if (st[i]) (void)hipStreamDestroy(st[i]);
        if (d_blob[i]) (void)hipFree(d_blob[i]);
    }
    if (rc != MELF_SUCCESS) {
        for (int i = 0; i < n; ++i)
            if (out[i]) { melf_ctx_destroy(out[i]); out[i] = nullptr; }
    }
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    return rc != MELF_SUCCESS ? fail(rc, keep) : MELF_SUCCESS;
}

extern "C" void melf_ctx_destroy(melf_ctx* c)
{
    if (!c) return;
    pool_forget_device(c->device);
    for (auto* j : c->files_jobs) { if (j->th.joinable()) j->th.join(); delete j; }
    c->files_jobs.clear();
    for (int a = 0; a < melf_ctx::NFJ; ++a) {
        if (c->file_arena[a]) (void)hipHostFree(c->file_arena[a]);
        if (c->files_parsed[a]) jpeg_parsed_free(c->files_parsed[a]);
    }
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (int l = 0; l < melf_ctx::NLANES; ++l)   // resident mode and the split modes run whole calls on the lanes' own streams
        if (c->lane_stream[l]) hipStreamSynchronize(c->lane_stream[l]);
    for (int b = 1; b < melf_ctx::NJ; ++b)
        if (c->jpeg_stream[b]) hipStreamSynchronize(c->jpeg_stream[b]);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    for (auto& e : c->events) { hipEventDestroy(e.start); hipEventDestroy(e.stop); }
    hipFree(c->d_atab);
    for (auto* ge : c->gen_cache) {
        hipFree(ge->dev.atab); hipFree(ge->dev.atabv); hipFree(ge->dev.tiles);
        delete ge;
    }
    for (int l = 0; l < melf_ctx::NLANES; ++l) {
        hipFree(c->d_lg[l]); hipFree(c->d_rsum[l]); hipFree(c->d_lpart[l]);
        if (c->lane_stream[l]) hipStreamDestroy(c->lane_stream[l]);
        if (c->ev_join[l]) hipEventDestroy(c->ev_join[l]);
    }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_order) hipEventDestroy(c->ev_order);
    for (int l = 0; l < melf_ctx::NLANES; ++l)
        if (c->ev_call[l]) hipEventDestroy(c->ev_call[l]);
    hipFree(c->d_tplT); hipFree(c->d_geom); hipFree(c->d_rowmasks); hipFree(c->d_fused_tables);
    hipFree(c->d_results); hipFree(c->d_stage_in); hipFree(c->d_stage_out);
    hipFree(c->d_crops);
    for (int b = 0; b < 2; ++b) {
        if (c->h_pin[b]) hipHostFree(c->h_pin[b]);
        if (c->ev_h2d[b]) hipEventDestroy(c->ev_h2d[b]);
    }
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    jpeg_workspace_free(c->jpeg);
    for (int b = 0; b < melf_ctx::NJ; ++b) {
        jpeg_workspace_free(c->jpeg_ring[b]);
        if (c->jpeg_stream[b]) hipStreamDestroy(c->jpeg_stream[b]);
        if (c->ev_jup[b]) hipEventDestroy(c->ev_jup[b]);
        if (c->ev_jdec[b]) hipEventDestroy(c->ev_jdec[b]);
    }
    for (int a = 0; a < melf_ctx::NJC; ++a) {
        if (c->h_jstatus[a]) hipHostFree(c->h_jstatus[a]);
        hipFree(c->d_jframes[a]); hipFree(c->d_jresults[a]);
        if (c->ev_jcall[a]) hipEventDestroy(c->ev_jcall[a]);
    }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// Waits for everything the context has enqueued on the caller streams it has been used with, and forgets those
// streams: call it before destroying a stream the context has worked on (a lane would otherwise still name it).
extern "C" int melf_ctx_sync(melf_ctx* c)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    for (int l = 0; l < melf_ctx::NLANES; ++l) {
        if (c->lane_owned[l]) HIP_TRY(hipStreamSynchronize(c->lane_owner[l]));
        c->lane_owned[l] = false;
        c->lane_owner[l] = nullptr;
        HIP_TRY(hipStreamSynchronize(c->lane_stream[l]));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MELF_SUCCESS;
}

extern "C" int melf_ctx_params(const melf_ctx* c, melf_params* out)
{
    if (!c || !out) return fail(MELF_ERR_INVALID, "NULL argument");
    *out = c->P;
    return MELF_SUCCESS;
}

extern "C" int melf_ctx_get_masks(const melf_ctx* c, uint8_t* masks)
{
    if (!c || !masks) return fail(MELF_ERR_INVALID, "NULL argument");
    memcpy(masks, c->h_masks.data(), c->h_masks.size());
    return MELF_SUCCESS;
}

extern "C" int melf_ctx_fused_table_ties(const melf_ctx* c_, int* count)
{
    if (!c_ || !count) return fail(MELF_ERR_INVALID, "NULL argument");
    melf_ctx* c = const_cast<melf_ctx*>(c_);
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = ensure_fused_tables(c)) return rc;
    *count = c->fused_ambiguous;
    return MELF_SUCCESS;
}

extern "C" int melf_ctx_set_frames_resident(melf_ctx* c, int on)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    c->frames_resident = on != 0;
    return MELF_SUCCESS;
}

// the general kernel's plan in melf_match_info terms (melf_ctx_last_match, melf_match_layout_query, melf_match_gen_plan_query)
static void fill_gen_info(melf_match_info* info, const GenPlan& pl)
{
    info->tiles = pl.ntiles; info->waves = pl.ntasks * pl.groups; info->rows_per_wave = pl.rc;
    info->reserved[0] = pl.nd; info->reserved[1] = pl.rows_pad; info->reserved[2] = pl.nxb_tile; info->reserved[3] = pl.nslices;
    info->reserved[4] = pl.vcols; info->reserved[5] = pl.ndv;
}

// Which kernel default dispatch picks for a shape and batch size (no context: assumes the tuned kernel's template tables
// exist whenever the shape is in its class, which is what melf_ctx_create arranges).
static int default_match_kind(int th, int tw, int rows, int cols, int n)
{
    const bool fast_ok = mfma_match_ok(th, tw, rows, cols);
    const bool gen_ok = gen_match_ok(th, tw, rows, cols);
    if (fast_ok) {
        // measured on MI355X, map 132 x 63 (tools/gpu_check_gen.sh): tuned / general kernel 175 / 246 us at 1024 frames,
        // 128 / 161 at 512, 72 / 87 at 256, 62 / 33 at 64; map 17 x 33 at 1024 frames: 67 / 42 us
        // round 5 (K slices: 4-row tiles, 2 or 4 waves per tile; tools/match_sweep.py, profiles/r05/match_sweep_k_slices.txt):
        // tuned / general 42 / 40 us at 128 frames, 44 / 51 at 192, 56 / 66 at 256, 73 / 70 at 320, 86 / 93 at 448, 98 / 153 at 512
        const int rh = rows - th + 1, rw = cols - tw + 1, groups = (n + 31) / 32;
        const MfmaPlan pl = mfma_plan(th, tw, rows, cols, n);
        // round 6 (profiles/r06/match_sweep_round6.txt): tuned / general 33 / 39 us at 128 frames (528 waves), 58 / 64 at 288, 61 / 67 at 320
        const bool few_waves = (long)pl.nparts * pl.groups < 500;          // even in four K slices the tuned kernel leaves half of the chip idle
        const bool small_map = rh < 64 && (long)((rh + 4) / 5) * groups * 2 <= 512;   // a few rows: the general kernel's tile shapes fit them better
        const bool odd_cols = rw > 32 && rw % 32 >= 1 && rw % 32 <= 4;     // a whole column block for <= 4 columns
        if (!gen_ok || !(few_waves || small_map || odd_cols)) return MK_FAST;
    }
    return gen_ok ? MK_GEN : MK_DOT4;
}

extern "C" int melf_match_gen_plan_query(int th, int tw, int rows, int cols, int n, melf_match_info* out, melf_gen_task* tasks,
                                         int cap, int32_t* ntasks)
{
    if (!out || n < 1 || cap < 0 || (cap > 0 && !tasks)) return fail(MELF_ERR_INVALID, "bad argument");
    memset(out, 0, sizeof(*out));
    out->n = n; out->rows = rows; out->cols = cols; out->groups = (n + 31) / 32;
    if (rows < th || cols < tw || th < 1 || tw < 1) return fail(MELF_ERR_INVALID, "image smaller than the template");
    if (!gen_match_ok(th, tw, rows, cols)) return fail(MELF_ERR_TOO_LARGE, "shape outside the general matrix-core kernel's limits");
    const GenPlan pl = gen_plan(th, tw, rows, cols, n);
    out->kernel = default_match_kind(th, tw, rows, cols, n);
    fill_gen_info(out, pl);
    if (ntasks) *ntasks = pl.ntasks;
    // one entry per WAVE: wave w of tile ti's workgroup, with the slice of the tile's K range the kernel gives it
    for (int i = 0; i < pl.ntasks && i < cap; ++i) {
        const int ti = i / pl.nslices, w = i - ti * pl.nslices;
        const GenTile& t = pl.tiles[ti];
        melf_gen_task& o = tasks[i];
        o.y0 = t.y0; o.rows = t.R; o.rows_computed = t.Rc; o.xb0 = t.xb0; o.nxb = t.nxb; o.tile = ti;
        o.slice = w; o.nslices = pl.nslices;
        o.k_lo = (int32_t)((long)w * t.klen / pl.nslices); o.k_hi = (int32_t)((long)(w + 1) * t.klen / pl.nslices);
        o.lds_bytes = (int32_t)pl.lds_bytes; o.reserved = 0;
    }
    return MELF_SUCCESS;
}

extern "C" int melf_match_layout_query(int th, int tw, int rows, int cols, int n, melf_match_info* out)
{
    if (!out || n < 1) return fail(MELF_ERR_INVALID, "bad argument");
    memset(out, 0, sizeof(*out));
    out->n = n; out->rows = rows; out->cols = cols; out->groups = (n + 31) / 32;
    if (rows < th || cols < tw || th < 1 || tw < 1) return fail(MELF_ERR_INVALID, "image smaller than the template");
    if (!mfma_match_ok(th, tw, rows, cols)) {
        out->kernel = gen_match_ok(th, tw, rows, cols) ? MK_GEN : MK_DOT4;
        if (out->kernel == MK_GEN) fill_gen_info(out, gen_plan(th, tw, rows, cols, n));
        return MELF_SUCCESS;
    }
    const MfmaPlan pl = mfma_plan(th, tw, rows, cols, n);
    out->kernel = MK_FAST;
    out->rows_per_wave = pl.rb; out->full_waves = pl.na; out->pair_waves = 2 * pl.np;
    out->waves = pl.nparts * pl.groups;
    out->tiles = pl.ntiles;
    out->reserved[0] = pl.th_pad; out->reserved[1] = pl.rows_pad; out->reserved[2] = pl.ks;
    return MELF_SUCCESS;
}

extern "C" int melf_ctx_last_match(const melf_ctx* c, melf_match_info* out)
{
    if (!c || !out) return fail(MELF_ERR_INVALID, "NULL argument");
    *out = c->last_match;
    return MELF_SUCCESS;
}

extern "C" int melf_ctx_set_profiling(melf_ctx* c, int on)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    // a _begin call's thread appends to the event list while it runs: no profiling calls in between (header: "no other call")
    if (!c->files_jobs.empty()) return fail(MELF_ERR_INVALID, "melf_ctx_set_profiling while a melf_jpeg_process_files_begin call is in flight");
    c->profiling = on < 0 ? 0 : (on > 2 ? 1 : on);
    return MELF_SUCCESS;
}

extern "C" int melf_ctx_timings(melf_ctx* c, double ms[MELF_K_COUNT], int64_t launches[MELF_K_COUNT])
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    if (!c->files_jobs.empty()) return fail(MELF_ERR_INVALID, "melf_ctx_timings while a melf_jpeg_process_files_begin call is in flight");
    HIP_TRY(hipSetDevice(c->device));
    for (auto& e : c->events) {
        HIP_TRY(hipEventSynchronize(e.stop));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e.start, e.stop));
        c->acc_ms[e.kernel] += t;
        c->acc_n[e.kernel] += 1;
        hipEventDestroy(e.start);
        hipEventDestroy(e.stop);
    }
    c->events.clear();
    for (int k = 0; k < MELF_K_COUNT; ++k) {
        if (ms) ms[k] = c->acc_ms[k];
        if (launches) launches[k] = c->acc_n[k];
        c->acc_ms[k] = 0;
        c->acc_n[k] = 0;
    }
    return MELF_SUCCESS;
}


// ------------------------------------------------------------ match stage ----
// Which kernel computes TM_CCOEFF for this shape.  The tuned matrix-core kernel (k_match_mfma) is instantiated for
// templates of 162..193 columns and maps of up to 64 columns; it is the fastest when its 5-row waves fill the chip.
// The general matrix-core kernel (k_match_gen) takes every other shape, and the shapes the tuned one handles badly:
// maps too small to fill the chip (it slices the K loop across waves) and map widths a few columns past 32 (it
// computes those columns in transposed form instead of a whole extra column block).  The VALU kernel (k_match)
// remains for templates wider than 256 columns and as an independent formulation in the tests.
static int pick_match_kind(const melf_ctx* c, int rows, int cols, int n)
{
    if (!c->use_mfma) return MK_DOT4;
    const int th = c->P.th, tw = c->P.tw;
    const bool fast_ok = c->d_atab && mfma_match_ok(th, tw, rows, cols);
    const bool gen_ok = gen_match_ok(th, tw, rows, cols);
    if (c->force_kind == MK_FAST && fast_ok) return MK_FAST;
    if (c->force_kind == MK_GEN && gen_ok) return MK_GEN;
    const int kind = default_match_kind(th, tw, rows, cols, n);
    if (kind == MK_FAST && !fast_ok) return gen_ok ? MK_GEN : MK_DOT4;   // no Toeplitz tables for this template
    return kind;
}

static int gen_entry(melf_ctx* c, int rows, int cols, int n, melf_ctx::GenEntry** out)
{
    const int groups = (n + 31) / 32;
    for (auto* ge : c->gen_cache)
        if (ge->rows == rows && ge->cols == cols && ge->groups == groups) { *out = ge; return MELF_SUCCESS; }
    if (c->gen_cache.size() >= 16) {  // a caller cycling through many shapes: drop the oldest plan
        HIP_TRY(hipDeviceSynchronize());
        auto* old = c->gen_cache.front();
        hipFree(old->dev.atab); hipFree(old->dev.atabv); hipFree(old->dev.tiles);
        delete old;
        c->gen_cache.erase(c->gen_cache.begin());
    }
    // the entry is built completely -- device tables included -- before the cache sees it: a failed allocation or upload
    // must not leave a half-built plan behind for the next call with this shape to launch from
    auto* ge = new melf_ctx::GenEntry();
    ge->rows = rows; ge->cols = cols; ge->groups = groups;
    ge->plan = gen_plan(c->P.th, c->P.tw, rows, cols, n);
    const GenPlan& p = ge->plan;
    if (diag_env("MELF_GEN_TRACE"))
        fprintf(stderr, "[melf gen] crop %dx%d n=%d: map %dx%d nd=%d tiles of %d rows x %d blocks, %d tiles (%d V columns) x %d groups = %d workgroups of %d waves, %zu B of LDS each\n",
                rows, cols, n, p.rh, p.rw, p.nd, p.rc, p.nxb_tile, p.ntiles, p.vcols, p.groups, p.ntiles * p.groups, p.nslices, p.lds_bytes);
    auto upload = [&]() -> int {
        std::vector<int8_t> atab(p.atab_bytes);
        gen_build_atab(c->h_templ.data(), c->P.th, c->P.tw, p, atab.data());
        HIP_TRY(hipMalloc((void**)&ge->dev.atab, atab.size()));
        HIP_TRY(hipMemcpy(ge->dev.atab, atab.data(), atab.size(), hipMemcpyHostToDevice));
        if (p.atabv_bytes) {
            std::vector<int8_t> atabv(p.atabv_bytes);
            gen_build_atabv(c->h_templ.data(), c->P.th, c->P.tw, p, atabv.data());
            HIP_TRY(hipMalloc((void**)&ge->dev.atabv, atabv.size()));
            HIP_TRY(hipMemcpy(ge->dev.atabv, atabv.data(), atabv.size(), hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMalloc((void**)&ge->dev.tiles, p.tiles.size() * sizeof(GenTile)));
        HIP_TRY(hipMemcpy(ge->dev.tiles, p.tiles.data(), p.tiles.size() * sizeof(GenTile), hipMemcpyHostToDevice));
        return MELF_SUCCESS;
    };
    if (int rc = upload()) {
        hipFree(ge->dev.atab); hipFree(ge->dev.atabv); hipFree(ge->dev.tiles);
        delete ge;
        return rc;
    }
    c->gen_cache.push_back(ge);
    *out = ge;
    return MELF_SUCCESS;
}

// prep + match of m images on stream ls with lane bl's work buffers; *parts / *nparts: per-frame (max, first arg-max)
// partials for the consumer (k_dials or the host fold of melf_match_ccoeff)
static int run_match_impl(melf_ctx* c, const MatchSrc& ms, bool from_bgr, int m, int bl, hipStream_t ls, float* d_map,
                          MatchPartial** parts, int* nparts, TimedEvent& ev);
static int run_match(melf_ctx* c, const MatchSrc& ms, bool from_bgr, int m, int bl, hipStream_t ls, float* d_map,
                     MatchPartial** parts, int* nparts)
{
    TimedEvent ev;
    ev.kernel = MELF_K_MATCH;
    ev.start = ev.stop = nullptr;
    const int rc = run_match_impl(c, ms, from_bgr, m, bl, ls, d_map, parts, nparts, ev);
    if (rc == MELF_SUCCESS && ev.start && ev.stop) {
        c->events.push_back(ev);
    } else {   // nothing was launched with them (an allocation failed on the way)
        if (ev.start) hipEventDestroy(ev.start);
        if (ev.stop) hipEventDestroy(ev.stop);
    }
    return rc;
}
static int run_match_impl(melf_ctx* c, const MatchSrc& ms, bool from_bgr, int m, int bl, hipStream_t ls, float* d_map,
                          MatchPartial** parts, int* nparts, TimedEvent& ev)
{
    const melf_params& P = c->P;
    const int kind = pick_match_kind(c, ms.rows, ms.cols, m);
    melf_match_info& info = c->last_match;
    memset(&info, 0, sizeof(info));
    info.kernel = kind; info.n = m; info.rows = ms.rows; info.cols = ms.cols; info.groups = (m + 31) / 32;
    static const bool trace = diag_env("MELF_MATCH_TRACE") != nullptr;
    if (c->profiling && kind != MK_DOT4) {  // the dispatch's own time stamps: no event-record packets around the kernel
        // timing only: without the system-scope fence (cache write-back and invalidate) a default event brings along
        // -- that fence put 7 us in front of the kernel and 5 us behind it (rocprofv3 kernel trace, round 2)
        HIP_TRY(hipEventCreateWithFlags(&ev.start, hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&ev.stop, hipEventDisableSystemFence));
    }
    if (kind == MK_FAST) {
        const MfmaPlan pl = mfma_plan(P.th, P.tw, ms.rows, ms.cols, m);
        *nparts = pl.nparts;
        // the match waves add up their window sums themselves from R (no column-sum launch, no window-sum array)
        if (int rc = grow(&c->d_lg[bl], &c->lg_cap[bl], pl.lg_bytes)) return rc;
        if (int rc = grow(&c->d_rsum[bl], &c->rsum_cap[bl], pl.r_bytes / sizeof(uint16_t))) return rc;
        if (int rc = grow(&c->d_lpart[bl], &c->lpart_cap[bl], (size_t)m * pl.nparts)) return rc;
        *parts = c->d_lpart[bl];
        {
            KernelTimer t(c, MELF_K_LPLANE, ls);
            launch_mfma_prep(ms, from_bgr, m, pl, P.th, P.tw, c->d_lg[bl], c->d_rsum[bl], ls);
        }
        info.rows_per_wave = pl.rb; info.full_waves = pl.na; info.pair_waves = 2 * pl.np;
        info.waves = pl.nparts * pl.groups; info.tiles = pl.ntiles;
        info.reserved[0] = pl.th_pad; info.reserved[1] = pl.rows_pad; info.reserved[2] = pl.ks;
        launch_mfma_match(m, pl, P.th, P.tw, c->tsum, c->mg.tmean, c->d_atab, c->d_lg[bl], (const uint32_t*)c->d_rsum[bl], d_map, *parts, ls,
                          ev.start, ev.stop);
    } else if (kind == MK_GEN) {
        melf_ctx::GenEntry* ge = nullptr;
        if (int rc = gen_entry(c, ms.rows, ms.cols, m, &ge)) return rc;
        const GenPlan& pl = ge->plan;
        *nparts = pl.ntiles;
        if (int rc = grow(&c->d_lg[bl], &c->lg_cap[bl], pl.lg_bytes)) return rc;
        if (int rc = grow(&c->d_rsum[bl], &c->rsum_cap[bl], pl.r_bytes / sizeof(uint16_t))) return rc;
        if (int rc = grow(&c->d_lpart[bl], &c->lpart_cap[bl], (size_t)m * pl.ntiles)) return rc;
        *parts = c->d_lpart[bl];
        {
            KernelTimer t(c, MELF_K_LPLANE, ls);
            launch_match_prep(ms, from_bgr, m, pl.groups, pl.rows_pad, pl.nkb, pl.rwp, P.tw, c->d_lg[bl], c->d_rsum[bl], ls);
        }
        const GenDev& dev = ge->dev;
        fill_gen_info(&info, pl);
        launch_gen_match(m, pl, ms.rows, P.th, P.tw, c->tsum, c->mg.tmean, dev, c->d_lg[bl], c->d_rsum[bl], d_map, *parts, ls, ev.start, ev.stop);
    } else {
        *nparts = match_parts(c->mg, ms.rows, ms.cols);
        if (int rc = grow(&c->d_lpart[bl], &c->lpart_cap[bl], (size_t)m * *nparts)) return rc;
        *parts = c->d_lpart[bl];
        KernelTimer t(c, MELF_K_MATCH, ls);
        launch_match(ms, from_bgr, m, c->mg, c->d_tplT, d_map, *parts, nullptr, ls);
        info.tiles = *nparts;
    }
    if (trace)
        fprintf(stderr, "[melf match] n=%d crop %dx%d: %s, %d groups, %d waves, rows per wave %d, full %d + pair %d per group\n", m, ms.rows,
                ms.cols, kind == MK_FAST ? "k_match_mfma" : (kind == MK_GEN ? "k_match_gen" : "k_match (dot4)"), info.groups, info.waves,
                info.rows_per_wave, info.full_waves, info.pair_waves);
    return MELF_SUCCESS;
}

// ------------------------------------------------------------ full path ----
static const int MAX_FRAMES_PER_LAUNCH = 32768;

// rect (optional): {x0, y0, x1, y1} of the meter crop inside the H x W frames instead of the context's meter_rect
// (the host-fed path uploads only the crop: its "frames" are the crops themselves)
static int process_batch_on(melf_ctx* c, const void* d_frames, int n, int H, int W, size_t frame_stride,
                            void* d_results, melf_result* out_host, hipStream_t st, const int* rect = nullptr, int row_stride = 0);

extern "C" int melf_process_batch_dev(melf_ctx* c, const void* d_frames, int n, int H, int W, size_t frame_stride,
                                      void* d_results, melf_result* out_host, void* stream_)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    if (n < 0 || H <= 0 || W <= 0) return fail(MELF_ERR_INVALID, "bad batch shape");
    if (n == 0) return MELF_SUCCESS;
    if (!d_frames) return fail(MELF_ERR_INVALID, "d_frames is NULL");
    if (frame_stride < (size_t)H * W * 3) return fail(MELF_ERR_INVALID, "frame_stride smaller than a frame");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream_;  // NULL = the null (legacy default) stream, as everywhere in HIP
    if (c->frames_resident && c->lanes == 1 && n <= MAX_FRAMES_PER_LAUNCH) {
        // frames promised complete: the call runs on the next lane's own stream, beside the previous call's kernels on the
        // other lane; only its dials kernel (which writes the records) waits for the caller's stream (process_batch_on)
        const int lane = c->resident_next_lane;
        c->resident_next_lane = (lane + 1) % melf_ctx::NLANES;
        hipStream_t ls = c->lane_stream[lane];
        if (int rc = claim_lane(c, lane, ls)) return rc;
        c->active_lane = lane;
        c->order_stream = st;
        c->order_valid = true;
        const int rc = process_batch_on(c, d_frames, n, H, W, frame_stride, d_results, nullptr, ls);
        c->order_valid = false;
        if (rc) return rc;
        HIP_TRY(hipEventRecord(c->ev_join[lane], ls));
        HIP_TRY(hipStreamWaitEvent(st, c->ev_join[lane], 0));   // what the caller enqueues next sees the records
        if (out_host) {
            const melf_result* res_dev = d_results ? (const melf_result*)d_results : c->d_results;
            HIP_TRY(hipMemcpyAsync(out_host, res_dev, (size_t)n * sizeof(melf_result), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        return MELF_SUCCESS;
    }
    if (c->lanes > 1) {  // the batch is split over both lanes
        if (int rc = claim_all_lanes(c, st)) return rc;
    } else if (int rc = acquire_lane(c, st, &c->active_lane)) {
        return rc;
    }
    return process_batch_on(c, d_frames, n, H, W, frame_stride, d_results, out_host, st);
}

static int process_batch_on(melf_ctx* c, const void* d_frames, int n, int H, int W, size_t frame_stride,
                            void* d_results, melf_result* out_host, hipStream_t st, const int* rect, int row_stride)
{
    if (row_stride <= 0) row_stride = W * 3;  // packed rows unless the caller's rows are padded (host-fed crops)
    const melf_params& P = c->P;
    // numpy slicing img[y0:y1, x0:x1] clamps to the image (meterelf/_image.py:54-55)
    const int rx0 = rect ? rect[0] : P.rect_x0, ry0 = rect ? rect[1] : P.rect_y0;
    const int rx1 = rect ? rect[2] : P.rect_x1, ry1 = rect ? rect[3] : P.rect_y1;
    const int x0 = rx0 < W ? rx0 : W, x1 = rx1 < W ? rx1 : W;
    const int y0 = ry0 < H ? ry0 : H, y1 = ry1 < H ? ry1 : H;
    const int crows = y1 - y0, ccols = x1 - x0;
    if (x0 < 0 || y0 < 0 || crows < P.th || ccols < P.tw)
        return fail(MELF_ERR_INVALID, "meter_rect crop is smaller than the dials template (cv2.matchTemplate would assert)");
    const bool mfma = pick_match_kind(c, crows, ccols, n) != MK_DOT4;
    const int rw = ccols - P.tw + 1;
    melf_result* res_dev = (melf_result*)d_results;
    if (!res_dev) {
        if (int rc = grow(&c->d_results, &c->results_cap, (size_t)n)) return rc;
        res_dev = c->d_results;
    }
    // work list: chunks of at most MAX_FRAMES_PER_LAUNCH frames, each split over the pipeline lanes
    // at a multiple of 32 frames (the MFMA group size)
    const bool split = mfma && c->lanes > 1 && n >= 128;
    if (split) {
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        for (int l = 0; l < melf_ctx::NLANES; ++l) HIP_TRY(hipStreamWaitEvent(c->lane_stream[l], c->ev_fork, 0));
    }
    for (int f0 = 0; f0 < n; f0 += MAX_FRAMES_PER_LAUNCH) {
        const int mtot = n - f0 < MAX_FRAMES_PER_LAUNCH ? n - f0 : MAX_FRAMES_PER_LAUNCH;
        const int nl = split ? c->lanes : 1;
        const int per = split ? ((mtot / nl + 31) / 32) * 32 : mtot;
        for (int l = 0; l < nl; ++l) {
            const int g0 = f0 + l * per;
            const int m = l == nl - 1 ? f0 + mtot - g0 : per;
            if (m <= 0) continue;
            hipStream_t ls = split ? c->lane_stream[l] : st;
            const int bl = split ? l : c->active_lane;  // whose work buffers
            const uint8_t* base = (const uint8_t*)d_frames + (size_t)g0 * frame_stride;
            MatchSrc ms;
            ms.base = base; ms.frame_stride = frame_stride; ms.row_stride = row_stride;
            ms.x0 = x0; ms.y0 = y0; ms.rows = crows; ms.cols = ccols;
            ms.readable = (size_t)(m - 1) * frame_stride + (size_t)H * row_stride;
            int nparts = 0;
            MatchPartial* parts = nullptr;
            if (int rc = run_match(c, ms, true, m, bl, ls, nullptr, &parts, &nparts)) return rc;
            DialsSrc ds;
            ds.base = base; ds.frame_stride = frame_stride; ds.row_stride = row_stride;
            ds.x0 = x0; ds.y0 = y0; ds.crop_rows = crows; ds.crop_cols = ccols;
            ds.readable = (size_t)(m - 1) * frame_stride + (size_t)H * row_stride;
            if (c->order_valid) {
                // resident mode: prep and match above ran unordered with the caller's stream (they touch only the frames
                // and the lane's buffers); the kernel that writes the caller's records waits for everything that stream
                // held when the call was made
                if (!c->ev_call[bl]) HIP_TRY(hipEventCreateWithFlags(&c->ev_call[bl], hipEventDisableTiming));
                HIP_TRY(hipEventRecord(c->ev_call[bl], c->order_stream));
                HIP_TRY(hipStreamWaitEvent(ls, c->ev_call[bl], 0));
            }
            {
                KernelTimer t(c, MELF_K_DIALS, ls);
                launch_dials(ds, false, m, P, c->d_geom, c->d_rowmasks, parts, nparts, rw, res_dev + g0, ls, c->ws_max);
            }
            HIP_TRY(hipGetLastError());
        }
    }
    if (split) {
        for (int l = 0; l < melf_ctx::NLANES; ++l) {
            HIP_TRY(hipEventRecord(c->ev_join[l], c->lane_stream[l]));
            HIP_TRY(hipStreamWaitEvent(st, c->ev_join[l], 0));
        }
    }
    if (out_host) {
        HIP_TRY(hipMemcpyAsync(out_host, res_dev, (size_t)n * sizeof(melf_result), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return MELF_SUCCESS;
}

// A stream of batches in one call (frames resident in HBM): consecutive batches alternate between the context's
// two pipeline lanes (own work buffers and stream each), so that one batch's prep / dials kernels run in the
// tail of the other batch's match kernel, whose last waves leave three quarters of the SIMDs idle.
extern "C" int melf_process_stream_dev(melf_ctx* c, const void* d_frames, int nbatches, size_t batch_stride, int n, int H,
                                       int W, size_t frame_stride, void* d_results, size_t results_stride, void* stream_)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    if (nbatches < 0 || n < 0) return fail(MELF_ERR_INVALID, "bad batch shape");
    if (nbatches == 0 || n == 0) return MELF_SUCCESS;
    if (!d_frames || !d_results) return fail(MELF_ERR_INVALID, "NULL device pointer");
    HIP_TRY(hipSetDevice(c->device));
    if (H <= 0 || W <= 0 || frame_stride < (size_t)H * W * 3) return fail(MELF_ERR_INVALID, "bad frame shape / stride");
    hipStream_t st = (hipStream_t)stream_;  // NULL = the null (legacy default) stream, as everywhere in HIP
    if (int rc = claim_all_lanes(c, st)) return rc;
    if (nbatches == 1 || n > MAX_FRAMES_PER_LAUNCH)  // nothing to overlap / too large for one set of lane buffers
    {
        for (int b = 0; b < nbatches; ++b)
            if (int rc = process_batch_on(c, (const uint8_t*)d_frames + (size_t)b * batch_stride, n, H, W, frame_stride,
                                          (melf_result*)d_results + (size_t)b * results_stride, nullptr, st))
                return rc;
        return MELF_SUCCESS;
    }
    HIP_TRY(hipEventRecord(c->ev_fork, st));
    for (int l = 0; l < melf_ctx::NLANES; ++l) HIP_TRY(hipStreamWaitEvent(c->lane_stream[l], c->ev_fork, 0));
    const int saved = c->active_lane;
    int rc = MELF_SUCCESS;
    for (int b = 0; b < nbatches && rc == MELF_SUCCESS; ++b) {
        c->active_lane = b % melf_ctx::NLANES;
        rc = process_batch_on(c, (const uint8_t*)d_frames + (size_t)b * batch_stride, n, H, W, frame_stride,
                              (melf_result*)d_results + (size_t)b * results_stride, nullptr, c->lane_stream[c->active_lane]);
    }
    c->active_lane = saved;
    for (int l = 0; l < melf_ctx::NLANES; ++l) {
        HIP_TRY(hipEventRecord(c->ev_join[l], c->lane_stream[l]));
        HIP_TRY(hipStreamWaitEvent(st, c->ev_join[l], 0));
    }
    return rc;
}

// Frames in HOST memory (the reference's get_bgr_image + _crop_meter, meterelf/_image.py:46-55): the path only ever
// reads the meter_rect crop, so only the crop crosses PCIe -- 187 500 of a 640x480 frame's 921 600 bytes.  Chunks of
// frames are packed (on the host pool's threads) into one of two pinned staging buffers, copied by DMA on a copy
// stream and processed on the context's stream, so that packing chunk k+1, the copy of chunk k and the kernels of
// chunk k-1 overlap.  The kernels see the crops as frames of crop size with the rect at the origin.
extern "C" int melf_process_batch(melf_ctx* c, const uint8_t* frames_host, int n, int H, int W, size_t frame_stride,
                                  melf_result* out_host)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    if (n == 0) return MELF_SUCCESS;
    if (!frames_host || !out_host || n < 0 || H <= 0 || W <= 0) return fail(MELF_ERR_INVALID, "bad argument");
    if (frame_stride < (size_t)H * W * 3) return fail(MELF_ERR_INVALID, "frame_stride smaller than a frame");
    HIP_TRY(hipSetDevice(c->device));
    pool_use_device(c->device);
    const melf_params& P = c->P;
    const int x0 = P.rect_x0 < W ? P.rect_x0 : W, x1 = P.rect_x1 < W ? P.rect_x1 : W;
    const int y0 = P.rect_y0 < H ? P.rect_y0 : H, y1 = P.rect_y1 < H ? P.rect_y1 : H;
    const int crows = y1 - y0, ccols = x1 - x0;
    if (x0 < 0 || y0 < 0 || crows < P.th || ccols < P.tw)
        return fail(MELF_ERR_INVALID, "meter_rect crop is smaller than the dials template (cv2.matchTemplate would assert)");
    const size_t row_bytes = (size_t)ccols * 3;
    // crop rows at a 64-byte pitch, so that the host side can pack them with streaming (non-temporal) 16-byte stores:
    // a plain memcpy into the staging buffer reads every destination line before writing it, a third of the pack's
    // memory traffic; + 128 spare bytes per crop (the prep kernel's aligned 100-byte windows reach past the last pixel)
    const size_t pitch = (row_bytes + 63) & ~(size_t)63;
    const size_t crop_stride = (size_t)crows * pitch + 128;
    const int chunk = 128;  // frames per pipeline stage (a multiple of the 32-frame MFMA group)
    const size_t pin_need = (size_t)(n < chunk ? n : chunk) * crop_stride;
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        if (!c->ev_h2d[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_h2d[b], hipEventDisableTiming));
        if (c->pin_cap[b] < pin_need && (b == 0 || n > chunk)) {
            if (c->h_pin[b]) { HIP_TRY(hipStreamSynchronize(c->copy_stream)); HIP_TRY(hipHostFree(c->h_pin[b])); }
            c->h_pin[b] = nullptr;
            c->pin_cap[b] = 0;
            HIP_TRY(hipHostMalloc((void**)&c->h_pin[b], pin_need, hipHostMallocDefault));
            c->pin_cap[b] = pin_need;
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));  // the previous call's kernels may still read d_crops
    if (int rc = grow(&c->d_crops, &c->crops_cap, (size_t)n * crop_stride)) return rc;
    if (int rc = grow(&c->d_results, &c->results_cap, (size_t)n)) return rc;
    if (c->lanes > 1) {
        if (int rc = claim_all_lanes(c, c->stream)) return rc;
    } else if (int rc = acquire_lane(c, c->stream, &c->active_lane)) {
        return rc;
    }
    static const bool trace = diag_env("MELF_HOSTFED_TRACE") != nullptr;
    double pack_ms = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    const int rect[4] = {0, 0, ccols, crows};
    const uint8_t* frames_end = frames_host + (size_t)(n - 1) * frame_stride + (size_t)H * W * 3;
    int k = 0;
    for (int f0 = 0; f0 < n; f0 += chunk, ++k) {
        const int m = n - f0 < chunk ? n - f0 : chunk;
        const int b = k & 1;
        if (k >= 2) HIP_TRY(hipEventSynchronize(c->ev_h2d[b]));  // the copy that last read this staging buffer is done
        uint8_t* pin = c->h_pin[b];
        const auto tp0 = std::chrono::steady_clock::now();
        // work items of 32 crop rows: a chunk of 128 frames gives a 16-thread pool ~1000 items
        const int rblocks = (crows + 31) / 32;
        host_pool().run(m * rblocks, [&](int item) {
            const int i = item / rblocks, r0 = (item - i * rblocks) * 32, r1 = r0 + 32 < crows ? r0 + 32 : crows;
            const uint8_t* src = frames_host + (size_t)(f0 + i) * frame_stride + ((size_t)(y0 + r0) * W + x0) * 3;
            uint8_t* dst = pin + (size_t)i * crop_stride + (size_t)r0 * pitch;
            for (int y = r0; y < r1; ++y, src += (size_t)W * 3, dst += pitch) {
                if (src + pitch <= frames_end) {  // whole 64-byte pitch from the source row (the tail bytes are never looked at)
                    for (size_t o = 0; o < pitch; o += 16)
                        _mm_stream_si128((__m128i*)(dst + o), _mm_loadu_si128((const __m128i*)(src + o)));
                } else {
                    memcpy(dst, src, row_bytes);
                }
            }
            _mm_sfence();
        });
        pack_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count();
        uint8_t* d_chunk = c->d_crops + (size_t)f0 * crop_stride;
        HIP_TRY(hipMemcpyAsync(d_chunk, pin, (size_t)m * crop_stride, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(hipEventRecord(c->ev_h2d[b], c->copy_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_h2d[b], 0));
        if (int rc = process_batch_on(c, d_chunk, m, crows, ccols, crop_stride, c->d_results + f0, nullptr, c->stream, rect, (int)pitch))
            return rc;
    }
    HIP_TRY(hipMemcpyAsync(out_host, c->d_results, (size_t)n * sizeof(melf_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (trace)
        fprintf(stderr, "[melf host-fed] n=%d crop %dx%d: %.2f ms in all, %.2f ms of it packing (%d pool threads), %.1f MB over PCIe\n", n, ccols,
                crows, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), pack_ms,
                host_pool().size() + 1, (double)n * crop_stride / 1e6);
    return MELF_SUCCESS;
}

// ---------------------------------------------------------- stage entries ----
extern "C" int melf_bgr2hls(melf_ctx* c, const uint8_t* src_host, int rows, int cols, size_t row_stride, uint8_t* dst_host)
{
    if (!c || !src_host || !dst_host || rows <= 0 || cols <= 0 || row_stride < (size_t)cols * 3)
        return fail(MELF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t in_bytes = (size_t)rows * row_stride, out_bytes = (size_t)rows * cols * 3;
    if (int rc = grow(&c->d_stage_in, &c->stage_in_cap, in_bytes)) return rc;
    if (int rc = grow(&c->d_stage_out, &c->stage_out_cap, out_bytes)) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_stage_in, src_host, in_bytes, hipMemcpyHostToDevice, c->stream));
    {
        KernelTimer t(c, MELF_K_HLS, c->stream);
        launch_bgr2hls(c->d_stage_in, rows, cols, row_stride, c->P.hue_shift, c->d_stage_out, c->stream);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dst_host, c->d_stage_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MELF_SUCCESS;
}

extern "C" int melf_hls_inrange_close_dev(melf_ctx* c, const void* d_frames, int n, int H, int W, void* d_masks,
                                          void* stream_)
{
    if (!c || n < 0 || H <= 0 || W <= 0) return fail(MELF_ERR_INVALID, "bad argument");
    if (n == 0) return MELF_SUCCESS;
    if (!d_frames || !d_masks) return fail(MELF_ERR_INVALID, "NULL device pointer");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = ensure_fused_tables(c)) return rc;   // first use: builds them (synchronises the context's stream once)
    hipStream_t st = (hipStream_t)stream_;  // NULL = the null (legacy default) stream, as everywhere in HIP
    // One kernel per call.  Round 4 tried one call as two half-batch kernels on the context's two lane streams (what two caller
    // streams buy, inside one call): 0.1075 ms per B = 256 call against 0.0691 -- the fork / join events cost more than the
    // overlapped ramp and tail save (profiles/r04/fused_split_and_lds_stream.txt); not kept.
    for (int f0 = 0; f0 < n; f0 += MAX_FRAMES_PER_LAUNCH) {
        const int m = n - f0 < MAX_FRAMES_PER_LAUNCH ? n - f0 : MAX_FRAMES_PER_LAUNCH;
        const uint8_t* fin = (const uint8_t*)d_frames + (size_t)f0 * H * W * 3;
        uint8_t* fout = (uint8_t*)d_masks + (size_t)f0 * H * W;
        if (fused_mask_lut_ok(fin, fout, H, W) && !c->force_generic_mask) {
            TimedEvent ev;
            ev.kernel = MELF_K_FUSED_MASK;
            ev.start = ev.stop = nullptr;
            if (c->profiling == 1) {   // the dispatch's own time stamps, as for the match kernel (no event-record packets)
                HIP_TRY(hipEventCreateWithFlags(&ev.start, hipEventDisableSystemFence));
                HIP_TRY(hipEventCreateWithFlags(&ev.stop, hipEventDisableSystemFence));
                fused_mask_timing_events(ev.start, ev.stop);
            }
            launch_fused_mask_lut(fin, m, H, W, c->P.hue_shift, c->P.needle_lo, c->P.needle_hi, c->d_fused_tables,
                                  c->fused_variant, fout, st);
            if (ev.start) c->events.push_back(ev);
        } else {
            KernelTimer t(c, MELF_K_FUSED_MASK, st);
            launch_fused_mask(fin, m, H, W, c->P.hue_shift, c->P.needle_lo, c->P.needle_hi, fout, st);
        }
    }
    HIP_TRY(hipGetLastError());
    return MELF_SUCCESS;
}

#ifdef MELF_DIAG   // measurement aid of the diagnostic build (bench.py's stream_ceiling): not an entry point of the product library
extern "C" int melf_stream_probe_dev(melf_ctx* c, const void* d_in, size_t in_bytes, void* d_out, int chunks_per_block, void* stream_)
{
    if (!c || !d_in || !d_out) return fail(MELF_ERR_INVALID, "bad argument");
    if ((((size_t)d_in | (size_t)d_out) & 15) != 0) return fail(MELF_ERR_INVALID, "buffers must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = ensure_fused_tables(c)) return rc;   // the work queues live behind the tables
    TimedEvent ev;
    ev.kernel = MELF_K_STREAM_PROBE;
    ev.start = ev.stop = nullptr;
    if (c->profiling == 1) {
        HIP_TRY(hipEventCreateWithFlags(&ev.start, hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&ev.stop, hipEventDisableSystemFence));
    }
    launch_stream_probe(d_in, in_bytes, d_out, chunks_per_block, c->d_fused_tables, (hipStream_t)stream_, ev.start, ev.stop);
    if (ev.start) c->events.push_back(ev);
    HIP_TRY(hipGetLastError());
    return MELF_SUCCESS;
}
#endif

extern "C" int melf_hls_inrange_close(melf_ctx* c, const uint8_t* frames_host, int n, int H, int W, uint8_t* masks_host)
{
    if (!c || !frames_host || !masks_host || n < 0 || H <= 0 || W <= 0) return fail(MELF_ERR_INVALID, "bad argument");
    if (n == 0) return MELF_SUCCESS;
    HIP_TRY(hipSetDevice(c->device));
    const size_t in_bytes = (size_t)n * H * W * 3, out_bytes = (size_t)n * H * W;
    if (int rc = grow(&c->d_stage_in, &c->stage_in_cap, in_bytes)) return rc;
    if (int rc = grow(&c->d_stage_out, &c->stage_out_cap, out_bytes)) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_stage_in, frames_host, in_bytes, hipMemcpyHostToDevice, c->stream));
    if (int rc = melf_hls_inrange_close_dev(c, c->d_stage_in, n, H, W, c->d_stage_out, c->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(masks_host, c->d_stage_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MELF_SUCCESS;
}

extern "C" int melf_match_ccoeff(melf_ctx* c, const uint8_t* images_host, int n, int rows, int cols, float* max_val,
                                 int32_t* max_x, int32_t* max_y, float* result_map)
{
    if (!c || !images_host || n < 0) return fail(MELF_ERR_INVALID, "bad argument");
    if (n == 0) return MELF_SUCCESS;
    if (rows < c->P.th || cols < c->P.tw) return fail(MELF_ERR_INVALID, "image smaller than the template");
    if (n > MAX_FRAMES_PER_LAUNCH) return fail(MELF_ERR_INVALID, "too many images for one stage call");
    HIP_TRY(hipSetDevice(c->device));
    const int rh = rows - c->P.th + 1, rw = cols - c->P.tw + 1;
    const size_t in_bytes = (size_t)n * rows * cols;
    const size_t map_bytes = result_map ? (size_t)n * rh * rw * sizeof(float) : 0;
    if (int rc = grow(&c->d_stage_in, &c->stage_in_cap, in_bytes)) return rc;
    if (int rc = grow(&c->d_stage_out, &c->stage_out_cap, map_bytes + 16)) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_stage_in, images_host, in_bytes, hipMemcpyHostToDevice, c->stream));
    if (int rc = claim_lane(c, 0, c->stream)) return rc;
    MatchSrc ms;
    ms.base = c->d_stage_in; ms.frame_stride = (size_t)rows * cols; ms.row_stride = cols;
    ms.x0 = 0; ms.y0 = 0; ms.rows = rows; ms.cols = cols;
    ms.readable = (size_t)n * rows * cols;
    int nparts = 0;
    MatchPartial* d_parts = nullptr;
    if (int rc = run_match(c, ms, false, n, 0, c->stream, result_map ? (float*)c->d_stage_out : nullptr, &d_parts, &nparts)) return rc;
    HIP_TRY(hipGetLastError());
    std::vector<MatchPartial> parts((size_t)n * nparts);
    HIP_TRY(hipMemcpyAsync(parts.data(), d_parts, parts.size() * sizeof(MatchPartial), hipMemcpyDeviceToHost, c->stream));
    if (result_map) HIP_TRY(hipMemcpyAsync(result_map, c->d_stage_out, map_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // fold the per-tile (max, first-argmax) pairs exactly as K3's prologue does on the device
    for (int f = 0; f < n; ++f) {
        float bv = 0.f;
        int bi = INT32_MAX;
        for (int k = 0; k < nparts; ++k) {
            const MatchPartial& p = parts[(size_t)f * nparts + k];
            if (p.idx != INT32_MAX && (bi == INT32_MAX || p.val > bv || (p.val == bv && p.idx < bi))) { bv = p.val; bi = p.idx; }
        }
        if (max_val) max_val[f] = bv;
        if (max_x) max_x[f] = bi % rw;
        if (max_y) max_y[f] = bi / rw;
    }
    return MELF_SUCCESS;
}

extern "C" int melf_read_dials(melf_ctx* c, const uint8_t* dials_hls_host, int n, melf_result* out_host)
{
    if (!c || !dials_hls_host || !out_host || n < 0) return fail(MELF_ERR_INVALID, "bad argument");
    if (n == 0) return MELF_SUCCESS;
    HIP_TRY(hipSetDevice(c->device));
    const melf_params& P = c->P;
    const size_t per = (size_t)P.th * P.tw * 3;
    if (int rc = grow(&c->d_stage_in, &c->stage_in_cap, (size_t)n * per)) return rc;
    if (int rc = grow(&c->d_results, &c->results_cap, (size_t)n)) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_stage_in, dials_hls_host, (size_t)n * per, hipMemcpyHostToDevice, c->stream));
    DialsSrc ds;
    ds.base = c->d_stage_in; ds.frame_stride = per; ds.row_stride = P.tw * 3;
    ds.x0 = 0; ds.y0 = 0; ds.crop_rows = P.th; ds.crop_cols = P.tw;
    ds.readable = (size_t)n * per;
    {
        KernelTimer t(c, MELF_K_DIALS, c->stream);
        launch_dials(ds, true, n, P, c->d_geom, c->d_rowmasks, nullptr, 0, 1, c->d_results, c->stream, c->ws_max);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_host, c->d_results, (size_t)n * sizeof(melf_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MELF_SUCCESS;
}

// ------------------------------------------------------- calibration stages ----
extern "C" int melf_aligned_average(melf_ctx* c, const uint8_t* frames_host, int n, int H, int W, size_t frame_stride,
                                    const int32_t* match_x, const int32_t* match_y, int align_x, int align_y,
                                    uint8_t* out_crop_host)
{
    if (!c || !frames_host || !match_x || !match_y || !out_crop_host || n < 1 || H <= 0 || W <= 0)
        return fail(MELF_ERR_INVALID, "bad argument");
    if (frame_stride < (size_t)H * W * 3) return fail(MELF_ERR_INVALID, "frame_stride smaller than a frame");
    HIP_TRY(hipSetDevice(c->device));
    const melf_params& P = c->P;
    const int x0 = P.rect_x0 < W ? P.rect_x0 : W, x1 = P.rect_x1 < W ? P.rect_x1 : W;
    const int y0 = P.rect_y0 < H ? P.rect_y0 : H, y1 = P.rect_y1 < H ? P.rect_y1 : H;
    const int rows = y1 - y0, cols = x1 - x0;
    if (x0 < 0 || y0 < 0 || rows < 1 || cols < 1) return fail(MELF_ERR_INVALID, "empty meter_rect crop");
    const size_t in_bytes = (size_t)n * frame_stride, out_bytes = (size_t)rows * cols * 3;
    if (int rc = grow(&c->d_stage_in, &c->stage_in_cap, in_bytes)) return rc;
    if (int rc = grow(&c->d_stage_out, &c->stage_out_cap, out_bytes + (size_t)n * 8 + 64)) return rc;
    int32_t* d_mx = (int32_t*)(c->d_stage_out + ((out_bytes + 15) & ~(size_t)15));
    int32_t* d_my = d_mx + n;
    HIP_TRY(hipMemcpyAsync(c->d_stage_in, frames_host, in_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_mx, match_x, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_my, match_y, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    launch_aligned_average(c->d_stage_in, n, frame_stride, W * 3, x0, y0, rows, cols, d_mx, d_my, align_x, align_y,
                           c->d_stage_out, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_crop_host, c->d_stage_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MELF_SUCCESS;
}

extern "C" int melf_inrange(melf_ctx* c, const uint8_t* img_host, int rows, int cols, const int32_t lo[3],
                            const int32_t hi[3], uint8_t* mask_host)
{
    if (!c || !img_host || !lo || !hi || !mask_host || rows < 1 || cols < 1) return fail(MELF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t npx = (size_t)rows * cols;
    if (int rc = grow(&c->d_stage_in, &c->stage_in_cap, npx * 3)) return rc;
    if (int rc = grow(&c->d_stage_out, &c->stage_out_cap, npx)) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_stage_in, img_host, npx * 3, hipMemcpyHostToDevice, c->stream));
    launch_inrange3(c->d_stage_in, (int)npx, lo, hi, c->d_stage_out, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mask_host, c->d_stage_out, npx, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MELF_SUCCESS;
}

// ------------------------------------------------------------ JPEG decode ----
extern "C" int melf_jpeg_probe(const uint8_t* data, size_t size, int32_t* H, int32_t* W, int32_t* supported)
{
    if (!data || !H || !W || !supported) return fail(MELF_ERR_INVALID, "bad argument");
    int h = 0, w = 0, ok = 0;
    std::string why;
    jpeg_probe(data, size, &h, &w, &ok, &why);
    *H = h; *W = w; *supported = ok;
    g_err = why;
    return MELF_SUCCESS;
}

extern "C" int melf_jpeg_probe_batch(const uint8_t* const* data, const size_t* sizes, int n, int32_t* H, int32_t* W,
                                     int32_t* supported)
{
    if (n < 0 || (n > 0 && (!data || !sizes || !H || !W || !supported))) return fail(MELF_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) {
        int h = 0, w = 0, ok = 0;
        if (data[i] && sizes[i]) jpeg_probe(data[i], sizes[i], &h, &w, &ok, nullptr);
        H[i] = h; W[i] = w; supported[i] = ok;
    }
    return MELF_SUCCESS;
}

namespace {
struct JpegTimers {
    melf_ctx* c;
    hipStream_t s;
    TimedEvent ev[3];
    bool on[3];
};
void jpeg_timer_hook(void* arg, int k, int stop)
{
    JpegTimers* t = (JpegTimers*)arg;
    if (t->c->profiling != 1) return;
    if (!stop) {
        t->ev[k].kernel = MELF_K_JPEG_HUFF + k;
        t->on[k] = hipEventCreate(&t->ev[k].start) == hipSuccess && hipEventCreate(&t->ev[k].stop) == hipSuccess;
        if (t->on[k]) hipEventRecord(t->ev[k].start, t->s);
    } else if (t->on[k]) {
        hipEventRecord(t->ev[k].stop, t->s);
        t->c->events.push_back(t->ev[k]);
    }
}

// decode into device memory `d_frames` (n*H*W*3 bytes) on the context's stream; fills status (host)
int jpeg_decode_to_device(melf_ctx* c, const uint8_t* const* data, const size_t* sizes, int n, int H, int W,
                          uint8_t* d_frames, int32_t* status, const int* rect)
{
    HIP_TRY(hipStreamSynchronize(c->stream));  // the pinned stage buffer of the previous batch is free again
    static const bool trace = diag_env("MELF_JPEG_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int32_t> hstat(n);
    std::string err;
    if (int rc = jpeg_prepare_batch(&c->jpeg, data, sizes, n, H, W, hstat.data(), &err)) return fail(rc, err);
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<int32_t> dstat(n);
    JpegTimers t{c, c->stream, {}, {false, false, false}};
    if (int rc = jpeg_launch_batch(c->jpeg, n, H, W, d_frames, dstat.data(), c->stream, &err, jpeg_timer_hook, &t, rect)) return fail(rc, err);
    if (trace) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[melf jpeg] n=%d host prepare %.2f ms, H2D + kernels + status %.2f ms\n", n,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    for (int i = 0; i < n; ++i) status[i] = hstat[i] ? hstat[i] : (dstat[i] ? MELF_JPEG_CORRUPT : MELF_JPEG_OK);
    return MELF_SUCCESS;
}
}  // namespace

extern "C" int melf_jpeg_decode_batch(melf_ctx* c, const uint8_t* const* data, const size_t* sizes, int n, int H, int W,
                                      void* out, int out_on_device, int32_t* status)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    if (n == 0) return MELF_SUCCESS;
    if (!data || !sizes || !out || !status || n < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || n > 32768)
        return fail(MELF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    pool_use_device(c->device);
    const size_t bytes = (size_t)n * H * W * 3;
    uint8_t* d = (uint8_t*)out;
    if (!out_on_device) {
        if (int rc = grow(&c->d_stage_in, &c->stage_in_cap, bytes)) return rc;
        d = c->d_stage_in;
    }
    if (int rc = jpeg_decode_to_device(c, data, sizes, n, H, W, d, status, nullptr)) return rc;
    if (!out_on_device) {
        HIP_TRY(hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return MELF_SUCCESS;
}

// melf_jpeg_process_files_begin: the call slot of the thread's call, and what to do once everything of the call is
// ENQUEUED (the next call in flight may then start its own preparation and enqueue behind it).  NULL: a plain call.
// tl_jpeg_files_thread: the call runs on the thread of a melf_jpeg_process_files_begin call, i.e. the PREVIOUS such call of
// the context may still have kernels running and its host thread may still be in its tail (waiting for its "call done"
// event, reading its slot's pinned status, filling the caller's records).  Such a call therefore uses the call slot of its
// own ticket -- for EVERY frame-size group of its file list, not only for the one that carries the hand-over hook (round 3
// took slot 0 for the other groups and could overwrite the previous call's status buffer under its reader) -- and never
// assumes the GPU is its own.
// INVARIANT: once a call has run (*tl_jpeg_enqueued)() it touches nothing of the context but its own call slot
// (d_jframes / d_jresults / h_jstatus / ev_jcall [slot]); everything else belongs to the next call from that moment on.
static size_t jrecs_offset(size_t n) { return (n * sizeof(int32_t) + 127) / 64 * 64; }   // records behind n status words
static double trace_clock_ms(std::chrono::steady_clock::time_point t)
{
    static const auto epoch = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(t - epoch).count();
}
static thread_local int tl_jpeg_slot = 0;
static thread_local bool tl_jpeg_files_thread = false;
static thread_local std::function<void()>* tl_jpeg_enqueued = nullptr;

// Decode of n files into d_frames in chunks, pipelined: while chunk k's kernels run (on one of two decode streams),
// the host parses chunk k + 1, builds its Huffman data, copies its scan bytes (as they are: the GPU cleans them) into the other workspace's pinned stage
// buffer, and the copy stream uploads it.  Serial, a 1024-file call spent more than half its time in host preparation,
// the upload and the status read-back with the GPU idle.  The Huffman kernel is bound by its critical path (every
// workgroup walks the same ~12 synchronisation rounds whatever the batch size: 0.8 ms for 256 files, 1.07 ms for 1024),
// so consecutive chunks go round a ring of NJ workspaces and streams and their kernels overlap on the GPU.  Leaves the per-file
// status in h_status (pinned) once the context's stream has passed the point this function leaves it at.
// overlapped: another call's kernels may still be running (melf_jpeg_process_files_begin, two calls in flight): nothing
// here waits for the context's stream; what protects a ring slot is its own pair of events, across calls as within one.
static int process_batch_on(melf_ctx* c, const void* d_frames, int n, int H, int W, size_t frame_stride,
                            void* d_results, melf_result* out_host, hipStream_t st, const int* rect, int row_stride);
// What a caller inside the library may already have of the files it hands to the decode path (the file-name entry points
// do): the parsed headers + Huffman decode data (of file index[k] of that parse for the call's file k), and the pinned buffer
// the files' bytes lie in.
struct JpegSource {
    const JpegParsed* parsed;
    const int* index;
    const uint8_t* pin_base;
    size_t pin_len;
};
// read_chunks: the reading path runs per chunk, right behind the chunk's decode on the chunk's stream (records into
// c->d_results), instead of once over all frames afterwards -- its kernels then run beside the later chunks' Huffman
// kernels, which leave most of the chip's throughput unused.
// first_len > 0: the first `first_len` files form a chunk of their own (melf_jpeg_process_batch puts the files with very
// few bits per block there: their Huffman kernel runs several times as long as a normal chunk's and, started first,
// does so beside the other chunks' preparation and kernels instead of at the end of one of them).
static int jpeg_decode_pipelined(melf_ctx* c, const uint8_t* const* data, const size_t* sizes, int n, int H, int W, const int* rect,
                                 std::vector<int32_t>& hstat, bool read_chunks, int first_len, uint8_t* d_frames, melf_result* d_results,
                                 int32_t* h_status, bool overlapped, const JpegSource* src)
{
    // 256 files per chunk.  A call's critical path is: first chunk parsed and prepared -> ALL uploads back to back (36 MB at
    // ~42 GB/s: 0.85 ms per 1024 fixture files, the longest item) -> the LAST chunk's Huffman kernel (0.4-0.5 ms whatever the
    // chunk's size: its rounds are a latency chain) -> IDCT / colour of that chunk -> the reading path.  Smaller chunks start
    // the uploads earlier and leave less behind the last one; below 256 the per-chunk launches cost more than that gains
    // (round 4, tools/jpeg_call_rate.py, medians of 96 interleaved calls: 512 / 256 / 128+4x256.. = 2.37 / 2.30 / 2.36 ms on
    // sample-images1, 1.85 / 1.83 / 1.92 on sample-images2; round 3 with every header parsed up front: 2.43 / 1.93).
    // MELF_JPEG_CHUNK: one size, or a comma-separated plan "a,b,c" (the last entry repeats)
    std::vector<int> plan;
    if (const char* e = getenv("MELF_JPEG_CHUNK")) {
        for (const char* q = e; *q;) {
            plan.push_back(std::max(32, atoi(q)));
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
    }
    if (plan.empty()) plan.push_back(256);
    constexpr int NJ = melf_ctx::NJ;
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < NJ; ++b) {
        if (b && !c->jpeg_stream[b]) HIP_TRY(hipStreamCreateWithFlags(&c->jpeg_stream[b], hipStreamNonBlocking));
        if (!c->ev_jup[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_jup[b], hipEventDisableTiming));
        if (!c->ev_jdec[b]) HIP_TRY(hipEventCreateWithFlags(&c->ev_jdec[b], hipEventDisableTiming));
    }
    hstat.assign(n, 0);
    hipStream_t dstream[NJ];
    JpegWorkspace** ws[NJ];
    for (int b = 0; b < NJ; ++b) { dstream[b] = b ? c->jpeg_stream[b] : c->stream; ws[b] = b ? &c->jpeg_ring[b] : &c->jpeg; }
    if (!overlapped) {
        // the other streams start behind whatever the context's stream holds (a caller's earlier work on it)
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        for (int b = 1; b < NJ; ++b) HIP_TRY(hipStreamWaitEvent(c->jpeg_stream[b], c->ev_fork, 0));
        HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_fork, 0));
    }
    std::string err;
    const uint64_t seq0 = c->jpeg_chunk_seq;
    int k = 0;
    // the ring position moves on past every chunk this call touched, also when it fails half way through one
    struct SeqGuard { melf_ctx* c; uint64_t seq0; const int* k; bool done; ~SeqGuard() { c->jpeg_chunk_seq = seq0 + (uint64_t)*k + (done ? 0 : 1); } } seq_guard{c, seq0, &k, false};
    // every file's headers in one parallel pass; the chunks then only build tables and clean scans
    // (MELF_JPEG_PARSE=all, the round-3 arrangement; by default each chunk parses its own files, so that only the first
    // chunk's headers are parsed before the first upload can start: 0.24 ms of a 1024-file call's head otherwise)
    const char* pmode = diag_env("MELF_JPEG_PARSE");
    const bool parse_all = pmode && !strcmp(pmode, "all");
    struct ParsedGuard { JpegParsed* p; ~ParsedGuard() { if (p) jpeg_parsed_free(p); } } parsed{
        parse_all && !src ? jpeg_parse_files(data, sizes, n, H, W, hstat.data()) : nullptr};
    // (a caller that brings the headers has checked them: every file is one for the GPU, of this frame size: hstat stays 0)
    for (int f0 = 0, m = 0, planned = 0; f0 < n; f0 += m, ++k) {
        m = (k == 0 && first_len > 0) ? first_len : plan[std::min(planned++, (int)plan.size() - 1)];
        if (m > n - f0) m = n - f0;
        const uint64_t seq = seq0 + (uint64_t)k;
        const int b = (int)(seq % NJ);
        static const bool trace = diag_env("MELF_JPEG_TRACE") != nullptr;
        double tt[5] = {};
        if (trace) tt[0] = trace_clock_ms(std::chrono::steady_clock::now());
        // the upload that last read this workspace's pinned stage buffer must be done before the host refills it
        if (seq >= (uint64_t)NJ) HIP_TRY(hipEventSynchronize(c->ev_jup[b]));
        if (trace) tt[1] = trace_clock_ms(std::chrono::steady_clock::now());
        if (int rc = src ? jpeg_prepare_batch(ws[b], data + f0, sizes + f0, m, H, W, hstat.data() + f0, &err, src->parsed, f0, src->index,
                                              src->pin_base, src->pin_len)
                         : jpeg_prepare_batch(ws[b], data + f0, sizes + f0, m, H, W, hstat.data() + f0, &err, parsed.p, f0))
            return fail(rc, err);
        if (trace) tt[2] = trace_clock_ms(std::chrono::steady_clock::now());
        // ... and the kernels that last read its device buffers before the upload overwrites them
        if (seq >= (uint64_t)NJ) HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_jdec[b], 0));
        if (int rc = jpeg_upload_batch(*ws[b], m, c->copy_stream, &err)) return fail(rc, err);
        double t_up = 0;
        if (trace) t_up = trace_clock_ms(std::chrono::steady_clock::now());
        HIP_TRY(hipEventRecord(c->ev_jup[b], c->copy_stream));
        HIP_TRY(hipStreamWaitEvent(dstream[b], c->ev_jup[b], 0));
        JpegTimers t{c, dstream[b], {}, {false, false, false}};
        if (int rc = jpeg_decode_batch_kernels(*ws[b], m, H, W, d_frames + (size_t)f0 * H * W * 3, dstream[b], &err, jpeg_timer_hook, &t, rect))
            return fail(rc, err);
        HIP_TRY(hipMemcpyAsync(h_status + f0, jpeg_device_status(*ws[b]), (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, dstream[b]));
        if (trace) tt[3] = trace_clock_ms(std::chrono::steady_clock::now());
        if (read_chunks) {
            c->active_lane = k % melf_ctx::NLANES;
            if (int rc = claim_lane(c, c->active_lane, dstream[b])) return rc;
            if (int rc = process_batch_on(c, d_frames + (size_t)f0 * H * W * 3, m, H, W, (size_t)H * W * 3, d_results + f0, nullptr,
                                          dstream[b], nullptr, 0))
                return rc;
        }
        HIP_TRY(hipEventRecord(c->ev_jdec[b], dstream[b]));
        if (trace) {
            tt[4] = trace_clock_ms(std::chrono::steady_clock::now());
            fprintf(stderr, "[melf jpeg]   chunk %d (%d files, ring slot %d): at %.2f, slot free %.2f, prepared %.2f, uploads enqueued %.2f, decode enqueued %.2f, all enqueued %.2f\n",
                    k, m, b, tt[0], tt[1], tt[2], t_up, tt[3], tt[4]);
        }
    }
    // the context's stream continues when this call's decode streams are done (a slot's latest record covers its earlier ones)
    for (int q = k > NJ ? k - NJ : 0; q < k; ++q) {
        const int b = (int)((seq0 + (uint64_t)q) % NJ);
        if (b) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_jdec[b], 0));
    }
    seq_guard.done = true;
    return MELF_SUCCESS;
}

static int jpeg_process_batch_from(melf_ctx* c, const uint8_t* const* data, const size_t* sizes, int n, int H, int W,
                                   melf_result* out_host, int32_t* status, const JpegSource* src);
extern "C" int melf_jpeg_process_batch(melf_ctx* c, const uint8_t* const* data, const size_t* sizes, int n, int H, int W,
                                       melf_result* out_host, int32_t* status)
{
    return jpeg_process_batch_from(c, data, sizes, n, H, W, out_host, status, nullptr);
}
static int jpeg_process_batch_from(melf_ctx* c, const uint8_t* const* data, const size_t* sizes, int n, int H, int W,
                                   melf_result* out_host, int32_t* status, const JpegSource* src)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    if (n == 0) return MELF_SUCCESS;
    if (!data || !sizes || !out_host || !status || n < 0 || H <= 0 || W <= 0 || H > 65535 || W > 65535 || n > 32768)
        return fail(MELF_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    pool_use_device(c->device);
    const size_t bytes = (size_t)n * H * W * 3;
    // overlapped: called on the thread of a melf_jpeg_process_files_begin call while the previous such call may still have
    // kernels running (its buffers are another call slot's; the ring slots are guarded by their events)
    const bool overlapped = tl_jpeg_files_thread;
    const int cs = overlapped ? tl_jpeg_slot % melf_ctx::NJC : 0;
    // the reading path only looks at the meter_rect crop: IDCT and colour conversion are limited to it
    const int rect[4] = {c->P.rect_x0, c->P.rect_y0, c->P.rect_x1, c->P.rect_y1};
    const bool serial = diag_env("MELF_JPEG_SERIAL") != nullptr;   // A/B and tests: the one-piece path
    if (!overlapped) HIP_TRY(hipStreamSynchronize(c->stream));   // a caller's earlier work on the context's stream
    if (serial || n <= 64) {
        if (overlapped) HIP_TRY(hipDeviceSynchronize());   // the one-piece path shares its buffers with every call: alone on the GPU
        if (int rc = grow(&c->d_stage_in, &c->stage_in_cap, bytes)) return rc;
        if (int rc = jpeg_decode_to_device(c, data, sizes, n, H, W, c->d_stage_in, status, rect)) return rc;
        return melf_process_batch_dev(c, c->d_stage_in, n, H, W, (size_t)H * W * 3, nullptr, out_host, c->stream);
    }
    static const bool trace = diag_env("MELF_JPEG_TRACE") != nullptr;
    const auto tc0 = std::chrono::steady_clock::now();
    // Files with very few bits per block (a dark, flat frame: "DC difference 0, end of block" and nothing else) keep the
    // Huffman kernel's wrongly-phased decoders in step with their garbage, so the truth crawls one segment per round and
    // the image's workgroup runs ~3x as long as a normal one -- and with it the whole chunk's kernel.  They go FIRST, in a
    // chunk of their own: its kernel then runs beside the other chunks' preparation and kernels.  Scheduling only: the
    // records come back in the caller's order.
    std::vector<int> perm;
    std::vector<const uint8_t*> pdata;
    std::vector<size_t> psizes;
    std::vector<int> pindex;
    JpegSource psrc;
    int nsparse = 0;
    if (!diag_env("MELF_JPEG_NO_REORDER")) {
        const double blocks = ((H + 7) / 8) * (double)((W + 7) / 8) * 1.5;
        for (int i = 0; i < n; ++i) nsparse += (double)sizes[i] * 8.0 < 12.0 * blocks ? 1 : 0;
        if (nsparse > 0 && nsparse < n) {
            perm.resize(n); pdata.resize(n); psizes.resize(n);
            int a = 0, b = nsparse;
            for (int i = 0; i < n; ++i) perm[(double)sizes[i] * 8.0 < 12.0 * blocks ? a++ : b++] = i;
            for (int i = 0; i < n; ++i) { pdata[i] = data[perm[i]]; psizes[i] = sizes[perm[i]]; }
            data = pdata.data();
            sizes = psizes.data();
            if (src) {   // the caller's headers follow their files
                pindex.resize(n);
                for (int i = 0; i < n; ++i) pindex[i] = src->index ? src->index[perm[i]] : perm[i];
                psrc = {src->parsed, pindex.data(), src->pin_base, src->pin_len};
                src = &psrc;
            }
        } else {
            nsparse = 0;
        }
    }
    std::vector<int32_t> hstat;
    // the reading path: ONE pass over all n frames behind the last chunk (the tuned match kernel in its full-batch layout);
    // MELF_JPEG_READ=chunk runs it per chunk on the chunk's stream instead (3 % faster on sample-images1, 1 % slower on
    // sample-images2, equal with three calls in flight: not the default, the full-batch layout is what the tests assert)
    const char* rmode = diag_env("MELF_JPEG_READ");
    const bool read_chunks = rmode && !strcmp(rmode, "chunk");
    if (int rc = grow(&c->d_jframes[cs], &c->jframes_cap[cs], bytes)) return rc;
    if (int rc = grow(&c->d_jresults[cs], &c->jresults_cap[cs], (size_t)n)) return rc;
    if (c->jstatus_cap[cs] < (size_t)n) {
        if (c->h_jstatus[cs]) HIP_TRY(hipHostFree(c->h_jstatus[cs]));
        c->h_jstatus[cs] = nullptr; c->jstatus_cap[cs] = 0;
        // status words, then (64-byte aligned) the call's records: hipMemcpyAsync into the CALLER's pageable array would
        // not return before the copy has run, i.e. before all of the call's kernels have -- the hand-over below would come
        // at the end of the call instead of at "enqueued" (rounds 3-4 measured two calls in flight no faster than one)
        HIP_TRY(hipHostMalloc((void**)&c->h_jstatus[cs], jrecs_offset((size_t)n) + (size_t)n * sizeof(melf_result), hipHostMallocDefault));
        c->jstatus_cap[cs] = (size_t)n;
    }
    // blocking sync: the thread that waits for a call's kernels sleeps instead of spinning (several calls wait at any time --
    // three per context, times the contexts of a process -- and the cores are needed by the I/O pool's readers)
    static const bool spin = diag_env("MELF_JPEG_SPIN_WAIT") != nullptr;   // A/B
    if (!c->ev_jcall[cs]) HIP_TRY(hipEventCreateWithFlags(&c->ev_jcall[cs], hipEventDisableTiming | (spin ? 0 : hipEventBlockingSync)));
    int rc = jpeg_decode_pipelined(c, data, sizes, n, H, W, rect, hstat, read_chunks, nsparse, c->d_jframes[cs], c->d_jresults[cs], c->h_jstatus[cs],
                                   overlapped, src);
    if (rc == MELF_SUCCESS && !read_chunks) {
        if (c->lanes > 1) rc = claim_all_lanes(c, c->stream);
        else rc = acquire_lane(c, c->stream, &c->active_lane);
        if (rc == MELF_SUCCESS)
            rc = process_batch_on(c, c->d_jframes[cs], n, H, W, (size_t)H * W * 3, c->d_jresults[cs], nullptr, c->stream, nullptr, 0);
    }
    if (rc != MELF_SUCCESS) {
        (void)hipDeviceSynchronize();   // nothing of the aborted pipeline may outlive the call
        return rc;
    }
    // records and the decode status back, one synchronisation
    const auto tc1 = std::chrono::steady_clock::now();
    melf_result* const recs = (melf_result*)((uint8_t*)c->h_jstatus[cs] + jrecs_offset(c->jstatus_cap[cs]));   // pinned
    HIP_TRY(hipMemcpyAsync(recs, c->d_jresults[cs], (size_t)n * sizeof(melf_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_jcall[cs], c->stream));
    // everything of this call is enqueued: from here on it touches only its own slot's buffers, and the next call in flight
    // may prepare and enqueue behind it
    if (tl_jpeg_enqueued) (*tl_jpeg_enqueued)();
    HIP_TRY(hipEventSynchronize(c->ev_jcall[cs]));
    if (trace) {
        const auto tc2 = std::chrono::steady_clock::now();  // absolute times (process clock) show how consecutive calls overlap
        fprintf(stderr, "[melf jpeg] pipelined call n=%d (%d sparse files first): %.2f ms enqueueing (host prepare + launches), %.2f ms waiting for the GPU"
                        " [start %.2f, enqueued %.2f, done %.2f]\n", n,
                nsparse, std::chrono::duration<double, std::milli>(tc1 - tc0).count(), std::chrono::duration<double, std::milli>(tc2 - tc1).count(),
                trace_clock_ms(tc0), trace_clock_ms(tc1), trace_clock_ms(tc2));
    }
    for (int i = 0; i < n; ++i) {
        const int o = perm.empty() ? i : perm[i];
        status[o] = hstat[i] ? hstat[i] : (c->h_jstatus[cs][i] ? MELF_JPEG_CORRUPT : MELF_JPEG_OK);
        out_host[o] = recs[i];
    }
    return MELF_SUCCESS;
}

// get_meter_values' inner loop for file names (meterelf/_api.py:22-33): the files are read here, on threads,
// so that a scripting host pays one call per chunk instead of an open/read per file.
struct FilesRead {  // what the read stage hands to the decode stage
    std::vector<int> hs, ws, oks;
    std::vector<const uint8_t*> where;  // file i's bytes (len[i] of them, 64-byte aligned, 64 spare bytes behind), NULL: not read
    std::vector<size_t> off;     // ... = base + off[i] for a file in the arena
    std::vector<size_t> len;
    uint8_t* base = nullptr;     // the context's pinned arena of this call's slot
    size_t cap = 0;
    std::vector<uint8_t> spill;  // files the arena had no room for
    JpegParsed* parsed = nullptr;  // every file's header and Huffman decode data, made by the thread that read the file
                                   // (the slot's object: 3 MB per 1024 files, allocated -- and its pages faulted in -- once)
    FilesRead() = default;
    FilesRead(const FilesRead&) = delete;
    FilesRead& operator=(const FilesRead&) = delete;
};

// Stage 1: the files' bytes into arena `slot` of the context, header check.  Touches nothing else of the context.
static int jpeg_files_read(melf_ctx* c, int slot, const char* const* paths, int n, int32_t* H_used, int32_t* W_used,
                           melf_result* out_host, int32_t* status, FilesRead& R)
{
    if (n == 0) return MELF_SUCCESS;
    if (!paths || !out_host || !status || !H_used || !W_used || n < 0 || n > 32768) return fail(MELF_ERR_INVALID, "bad argument");
    pool_use_device(c->device);
    // ONE pass on the I/O pool (round 4; two before: a stat pass for the sizes, then the reads at known offsets): open, fstat,
    // claim the file's place in the slot's pinned arena from a bump counter, read, close, parse.  The arena is grow-only (no
    // per-file allocation, no zero fill, no fresh pages to fault in or to pin after the first calls); a file that no longer
    // fits -- the first call of a context, or a list of larger files than any before -- is read again behind the pass into
    // pageable memory (the decode stage then copies it like a caller's buffer), and the arena is replaced by a larger one
    // before the slot's NEXT call reads into it.  Files sit in the arena in the order the threads claimed their places; the
    // decode stage forms its chunks in arena order, so that a chunk's upload is still one contiguous span.
    static const bool trace = diag_env("MELF_JPEG_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int>&hs = R.hs, &ws = R.ws, &oks = R.oks;
    std::vector<size_t>&off = R.off, &len = R.len;
    try {
        hs.assign(n, 0); ws.assign(n, 0); oks.assign(n, 0);
        off.assign(n, 0);
        len.assign(n, 0);
        R.where.assign(n, nullptr);
        if (!c->files_parsed[slot]) c->files_parsed[slot] = jpeg_parsed_new(n, true);
        else jpeg_parsed_resize(c->files_parsed[slot], n);
        R.parsed = c->files_parsed[slot];
    } catch (const std::exception&) {
        return fail(MELF_ERR_INVALID, "out of host memory");
    }
    // a camera frame is tens of KiB; a "JPEG" of more than 64 MiB is not one of ours (and n of them would not fit)
    const off_t max_file = (off_t)64 << 20;
    {
        // pinned: the uploads start from this buffer.  Nothing of the GPU reads the old one any more: the slot's previous
        // call has been collected (its _end) before this one could be begun.
        // what the slot's last call needed per file (48 KiB before there was one) x this call's files, + 25 %
        const size_t per_file = c->file_arena_per_file[slot] ? c->file_arena_per_file[slot] : (size_t)48 << 10;
        // (grown with room to spare: lists of slightly different sizes must not replace the arena call after call)
        const size_t need = std::min<size_t>((size_t)n * (per_file + per_file / 8) + (1u << 20), (size_t)2 << 30);
        size_t want = need > c->file_arena_cap[slot] ? need + need / 2 : 0;
        // ... and given back (advisor, round 4: the arenas were grow-only, up to 3 GiB of pinned memory per slot after one list of
        // large files): a slot whose arena is more than four times what its calls need, sixteen calls in a row, gets a fitting one
        if (!want && c->file_arena_cap[slot] > ((size_t)64 << 20) && c->file_arena_cap[slot] > 4 * need) {
            if (++c->file_arena_oversized[slot] >= 16) want = need + need / 2;
        } else {
            c->file_arena_oversized[slot] = 0;
        }
        if (want && want != c->file_arena_cap[slot]) {
            HIP_TRY(hipSetDevice(c->device));
            if (c->file_arena[slot]) HIP_TRY(hipHostFree(c->file_arena[slot]));
            c->file_arena[slot] = nullptr;
            c->file_arena_cap[slot] = 0;
            HIP_TRY(hipHostMalloc((void**)&c->file_arena[slot], want, hipHostMallocDefault));
            c->file_arena_cap[slot] = want;
        }
    }
    uint8_t* const base = R.base = c->file_arena[slot];
    const size_t cap = R.cap = c->file_arena_cap[slot];
    const auto t_arena = std::chrono::steady_clock::now();
    std::atomic<size_t> bump{0};
    std::atomic<int> spilled{0};
    auto read_all = [](int fd, uint8_t* dst, size_t sz) {
        size_t got = 0;
        while (got < sz) {
            const ssize_t r = read(fd, dst + got, sz - got);
            if (r > 0) got += (size_t)r;
            else if (r < 0 && errno == EINTR) continue;
            else break;
        }
        return got;
    };
    io_pool().run(n, [&](int i) {
        status[i] = MELF_JPEG_UNREADABLE;
        const int fd = paths[i] ? open(paths[i], O_RDONLY | O_CLOEXEC) : -1;
        if (fd < 0) return;
        struct stat sb;
        if (fstat(fd, &sb) != 0) { close(fd); return; }
        if (!S_ISREG(sb.st_mode) || sb.st_size <= 0 || sb.st_size > max_file) {
            if (S_ISREG(sb.st_mode) && sb.st_size > max_file) status[i] = MELF_JPEG_UNSUPPORTED;
            close(fd);
            return;
        }
        const size_t sz = len[i] = (size_t)sb.st_size;
        // 64-byte aligned, 64 spare bytes behind each file (the GPU's scan cleaner reads a few bytes past a file's end)
        const size_t space = (sz + 64 + 63) / 64 * 64;
        const size_t o = bump.fetch_add(space, std::memory_order_relaxed);
        if (o + space + 64 > cap) {   // no room (any more): second pass below
            close(fd);
            spilled.fetch_add(1, std::memory_order_relaxed);
            status[i] = MELF_JPEG_OK;
            return;
        }
        off[i] = o;
        const size_t got = read_all(fd, base + o, sz);
        close(fd);
        if (got != sz) return;   // the file shrank meanwhile (one that grew: the decoder sees the first part and reports a corrupt stream)
        status[i] = MELF_JPEG_OK;
        R.where[i] = base + o;
        // header and Huffman decode data right here, while the file's first lines are in this core's cache: the decode stage
        // then has nothing to compute per file
        jpeg_parse_one(R.parsed, i, base + o, sz, &hs[i], &ws[i], &oks[i]);
    });
    const auto t_read = std::chrono::steady_clock::now();
    c->file_arena_per_file[slot] = std::max<size_t>(bump.load() / (size_t)n, 4096);
    if (spilled.load() > 0) {
        std::vector<size_t> so((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) so[(size_t)i + 1] = so[i] + ((status[i] == MELF_JPEG_OK && !R.where[i]) ? (len[i] + 64 + 63) / 64 * 64 : 0);
        try {
            R.spill.resize(so[n] + 128);
        } catch (const std::exception&) {
            return fail(MELF_ERR_INVALID, "out of host memory");
        }
        io_pool().run(n, [&](int i) {
            if (status[i] != MELF_JPEG_OK || R.where[i]) return;
            status[i] = MELF_JPEG_UNREADABLE;
            const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
            if (fd < 0) return;
            uint8_t* dst = R.spill.data() + so[i];
            const size_t got = read_all(fd, dst, len[i]);
            close(fd);
            if (got != len[i]) return;
            status[i] = MELF_JPEG_OK;
            R.where[i] = dst;
            jpeg_parse_one(R.parsed, i, dst, len[i], &hs[i], &ws[i], &oks[i]);
        });
    }
    if (trace) {
        const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        fprintf(stderr, "[melf jpeg] n=%d files read (%.1f MB): arena %.2f ms, open + read + parse %.2f ms, %d file(s) read again behind a full arena %.2f ms\n",
                n, bump.load() / 1e6, ms(t0, t_arena), ms(t_arena, t_read), spilled.load(), ms(t_read, std::chrono::steady_clock::now()));
    }
    return MELF_SUCCESS;
}

// Stage 2: decode + reading path of the files stage 1 accepted (the context's GPU state: one call at a time).  Files of
// several frame sizes in one list (a camera that was turned at some point) are processed size by size, in the order in
// which the sizes first appear; H_used / W_used report the first one.
static int jpeg_files_decode(melf_ctx* c, int n, int32_t* H_used, int32_t* W_used, melf_result* out_host, int32_t* status, const FilesRead& R,
                             std::function<void()>* enqueued = nullptr, int call_slot = 0, bool files_thread = false)
{
    if (n == 0) return MELF_SUCCESS;
    *H_used = 0; *W_used = 0;
    std::vector<char> todo(n, 0);
    int left = 0;
    for (int i = 0; i < n; ++i) {
        if (status[i] != MELF_JPEG_OK) continue;
        if (!R.where[i]) { status[i] = MELF_JPEG_UNREADABLE; continue; }
        if (!R.oks[i]) { status[i] = R.hs[i] > 0 ? MELF_JPEG_UNSUPPORTED : MELF_JPEG_CORRUPT; continue; }
        todo[i] = 1;
        ++left;
    }
    std::vector<const uint8_t*> ptr;
    std::vector<size_t> len;
    std::vector<int> where;
    std::vector<melf_result> res;
    std::vector<int32_t> st;
    while (left > 0) {
        int H = 0, W = 0;
        ptr.clear(); len.clear(); where.clear();
        for (int i = 0; i < n; ++i) {
            if (!todo[i]) continue;
            if (!H) { H = R.hs[i]; W = R.ws[i]; }
            if (R.hs[i] != H || R.ws[i] != W) continue;
            where.push_back(i);
            todo[i] = 0;
        }
        // in the order the files lie in memory: a chunk of the pipelined decode is then one contiguous span of the pinned arena
        // (the upload starts there); which files share a chunk is scheduling only, the records go back by `where`
        std::sort(where.begin(), where.end(), [&](int a, int b) { return (uintptr_t)R.where[a] < (uintptr_t)R.where[b]; });
        for (int i : where) { ptr.push_back(R.where[i]); len.push_back(R.len[i]); }
        if (!*H_used) { *H_used = H; *W_used = W; }
        const int m = (int)ptr.size();
        left -= m;
        res.resize(m);
        st.resize(m);
        // the LAST group's call hands the context on as soon as it has enqueued everything (melf_jpeg_process_batch)
        tl_jpeg_enqueued = left == 0 ? enqueued : nullptr;
        tl_jpeg_slot = call_slot;
        tl_jpeg_files_thread = files_thread;
        // the files' headers are parsed (file where[k] of the read stage's pass) and their bytes lie in the pinned arena
        const JpegSource src = {R.parsed, where.data(), R.base, R.cap};
        const int rc = jpeg_process_batch_from(c, ptr.data(), len.data(), m, H, W, res.data(), st.data(), &src);
        tl_jpeg_enqueued = nullptr;
        tl_jpeg_files_thread = false;
        if (rc) return rc;
        for (int k = 0; k < m; ++k) { out_host[where[k]] = res[k]; status[where[k]] = st[k]; }
    }
    return MELF_SUCCESS;
}

extern "C" int melf_jpeg_process_files(melf_ctx* c, const char* const* paths, int n, int32_t* H_used, int32_t* W_used,
                                       melf_result* out_host, int32_t* status)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    if (!c->files_jobs.empty()) return fail(MELF_ERR_INVALID, "melf_jpeg_process_files: a _begin is waiting for its _end on this context");
    try {
        FilesRead R;
        if (int rc = jpeg_files_read(c, 0, paths, n, H_used, W_used, out_host, status, R)) return rc;
        return jpeg_files_decode(c, n, H_used, W_used, out_host, status, R);
    } catch (const std::exception& e) {
        return fail(MELF_ERR_INVALID, std::string("out of host memory: ") + e.what());
    }
}

// The same call split in two for a scripting host: _begin returns at once, the work (file reads, Huffman tables,
// upload, kernels, records) runs on a thread of the library, _end waits for the OLDEST call begun and returns its
// code.  Between the two the caller can turn the previous chunk's records into its own objects -- with a helper thread
// of the host language instead, the interpreter lock's hand-over (5 ms in CPython) eats the overlap.  Up to
// MELF_FILES_IN_FLIGHT_MAX calls may be in flight per context: a later one's files are read while an earlier one decodes
// (the decode stages run one after the other, in the order of the _begin calls).  All pointers must stay valid until the call's _end; no
// other call on the context while any is in flight.
extern "C" int melf_jpeg_process_files_begin(melf_ctx* c, const char* const* paths, int n, int32_t* H_used, int32_t* W_used,
                                             melf_result* out_host, int32_t* status)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    if ((int)c->files_jobs.size() >= melf_ctx::NFJ)
        return fail(MELF_ERR_INVALID, "melf_jpeg_process_files_begin: as many calls as the context takes are already in flight (MELF_FILES_IN_FLIGHT_MAX)");
    melf_ctx::FilesJob* job = nullptr;
    try {
        job = new melf_ctx::FilesJob();
        const uint64_t ticket = c->files_next_ticket;
        job->th = std::thread([=]() {
            // whatever happens in the stages, the ticket must be handed on: the next call's decode stage waits for it
            int rc = MELF_SUCCESS;
            FilesRead R;
            static const bool trace = diag_env("MELF_JPEG_TRACE") != nullptr;
            const auto tb0 = std::chrono::steady_clock::now();
            try {
                rc = jpeg_files_read(c, (int)(ticket % melf_ctx::NFJ), paths, n, H_used, W_used, out_host, status, R);
            } catch (const std::exception& e) {
                rc = fail(MELF_ERR_INVALID, std::string("out of host memory: ") + e.what());
            }
            const auto tb1 = std::chrono::steady_clock::now();
            {
                std::unique_lock<std::mutex> lk(c->files_m);
                c->files_cv.wait(lk, [&]() { return c->files_decode_turn == ticket; });
            }
            const auto tb2 = std::chrono::steady_clock::now();
            if (trace)
                fprintf(stderr, "[melf jpeg] call %llu: thread up at %.2f, files read by %.2f, its turn at %.2f (process clock, ms)\n",
                        (unsigned long long)ticket, trace_clock_ms(tb0), trace_clock_ms(tb1), trace_clock_ms(tb2));
            // the turn goes on as soon as this call has ENQUEUED all its GPU work (the next call then prepares and enqueues
            // while this one's kernels run), at the latest when the call is over
            bool released = false;
            auto tb3 = tb2;
            std::function<void()> release = [&]() {
                if (released) return;
                released = true;
                tb3 = std::chrono::steady_clock::now();
                {
                    std::lock_guard<std::mutex> lk(c->files_m);
                    ++c->files_decode_turn;
                }
                c->files_cv.notify_all();
            };
            if (rc == MELF_SUCCESS) {
                try {
                    rc = jpeg_files_decode(c, n, H_used, W_used, out_host, status, R, diag_env("MELF_FILES_NO_OVERLAP") ? nullptr : &release,
                                           (int)(ticket % melf_ctx::NJC), true);
                } catch (const std::exception& e) {
                    rc = fail(MELF_ERR_INVALID, std::string("out of host memory: ") + e.what());
                }
            }
            job->rc = rc;
            if (rc) job->err = g_err;  // this thread's message, for the thread that calls _end
            release();
            {
                const auto tb4 = std::chrono::steady_clock::now();
                const auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
                    return std::chrono::duration<double, std::milli>(b - a).count();
                };
                std::lock_guard<std::mutex> lk(c->files_m);
                auto& st = c->files_stats;
                st.calls += 1; st.files += n;
                st.ms_read += ms(tb0, tb1); st.ms_turn_wait += ms(tb1, tb2); st.ms_enqueue += ms(tb2, tb3); st.ms_gpu_wait += ms(tb3, tb4);
            }
        });
    } catch (const std::exception& e) {
        delete job;
        return fail(MELF_ERR_INVALID, std::string("cannot start a thread: ") + e.what());
    }
    ++c->files_next_ticket;
    c->files_jobs.push_back(job);
    return MELF_SUCCESS;
}

extern "C" int melf_jpeg_files_in_flight_max(void) { return MELF_FILES_IN_FLIGHT_MAX; }

// What the file system gives the read stage: open() + close() of every path on the device's I/O pool, nothing else (0.5-0.6 ms
// per 1024 files with 12 threads on the pool's boxes -- about a third of the read stage, the rest being read() into the pinned
// arena and the header parse: HISTORY.md, round 5; bench.py puts the figure on the line as `open_close_probe`).
extern "C" int melf_files_open_probe(const char* const* paths, int n, int device, double* ms, int* threads)
{
    if (!paths || n < 0 || !ms) return fail(MELF_ERR_INVALID, "bad argument");
    pool_use_device(device);
    WorkerPool& pool = io_pool();
    std::atomic<int> failed{0};
    const auto t0 = std::chrono::steady_clock::now();
    pool.run(n, [&](int i) {
        const int fd = paths[i] ? open(paths[i], O_RDONLY | O_CLOEXEC) : -1;
        if (fd < 0) { failed.fetch_add(1, std::memory_order_relaxed); return; }
        close(fd);
    });
    *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (threads) *threads = pool.size() + 1;
    return failed.load() ? fail(MELF_ERR_INVALID, "melf_files_open_probe: a file could not be opened") : MELF_SUCCESS;
}

extern "C" int melf_ctx_files_stats(melf_ctx* c, double out[MELF_FILES_STATS_COUNT], int reset)
{
    if (!c || !out) return fail(MELF_ERR_INVALID, "NULL argument");
    pool_use_device(c->device);
    {
        std::lock_guard<std::mutex> lk(c->files_m);
        if (!c->files_jobs.empty()) return fail(MELF_ERR_INVALID, "melf_ctx_files_stats while a melf_jpeg_process_files_begin call is in flight");
        const auto& st = c->files_stats;
        out[0] = st.calls; out[1] = st.files; out[2] = st.ms_read; out[3] = st.ms_turn_wait; out[4] = st.ms_enqueue; out[5] = st.ms_gpu_wait;
        if (reset) c->files_stats = melf_ctx::FilesStats();
    }
    out[6] = io_pool().size() + 1;      // threads of the read stage, the calling thread included
    out[7] = host_pool().size() + 1;    // threads of the other parallel host loops
    out[8] = pool_cores();              // cores this process may run on (affinity mask)
    out[9] = pool_devices();            // devices this process has contexts on (the pools' divisor)
    return MELF_SUCCESS;
}

extern "C" int melf_jpeg_process_files_end(melf_ctx* c)
{
    if (!c) return fail(MELF_ERR_INVALID, "ctx is NULL");
    if (c->files_jobs.empty()) return fail(MELF_ERR_INVALID, "melf_jpeg_process_files_end without _begin");
    melf_ctx::FilesJob* job = c->files_jobs.front();
    c->files_jobs.pop_front();
    job->th.join();
    const int rc = job->rc;
    const std::string err = job->err;
    delete job;
    if (rc) return fail(rc, err);
    return MELF_SUCCESS;
}
