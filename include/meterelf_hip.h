/*
 * meterelf_hip.h -- C ABI of libmeterelf_hip.so, the MI355X (gfx950) drop-in
 * for meterelf's per-image hot path.
 *
 * The reference (suutari/meterelf) has no FFI/plugin boundary: its hot path is
 * Python calling cv2.  The boundary is therefore its Python API, which
 * meterelf_amd/ keeps (get_meter_values, MeterImageData, ImageFile,
 * get_meter_value), and this header is what that host layer binds through
 * ctypes.  Each entry point cites the reference code it replaces (paths relative
 * to the reference checkout).  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function
 * returns 0 on success or a negative melf_status; the message is available from
 * melf_last_error() (thread-local).  A context belongs to one GPU and is driven
 * by one host thread at a time.  The caller owns every buffer it passes in; the
 * context owns its device memory.  "_dev" entry points take device pointers
 * (e.g. torch tensors' data_ptr()) and enqueue on `stream` (a hipStream_t passed
 * as void*; NULL = the null / legacy default stream, exactly as in a HIP launch,
 * which is also what torch's default stream handle 0 means) without synchronising
 * it unless stated: work the caller enqueued on `stream` before the call is seen
 * by the kernels, and whatever the caller enqueues on it afterwards sees the
 * records.  A context owns two sets of work buffers ("lanes") and hands a lane
 * to each caller stream: *_dev calls that arrive on two different streams run
 * concurrently on the GPU, a third stream (or a call that needs the lane
 * another stream used last) is ordered behind that stream's work by an event.
 * Two batches' kernels never run against the same buffers at once.
 */
#ifndef METERELF_HIP_H
#define METERELF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define MELF_MAX_DIALS 8
#define MELF_ABI_VERSION 3

/* API status codes (negative) */
enum {
    MELF_SUCCESS = 0,
    MELF_ERR_INVALID = -1,  /* bad argument / shape mismatch          */
    MELF_ERR_HIP = -2,      /* HIP runtime error, see melf_last_error  */
    MELF_ERR_NO_DEVICE = -3,
    MELF_ERR_TOO_LARGE = -4 /* template/dial does not fit the kernels' LDS tiles */
};

/* Per-frame status: mirrors the reference's exception classes
 * (meterelf/exceptions.py:35-52). */
enum {
    MELF_FRAME_OK = 0,
    MELF_FRAME_DIALS_NOT_FOUND = 1,           /* DialsNotFoundError, meterelf/_image.py:62-64      */
    MELF_FRAME_NEEDLE_CONTOURS_NOT_FOUND = 2, /* NeedleContoursNotFoundError, _reading.py:137-138  */
    MELF_FRAME_ANGLE_UNDETERMINED = 3         /* DialAngleDeterminingError, _reading.py:98-106     */
};

/* One entry of params.yml `needle_data` (meterelf/_params.py:71-81). */
typedef struct {
    double cx, cy;            /* center                                   */
    double angle_of_zero;     /* degrees                                  */
    int32_t range_h, range_l, range_s; /* color_range                     */
    int32_t negative_momentum;
    int32_t diameter;
    int32_t dist_from_center;
    int32_t circle_thickness;
    int32_t reserved;
} melf_dial;

/* Scalars of params.yml (meterelf/_params.py:30-64). */
typedef struct {
    int32_t abi_version;      /* MELF_ABI_VERSION                          */
    int32_t rect_x0, rect_y0, rect_x1, rect_y1; /* meter_rect              */
    int32_t th, tw;           /* template rows, cols (dials_template_size as (h, w)) */
    int32_t hue_shift;
    int32_t ndials;
    int32_t needle_lo[3];     /* fixed H,L,S bounds of the fused full-frame */
    int32_t needle_hi[3];     /*   stage: needle_color -/+ needle_color_range, clamped */
    int32_t name_order[MELF_MAX_DIALS]; /* dial indices sorted by name string (_reading.py:171) */
    int32_t reserved;
    double match_threshold;   /* dials_template_match_threshold            */
    melf_dial dial[MELF_MAX_DIALS];
} melf_params;

/* Per-frame record.  The Python host rebuilds the reference's return dict /
 * exception objects from it (meterelf/_reading.py:19-115). */
typedef struct {
    int32_t status;           /* MELF_FRAME_*                              */
    int32_t match_x, match_y; /* minMaxLoc max_loc, meterelf/_utils.py:94-95 */
    int32_t failed_dial;      /* NEEDLE_CONTOURS_NOT_FOUND: dial index, else -1 */
    uint32_t unreadable_mask; /* ANGLE_UNDETERMINED: bit d = dial d unreadable */
    float match_val;          /* minMaxLoc max_val (float32)               */
    double pos[MELF_MAX_DIALS];   /* dial positions in [0, 10)             */
    double angle[MELF_MAX_DIALS]; /* needle angle in turns, before angle_of_zero */
    double value;             /* determine_value_by_dial_positions, valid iff OK and ndials == 4 */
} melf_result;

typedef struct melf_ctx melf_ctx;

const char* melf_last_error(void);
int melf_abi_version(void);
int melf_device_count(int* count);

/* Host precompute of the per-dial masks, replaces _dial_data._get_dial_data
 * (meterelf/_dial_data.py:22-55): masks[ndials][2][th*tw], plane 0 = `mask`
 * (disk), plane 1 = `circle_mask` (annulus). */
int melf_build_dial_masks(const melf_params* p, uint8_t* masks);

/* The calibration blob = params + template + dial masks, the only shared
 * read-only state of the path (meterelf/_image.py:69-81, _dial_data.py:11-19).
 * Rank 0 packs it, the host layer broadcasts it (RCCL) and every rank creates
 * its context from the same bytes. */
size_t melf_blob_size(const melf_params* p);
int melf_blob_pack(const melf_params* p, const uint8_t* templ /* th*tw */, void* blob, size_t blob_bytes);
int melf_blob_params(const void* blob, size_t blob_bytes, melf_params* out);

/* One context per GPU.  `blob` is a host pointer unless blob_on_device != 0. */
int melf_ctx_create(int device, const void* blob, size_t blob_bytes, int blob_on_device, melf_ctx** out);
/* One process, n GPUs (SURVEY 8b's `melf_ctx_bcast`, 8e; the reference has one process and no GPU: meterelf/_api.py:16-33):
 * the host blob goes to devices[0] and from there to every other listed GPU by ONE ncclBroadcast (ncclUint8, root 0; RCCL
 * over xGMI, bound with dlopen at the first call), and out[i] is created on devices[i] from that GPU's copy.  A device may
 * not be listed twice.  All or nothing: on failure every out[i] is NULL.  (One process PER GPU broadcasts the same bytes
 * through its own communicator -- torch.distributed in meterelf_amd/_dist.py -- and calls melf_ctx_create with
 * blob_on_device = 1.) */
int melf_ctx_create_bcast(const int* devices, int n, const void* blob, size_t blob_bytes, melf_ctx** out);
void melf_ctx_destroy(melf_ctx* ctx);
int melf_ctx_params(const melf_ctx* ctx, melf_params* out);
/* Waits for all work the context has enqueued on caller streams and forgets those streams.  Call it before
 * destroying a stream that *_dev calls of this context were issued on. */
int melf_ctx_sync(melf_ctx* ctx);
/* copy the context's dial masks back (tests) */
int melf_ctx_get_masks(const melf_ctx* ctx, uint8_t* masks);

/* ---- the whole path: replaces get_meter_value(imgf) per frame
 * (meterelf/_reading.py:19-115 with meterelf/_image.py:23-66) ---------------
 * frames: n full camera frames, H x W x 3 u8 BGR (cv2.imread layout), frame f at
 * frames + f*frame_stride bytes.  The meter_rect crop is taken inside, with
 * numpy-slice clamping (meterelf/_image.py:54-55).  Host frames: only that crop
 * crosses PCIe (packed by host threads into pinned staging buffers, copied by
 * DMA in chunks that overlap the previous chunk's kernels). */
int melf_process_batch(melf_ctx* ctx, const uint8_t* frames_host, int n, int H, int W,
                       size_t frame_stride, melf_result* out_host);
/* frames already in HBM; results go to d_results (device, may be NULL) and/or
 * out_host (host, may be NULL; when given the call synchronises the stream). */
int melf_process_batch_dev(melf_ctx* ctx, const void* d_frames, int n, int H, int W,
                           size_t frame_stride, void* d_results, melf_result* out_host, void* stream);

/* ---- stage entry points (parity tests and roofline runs) ----------------- */

/* convert_to_hls (meterelf/_utils.py:100-102): cvtColor(BGR2HLS_FULL) + uint8
 * hue shift.  src rows x cols x 3 u8 with row stride in bytes; dst packed. */
int melf_bgr2hls(melf_ctx* ctx, const uint8_t* src_host, int rows, int cols, size_t row_stride,
                 uint8_t* dst_host);

/* Fused full-frame stage (BASELINE config 2): HLS(+shift) -> inRange with the
 * context's fixed needle bounds (get_mask_by_color, meterelf/_utils.py:113-119,
 * bounds as meterelf/_calibration.py:82-84) -> dilate 3x3 -> erode 3x3
 * (meterelf/_reading.py:128-130).  n frames H x W x 3 -> n masks H x W u8 {0,255}. */
int melf_hls_inrange_close(melf_ctx* ctx, const uint8_t* frames_host, int n, int H, int W,
                           uint8_t* masks_host);
int melf_hls_inrange_close_dev(melf_ctx* ctx, const void* d_frames, int n, int H, int W,
                               void* d_masks, void* stream);

#ifdef MELF_DIAG
/* DIAGNOSTIC BUILD ONLY (make -C meterelf_amd/csrc diag -> libmeterelf_hip_diag.so; the product library does not export it).
 * Measurement aid for the fused stage's roofline (bench.py: fused_mask.stream_ceiling), nothing the reference has: one launch
 * of a BARE persistent stream with the fused kernel's traffic mix and launch shape -- 48 bytes read and 16 bytes written per
 * thread and step, no pixel arithmetic -- over the caller's device buffers: floor(in_bytes / 48 KiB) chunks of d_in are read,
 * a third as many bytes of d_out are overwritten with garbage (XOR of the input: point it at a mask buffer that is rewritten
 * afterwards).  chunks_per_block = 0: static grid-stride split; -1: the same with the kernel's register prefetch (the next
 * chunk requested before this one is stored); -2 .. -5: every workgroup walks its own contiguous run of chunks like the kernel's
 * segments (-3: + the kernel's 48-byte lane stride, -4: + two barriers and an LDS hand-over per step, -5: + 64 KiB of LDS
 * tables filled first); > 0: blocks of that many chunks from a work queue.  The launch
 * is timed like the fused kernel's (melf_ctx_set_profiling(1), entry MELF_K_STREAM_PROBE of melf_ctx_timings). */
int melf_stream_probe_dev(melf_ctx* ctx, const void* d_in, size_t in_bytes, void* d_out, int chunks_per_block, void* stream);
#endif

/* Number of entries of the fused stage's hue lookup table whose in-range answer
 * depends on the float32 rounding of the individual BGR triple (exact rounding
 * ties at a bound).  > 0 selects the kernel variant that re-evaluates those
 * pixels with the exact float path; results are identical either way. */
int melf_ctx_fused_table_ties(const melf_ctx* ctx, int* count);

/* match_template (meterelf/_utils.py:91-97): TM_CCOEFF of n single-channel u8
 * images (rows x cols, packed) against the context's template + minMaxLoc.
 * result_map (optional) receives n*(rows-th+1)*(cols-tw+1) float32. */
int melf_match_ccoeff(melf_ctx* ctx, const uint8_t* images_host, int n, int rows, int cols,
                      float* max_val, int32_t* max_x, int32_t* max_y, float* result_map);

/* Per-dial reading on n already-located dials crops (th x tw x 3 HLS u8, packed):
 * get_needle_points + angle estimate + digit combine
 * (meterelf/_reading.py:28-111, :118-182). */
int melf_read_dials(melf_ctx* ctx, const uint8_t* dials_hls_host, int n, melf_result* out_host);

/* ---- calibration stages (reference: meterelf/_calibration.py, offline) ------ */

/* get_average_meter_image (meterelf/_calibration.py:60-63 with _image.py:34-44 and
 * _utils.py:64-88): the meter_rect crop of every frame is translated so that its dial match
 * (match_x[i], match_y[i]) lands at (align_x, align_y), the float64 running mean is taken in the
 * reference's operation order and denormalised to u8.  out_crop: crop_rows x crop_cols x 3. */
int melf_aligned_average(melf_ctx* ctx, const uint8_t* frames_host, int n, int H, int W, size_t frame_stride,
                         const int32_t* match_x, const int32_t* match_y, int align_x, int align_y,
                         uint8_t* out_crop_host);

/* cv2.inRange on a packed 3-channel u8 image (get_mask_by_color, meterelf/_utils.py:113-119). */
int melf_inrange(melf_ctx* ctx, const uint8_t* img_host, int rows, int cols, const int32_t lo[3],
                 const int32_t hi[3], uint8_t* mask_host);

/* nbatches batches of n frames each, already in HBM (batch b at d_frames + b * batch_stride bytes, its records at
 * d_results + b * results_stride records; a stride of 0 re-reads / overwrites the same batch), enqueued from
 * `stream`: consecutive batches run on the context's two pipeline lanes, so that their kernels overlap; `stream`
 * continues when all of them are done.  Results as melf_process_batch_dev. */
int melf_process_stream_dev(melf_ctx* ctx, const void* d_frames, int nbatches, size_t batch_stride, int n, int H, int W,
                            size_t frame_stride, void* d_results, size_t results_stride, void* stream);

/* ---- JPEG decode (reference: cv2.imread in ImageFile.get_bgr_image, meterelf/_image.py:46-51) ----
 * Baseline sequential 8-bit Huffman JPEGs (one interleaved scan; YCbCr 4:2:0 / 4:2:2 / 4:4:4 or
 * greyscale; restart intervals allowed) are decoded on the GPU to the bytes libjpeg produces with its
 * defaults (ISLOW IDCT, fancy upsampling), i.e. what cv2.imread returns: H x W x 3 BGR u8.
 * Per-file status: 0 decoded, 1 valid JPEG outside that subset (decode it on the host instead) -- also a file
 * WITHOUT restart markers whose entropy-coded data exceed 4 MB (the segment-parallel Huffman kernel addresses
 * 1024 segments of at most 32 000 bits; melf_jpeg_probe cannot tell, the decode calls report it) --,
 * 2 unreadable / corrupt, 3 its size is not H x W.  Frames with a non-zero status are zero-filled. */
enum { MELF_JPEG_OK = 0, MELF_JPEG_UNSUPPORTED = 1, MELF_JPEG_CORRUPT = 2, MELF_JPEG_SIZE_MISMATCH = 3,
       MELF_JPEG_UNREADABLE = 4 /* melf_jpeg_process_files: the file could not be opened or read */ };

/* Header check only (no GPU, no context): image size and whether the GPU decoder handles the file. */
int melf_jpeg_probe(const uint8_t* data, size_t size, int32_t* H, int32_t* W, int32_t* supported);

/* The same check for n files at once (one call instead of n from a scripting host). */
int melf_jpeg_probe_batch(const uint8_t* const* data, const size_t* sizes, int n, int32_t* H, int32_t* W,
                          int32_t* supported);

/* Decodes n files of H x W pixels.  out: n*H*W*3 bytes, on the host (out_on_device = 0) or in HBM.
 * Synchronises the context's stream. */
int melf_jpeg_decode_batch(melf_ctx* ctx, const uint8_t* const* data, const size_t* sizes, int n, int H, int W,
                           void* out, int out_on_device, int32_t* status);

/* Stage entry point (parity tests): the first decode stage alone -- byte stuffing (FF 00), fill bytes and RSTn markers taken out
 * of ONE entropy-coded segment on the GPU, as libjpeg's bit reader does while it reads (cv2.imread, meterelf/_image.py:49).
 * raw[n]: any bytes; restart_expected > 0: the segment of a file with restart intervals, rst receives the bit offsets of its
 * restart_expected + 1 intervals (as the kernel leaves them) and rst_cnt the number found.  out must hold n + 192 bytes (clean
 * bytes, then zero fill); out_len receives the cleaned length.  Returns 0, or -1 on a bad argument / HIP error. */
int melf_jpeg_clean_segment(const uint8_t* raw, int n, int restart_expected, uint8_t* out, int32_t* out_len, uint32_t* rst,
                            int32_t* rst_cnt);

/* get_meter_value for n JPEG files (meterelf/_api.py:22-33 with _image.py:46-51): decode on the GPU
 * straight into HBM, then the same path as melf_process_batch.  Records of files whose status is
 * non-zero are meaningless. */
int melf_jpeg_process_batch(melf_ctx* ctx, const uint8_t* const* data, const size_t* sizes, int n, int H, int W,
                            melf_result* out_host, int32_t* status);

/* The same for n file names (the loop of get_meter_values, meterelf/_api.py:22-33): the library reads the files
 * (on threads) and processes them frame size by frame size (a list may mix sizes; H_used / W_used return the size of
 * the first file its decoder accepts); files it does not decode come back with status 1 / 2 / 4 for the caller to
 * route (host decode for another format).  Status 3 is not used by this call.  The files are read straight into a
 * pinned buffer of the context, from which they are uploaded as they are (byte stuffing and restart markers are taken
 * out on the GPU): the host touches no byte of a file after read() has written it. */
int melf_jpeg_process_files(melf_ctx* ctx, const char* const* paths, int n, int32_t* H_used, int32_t* W_used,
                            melf_result* out_host, int32_t* status);

/* The same call in two halves, for a host that wants to work on the previous chunk's records meanwhile
 * (get_meter_values does: meterelf_amd/_api.py): _begin returns at once and the call runs on a thread of the library,
 * _end waits for the OLDEST call begun and returns its status code (message via melf_last_error as usual).  Up to
 * MELF_FILES_IN_FLIGHT_MAX calls may be in flight per context -- one reading its files, one preparing and enqueueing
 * its GPU work, one waiting for its kernels (one more _begin fails with MELF_ERR_INVALID); every pointer must stay
 * valid until the call's own _end; no other call on the context while any is in flight. */
#define MELF_FILES_IN_FLIGHT_MAX 3
int melf_jpeg_files_in_flight_max(void); /* the value the library was built with */
int melf_jpeg_process_files_begin(melf_ctx* ctx, const char* const* paths, int n, int32_t* H_used, int32_t* W_used,
                                  melf_result* out_host, int32_t* status);
int melf_jpeg_process_files_end(melf_ctx* ctx);
/* Where the _begin calls of this context spent their host time since the last reset (sums, milliseconds): out[0] calls,
 * [1] files, [2] read stage (open / fstat / read / close / header parse on the I/O pool), [3] waiting for the call's turn at the
 * context, [4] enqueueing (chunk layout, uploads, launches) until the context is handed to the next call, [5] waiting for the
 * call's kernels and records; [6] threads of the read stage, [7] of the other host loops (the caller included), [8] cores the
 * process may use (affinity mask cut down to the cgroup CPU quota), [9] devices the process has contexts on (what the pools divide the cores by).
 * Not while a _begin call is in flight. */
/* Measurement aid: open() + close() of every path on the I/O pool of `device` (no context needed, nothing is read):
 * milliseconds for the n files and the threads that took part -- what the file system allows the read stage. */
int melf_files_open_probe(const char* const* paths, int n, int device, double* ms, int* threads);
#define MELF_FILES_STATS_COUNT 10
int melf_ctx_files_stats(melf_ctx* ctx, double out[MELF_FILES_STATS_COUNT], int reset);

/* Promise that the frames handed to melf_process_batch_dev are complete in device memory at the time of each call (they
 * do not depend on work still pending on the call's stream -- e.g. frames that were uploaded or decoded earlier and
 * synchronised).  Consecutive calls, also on ONE caller stream, then alternate between the context's two lanes: a call's
 * prep and match kernels start at once on the lane's own stream, beside the previous call's kernels; only the kernel
 * that writes the records waits for the work the caller's stream held at the time of the call, and the caller's stream
 * continues when the call is done (what is enqueued on it afterwards sees the records, as without the promise).
 * Same results; steps 10-30 % shorter.  Off by default: without the promise every kernel of a call is ordered behind
 * the stream's earlier work. */
int melf_ctx_set_frames_resident(melf_ctx* ctx, int on);

/* ---- measurement ---------------------------------------------------------
 * With profiling on, every kernel launched by a *_dev entry point is bracketed
 * by hipEvents on its stream; melf_ctx_timings drains them (synchronising) and
 * returns per-kernel accumulated milliseconds and launch counts. */
enum { MELF_K_LPLANE = 0, MELF_K_MATCH = 1, MELF_K_DIALS = 2, MELF_K_FUSED_MASK = 3, MELF_K_HLS = 4,
       MELF_K_JPEG_HUFF = 5, MELF_K_JPEG_IDCT = 6, MELF_K_JPEG_COLOR = 7, MELF_K_STREAM_PROBE = 8, MELF_K_COUNT = 9 };
/* Which kernel, in which layout, computed the template match (meterelf/_utils.py:91-97: ONE cv2.matchTemplate code
 * path in the reference; here the batch size and the crop shape select among three kernels and, for the tuned one,
 * among wave layouts) of the context's most recent call.  Tests assert it, so that a change of a dispatch threshold
 * cannot silently move a parity test onto another kernel. */
enum { MELF_MATCH_KERNEL_DOT4 = 0, MELF_MATCH_KERNEL_MFMA = 1, MELF_MATCH_KERNEL_GEN = 2 };
typedef struct {
    int32_t kernel;          /* MELF_MATCH_KERNEL_*                                                      */
    int32_t n, rows, cols;   /* images of the launch, searched image size                                */
    int32_t groups;          /* 32-frame groups                                                          */
    int32_t waves;           /* waves of the launch (matrix-core kernels)                                */
    int32_t rows_per_wave;   /* tuned kernel: map rows of a full-row wave (RB); general kernel: tile rows */
    int32_t full_waves;      /* tuned kernel, per group: waves of RB full rows                           */
    int32_t pair_waves;      /* tuned kernel, per group: waves of RB + 1 rows that share a middle row    */
    int32_t tiles;           /* general / dot4 kernel: tiles (partials) per frame                        */
    int32_t reserved[6];
} melf_match_info;
int melf_ctx_last_match(const melf_ctx* ctx, melf_match_info* out);
/* The tuned kernel's wave layout for a template / searched-image shape and a batch of n images, without a GPU or a
 * context (host logic; kernel = MELF_MATCH_KERNEL_MFMA when the shape belongs to the tuned kernel's class, else the
 * kernel that takes it).  reserved[0] = padded template rows, reserved[1] = L-plane rows per frame group. */
int melf_match_layout_query(int th, int tw, int rows, int cols, int n, melf_match_info* out);
/* The GENERAL matrix-core kernel's plan for a shape and batch size, without a GPU (host logic; tests pin its invariants for
 * every batch size).  out->kernel = the kernel DEFAULT dispatch launches for this shape and n (the plan returned is the
 * general kernel's either way: MELF_MATCH=gen forces it).  For the general kernel -- here, in melf_match_layout_query and
 * in melf_ctx_last_match -- rows_per_wave = rows a tile computes, tiles = tiles (= workgroups, = partials) per frame group,
 * waves = waves of the launch, reserved[0] = Toeplitz blocks per template row, [1] = L-plane rows per frame group,
 * [2] = column blocks per tile, [3] = K slices per tile = waves per workgroup, [4] = remainder ("V form") map columns,
 * [5] = image blocks per V-form row.
 * tasks (optional, cap entries): one entry per wave of a frame group's workgroups, *ntasks = how many there are.
 * A wave computes map rows y0 .. y0 + rows - 1 (rows_computed >= rows are accumulated) of column blocks xb0 .. xb0 + nxb - 1
 * (32 map columns each) over the slice [k_lo, k_hi) of the tile's K range (Toeplitz block x template row); rows == 0: a
 * V-form tile = ONE map column (remainder column index xb0) x the 32 map rows from y0, K range = (image row - y0, image
 * block).  The nslices waves of a tile are one workgroup and add their accumulators up in its LDS (lds_bytes). */
typedef struct {
    int32_t y0, rows, rows_computed, xb0, nxb, tile, slice, nslices, k_lo, k_hi, lds_bytes, reserved;
} melf_gen_task;
int melf_match_gen_plan_query(int th, int tw, int rows, int cols, int n, melf_match_info* out, melf_gen_task* tasks, int cap,
                              int32_t* ntasks);

int melf_ctx_set_profiling(melf_ctx* ctx, int on);  /* 0 off, 1 every kernel, 2 only the match kernel (two event records per batch instead of eight) */
int melf_ctx_timings(melf_ctx* ctx, double ms[MELF_K_COUNT], int64_t launches[MELF_K_COUNT]);
const char* melf_kernel_name(int k);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* METERELF_HIP_H */
