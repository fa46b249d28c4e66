// Internal declarations shared by the translation units of libmeterelf_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <stdint.h>

#include "../../include/meterelf_hip.h"

namespace melf {

// Environment switches.  The PRODUCT library reads only the documented ones (README.md, "Environment switches": MELF_MATCH,
// MELF_MATCH_LAYOUT, MELF_GEN_SHAPE, MELF_FUSED_VARIANT, MELF_FORCE_GENERIC_MASK, MELF_JPEG_CHUNK, MELF_IO_THREADS,
// MELF_HOST_THREADS).  Trace output and the A/B switches of past experiments exist only in the diagnostic build
// (`make -C meterelf_amd/csrc diag`, -DMELF_DIAG, loaded through MELF_LIB_PATH): there diag_env is getenv, here it is nothing.
#ifdef MELF_DIAG
inline const char* diag_env(const char* name) { return getenv(name); }
#else
inline const char* diag_env(const char*) { return nullptr; }
#endif

// ---- K2: template match -----------------------------------------------------
// One partial (max, first-argmax) per workgroup tile of the correlation map.
struct MatchPartial {
    float val;
    int32_t idx;  // y * rw + x in the frame's correlation map; INT32_MAX = empty
};

constexpr int MATCH_R = 11;           // output rows per lane
constexpr int MATCH_WAVES = 4;        // waves per workgroup
constexpr int MATCH_RBLK = MATCH_R * MATCH_WAVES;  // output rows per workgroup
constexpr int MATCH_CBLK = 64;        // output cols per workgroup (one per lane)

struct MatchGeom {
    int th, tw;        // template rows, cols
    int tw4;           // template row length in dwords (tw padded to 4)
    int trows;         // padded template rows: th + 2*(MATCH_R-1)
    uint32_t last_ones;  // byte mask (0x01 per valid byte) of the last template dword
    int ldsw;          // LDS tile row width in dwords
    int lds_rows;      // LDS tile rows: MATCH_RBLK + th - 1
    double tmean;      // cv::mean(template) = sum * (1.0 / N)
};

// source description for K2: either packed single-channel u8 images or BGR frames
struct MatchSrc {
    const uint8_t* base;
    size_t frame_stride;  // bytes between consecutive images/frames
    int row_stride;       // bytes between rows
    int x0, y0;           // origin of the searched image inside the frame (pixels)
    int rows, cols;       // searched image size
    size_t readable;      // bytes from `base` the caller guarantees readable: (images - 1) * frame_stride + the last image's rows
};

void launch_match(const MatchSrc& src, bool from_bgr, int n, const MatchGeom& g, const uint32_t* d_tplT,
                  float* d_result_map, MatchPartial* d_partials, int* nparts_out, hipStream_t stream);
int match_parts(const MatchGeom& g, int rows, int cols);

// ---- K2 on the matrix cores (k_match_mfma.hip) --------------------------------
struct MfmaPlan {
    int rh, rw, nxb, nkb, th_pad, nparts, rows_pad, groups;
    int rb, na, np;   // layout: na waves of rb full map rows + np pairs of (rb + 1)-row waves sharing their middle row
    int ks, ntiles;   // K slices per tile (waves of a tile's workgroup; nparts = ntiles x ks), tiles per frame group = na + 2 np
    size_t lg_bytes, r_bytes;
};
bool mfma_match_ok(int th, int tw, int rows, int cols);
MfmaPlan mfma_plan(int th, int tw, int rows, int cols, int nframes);
size_t mfma_atab_bytes(int th);
void mfma_build_atab(const uint8_t* templ, int th, int tw, int8_t* atab);
void launch_mfma_prep(const MatchSrc& src, bool from_bgr, int n, const MfmaPlan& p, int th, int tw, int8_t* d_lg,
                      uint16_t* d_r, hipStream_t stream);
// launch_mfma_match: d_ws = the row-window sums R in epilogue order (k_prep_lplane); the waves add them up
void launch_mfma_match(int n, const MfmaPlan& p, int th, int tw, long tsum, double tmean, const int8_t* d_atab,
                       const int8_t* d_lg, const uint32_t* d_ws, float* d_result_map, MatchPartial* d_partials,
                       hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---- K2, general form (k_match_gen.hip): any template up to 256 columns, any map size ----
struct GenTile {           // one workgroup's tile of the correlation map (per frame group)
    int16_t y0;            // first map row of the tile (V form: first of its 32 map rows)
    int8_t R, Rc;          // map rows of the tile, rows computed (R rounded up to 2/4/6/8); R == 0: V-form tile
    int8_t nxb;            // column blocks (1 or 2)
    int8_t pad0;
    int16_t xb0;           // first column block (V form: index of the remainder column)
    int32_t klen;          // length of the tile's linearised K range; wave w of ns takes [w klen / ns, (w + 1) klen / ns)
};
struct GenPlan {
    int rh, rw, rwp, nd, nkb, rows_pad, groups, ntiles, ntasks, rc;
    int nxb_tile, nslices;   // column blocks per H-form tile, K slices = waves per workgroup (the planner's choice)
    int nxb_h;               // column blocks the H form covers
    int vcols, vx0, vkb0, ndv, ndelta;
    size_t lg_bytes, r_bytes, atab_bytes, atabv_bytes, lds_bytes;
    std::vector<GenTile> tiles;
};
struct GenDev {            // device copies that belong to one plan
    int8_t* atab = nullptr;
    int8_t* atabv = nullptr;
    GenTile* tiles = nullptr;
};
bool gen_match_ok(int th, int tw, int rows, int cols);
GenPlan gen_plan(int th, int tw, int rows, int cols, int nframes);
void gen_build_atab(const uint8_t* templ, int th, int tw, const GenPlan& p, int8_t* atab);
void gen_build_atabv(const uint8_t* templ, int th, int tw, const GenPlan& p, int8_t* atabv);
void launch_gen_match(int n, const GenPlan& p, int rows, int th, int tw, long tsum, double tmean, const GenDev& dev, const int8_t* d_lg,
                      const uint16_t* d_r, float* d_result_map, MatchPartial* d_partials, hipStream_t stream,
                      hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// prep for either matrix-core kernel: Lg (fragment order) and the row-window sums R in the match waves' epilogue order
// (pairs > 0: the tuned kernel's paired-operand row layout, see k_prep_lplane)
void launch_match_prep(const MatchSrc& src, bool from_bgr, int n, int groups, int rows_pad, int nkb, int rwp, int tw, int8_t* d_lg,
                       uint16_t* d_r, hipStream_t stream, int pairs = 0);

// ---- K3: per-dial reading ---------------------------------------------------
struct DialGeom {
    int32_t wx0, wy0;    // window origin in dials-crop coordinates
    int32_t ws;          // window size (2R+5 <= 64)
    int32_t core_x, core_y;  // int(cx), int(cy): centre of the 5x5 colour core
};

struct DialsSrc {
    const uint8_t* base;   // BGR frames or packed HLS dials crops
    size_t frame_stride;
    int row_stride;        // bytes
    int x0, y0;            // origin of the meter crop inside the frame (BGR mode)
    int crop_rows, crop_cols;  // meter crop size (BGR mode): cvtColor image width for the tail rule
    size_t readable;       // bytes from `base` the caller guarantees readable: (frames - 1) * frame_stride + the last frame's rows
};

void launch_dials(const DialsSrc& src, bool from_hls, int n, const melf_params& P, const DialGeom* d_geom,
                  const uint64_t* d_rowmasks /* [ndials][3][64] */, const MatchPartial* d_partials,
                  int nparts, int rw, melf_result* d_results, hipStream_t stream, int ws_max /* largest DialGeom::ws */);

// ---- K1b / HLS --------------------------------------------------------------
void launch_bgr2hls(const uint8_t* d_src, int rows, int cols, size_t row_stride, int hue_shift,
                    uint8_t* d_dst, hipStream_t stream);
void launch_fused_mask(const uint8_t* d_frames, int n, int H, int W, int hue_shift, const int lo[3],
                       const int hi[3], uint8_t* d_masks, hipStream_t stream);
// table-driven fast path of K1b (W % 16 == 0, 16-byte aligned buffers)
constexpr int FUSED_TABLE_DWORDS = 3 * 65536 * 2 / 32 + 65536 / 32 + 3 * 65536 / 32 + 3 * (512 * 512 / 32) + 3 * (256 + 512) + 16;
// behind the tables: the work queues of the launches in flight (round 5: a launch's persistent workgroups take their segments
// from a counter), FUSED_QUEUE_SLOTS of them in rotation, one 64-byte line each: {next segment, workgroups done}; a launch's
// last workgroup leaves its slot zeroed
constexpr int FUSED_QUEUE_SLOTS = 64, FUSED_QUEUE_DWORDS = FUSED_QUEUE_SLOTS * 16;
constexpr int FUSED_BUF_DWORDS = FUSED_TABLE_DWORDS + FUSED_QUEUE_DWORDS;
int fused_tables_count_offset();
int fused_tables_active_offset();
int fused_tables_noniv_offset();
void launch_build_fused_tables(int hue_shift, const int lo[3], const int hi[3], uint32_t* d_tables, hipStream_t stream);
bool fused_mask_lut_ok(const void* d_frames, const void* d_masks, int H, int W);
// the next launch_fused_mask_lut of this thread carries these events as its dispatch's own start / stop stamps
void fused_mask_timing_events(hipEvent_t start, hipEvent_t stop);
void launch_fused_mask_lut(const uint8_t* d_frames, int n, int H, int W, int hue_shift, const int lo[3],
                           const int hi[3], uint32_t* d_tables /* FUSED_BUF_DWORDS: tables + work queues */, int variant, uint8_t* d_masks,
                           hipStream_t stream);

// measurement aid: bare 3:1 stream over the caller's buffers (bench.py's stream_ceiling); returns the bytes moved
size_t launch_stream_probe(const void* d_in, size_t in_bytes, void* d_out, int chunks_per_block, uint32_t* d_tables, hipStream_t stream,
                           hipEvent_t ev_start, hipEvent_t ev_stop);

// ---- calibration stage kernels ------------------------------------------------
void launch_aligned_average(const uint8_t* d_frames, int n, size_t frame_stride, int row_stride, int x0, int y0, int rows,
                            int cols, const int32_t* d_mx, const int32_t* d_my, int ax, int ay, uint8_t* d_out,
                            hipStream_t stream);
void launch_inrange3(const uint8_t* d_img, int npx, const int lo[3], const int hi[3], uint8_t* d_out, hipStream_t stream);

// ---- k_jpeg.hip: baseline JPEG decode ----
struct JpegWorkspace;
int jpeg_probe(const uint8_t* data, size_t size, int* H, int* W, int* supported, std::string* why);
struct JpegParsed;   // headers (and, when asked for, the Huffman decode data built from them) of n files
JpegParsed* jpeg_parse_files(const uint8_t* const* data, const size_t* sizes, int n, int H, int W, int32_t* host_status);
// the same file by file, from any thread (one thread per index): the file-name entry points parse a file right after
// reading it, on the thread that read it, and build its decode tables there too
JpegParsed* jpeg_parsed_new(int n, bool with_tables);
void jpeg_parsed_resize(JpegParsed* p, int n);   // room for n files; nothing is cleared (jpeg_parse_one resets its entry)
void jpeg_parse_one(JpegParsed* p, int i, const uint8_t* data, size_t size, int* H, int* W, int* supported);
void jpeg_parsed_free(JpegParsed* p);
// parsed: headers made earlier -- of file pidx[first + j] for the batch's file j (pidx NULL: of file first + j); host_status is
// then the caller's.  pin_base / pin_len: a pinned host buffer the files' bytes may already lie in (the file-name entry points
// read into one): a batch whose files all do is uploaded from there, no byte of it is copied on the host.
int jpeg_prepare_batch(JpegWorkspace** ws, const uint8_t* const* data, const size_t* sizes, int n, int H, int W,
                       int32_t* host_status, std::string* err, const JpegParsed* parsed = nullptr, int first = 0,
                       const int* pidx = nullptr, const uint8_t* pin_base = nullptr, size_t pin_len = 0);
int jpeg_launch_batch(JpegWorkspace* ws, int n, int H, int W, uint8_t* d_frames, int32_t* status_out_host,
                      hipStream_t stream, std::string* err, void (*timer)(void*, int, int), void* timer_arg,
                      const int* rect /* x0, y0, x1, y1: only this part of each frame is needed; NULL = all */);
int jpeg_upload_batch(JpegWorkspace* ws, int n, hipStream_t copy_stream, std::string* err);
int jpeg_decode_batch_kernels(JpegWorkspace* ws, int n, int H, int W, uint8_t* d_frames, hipStream_t stream, std::string* err,
                              void (*timer)(void*, int, int), void* timer_arg, const int* rect);
const int32_t* jpeg_device_status(const JpegWorkspace* ws);
void jpeg_workspace_free(JpegWorkspace* ws);

}  // namespace melf
