// Internal declarations shared by the orbx translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/orbx.h"

namespace orbx {

constexpr int kEdge = 19;        // EDGE_THRESHOLD, src/ORBextractor.cc:73
constexpr int kBorder = 16;      // EDGE_THRESHOLD-3 = minBorderX/Y, src/ORBextractor.cc:789-790
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE, src/ORBextractor.cc:72
constexpr int kPatchSize = 31;   // PATCH_SIZE, src/ORBextractor.cc:71
constexpr int kMaxLevels = 16;
constexpr int kMaxRoots = 8;
constexpr int kMaxDim = 4095;    // packed point format: 12 bits per coordinate

// packed point: x | y << 12 | score << 24
__host__ __device__ inline uint32_t pack_pt(int x, int y, int s) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24); }
__host__ __device__ inline int pt_x(uint32_t p) { return (int)(p & 0xfffu); }
__host__ __device__ inline int pt_y(uint32_t p) { return (int)((p >> 12) & 0xfffu); }
__host__ __device__ inline int pt_s(uint32_t p) { return (int)(p >> 24); }

struct XTab { uint16_t s0, s1; int16_t a0, a1; };  // resize tables: source index pair + 11-bit weights

struct LevelGeom {
  int w, h, pitch;          // pixels; pitch in bytes (levels >= 1; level 0 uses the caller's strides)
  int64_t plane_off;        // byte offset of this level inside one frame's pyramid block (levels >= 1)
  int ncols, nrows, wcell, hcell;
  int cell_begin, ncells;   // range in the cell table
  int cand_off, cand_cap;   // range in one frame's candidate-slot array (u32 entries)
  int quota, kp_off, kp_cap;  // quadtree output range in one frame's level-keypoint array
  int nroots;
  int root_x0[kMaxRoots], root_x1[kMaxRoots];
  float hX, scale;
  int scaled_patch;         // (int)(PATCH_SIZE*scale): KeyPoint::size
  int xtab_off, ytab_off;   // offsets into the device resize tables (entries)
  int rs_lds_pitch, rs_lds_rows;  // LDS source tile of k_resize for this level (bytes per row, rows)
  int64_t bplane_off;       // byte offset of this level inside one frame's BLURRED pyramid block (all levels)
  int btile_begin, btiles_x, btiles_y;  // 64x32 tiles of the blur kernel
};

struct CellGeom {
  int16_t level, x0, y0, cw, ch;  // sub-image origin (level coords) and size incl. the +6 margin
  int16_t relx, rely;             // j*wCell, i*hCell (added to FAST coordinates, src/ORBextractor.cc:865-866)
  int32_t slot_off, slot_cap;     // range in one frame's candidate-slot array
  int32_t pitch;                  // of the cell's pyramid plane (levels >= 1), copied here so that the FAST workgroup needs one
  int64_t plane_off;              // dependent scalar load less before its first pixel fetch
};

struct Geometry {
  int rows = 0, cols = 0, nlevels = 0;
  std::vector<LevelGeom> lv;
  std::vector<CellGeom> cells;
  std::vector<XTab> xtab, ytab;
  int64_t pyr_bytes = 0;      // per frame, levels >= 1
  int64_t blur_bytes = 0;     // per frame, all levels
  int btiles_total = 0;
  int cand_total = 0;         // per frame
  int kp_total = 0;           // per frame (sum of kp_cap)
  int max_cell_w = 0, max_cell_h = 0, max_cells_per_level = 0, max_quota = 0;
};

struct DeviceLevel {  // POD copy of LevelGeom fields the kernels need
  int w, h, pitch;
  long long plane_off;
  int cell_begin, ncells, cand_off, cand_cap, quota, kp_off, kp_cap, nroots;
  int root_x0[kMaxRoots], root_x1[kMaxRoots];
  float hX, scale;
  int scaled_patch, xtab_off, ytab_off;
  long long bplane_off;
  int btile_begin, btiles_x, btiles_y;
  uint32_t m_btiles_x;  // fast_div magic of btiles_x
};

struct DeviceGeom {
  int nlevels, rows, cols, ncells_total, cand_total, kp_total, out_cap, btiles_total;
  uint32_t m_ncells, m_btiles;  // fast_div magics of ncells_total, btiles_total
  int btile_begin_all[kMaxLevels];  // lv[l].btile_begin side by side (INT_MAX beyond nlevels): the level of a blur tile from ONE scalar load
  DeviceLevel lv[kMaxLevels];
};

}  // namespace orbx

namespace orbx {
// Grow-only device scratch for the host-buffer convenience entry points (grid / search / stereo): a bump allocator that
// is rewound at the start of each call, so steady-state calls do no hipMalloc / hipFree.  Each of those entry points
// synchronises its stream before returning, so nothing is in flight when the next call rewinds.
struct DeviceArena {
  std::vector<std::pair<uint8_t*, size_t>> blocks;
  size_t off = 0, hint = 1 << 20;
  void* alloc(size_t bytes, hipError_t* err) {
    bytes = (bytes + 255) / 256 * 256;
    if (bytes == 0) bytes = 256;
    if (blocks.empty() || off + bytes > blocks.back().second) {
      size_t total = 0;
      for (auto& b : blocks) total += b.second;
      const size_t sz = std::max(std::max(bytes, hint), 2 * total);
      uint8_t* p = nullptr;
      *err = hipMalloc((void**)&p, sz);
      if (*err != hipSuccess) return nullptr;
      blocks.push_back(std::make_pair(p, sz));
      off = 0;
    }
    *err = hipSuccess;
    void* r = blocks.back().first + off;
    off += bytes;
    return r;
  }
  void rewind() {
    if (blocks.size() > 1) {   // consolidate: next call gets one block large enough for everything seen so far
      size_t total = 0;
      for (auto& b : blocks) { total += b.second; (void)hipFree(b.first); }
      blocks.clear();
      hint = std::max(hint, total);
    }
    off = 0;
  }
  void release() {
    for (auto& b : blocks) (void)hipFree(b.first);
    blocks.clear();
    off = 0;
  }
};
}  // namespace orbx

struct orbx_ctx {
  int nfeatures, nlevels, ini_th, min_th, device;
  double scale_factor;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> quota;
  int umax[16];
  int out_cap;  // nfeatures + 3*nlevels
  int fast_threads = 128;  // workgroup size of k_fast_cells
  int qt_threads = 0;      // workgroup size of k_quadtree (0: by batch size)
  bool fast_split = true;  // k_fast_cells launched per group of levels, each with its own LDS footprint
  int desc_k = 8;          // keypoints per wave of k_describe
  bool desc_k_user = false; // set by ORBX_DESC_K / "desc_k": the single-frame path then keeps it instead of choosing 1

  hipStream_t stream = nullptr;
  static constexpr int kMaxAux = 8;
  hipStream_t aux[kMaxAux] = {nullptr};     // sub-batch streams of orbx_extract_batch_device
  hipEvent_t ev_fork = nullptr, ev_join[kMaxAux] = {nullptr};
  hipEvent_t ev_blur_fork[2] = {nullptr, nullptr}, ev_blur_join[2] = {nullptr, nullptr};
  int nstreams = 1;
  hipEvent_t ev_f0_fork[2] = {nullptr, nullptr}, ev_f0_join[2] = {nullptr, nullptr};
  hipEvent_t ev_qt_fork[2] = {nullptr, nullptr}, ev_qt_join[2] = {nullptr, nullptr};
  bool fork_blur = true, fork_fast0 = false, fork_qt = true, fast_pk = true, desc_lds = true;
  int desc_fused_blur = -1;   // batches under the default blur arithmetic: the Gaussian inside the descriptor kernel (k_describe_blur), no k_blur7 launch.
                              // -1: where it pays (>= kFusedBlurPxPerKp = 800 pyramid pixels per keypoint slot), 0: never, 1: wherever the arithmetic allows
  bool realign = true;          // batch calls: frames whose rows are not dword-aligned are copied into an aligned buffer first
  uint8_t* d_realign = nullptr;
  size_t realign_bytes = 0;
  int replay_alternate = -1;    // lane schedule of a replay engine whose lane 0 this context is: -1 default (alternate for >= 2 lanes), 0 split, 1 alternate
  int fast_passes = 2;          // batch FAST: 2 = the reference's two-threshold cell loop literally (iniTh, then minTh where the cell stayed empty); 1 = one pass at minTh
  bool fast_stage_dma = true;   // FAST: the cell's tile by LDS-DMA loads (aligned sources)
  int qt_points = 2048;       // LDS-resident candidate capacity per (frame, level) of k_quadtree's big levels ("qt_points" / ORBX_QT_POINTS)
  int chain_threads = 1024;   // workgroup size of k_resize_chain (ORBX_CHAIN_THREADS)
  int chain_first = 7;           // levels in the first chain launch of a single frame (2 .. 7)
  int chain_long_tile = 16;      // tile of the last level of a long chain
  // which cv::GaussianBlur the blur equals (orbx_set_option "gauss_kernel" / "gauss_round"; include/orbx.h):
  // kernel 0 = {18,34,48,56,...} (error-diffused, sum 256), 1 = {18,34,49,55,...} (each coefficient rounded, sum 257);
  // round 0 = (acc + 2^15) >> 16, 1 = half to even (last w mod 4 columns half up), 2 = floor; all saturated to 255
  // gauss_tail V: the last (w mod V) columns of a row round half up whatever gauss_round says (the scalar tail of a SIMD column pass)
  // atan_fma: cv::fastAtan2's polynomial with the contractions a compiler makes under -mfma (OpenCV's AVX2 dispatch), 0 = separate mul / add
  // brief_fma: the pattern rotation of src/ORBextractor.cc:118-120 as a -march=native build of the reference contracts it (fma(x, b, y*a))
  int gauss_kernel = 0, gauss_round = 0, gauss_tail = 0, atan_fma = 0, brief_fma = 0;
  bool qt_fused = true;          // quadtree: the first passes fused into one sweep (level_base bit 9 switches it off)
  bool describe_direct = true;   // single frame, trivial lapping area: no assembly pass, k_describe reads the quadtree's per-level output
  bool chain_long = true;        // single-frame pyramid: levels 1-2 in one launch, then up to five small levels per launch
  bool chain_batch = false;      // batches too build the pyramid with the chain launches (k_resize_chain) instead of one launch per level
  int qt_big_levels = 2;         // levels of a batch launched with the large quadtree workgroup configuration
  int qt_threads_small = 128;    // threads of the small-level quadtree launch of a batch (0 = as the big levels): 127.9 -> 115.4 us per 256 frames
  bool qt_level_major = true, qt_one_launch = false;   // batch quadtree launch order experiments ("qt_level_major", "qt_one_launch")
  bool small_fused = true;    // small batches (single-frame graph): FAST + blur in one launch, assemble as the quadtree's tail (ORBX_SMALL_FUSED=0 / "small_fused")
  int32_t* d_qt_fin = nullptr;   // per-frame finished-level counters of k_quadtree_assemble (zero between launches)

  std::string err;

  // geometry + device buffers for the current (rows, cols, batch capacity)
  orbx::Geometry geo;
  orbx::DeviceGeom* d_geo = nullptr;
  orbx::CellGeom* d_cells = nullptr;
  orbx::XTab* d_xtab = nullptr;
  orbx::XTab* d_ytab = nullptr;
  int batch_cap = 0;
  int blur_cap = 0;        // frames d_blur holds: the blurred planes exist only for extractions that launch k_blur7 (ensure_blur)
  bool last_fused_blur = false;   // the last extraction blurred inside the descriptor kernel: there are no blurred planes to read back
  uint8_t* d_pyr = nullptr;        // [batch][pyr_bytes]
  uint8_t* d_blur = nullptr;       // [batch][blur_bytes] 7x7 Gaussian of every level
  uint32_t* d_cand = nullptr;      // [batch][cand_total]
  int32_t* d_cell_cnt = nullptr;   // [batch][ncells]
  uint32_t* d_pts = nullptr;       // [batch][2][cand_total]  quadtree ping-pong
  uint8_t* d_qt_nodes = nullptr;   // node arrays of the levels whose quota does not fit the LDS: [level slot][batch][qt_node_stride]
  size_t qt_node_stride = 0; int qt_node_slot[orbx::kMaxLevels] = {0}; int qt_node_slots = 0;
  unsigned long long* d_asm_scan = nullptr;   // k_assemble's scan array when capacity * 8 B does not fit the LDS: [batch][out_cap]
  uint32_t* d_lvl_kp = nullptr;    // [batch][kp_total]
  int32_t* d_lvl_n = nullptr;      // [batch][nlevels]
  uint2* d_kp_list = nullptr;      // [batch][out_cap] {packed point, level | output slot << 8}, level-major order
  // single-frame staging (orbx_extract)
  uint8_t* d_stage_img = nullptr; size_t stage_img_bytes = 0;
  orbx::XTab* d_ingest_tab = nullptr; int ingest_key[4] = {0, 0, 0, 0}; int ingest_ytab_off = 0, ingest_lds_pitch = 0, ingest_lds_rows = 0;   // resize ingestion (orbx_extract_resized)
  uint8_t* d_color = nullptr; size_t color_bytes = 0;   // interleaved colour frames of orbx_extract_color, before conversion
  uint8_t* d_stage_out = nullptr;   // [keypoints | descriptors | counts] of the host-buffer entry points
  uint8_t* h_stage_out = nullptr;   // pinned host mirror of d_stage_out (one D2H copy per call)
  int stage_frames = 0;
  orbx::DeviceArena arena;   // scratch of the grid / search / stereo entry points
  bool window_direct = true;   // window passes write their results into mapped pinned memory and the host polls (ORBX_WINDOW_DIRECT=0 / "window_direct": copies + sync)
  uint8_t* h_tgt = nullptr; size_t h_tgt_bytes = 0; hipEvent_t ev_tgt = nullptr;   // pinned staging of orbx_target uploads + "staging free again"
  std::vector<int32_t> view_row_ptr; int view_pool_cap = 16384;   // orbx_target_search_view: scratch and the candidate-pool capacity learnt from earlier calls
  int32_t* d_win_ctr = nullptr; bool win_ctr_dirty = true;   // the two self-resetting counters of resident-target window passes
  uint8_t* h_view[2] = {nullptr, nullptr}; size_t h_view_bytes[2] = {0, 0}; int view_par = 0;   // the blobs of view calls, used alternately
  // orbx_target_search_view_begin / _end: a view call split into issue and wait (a caller with two halves of work overlaps its own host
  // passes with the device: ref_adapter/ORBmatcher.cc).  One pending call per blob.
  struct ViewCall {
    bool pending = false, sync_fallback = false;
    const orbx_target* target = nullptr; const uint8_t* kp_skip = nullptr;
    const float *qx = nullptr, *qy = nullptr, *qr = nullptr, *q_xr = nullptr; const int32_t *qlo = nullptr, *qhi = nullptr; const uint8_t* q_desc = nullptr;
    int nq = 0, pool_cap = 0; size_t in_size = 0, p_hdr = 0, p_q = 0, p_pool = 0;
  } view_call[2];
  uint8_t* h_call = nullptr; size_t h_call_bytes = 0;   // pinned [inputs | outputs] blob of the window / nn entry points (orbx_window.hip)
  int win_guess = 0;            // candidates of the last window call: how much of the pool the first read-back copy takes
  // single-frame operator() path as a replayed hipGraph (H2D, the 13 launches, D2H): one graph launch per frame instead
  // of ~16 API calls; re-captured when the shape / lapping area / buffers change, disabled on any capture failure
  bool graph_timing = false; hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr; double last_graph_us = -1.0;   // "graph_timing": events around the replayed graph
  bool window_timing = false; hipEvent_t ev_w0 = nullptr, ev_w1 = nullptr; double last_window_us = -1.0; // "window_timing": events around a resident-target window pass
  bool use_graph = true;
  hipGraphExec_t graph_exec = nullptr;
  int graph_key[6] = {0, 0, 0, 0, 0, 0};
  int buf_epoch = 0;            // bumped whenever a buffer a captured graph points to is reallocated
  uint8_t* h_in = nullptr; size_t h_in_bytes = 0;   // pinned copy of the caller's frame
  unsigned long long* d_knn_ws = nullptr; size_t knn_ws_bytes = 0;  // per-segment partial top-2 of orbx_knn2_allpairs*
  // host mirror of frame 0's pyramid levels >= 1 (pinned), refreshed by orbx_extract when keep_host_pyr is set
  bool keep_host_pyr = false;
  uint8_t* h_pyr = nullptr; size_t h_pyr_bytes = 0; bool h_pyr_valid = false;
  // last extraction (for orbx_pyramid_level / debug dumps)
  const uint8_t* last_imgs = nullptr; size_t last_row_stride = 0, last_frame_stride = 0; int last_nframes = 0;
  // the last host-buffer extraction: its sequence number, its keypoint count (frame 0; -1 while one is running or after a failed one), and
  // whether one is running right now (orbx_voc_destroy waits for it).  All under orbx_extractor.hip's g_pub_mu
  unsigned long long extract_seq = 0;
  int last_n0 = -1; bool in_extract = false;
  // vocabulary attached to this extractor context (orbx_bow_transform_published attaches the first vocabulary that asks for the words of
  // one of its extractions): the single-frame graph then ends with the tree descent of the frame's descriptors and leaves the
  // {word, node, weight} records in the pinned result block — Frame::ComputeBoW costs no device round trip of its own.  bow_voc /
  // bow_levelsup are written under g_pub_mu by any thread; the extracting thread takes its snapshot (bow_active*) at the start of a call
  orbx_voc* bow_voc = nullptr; int bow_levelsup = 0;
  orbx_voc* bow_active = nullptr; int bow_active_levelsup = 0;
  unsigned long long bow_gen = 0, bow_active_gen = 0;   // bumped when the attached vocabulary is destroyed: a later one at the same address is another tree
  unsigned long long bow_seq = ~0ull;   // extract_seq of the extraction whose records the pinned block holds
  size_t bow_off = 0;                   // offset of the records in h_stage_out
  bool bow_in_graph = false;            // the captured single-frame graph ends with the descent
  // the host buffer that holds (a copy of) the rows of the last extraction (orbx_publish_descriptors), until the next extraction begins
  const void* rows_host = nullptr; int rows_n = 0; uint64_t rows_digest_v = 0;
  hipStream_t last_ext_stream = nullptr;   // caller's stream of the last orbx_extract_batch_device (its work may still use our buffers)
  // profiling
  bool profiling = false;
  double prof_ms[ORBX_NUM_KERNELS] = {0};
  int64_t prof_n[ORBX_NUM_KERNELS] = {0};
  std::vector<hipEvent_t> ev_pool;
};

namespace orbx {
int set_err(orbx_ctx* ctx, int code, const std::string& msg);

// No entry point touches the legacy (null) stream: a null-stream copy or a device-wide sync issued by one thread while
// ANOTHER context captures its single-frame graph is rejected by the runtime and poisons that capture (two extractors
// run from two threads in stereo, src/Frame.cc:122-125).  Synchronous copies go through the context's own stream.
inline hipError_t copy_sync(orbx_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
  return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}
inline hipError_t copy2d_sync(orbx_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                              hipMemcpyKind kind) {
  hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, ctx->stream);
  return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}
// everything this context may have in flight: its stream, its aux streams, and the caller's stream of the last
// device-resident batch
inline hipError_t sync_ctx(orbx_ctx* ctx) {
  hipError_t e = hipStreamSynchronize(ctx->stream);
  for (int i = 0; i < orbx_ctx::kMaxAux && e == hipSuccess; i++)
    if (ctx->aux[i]) e = hipStreamSynchronize(ctx->aux[i]);
  if (e == hipSuccess && ctx->last_ext_stream) e = hipStreamSynchronize(ctx->last_ext_stream);
  return e;
}
// hipFuncAttributeMaxDynamicSharedMemorySize is process-wide state of a kernel: contexts on different threads (two stereo
// extractors, the replay lanes, the per-thread matcher contexts) must never lower it under a launch that needs more, so the
// value is only ever raised, under a mutex (orbx_extractor.hip).
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);
// orbx_window.hip: pinned staging (grow-only) and the fused window pass behind orbx_window_search* / orbx_window_nearest
hipError_t host_stage(orbx_ctx* ctx, size_t bytes, uint8_t** p);
// orbx_extractor.hip: orbx_destroy — the context leaves the publisher / attachment lists
void unpublish_context(orbx_ctx* ctx);
// orbx_matcher.hip: the tree descent of the single-frame graph — descriptors and the frame's keypoint count read on the device, records
// {word, node, weight} (16 bytes each) written to rec_out; nullptr stream error codes as hipError_t.  voc_device: the GPU the tree lives on
hipError_t launch_bow_records(const orbx_voc* v, const uint8_t* d_desc, const int32_t* d_counts, int cap, int levelsup, void* rec_out, hipStream_t st);
int voc_device(const orbx_voc* v);
// orbx_extractor.hip: a vocabulary is going away — no context may keep it attached
void voc_detach_all(const orbx_voc* v);
// 64-bit digest of n descriptor rows: the first, the middle and the last row whole, one 8-byte word of every other row (1.5 us for 1000 rows)
inline uint64_t rows_digest(const uint8_t* rows, int n) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)(uint32_t)n;
  auto mix = [&h](uint64_t v) { h = (h ^ v) * 0xff51afd7ed558ccdull; h ^= h >> 32; };
  auto word = [rows](size_t off) { uint64_t v; std::memcpy(&v, rows + off, 8); return v; };
  for (int i = 0; i < n; i++) mix(word((size_t)i * 32 + 8 * (size_t)(i & 3)));
  if (n > 0) for (int r : {0, n / 2, n - 1}) for (int k = 0; k < 4; k++) mix(word((size_t)r * 32 + 8 * (size_t)k));
  return h;
}
}  // namespace orbx
// A Frame's / KeyFrame's keypoints, descriptors and grid resident in HBM (orbx_target_create, orbx_window.hip)
struct orbx_target {
  orbx_ctx* ctx = nullptr;
  uint8_t* dev = nullptr;
  size_t cap = 0, bytes = 0, o_kps = 0, o_desc = 0, o_ur = 0, o_sig = 0, o_cs = 0, o_ci = 0;
  int n = 0, ngrid = 0, nlevels = 0;
  bool has_ur = false, has_sig = false;
  bool valid = false;   // false after a failed (re)fill: searches return ORBX_E_INVALID instead of treating it as an empty target
  float min_x = 0, min_y = 0, inv_w = 0, inv_h = 0;
};
namespace orbx {
// offsets of the pieces of one packed blob (256-byte aligned pieces)
struct BlobLayout {
  size_t size = 0;
  size_t add(size_t bytes) { const size_t o = size; size = (size + bytes + 255) & ~(size_t)255; return o; }
};
int window_call(orbx_ctx* ctx, const char* who, const orbx_keypoint* kps, const uint8_t* desc, int n, const orbx_grid* grid,
                const uint8_t* kp_skip, const float* kp_uright, const float* inv_sigma2, int nlevels, const float* qx, const float* qy,
                const float* qr, const int32_t* qlo, const int32_t* qhi, const float* qaux, const uint8_t* q_desc, int nq, bool lists,
                int32_t* row_ptr, int32_t* cand, int32_t* dist, int cand_cap, int32_t* best_idx, int32_t* best_dist, int32_t* second_idx,
                int32_t* second_dist);
#define ORBX_HIP(ctx, expr)                                                                          \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return orbx::set_err((ctx), ORBX_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
}  // namespace orbx
