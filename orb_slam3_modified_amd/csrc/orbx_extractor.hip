// orbx extractor: host side of the C ABI (include/orbx.h) — scale tables, level/cell geometry, device
// buffers sized for batch replay, kernel launch sequence.  No CPU compute fallback: every entry point that
// produces features needs a HIP device and reports ORBX_E_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <condition_variable>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <new>

#include "orbx_internal.h"
#include "orbx_kernels.hip"

namespace orbx {

// Workgroup size of k_fast_cells: 128 by default (measured best); ORBX_FAST_THREADS=64|128|256 in the environment overrides it
// (tuning knob, read once per context).
constexpr int kQtLdsPoints = 2048;  // LDS-resident candidate capacity per (frame, level) of k_quadtree (big levels)
static_assert(kQtLdsPoints <= 4096, "the chunk prefix of the quadtree full pass is one wave wide");
constexpr int kQtBigLevels = 2;     // levels launched with the large quadtree workgroup configuration
constexpr int kSmallBatchFrames = 512;   // a small batch is nframes * nlevels <= 512

static int fast_threads_from_env() {
  const char* e = getenv("ORBX_FAST_THREADS");
  const int v = e ? atoi(e) : 128;
  return (v == 64 || v == 128 || v == 256) ? v : 128;
}

hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  // the attribute belongs to the function ON THE CURRENT DEVICE: one process may drive several GPUs (contexts carry their device)
  static std::map<std::pair<int, const void*>, int> raised;
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  int& cur = raised[std::make_pair(dev, kernel)];
  if (bytes <= cur) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) cur = bytes;
  else (void)hipGetLastError();   // do not leave the sticky error for an unrelated later call
  return e;
}

int set_err(orbx_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

static inline int cv_round(float v) { return (int)lrintf(v); }
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }
// fast_div magic: ceil(2^32 / d), 0 for d == 1; n / d == umulhi(n, M) while n * d < 2^32
static inline uint32_t div_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); }
static inline bool div_ok(uint64_t nmax, uint64_t d) { return nmax * d < (1ull << 32); }
static inline unsigned xcd_grid(int nitems) { return (unsigned)(8 * ((nitems + 7) / 8)); }  // see xcd_logical_block

// resize tables (cv::resize INTER_LINEAR 8u): per destination index the two source
// indices and the two 11-bit weights.  The clamped tail (`S[sx]*2048`) is encoded as weights (2048, 0).
static void build_axis_table(int ssize, int dsize, bool is_x, std::vector<XTab>& out) {
  const double scale = 1.0 / ((double)dsize / ssize);
  int dmax = dsize;
  for (int d = 0; d < dsize; d++) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f -= s;
    XTab t;
    if (is_x) {
      if (s < 0) { f = 0; s = 0; }
      if (s + 1 >= ssize) {
        dmax = std::min(dmax, d);
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
      }
      int a0 = std::min(std::max(cv_round((1.f - f) * 2048.f), -32768), 32767);
      int a1 = std::min(std::max(cv_round(f * 2048.f), -32768), 32767);
      if (d >= dmax) { t.s0 = (uint16_t)s; t.s1 = (uint16_t)s; t.a0 = 2048; t.a1 = 0; }
      else { t.s0 = (uint16_t)s; t.s1 = (uint16_t)(s + 1); t.a0 = (int16_t)a0; t.a1 = (int16_t)a1; }
    } else {
      // rows: the index is clamped, the weights are not (OpenCV keeps fy)
      int a0 = std::min(std::max(cv_round((1.f - f) * 2048.f), -32768), 32767);
      int a1 = std::min(std::max(cv_round(f * 2048.f), -32768), 32767);
      auto clip = [&](int v) { return v < 0 ? 0 : (v >= ssize ? ssize - 1 : v); };
      t.s0 = (uint16_t)clip(s); t.s1 = (uint16_t)clip(s + 1); t.a0 = (int16_t)a0; t.a1 = (int16_t)a1;
    }
    out.push_back(t);
  }
}

// Level / cell geometry: src/ORBextractor.cc:1174-1175 (level sizes), :789-822 (cells), :559-579 (roots).
static int build_geometry(orbx_ctx* ctx, int rows, int cols, Geometry& geo) {
  geo = Geometry();
  geo.rows = rows; geo.cols = cols; geo.nlevels = ctx->nlevels;
  if (rows > kMaxDim || cols > kMaxDim) return set_err(ctx, ORBX_E_INVALID, "image larger than 4095 px per side");
  geo.lv.resize(ctx->nlevels);
  int cand_off = 0, kp_off = 0, btile = 0;
  int64_t plane_off = 0, bplane_off = 0;
  for (int l = 0; l < ctx->nlevels; l++) {
    LevelGeom& L = geo.lv[l];
    std::memset(&L, 0, sizeof(L));
    const float s = ctx->inv_scale[l];
    L.w = cv_round((float)cols * s);
    L.h = cv_round((float)rows * s);
    L.pitch = round_up(L.w, 64);
    L.plane_off = plane_off;
    if (l > 0) plane_off += (int64_t)L.pitch * L.h;
    L.bplane_off = bplane_off;
    bplane_off += (int64_t)L.pitch * L.h;
    L.btile_begin = btile; L.btiles_x = (L.w + kBT_W - 1) / kBT_W; L.btiles_y = (L.h + kBT_H - 1) / kBT_H;
    btile += L.btiles_x * L.btiles_y;
    const int minB = kBorder, maxBX = L.w - kBorder, maxBY = L.h - kBorder;
    const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
    const float W = 35;
    L.ncols = (int)(width / W);
    L.nrows = (int)(height / W);
    if (L.ncols < 1 || L.nrows < 1)
      return set_err(ctx, ORBX_E_INVALID, "image too small for this number of pyramid levels (level < 67 px)");
    L.wcell = (int)std::ceil(width / L.ncols);
    L.hcell = (int)std::ceil(height / L.nrows);
    L.cell_begin = (int)geo.cells.size();
    L.cand_off = cand_off;
    for (int i = 0; i < L.nrows; i++) {
      const float iniY = (float)(minB + i * L.hcell);
      float maxY = iniY + L.hcell + 6;
      if (iniY >= maxBY - 3) continue;
      if (maxY > maxBY) maxY = (float)maxBY;
      for (int j = 0; j < L.ncols; j++) {
        const float iniX = (float)(minB + j * L.wcell);
        float maxX = iniX + L.wcell + 6;
        if (iniX >= maxBX - 6) continue;
        if (maxX > maxBX) maxX = (float)maxBX;
        CellGeom c{};
        c.level = (int16_t)l;
        c.x0 = (int16_t)iniX; c.y0 = (int16_t)iniY;
        c.cw = (int16_t)((int)maxX - (int)iniX); c.ch = (int16_t)((int)maxY - (int)iniY);
        c.relx = (int16_t)(j * L.wcell); c.rely = (int16_t)(i * L.hcell);
        const int dw = c.cw - 6, dh = c.ch - 6;
        if (dw <= 0 || dh <= 0) continue;  // FAST has no interior pixel: the reference gets no keypoint here
        c.pitch = L.pitch; c.plane_off = L.plane_off;
        c.slot_off = cand_off;
        c.slot_cap = ((dw + 1) / 2) * ((dh + 1) / 2);  // strict 3x3 NMS: survivors are never 8-adjacent
        cand_off += c.slot_cap;
        geo.cells.push_back(c);
        geo.max_cell_w = std::max(geo.max_cell_w, (int)c.cw);
        geo.max_cell_h = std::max(geo.max_cell_h, (int)c.ch);
      }
    }
    L.ncells = (int)geo.cells.size() - L.cell_begin;
    L.cand_cap = cand_off - L.cand_off;
    geo.max_cells_per_level = std::max(geo.max_cells_per_level, L.ncells);
    L.quota = ctx->quota[l];
    geo.max_quota = std::max(geo.max_quota, L.quota);
    L.nroots = (int)std::round((float)(maxBX - minB) / (maxBY - minB));
    if (L.nroots < 1 || L.nroots > kMaxRoots)
      return set_err(ctx, ORBX_E_INVALID, "aspect ratio outside the supported 0.5 .. 8.5 range");
    L.hX = (float)(maxBX - minB) / L.nroots;
    for (int i = 0; i < L.nroots; i++) {
      L.root_x0[i] = (int)(L.hX * (float)i);
      L.root_x1[i] = (int)(L.hX * (float)(i + 1));
    }
    L.kp_off = kp_off;
    L.kp_cap = std::max(L.quota + 3, 4 * kMaxRoots);
    kp_off += L.kp_cap;
    L.scale = ctx->scale[l];
    L.scaled_patch = (int)(kPatchSize * ctx->scale[l]);
    if (l > 0) {
      while (geo.xtab.size() % 4) geo.xtab.push_back(XTab{0, 0, 0, 0});  // 32-byte aligned level tables (uint4 loads)
      L.xtab_off = (int)geo.xtab.size();
      build_axis_table(geo.lv[l - 1].w, L.w, true, geo.xtab);
      L.ytab_off = (int)geo.ytab.size();
      build_axis_table(geo.lv[l - 1].h, L.h, false, geo.ytab);
      // LDS tile of k_resize: the largest source rectangle any 64x16 output tile of this level needs
      int mw = 1, mh = 1;
      for (int x0 = 0; x0 < L.w; x0 += kRT_W) {
        const XTab a = geo.xtab[L.xtab_off + x0], b = geo.xtab[L.xtab_off + std::min(x0 + kRT_W, L.w) - 1];
        mw = std::max(mw, std::max((int)b.s0, (int)b.s1) - ((int)a.s0 & ~3) + 1);
      }
      for (int y0 = 0; y0 < L.h; y0 += kRT_H) {
        const XTab a = geo.ytab[L.ytab_off + y0], b = geo.ytab[L.ytab_off + std::min(y0 + kRT_H, L.h) - 1];
        mh = std::max(mh, std::max((int)b.s0, (int)b.s1) - (int)a.s0 + 1);
      }
      L.rs_lds_pitch = round_up(mw, 4);
      L.rs_lds_rows = mh;
      if ((size_t)L.rs_lds_pitch * L.rs_lds_rows > 60 * 1024)
        return set_err(ctx, ORBX_E_CAPACITY, "scale factor too large for the resize kernel's LDS tile");
    }
  }
  geo.pyr_bytes = plane_off;
  geo.blur_bytes = bplane_off;
  geo.btiles_total = btile;
  geo.cand_total = cand_off;
  geo.kp_total = kp_off;
  return ORBX_OK;
}

static void free_buffers(orbx_ctx* ctx) {
  auto fr = [](auto*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } };
  fr(ctx->d_geo); fr(ctx->d_cells); fr(ctx->d_xtab); fr(ctx->d_ytab);
  fr(ctx->d_pyr); fr(ctx->d_blur); fr(ctx->d_cand); fr(ctx->d_cell_cnt); fr(ctx->d_pts); fr(ctx->d_lvl_kp); fr(ctx->d_lvl_n); fr(ctx->d_kp_list); fr(ctx->d_qt_nodes); fr(ctx->d_asm_scan);
  ctx->batch_cap = 0; ctx->blur_cap = 0;
}

// LDS bytes of one quadtree workgroup over levels with at most `mq` quota and `mc` cells (see k_quadtree's carve-up)
constexpr size_t kLdsMax = 160 * 1024;   // one workgroup may use the whole LDS of a CU, not more
constexpr int kQtMaxNodes = 65532;        // 16-bit positions in the exact sort
// mq: largest quota, mc: most cells, mp: most candidate slots of the levels served (a node keeps >= 1 point, so the list is
// never longer than the candidates either)
static void qt_caps(int mq, int mc, int mp, int& node_cap, int& scan_cap) {
  node_cap = std::min(round_up(std::min(mq, mp) + 4 * kMaxRoots + 8, 4), kQtMaxNodes);
  scan_cap = round_up(std::max(node_cap, mc) + 8, 4);
}
static size_t qt_node_bytes(int node_cap, int scan_cap) {
  return (size_t)node_cap * (2 * sizeof(QNode) + 2 * 8 + sizeof(int4) + 4) + (size_t)scan_cap * 8;
}
// LDS-resident points of one quadtree workgroup: two point buffers, two node-index buffers (16 bit) and the chunk tables of
// the thread-per-point full passes (counts, prefix and four class ballots per 64 positions, one spare entry); pts_cap <= 4096
static size_t qt_point_bytes(int pts_cap) {
  return pts_cap ? (size_t)pts_cap * (2 * sizeof(uint32_t) + 2 * sizeof(uint16_t)) + (size_t)(pts_cap / 64 + 1) * 6 * 8 : 0;
}

static int ensure_buffers(orbx_ctx* ctx, int rows, int cols, int nframes) {
  const bool same_shape = ctx->geo.rows == rows && ctx->geo.cols == cols && ctx->d_geo;
  if (same_shape && nframes <= ctx->batch_cap) return ORBX_OK;
  ORBX_HIP(ctx, sync_ctx(ctx));
  Geometry geo;
  int rc = build_geometry(ctx, rows, cols, geo);
  if (rc != ORBX_OK) return rc;
  free_buffers(ctx);
  ctx->buf_epoch++;
  ctx->geo = geo;
  DeviceGeom dg;
  std::memset(&dg, 0, sizeof(dg));
  dg.nlevels = geo.nlevels; dg.rows = rows; dg.cols = cols;
  dg.ncells_total = (int)geo.cells.size(); dg.cand_total = geo.cand_total; dg.kp_total = geo.kp_total;
  dg.out_cap = ctx->out_cap; dg.btiles_total = geo.btiles_total;
  dg.m_ncells = div_magic((uint32_t)geo.cells.size()); dg.m_btiles = div_magic((uint32_t)geo.btiles_total);
  for (int l = 0; l < kMaxLevels; l++) dg.btile_begin_all[l] = l < geo.nlevels ? geo.lv[l].btile_begin : INT_MAX;
  for (int l = 0; l < geo.nlevels; l++) {
    const LevelGeom& L = geo.lv[l];
    DeviceLevel& D = dg.lv[l];
    D.w = L.w; D.h = L.h; D.pitch = L.pitch; D.plane_off = L.plane_off;
    D.cell_begin = L.cell_begin; D.ncells = L.ncells; D.cand_off = L.cand_off; D.cand_cap = L.cand_cap;
    D.quota = L.quota; D.kp_off = L.kp_off; D.kp_cap = L.kp_cap; D.nroots = L.nroots;
    for (int i = 0; i < kMaxRoots; i++) { D.root_x0[i] = L.root_x0[i]; D.root_x1[i] = L.root_x1[i]; }
    D.hX = L.hX; D.scale = L.scale; D.scaled_patch = L.scaled_patch; D.xtab_off = L.xtab_off; D.ytab_off = L.ytab_off;
    D.bplane_off = L.bplane_off; D.btile_begin = L.btile_begin; D.btiles_x = L.btiles_x; D.btiles_y = L.btiles_y; D.m_btiles_x = div_magic(L.btiles_x);
  }
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_geo, sizeof(DeviceGeom)));
  ORBX_HIP(ctx, copy_sync(ctx, ctx->d_geo, &dg, sizeof(dg), hipMemcpyHostToDevice));
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_cells, sizeof(CellGeom) * geo.cells.size()));
  ORBX_HIP(ctx, copy_sync(ctx, ctx->d_cells, geo.cells.data(), sizeof(CellGeom) * geo.cells.size(), hipMemcpyHostToDevice));
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_xtab, sizeof(XTab) * std::max<size_t>(geo.xtab.size(), 1)));
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_ytab, sizeof(XTab) * std::max<size_t>(geo.ytab.size(), 1)));
  if (!geo.xtab.empty()) {
    ORBX_HIP(ctx, copy_sync(ctx, ctx->d_xtab, geo.xtab.data(), sizeof(XTab) * geo.xtab.size(), hipMemcpyHostToDevice));
    ORBX_HIP(ctx, copy_sync(ctx, ctx->d_ytab, geo.ytab.data(), sizeof(XTab) * geo.ytab.size(), hipMemcpyHostToDevice));
  }
  const size_t B = (size_t)nframes;
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_pyr, std::max<size_t>(B * (size_t)geo.pyr_bytes, 64)));
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_cand, B * geo.cand_total * sizeof(uint32_t)));
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_cell_cnt, B * geo.cells.size() * sizeof(int32_t)));
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_pts, B * 2 * geo.cand_total * sizeof(uint32_t)));
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_lvl_kp, B * geo.kp_total * sizeof(uint32_t)));
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_lvl_n, B * geo.nlevels * sizeof(int32_t)));
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_kp_list, B * ctx->out_cap * sizeof(uint2)));
  // levels whose node arrays do not fit the LDS even with the points in HBM get an HBM slice per (level, frame)
  ctx->qt_node_stride = 0; ctx->qt_node_slots = 0;
  for (int l = 0; l < geo.nlevels; l++) {
    int nc, sc;
    qt_caps(geo.lv[l].quota, geo.lv[l].ncells, geo.lv[l].cand_cap, nc, sc);
    ctx->qt_node_slot[l] = -1;
    if (qt_node_bytes(nc, sc) > kLdsMax) {
      ctx->qt_node_slot[l] = ctx->qt_node_slots++;
      ctx->qt_node_stride = std::max(ctx->qt_node_stride, (size_t)round_up((int)qt_node_bytes(nc, sc), 256));
    }
  }
  if (ctx->qt_node_slots) ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_qt_nodes, B * ctx->qt_node_slots * ctx->qt_node_stride));
  if ((size_t)ctx->out_cap * 8 + 64 > kLdsMax) ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_asm_scan, B * (size_t)ctx->out_cap * 8));
  ctx->batch_cap = nframes;
  return ORBX_OK;
}

// Does an extraction of `nframes` frames take the Gaussian inside the descriptor kernel (k_describe_blur) instead of launching k_blur7?  It blurs a
// window per keypoint (only the positions the rotated pattern can read: fb_items.inc) where k_blur7 blurs every pixel once, so its cost goes with
// the keypoints, not with the pixels.  Measured per 256 frames: 640 x 480 with 1000 features (975 pyramid pixels per keypoint slot) 0.264 ms
// against 0.171 + 0.168, the step +6 %; 1024 x 1024 with 2000 features (1690) the step +18 %.  The other blur arithmetics (the kernel's
// general path), the single-frame graph and desc_lds = 0 keep the separate kernel.
static bool small_fused_launch(const orbx_ctx* ctx, int nframes) {
  return nframes * ctx->geo.nlevels <= 512 && nframes <= 4 && ctx->small_fused && !ctx->profiling;
}
static bool use_fused_blur(const orbx_ctx* ctx, int nframes) {
  const Geometry& geo = ctx->geo;
  long long pyr_px = 0;
  for (int l = 0; l < geo.nlevels; l++) pyr_px += (long long)geo.lv[l].w * geo.lv[l].h;
  constexpr long long kFusedBlurPxPerKp = 800;
  const bool pays = ctx->desc_fused_blur > 0 || (ctx->desc_fused_blur < 0 && pyr_px >= kFusedBlurPxPerKp * ctx->out_cap);
  const bool general_blur = ctx->gauss_kernel != 0 || ctx->gauss_round != 0;
  bool same_umax = true;   // the kernel's moment weights are a generated table (fb_items.inc) for exactly these circle half-widths
  for (int i = 0; i < 16; i++) same_umax = same_umax && ctx->umax[i] == kFB_UMAX[i];
  return pays && !small_fused_launch(ctx, nframes) && !general_blur && ctx->desc_lds && same_umax;
}
// The blurred planes ([batch][blur_bytes]) exist only for extractions that launch k_blur7; allocated (for the whole batch capacity) by the entry
// points before they queue anything — never inside launch_pipeline, which may run under stream capture.
static int ensure_blur(orbx_ctx* ctx, int nframes) {
  if (use_fused_blur(ctx, nframes) || ctx->blur_cap >= ctx->batch_cap) return ORBX_OK;
  ORBX_HIP(ctx, sync_ctx(ctx));
  if (ctx->d_blur) { (void)hipFree(ctx->d_blur); ctx->d_blur = nullptr; }
  ctx->blur_cap = 0;
  ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_blur, (size_t)ctx->batch_cap * (size_t)ctx->geo.blur_bytes));
  ctx->blur_cap = ctx->batch_cap;
  return ORBX_OK;
}

struct ProfScope {
  orbx_ctx* ctx; int slot; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr;
  ProfScope(orbx_ctx* c, int s, hipStream_t stream) : ctx(c), slot(s), st(stream) {
    if (ctx->profiling) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
  }
  ~ProfScope() {
    if (ctx->profiling) {
      (void)hipEventRecord(e1, st);
      ctx->ev_pool.push_back(e0); ctx->ev_pool.push_back(e1);
      ctx->prof_n[slot] += 1;
      // slot is recovered from the pool position when the times are read
      ctx->ev_pool.push_back((hipEvent_t)(intptr_t)(slot + 1));
    }
  }
};

static void gaussian_kernel7(int k[7], int variant) {
  // cv::getGaussianKernel(7, 2) in 8.8 fixed point.  variant 0: rounding error diffused from the outside in, centre =
  // 256 - 2 * sum(others) — {18,34,48,56,...}, OpenCV >= 4.5.1; variant 1: every coefficient rounded on its own — {18,34,49,55,...}, sum
  // 257, the kernel of OpenCV 3.x .. 4.5.0
  double v[7], sum = 0;
  for (int i = 0; i < 7; i++) { const double x = i - 3; v[i] = std::exp(-0.5 * x * x / 4.0); sum += v[i]; }
  if (variant == 1) { for (int i = 0; i < 7; i++) k[i] = cv_round(v[i] / sum * 256.0); return; }
  double err = 0; int s = 0;
  for (int i = 0; i < 3; i++) {
    const double adj = v[i] / sum * 256.0 + err;
    const int q = cv_round(adj);
    err = adj - q; k[i] = k[6 - i] = q; s += q;
  }
  k[3] = 256 - 2 * s;
}

// One pass of the hot path over frames [f0, f0 + nframes) of the batch, on stream `st`.  Every buffer is indexed by
// frame, so a sub-batch is the same launch sequence on offset base pointers.
// mirror (optional): device-visible addresses of the caller's pinned result block — the single-frame graph lets the last kernels write
// keypoints, descriptors and counts there as well, instead of a download node; *mirrored tells whether that happened
struct ResultMirror { orbx_keypoint* kps = nullptr; uint8_t* desc = nullptr; int32_t* counts = nullptr; };
static int launch_pipeline(orbx_ctx* ctx, const uint8_t* d_imgs, int f0, int nframes, int rows, int cols, size_t row_stride,
                           size_t frame_stride, int lap0, int lap1, orbx_keypoint* d_kps, uint8_t* d_desc,
                           int32_t* d_counts, hipStream_t st, const ResultMirror* mirror = nullptr, bool* mirrored = nullptr) {
  if (mirrored) *mirrored = false;
  const Geometry& geo = ctx->geo;
  d_imgs += (size_t)f0 * frame_stride;
  d_kps += (size_t)f0 * ctx->out_cap;
  d_desc += (size_t)f0 * ctx->out_cap * 32;
  d_counts += (size_t)f0 * 2;
  uint8_t* const b_pyr = ctx->d_pyr + (size_t)f0 * geo.pyr_bytes;
  uint8_t* const b_blur = ctx->d_blur ? ctx->d_blur + (size_t)f0 * geo.blur_bytes : nullptr;   // only when k_blur7 runs (ensure_blur)
  uint32_t* const b_cand = ctx->d_cand + (size_t)f0 * geo.cand_total;
  int32_t* const b_cell_cnt = ctx->d_cell_cnt + (size_t)f0 * geo.cells.size();
  uint32_t* const b_pts = ctx->d_pts + (size_t)f0 * 2 * geo.cand_total;
  uint32_t* const b_lvl_kp = ctx->d_lvl_kp + (size_t)f0 * geo.kp_total;
  int32_t* const b_lvl_n = ctx->d_lvl_n + (size_t)f0 * geo.nlevels;
  uint2* const b_kp_list = ctx->d_kp_list + (size_t)f0 * ctx->out_cap;
  // K2 (level 0): FAST over the cells of level 0 only needs the input image, so it is forked onto a second stream and
  // runs concurrently with the (latency-bound) pyramid chain; joined before the quadtree.  Off by default for a lone
  // context (no gain: both kernels fill the CUs); the replay lanes switch it on (orbx_set_option), where it pays.
  const int need = geo.max_cell_w + 3;  // +3: alignment shift of the dword-staged rows
  const int pitchB = need <= 64 ? 64 : 96;   // LDS pitch of the FAST tile (another pitch against the bank conflicts was measured slower: HISTORY.md, round 5)
  if (need > 96 || geo.max_cell_h > 127 + 6 || round_up((geo.max_cell_w - 6) * (geo.max_cell_h - 6), 8) > 8192)
    return set_err(ctx, ORBX_E_CAPACITY, "FAST cell larger than the kernel's LDS tile");
  if (!div_ok((uint64_t)geo.cells.size() * nframes + 8, geo.cells.size()) ||
      !div_ok((uint64_t)geo.btiles_total * nframes + 8, geo.btiles_total))
    return set_err(ctx, ORBX_E_CAPACITY, "batch too large for 32-bit tile indexing");
  const int ft = ctx->fast_threads;
  // the reference runs cv::FAST at iniThFAST and, where that leaves a cell empty, again at minThFAST (src/ORBextractor.cc:826-850): with
  // minThFAST > iniThFAST the second run can only find a subset of nothing, i.e. the cell's result is iniThFAST's whatever minThFAST says
  const int th_min = std::min(ctx->min_th, ctx->ini_th);
  // LDS of a FAST workgroup over cells [c0, c1): tile + score plane of the tallest cell, list for the largest detection domain,
  // bitmap + word prefix (64 words up to 2048 pixels, 256 beyond)
  struct FastLds { int tile_rows, list_cap, nwords; size_t bytes; };
  auto fast_lds_of = [&](int c0, int c1) {
    FastLds f{1, 8, 64, 0};
    for (int c = c0; c < c1; c++) {
      const CellGeom& cg = geo.cells[c];
      f.tile_rows = std::max(f.tile_rows, (int)cg.ch);
      const int npx = (cg.cw - 6) * (cg.ch - 6);
      f.list_cap = std::max(f.list_cap, round_up(npx, 8));
      if (npx > 2048) f.nwords = 256;
    }
    f.bytes = 16 + (size_t)pitchB * f.tile_rows * 2 + (size_t)f.list_cap * 2 + (size_t)f.nwords * 8;
    return f;
  };
  auto fast_kern = pitchB == 64 ? (ft == 64 ? k_fast_cells<64, 64> : ft == 128 ? k_fast_cells<128, 64> : k_fast_cells<256, 64>)
                                : (ft == 64 ? k_fast_cells<64, 96> : ft == 128 ? k_fast_cells<128, 96> : k_fast_cells<256, 96>);
  if (ctx->fast_pk && ft == 128) fast_kern = pitchB == 64 ? k_fast_cells<128, 64, true> : k_fast_cells<128, 96, true>;
  if (ctx->fast_pk && ft == 64) fast_kern = pitchB == 64 ? k_fast_cells<64, 64, true> : k_fast_cells<64, 96, true>;
  auto launch_fast_range = [&](int cell_base, int ncells_sub, hipStream_t s) {
    const int nitems = ncells_sub * nframes;
    if (nitems <= 0) return;
    const FastLds f = fast_lds_of(cell_base, cell_base + ncells_sub);
    hipLaunchKernelGGL(fast_kern, dim3(xcd_grid(nitems)), dim3(ft), f.bytes, s, ctx->d_geo, ctx->d_cells, d_imgs,
                       (long long)row_stride, (long long)frame_stride, b_pyr, (long long)geo.pyr_bytes, b_cand, b_cell_cnt,
                       ctx->ini_th, th_min, f.tile_rows, nitems, cell_base, ncells_sub, div_magic((uint32_t)ncells_sub),
                       (ctx->fast_stage_dma ? 1 : 0) | (ctx->fast_passes == 2 ? 2 : 0), f.list_cap, f.nwords);
  };
  // The cells of the small levels are taller (fewer rows of cells share the same height): one launch over all levels would give
  // every workgroup the LDS of the tallest cell and cost the many cells of the large levels their residency.  A range of cells
  // is therefore split once, at the level boundary that maximises (cells x workgroups per CU), when that gains >= 5 %
  // (batch calls only: a single frame is bound by the number of launches).
  auto wgs_per_cu = [&](size_t bytes) { return (int)std::min<size_t>(32 / (ft / 64), (160 * 1024) / (bytes + 64 + 512)); };
  auto launch_fast = [&](int cell_base, int ncells_sub, hipStream_t s) {
    const int cend = cell_base + ncells_sub;
    int best_split = -1;
    if (nframes * geo.nlevels > 512 && ctx->fast_split) {
      const long whole = (long)ncells_sub * wgs_per_cu(fast_lds_of(cell_base, cend).bytes);
      long best = whole + whole / 20;
      for (int l = 0; l + 1 < geo.nlevels; l++) {
        const int b = geo.lv[l].cell_begin + geo.lv[l].ncells;   // first cell of level l + 1
        if (b <= cell_base || b >= cend) continue;
        const long sc = (long)(b - cell_base) * wgs_per_cu(fast_lds_of(cell_base, b).bytes) + (long)(cend - b) * wgs_per_cu(fast_lds_of(b, cend).bytes);
        if (sc > best) { best = sc; best_split = b; }
      }
    }
    if (best_split < 0) { launch_fast_range(cell_base, ncells_sub, s); return; }
    launch_fast_range(cell_base, best_split - cell_base, s);
    launch_fast_range(best_split, cend - best_split, s);
  };
  const int ncells0 = geo.lv[0].ncells, ncells_all = (int)geo.cells.size();
  // small batches (the single-frame operator() path) are latency-bound: a stream fork costs more than it hides there — also
  // inside the captured graph, where the runtime pays for a multi-stream graph on the host (measured: launch call 8 -> 56 us,
  // operator() 0.175 -> 0.187 ms with level-0 FAST and the blur on side branches)
  const bool small_batch = nframes * geo.nlevels <= 512;
  // every aux stream forked off `st` below is joined back into it when this function returns — also on an error return, so
  // that a caller's stream capture stays valid and no buffer is reused while forked work is pending
  struct ForkGuard {
    hipStream_t st;
    std::vector<std::pair<hipStream_t, hipEvent_t> > open;
    void forked(hipStream_t s, hipEvent_t join_ev) { open.push_back(std::make_pair(s, join_ev)); }
    void joined(hipStream_t s) { for (size_t i = 0; i < open.size(); i++) if (open[i].first == s) { open.erase(open.begin() + i); break; } }
    ~ForkGuard() { for (auto& o : open) { (void)hipEventRecord(o.second, o.first); (void)hipStreamWaitEvent(st, o.second, 0); } }
  } forks{st, {}};
  // checks that can fail come before the first fork
  for (int l = 1; l < geo.nlevels; l++) {
    const LevelGeom& D = geo.lv[l];
    const int nbx = (D.w + kRT_W - 1) / kRT_W, nby = (D.h + kRT_H - 1) / kRT_H;
    if (!div_ok((uint64_t)nbx * nby * nframes + 8, (uint64_t)nbx * nby)) return set_err(ctx, ORBX_E_CAPACITY, "batch too large for 32-bit tile indexing");
  }
  const bool fork_fast0 = ctx->fork_fast0 && !ctx->profiling && geo.nlevels > 1 && !small_batch;
  const int sb = f0 != 0;  // sub-batch slot of the fork events / streams
  if (fork_fast0) {
    hipStream_t fst = ctx->aux[orbx_ctx::kMaxAux - 3 - sb];
    ORBX_HIP(ctx, hipEventRecord(ctx->ev_f0_fork[sb], st));
    ORBX_HIP(ctx, hipStreamWaitEvent(fst, ctx->ev_f0_fork[sb], 0));
    forks.forked(fst, ctx->ev_f0_join[sb]);
    launch_fast(0, ncells0, fst);
    ORBX_HIP(ctx, hipEventRecord(ctx->ev_f0_join[sb], fst));
  }
  // small batches (the single-frame graph): every launch costs about 5 us of device time whatever it does, so FAST and the blur —
  // both only read the finished pyramid — share one launch, and the assembly runs as the tail of the quadtree launch
  // (latency-bound calls only: at 32 frames of 1024 x 1024 — config 4's replay lanes, still a "small batch" by the fork rule above — the shared
  // launch with its 256-thread FAST workgroups and worst-case LDS costs 20 % of the throughput)
  const bool small_fused = small_batch && nframes <= 4 && ctx->small_fused && !ctx->profiling;
  // K1, small batches: groups of consecutive levels in one launch each (k_resize_chain); the plan (which levels, LDS rectangles) and
  // the coverage check run on the host tables; anything the plan cannot serve takes the launch per level below
  int chained_upto = 0;   // levels 1 .. chained_upto are produced by chain launches
  if ((small_fused || ctx->chain_batch) && geo.nlevels > 2) {
    ProfScope ps(ctx, 0, st);
    int l = 1;
    while (l < geo.nlevels) {
      const int left = geo.nlevels - l;
      // which levels share a launch.  A single frame: ALL of them (up to seven per launch), on 16 x 16 tiles of the last level — 108 workgroups
      // at 640 x 480, each recomputing the rectangles of the six levels above its tile (about twice the pixels of the pyramid in total: a few
      // microseconds of arithmetic), against ≈ 8 us of device time for every launch saved inside the replayed graph.  Measured on the whole
      // operator(): device 110.8 us with the groups (1,2)(3,4)(5,6,7) of 64 x 64 tiles, 104.8 / 102.2 with (1,2)(3..7) on 32 / 16-px tiles,
      // 96.7 with one launch ("chain_long" = 0: the groups of two or three; "chain_first": levels in the first launch; "chain_long_tile")
      int K = left == 3 ? 3 : (left >= 2 ? 2 : 1);
      if (ctx->chain_long && !ctx->chain_batch && left >= 2) K = l > 1 ? std::min(left, 7) : std::min(left, ctx->chain_first);
      if (K == 1) break;
      // rectangles of every tile of the group's last level, as the kernel derives them
      const LevelGeom& LD = geo.lv[l + K - 1];
      const int tile = K >= 4 ? ctx->chain_long_tile : kRT_W;   // long chains: smaller tiles, more workgroups
      const int cthreads = K >= 4 ? std::min(ctx->chain_threads, 512) : ctx->chain_threads;
      const int nbx = (LD.w + tile - 1) / tile, nby = (LD.h + tile - 1) / tile;
      int mw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mh[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      std::vector<std::vector<uint8_t> > colhit(K), rowhit(K);
      for (int k = 1; k < K; k++) { colhit[k].assign(geo.lv[l + k - 1].w, 0); rowhit[k].assign(geo.lv[l + k - 1].h, 0); }
      for (int b = 0; b < std::max(nbx, nby); b++) {   // columns and rows are independent: one sweep over tile columns, one over tile rows
        int X0 = b * tile, X1 = std::min(b * tile + tile, LD.w) - 1, Y0 = b * tile, Y1 = std::min(b * tile + tile, LD.h) - 1;
        for (int k = K; k >= 1; k--) {
          const LevelGeom& L = geo.lv[l + k - 1];
          if (b < nbx) {
            const XTab ta = geo.xtab[L.xtab_off + X0], tb = geo.xtab[L.xtab_off + std::min(X1, L.w - 1)];
            X0 = (int)ta.s0 & ~3; X1 = std::max((int)tb.s0, (int)tb.s1) | 3;
            mw[k - 1] = std::max(mw[k - 1], X1 - X0 + 1);
            if (k - 1 >= 1) for (int x = X0; x <= std::min(X1, (int)colhit[k - 1].size() - 1); x++) colhit[k - 1][x] = 1;
          }
          if (b < nby) {
            const XTab ua = geo.ytab[L.ytab_off + Y0], ub = geo.ytab[L.ytab_off + Y1];
            Y0 = ua.s0; Y1 = std::max((int)ub.s0, (int)ub.s1);
            mh[k - 1] = std::max(mh[k - 1], Y1 - Y0 + 1);
            if (k - 1 >= 1) for (int y = Y0; y <= Y1; y++) rowhit[k - 1][y] = 1;
          }
        }
      }
      bool covered = true;
      for (int k = 1; k < K && covered; k++) {
        for (uint8_t h : colhit[k]) covered = covered && h;
        for (uint8_t h : rowhit[k]) covered = covered && h;
      }
      size_t lds = 0;
      int off[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < K; k++) { off[k] = (int)lds; lds += (size_t)mw[k] * mh[k]; lds = (lds + 15) & ~(size_t)15; }
      if (!covered || lds > kLdsMax) break;
      const void* chain_kern = K == 2 ? (const void*)k_resize_chain<2> : K == 3 ? (const void*)k_resize_chain<3> : K == 4 ? (const void*)k_resize_chain<4>
                               : K == 5 ? (const void*)k_resize_chain<5> : K == 6 ? (const void*)k_resize_chain<6> : (const void*)k_resize_chain<7>;
      if (lds > 64 * 1024 && ensure_dynamic_lds(chain_kern, (int)lds) != hipSuccess) { (void)hipGetLastError(); break; }
      auto fill = [&](auto& ca) {
        if (l == 1) { ca.src = d_imgs; ca.src_frame_stride = (long long)frame_stride; ca.src_pitch = (int)row_stride; }
        else { ca.src = b_pyr + geo.lv[l - 1].plane_off; ca.src_frame_stride = geo.pyr_bytes; ca.src_pitch = geo.lv[l - 1].pitch; }
        ca.sw = geo.lv[l - 1].w; ca.dst_frame_stride = (long long)geo.pyr_bytes; ca.nbx = nbx; ca.nby = nby; ca.tile = tile;
        for (int k = 0; k < K; k++) {
          const LevelGeom& L = geo.lv[l + k];
          ca.lv[k].xt = ctx->d_xtab + L.xtab_off; ca.lv[k].yt = ctx->d_ytab + L.ytab_off; ca.lv[k].dst = b_pyr + L.plane_off;
          ca.lv[k].pitch = L.pitch; ca.lv[k].w = L.w; ca.lv[k].h = L.h;
          ca.buf_off[k] = off[k]; ca.buf_pitch[k] = mw[k];
        }
      };
      if (K == 2) { ChainArgs<2> ca; fill(ca); hipLaunchKernelGGL(k_resize_chain<2>, dim3(nbx * nby, nframes), dim3(cthreads), lds, st, ca); }
      else if (K == 3) { ChainArgs<3> ca; fill(ca); hipLaunchKernelGGL(k_resize_chain<3>, dim3(nbx * nby, nframes), dim3(cthreads), lds, st, ca); }
      else if (K == 4) { ChainArgs<4> ca; fill(ca); hipLaunchKernelGGL(k_resize_chain<4>, dim3(nbx * nby, nframes), dim3(cthreads), lds, st, ca); }
      else if (K == 5) { ChainArgs<5> ca; fill(ca); hipLaunchKernelGGL(k_resize_chain<5>, dim3(nbx * nby, nframes), dim3(cthreads), lds, st, ca); }
      else if (K == 6) { ChainArgs<6> ca; fill(ca); hipLaunchKernelGGL(k_resize_chain<6>, dim3(nbx * nby, nframes), dim3(cthreads), lds, st, ca); }
      else { ChainArgs<7> ca; fill(ca); hipLaunchKernelGGL(k_resize_chain<7>, dim3(nbx * nby, nframes), dim3(cthreads), lds, st, ca); }
      chained_upto = l + K - 1;
      l += K;
    }
  }
  // K1: pyramid chain (levels depend on each other: one launch per level over the whole batch)
  {
    ProfScope ps(ctx, 0, st);
    for (int l = chained_upto + 1; l < geo.nlevels; l++) {
      const LevelGeom& D = geo.lv[l];
      const LevelGeom& S = geo.lv[l - 1];
      const uint8_t* src; long long sfs; int sp;
      if (l == 1) { src = d_imgs; sfs = (long long)frame_stride; sp = (int)row_stride; }
      else { src = b_pyr + S.plane_off; sfs = geo.pyr_bytes; sp = S.pitch; }
      const int nbx = (D.w + kRT_W - 1) / kRT_W, nby = (D.h + kRT_H - 1) / kRT_H, nitems = nbx * nby * nframes;
      hipLaunchKernelGGL(k_resize, dim3(xcd_grid(nitems)), dim3(256), (size_t)D.rs_lds_pitch * D.rs_lds_rows, st, src, sfs, sp,
                         S.w, b_pyr + D.plane_off, (long long)geo.pyr_bytes, D.pitch, D.w, D.h, ctx->d_xtab + D.xtab_off,
                         ctx->d_ytab + D.ytab_off, nbx, nby, nitems, D.rs_lds_pitch, D.rs_lds_rows,
                         div_magic((uint32_t)(nbx * nby)), div_magic((uint32_t)nbx), div_magic((uint32_t)(D.rs_lds_pitch / 4)),
                         256 / (D.rs_lds_pitch / 4));
    }
  }
  BlurConsts bc;
  {
    int gk[7];
    gaussian_kernel7(gk, ctx->gauss_kernel);
    // the default variant needs neither saturation nor a tie rule nor a tail: the kernel's fast path (flags == 0)
    const bool general = ctx->gauss_kernel != 0 || ctx->gauss_round != 0;
    bc.radd = general ? 0u : 32768u;
    bc.flags = general ? (1u | ((uint32_t)ctx->gauss_round << 1)) : 0u;
    bc.tail_mask = (general && ctx->gauss_round != 0 && ctx->gauss_tail > 1) ? (uint32_t)ctx->gauss_tail - 1u : 0u;
    auto b4 = [&](int a, int b, int c, int d) {   // weights of the four bytes of a dword; index -1 = no tap
      auto w = [&](int i) { return i < 0 ? 0u : (uint32_t)gk[i]; };
      return w(a) | (w(b) << 8) | (w(c) << 16) | (w(d) << 24);
    };
    bc.hw[0] = b4(-1, 0, 1, 2); bc.hw[1] = b4(3, 4, 5, 6);                              // output x: bytes x+1 .. x+7 of d0 d1 d2
    bc.hw[2] = b4(-1, -1, 0, 1); bc.hw[3] = b4(2, 3, 4, 5); bc.hw[4] = b4(6, -1, -1, -1);
    bc.hw[5] = b4(-1, -1, -1, 0); bc.hw[6] = b4(1, 2, 3, 4); bc.hw[7] = b4(5, 6, -1, -1);
    bc.hw[8] = b4(0, 1, 2, 3); bc.hw[9] = b4(4, 5, 6, -1);
    const uint32_t k0 = gk[0], k1 = gk[1], k2 = gk[2], k3 = gk[3], k4 = gk[4], k5 = gk[5], k6 = gk[6];
    bc.we[0] = k0 | k1 << 16; bc.we[1] = k2 | k3 << 16; bc.we[2] = k4 | k5 << 16; bc.we[3] = k6;
    bc.wo[0] = k0 << 16; bc.wo[1] = k1 | k2 << 16; bc.wo[2] = k3 | k4 << 16; bc.wo[3] = k5 | k6 << 16;
  }
  // K2: FAST cells of the remaining levels (or of all levels when level 0 is not forked)
  if (small_fused) {
    const FastLds f = fast_lds_of(0, ncells_all);
    const int nfast = ncells_all * nframes, nblur = geo.btiles_total * nframes;
    auto fk = pitchB == 64 ? (ctx->fast_pk ? k_fast_blur<64, true> : k_fast_blur<64, false>) : (ctx->fast_pk ? k_fast_blur<96, true> : k_fast_blur<96, false>);
    hipLaunchKernelGGL(fk, dim3(nfast + nblur), dim3(256), f.bytes, st, ctx->d_geo, ctx->d_cells, d_imgs, (long long)row_stride, (long long)frame_stride,
                       b_pyr, (long long)geo.pyr_bytes, b_cand, b_cell_cnt, ctx->ini_th, th_min, f.tile_rows, nfast, ncells_all,
                       div_magic((uint32_t)ncells_all), ctx->fast_stage_dma ? 1 : 0, f.list_cap, f.nwords, b_blur, (long long)geo.blur_bytes, bc);
  } else {
    ProfScope ps(ctx, 1, st);
    if (fork_fast0) launch_fast(ncells0, ncells_all - ncells0, st);
    else launch_fast(0, ncells_all, st);
  }
  // K4a: 7x7 fixed-point Gaussian of every level (the reference blurs each level that holds keypoints).  It only needs
  // the pyramid, and it is VALU-bound while the quadtree that follows FAST is latency-bound with few workgroups, so it
  // is forked onto a second stream behind FAST and joined before the descriptors (ORBX_FORK_BLUR=0 disables).
  // Batches under the default blur arithmetic take the Gaussian inside the descriptor kernel (k_describe_blur): no k_blur7 launch, no blurred
  // planes (use_fused_blur has the rule).
  const bool fused_blur = use_fused_blur(ctx, nframes);
  ctx->last_fused_blur = fused_blur;
  if (!fused_blur && (!ctx->d_blur || ctx->blur_cap < f0 + nframes)) return set_err(ctx, ORBX_E_DEVICE, "internal: blurred planes not allocated");
  const bool fork_blur = ctx->fork_blur && !ctx->profiling && !small_batch && !fused_blur;
  hipStream_t bst = st;
  if (fork_blur) {
    bst = ctx->aux[orbx_ctx::kMaxAux - 1 - (f0 != 0)];
    ORBX_HIP(ctx, hipEventRecord(ctx->ev_blur_fork[f0 != 0], st));
    ORBX_HIP(ctx, hipStreamWaitEvent(bst, ctx->ev_blur_fork[f0 != 0], 0));
    forks.forked(bst, ctx->ev_blur_join[f0 != 0]);
  }
  if (!small_fused && !fused_blur) {
    ProfScope ps(ctx, 4, bst);
    const int nitems = geo.btiles_total * nframes;
    hipLaunchKernelGGL(k_blur7, dim3(xcd_grid(nitems)), dim3(256), 0, bst, ctx->d_geo, d_imgs, (long long)row_stride,
                       (long long)frame_stride, b_pyr, (long long)geo.pyr_bytes, b_blur, (long long)geo.blur_bytes, bc,
                       nitems);
  }
  if (fork_blur) ORBX_HIP(ctx, hipEventRecord(ctx->ev_blur_join[f0 != 0], bst));
  // K3: quadtree
  bool assembled = false;
  int direct_mode = 0;
  if (fork_fast0) { ORBX_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_f0_join[sb], 0)); forks.joined(ctx->aux[orbx_ctx::kMaxAux - 3 - sb]); }
  {
    ProfScope ps(ctx, 2, st);
    // The workgroup of (frame, level) is sized by the level's quota; one LDS size for all levels would let the big
    // levels halve the residency of the small ones, so the levels run as two launches: the first kQtBigLevels levels
    // (large quota, LDS points for 2048 candidates) and the rest (small node arrays, 1024 LDS points), the second one
    // forked onto an aux stream so both fill the CUs together.  Candidates beyond the LDS capacity use HBM buffers.
    std::function<int(int, int, int, hipStream_t)> launch_qt = [&](int l0, int l1, int pts_cap, hipStream_t s) -> int {
      int mq = 1, mc = 1, mp = 1;
      for (int l = l0; l < l1; l++) {
        mq = std::max(mq, geo.lv[l].quota); mc = std::max(mc, geo.lv[l].ncells); mp = std::max(mp, geo.lv[l].cand_cap);
      }
      int node_cap, scan_cap;
      qt_caps(mq, mc, mp, node_cap, scan_cap);   // beyond 65 532 nodes the kernel refuses at run time (never seen: that many corners on one level)
      size_t lds = qt_node_bytes(node_cap, scan_cap) + qt_point_bytes(pts_cap);
      uint8_t* gnodes = nullptr;
      if (lds > kLdsMax) {
        if (l1 - l0 > 1) {   // let every level of the group choose for itself
          for (int l = l0; l < l1; l++) { const int rc = launch_qt(l, l + 1, pts_cap, s); if (rc != ORBX_OK) return rc; }
          return ORBX_OK;
        }
        pts_cap = 0;         // points in HBM
        lds = qt_node_bytes(node_cap, scan_cap);
        if (lds > kLdsMax) {  // node arrays in HBM too (slow, but the level is served)
          if (ctx->qt_node_slot[l0] < 0 || !ctx->d_qt_nodes) return set_err(ctx, ORBX_E_CAPACITY, "quadtree node scratch missing");
          gnodes = ctx->d_qt_nodes + ((size_t)ctx->qt_node_slot[l0] * ctx->batch_cap + f0) * ctx->qt_node_stride;
          lds = 0;
        }
      }
      if (lds > 64 * 1024) {
        if (ensure_dynamic_lds((const void*)k_quadtree, (int)lds) != hipSuccess)
          return set_err(ctx, ORBX_E_CAPACITY, "nfeatures too large for the quadtree kernel's LDS");
      }
      // small batches are latency-bound on the big levels' workgroups: more waves split more nodes at a time
      // one to four frames are latency-bound on the big levels' workgroups (more waves split more nodes at a time: 512); from there on the
      // launch is throughput-bound and 256 wins (config 4's 32-frame lanes of 1024 x 1024: 134.1 k -> 144.6 k features/ms)
      int qthreads = ctx->qt_threads ? ctx->qt_threads : nframes <= 4 ? 512 : 256;
      if (!small_batch && l0 >= ctx->qt_big_levels && l0 > 0 && ctx->qt_threads_small) qthreads = ctx->qt_threads_small;   // the small levels' launch of a batch
      // level-major order (LDS-resident node arrays only: the HBM node slices are indexed frame-major): all workgroups of the largest
      // level of the launch are dispatched first, the short ones fill the CUs behind them
      if (ctx->qt_level_major && !gnodes && !small_batch)
        hipLaunchKernelGGL(k_quadtree, dim3(nframes, l1 - l0, 1), dim3(qthreads), lds, s, ctx->d_geo, ctx->d_cells, b_cand, b_cell_cnt,
                           b_pts, b_lvl_kp, b_lvl_n, node_cap, scan_cap, pts_cap, l0 | 0x100 | (ctx->qt_fused ? 0 : 0x200), gnodes, (long long)ctx->qt_node_stride);
      else
        hipLaunchKernelGGL(k_quadtree, dim3(l1 - l0, nframes, 1), dim3(qthreads), lds, s, ctx->d_geo, ctx->d_cells, b_cand, b_cell_cnt,
                           b_pts, b_lvl_kp, b_lvl_n, node_cap, scan_cap, pts_cap, l0 | (ctx->qt_fused ? 0 : 0x200), gnodes, (long long)ctx->qt_node_stride);
      return ORBX_OK;
    };
    const int qt_pts = ctx->qt_points;   // measured: 1024 ... 2048 points make no difference to the launch (128 VGPRs hold it at four workgroups per CU)
    const int nbig = (geo.nlevels >= 4 && !small_batch && !ctx->qt_one_launch) ? std::min(ctx->qt_big_levels, geo.nlevels) : geo.nlevels;  // small batch: one launch, all levels
    int qrc = ORBX_OK;
    // a single frame whose lapping area holds every keypoint or none (every configuration but the fisheye rigs and images wider than the
    // 1000 columns Frame.cc passes): no assembly at all — k_describe finds its keypoints in the quadtree's per-level output (DIRECT)
    if (small_fused && nframes == 1 && ctx->describe_direct && !ctx->desc_k_user)
      direct_mode = (lap0 <= 0 && lap1 >= cols) ? 1 : (lap1 < 16 || lap0 >= cols) ? 2 : 0;
    if (direct_mode) {
    } else if (small_fused && ctx->d_qt_fin && !ctx->d_asm_scan && nframes <= kSmallBatchFrames) {
      // all levels in one launch with the assembly as its tail (k_quadtree_assemble), when everything is LDS-resident
      int mq = 1, mc = 1, mp = 1;
      for (int l = 0; l < geo.nlevels; l++) { mq = std::max(mq, geo.lv[l].quota); mc = std::max(mc, geo.lv[l].ncells); mp = std::max(mp, geo.lv[l].cand_cap); }
      int node_cap, scan_cap;
      qt_caps(mq, mc, mp, node_cap, scan_cap);
      const size_t lds = std::max(qt_node_bytes(node_cap, scan_cap) + qt_point_bytes(qt_pts), (size_t)ctx->out_cap * 8 + 64);
      if (lds <= kLdsMax && (lds <= 64 * 1024 || ensure_dynamic_lds((const void*)k_quadtree_assemble, (int)lds) == hipSuccess)) {
        QtaArgs qa;
        qa.g = ctx->d_geo; qa.cells = ctx->d_cells; qa.cand = b_cand; qa.cell_cnt = b_cell_cnt; qa.pts = b_pts; qa.lvl_kp = b_lvl_kp; qa.lvl_n = b_lvl_n;
        qa.node_cap = node_cap; qa.scan_cap = scan_cap; qa.pts_cap = qt_pts;
        qa.kp_list = b_kp_list; qa.counts = d_counts; qa.lap0 = lap0; qa.lap1 = lap1; qa.fin = ctx->d_qt_fin;
        qa.mirror_counts = (mirror && nframes == 1) ? mirror->counts : nullptr;
        hipLaunchKernelGGL(k_quadtree_assemble, dim3(geo.nlevels, nframes, 1), dim3(ctx->qt_threads ? ctx->qt_threads : 512), lds, st, qa);
        assembled = true;
      }
    }
    if (assembled) {
    } else if (direct_mode) {
      qrc = launch_qt(0, geo.nlevels, qt_pts, st);
      if (qrc != ORBX_OK) return qrc;
    } else if (nbig < geo.nlevels && !ctx->profiling && ctx->fork_qt) {
      hipStream_t qst = ctx->aux[orbx_ctx::kMaxAux - 5 - sb];
      ORBX_HIP(ctx, hipEventRecord(ctx->ev_qt_fork[sb], st));
      ORBX_HIP(ctx, hipStreamWaitEvent(qst, ctx->ev_qt_fork[sb], 0));
      forks.forked(qst, ctx->ev_qt_join[sb]);
      qrc = launch_qt(nbig, geo.nlevels, qt_pts / 2, qst);
      if (qrc != ORBX_OK) return qrc;
      ORBX_HIP(ctx, hipEventRecord(ctx->ev_qt_join[sb], qst));
      qrc = launch_qt(0, nbig, qt_pts, st);
      if (qrc != ORBX_OK) return qrc;
      ORBX_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_qt_join[sb], 0));
      forks.joined(qst);
    } else {
      qrc = launch_qt(0, nbig, qt_pts, st);
      if (qrc != ORBX_OK) return qrc;
      if (nbig < geo.nlevels) { qrc = launch_qt(nbig, geo.nlevels, qt_pts / 2, st); if (qrc != ORBX_OK) return qrc; }
    }
  }
  // K3b: output slots
  if (!assembled && !direct_mode) {
    ProfScope ps(ctx, 3, st);
    const bool gs = ctx->d_asm_scan != nullptr;
    hipLaunchKernelGGL(k_assemble, dim3(nframes), dim3(256), gs ? 0 : (size_t)ctx->out_cap * 8 + 64, st, ctx->d_geo, b_lvl_kp,
                       b_lvl_n, b_kp_list, d_counts, lap0, lap1, gs ? ctx->d_asm_scan + (size_t)f0 * ctx->out_cap : nullptr);
  }
  // K4b: orientation + descriptors
  if (fork_blur) { ORBX_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_blur_join[f0 != 0], 0)); forks.joined(bst); }
  {
    ProfScope ps(ctx, 5, st);
    const bool use_mirror = mirror && (assembled || direct_mode) && nframes == 1 && mirror->kps && mirror->desc && mirror->counts;   // counts: mirrored by the quadtree's tail, or by k_describe (DIRECT)
    DescConsts dc;
    for (int i = 0; i < 16; i++) dc.umax[i] = ctx->umax[i];
    // keypoints per wave: K = 4 amortises the trig pass in a batch; a single frame has 1000 keypoints for 1024 SIMDs and is served fastest by
    // one keypoint per wave (operator() -6 us) unless the caller chose a value
    const int K = (small_fused && !ctx->desc_k_user) ? 1 : ctx->desc_k;
    const int gpf = (ctx->out_cap + 4 * K - 1) / (4 * K), nitems = gpf * nframes;
    auto kern = K == 1 ? k_describe<1> : K == 2 ? k_describe<2> : K == 4 ? k_describe<4> : K == 8 ? k_describe<8> : k_describe<16>;
    if (ctx->desc_lds && (K == 2 || K == 4 || K == 8)) kern = K == 2 ? k_describe<2, true> : K == 4 ? k_describe<4, true> : k_describe<8, true>;
    if (fused_blur) {
      constexpr int KF = 2;   // keypoints per wave and round: two raw slices + one row-pair buffer per wave keep five workgroups on a CU (measured: K = 1 and K = 4 are 9 % slower)
      constexpr int RF = 1;   // rounds per workgroup (k_describe_blur says why 1)
      const int gpf_f = (ctx->out_cap + 4 * KF * RF - 1) / (4 * KF * RF), nitems_f = gpf_f * nframes;
      hipLaunchKernelGGL((k_describe_blur<KF, RF>), dim3(xcd_grid(nitems_f)), dim3(256), 0, st, ctx->d_geo, d_imgs, (long long)row_stride,
                         (long long)frame_stride, b_pyr, (long long)geo.pyr_bytes, b_kp_list, d_counts, d_kps, d_desc, bc, gpf_f, nitems_f,
                         div_magic((uint32_t)gpf_f), ctx->atan_fma, ctx->brief_fma);
    } else if (direct_mode)
      hipLaunchKernelGGL((k_describe<1, false, true>), dim3(xcd_grid(nitems)), dim3(256), 0, st, ctx->d_geo, d_imgs, (long long)row_stride,
                         (long long)frame_stride, b_pyr, (long long)geo.pyr_bytes, b_blur, (long long)geo.blur_bytes,
                         b_kp_list, d_counts, d_kps, d_desc, dc, gpf, nitems, div_magic((uint32_t)gpf),
                         use_mirror ? mirror->kps : (orbx_keypoint*)nullptr, use_mirror ? mirror->desc : (uint8_t*)nullptr,
                         (const uint32_t*)b_lvl_kp, (const int32_t*)b_lvl_n, direct_mode, d_counts, use_mirror ? mirror->counts : (int32_t*)nullptr,
                         ctx->atan_fma, ctx->brief_fma);
    else
    hipLaunchKernelGGL(kern, dim3(xcd_grid(nitems)), dim3(256), 0, st, ctx->d_geo, d_imgs, (long long)row_stride,
                       (long long)frame_stride, b_pyr, (long long)geo.pyr_bytes, b_blur, (long long)geo.blur_bytes,
                       b_kp_list, d_counts, d_kps, d_desc, dc, gpf, nitems, div_magic((uint32_t)gpf),
                       use_mirror ? mirror->kps : (orbx_keypoint*)nullptr, use_mirror ? mirror->desc : (uint8_t*)nullptr,
                       (const uint32_t*)nullptr, (const int32_t*)nullptr, 0, (int32_t*)nullptr, (int32_t*)nullptr, ctx->atan_fma, ctx->brief_fma);
    if (use_mirror && mirrored) *mirrored = true;
  }
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

}  // namespace orbx

// ---- whose rows are these?  (Frame::ComputeBoW without a device round trip of its own) ---------------------------------------------------
// The extractor adapter copies the rows it got from orbx_extract into the caller's cv::Mat (Frame::mDescriptors) and tells the library
// which host buffer now holds the rows of the context's last extraction (orbx_publish_descriptors).  ORBVocabulary::transform is handed
// row headers into that very buffer by the unmodified Frame.cc — the host address (+ row count + a digest of the bytes) is the only key
// the reference's call chain offers — and finds the {word, node, weight} records the extraction graph left in the pinned result block
// (orbx_bow_transform_published).  Contract, as in the reference: nobody writes into mDescriptors after ExtractORB.
// (Rounds 3-4 also handed the rows over to the first search TARGET made from that buffer, device to device; measured at 58.0 us for the
// first search of a new frame against 51.2 us with the host rows and a caller-held grid — profiles/target_latency_r4.txt — it was removed
// in round 5 together with its publication list, in-flight counts and per-target digests.)
namespace {
std::mutex g_pub_mu;
std::condition_variable g_pub_cv;          // a context left its extraction (in_extract dropped)
std::vector<orbx_ctx*> g_bow_attached;     // contexts with a vocabulary attached (orbx_bow_transform_published)
std::vector<orbx_ctx*> g_publishers;       // every context that has ever published (until it is destroyed): who extracted the rows in this buffer?
}  // namespace

// a new extraction begins on ctx: its pinned result block is about to be overwritten, what was published about it is void
static void begin_extraction(orbx_ctx* ctx) {
  std::lock_guard<std::mutex> lock(g_pub_mu);
  ctx->extract_seq++;
  ctx->in_extract = true;
  ctx->last_n0 = -1;
  ctx->rows_host = nullptr; ctx->rows_n = 0;
  // attached / detached since the last call (or the attached vocabulary was destroyed and another one now lives at its address: bow_gen):
  // the graph changes
  if (ctx->bow_active != ctx->bow_voc || ctx->bow_active_levelsup != ctx->bow_levelsup || ctx->bow_active_gen != ctx->bow_gen) {
    ctx->bow_active = ctx->bow_voc; ctx->bow_active_levelsup = ctx->bow_levelsup; ctx->bow_active_gen = ctx->bow_gen;
    ctx->buf_epoch++;
  }
}
// the extraction has finished (n0 >= 0: its rows are in the pinned block; < 0: it failed) — on every path out of extract_batch_impl
static void end_extraction(orbx_ctx* ctx, int n0, bool bow_records) {
  std::lock_guard<std::mutex> lock(g_pub_mu);
  ctx->last_n0 = n0;
  if (n0 >= 0 && bow_records) ctx->bow_seq = ctx->extract_seq;   // the pinned block holds this extraction's {word, node, weight} records
  ctx->in_extract = false;
  g_pub_cv.notify_all();
}
namespace {
struct ExtractionScope {   // begin_extraction ... end_extraction around one host-buffer extraction, whatever path leaves it
  orbx_ctx* ctx; int n0 = -1; bool bow = false;
  explicit ExtractionScope(orbx_ctx* c) : ctx(c) { begin_extraction(c); }
  ~ExtractionScope() { end_extraction(ctx, n0, bow); }
};
}  // namespace
void orbx::unpublish_context(orbx_ctx* ctx) {   // orbx_destroy
  std::lock_guard<std::mutex> lock(g_pub_mu);
  ctx->last_n0 = -1;
  ctx->bow_voc = nullptr; ctx->rows_host = nullptr;
  g_bow_attached.erase(std::remove(g_bow_attached.begin(), g_bow_attached.end(), ctx), g_bow_attached.end());
  g_publishers.erase(std::remove(g_publishers.begin(), g_publishers.end(), ctx), g_publishers.end());
}
// orbx_voc_destroy: no context may go on descending this tree.  Detaches it everywhere, invalidates the snapshots (bow_gen: a vocabulary
// created later at the same address is a different one) and waits for extractions in flight that already took their snapshot — a
// single-frame extraction is synchronous, so once in_extract has dropped its graph is done with the tree.
void orbx::voc_detach_all(const orbx_voc* v) {
  std::unique_lock<std::mutex> lock(g_pub_mu);
  std::vector<orbx_ctx*> users;
  for (orbx_ctx* c : g_bow_attached)
    if (c->bow_voc == v || c->bow_active == v) { c->bow_voc = nullptr; c->bow_seq = ~0ull; c->bow_gen++; users.push_back(c); }
  g_bow_attached.erase(std::remove_if(g_bow_attached.begin(), g_bow_attached.end(), [](orbx_ctx* c) { return c->bow_voc == nullptr; }), g_bow_attached.end());
  g_pub_cv.wait(lock, [&] { for (orbx_ctx* c : users) if (c->in_extract && c->bow_active == v) return false; return true; });
}

using namespace orbx;

extern "C" {

int orbx_create(orbx_ctx** out, int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int device_id) {
  if (!out) return ORBX_E_INVALID;
  *out = nullptr;
  if (nfeatures < 1 || nlevels < 1 || nlevels > kMaxLevels || !(scale_factor > 1.0f) || ini_th < 0 || min_th < 0)
    return ORBX_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ORBX_E_DEVICE;
  if (device_id < 0) { if (hipGetDevice(&device_id) != hipSuccess) return ORBX_E_DEVICE; }
  if (device_id >= ndev) return ORBX_E_INVALID;
  orbx_ctx* ctx = new (std::nothrow) orbx_ctx();
  if (!ctx) return ORBX_E_CAPACITY;
  ctx->nfeatures = nfeatures; ctx->nlevels = nlevels; ctx->ini_th = std::min(std::max(ini_th, 0), 255);
  ctx->min_th = std::min(std::max(min_th, 0), 255); ctx->device = device_id;
  ctx->scale_factor = scale_factor;  // include/ORBextractor.h:96: double member initialised from the float argument
  // src/ORBextractor.cc:414-430
  ctx->scale.resize(nlevels); ctx->sigma2.resize(nlevels); ctx->inv_scale.resize(nlevels); ctx->inv_sigma2.resize(nlevels);
  ctx->scale[0] = 1.0f; ctx->sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    ctx->scale[i] = (float)(ctx->scale[i - 1] * ctx->scale_factor);
    ctx->sigma2[i] = ctx->scale[i] * ctx->scale[i];
  }
  for (int i = 0; i < nlevels; i++) { ctx->inv_scale[i] = 1.0f / ctx->scale[i]; ctx->inv_sigma2[i] = 1.0f / ctx->sigma2[i]; }
  // src/ORBextractor.cc:434-445
  ctx->quota.resize(nlevels);
  const float factor = (float)(1.0f / ctx->scale_factor);
  float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    ctx->quota[l] = cv_round(desired);
    sum += ctx->quota[l];
    desired *= factor;
  }
  ctx->quota[nlevels - 1] = std::max(nfeatures - sum, 0);
  // src/ORBextractor.cc:453-468
  {
    int* umax = ctx->umax;
    int v, v0, vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }
  ctx->fast_threads = fast_threads_from_env();
  { const char* e = getenv("ORBX_SMALL_FUSED"); ctx->small_fused = e ? atoi(e) != 0 : true; }
  { const char* e = getenv("ORBX_QT_THREADS_SMALL"); const int v = e ? atoi(e) : 128; ctx->qt_threads_small = (v == 64 || v == 128 || v == 192 || v == 256 || v == 512) ? v : 0; }
  { const char* e = getenv("ORBX_QT_BIG_LEVELS"); const int v = e ? atoi(e) : 0; ctx->qt_big_levels = (v >= 1 && v <= 8) ? v : kQtBigLevels; }
  { const char* e = getenv("ORBX_CHAIN_LONG"); ctx->chain_long = e ? atoi(e) != 0 : true; }
  { const char* e = getenv("ORBX_DESCRIBE_DIRECT"); ctx->describe_direct = e ? atoi(e) != 0 : true; }
  { const char* e = getenv("ORBX_CHAIN_LONG_TILE"); const int v = e ? atoi(e) : 16; ctx->chain_long_tile = (v >= 8 && v <= 64 && v % 4 == 0) ? v : 16; }
  { const char* e = getenv("ORBX_CHAIN_FIRST"); const int v = e ? atoi(e) : 7; ctx->chain_first = (v >= 2 && v <= 7) ? v : 7; }
  { const char* e = getenv("ORBX_CHAIN_BATCH"); ctx->chain_batch = e ? atoi(e) != 0 : false; }
  { const char* e = getenv("ORBX_QT_LEVEL_MAJOR"); ctx->qt_level_major = e ? atoi(e) != 0 : true; }
  { const char* e = getenv("ORBX_QT_FUSED"); ctx->qt_fused = e ? atoi(e) != 0 : true; }
  { const char* e = getenv("ORBX_QT_ONE_LAUNCH"); ctx->qt_one_launch = e ? atoi(e) != 0 : false; }
  { const char* e = getenv("ORBX_CHAIN_THREADS"); const int v = e ? atoi(e) : 1024; ctx->chain_threads = (v == 256 || v == 512 || v == 1024) ? v : 1024; }
  { const char* e = getenv("ORBX_QT_POINTS"); const int v = e ? atoi(e) : kQtLdsPoints; ctx->qt_points = (v >= 256 && v <= 4096 && v % 128 == 0) ? v : kQtLdsPoints; }
  { const char* e = getenv("ORBX_FAST_SPLIT"); ctx->fast_split = e ? atoi(e) != 0 : true; }
  { const char* e = getenv("ORBX_WINDOW_DIRECT"); ctx->window_direct = e ? atoi(e) != 0 : true; }
  { const char* e = getenv("ORBX_QT_THREADS"); const int v = e ? atoi(e) : 0; ctx->qt_threads = (v >= 64 && v <= 512 && v % 64 == 0) ? v : 0; }
  {
    const char* e = getenv("ORBX_DESC_K");  // keypoints per wave of k_describe (tuning knob)
    const int v = e ? atoi(e) : 4;
    ctx->desc_k = (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) ? v : 4;
    ctx->desc_k_user = e != nullptr;
  }
  ctx->out_cap = 0;
  for (int l = 0; l < nlevels; l++) ctx->out_cap += std::max(ctx->quota[l] + 3, 4 * kMaxRoots);
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return ORBX_E_DEVICE;
  }
  if (hipMalloc((void**)&ctx->d_qt_fin, kSmallBatchFrames * sizeof(int32_t)) != hipSuccess ||
      hipMemsetAsync(ctx->d_qt_fin, 0, kSmallBatchFrames * sizeof(int32_t), ctx->stream) != hipSuccess ||   // never the legacy stream (orbx_internal.h):
      hipStreamSynchronize(ctx->stream) != hipSuccess) {                                                        // contexts are created lazily, per thread, at any time
    (void)hipGetLastError();
    if (ctx->d_qt_fin) (void)hipFree(ctx->d_qt_fin);
    ctx->d_qt_fin = nullptr;   // the separate launches serve instead
  }
  {
    const char* e = getenv("ORBX_STREAMS");  // concurrent sub-batches of the batch entry point (1..orbx_ctx::kMaxAux)
    const int v = e ? atoi(e) : 1;
    ctx->nstreams = std::min(std::max(v, 1), 2);  // the blur fork owns the last two aux streams
    const char* gr = getenv("ORBX_GRAPH");   // 0 disables the replayed-graph single-frame path
    ctx->use_graph = gr ? atoi(gr) != 0 : true;
    const char* fb = getenv("ORBX_FORK_BLUR");
    ctx->fork_blur = fb ? atoi(fb) != 0 : true;
    const char* ff = getenv("ORBX_FORK_FAST0");
    { const char* e = getenv("ORBX_GAUSS_KERNEL"); const int v = e ? atoi(e) : 0; ctx->gauss_kernel = v == 1 ? 1 : 0; }
    { const char* e = getenv("ORBX_GAUSS_ROUND"); const int v = e ? atoi(e) : 0; ctx->gauss_round = (v >= 0 && v <= 2) ? v : 0; }
    { const char* e = getenv("ORBX_GAUSS_TAIL"); const int v = e ? atoi(e) : 0; ctx->gauss_tail = (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) ? v : 0; }
    { const char* e = getenv("ORBX_ATAN_FMA"); ctx->atan_fma = e && atoi(e) == 1 ? 1 : 0; }
    { const char* e = getenv("ORBX_BRIEF_FMA"); ctx->brief_fma = e && atoi(e) == 1 ? 1 : 0; }
    const char* fpk = getenv("ORBX_FAST_PK");   // packed 16-bit necessary test in k_fast_cells (128-thread workgroups)
    ctx->fast_pk = fpk ? atoi(fpk) != 0 : true;
    { const char* e = getenv("ORBX_REALIGN"); ctx->realign = e ? atoi(e) != 0 : true; }
    { const char* e = getenv("ORBX_FAST_STAGE_DMA"); ctx->fast_stage_dma = e ? atoi(e) != 0 : true; }
    { const char* e = getenv("ORBX_FAST_PASSES"); ctx->fast_passes = e && atoi(e) == 1 ? 1 : 2; }
    const char* dl = getenv("ORBX_DESC_LDS");   // blurred 37x37 window staged in LDS for the descriptor taps
    ctx->desc_lds = dl ? atoi(dl) != 0 : true;
    { const char* e = getenv("ORBX_DESC_FUSED_BLUR"); ctx->desc_fused_blur = e ? std::max(-1, std::min(1, atoi(e))) : -1; }
    const char* fq = getenv("ORBX_FORK_QT");
    ctx->fork_qt = fq ? atoi(fq) != 0 : true;
    ctx->fork_fast0 = ff ? atoi(ff) != 0 : false;  // measured: no gain (both kernels already fill the CUs), kept as a knob
    bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; i++)
      ok = hipEventCreateWithFlags(&ctx->ev_blur_fork[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_blur_join[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_qt_fork[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_qt_join[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_f0_fork[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_f0_join[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < orbx_ctx::kMaxAux && ok; i++)
      ok = hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { orbx_destroy(ctx); return ORBX_E_DEVICE; }
  }
  *out = ctx;
  return ORBX_OK;
}

void orbx_destroy(orbx_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_buffers(ctx);
  auto fr = [](auto*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } };
  fr(ctx->d_stage_img); fr(ctx->d_stage_out); fr(ctx->d_knn_ws); fr(ctx->d_realign);
  ctx->arena.release();
  if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
  if (ctx->h_in) { (void)hipHostFree(ctx->h_in); ctx->h_in = nullptr; }
  if (ctx->h_call) { (void)hipHostFree(ctx->h_call); ctx->h_call = nullptr; }
  for (int i = 0; i < 2; i++) if (ctx->h_view[i]) { (void)hipHostFree(ctx->h_view[i]); ctx->h_view[i] = nullptr; }
  unpublish_context(ctx);
  if (ctx->d_win_ctr) { (void)hipFree(ctx->d_win_ctr); ctx->d_win_ctr = nullptr; }
  if (ctx->d_qt_fin) { (void)hipFree(ctx->d_qt_fin); ctx->d_qt_fin = nullptr; }
  if (ctx->ev_g0) { (void)hipEventDestroy(ctx->ev_g0); ctx->ev_g0 = nullptr; }
  if (ctx->ev_g1) { (void)hipEventDestroy(ctx->ev_g1); ctx->ev_g1 = nullptr; }
  if (ctx->ev_w0) { (void)hipEventDestroy(ctx->ev_w0); ctx->ev_w0 = nullptr; }
  if (ctx->ev_w1) { (void)hipEventDestroy(ctx->ev_w1); ctx->ev_w1 = nullptr; }
  if (ctx->h_tgt) { (void)hipHostFree(ctx->h_tgt); ctx->h_tgt = nullptr; }
  if (ctx->ev_tgt) { (void)hipEventDestroy(ctx->ev_tgt); ctx->ev_tgt = nullptr; }
  if (ctx->d_color) { (void)hipFree(ctx->d_color); ctx->d_color = nullptr; }
  if (ctx->d_ingest_tab) { (void)hipFree(ctx->d_ingest_tab); ctx->d_ingest_tab = nullptr; }
  if (ctx->h_stage_out) { (void)hipHostFree(ctx->h_stage_out); ctx->h_stage_out = nullptr; }
  if (ctx->h_pyr) { (void)hipHostFree(ctx->h_pyr); ctx->h_pyr = nullptr; }
  for (int i = 0; i < orbx_ctx::kMaxAux; i++) {
    if (ctx->aux[i]) { (void)hipStreamSynchronize(ctx->aux[i]); (void)hipStreamDestroy(ctx->aux[i]); }
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  for (int i = 0; i < 2; i++) {
    if (ctx->ev_blur_fork[i]) (void)hipEventDestroy(ctx->ev_blur_fork[i]);
    if (ctx->ev_blur_join[i]) (void)hipEventDestroy(ctx->ev_blur_join[i]);
    if (ctx->ev_f0_fork[i]) (void)hipEventDestroy(ctx->ev_f0_fork[i]);
    if (ctx->ev_qt_fork[i]) (void)hipEventDestroy(ctx->ev_qt_fork[i]);
    if (ctx->ev_qt_join[i]) (void)hipEventDestroy(ctx->ev_qt_join[i]);
    if (ctx->ev_f0_join[i]) (void)hipEventDestroy(ctx->ev_f0_join[i]);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* orbx_last_error(const orbx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int orbx_keypoint_capacity(const orbx_ctx* ctx) { return ctx ? ctx->out_cap : ORBX_E_INVALID; }
int orbx_levels(const orbx_ctx* ctx) { return ctx ? ctx->nlevels : ORBX_E_INVALID; }

int orbx_scale_tables(const orbx_ctx* ctx, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                      int32_t* features_per_level) {
  if (!ctx) return ORBX_E_INVALID;
  for (int i = 0; i < ctx->nlevels; i++) {
    if (scale) scale[i] = ctx->scale[i];
    if (inv_scale) inv_scale[i] = ctx->inv_scale[i];
    if (sigma2) sigma2[i] = ctx->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = ctx->inv_sigma2[i];
    if (features_per_level) features_per_level[i] = ctx->quota[i];
  }
  return ORBX_OK;
}

int orbx_extract_batch_device(orbx_ctx* ctx, const uint8_t* d_imgs, int nframes, int rows, int cols, size_t row_stride,
                              size_t frame_stride, int lap0, int lap1, orbx_keypoint* d_kps, uint8_t* d_desc,
                              int32_t* d_counts, void* stream) {
  if (!ctx) return ORBX_E_INVALID;
  if (!d_imgs || rows <= 0 || cols <= 0 || nframes <= 0) return set_err(ctx, ORBX_E_EMPTY, "empty image");
  if (!d_kps || !d_desc || !d_counts || row_stride < (size_t)cols || nframes > 65535)
    return set_err(ctx, ORBX_E_INVALID, "bad batch arguments");
  if (row_stride >= (1u << 23) || (unsigned long long)row_stride * (unsigned long long)rows >= (1ull << 31))
    return set_err(ctx, ORBX_E_INVALID, "row stride / frame size beyond the kernels' 32-bit in-frame offsets");
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_buffers(ctx, rows, cols, nframes);
  if (rc == ORBX_OK) rc = ensure_blur(ctx, nframes);
  if (rc != ORBX_OK) return rc;
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  ctx->last_ext_stream = stream ? (hipStream_t)stream : nullptr;
  // frames whose rows are not dword-aligned take one pass into an aligned copy ("realign" option; k_realign_rows says why); level 0 of this
  // extraction — what the stereo pass, the debug readers and the pyramid accessors see — is then that copy
  if (ctx->realign && nframes > 1 && ((row_stride & 3) != 0 || (frame_stride & 3) != 0 || ((uintptr_t)d_imgs & 3) != 0)) {
    const size_t ap = (size_t)round_up(cols, 64), need = (size_t)nframes * rows * ap;
    if (need > ctx->realign_bytes) {
      ORBX_HIP(ctx, sync_ctx(ctx));
      if (ctx->d_realign) (void)hipFree(ctx->d_realign);
      ctx->d_realign = nullptr; ctx->realign_bytes = 0;
      ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_realign, need));
      ctx->realign_bytes = need;
    }
    hipLaunchKernelGGL(k_realign_rows, dim3((unsigned)((ap / 4 + 255) / 256), (unsigned)rows, (unsigned)nframes), dim3(256), 0, st, d_imgs,
                       (long long)row_stride, (long long)frame_stride, ctx->d_realign, (int)ap, rows, cols);
    d_imgs = ctx->d_realign; row_stride = ap; frame_stride = (size_t)rows * ap;
  }
  ctx->last_imgs = d_imgs; ctx->last_row_stride = row_stride; ctx->last_frame_stride = frame_stride;
  ctx->last_nframes = nframes;
  // Sub-batches on concurrent streams: frames are independent, and the pipeline alternates VALU-bound kernels
  // (FAST, blur) with latency-bound ones (quadtree, descriptors, the pyramid chain), so two or more sub-batches in
  // flight keep the CUs busy across kernel boundaries.  Fork/join with events on the caller's stream (capturable).
  int ns = ctx->profiling ? 1 : std::min(ctx->nstreams, std::max(1, nframes / 32));
  if (ns <= 1)
    return launch_pipeline(ctx, d_imgs, 0, nframes, rows, cols, row_stride, frame_stride, lap0, lap1, d_kps, d_desc, d_counts, st);
  ORBX_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
  const int per = (nframes + ns - 1) / ns;
  for (int i = 0; i < ns; i++) {
    const int f0 = i * per, nf = std::min(per, nframes - f0);
    if (nf <= 0) break;
    hipStream_t s = ctx->aux[i];
    ORBX_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_fork, 0));
    rc = launch_pipeline(ctx, d_imgs, f0, nf, rows, cols, row_stride, frame_stride, lap0, lap1, d_kps, d_desc, d_counts, s);
    // the sub-batch's stream is joined back into the caller's stream whether or not its launches succeeded
    const hipError_t e1 = hipEventRecord(ctx->ev_join[i], s), e2 = hipStreamWaitEvent(st, ctx->ev_join[i], 0);
    if (rc != ORBX_OK) return rc;
    ORBX_HIP(ctx, e1);
    ORBX_HIP(ctx, e2);
  }
  return ORBX_OK;
}

// Staging of the host-buffer entry points: one device block [keypoints | descriptors | counts] per call and a pinned
// host mirror of it, so the results come back in ONE device-to-host copy.
struct StageLayout { size_t kps_off, desc_off, counts_off, bow_off, bytes; };
static StageLayout stage_layout(const orbx_ctx* ctx, int nframes) {
  StageLayout L;
  L.kps_off = 0;
  L.desc_off = ((size_t)nframes * ctx->out_cap * sizeof(orbx_keypoint) + 255) / 256 * 256;
  L.counts_off = L.desc_off + ((size_t)nframes * ctx->out_cap * 32 + 255) / 256 * 256;
  L.bow_off = L.counts_off + ((size_t)nframes * 2 * sizeof(int32_t) + 255) / 256 * 256;   // {word, node, weight} records of frame 0 (single-frame graph)
  L.bytes = L.bow_off + ((size_t)ctx->out_cap * 16 + 255) / 256 * 256;
  return L;
}

static int ensure_stage(orbx_ctx* ctx, int nframes, size_t img_bytes) {
  if (img_bytes > ctx->stage_img_bytes) {
    if (ctx->d_stage_img) (void)hipFree(ctx->d_stage_img);
    ctx->d_stage_img = nullptr; ctx->stage_img_bytes = 0;
    ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_stage_img, img_bytes));
    ctx->stage_img_bytes = img_bytes;
    ctx->buf_epoch++;
  }
  if (nframes > ctx->stage_frames) {
    if (ctx->d_stage_out) (void)hipFree(ctx->d_stage_out);
    if (ctx->h_stage_out) (void)hipHostFree(ctx->h_stage_out);
    ctx->d_stage_out = nullptr; ctx->h_stage_out = nullptr; ctx->stage_frames = 0;
    const StageLayout L = stage_layout(ctx, nframes);
    ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_stage_out, L.bytes));
    ORBX_HIP(ctx, hipHostMalloc((void**)&ctx->h_stage_out, L.bytes, hipHostMallocMapped));
    ctx->stage_frames = nframes;
    ctx->buf_epoch++;
  }
  return ORBX_OK;
}

// The single-frame path as a replayed graph.  Returns 1 when it handled the frame, 0 when the caller must take the
// ordinary path (graphs disabled or capture failed), < 0 on a real error.
static int extract_one_graph(orbx_ctx* ctx, const uint8_t* img, int rows, int cols, size_t row_stride, int lap0, int lap1,
                             size_t pitch, size_t fbytes, const StageLayout& L) {
  if (!ctx->use_graph || ctx->profiling) return 0;
  hipStream_t st = ctx->stream;
  const size_t in_bytes = (size_t)rows * cols;
  if (in_bytes > ctx->h_in_bytes) {
    if (ctx->h_in) (void)hipHostFree(ctx->h_in);
    ctx->h_in = nullptr; ctx->h_in_bytes = 0;
    if (hipHostMalloc((void**)&ctx->h_in, in_bytes, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); ctx->use_graph = false; return 0; }
    ctx->h_in_bytes = in_bytes;
    ctx->buf_epoch++;
  }
  const bool keep = ctx->keep_host_pyr;
  int rc = ensure_buffers(ctx, rows, cols, 1);
  if (rc == ORBX_OK) rc = ensure_blur(ctx, 1);
  if (rc != ORBX_OK) return rc;
  if (keep && ctx->geo.pyr_bytes > 0 && (size_t)ctx->geo.pyr_bytes > ctx->h_pyr_bytes) {
    if (ctx->h_pyr) (void)hipHostFree(ctx->h_pyr);
    ctx->h_pyr = nullptr; ctx->h_pyr_bytes = 0;
    ORBX_HIP(ctx, hipHostMalloc((void**)&ctx->h_pyr, (size_t)ctx->geo.pyr_bytes, hipHostMallocDefault));
    ctx->h_pyr_bytes = (size_t)ctx->geo.pyr_bytes;
    ctx->buf_epoch++;
  }
  const int key[6] = {rows, cols, lap0, lap1, (keep ? 1 : 0) | (ctx->stage_frames == 1 ? 2 : 0), ctx->buf_epoch};
  if (!ctx->graph_exec || std::memcmp(key, ctx->graph_key, sizeof(key)) != 0) {
    if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
    uint8_t* d = ctx->d_stage_out;
    hipGraph_t graph = nullptr;
    bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess;
    if (ok) {
      // the upload: a kernel pulling the rows from the mapped pinned buffer with 16-byte loads (k_upload_rows) where the shape allows,
      // otherwise a copy node
      uint8_t* hdev = nullptr;
      static const int upload_kernel = getenv("ORBX_UPLOAD_KERNEL") ? atoi(getenv("ORBX_UPLOAD_KERNEL")) : 1;   // A/B knob
      if (upload_kernel && (cols & 15) == 0 && hipHostGetDevicePointer((void**)&hdev, ctx->h_in, 0) == hipSuccess && hdev) {
        const int row16 = cols / 16, n = rows * row16;
        hipLaunchKernelGGL(k_upload_rows, dim3(std::min((n + 255) / 256, 512)), dim3(256), 0, st, (const uint4*)hdev, ctx->d_stage_img, rows, row16, (int)pitch);
        ok = hipGetLastError() == hipSuccess;
      } else {
        (void)hipGetLastError();
        ok = hipMemcpy2DAsync(ctx->d_stage_img, pitch, ctx->h_in, (size_t)cols, (size_t)cols, (size_t)rows, hipMemcpyHostToDevice, st) == hipSuccess;
      }
      // results: the last kernels write them into the pinned block themselves (mapped view) where the fused single-frame launches are
      // in use; otherwise ONE copy node (every node of the graph costs about 5 us of device time, whatever it moves)
      ResultMirror mir;
      uint8_t* hout_dev = nullptr;
      static const bool mirror_on = !(getenv("ORBX_MIRROR") && atoi(getenv("ORBX_MIRROR")) == 0);   // A/B knob
      if (mirror_on && hipHostGetDevicePointer((void**)&hout_dev, ctx->h_stage_out, 0) == hipSuccess && hout_dev) {
        mir.kps = (orbx_keypoint*)(hout_dev + L.kps_off); mir.desc = hout_dev + L.desc_off; mir.counts = (int32_t*)(hout_dev + L.counts_off);
      } else (void)hipGetLastError();
      bool mirrored = false;
      if (ok) ok = launch_pipeline(ctx, ctx->d_stage_img, 0, 1, rows, cols, pitch, fbytes, lap0, lap1, (orbx_keypoint*)(d + L.kps_off),
                                   d + L.desc_off, (int32_t*)(d + L.counts_off), st, &mir, &mirrored) == ORBX_OK;
      const size_t kb = (size_t)ctx->out_cap * sizeof(orbx_keypoint), db = (size_t)ctx->out_cap * 32, cb = 2 * sizeof(int32_t);
      if (mirrored) {
      } else if (ctx->stage_frames == 1) {
        if (ok) ok = hipMemcpyAsync(ctx->h_stage_out, d, L.counts_off + cb, hipMemcpyDeviceToHost, st) == hipSuccess;
      } else {   // the staging block was laid out for an earlier, larger batch: this frame's three ranges
        if (ok) ok = hipMemcpyAsync(ctx->h_stage_out + L.kps_off, d + L.kps_off, kb, hipMemcpyDeviceToHost, st) == hipSuccess;
        if (ok) ok = hipMemcpyAsync(ctx->h_stage_out + L.desc_off, d + L.desc_off, db, hipMemcpyDeviceToHost, st) == hipSuccess;
        if (ok) ok = hipMemcpyAsync(ctx->h_stage_out + L.counts_off, d + L.counts_off, cb, hipMemcpyDeviceToHost, st) == hipSuccess;
      }
      // a vocabulary is attached: the descent of the frame's descriptors as the graph's tail, records into the pinned block
      ctx->bow_in_graph = false;
      if (ok && ctx->bow_active && hout_dev && voc_device(ctx->bow_active) == ctx->device) {
        ok = launch_bow_records(ctx->bow_active, d + L.desc_off, (const int32_t*)(d + L.counts_off), ctx->out_cap, ctx->bow_active_levelsup, hout_dev + L.bow_off, st) == hipSuccess;
        ctx->bow_in_graph = ok;
      }
      if (ok && keep && ctx->geo.pyr_bytes > 0)
        ok = hipMemcpyAsync(ctx->h_pyr, ctx->d_pyr, (size_t)ctx->geo.pyr_bytes, hipMemcpyDeviceToHost, st) == hipSuccess;
      const hipError_t ee = hipStreamEndCapture(st, &graph);   // always leave capture mode
      ok = ok && ee == hipSuccess && graph != nullptr;
    }
    if (ok) ok = hipGraphInstantiate(&ctx->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
      (void)hipGetLastError();
      if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
      ctx->use_graph = false;   // this runtime cannot capture the pipeline: fall back to plain launches for good
      return 0;
    }
    std::memcpy(ctx->graph_key, key, sizeof(key));
  }
  static const bool trace = getenv("ORBX_TRACE_EXTRACT") != nullptr;   // phase times of every call on stderr (diagnostics)
  const auto tr0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count(); };
  if (row_stride == (size_t)cols) std::memcpy(ctx->h_in, img, in_bytes);
  else for (int r = 0; r < rows; r++) std::memcpy(ctx->h_in + (size_t)r * cols, img + (size_t)r * row_stride, (size_t)cols);
  const double us_in = since();
  ctx->last_imgs = ctx->d_stage_img; ctx->last_row_stride = pitch; ctx->last_frame_stride = fbytes; ctx->last_nframes = 1;
  ctx->h_pyr_valid = false;
  if (ctx->graph_timing) {
    if (!ctx->ev_g0) { ORBX_HIP(ctx, hipEventCreate(&ctx->ev_g0)); ORBX_HIP(ctx, hipEventCreate(&ctx->ev_g1)); }
    ORBX_HIP(ctx, hipEventRecord(ctx->ev_g0, st));
  }
  ORBX_HIP(ctx, hipGraphLaunch(ctx->graph_exec, st));
  if (ctx->graph_timing) ORBX_HIP(ctx, hipEventRecord(ctx->ev_g1, st));
  const double us_launch = since();
  ORBX_HIP(ctx, hipStreamSynchronize(st));
  if (ctx->graph_timing) { float ms = 0.f; if (hipEventElapsedTime(&ms, ctx->ev_g0, ctx->ev_g1) == hipSuccess) ctx->last_graph_us = 1e3 * ms; else (void)hipGetLastError(); }
  if (trace) std::fprintf(stderr, "[orbx extract] %dx%d: image into the pinned buffer %.1f us, graph launch %.1f, wait %.1f\n", cols, rows, us_in,
                          us_launch - us_in, since() - us_launch);
  ctx->h_pyr_valid = keep && ctx->geo.pyr_bytes > 0;
  return 1;
}

// cv::resize(im, resizedIm, settings_->newImSize()) of System::TrackMonocular / TrackStereo / TrackRGBD (src/System.cc:441-446)
// fused behind the upload: the caller's grey image lands in d_color, k_resize (the pyramid's INTER_LINEAR kernel, with tables
// for this source / destination size) or k_box2 (OpenCV's INTER_AREA shortcut for an exact 2 x 2 downscale) writes level 0.
static int ingest_resized(orbx_ctx* ctx, const uint8_t* img, int src_rows, int src_cols, size_t row_stride, int rows, int cols, size_t pitch) {
  if (src_rows > 65535 || src_cols > 65535) return set_err(ctx, ORBX_E_INVALID, "orbx_extract_resized: source image side beyond 65535 px");
  const size_t spitch = (size_t)round_up(src_cols, 64), sbytes = spitch * src_rows;
  if (spitch >= (1u << 23)) return set_err(ctx, ORBX_E_INVALID, "orbx_extract_resized: source row beyond the kernels' offsets");
  if (sbytes > ctx->color_bytes) {
    ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_color) (void)hipFree(ctx->d_color);
    ctx->d_color = nullptr; ctx->color_bytes = 0;
    ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_color, sbytes));
    ctx->color_bytes = sbytes;
  }
  hipStream_t st = ctx->stream;
  ORBX_HIP(ctx, hipMemcpy2DAsync(ctx->d_color, spitch, img, row_stride, (size_t)src_cols, (size_t)src_rows, hipMemcpyHostToDevice, st));
  if (src_cols == 2 * cols && src_rows == 2 * rows) {
    hipLaunchKernelGGL(k_box2, dim3((cols + 63) / 64, (rows + 3) / 4), dim3(256), 0, st, ctx->d_color, (int)spitch, ctx->d_stage_img, (int)pitch, cols, rows);
    ORBX_HIP(ctx, hipGetLastError());
    return ORBX_OK;
  }
  const int key[4] = {src_rows, src_cols, rows, cols};
  if (!ctx->d_ingest_tab || std::memcmp(key, ctx->ingest_key, sizeof(key)) != 0) {
    std::vector<XTab> xt, yt;
    build_axis_table(src_cols, cols, true, xt);
    while (xt.size() % 4) xt.push_back(XTab{0, 0, 0, 0});   // the kernel reads the column table four entries at a time
    build_axis_table(src_rows, rows, false, yt);
    int mw = 1, mh = 1;   // LDS tile: the largest source rectangle any 64 x 64 output tile needs (as for the pyramid levels)
    for (int x0 = 0; x0 < cols; x0 += kRT_W) {
      const XTab a = xt[x0], b = xt[std::min(x0 + kRT_W, cols) - 1];
      mw = std::max(mw, std::max((int)b.s0, (int)b.s1) - ((int)a.s0 & ~3) + 1);
    }
    for (int y0 = 0; y0 < rows; y0 += kRT_H) {
      const XTab a = yt[y0], b = yt[std::min(y0 + kRT_H, rows) - 1];
      mh = std::max(mh, std::max((int)b.s0, (int)b.s1) - (int)a.s0 + 1);
    }
    const int lp = round_up(mw, 4);
    if ((size_t)lp * mh > 60 * 1024 || lp / 4 > 256) return set_err(ctx, ORBX_E_CAPACITY, "orbx_extract_resized: scale too large for the resize kernel's LDS tile");
    ORBX_HIP(ctx, hipStreamSynchronize(st));
    if (ctx->d_ingest_tab) (void)hipFree(ctx->d_ingest_tab);
    ctx->d_ingest_tab = nullptr;
    ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_ingest_tab, sizeof(XTab) * (xt.size() + yt.size())));
    ORBX_HIP(ctx, copy_sync(ctx, ctx->d_ingest_tab, xt.data(), sizeof(XTab) * xt.size(), hipMemcpyHostToDevice));
    ORBX_HIP(ctx, copy_sync(ctx, ctx->d_ingest_tab + xt.size(), yt.data(), sizeof(XTab) * yt.size(), hipMemcpyHostToDevice));
    std::memcpy(ctx->ingest_key, key, sizeof(key));
    ctx->ingest_ytab_off = (int)xt.size(); ctx->ingest_lds_pitch = lp; ctx->ingest_lds_rows = mh;
  }
  const int nbx = (cols + kRT_W - 1) / kRT_W, nby = (rows + kRT_H - 1) / kRT_H, nitems = nbx * nby;
  const int lp = ctx->ingest_lds_pitch, lr = ctx->ingest_lds_rows;
  hipLaunchKernelGGL(k_resize, dim3(xcd_grid(nitems)), dim3(256), (size_t)lp * lr, st, ctx->d_color, (long long)sbytes, (int)spitch, src_cols,
                     ctx->d_stage_img, (long long)(pitch * rows), (int)pitch, cols, rows, ctx->d_ingest_tab, ctx->d_ingest_tab + ctx->ingest_ytab_off,
                     nbx, nby, nitems, lp, lr, div_magic((uint32_t)(nbx * nby)), div_magic((uint32_t)nbx), div_magic((uint32_t)(lp / 4)),
                     256 / (lp / 4));
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

int orbx_publish_descriptors(orbx_ctx* ctx, const void* host_desc, int n) {
  if (!ctx || !host_desc || n < 0) return ORBX_E_INVALID;
  const uint64_t dig = rows_digest((const uint8_t*)host_desc, n);   // the caller's buffer as it is NOW (outside the lock: host memory only)
  std::lock_guard<std::mutex> lock(g_pub_mu);
  for (orbx_ctx* c : g_publishers) if (c->rows_host == host_desc) { c->rows_host = nullptr; c->rows_n = 0; }   // a reused buffer: the old entry dies
  ctx->rows_host = nullptr; ctx->rows_n = 0;                      // a context names the buffer of its last extraction once
  if (n != ctx->last_n0 || n == 0) return ORBX_OK;                // not what this context last extracted: nothing to look up later
  ctx->rows_host = host_desc; ctx->rows_n = n; ctx->rows_digest_v = dig;
  if (std::find(g_publishers.begin(), g_publishers.end(), ctx) == g_publishers.end()) g_publishers.push_back(ctx);
  return ORBX_OK;
}

int orbx_bow_transform_published(orbx_voc* voc, const void* host_desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node) {
  if (!voc || !host_desc || n <= 0 || !word || !weight || !node) return ORBX_E_INVALID;
  const uint64_t dig = rows_digest((const uint8_t*)host_desc, n);
  std::lock_guard<std::mutex> lock(g_pub_mu);
  for (orbx_ctx* c : g_publishers)
    if (c->rows_host == host_desc && c->rows_n == n && c->rows_digest_v == dig) {
      if (c->bow_active == voc && c->bow_active_levelsup == levelsup && c->bow_seq == c->extract_seq && c->h_stage_out) {
        struct Rec { uint32_t word, node; double weight; };
        const Rec* rec = (const Rec*)(c->h_stage_out + c->bow_off);   // the extracting thread cannot overwrite the block while g_pub_mu is held
        for (int i = 0; i < n; i++) { word[i] = rec[i].word; node[i] = rec[i].node; weight[i] = rec[i].weight; }
        return ORBX_OK;
      }
      // this extractor's frames are being transformed with `voc`: from its next extraction on the graph does the descent itself
      static const bool attach = !(getenv("ORBX_BOW_IN_GRAPH") && atoi(getenv("ORBX_BOW_IN_GRAPH")) == 0);   // A/B knob
      if (attach && voc_device(voc) == c->device && (c->bow_voc != voc || c->bow_levelsup != levelsup)) {
        c->bow_voc = voc; c->bow_levelsup = levelsup;
        if (std::find(g_bow_attached.begin(), g_bow_attached.end(), c) == g_bow_attached.end()) g_bow_attached.push_back(c);
      }
      return 1;
    }
  return 1;
}

// channels == 1: grey frames.  channels == 3 / 4: interleaved colour frames, converted on the device behind the upload
// (rgb_order != 0: R first, else B first).
static int extract_batch_impl(orbx_ctx* ctx, const uint8_t* imgs, int nframes, int rows, int cols, size_t row_stride,
                              size_t frame_stride, int channels, int rgb_order, int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc,
                              int32_t* counts, int src_rows = 0, int src_cols = 0) {
  if (!ctx) return ORBX_E_INVALID;
  if (!imgs || rows <= 0 || cols <= 0 || nframes <= 0) return set_err(ctx, ORBX_E_EMPTY, "empty image");
  const bool resized = src_rows > 0;   // rows x cols = the size AFTER cv::resize; the caller's image is src_rows x src_cols
  if (!kps || !desc || !counts || row_stride < (size_t)(resized ? src_cols : cols) * channels) return set_err(ctx, ORBX_E_INVALID, "bad arguments");
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  ExtractionScope scope(ctx);   // what was published about the last extraction is void from here; ends on every path out
  const size_t pitch = (size_t)round_up(cols, 64), fbytes = pitch * rows;
  int rc = ensure_stage(ctx, nframes, fbytes * nframes);
  if (rc != ORBX_OK) return rc;
  const StageLayout L = stage_layout(ctx, ctx->stage_frames);   // the block was laid out for its allocated capacity
  const size_t kb = (size_t)nframes * ctx->out_cap * sizeof(orbx_keypoint), db = (size_t)nframes * ctx->out_cap * 32,
               cb = (size_t)nframes * 2 * sizeof(int32_t);
  if (nframes == 1 && channels == 1 && !resized) {   // the live-SLAM path: replay the captured graph
    if (row_stride >= (1u << 23) || (unsigned long long)row_stride * (unsigned long long)rows >= (1ull << 31))
      return set_err(ctx, ORBX_E_INVALID, "row stride / frame size beyond the kernels' 32-bit in-frame offsets");
    rc = extract_one_graph(ctx, imgs, rows, cols, row_stride, lap0, lap1, pitch, fbytes, L);
    if (rc < 0) return rc;
    if (rc == 1) {
      std::memcpy(kps, ctx->h_stage_out + L.kps_off, kb);
      std::memcpy(desc, ctx->h_stage_out + L.desc_off, db);
      std::memcpy(counts, ctx->h_stage_out + L.counts_off, cb);
      if (counts[0] < 0) return set_err(ctx, ORBX_E_CAPACITY, "quadtree produced more nodes than the level capacity");
      ctx->bow_off = L.bow_off;
      scope.n0 = counts[0]; scope.bow = ctx->bow_in_graph;
      return ORBX_OK;
    }
  }
  if (resized) {
    rc = ingest_resized(ctx, imgs, src_rows, src_cols, row_stride, rows, cols, pitch);
    if (rc != ORBX_OK) return rc;
  } else if (channels == 1) {
    for (int f = 0; f < nframes; f++)
      ORBX_HIP(ctx, hipMemcpy2DAsync(ctx->d_stage_img + f * fbytes, pitch, imgs + f * frame_stride, row_stride, cols, rows,
                                     hipMemcpyHostToDevice, ctx->stream));
  } else {
    const size_t cpitch = (size_t)round_up(cols * channels, 64), cbytes = cpitch * rows;
    if (cpitch >= (1u << 23)) return set_err(ctx, ORBX_E_INVALID, "colour row beyond the kernels' offsets");
    if (cbytes * nframes > ctx->color_bytes) {
      ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->d_color) (void)hipFree(ctx->d_color);
      ctx->d_color = nullptr; ctx->color_bytes = 0;
      ORBX_HIP(ctx, hipMalloc((void**)&ctx->d_color, cbytes * nframes));
      ctx->color_bytes = cbytes * nframes;
    }
    for (int f = 0; f < nframes; f++)
      ORBX_HIP(ctx, hipMemcpy2DAsync(ctx->d_color + f * cbytes, cpitch, imgs + f * frame_stride, row_stride, (size_t)cols * channels, rows,
                                     hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_color_to_gray, dim3((cols + 255) / 256, (rows + 3) / 4, nframes), dim3(256), 0, ctx->stream, ctx->d_color,
                       (long long)cbytes, (int)cpitch, channels, rgb_order ? 0 : 2, rgb_order ? 2 : 0, ctx->d_stage_img, (long long)fbytes,
                       (int)pitch, rows, cols);
    ORBX_HIP(ctx, hipGetLastError());
  }
  uint8_t* d = ctx->d_stage_out;
  rc = orbx_extract_batch_device(ctx, ctx->d_stage_img, nframes, rows, cols, pitch, fbytes, lap0, lap1,
                                 (orbx_keypoint*)(d + L.kps_off), d + L.desc_off, (int32_t*)(d + L.counts_off), ctx->stream);
  if (rc != ORBX_OK) return rc;
  // one copy for everything the caller gets back (sized by what this call produced)
  if (nframes == ctx->stage_frames) {
    ORBX_HIP(ctx, hipMemcpyAsync(ctx->h_stage_out, d, L.counts_off + cb, hipMemcpyDeviceToHost, ctx->stream));
  } else {  // a smaller batch than the block was laid out for: three ranges
    ORBX_HIP(ctx, hipMemcpyAsync(ctx->h_stage_out + L.kps_off, d + L.kps_off, kb, hipMemcpyDeviceToHost, ctx->stream));
    ORBX_HIP(ctx, hipMemcpyAsync(ctx->h_stage_out + L.desc_off, d + L.desc_off, db, hipMemcpyDeviceToHost, ctx->stream));
    ORBX_HIP(ctx, hipMemcpyAsync(ctx->h_stage_out + L.counts_off, d + L.counts_off, cb, hipMemcpyDeviceToHost, ctx->stream));
  }
  ctx->h_pyr_valid = false;
  if (ctx->keep_host_pyr && ctx->geo.pyr_bytes > 0) {  // frame 0's levels >= 1 in ONE pinned copy, overlapped with the rest
    const size_t need = (size_t)ctx->geo.pyr_bytes;
    if (need > ctx->h_pyr_bytes) {
      if (ctx->h_pyr) (void)hipHostFree(ctx->h_pyr);
      ctx->h_pyr = nullptr; ctx->h_pyr_bytes = 0;
      ORBX_HIP(ctx, hipHostMalloc((void**)&ctx->h_pyr, need, hipHostMallocDefault));
      ctx->h_pyr_bytes = need;
      ctx->buf_epoch++;
    }
    ORBX_HIP(ctx, hipMemcpyAsync(ctx->h_pyr, ctx->d_pyr, need, hipMemcpyDeviceToHost, ctx->stream));
    ctx->h_pyr_valid = true;
  }
  ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::memcpy(kps, ctx->h_stage_out + L.kps_off, kb);
  std::memcpy(desc, ctx->h_stage_out + L.desc_off, db);
  std::memcpy(counts, ctx->h_stage_out + L.counts_off, cb);
  // k_assemble reports a quadtree capacity overflow (never expected) as a negative keypoint count: fail loudly
  for (int f = 0; f < nframes; f++)
    if (counts[2 * f] < 0) return set_err(ctx, ORBX_E_CAPACITY, "quadtree produced more nodes than the level capacity");
  scope.n0 = counts[0];
  return ORBX_OK;
}

int orbx_extract_batch(orbx_ctx* ctx, const uint8_t* imgs, int nframes, int rows, int cols, size_t row_stride,
                       size_t frame_stride, int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc, int32_t* counts) {
  return extract_batch_impl(ctx, imgs, nframes, rows, cols, row_stride, frame_stride, 1, 0, lap0, lap1, kps, desc, counts);
}

int orbx_extract_color(orbx_ctx* ctx, const uint8_t* img, int rows, int cols, size_t stride, int channels, int rgb_order, int lap0,
                       int lap1, orbx_keypoint* kps, uint8_t* desc, int* n_out, int* mono_index_out) {
  if (n_out) *n_out = 0;
  if (mono_index_out) *mono_index_out = 0;
  if (!ctx) return ORBX_E_INVALID;
  if (channels != 3 && channels != 4) return set_err(ctx, ORBX_E_INVALID, "orbx_extract_color: 3 or 4 interleaved channels expected");
  if (!img || rows <= 0 || cols <= 0) return set_err(ctx, ORBX_E_EMPTY, "empty image");
  int32_t counts[2] = {0, 0};
  int rc = extract_batch_impl(ctx, img, 1, rows, cols, stride, stride * rows, channels, rgb_order, lap0, lap1, kps, desc, counts);
  if (rc != ORBX_OK) return rc;
  if (n_out) *n_out = counts[0];
  if (mono_index_out) *mono_index_out = counts[1];
  return ORBX_OK;
}

int orbx_extract_resized(orbx_ctx* ctx, const uint8_t* img, int rows, int cols, size_t stride, int new_rows, int new_cols, int lap0, int lap1,
                         orbx_keypoint* kps, uint8_t* desc, int* n_out, int* mono_index_out) {
  if (n_out) *n_out = 0;
  if (mono_index_out) *mono_index_out = 0;
  if (!ctx) return ORBX_E_INVALID;
  if (!img || rows <= 0 || cols <= 0 || new_rows <= 0 || new_cols <= 0) return set_err(ctx, ORBX_E_EMPTY, "empty image");
  if (new_rows == rows && new_cols == cols) return orbx_extract(ctx, img, rows, cols, stride, lap0, lap1, kps, desc, n_out, mono_index_out);
  int32_t counts[2] = {0, 0};
  int rc = extract_batch_impl(ctx, img, 1, new_rows, new_cols, stride, stride * rows, 1, 0, lap0, lap1, kps, desc, counts, rows, cols);
  if (rc != ORBX_OK) return rc;
  if (n_out) *n_out = counts[0];
  if (mono_index_out) *mono_index_out = counts[1];
  return ORBX_OK;
}

int orbx_extract(orbx_ctx* ctx, const uint8_t* img, int rows, int cols, size_t stride, int lap0, int lap1,
                 orbx_keypoint* kps, uint8_t* desc, int* n_out, int* mono_index_out) {
  if (n_out) *n_out = 0;
  if (mono_index_out) *mono_index_out = 0;
  if (!ctx) return ORBX_E_INVALID;
  if (!img || rows <= 0 || cols <= 0) return set_err(ctx, ORBX_E_EMPTY, "empty image");
  int32_t counts[2] = {0, 0};
  int rc = orbx_extract_batch(ctx, img, 1, rows, cols, stride, stride * rows, lap0, lap1, kps, desc, counts);
  if (rc != ORBX_OK) return rc;
  if (n_out) *n_out = counts[0];
  if (mono_index_out) *mono_index_out = counts[1];
  return ORBX_OK;
}

int orbx_pyramid_level(orbx_ctx* ctx, int frame, int level, uint8_t* dst, size_t dst_stride, int* w, int* h) {
  if (!ctx || !ctx->d_geo || level < 0 || level >= ctx->nlevels || frame < 0 || frame >= ctx->last_nframes)
    return ctx ? set_err(ctx, ORBX_E_INVALID, "no such pyramid level") : ORBX_E_INVALID;
  const LevelGeom& L = ctx->geo.lv[level];
  if (w) *w = L.w;
  if (h) *h = L.h;
  if (!dst) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  ORBX_HIP(ctx, sync_ctx(ctx));   // the pyramid may still be in flight on an aux stream or on the caller's stream of the last batch
  const uint8_t* src; size_t sp;
  if (level == 0) { src = ctx->last_imgs + (size_t)frame * ctx->last_frame_stride; sp = ctx->last_row_stride; }
  else { src = ctx->d_pyr + (size_t)frame * ctx->geo.pyr_bytes + L.plane_off; sp = L.pitch; }
  ORBX_HIP(ctx, copy2d_sync(ctx, dst, dst_stride, src, sp, L.w, L.h, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_reserve(orbx_ctx* ctx, int rows, int cols, int nframes) {
  if (!ctx || rows <= 0 || cols <= 0 || nframes <= 0 || nframes > 65535) return ctx ? set_err(ctx, ORBX_E_INVALID, "orbx_reserve: bad arguments") : ORBX_E_INVALID;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  const int rc = ensure_buffers(ctx, rows, cols, nframes);
  return rc == ORBX_OK ? ensure_blur(ctx, nframes) : rc;
}

int orbx_set_option(orbx_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return ORBX_E_INVALID;
  const std::string n(name);
  if (n == "fork_blur") ctx->fork_blur = value != 0;
  else if (n == "fork_fast0") ctx->fork_fast0 = value != 0;
  else if (n == "fork_qt") ctx->fork_qt = value != 0;
  else if (n == "graph") ctx->use_graph = value != 0;
  else if (n == "graph_timing") { ctx->graph_timing = value != 0; return ORBX_OK; }   // no re-capture needed
  else if (n == "window_timing") { ctx->window_timing = value != 0; return ORBX_OK; }   // HIP events around every resident-target window pass (orbx_last_window_device_us)
  else if (n == "fast_pk") ctx->fast_pk = value != 0;
  else if (n == "realign") ctx->realign = value != 0;   // batch frames with rows that are not dword-aligned: one pass into an aligned copy first
  else if (n == "fast_stage_dma") ctx->fast_stage_dma = value != 0;   // FAST tile staged by LDS-DMA loads instead of load + ds_write
  else if (n == "replay_alternate" && (value == 0 || value == 1)) { ctx->replay_alternate = value; return ORBX_OK; }   // read by orbx_replay_prepare from lane 0
  else if (n == "fast_passes" && (value == 1 || value == 2)) ctx->fast_passes = value;   // batch FAST: 2 = iniTh first, minTh where the cell stayed empty; 1 = one pass at minTh
  else if (n == "gauss_kernel" && (value == 0 || value == 1)) ctx->gauss_kernel = value;   // which OpenCV's 8-bit Gaussian weights (include/orbx.h)
  else if (n == "gauss_round" && value >= 0 && value <= 2) ctx->gauss_round = value;       // ... which rounding of the column pass
  else if (n == "gauss_tail" && (value == 0 || value == 4 || value == 8 || value == 16 || value == 32 || value == 64)) ctx->gauss_tail = value;   // ... and its scalar tail
  else if (n == "atan_fma" && (value == 0 || value == 1)) ctx->atan_fma = value;          // cv::fastAtan2 compiled with / without FMA contraction
  else if (n == "brief_fma" && (value == 0 || value == 1)) ctx->brief_fma = value;        // the reference's own pattern rotation compiled with / without it
  else if (n == "qt_points" && value >= 256 && value <= 4096 && value % 128 == 0) ctx->qt_points = value;   // LDS-resident candidates per (frame, level) of the big quadtree levels (half of it for the small ones)
  else if (n == "small_fused") ctx->small_fused = value != 0;
  else if (n == "qt_level_major") ctx->qt_level_major = value != 0;
  else if (n == "qt_fused") ctx->qt_fused = value != 0;   // the first (up to three) quadtree passes in one sweep over the points
  else if (n == "chain_batch") ctx->chain_batch = value != 0;
  else if (n == "chain_long") ctx->chain_long = value != 0;
  else if (n == "describe_direct") ctx->describe_direct = value != 0;
  else if (n == "chain_long_tile" && value >= 8 && value <= 64 && value % 4 == 0) ctx->chain_long_tile = value;
  else if (n == "chain_first" && value >= 2 && value <= 7) ctx->chain_first = value;
  else if (n == "chain_threads" && value >= 64 && value <= 1024 && value % 64 == 0) ctx->chain_threads = value;
  else if (n == "qt_big_levels" && value >= 1 && value <= 8) ctx->qt_big_levels = value;
  else if (n == "qt_threads_small" && (value == 0 || value == 64 || value == 128 || value == 192 || value == 256 || value == 512)) ctx->qt_threads_small = value;
  else if (n == "qt_one_launch") ctx->qt_one_launch = value != 0;
  else if (n == "window_direct") ctx->window_direct = value != 0;
  else if (n == "fast_split") ctx->fast_split = value != 0;   // FAST launched per group of levels with its own LDS size (batch calls)
  else if (n == "desc_lds") ctx->desc_lds = value != 0;
  else if (n == "desc_fused_blur" && value >= -1 && value <= 1) ctx->desc_fused_blur = value;
  else if (n == "fast_threads" && (value == 64 || value == 128 || value == 256)) ctx->fast_threads = value;
  else if (n == "qt_threads" && value >= 0 && value <= 512 && value % 64 == 0) ctx->qt_threads = value;   // 0: chosen by batch size
  else if (n == "desc_k" && (value == 1 || value == 2 || value == 4 || value == 8 || value == 16)) { ctx->desc_k = value; ctx->desc_k_user = true; }
  else if (n == "streams" && value >= 1 && value <= 2) ctx->nstreams = value;
  else if (n == "view_pool_cap" && value >= 16) { ctx->view_pool_cap = value; return ORBX_OK; }   // test hook: the learnt candidate-pool capacity of orbx_target_search_view
  else return set_err(ctx, ORBX_E_INVALID, "orbx_set_option: unknown option or value out of range: " + n);
  ctx->buf_epoch++;   // a captured single-frame graph holds the launch shape of the old options: capture again
  return ORBX_OK;
}

// the current value of an option (all are >= 0), ORBX_E_INVALID for an unknown name
int orbx_get_option(const orbx_ctx* ctx, const char* name) {
  if (!ctx || !name) return ORBX_E_INVALID;
  const std::string n(name);
  const struct { const char* name; int value; } tab[] = {
      {"fork_blur", ctx->fork_blur}, {"fork_fast0", ctx->fork_fast0}, {"fork_qt", ctx->fork_qt}, {"graph", ctx->use_graph}, {"graph_timing", ctx->graph_timing}, {"window_timing", ctx->window_timing},
      {"fast_pk", ctx->fast_pk}, {"fast_passes", ctx->fast_passes}, {"realign", ctx->realign}, {"fast_stage_dma", ctx->fast_stage_dma},
      {"gauss_kernel", ctx->gauss_kernel}, {"gauss_round", ctx->gauss_round}, {"gauss_tail", ctx->gauss_tail}, {"atan_fma", ctx->atan_fma}, {"brief_fma", ctx->brief_fma},
      {"qt_points", ctx->qt_points}, {"small_fused", ctx->small_fused}, {"qt_level_major", ctx->qt_level_major}, {"qt_fused", ctx->qt_fused},
      {"chain_batch", ctx->chain_batch}, {"chain_long", ctx->chain_long}, {"describe_direct", ctx->describe_direct}, {"chain_long_tile", ctx->chain_long_tile},
      {"chain_first", ctx->chain_first}, {"chain_threads", ctx->chain_threads}, {"qt_big_levels", ctx->qt_big_levels}, {"qt_threads_small", ctx->qt_threads_small},
      {"qt_one_launch", ctx->qt_one_launch}, {"window_direct", ctx->window_direct}, {"fast_split", ctx->fast_split}, {"desc_lds", ctx->desc_lds}, {"desc_fused_blur", ctx->desc_fused_blur},
      {"fast_threads", ctx->fast_threads}, {"qt_threads", ctx->qt_threads}, {"desc_k", ctx->desc_k}, {"streams", ctx->nstreams}, {"view_pool_cap", ctx->view_pool_cap}};
  for (const auto& t : tab) if (n == t.name) return t.value;
  return ORBX_E_INVALID;
}

// ---- named sets of the five result-changing options: WHICH build of "the reference CPU path" the output equals ----
namespace {
struct CpuProfile { const char* name; const char* alias; int gauss_kernel, gauss_round, gauss_tail; bool has_avx2_atan; const char* what; };
// the release ranges are recalled (no OpenCV of any version in the build container): tools/validate_opencv.cpp is how a maintainer checks them
const CpuProfile kProfiles[] = {
    {"opencv>=4.5.1", "default", 0, 0, 0, true, "cv::GaussianBlur 8u with the error-diffused kernel {18,34,48,56,...}, (acc + 2^15) >> 16: OpenCV >= 4.5.1 and every scalar path"},
    {"opencv-4.4", "opencv-4.4-avx2", 1, 2, 16, true, "OpenCV 3.4.2 .. 4.5.0 (CMakeLists.txt:33 asks for 4.4): kernel {18,34,49,55,...}, SIMD column pass that floors, v_uint16 of 16 lanes (AVX2 dispatch)"},
    {"opencv-4.4-sse", nullptr, 1, 2, 8, false, "the same releases on the SSE baseline: 8 lanes"},
    {"opencv-4.4-avx512", nullptr, 1, 2, 32, true, "the same releases with the AVX-512 dispatch: 32 lanes"},
    {"opencv-4.4-scalar", nullptr, 1, 0, 0, false, "the same releases without SIMD (CV_DISABLE_OPTIMIZATION / non-x86): kernel {18,34,49,55,...}, half-up rounding"},
    {"opencv-3.2", "opencv<=3.4.1", 1, 1, 4, false, "OpenCV <= 3.4.1 (README.md:560 names 3.2.0): integer sepFilter2D, SSE2 column pass (float sum + cvtps2dq: ties to even), 4-column tail"},
};
const CpuProfile* find_profile(const char* name) {
  if (!name) return nullptr;
  for (const CpuProfile& p : kProfiles) if (!strcmp(name, p.name) || (p.alias && !strcmp(name, p.alias))) return &p;
  return nullptr;
}
}  // namespace

const char* orbx_cpu_profile_name(int i) { return i >= 0 && i < (int)(sizeof(kProfiles) / sizeof(kProfiles[0])) ? kProfiles[i].name : nullptr; }
const char* orbx_cpu_profile_description(const char* name) { const CpuProfile* p = find_profile(name); return p ? p->what : nullptr; }

int orbx_cpu_profile_values(const char* name, int fma_build, int values[5]) {
  const CpuProfile* p = find_profile(name);
  if (!p || !values || fma_build < 0 || fma_build > 3) return ORBX_E_INVALID;
  values[0] = p->gauss_kernel; values[1] = p->gauss_round; values[2] = p->gauss_tail;
  values[3] = (fma_build & 2) ? 1 : 0;   // atan_fma: OpenCV's AVX2 dispatch copy of cv::fastAtan2 runs (an FMA machine AND a release / build that has one)
  values[4] = (fma_build & 1) ? 1 : 0;   // brief_fma: src/ORBextractor.cc itself built with -march=native on an FMA machine
  if ((fma_build & 2) && !p->has_avx2_atan) return ORBX_E_INVALID;   // this profile's OpenCV has no FMA copy of fastAtan2
  return ORBX_OK;
}

int orbx_set_cpu_profile(orbx_ctx* ctx, const char* name, int fma_build) {
  if (!ctx) return ORBX_E_INVALID;
  int v[5];
  if (orbx_cpu_profile_values(name, fma_build, v) != ORBX_OK)
    return set_err(ctx, ORBX_E_INVALID, std::string("orbx_set_cpu_profile: unknown profile or fma_build out of range for it: ") + (name ? name : "(null)"));
  static const char* opt[5] = {"gauss_kernel", "gauss_round", "gauss_tail", "atan_fma", "brief_fma"};
  for (int i = 0; i < 5; i++) { const int rc = orbx_set_option(ctx, opt[i], v[i]); if (rc != ORBX_OK) return rc; }
  return ORBX_OK;
}

int orbx_get_cpu_profile(const orbx_ctx* ctx, char* buf, size_t buf_bytes, int values[5]) {
  if (!ctx) return ORBX_E_INVALID;
  const int v[5] = {ctx->gauss_kernel, ctx->gauss_round, ctx->gauss_tail, ctx->atan_fma, ctx->brief_fma};
  if (values) for (int i = 0; i < 5; i++) values[i] = v[i];
  if (buf && buf_bytes) {
    const char* nm = "custom";
    for (const CpuProfile& p : kProfiles) if (p.gauss_kernel == v[0] && p.gauss_round == v[1] && p.gauss_tail == v[2]) { nm = p.name; break; }
    snprintf(buf, buf_bytes, "%s%s%s (gauss_kernel=%d gauss_round=%d gauss_tail=%d atan_fma=%d brief_fma=%d)", nm, v[3] ? " +avx2-atan" : "", v[4] ? " +native-build" : "",
             v[0], v[1], v[2], v[3], v[4]);
  }
  return ORBX_OK;
}

double orbx_last_graph_device_us(orbx_ctx* ctx) { return ctx ? ctx->last_graph_us : -1.0; }
double orbx_last_window_device_us(orbx_ctx* ctx) { return ctx ? ctx->last_window_us : -1.0; }

int orbx_set_host_pyramid(orbx_ctx* ctx, int on) {
  if (!ctx) return ORBX_E_INVALID;
  ctx->keep_host_pyr = on != 0;
  if (!on) ctx->h_pyr_valid = false;
  return ORBX_OK;
}

int orbx_host_pyramid_level(orbx_ctx* ctx, int level, const uint8_t** data, size_t* stride, int* w, int* h) {
  if (!ctx || !data || !ctx->d_geo || level < 0 || level >= ctx->nlevels)
    return ctx ? set_err(ctx, ORBX_E_INVALID, "no such pyramid level") : ORBX_E_INVALID;
  const LevelGeom& L = ctx->geo.lv[level];
  if (w) *w = L.w;
  if (h) *h = L.h;
  if (level == 0) { *data = nullptr; if (stride) *stride = 0; return ORBX_OK; }  // level 0 is the caller's own image
  if (!ctx->h_pyr_valid) return set_err(ctx, ORBX_E_INVALID, "host pyramid not enabled (orbx_set_host_pyramid) or no extraction yet");
  *data = ctx->h_pyr + L.plane_off;
  if (stride) *stride = (size_t)L.pitch;
  return ORBX_OK;
}

int orbx_profile_enable(orbx_ctx* ctx, int on) {
  if (!ctx) return ORBX_E_INVALID;
  ctx->profiling = on != 0;
  return ORBX_OK;
}

int orbx_profile_read(orbx_ctx* ctx, double ms[ORBX_NUM_KERNELS], int64_t launches[ORBX_NUM_KERNELS]) {
  if (!ctx) return ORBX_E_INVALID;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  ORBX_HIP(ctx, sync_ctx(ctx));
  for (size_t i = 0; i + 3 <= ctx->ev_pool.size(); i += 3) {
    hipEvent_t e0 = ctx->ev_pool[i], e1 = ctx->ev_pool[i + 1];
    const int slot = (int)(intptr_t)ctx->ev_pool[i + 2] - 1;
    float t = 0;
    if (hipEventElapsedTime(&t, e0, e1) == hipSuccess && slot >= 0 && slot < ORBX_NUM_KERNELS) ctx->prof_ms[slot] += t;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  ctx->ev_pool.clear();
  for (int i = 0; i < ORBX_NUM_KERNELS; i++) {
    if (ms) ms[i] = ctx->prof_ms[i];
    if (launches) launches[i] = ctx->prof_n[i];
    ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0;
  }
  return ORBX_OK;
}

#ifdef ORBX_DEBUG_ABI   // the diagnostic ABI (include/orbx_debug.h): compiled into liborbx_debug.so only (csrc/orbx_debug.hip), never into liborbx.so
int orbx_debug_blur_level(orbx_ctx* ctx, int frame, int level, uint8_t* dst, size_t dst_stride) {
  if (!ctx || !ctx->d_geo || !dst || level < 0 || level >= ctx->nlevels || frame < 0 || frame >= ctx->last_nframes)
    return ctx ? set_err(ctx, ORBX_E_INVALID, "no such pyramid level") : ORBX_E_INVALID;
  const LevelGeom& L = ctx->geo.lv[level];
  if (ctx->last_fused_blur || !ctx->d_blur || frame >= ctx->blur_cap)
    return set_err(ctx, ORBX_E_INVALID, "no blurred planes: the last extraction blurred inside the descriptor kernel (desc_fused_blur)");
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  ORBX_HIP(ctx, sync_ctx(ctx));
  ORBX_HIP(ctx, copy2d_sync(ctx, dst, dst_stride, ctx->d_blur + (size_t)frame * ctx->geo.blur_bytes + L.bplane_off, L.pitch, L.w, L.h,
                            hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_debug_level_points(orbx_ctx* ctx, int frame, int level, int stage, uint32_t* dst, int cap) {
  if (!ctx || !ctx->d_geo || level < 0 || level >= ctx->nlevels || frame < 0 || frame >= ctx->last_nframes)
    return ctx ? set_err(ctx, ORBX_E_INVALID, "no such level") : ORBX_E_INVALID;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  ORBX_HIP(ctx, sync_ctx(ctx));
  const Geometry& geo = ctx->geo;
  const LevelGeom& L = geo.lv[level];
  if (stage == 0) {
    std::vector<int32_t> cnt(L.ncells);
    ORBX_HIP(ctx, copy_sync(ctx, cnt.data(), ctx->d_cell_cnt + (size_t)frame * geo.cells.size() + L.cell_begin,
                            sizeof(int32_t) * L.ncells, hipMemcpyDeviceToHost));
    std::vector<uint32_t> slots(L.cand_cap);
    ORBX_HIP(ctx, copy_sync(ctx, slots.data(), ctx->d_cand + (size_t)frame * geo.cand_total + L.cand_off,
                            sizeof(uint32_t) * L.cand_cap, hipMemcpyDeviceToHost));
    int n = 0;
    for (int c = 0; c < L.ncells; c++) {
      const CellGeom& cg = geo.cells[L.cell_begin + c];
      for (int e = 0; e < cnt[c]; e++) {
        if (dst && n < cap) dst[n] = slots[cg.slot_off - L.cand_off + e];
        n++;
      }
    }
    return n;
  }
  int32_t n = 0;
  ORBX_HIP(ctx, copy_sync(ctx, &n, ctx->d_lvl_n + (size_t)frame * geo.nlevels + level, sizeof(int32_t), hipMemcpyDeviceToHost));
  if (n < 0) return set_err(ctx, ORBX_E_CAPACITY, "quadtree level overflow");
  if (dst && n > 0)
    ORBX_HIP(ctx, copy_sync(ctx, dst, ctx->d_lvl_kp + (size_t)frame * geo.kp_total + L.kp_off, sizeof(uint32_t) * std::min(n, cap),
                            hipMemcpyDeviceToHost));
  return n;
}

int orbx_debug_trig(orbx_ctx* ctx, const float* y, const float* x, int n, int angle_is_input, float* angle, float* a,
                    float* b) {
  if (!ctx || n < 0 || !y || !angle || !a || !b || (!angle_is_input && !x)) return ORBX_E_INVALID;
  if (n == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  float *dy = nullptr, *dx = nullptr, *dang = nullptr, *da = nullptr, *db = nullptr;
  const size_t bytes = (size_t)n * sizeof(float);
  ORBX_HIP(ctx, hipMalloc((void**)&dy, bytes)); ORBX_HIP(ctx, hipMalloc((void**)&dx, bytes));
  ORBX_HIP(ctx, hipMalloc((void**)&dang, bytes)); ORBX_HIP(ctx, hipMalloc((void**)&da, bytes)); ORBX_HIP(ctx, hipMalloc((void**)&db, bytes));
  ORBX_HIP(ctx, copy_sync(ctx, dy, y, bytes, hipMemcpyHostToDevice));
  if (x) ORBX_HIP(ctx, copy_sync(ctx, dx, x, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_debug_trig, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dy, dx, n, angle_is_input, dang, da, db, ctx->atan_fma);
  ORBX_HIP(ctx, hipGetLastError());
  ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ORBX_HIP(ctx, copy_sync(ctx, angle, dang, bytes, hipMemcpyDeviceToHost));
  ORBX_HIP(ctx, copy_sync(ctx, a, da, bytes, hipMemcpyDeviceToHost));
  ORBX_HIP(ctx, copy_sync(ctx, b, db, bytes, hipMemcpyDeviceToHost));
  (void)hipFree(dy); (void)hipFree(dx); (void)hipFree(dang); (void)hipFree(da); (void)hipFree(db);
  return ORBX_OK;
}

int orbx_debug_gnu_sort(orbx_ctx* ctx, uint64_t* elems, int n, int threads) {
  if (!ctx || n < 0 || n > 2048 || (n && !elems) || threads < 64 || threads > 512 || (threads & 63)) return ORBX_E_INVALID;
  if (n == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  unsigned long long* d = nullptr;
  ORBX_HIP(ctx, hipMalloc((void**)&d, (size_t)n * 8));
  ORBX_HIP(ctx, copy_sync(ctx, d, elems, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_debug_gnu_sort, dim3(1), dim3(threads), 0, ctx->stream, d, n);
  ORBX_HIP(ctx, hipGetLastError());
  ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ORBX_HIP(ctx, copy_sync(ctx, elems, d, (size_t)n * 8, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return ORBX_OK;
}

static int run_hash_kernel(orbx_ctx* ctx, int which, uint32_t a, uint32_t count, uint64_t* hash);

int orbx_debug_atan_hash(orbx_ctx* ctx, uint32_t seed, uint32_t count, uint64_t* hash) {
  if (!ctx || !hash) return ORBX_E_INVALID;
  return run_hash_kernel(ctx, 1, seed, count, hash);
}

int orbx_debug_brief_hash(orbx_ctx* ctx, uint32_t first_bits, uint32_t count, uint64_t* hash) {
  if (!ctx || !hash) return ORBX_E_INVALID;
  return run_hash_kernel(ctx, 2, first_bits, count, hash);
}

int orbx_debug_trig_hash(orbx_ctx* ctx, uint32_t first_bits, uint32_t count, uint64_t* hash) {
  if (!ctx || !hash) return ORBX_E_INVALID;
  return run_hash_kernel(ctx, 0, first_bits, count, hash);
}

static int run_hash_kernel(orbx_ctx* ctx, int which, uint32_t first_bits, uint32_t count, uint64_t* hash) {
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  unsigned long long* d = nullptr;
  ORBX_HIP(ctx, hipMalloc((void**)&d, sizeof(unsigned long long)));
  ORBX_HIP(ctx, hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
  if (count && which == 0) hipLaunchKernelGGL(k_debug_trig_hash, dim3(256 * 32), dim3(256), 0, ctx->stream, first_bits, count, d);
  if (count && which == 1) hipLaunchKernelGGL(k_debug_atan_hash, dim3(256 * 32), dim3(256), 0, ctx->stream, first_bits, count, d, ctx->atan_fma);
  if (count && which == 2) hipLaunchKernelGGL(k_debug_brief_hash, dim3(256 * 32), dim3(256), 0, ctx->stream, first_bits, count, d, ctx->brief_fma);
  ORBX_HIP(ctx, hipGetLastError());
  unsigned long long h = 0;
  ORBX_HIP(ctx, hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  ORBX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(d);
  *hash = (uint64_t)h;
  return ORBX_OK;
}

int orbx_debug_calib_copy(orbx_ctx* ctx, const void* d_src, void* d_dst, size_t nbytes, int width, void* stream) {
  if (!ctx || !d_src || !d_dst || (width != 1 && width != 4 && width != 16) || nbytes % 16 != 0) return ORBX_E_INVALID;
  if (nbytes == 0) return ORBX_OK;
  ORBX_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const long long n = (long long)(nbytes / (size_t)width);
  const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 256 * 32);
  if (width == 1) hipLaunchKernelGGL(k_calib_copy<uint8_t>, dim3(blocks), dim3(256), 0, st, (const uint8_t*)d_src, (uint8_t*)d_dst, n);
  else if (width == 4) hipLaunchKernelGGL(k_calib_copy<uint32_t>, dim3(blocks), dim3(256), 0, st, (const uint32_t*)d_src, (uint32_t*)d_dst, n);
  else hipLaunchKernelGGL(k_calib_copy<uint4>, dim3(blocks), dim3(256), 0, st, (const uint4*)d_src, (uint4*)d_dst, n);
  ORBX_HIP(ctx, hipGetLastError());
  return ORBX_OK;
}

#ifdef ORBX_QT_PROFILE
int orbx_debug_qt_profile(orbx_ctx* ctx, long long* out, int reset) {
  (void)sync_ctx(ctx);
  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qt_prof), sizeof(long long) * kMaxLevels * 8);
  if (reset) { long long z[kMaxLevels * 8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_qt_prof), z, sizeof(z)); }
  return 0;
}
#endif

#endif  // ORBX_DEBUG_ABI

const char* orbx_kernel_name(int slot) {
  static const char* names[ORBX_NUM_KERNELS] = {"k_resize(pyramid chain)", "k_fast_cells", "k_quadtree", "k_assemble",
                                                "k_blur7", "k_describe"};
  return slot >= 0 && slot < ORBX_NUM_KERNELS ? names[slot] : "";
}

}  // extern "C"
