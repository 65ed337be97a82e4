// orbx — batch replay (BASELINE.json config 5): camera streams sharded one per GPU, one process per GPU, and ONE exchange
// per step: an all-gather of every rank's fixed-size feature block, asynchronous on its own stream and double-buffered so that it overlaps
// the next step's kernels.  RCCL is called directly (ncclAllGather over xGMI); the host — C++ like the reference's (src/System.cc:197-264
// owns every thread) — needs nothing but this C ABI (include/orbx.h: orbx_replay_*).  Rounds 1-4 had this loop in Python over
// torch.distributed; orb_slam3_modified_amd/replay.py is now a ctypes wrapper over these entry points.
//
// RCCL is bound at run time (dlopen + dlsym): liborbx.so has no link-time dependency on it, a process that already carries an RCCL (PyTorch
// ships its own) shares that instance instead of loading a second one, and a host without RCCL can still use everything else.  A caller
// without RCCL between its ranks can hand in a host all-gather (MPI, gloo, sockets) instead: the block then crosses PCIe twice.
//
// Feature block of one rank and step (one contiguous buffer, fixed size so that the collective is regular):
//     [B][cap] orbx_keypoint (28 B) | [B][cap][32] descriptor bytes | [B][2] int32 (n, monoIndex)        each part 256-byte aligned
// gather "descriptors" moves the tail (descriptor rows + counts: north_star's exchange), "blocks" the whole block (SURVEY §8(e)'s).
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "orbx_internal.h"

namespace {

// the few RCCL types / constants this file needs (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE, ncclUint8 = 1, ncclSuccess = 0)
struct RcclUniqueId { char internal[ORBX_REPLAY_UNIQUE_ID_BYTES]; };
typedef void* RcclComm;
struct Rccl {
  void* lib = nullptr;
  std::string where;   // which library, which version
  int version = 0;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*CommAbort)(RcclComm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};
std::mutex g_rccl_mu;
Rccl g_rccl;
std::string g_rccl_err;

// one RCCL per process: the one already loaded (PyTorch's) if there is one, else the system's
const Rccl* rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.lib) return &g_rccl;
  if (!g_rccl_err.empty()) return nullptr;
  const char* env = getenv("ORBX_RCCL_LIB");
  std::vector<std::pair<std::string, int>> tries;
  if (env && *env) tries.push_back({env, RTLD_NOW | RTLD_LOCAL});
  else {
    for (const char* n : {"librccl.so", "librccl.so.1"}) tries.push_back({n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});   // already in the process?
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) tries.push_back({n, RTLD_NOW | RTLD_LOCAL});
  }
  std::string errs;
  for (const auto& t : tries) {
    void* h = dlopen(t.first.c_str(), t.second);
    if (!h) { if (!(t.second & RTLD_NOLOAD)) { const char* e = dlerror(); errs += t.first + ": " + (e ? e : "?") + "; "; } continue; }
    Rccl r;
    r.lib = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.GetVersion = (decltype(r.GetVersion))dlsym(h, "ncclGetVersion");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) { errs += t.first + ": not an RCCL (symbols missing); "; dlclose(h); continue; }
    if (r.GetVersion) (void)r.GetVersion(&r.version);
    char buf[160];
    // NCCL_VERSION_CODE = major * 10000 + minor * 100 + patch (since 2.9)
    snprintf(buf, sizeof buf, "rccl %d.%d.%d (%s%s)", r.version / 10000, (r.version / 100) % 100, r.version % 100, t.first.c_str(),
             (t.second & RTLD_NOLOAD) ? ", the instance already loaded in this process" : "");
    r.where = buf;
    g_rccl = r;
    return &g_rccl;
  }
  g_rccl_err = "no RCCL library could be loaded (" + errs + "set ORBX_RCCL_LIB to its path)";
  return nullptr;
}

size_t up256(size_t n) { return (n + 255) / 256 * 256; }

constexpr int kTimingPairs = 64;

}  // namespace

struct orbx_replay {
  std::vector<orbx_ctx*> lanes;
  std::vector<std::pair<int, int>> ranges;   // frames [f0, f1) of the step's batch per lane
  int device = 0, B = 0, rows = 0, cols = 0, cap = 0;
  int rank = 0, world = 1, gather_what = ORBX_GATHER_NONE;
  bool gather_on = false;
  size_t kps_bytes = 0, desc_bytes = 0, counts_bytes = 0, desc_off = 0, counts_off = 0, nbytes = 0, send_off = 0, send_bytes = 0;
  uint8_t* blocks[2] = {nullptr, nullptr};
  uint8_t* gathered[2] = {nullptr, nullptr};
  std::vector<hipStream_t> streams;
  hipStream_t gstream = nullptr;
  std::vector<hipEvent_t> lane_done[2];
  hipEvent_t gather_done[2] = {nullptr, nullptr};
  bool pending[2] = {false, false};   // a collective that read blocks[i] has been queued and no lane has waited for it yet
  bool gathered_valid[2] = {false, false};     // gather_done[i] has been recorded at least once: gathered buffer i holds (or will hold) a step
  // a consumer of gathered buffer i on a stream of its own (orbx_replay_wait_gathered / _release_gathered): the next collective INTO buffer i
  // waits for the event the consumer's release recorded
  hipEvent_t consumer_done[2] = {nullptr, nullptr};
  bool consumer_pending[2] = {false, false};
  // what create changed in the caller's lane contexts (restored on destroy): fork_blur, fork_fast0, fork_qt
  std::vector<int> saved_forks;
  bool connected = false;                      // transport up (ncclCommInitRank done, or host transport / none)
  // failure containment: once a lane failed, this rank keeps taking part in every step's exchange with a POISONED block (all counts -1), so that
  // the other ranks' collectives complete; every later step returns the first error again
  int failed_code = 0; std::string failed_msg;
  // transport
  RcclComm comm = nullptr;
  orbx_host_exchange_fn host_fn = nullptr; void* host_user = nullptr;
  uint8_t* h_send = nullptr; uint8_t* h_recv = nullptr;
  std::string transport = "none";
  // device time of the collectives (HIP events on the gather stream): a ring of pairs, read back when a pair is reused
  hipEvent_t t0[kTimingPairs] = {}, t1[kTimingPairs] = {};
  bool t_live[kTimingPairs] = {};
  double t_sum_ms = 0; long long t_n = 0;
  unsigned long long step_idx = 0;
  std::string err;
  // lane schedule (round 6).  alternate: the lanes take WHOLE steps in turn — step k runs on lane k mod L over all `frames` frames while the other
  // lanes are still busy with the steps before it — instead of every lane working on its share of every step.  Same results, the same overlap of
  // two free-running streams, but every launch covers the whole batch: +3.5 % at 256 frames per step (0.912 -> 0.878 ms, profiles/replay_alternate_ab_r6.txt);
  // a step's results are complete one step later.  Default for two and more lanes; lane 0's option "replay_alternate" = 0 (or ORBX_REPLAY_ALTERNATE=0)
  // keeps the split form.
  bool alternate = false;
};

namespace {

int rfail(orbx_replay* r, int code, const std::string& msg) { if (r) r->err = msg; return code; }
#define RHIP(r, expr)                                                                                    \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) return rfail((r), ORBX_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

void harvest(orbx_replay* r, int k) {
  if (!r->t_live[k]) return;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, r->t0[k], r->t1[k]) == hipSuccess) { r->t_sum_ms += ms; r->t_n++; }
  else (void)hipGetLastError();
  r->t_live[k] = false;
}

void release(orbx_replay* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  for (hipStream_t s : r->streams) if (s) (void)hipStreamSynchronize(s);
  if (r->gstream) (void)hipStreamSynchronize(r->gstream);
  if (r->comm) { const Rccl* R = rccl(); if (R) (void)R->CommDestroy(r->comm); r->comm = nullptr; }
  for (size_t j = 0; j * 3 + 2 < r->saved_forks.size() && j < r->lanes.size(); j++) {   // the lanes are the caller's: give them back as they came
    (void)orbx_set_option(r->lanes[j], "fork_blur", r->saved_forks[j * 3]);
    (void)orbx_set_option(r->lanes[j], "fork_fast0", r->saved_forks[j * 3 + 1]);
    (void)orbx_set_option(r->lanes[j], "fork_qt", r->saved_forks[j * 3 + 2]);
  }
  for (int i = 0; i < 2; i++) {
    if (r->consumer_done[i]) (void)hipEventDestroy(r->consumer_done[i]);
    if (r->blocks[i]) (void)hipFree(r->blocks[i]);
    if (r->gathered[i]) (void)hipFree(r->gathered[i]);
    for (hipEvent_t e : r->lane_done[i]) if (e) (void)hipEventDestroy(e);
    if (r->gather_done[i]) (void)hipEventDestroy(r->gather_done[i]);
  }
  for (int k = 0; k < kTimingPairs; k++) { if (r->t0[k]) (void)hipEventDestroy(r->t0[k]); if (r->t1[k]) (void)hipEventDestroy(r->t1[k]); }
  if (r->h_send) (void)hipHostFree(r->h_send);
  if (r->h_recv) (void)hipHostFree(r->h_recv);
  for (hipStream_t s : r->streams) if (s) (void)hipStreamDestroy(s);
  if (r->gstream) (void)hipStreamDestroy(r->gstream);
  delete r;
}

}  // namespace

extern "C" {

int orbx_replay_unique_id(uint8_t id[ORBX_REPLAY_UNIQUE_ID_BYTES]) {
  if (!id) return ORBX_E_INVALID;
  const Rccl* R = rccl();
  if (!R) return ORBX_E_DEVICE;
  RcclUniqueId u;
  if (R->GetUniqueId(&u) != 0) return ORBX_E_DEVICE;
  std::memcpy(id, u.internal, ORBX_REPLAY_UNIQUE_ID_BYTES);
  return ORBX_OK;
}

const char* orbx_replay_rccl_info(void) {
  const Rccl* R = rccl();
  return R ? R->where.c_str() : g_rccl_err.c_str();
}

// Everything a rank can fail at ON ITS OWN (arguments, lane parameters, buffers, streams, the RCCL library itself) happens here and touches no
// other rank: a multi-rank host all-reduces "did prepare succeed" over its control plane BEFORE any rank enters ncclCommInitRank
// (orbx_replay_connect), in which a rank whose peers never arrive would wait forever.
int orbx_replay_prepare(orbx_replay** out, orbx_ctx* const* lanes, int nlanes, int frames, int rows, int cols, int gather_what, int rank, int world,
                        int use_rccl, orbx_host_exchange_fn host_exchange, void* user) {
  if (!out) return ORBX_E_INVALID;
  *out = nullptr;
  if (!lanes || nlanes < 1 || nlanes > 16 || frames < nlanes || frames > 65535 || rows <= 0 || cols <= 0 || world < 1 || rank < 0 || rank >= world ||
      gather_what < ORBX_GATHER_NONE || gather_what > ORBX_GATHER_BLOCKS || (use_rccl && host_exchange))
    return ORBX_E_INVALID;
  for (int j = 0; j < nlanes; j++) if (!lanes[j]) return ORBX_E_INVALID;
  for (int j = 0; j < nlanes; j++) for (int k = 0; k < j; k++) if (lanes[j] == lanes[k]) return ORBX_E_INVALID;
  // "identical parameters": everything that decides a result byte — frame ranges of ONE block must not follow different builds of the reference
  for (int j = 1; j < nlanes; j++) {
    const orbx_ctx *a = lanes[0], *b = lanes[j];
    const char* what = nullptr;
    if (b->device != a->device) what = "device";
    else if (b->nfeatures != a->nfeatures || b->out_cap != a->out_cap) what = "nfeatures";
    else if (b->nlevels != a->nlevels || b->scale_factor != a->scale_factor) what = "pyramid (levels, scale factor)";
    else if (b->ini_th != a->ini_th || b->min_th != a->min_th) what = "FAST thresholds";
    else if (b->gauss_kernel != a->gauss_kernel || b->gauss_round != a->gauss_round || b->gauss_tail != a->gauss_tail || b->atan_fma != a->atan_fma ||
             b->brief_fma != a->brief_fma) what = "CPU-path profile (gauss_kernel / gauss_round / gauss_tail / atan_fma / brief_fma)";
    if (what) {
      orbx::set_err(lanes[0], ORBX_E_INVALID, std::string("orbx_replay_prepare: lane ") + std::to_string(j) + " differs from lane 0 in its " + what +
                                                  ": the lanes of one engine must compute the same reference build");
      return ORBX_E_INVALID;
    }
  }
  if (gather_what != ORBX_GATHER_NONE && world > 1 && !use_rccl && !host_exchange) return ORBX_E_INVALID;   // more than one rank needs a way to reach the others
  orbx_replay* r = new orbx_replay;
  r->lanes.assign(lanes, lanes + nlanes);
  r->device = lanes[0]->device; r->B = frames; r->rows = rows; r->cols = cols; r->cap = lanes[0]->out_cap;
  r->rank = rank; r->world = world; r->gather_what = gather_what; r->gather_on = gather_what != ORBX_GATHER_NONE;
  r->kps_bytes = (size_t)frames * r->cap * sizeof(orbx_keypoint); r->desc_bytes = (size_t)frames * r->cap * 32; r->counts_bytes = (size_t)frames * 8;
  r->desc_off = up256(r->kps_bytes); r->counts_off = r->desc_off + up256(r->desc_bytes); r->nbytes = r->counts_off + up256(r->counts_bytes);
  r->send_off = gather_what == ORBX_GATHER_DESCRIPTORS ? r->desc_off : 0;
  r->send_bytes = r->nbytes - r->send_off;
  const int per = (frames + nlanes - 1) / nlanes;
  {
    int want = lanes[0]->replay_alternate;   // -1 = by default, 0 / 1 = the caller's choice
    if (const char* ae = getenv("ORBX_REPLAY_ALTERNATE")) want = atoi(ae) != 0;
    r->alternate = nlanes >= 2 && want != 0;
  }
  if (r->alternate) r->ranges.assign(nlanes, {0, frames});
  else for (int j = 0; j < nlanes; j++) if (j * per < frames) r->ranges.push_back({j * per, std::min(frames, (j + 1) * per)});
  r->lanes.resize(r->ranges.size());
  auto bail = [&](int code, const std::string& msg) { if (r->lanes[0]) orbx::set_err(r->lanes[0], code, msg); release(r); return code; };
  if (hipSetDevice(r->device) != hipSuccess) return bail(ORBX_E_DEVICE, "orbx_replay_prepare: hipSetDevice failed");
  // With a second lane filling the idle issue slots the in-lane forks that pay differ from a lone context's.  Measured on all eight
  // combinations (2 lanes x 128 frames, round 2): blur forked behind FAST + level-0 FAST beside the pyramid chain + the quadtree as one
  // launch.  The ORBX_* environment variables still win.  The contexts are the caller's: destroy puts the three options back.
  for (orbx_ctx* c : r->lanes) { r->saved_forks.push_back(c->fork_blur); r->saved_forks.push_back(c->fork_fast0); r->saved_forks.push_back(c->fork_qt); }
  if (r->lanes.size() > 1)
    for (orbx_ctx* c : r->lanes) {
      if (!getenv("ORBX_FORK_BLUR")) (void)orbx_set_option(c, "fork_blur", 1);
      if (!getenv("ORBX_FORK_FAST0")) (void)orbx_set_option(c, "fork_fast0", 1);
      if (!getenv("ORBX_FORK_QT")) (void)orbx_set_option(c, "fork_qt", 0);
    }
  for (size_t j = 0; j < r->lanes.size(); j++) {   // every lane's buffers now, not inside the first (possibly timed) step
    const int rc = orbx_reserve(r->lanes[j], rows, cols, r->ranges[j].second - r->ranges[j].first);
    if (rc != ORBX_OK) return bail(rc, std::string("orbx_replay_prepare: orbx_reserve: ") + orbx_last_error(r->lanes[j]));
  }
  // Explicit non-blocking streams carry the lanes; the collective ALWAYS runs on its own stream behind every lane of the step: on a lane's
  // stream step k + 1's kernels would queue behind step k's collective and the overlap would be gone.
  hipError_t e = hipSuccess;
  r->streams.assign(r->lanes.size(), nullptr);
  for (size_t j = 0; j < r->lanes.size() && e == hipSuccess; j++) e = hipStreamCreateWithFlags(&r->streams[j], hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->gstream, hipStreamNonBlocking);
  // (no entry point of this library touches the legacy stream — orbx_internal.h: a null-stream operation of one thread poisons another
  // thread's graph capture — so the buffers are cleared on the engine's own stream)
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&r->blocks[i], r->nbytes);
    if (e == hipSuccess) e = hipMemsetAsync(r->blocks[i], 0, r->nbytes, r->gstream);
    if (e == hipSuccess && r->gather_on) {
      e = hipMalloc((void**)&r->gathered[i], r->send_bytes * (size_t)world);
      if (e == hipSuccess) e = hipMemsetAsync(r->gathered[i], 0, r->send_bytes * (size_t)world, r->gstream);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->gather_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->consumer_done[i], hipEventDisableTiming);
    r->lane_done[i].assign(r->lanes.size(), nullptr);
    for (size_t j = 0; j < r->lanes.size() && e == hipSuccess; j++) e = hipEventCreateWithFlags(&r->lane_done[i][j], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(r->gstream);
  for (int k = 0; k < kTimingPairs && e == hipSuccess && r->gather_on; k++) { e = hipEventCreate(&r->t0[k]); if (e == hipSuccess) e = hipEventCreate(&r->t1[k]); }
  if (e != hipSuccess) return bail(ORBX_E_DEVICE, std::string("orbx_replay_prepare: ") + hipGetErrorString(e));
  if (r->gather_on) {
    if (host_exchange) {
      r->host_fn = host_exchange; r->host_user = user;
      e = hipHostMalloc((void**)&r->h_send, r->send_bytes, hipHostMallocDefault);
      if (e == hipSuccess) e = hipHostMalloc((void**)&r->h_recv, r->send_bytes * (size_t)world, hipHostMallocDefault);
      if (e != hipSuccess) return bail(ORBX_E_DEVICE, std::string("orbx_replay_prepare: pinned staging: ") + hipGetErrorString(e));
      r->transport = "host all-gather supplied by the caller (block staged through pinned memory, synchronous)";
      r->connected = true;
    } else if (!rccl()) return bail(ORBX_E_DEVICE, "orbx_replay_prepare: " + g_rccl_err);   // the library is resolved HERE, not inside connect
  } else r->connected = true;
  *out = r;
  return ORBX_OK;
}

// The one step that meets the other ranks: ncclCommInitRank.  unique_id NULL: world must be 1 (a one-rank group: the self-gather).
// On failure the engine stays usable for the sharded extraction after orbx_replay_set_gather(r, 0).
int orbx_replay_connect(orbx_replay* r, const uint8_t* unique_id) {
  if (!r) return ORBX_E_INVALID;
  if (r->connected) return ORBX_OK;
  if (!unique_id && r->world > 1) return rfail(r, ORBX_E_INVALID, "orbx_replay_connect: more than one rank needs the ncclUniqueId of orbx_replay_unique_id()");
  const Rccl* R = rccl();
  if (!R) return rfail(r, ORBX_E_DEVICE, "orbx_replay_connect: " + g_rccl_err);
  RHIP(r, hipSetDevice(r->device));
  RcclUniqueId u;
  if (unique_id) std::memcpy(u.internal, unique_id, sizeof u.internal);
  else if (R->GetUniqueId(&u) != 0) return rfail(r, ORBX_E_DEVICE, "orbx_replay_connect: ncclGetUniqueId failed");
  const int rc = R->CommInitRank(&r->comm, r->world, u, r->rank);
  if (rc != 0) { r->comm = nullptr; return rfail(r, ORBX_E_DEVICE, std::string("orbx_replay_connect: ncclCommInitRank: ") + (R->GetErrorString ? R->GetErrorString(rc) : "error")); }
  r->transport = "ncclAllGather, " + R->where;
  r->connected = true;
  return ORBX_OK;
}

int orbx_replay_create(orbx_replay** out, orbx_ctx* const* lanes, int nlanes, int frames, int rows, int cols, int gather_what, int rank, int world,
                       const uint8_t* unique_id, orbx_host_exchange_fn host_exchange, void* user) {
  if (unique_id && host_exchange) return ORBX_E_INVALID;
  if (gather_what != ORBX_GATHER_NONE && world > 1 && !unique_id && !host_exchange) return ORBX_E_INVALID;
  const int use_rccl = gather_what != ORBX_GATHER_NONE && !host_exchange;
  int rc = orbx_replay_prepare(out, lanes, nlanes, frames, rows, cols, gather_what, rank, world, use_rccl, host_exchange, user);
  if (rc != ORBX_OK) return rc;
  if (!(*out)->connected) {
    rc = orbx_replay_connect(*out, unique_id);
    if (rc != ORBX_OK) {
      orbx::set_err((*out)->lanes[0], rc, "orbx_replay_create: " + (*out)->err);
      release(*out);
      *out = nullptr;
      return rc;
    }
  }
  return ORBX_OK;
}

void orbx_replay_destroy(orbx_replay* r) { release(r); }

const char* orbx_replay_last_error(const orbx_replay* r) { return r ? r->err.c_str() : "null replay engine"; }
const char* orbx_replay_transport(const orbx_replay* r) { return r ? r->transport.c_str() : ""; }

int orbx_replay_layout(const orbx_replay* r, int* frames, int* capacity, size_t* block_bytes, size_t* desc_off, size_t* counts_off, size_t* send_off,
                       size_t* send_bytes, int* nlanes) {
  if (!r) return ORBX_E_INVALID;
  if (frames) *frames = r->B;
  if (capacity) *capacity = r->cap;
  if (block_bytes) *block_bytes = r->nbytes;
  if (desc_off) *desc_off = r->desc_off;
  if (counts_off) *counts_off = r->counts_off;
  if (send_off) *send_off = r->send_off;
  if (send_bytes) *send_bytes = r->send_bytes;
  if (nlanes) *nlanes = (int)r->lanes.size();
  return ORBX_OK;
}

int orbx_replay_lane_range(const orbx_replay* r, int lane, int* f0, int* f1) {
  if (!r || lane < 0 || lane >= (int)r->ranges.size()) return ORBX_E_INVALID;
  if (f0) *f0 = r->ranges[lane].first;
  if (f1) *f1 = r->ranges[lane].second;
  return ORBX_OK;
}

int orbx_replay_set_gather(orbx_replay* r, int on) {
  if (!r) return ORBX_E_INVALID;
  if (on && r->gather_what == ORBX_GATHER_NONE) return rfail(r, ORBX_E_INVALID, "orbx_replay_set_gather: the engine was created without an exchange");
  r->gather_on = on != 0;
  return ORBX_OK;
}

int orbx_replay_step(orbx_replay* r, const uint8_t* d_frames, size_t row_stride, size_t frame_stride, int lap0, int lap1) {
  if (!r || !d_frames) return r ? rfail(r, ORBX_E_INVALID, "orbx_replay_step: bad arguments") : ORBX_E_INVALID;
  if (r->gather_on && !r->connected) return rfail(r, ORBX_E_INVALID, "orbx_replay_step: the exchange is on but orbx_replay_connect has not succeeded");
  RHIP(r, hipSetDevice(r->device));
  const int i = (int)(r->step_idx & 1);
  uint8_t* const base = r->blocks[i];
  // A lane that fails must not keep this rank out of the step's collective: the other ranks are (or will be) inside theirs and would wait for
  // ever.  The rank goes on taking part — this step and every later one — with a POISONED block: every count of the block is {-1, -1}, which
  // no extraction produces, so a reader of the gathered buffer sees which rank dropped out at which step; the caller of THIS rank gets the
  // lane's error from this call and from every later one (orbx_replay_failed), and decides when to leave (orbx_replay_destroy after the
  // others have been told over the host's own control plane, or orbx_replay_abort).
  const size_t active = r->alternate ? (size_t)(r->step_idx % r->lanes.size()) : 0;
  for (size_t j = 0; j < r->lanes.size() && !r->failed_code; j++) {
    if (r->alternate && j != active) continue;
    const int f0 = r->ranges[j].first, f1 = r->ranges[j].second;
    // the collective that last read this block must be done before a lane overwrites it (a device-side wait: no host stall)
    if (r->pending[i]) RHIP(r, hipStreamWaitEvent(r->streams[j], r->gather_done[i], 0));
    if (r->alternate)   // ... and so must the lane that wrote this block two steps ago, when that was another lane
      for (size_t jj = 0; jj < r->lanes.size(); jj++) if (jj != j) (void)hipStreamWaitEvent(r->streams[j], r->lane_done[i][jj], 0);
    const int rc = orbx_extract_batch_device(r->lanes[j], d_frames + (size_t)f0 * frame_stride, f1 - f0, r->rows, r->cols, row_stride, frame_stride, lap0, lap1,
                                             (orbx_keypoint*)(base + (size_t)f0 * r->cap * sizeof(orbx_keypoint)), base + r->desc_off + (size_t)f0 * r->cap * 32,
                                             (int32_t*)(base + r->counts_off + (size_t)f0 * 8), r->streams[j]);
    if (rc != ORBX_OK) {
      r->failed_code = rc;
      r->failed_msg = std::string("orbx_replay_step: step ") + std::to_string(r->step_idx) + ", lane " + std::to_string(j) + ": " +
                      orbx_last_error(r->lanes[j]);
    }
  }
  if (r->failed_code) {
    r->err = r->failed_msg;
    if (!r->gather_on) { r->step_idx++; return r->failed_code; }
    // the poison, behind whatever the lanes of this step already queued and behind the collective that last read the block (alternate schedule:
    // on the step's own lane, whose stream already waits for the block's previous writers)
    hipStream_t ps = r->streams[r->alternate ? active : 0];
    if (r->pending[i]) (void)hipStreamWaitEvent(ps, r->gather_done[i], 0);
    if (!r->alternate)
      for (size_t j = 1; j < r->lanes.size(); j++)
        if (hipEventRecord(r->lane_done[i][j], r->streams[j]) == hipSuccess) (void)hipStreamWaitEvent(ps, r->lane_done[i][j], 0);
    (void)hipMemsetAsync(base + r->counts_off, 0xff, r->counts_bytes, ps);
  }
  if (r->gather_on || r->alternate)
    for (size_t j = 0; j < r->lanes.size(); j++) {
      if (r->alternate && j != active) continue;
      if (hipEventRecord(r->lane_done[i][j], r->streams[j]) != hipSuccess && !r->failed_code) return rfail(r, ORBX_E_DEVICE, "orbx_replay_step: hipEventRecord failed");
    }
  r->pending[i] = false;
  int xrc = ORBX_OK;   // the exchange's own failure (reported when the lanes were fine)
  if (r->gather_on) {   // queued behind this step's kernels, overlaps the next step's
    for (hipEvent_t ev : r->lane_done[i]) (void)hipStreamWaitEvent(r->gstream, ev, 0);
    // a consumer that announced it is still reading gathered buffer i (orbx_replay_release_gathered) is waited for on the device
    if (r->consumer_pending[i]) { (void)hipStreamWaitEvent(r->gstream, r->consumer_done[i], 0); r->consumer_pending[i] = false; }
    const uint8_t* send = base + r->send_off;
    if (r->comm) {
      const int k = (int)(r->step_idx % kTimingPairs);
      harvest(r, k);
      (void)hipEventRecord(r->t0[k], r->gstream);
      const Rccl* R = rccl();
      const int rc = R->AllGather(send, r->gathered[i], r->send_bytes, /* ncclUint8 */ 1, r->comm, r->gstream);
      if (rc != 0) xrc = rfail(r, ORBX_E_DEVICE, std::string("ncclAllGather: ") + (R->GetErrorString ? R->GetErrorString(rc) : "error"));
      (void)hipEventRecord(r->t1[k], r->gstream);
      r->t_live[k] = xrc == ORBX_OK;
    } else {
      hipError_t e = hipMemcpyAsync(r->h_send, send, r->send_bytes, hipMemcpyDeviceToHost, r->gstream);
      if (e == hipSuccess) e = hipStreamSynchronize(r->gstream);
      if (e != hipSuccess) {   // even then the other ranks are waiting in their host all-gather: send them the poison from the host side
        (void)hipGetLastError();
        std::memset(r->h_send, 0, r->send_bytes);
        std::memset(r->h_send + (r->counts_off - r->send_off), 0xff, r->counts_bytes);
        if (!r->failed_code) { r->failed_code = ORBX_E_DEVICE; r->failed_msg = std::string("orbx_replay_step: staging the block: ") + hipGetErrorString(e); r->err = r->failed_msg; }
      }
      const int rc = r->host_fn(r->host_user, r->h_send, r->h_recv, r->send_bytes);
      if (rc != 0) xrc = rfail(r, ORBX_E_DEVICE, "orbx_replay_step: the caller's host all-gather failed with " + std::to_string(rc));
      else if (hipMemcpyAsync(r->gathered[i], r->h_recv, r->send_bytes * (size_t)r->world, hipMemcpyHostToDevice, r->gstream) != hipSuccess)
        xrc = rfail(r, ORBX_E_DEVICE, "orbx_replay_step: gathered buffer upload failed");
    }
    if (hipEventRecord(r->gather_done[i], r->gstream) == hipSuccess) { r->pending[i] = true; r->gathered_valid[i] = true; }
  }
  r->step_idx++;
  if (r->failed_code) { r->err = r->failed_msg; return r->failed_code; }
  if (xrc != ORBX_OK) return xrc;
  return i;
}

// ---- failure containment (see orbx_replay_step)
int orbx_replay_failed(const orbx_replay* r) { return r ? r->failed_code : ORBX_E_INVALID; }

// Leaving a group whose other ranks may be gone: ncclCommAbort frees the communicator WITHOUT the collective hand-shake of ncclCommDestroy and
// releases collectives of this rank that are stuck on the gather stream.  The engine keeps working with the exchange off.
int orbx_replay_abort(orbx_replay* r) {
  if (!r) return ORBX_E_INVALID;
  (void)hipSetDevice(r->device);
  if (r->comm) {
    const Rccl* R = rccl();
    if (R && R->CommAbort) (void)R->CommAbort(r->comm);
    else if (R) (void)R->CommDestroy(r->comm);
    r->comm = nullptr;
  }
  r->gather_on = false;
  r->transport += " [aborted]";
  return ORBX_OK;
}

// ---- ordering a consumer against ONE step's exchange, without draining the engine
// orbx_replay_step returns at once and the collective into gathered buffer i runs on the engine's own gather stream: a kernel the caller
// launches on ITS stream knows nothing of it.  wait: `consumer` waits (on the device) for the last collective queued into buffer i.
// release: the consumer is done with buffer i as far as `consumer` has been fed so far — the next collective INTO buffer i (two steps later)
// waits for that point instead of overwriting what is being read.
int orbx_replay_wait_gathered(orbx_replay* r, int i, void* consumer_stream) {
  hipStream_t consumer = (hipStream_t)consumer_stream;
  if (!r || i < 0 || i > 1) return ORBX_E_INVALID;
  if (!r->gathered[i] || !r->gathered_valid[i]) return rfail(r, ORBX_E_INVALID, "orbx_replay_wait_gathered: no exchange has been queued into this buffer");
  RHIP(r, hipSetDevice(r->device));
  RHIP(r, hipStreamWaitEvent(consumer, r->gather_done[i], 0));
  return ORBX_OK;
}

int orbx_replay_release_gathered(orbx_replay* r, int i, void* consumer_stream) {
  hipStream_t consumer = (hipStream_t)consumer_stream;
  if (!r || i < 0 || i > 1) return ORBX_E_INVALID;
  if (!r->gathered[i]) return rfail(r, ORBX_E_INVALID, "orbx_replay_release_gathered: the engine was created without an exchange");
  RHIP(r, hipSetDevice(r->device));
  RHIP(r, hipEventRecord(r->consumer_done[i], consumer));
  r->consumer_pending[i] = true;
  return ORBX_OK;
}

// host variant of wait: returns ORBX_OK when the last collective into buffer i has completed, ORBX_E_TIMEOUT after timeout_ms (< 0: no limit)
// — a rank whose peer died never completes its collective: the caller then tells its control plane and calls orbx_replay_abort
int orbx_replay_wait_gathered_host(orbx_replay* r, int i, int timeout_ms) {
  if (!r || i < 0 || i > 1) return ORBX_E_INVALID;
  if (!r->gathered[i] || !r->gathered_valid[i]) return rfail(r, ORBX_E_INVALID, "orbx_replay_wait_gathered_host: no exchange has been queued into this buffer");
  RHIP(r, hipSetDevice(r->device));
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(r->gather_done[i]);
    if (e == hipSuccess) return ORBX_OK;
    if (e != hipErrorNotReady) return rfail(r, ORBX_E_DEVICE, std::string("orbx_replay_wait_gathered_host: ") + hipGetErrorString(e));
    if (timeout_ms >= 0 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double)timeout_ms)
      return rfail(r, ORBX_E_TIMEOUT, "orbx_replay_wait_gathered_host: the exchange into buffer " + std::to_string(i) + " did not complete within " +
                                          std::to_string(timeout_ms) + " ms (a peer that left?)");
    std::this_thread::yield();
  }
}

int orbx_replay_drain(orbx_replay* r) {
  if (!r) return ORBX_E_INVALID;
  RHIP(r, hipSetDevice(r->device));
  for (hipStream_t s : r->streams) RHIP(r, hipStreamSynchronize(s));
  RHIP(r, hipStreamSynchronize(r->gstream));
  r->pending[0] = r->pending[1] = false;   // nothing in flight any more
  return ORBX_OK;
}

int orbx_replay_block(orbx_replay* r, int i, uint8_t** d_block) {
  if (!r || !d_block || i < 0 || i > 1) return ORBX_E_INVALID;
  *d_block = r->blocks[i];
  return ORBX_OK;
}

int orbx_replay_gathered(orbx_replay* r, int i, int rank, const uint8_t** d_part) {
  if (!r || !d_part || i < 0 || i > 1 || rank < 0 || rank >= r->world) return ORBX_E_INVALID;
  if (!r->gathered[i]) return rfail(r, ORBX_E_INVALID, "orbx_replay_gathered: the engine was created without an exchange");
  *d_part = r->gathered[i] + (size_t)rank * r->send_bytes;
  return ORBX_OK;
}

int orbx_replay_read(orbx_replay* r, int what, int i, void* host_dst, size_t offset, size_t nbytes) {
  if (!r || !host_dst || i < 0 || i > 1 || (what != 0 && what != 1)) return ORBX_E_INVALID;
  const size_t total = what == 0 ? r->nbytes : r->send_bytes * (size_t)r->world;
  const uint8_t* src = what == 0 ? r->blocks[i] : r->gathered[i];
  if (!src || offset > total || nbytes > total - offset) return rfail(r, ORBX_E_INVALID, "orbx_replay_read: range outside the buffer");
  const int rc = orbx_replay_drain(r);
  if (rc != ORBX_OK) return rc;
  RHIP(r, hipMemcpyAsync(host_dst, src + offset, nbytes, hipMemcpyDeviceToHost, r->gstream));   // never the legacy stream (orbx_internal.h)
  RHIP(r, hipStreamSynchronize(r->gstream));
  return ORBX_OK;
}

int orbx_replay_write_block(orbx_replay* r, int i, const void* host_src, size_t offset, size_t nbytes) {
  if (!r || !host_src || i < 0 || i > 1 || offset > r->nbytes || nbytes > r->nbytes - offset) return ORBX_E_INVALID;
  const int rc = orbx_replay_drain(r);
  if (rc != ORBX_OK) return rc;
  RHIP(r, hipMemcpyAsync(r->blocks[i] + offset, host_src, nbytes, hipMemcpyHostToDevice, r->gstream));
  RHIP(r, hipStreamSynchronize(r->gstream));
  return ORBX_OK;
}

int orbx_replay_gather_ms(orbx_replay* r, double* avg_ms, long long* n, int reset) {
  if (!r) return ORBX_E_INVALID;
  const int rc = orbx_replay_drain(r);
  if (rc != ORBX_OK) return rc;
  for (int k = 0; k < kTimingPairs; k++) harvest(r, k);
  if (avg_ms) *avg_ms = r->t_n ? r->t_sum_ms / (double)r->t_n : -1.0;
  if (n) *n = r->t_n;
  if (reset) { r->t_sum_ms = 0; r->t_n = 0; }
  return ORBX_OK;
}


}  // extern "C"
