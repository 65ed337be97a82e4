// TEST INFRASTRUCTURE — and the example INTEGRATION.md section 8 points to: batch replay driven from a plain C++ host through include/orbx.h
// alone (no Python, no torch).  One rank: two lanes, the RCCL self-gather of whole blocks.  Prints, per step, position-weighted 64-bit sums
// of the feature block and of the gathered buffer and the step's keypoint count; tests/test_gpu_replay.py compares them with what the ctypes
// mirror of the same entry points produces on the same frames.
//   replay_host frames.u8 nframes rows cols steps        (frames.u8: nframes x rows x cols bytes)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbx.h"

static uint64_t digest(const std::vector<uint8_t>& b) {   // sum over 8-byte words w_k of w_k * (2k + 1), modulo 2^64 (the tail zero-padded)
  uint64_t h = 0;
  const size_t n = b.size() / 8;
  for (size_t k = 0; k < n; k++) { uint64_t w; std::memcpy(&w, &b[8 * k], 8); h += w * (2 * (uint64_t)k + 1); }
  if (b.size() % 8) { uint64_t w = 0; std::memcpy(&w, &b[8 * n], b.size() % 8); h += w * (2 * (uint64_t)n + 1); }
  return h;
}
#define CHECK(x) do { const int _rc = (x); if (_rc < 0) { std::fprintf(stderr, "%s -> %d (%s)\n", #x, _rc, eng ? orbx_replay_last_error(eng) : orbx_last_error(lanes[0])); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: replay_host frames.u8 nframes rows cols steps\n"); return 2; }
  const int nframes = std::atoi(argv[2]), rows = std::atoi(argv[3]), cols = std::atoi(argv[4]), steps = std::atoi(argv[5]);
  std::vector<uint8_t> host((size_t)nframes * rows * cols);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(host.data(), 1, host.size(), f) != host.size()) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  std::fclose(f);
  uint8_t* d_frames = nullptr;
  if (hipMalloc((void**)&d_frames, host.size()) != hipSuccess || hipMemcpy(d_frames, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return 3;

  // ---- the loop of INTEGRATION.md section 8
  orbx_ctx* lanes[2] = {nullptr, nullptr};
  orbx_replay* eng = nullptr;
  for (orbx_ctx*& c : lanes) if (orbx_create(&c, 1000, 1.2f, 8, 20, 7, 0) != ORBX_OK) { std::fprintf(stderr, "orbx_create failed\n"); return 3; }
  // one rank: no unique id, no host transport -> a one-rank RCCL group.  N ranks: rank 0 calls orbx_replay_unique_id(), every rank passes the 128 bytes
  CHECK(orbx_replay_create(&eng, lanes, 2, nframes, rows, cols, ORBX_GATHER_BLOCKS, /* rank */ 0, /* world */ 1, nullptr, nullptr, nullptr));
  std::printf("transport %s\n", orbx_replay_transport(eng));
  size_t block_bytes = 0, counts_off = 0, send_bytes = 0;
  CHECK(orbx_replay_layout(eng, nullptr, nullptr, &block_bytes, nullptr, &counts_off, nullptr, &send_bytes, nullptr));
  std::vector<uint8_t> block(block_bytes), gathered(send_bytes);
  for (int s = 0; s < steps; s++) {
    const int i = orbx_replay_step(eng, d_frames, (size_t)cols, (size_t)rows * cols, 0, 1000);   // asynchronous: returns the buffer index
    CHECK(i);
    CHECK(orbx_replay_read(eng, 0, i, block.data(), 0, block.size()));                            // (drains; a real host would consume on the device)
    CHECK(orbx_replay_read(eng, 1, i, gathered.data(), 0, gathered.size()));
    long long nkp = 0;
    for (int fr = 0; fr < nframes; fr++) { int32_t c[2]; std::memcpy(c, &block[counts_off + 8 * (size_t)fr], 8); nkp += c[0]; }
    std::printf("step %d buffer %d block %016llx gathered %016llx keypoints %lld\n", s, i, (unsigned long long)digest(block), (unsigned long long)digest(gathered), nkp);
  }
  double ms = 0; long long n = 0;
  CHECK(orbx_replay_gather_ms(eng, &ms, &n, 0));
  std::printf("gather_ms %.4f over %lld collectives\n", ms, n);
  // ---- the same loop the way a host with device code of its own runs it: NO drain; a consumer stream is ordered against each step's
  // exchange by orbx_replay_wait_gathered / _release_gathered and takes its copy of the gathered buffer while the engine runs on
  {
    hipStream_t consumer = nullptr;
    if (hipStreamCreateWithFlags(&consumer, hipStreamNonBlocking) != hipSuccess) return 3;
    std::vector<uint8_t*> d_copy(steps, nullptr);
    for (int s = 0; s < steps; s++) if (hipMalloc((void**)&d_copy[s], send_bytes) != hipSuccess) return 3;
    for (int s = 0; s < steps; s++) {
      const int i = orbx_replay_step(eng, d_frames, (size_t)cols, (size_t)rows * cols, 0, 1000);
      CHECK(i);
      const uint8_t* part = nullptr;
      CHECK(orbx_replay_gathered(eng, i, /* rank */ 0, &part));
      CHECK(orbx_replay_wait_gathered(eng, i, consumer));             // device-side wait for THIS step's collective
      if (hipMemcpyAsync(d_copy[s], part, send_bytes, hipMemcpyDeviceToDevice, consumer) != hipSuccess) return 3;   // "your own kernels"
      CHECK(orbx_replay_release_gathered(eng, i, consumer));          // step s + 2's collective into buffer i waits for this point
    }
    if (hipStreamSynchronize(consumer) != hipSuccess) return 3;
    int same = 0;
    for (int s = 0; s < steps; s++) {
      std::vector<uint8_t> c(send_bytes);
      if (hipMemcpy(c.data(), d_copy[s], send_bytes, hipMemcpyDeviceToHost) != hipSuccess) return 3;
      same += digest(c) == digest(gathered);                          // every step works on the same frames: every copy = the drained loop's last buffer
      (void)hipFree(d_copy[s]);
    }
    std::printf("consumer stream without drain: %d of %d copies equal the gathered buffer\n", same, steps);
    (void)hipStreamDestroy(consumer);
    if (same != steps) return 5;
  }
  orbx_replay_destroy(eng);
  for (orbx_ctx* c : lanes) orbx_destroy(c);
  (void)hipFree(d_frames);
  std::printf("ok\n");
  return 0;
}
