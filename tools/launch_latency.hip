// Launch-to-visible latency on the GPU box: what a host thread waits for one small kernel whose result it needs (the shape of the matcher's
// window pass).  hipcc --offload-arch=gfx950 -O3 tools/launch_latency.hip -o /tmp/launch_latency && /tmp/launch_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <algorithm>

__global__ void k_flag(unsigned long long* done) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { __atomic_store_n(done, 1ull, __ATOMIC_RELEASE); __threadfence_system(); }
}
// nblocks workgroups each read `words` dwords of mapped host memory, the last one to finish raises the flag
__global__ void k_read_flag(const unsigned* src, int words, unsigned* ctr, unsigned long long* done, unsigned* sink) {
  unsigned acc = 0;
  for (int i = threadIdx.x; i < words; i += blockDim.x) acc += src[i];
  if (acc == 0xdeadbeefu) *sink = acc;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && atomicInc(ctr, gridDim.x - 1) == gridDim.x - 1) { __atomic_store_n(done, 1ull, __ATOMIC_RELEASE); __threadfence_system(); }
}

int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  unsigned char* h = nullptr; hipHostMalloc((void**)&h, 1 << 20, hipHostMallocMapped | hipHostMallocCoherent);
  unsigned char* hd = nullptr; hipHostGetDevicePointer((void**)&hd, h, 0);
  unsigned* d = nullptr; hipMalloc((void**)&d, 4096); hipMemset(d, 0, 4096);
  volatile unsigned long long* done = (volatile unsigned long long*)h;
  auto run = [&](const char* name, int mode, int idle_us) {
    std::vector<double> v;
    for (int it = 0; it < 220; it++) {
      if (idle_us) std::this_thread::sleep_for(std::chrono::microseconds(idle_us));
      *done = 0;
      const auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, (unsigned long long*)hd); hipStreamSynchronize(st); }
      else if (mode == 1) { hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, (unsigned long long*)hd); while (!__atomic_load_n(done, __ATOMIC_ACQUIRE)) __builtin_ia32_pause(); }
      else { hipLaunchKernelGGL(k_read_flag, dim3(192), dim3(256), 0, st, (const unsigned*)(hd + 4096), 12288, d, (unsigned long long*)hd, d + 16);
             while (!__atomic_load_n(done, __ATOMIC_ACQUIRE)) __builtin_ia32_pause(); }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (it >= 20) v.push_back(us);
    }
    std::sort(v.begin(), v.end());
    std::printf("%-78s idle %4d us between calls: median %6.1f us, p10 %6.1f, p90 %6.1f\n", name, idle_us, v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
  };
  for (int idle : {0, 100, 1000}) {
    run("empty kernel + hipStreamSynchronize", 0, idle);
    run("kernel raises a flag in mapped host memory, host polls", 1, idle);
    run("192 workgroups read 48 KB of mapped host memory each, last raises the flag, host polls", 2, idle);
  }
  return 0;
}
