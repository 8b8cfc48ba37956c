// How many streams of one process run kernels concurrently on this runtime?  N streams, each a chain of K short kernels (G workgroups
// spinning for T us); wall time for N = 1..8.  Also the same chains as one captured single-stream graph per stream, replayed.
//   hipcc --offload-arch=gfx950 -O2 tools/stream_conc_probe.hip -o tools/bin/stream_conc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned* sink) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long now = t0;
  int guard = 0;
  while (now - t0 < ticks && guard < (1 << 22)) { now = __builtin_readcyclecounter(); ++guard; }   // s_memtime: 100 MHz on gfx950
  if (ticks == 0xffffffffffffffffull) *sink = (unsigned)now;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 300;        // kernels per chain
  const int G = argc > 2 ? atoi(argv[2]) : 64;         // workgroups per kernel
  const int us = argc > 3 ? atoi(argv[3]) : 20;        // spin per kernel
  unsigned* sink; CHECK(hipMalloc(&sink, 4));
  const unsigned long long ticks = (unsigned long long)us * 100;   // 100 MHz
  std::vector<hipStream_t> st(8);
  for (auto& s : st) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<hipGraphExec_t> ge(8, nullptr);
    if (mode == 1)
      for (int i = 0; i < 8; ++i) {
        hipGraph_t g;
        CHECK(hipStreamBeginCapture(st[i], hipStreamCaptureModeRelaxed));
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), 0, st[i], ticks, sink);
        CHECK(hipStreamEndCapture(st[i], &g));
        CHECK(hipGraphInstantiate(&ge[i], g, nullptr, nullptr, 0));
        CHECK(hipGraphDestroy(g));
      }
    for (int n = 1; n <= 8; ++n) {
      double best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        if (mode == 0) {
          for (int k = 0; k < K; ++k)
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), 0, st[i], ticks, sink);
        } else {
          for (int i = 0; i < n; ++i) CHECK(hipGraphLaunch(ge[i], st[i]));
        }
        CHECK(hipDeviceSynchronize());
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
      printf("%s  %d stream(s) x %d kernels of %d us (%d workgroups): %.2f ms  (one chain alone would take >= %.2f)\n", mode ? "graph" : "eager", n, K, us, G, best,
             K * us * 1e-3);
    }
  }
  return 0;
}
