// Dependent-launch cost on this stack: a chain of small kernels launched into a stream, and the same chain captured into a hipGraph
// (MI355X, ROCm 7.2: 2.9 us vs 1.6 us per kernel).  build: hipcc -O2 --offload-arch=gfx950 tools/launch_gap.hip -o /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_small(unsigned *p, unsigned r) {
  // a little dependent work per wave, like a narrow round kernel
  unsigned v = p[(blockIdx.x * 64 + threadIdx.x) & 1023];
  if (v == 0xFFFFFFFFu) p[0] = r;
}
int main() {
  unsigned *d;
  CK(hipMalloc(&d, 4096));
  CK(hipMemset(d, 0, 4096));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int K = 96, REP = 40;
  for (int grid : {1, 1792}) {
    // plain launches
    for (int w = 0; w < 2; w++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < REP; r++)
        for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, d, (unsigned)i);
      CK(hipStreamSynchronize(s));
      auto t1 = std::chrono::steady_clock::now();
      if (w) printf("grid %d plain : %.2f us per kernel\n", grid, std::chrono::duration<double, std::micro>(t1 - t0).count() / (K * REP));
    }
    // graph
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < K; i++) hipLaunchKernelGGL(k_small, dim3(grid), dim3(256), 0, s, d, (unsigned)i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 2; w++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < REP; r++) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      auto t1 = std::chrono::steady_clock::now();
      if (w) printf("grid %d graph : %.2f us per kernel\n", grid, std::chrono::duration<double, std::micro>(t1 - t0).count() / (K * REP));
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
