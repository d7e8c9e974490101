// Minimal device-runtime layer used by engine.cc.  Product build: HIP on gfx950.  The PPCSR_SIM
// branch exists only for tests/hostsim (CPU debugging of the kernel source) and is never compiled
// into libppcsr_hip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(PPCSR_SIM)
#include <chrono>
#include <cstdlib>

#include "sim_runtime.h"
extern int g_sim_fail_alloc;  // (tests/hostsim/sim_xchg.cpp: failure injection, the n-th device allocation from now fails)
namespace gpu {
typedef int stream_t;
inline const char *err_str(int) { return "sim error"; }
inline int set_device(int) { return 0; }
inline int device_count(int *n) { *n = 1; return 0; }
inline int device_cus(int, int *n) { *n = 0; return 0; }  // (the emulator has no residency to fill)
inline int stream_create(stream_t *s) { *s = 0; return 0; }
inline stream_t stream_from_ptr(void *) { return 0; }
inline int stream_destroy(stream_t) { return 0; }
inline int dmalloc(void **p, size_t b) {
  if (g_sim_fail_alloc > 0 && --g_sim_fail_alloc == 0) {
    *p = nullptr;
    return 2;
  }
  *p = ::malloc(b ? b : 1);
  return *p ? 0 : 2;
}
inline int dfree(void *p) { ::free(p); return 0; }
inline int dfree_named(void *p, const char *, int) { ::free(p); return 0; }
inline int hmalloc(void **p, size_t b) { *p = ::malloc(b ? b : 1); return *p ? 0 : 2; }
inline int hfree(void *p) { ::free(p); return 0; }
inline int h2d(void *d, const void *s, size_t b, stream_t) { memcpy(d, s, b); return 0; }
inline int d2h(void *d, const void *s, size_t b, stream_t) { memcpy(d, s, b); return 0; }
inline int d2d(void *d, const void *s, size_t b, stream_t) { memmove(d, s, b); return 0; }
inline int dset(void *d, int v, size_t b, stream_t) { memset(d, v, b); return 0; }
inline int sync(stream_t) { return 0; }
inline int last_error() { return 0; }
struct Timer {
  std::chrono::steady_clock::time_point a, b;
  int init() { return 0; }
  void destroy() {}
  void start(stream_t) { a = std::chrono::steady_clock::now(); }
  void stop(stream_t) { b = std::chrono::steady_clock::now(); }
  double ms() { return std::chrono::duration<double, std::milli>(b - a).count(); }
};
struct Event {
  std::chrono::steady_clock::time_point t;
  int init() { return 0; }
  void destroy() {}
  void record(stream_t) { t = std::chrono::steady_clock::now(); }
  static double elapsed_ms(Event &a, Event &b) { return std::chrono::duration<double, std::milli>(b.t - a.t).count(); }
};
}  // namespace gpu
#define GPU_LAUNCH(stream, kernel, grid, block, ...) \
  sim::launch((uint32_t)(grid), (uint32_t)(block), [=]() { kernel(__VA_ARGS__); })
#else
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
namespace gpu {
typedef hipStream_t stream_t;
inline int dbg_report(const char *what, int e) {
  if (e && getenv("PPCSR_DEBUG")) fprintf(stderr, "[ppcsr] %s -> %s\n", what, hipGetErrorString((hipError_t)e));
  return e;
}
inline const char *err_str(int e) { return hipGetErrorString((hipError_t)e); }
inline int set_device(int d) { return (int)hipSetDevice(d); }
inline int device_count(int *n) { return (int)hipGetDeviceCount(n); }
inline int device_cus(int d, int *n) { return (int)hipDeviceGetAttribute(n, hipDeviceAttributeMultiprocessorCount, d); }
inline int stream_create(stream_t *s) { return (int)hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
inline stream_t stream_from_ptr(void *p) { return (stream_t)p; }
inline int stream_destroy(stream_t s) { return dbg_report("hipStreamDestroy", (int)hipStreamDestroy(s)); }
inline int dmalloc(void **p, size_t b) { return (int)hipMalloc(p, b ? b : 1); }
inline int dfree(void *p) { return p ? dbg_report("hipFree", (int)hipFree(p)) : 0; }
inline int dfree_named(void *p, const char *what, int line) {
  if (!p) return 0;
  int e = (int)hipFree(p);
  if (e && getenv("PPCSR_DEBUG")) fprintf(stderr, "[ppcsr] hipFree(%s) at engine.cc:%d -> %s\n", what, line, hipGetErrorString((hipError_t)e));
  return e;
}
inline int hmalloc(void **p, size_t b) { return (int)hipHostMalloc(p, b ? b : 1, hipHostMallocDefault); }
inline int hfree(void *p) { return p ? dbg_report("hipHostFree", (int)hipHostFree(p)) : 0; }
inline int h2d(void *d, const void *s, size_t b, stream_t st) { return (int)hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, st); }
inline int d2h(void *d, const void *s, size_t b, stream_t st) { return (int)hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, st); }
inline int d2d(void *d, const void *s, size_t b, stream_t st) { return (int)hipMemcpyAsync(d, s, b, hipMemcpyDeviceToDevice, st); }
inline int dset(void *d, int v, size_t b, stream_t st) { return (int)hipMemsetAsync(d, v, b, st); }
inline int sync(stream_t st) { return (int)hipStreamSynchronize(st); }
inline int last_error() { return dbg_report("hipGetLastError", (int)hipGetLastError()); }
struct Timer {
  hipEvent_t a = nullptr, b = nullptr;
  int init() {
    int e = (int)hipEventCreate(&a);
    if (e) return e;
    return (int)hipEventCreate(&b);
  }
  void destroy() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
  void start(stream_t s) { (void)hipEventRecord(a, s); }
  void stop(stream_t s) { (void)hipEventRecord(b, s); }
  double ms() {
    float f = 0;
    (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&f, a, b);
    return (double)f;
  }
};
struct Event {
  hipEvent_t e = nullptr;
  int init() { return (int)hipEventCreate(&e); }
  void destroy() {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  void record(stream_t s) { (void)hipEventRecord(e, s); }
  static double elapsed_ms(Event &a, Event &b) {
    float f = 0;
    (void)hipEventElapsedTime(&f, a.e, b.e);
    return (double)f;
  }
};
}  // namespace gpu
#define GPU_LAUNCH(stream, kernel, grid, block, ...) \
  hipLaunchKernelGGL(kernel, dim3((uint32_t)(grid)), dim3((uint32_t)(block)), 0, stream, __VA_ARGS__)
#endif
