// e3d_common.hpp -- shared host-side plumbing of libe3dhip.so (error state, device buffers,
// HIP call checking).  gfx950 only; no CUDA compatibility paths.
#pragma once

#include <cstdlib>

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace e3d {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& msg);

// Frees the device memory the library only keeps as a cache (the kNN workspace pool of e3d_normals.hip); true if anything
// was freed.  DevBuf calls it once when hipMalloc reports out-of-memory and retries.
bool release_cached_device_memory();

inline hipError_t malloc_with_retry(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipErrorOutOfMemory && release_cached_device_memory()) {
    (void)hipGetLastError();
    e = hipMalloc(p, bytes);
  }
  return e;
}

inline std::string fmt(const char* f, ...) {
  char buf[512];
  va_list ap; va_start(ap, f); vsnprintf(buf, sizeof buf, f, ap); va_end(ap);
  return std::string(buf);
}

#define E3D_HIP(expr)                                                                     \
  do {                                                                                    \
    hipError_t e3d_err__ = (expr);                                                        \
    if (e3d_err__ != hipSuccess)                                                          \
      throw ::e3d::Error(-3, ::e3d::fmt("%s failed: %s (%s:%d)", #expr,                   \
                                        hipGetErrorString(e3d_err__), __FILE__, __LINE__)); \
  } while (0)

// Owning device buffer (hipMalloc), grow-only reserve.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
  void reserve(size_t n) {
    if (n <= cap) return;
    release();
    E3D_HIP(malloc_with_retry((void**)&p, sizeof(T) * (n ? n : 1)));
    cap = n;
    if (poison_allocations()) { E3D_HIP(hipMemset(p, 0xFF, sizeof(T) * (n ? n : 1))); E3D_HIP(hipDeviceSynchronize()); }   // the library's streams do not wait for the null stream
  }
  // E3D_POISON=1: fresh device buffers are filled with 0xFF bytes (NaN as f32 / f64, -1 as integers) so that a kernel that
  // reads memory nobody wrote shows up in the parity tests instead of depending on what the allocator hands out
  static bool poison_allocations() {
    static const bool on = [] { const char* e = getenv("E3D_POISON"); return e && e[0] == '1'; }();
    return on;
  }
  // grow keeping contents
  void grow_keep(size_t n, size_t used, hipStream_t s) {
    if (n <= cap) return;
    T* q = nullptr;
    E3D_HIP(malloc_with_retry((void**)&q, sizeof(T) * n));
    if (poison_allocations()) { E3D_HIP(hipMemset(q, 0xFF, sizeof(T) * n)); E3D_HIP(hipDeviceSynchronize()); }
    if (p && used) E3D_HIP(hipMemcpyAsync(q, p, sizeof(T) * used, hipMemcpyDeviceToDevice, s));
    E3D_HIP(hipStreamSynchronize(s));
    if (p) (void)hipFree(p);
    p = q; cap = n;
  }
};

// Pinned host buffer.
template <typename T>
struct PinBuf {
  T* p = nullptr;
  size_t cap = 0;
  ~PinBuf() { if (p) (void)hipHostFree(p); }
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) (void)hipHostFree(p);
    E3D_HIP(hipHostMalloc((void**)&p, sizeof(T) * (n ? n : 1)));
    cap = n;
  }
};

// Copy n elements from a host-or-device pointer into device memory.
inline void copy_in(void* dst_dev, const void* src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return;
  E3D_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyDefault, s));
}
inline void copy_out(void* dst, const void* src_dev, size_t bytes, hipStream_t s) {
  if (bytes == 0) return;
  E3D_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDefault, s));
}

struct EventTimer {
  hipEvent_t a = nullptr, b = nullptr;
  EventTimer() { E3D_HIP(hipEventCreate(&a)); E3D_HIP(hipEventCreate(&b)); }
  ~EventTimer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  void start(hipStream_t s) { E3D_HIP(hipEventRecord(a, s)); }
  void stop(hipStream_t s) { E3D_HIP(hipEventRecord(b, s)); }
  float ms() { E3D_HIP(hipEventSynchronize(b)); float t = 0; E3D_HIP(hipEventElapsedTime(&t, a, b)); return t; }
};

inline size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }

// Streaming accesses (read or written once per launch, far larger than the caches): non-temporal, so that they do not displace the
// lines gathers hit in L2.  E3D_NT = 0 plain loads / stores everywhere, 1 only the correspondence planes of the LM passes, 2 every
// marked stream (measured per level: DESIGN 4.2).
#ifndef E3D_NT
#define E3D_NT 2
#endif
#ifdef __HIPCC__
template <int LEVEL = 2, typename T>
__device__ __forceinline__ T ld_stream(const T* __restrict__ p) {
  if constexpr (E3D_NT >= LEVEL) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int LEVEL = 2>
__device__ __forceinline__ float4 ld_stream(const float4* __restrict__ p) {
  if constexpr (E3D_NT >= LEVEL) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}
template <int LEVEL = 2, typename T>
__device__ __forceinline__ void st_stream(T* __restrict__ p, const T v) {
  if constexpr (E3D_NT >= LEVEL) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int LEVEL = 2>
__device__ __forceinline__ void st_stream(float4* __restrict__ p, const float4 v) {
  if constexpr (E3D_NT >= LEVEL) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<v4f*>(p));
  } else {
    *p = v;
  }
}
#endif

}  // namespace e3d
