// Shared host/device helpers for libmi_depth (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/mi_depth.h"

namespace md {

// ---- error plumbing ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
const char* get_error();

struct Error {
  int code;
};

#define MD_FAIL(code, ...)        \
  do {                            \
    ::md::set_error(__VA_ARGS__); \
    return (code);                \
  } while (0)

#define MD_HIP(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::md::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return MD_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

#define MD_TRY(expr)          \
  do {                        \
    int _s = (expr);          \
    if (_s != MD_OK) return _s; \
  } while (0)

// ---- fp8 (OCP e4m3fn) storage type: MFMA operands only ---------------------------------------
struct fp8_t {
  uint8_t bits;
};

// ---- bf16 / IEEE half storage types -----------------------------------------------------------
struct bf16_t {
  uint16_t bits;
};
struct f16_t {
  uint16_t bits;
};
// MD_PREC_F16X2: an element of a tensor kept as TWO IEEE-half planes, value = hi + lo with hi = f16(x), lo = f16(x - hi)
// (22 significant bits). A row of C logical channels is stored [hi: C | lo: C]; a product with an f16 weight is two
// v_mfma_f32_16x16x32_f16 (hi and lo against the same weight), three when the weight is itself kept as hi + lo.
struct f16s_t {
  uint16_t bits;
};

__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union {
    uint32_t u;
    float f;
  } c;
  c.u = (uint32_t)v.bits << 16;
  return c.f;
}

// Host-side RNE conversion (device code uses a plain cast to __bf16 -> v_cvt_pk_bf16_f32).
inline bf16_t f32_to_bf16_host(float f) {
  union {
    uint32_t u;
    float f;
  } c;
  c.f = f;
  uint32_t u = c.u;
  bf16_t r;
  if ((u & 0x7fffffffu) > 0x7f800000u) {  // NaN stays NaN
    r.bits = (uint16_t)((u >> 16) | 0x40);
    return r;
  }
  u = u + 0x7fffu + ((u >> 16) & 1u);
  r.bits = (uint16_t)(u >> 16);
  return r;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace md
