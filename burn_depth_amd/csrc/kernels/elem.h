// Storage element types of the MFMA operands (bf16_t / f16_t / float) and their conversions, shared by every kernel
// file. A kernel is a template over the storage type T; the host side picks the instantiation from md_precision with
// MD_BY_PREC. The fp32 -> T conversions are round-to-nearest-even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32); the f16 form
// SATURATES at +-65504 first (an infinity would poison every later sum; bf16 has the fp32 exponent range and needs none).
#pragma once

#include <type_traits>

#include "../md_common.h"

namespace md {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(2))) int i32x2_t;

template <typename T>
struct Native;  // the compiler's scalar type behind a 2-byte storage struct
template <>
struct Native<bf16_t> {
  typedef __bf16 type;
  typedef __attribute__((ext_vector_type(2))) __bf16 v2;
  typedef __attribute__((ext_vector_type(4))) __bf16 v4;
  typedef __attribute__((ext_vector_type(8))) __bf16 v8;
};
template <>
struct Native<f16_t> {
  typedef _Float16 type;
  typedef __attribute__((ext_vector_type(2))) _Float16 v2;
  typedef __attribute__((ext_vector_type(4))) _Float16 v4;
  typedef __attribute__((ext_vector_type(8))) _Float16 v8;
};

// the split-half element (MD_PREC_F16X2): an IEEE half in each of its two planes
template <>
struct Native<f16s_t> : Native<f16_t> {};
template <typename T>
struct is_split : std::false_type {};
template <>
struct is_split<f16s_t> : std::true_type {};
template <typename T>
struct is_half : std::integral_constant<bool, std::is_same<T, f16_t>::value || std::is_same<T, f16s_t>::value> {};
// planes per logical element of a tensor of T (host and device)
template <typename T>
constexpr int kPlanes = is_split<T>::value ? 2 : 1;

template <typename T>
__device__ __forceinline__ typename Native<T>::type cvt_elem(float v);
template <>
__device__ __forceinline__ __bf16 cvt_elem<bf16_t>(float v) {
  return (__bf16)v;
}
template <>
__device__ __forceinline__ _Float16 cvt_elem<f16_t>(float v) {
  return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
}

template <>
__device__ __forceinline__ _Float16 cvt_elem<f16s_t>(float v) {
  return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
}

// one element: store / load through fp32
template <typename T>
__device__ __forceinline__ void st1(T* p, float v) {
  if constexpr (sizeof(T) == 4)
    *(float*)p = v;
  else
    *(typename Native<T>::type*)p = cvt_elem<T>(v);
}
template <typename T>
__device__ __forceinline__ float ld1(const T* p) {
  if constexpr (sizeof(T) == 4)
    return *(const float*)p;
  else
    return (float)*(const typename Native<T>::type*)p;
}

// four / eight 2-byte elements packed into 8 / 16 bytes
template <typename T>
__device__ __forceinline__ i32x2_t pack4(f32x4_t v) {
  typename Native<T>::v4 b = {cvt_elem<T>(v[0]), cvt_elem<T>(v[1]), cvt_elem<T>(v[2]), cvt_elem<T>(v[3])};
  return __builtin_bit_cast(i32x2_t, b);
}
template <typename T>
__device__ __forceinline__ i32x4_t pack8(f32x4_t lo, f32x4_t hi) {
  typename Native<T>::v8 b = {cvt_elem<T>(lo[0]), cvt_elem<T>(lo[1]), cvt_elem<T>(lo[2]), cvt_elem<T>(lo[3]),
                              cvt_elem<T>(hi[0]), cvt_elem<T>(hi[1]), cvt_elem<T>(hi[2]), cvt_elem<T>(hi[3])};
  return __builtin_bit_cast(i32x4_t, b);
}
// two values -> one dword (the P fragment of the attention kernel: values in [0, 2^k], no saturation needed)
template <typename T>
__device__ __forceinline__ int pack2_nosat(float a, float b) {
  typename Native<T>::v2 v = {(typename Native<T>::type)a, (typename Native<T>::type)b};
  return __builtin_bit_cast(int, v);
}

// widen packed 2-byte elements to fp32. bf16 is a shift / mask of the dword (one VALU op per element), f16 a v_cvt.
template <typename T>
__device__ __forceinline__ void widen2(unsigned u, float& a, float& b) {
  if constexpr (std::is_same<T, bf16_t>::value) {  // (f16_t and f16s_t: v_cvt_f32_f16)
    a = __uint_as_float(u << 16);
    b = __uint_as_float(u & 0xffff0000u);
  } else {
    const typename Native<T>::v2 h = __builtin_bit_cast(typename Native<T>::v2, u);
    a = (float)h[0];
    b = (float)h[1];
  }
}
template <typename T>
__device__ __forceinline__ f32x4_t widen4(i32x2_t raw) {
  float a, b, c, d;
  widen2<T>((unsigned)raw[0], a, b);
  widen2<T>((unsigned)raw[1], c, d);
  return (f32x4_t){a, b, c, d};
}
template <typename T>
__device__ __forceinline__ void widen8(const i32x4_t& raw, f32x4_t& lo, f32x4_t& hi) {
  lo = widen4<T>((i32x2_t){raw[0], raw[1]});
  hi = widen4<T>((i32x2_t){raw[2], raw[3]});
}

// 4 / 8 consecutive elements <-> fp32 (global or LDS pointers, naturally aligned)
template <typename T>
__device__ __forceinline__ f32x4_t load4(const T* p) {
  if constexpr (sizeof(T) == 4)
    return *(const f32x4_t*)p;
  else
    return widen4<T>(*(const i32x2_t*)p);
}
template <typename T>
__device__ __forceinline__ void store4(T* p, f32x4_t v) {
  if constexpr (sizeof(T) == 4)
    *(f32x4_t*)p = v;
  else
    *(i32x2_t*)p = pack4<T>(v);
}
template <typename T>
__device__ __forceinline__ void store1(T* p, float v) {
  st1<T>(p, v);
}
template <typename T>
__device__ __forceinline__ void load8f(const T* p, float* v) {
  f32x4_t a, b;
  if constexpr (sizeof(T) == 4) {
    a = *(const f32x4_t*)p;
    b = *(const f32x4_t*)((const float*)p + 4);
  } else {
    widen8<T>(*(const i32x4_t*)p, a, b);
  }
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float* v) {
  const f32x4_t a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  if constexpr (sizeof(T) == 4) {
    *(f32x4_t*)p = a;
    *(f32x4_t*)((float*)p + 4) = b;
  } else {
    *(i32x4_t*)p = pack8<T>(a, b);
  }
}

// ---- split-half tensors (MD_PREC_F16X2): value = hi + lo, the lo plane `plane` elements behind the hi plane ----
// hi = f16(v) (saturating), lo = f16(v - hi): exact to 2^-22 relative while lo is a normal half (|v| >= 2^-3) and to
// 2^-25 absolute below that (lo subnormal; the f16 MFMAs and conversions keep subnormals: tools/probes/f16_denorm_probe.hip)
template <typename T>
__device__ __forceinline__ void split4(f32x4_t v, i32x2_t& hi, i32x2_t& lo) {
  hi = pack4<T>(v);
  lo = pack4<T>(v - widen4<T>(hi));
}
template <typename T>
__device__ __forceinline__ void split8(f32x4_t a, f32x4_t b, i32x4_t& hi, i32x4_t& lo) {
  hi = pack8<T>(a, b);
  f32x4_t ha, hb;
  widen8<T>(hi, ha, hb);
  lo = pack8<T>(a - ha, b - hb);
}
template <typename T>
__device__ __forceinline__ void store1s(T* p, long plane, float v) {
  const typename Native<T>::type h = cvt_elem<T>(v);
  *(typename Native<T>::type*)p = h;
  *(typename Native<T>::type*)(p + plane) = cvt_elem<T>(v - (float)h);
}
template <typename T>
__device__ __forceinline__ float load1s(const T* p, long plane) {
  return (float)*(const typename Native<T>::type*)p + (float)*(const typename Native<T>::type*)(p + plane);
}
template <typename T>
__device__ __forceinline__ void store4s(T* p, long plane, f32x4_t v) {
  i32x2_t hi, lo;
  split4<T>(v, hi, lo);
  *(i32x2_t*)p = hi;
  *(i32x2_t*)(p + plane) = lo;
}
template <typename T>
__device__ __forceinline__ f32x4_t load4s(const T* p, long plane) {
  return widen4<T>(*(const i32x2_t*)p) + widen4<T>(*(const i32x2_t*)(p + plane));
}
// plane-aware forms: `plane` is ignored (and costs nothing) for the one-plane types
template <typename T>
__device__ __forceinline__ void st1p(T* p, long plane, float v) {
  if constexpr (is_split<T>::value) store1s<T>(p, plane, v);
  else st1<T>(p, v);
}
template <typename T>
__device__ __forceinline__ float ld1p(const T* p, long plane) {
  if constexpr (is_split<T>::value) return load1s<T>(p, plane);
  else return ld1<T>(p);
}
template <typename T>
__device__ __forceinline__ void store4p(T* p, long plane, f32x4_t v) {
  if constexpr (is_split<T>::value) store4s<T>(p, plane, v);
  else store4<T>(p, v);
}
template <typename T>
__device__ __forceinline__ f32x4_t load4p(const T* p, long plane) {
  if constexpr (is_split<T>::value) return load4s<T>(p, plane);
  else return load4<T>(p);
}

// eight consecutive elements <-> fp32, plane-aware
template <typename T>
__device__ __forceinline__ void load8fp(const T* p, long plane, float* v) {
  load8f<T>(p, v);
  if constexpr (is_split<T>::value) {
    float l[8];
    load8f<T>(p + plane, l);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += l[i];
  }
}
template <typename T>
__device__ __forceinline__ void store8p(T* p, long plane, const float* v) {
  if constexpr (is_split<T>::value) {
    i32x4_t hi, lo;
    split8<T>((f32x4_t){v[0], v[1], v[2], v[3]}, (f32x4_t){v[4], v[5], v[6], v[7]}, hi, lo);
    *(i32x4_t*)p = hi;
    *(i32x4_t*)(p + plane) = lo;
  } else {
    store8<T>(p, v);
  }
}

// Host-side dispatch over the storage type of a precision mode: `T` is float, f16_t or bf16_t inside STMT.
// (MD_PREC_FP8 models store everything but the four ViT linear operands as bf16.)
#define MD_BY_PREC(prec, STMT)             \
  do {                                     \
    if ((prec) == MD_PREC_F32) {           \
      typedef float T;                     \
      STMT;                                \
    } else if ((prec) == MD_PREC_F16) {    \
      typedef ::md::f16_t T;               \
      STMT;                                \
    } else if ((prec) == MD_PREC_F16X2) {  \
      typedef ::md::f16s_t T;              \
      STMT;                                \
    } else {                               \
      typedef ::md::bf16_t T;              \
      STMT;                                \
    }                                      \
  } while (0)

}  // namespace md
