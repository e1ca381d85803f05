// pga_pk16.h -- two int16 lanes per VGPR (v_pk_*_i16): the DP kernels compute two target columns per instruction
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pga {

typedef short s2_t __attribute__((ext_vector_type(2)));
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int as_i(s2_t v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ s2_t as_s2(int v) { return __builtin_bit_cast(s2_t, v); }
__device__ __forceinline__ s2_t splat2(int x) { s2_t r; r.x = (short)x; r.y = (short)x; return r; }
__device__ __forceinline__ s2_t pmax(s2_t a, s2_t b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ s2_t pmin(s2_t a, s2_t b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s2_t pminu(s2_t a, s2_t b) { return __builtin_bit_cast(s2_t, __builtin_elementwise_min(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b))); }
__device__ __forceinline__ int bfi(int mask, int a, int b) { return (a & mask) | (b & ~mask); }   // mask ? a : b, bitwise

// two sign-extended bytes (a 16-bit LDS load) -> two int16 halves
__device__ __forceinline__ s2_t unpack_i8x2(uint32_t w) { return as_s2((int)((w & 0xffu) | (w & 0xff00u) << 8)) << 8 >> 8; }
// the low bytes of the two halves -> 16 bits for a 2-byte store
__device__ __forceinline__ uint16_t pack_i8x2(s2_t v) { const uint32_t w = (uint32_t)as_i(v); return (uint16_t)((w & 0xffu) | (w >> 8 & 0xff00u)); }
__device__ __forceinline__ s2_t pack2(int lo, int hi) { return as_s2((int)(((uint32_t)lo & 0xffffu) | (uint32_t)hi << 16)); }

} // namespace pga
