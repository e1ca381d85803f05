// pga_wave.h -- wave64 cross-lane primitives built on DPP moves (no LDS traffic, a few cycles per step).
// gfx950 keeps the GCN DPP controls used here: quad_perm, row_shr, row_mirror, row_half_mirror, row_bcast15/31, wave_shr.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pga {

// value of lane `l` (uniform), as a scalar broadcast
__device__ __forceinline__ int32_t rl(int32_t v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
// lane i receives lane i-1's value, lane 0 receives `first` (one DPP move: wave_shr:1)
__device__ __forceinline__ int32_t wave_shr1(int32_t v, int32_t first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); }

// minimum of a double over the wave with DPP moves only (no LDS traffic); the result is uniform
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_min_step(double v)
{
	const long long b = __double_as_longlong(v);
	const int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
	const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
	const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
	const double o = __longlong_as_double(((long long)ohi << 32) | (unsigned)olo);
	return o < v ? o : v;
}
__device__ __forceinline__ double wave_min_f64(double v)
{
	v = dpp_min_step<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
	v = dpp_min_step<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
	v = dpp_min_step<0x141, 0xf>(v);     // row_half_mirror
	v = dpp_min_step<0x140, 0xf>(v);     // row_mirror: every lane of a 16-lane row holds the row minimum
	v = dpp_min_step<0x142, 0xa>(v);     // row_bcast15 into rows 1 and 3
	v = dpp_min_step<0x143, 0xc>(v);     // row_bcast31 into rows 2 and 3: lane 63 holds the wave minimum
	const long long b = __double_as_longlong(v);
	const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
	return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_minu_step(uint32_t v)
{
	const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
	return o < v ? o : v;
}
// unsigned minimum over the wave, returned to every lane
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
	v = dpp_minu_step<0xB1, 0xf>(v);
	v = dpp_minu_step<0x4E, 0xf>(v);
	v = dpp_minu_step<0x141, 0xf>(v);
	v = dpp_minu_step<0x140, 0xf>(v);
	v = dpp_minu_step<0x142, 0xa>(v);
	v = dpp_minu_step<0x143, 0xc>(v);
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_maxu_step(uint32_t v)
{
	const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
	return o > v ? o : v;
}
// unsigned maximum over the wave, returned to every lane
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
	v = dpp_maxu_step<0xB1, 0xf>(v);
	v = dpp_maxu_step<0x4E, 0xf>(v);
	v = dpp_maxu_step<0x141, 0xf>(v);
	v = dpp_maxu_step<0x140, 0xf>(v);
	v = dpp_maxu_step<0x142, 0xa>(v);
	v = dpp_maxu_step<0x143, 0xc>(v);
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// minimum of a double over the wave (no NaNs), to every lane: two 32-bit reductions over the order-preserving integer image
__device__ __forceinline__ double wave_min_f64_key(double v)
{
	const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
	const unsigned long long key = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ULL);
	const uint32_t khi = (uint32_t)(key >> 32), klo = (uint32_t)key;
	const uint32_t mhi = wave_min_u32(khi);
	const uint32_t mlo = wave_min_u32(khi == mhi ? klo : 0xffffffffu);
	const unsigned long long mk = ((unsigned long long)mhi << 32) | mlo;
	return __longlong_as_double((long long)((mk >> 63) ? (mk & 0x7fffffffffffffffULL) : ~mk));
}

// inclusive prefix maximum over the wave (lane order), DPP only
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int32_t dpp_max_step(int32_t v)
{
	const int32_t o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
	return o > v ? o : v;
}
__device__ __forceinline__ int32_t wave_prefix_max_incl(int32_t v)
{
	v = dpp_max_step<0x111, 0xf>(v);     // row_shr:1
	v = dpp_max_step<0x112, 0xf>(v);     // row_shr:2
	v = dpp_max_step<0x114, 0xf>(v);     // row_shr:4
	v = dpp_max_step<0x118, 0xf>(v);     // row_shr:8
	v = dpp_max_step<0x142, 0xa>(v);     // row_bcast15
	v = dpp_max_step<0x143, 0xc>(v);     // row_bcast31
	return v;
}

template <int CTRL, int ROW_MASK> __device__ __forceinline__ int32_t dpp_min_step_i(int32_t v)
{
	const int32_t o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
	return o < v ? o : v;
}
__device__ __forceinline__ int32_t wave_min_i32(int32_t v)
{
	v = dpp_min_step_i<0xB1, 0xf>(v); v = dpp_min_step_i<0x4E, 0xf>(v); v = dpp_min_step_i<0x141, 0xf>(v); v = dpp_min_step_i<0x140, 0xf>(v);
	v = dpp_min_step_i<0x142, 0xa>(v); v = dpp_min_step_i<0x143, 0xc>(v);
	return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) { return -wave_min_i32(-v); }   // y is a sequence coordinate: never INT32_MIN

// inclusive prefix sum over the wave (lane order), DPP only
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_add_step(uint32_t v)
{
	return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_prefix_sum_incl(uint32_t v)
{
	v = dpp_add_step<0x111, 0xf>(v);     // row_shr:1
	v = dpp_add_step<0x112, 0xf>(v);     // row_shr:2
	v = dpp_add_step<0x114, 0xf>(v);     // row_shr:4
	v = dpp_add_step<0x118, 0xf>(v);     // row_shr:8
	v = dpp_add_step<0x142, 0xa>(v);     // row_bcast15
	v = dpp_add_step<0x143, 0xc>(v);     // row_bcast31
	return v;
}

// maximum of a signed 64-bit key over the wave; the result is uniform
template <int CTRL, int ROW_MASK> __device__ __forceinline__ long long dpp_max_step_i64(long long v)
{
	const int lo = (int)(v & 0xffffffffLL), hi = (int)(v >> 32);
	const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
	const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
	const long long o = ((long long)ohi << 32) | (unsigned)olo;
	return o > v ? o : v;
}
__device__ __forceinline__ long long wave_max_i64(long long v)
{
	v = dpp_max_step_i64<0xB1, 0xf>(v); v = dpp_max_step_i64<0x4E, 0xf>(v); v = dpp_max_step_i64<0x141, 0xf>(v); v = dpp_max_step_i64<0x140, 0xf>(v);
	v = dpp_max_step_i64<0x142, 0xa>(v); v = dpp_max_step_i64<0x143, 0xc>(v);
	const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
	return ((long long)hi << 32) | (unsigned)lo;
}

} // namespace pga
