// hip_emu.h -- barrier-synchronous HIP kernels on the host (development and CPU tests; g++ -std=c++17 -DPGA_EMU).
//
// There is no GPU in the build container.  A kernel that only uses threadIdx / blockIdx / blockDim, __shared__ arrays, __syncthreads() and
// atomicAdd / atomicMax / atomicOr (no wave intrinsics, no inline assembly) means the same thing when every thread of a workgroup is a FIBER and
// __syncthreads() hands control to the next one: a workgroup runs phase by phase (every thread up to its next barrier, then every thread up to the
// one after), workgroups run one after the other, `__shared__` becomes a function-local static (one workgroup at a time, contents undefined at
// kernel entry as on the device).  This checks the LOGIC of such a kernel -- indexing, barrier placement (threads of a workgroup that leave a kernel
// while others wait at a barrier are reported), what the result depends on -- not its speed, and not races inside a phase (fibers do not overlap; the
// order of atomics inside a phase is one of the orders the device may take).
// Use: #include "hip_emu.h" before the kernel header, then  emu_launch(dim3(grid), dim3(block), [&] { kernel(args...); });
#pragma once
#include <ucontext.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __shared__ static
#define __launch_bounds__(...)

namespace emu {
struct Fiber { ucontext_t ctx; std::vector<char> stack; bool done = false, at_barrier = false; };
inline ucontext_t main_ctx;
inline Fiber *cur = nullptr;
inline std::function<void()> *body = nullptr;
inline void entry() { (*body)(); cur->done = true; swapcontext(&cur->ctx, &main_ctx); }
}

inline void __syncthreads() { emu::cur->at_barrier = true; swapcontext(&emu::cur->ctx, &emu::main_ctx); }
template <class T> inline T atomicAdd(T *p, T v) { const T o = *p; *p = o + v; return o; }
template <class T> inline T atomicOr(T *p, T v) { const T o = *p; *p = o | v; return o; }
template <class T> inline T atomicMax(T *p, T v) { const T o = *p; if (o < v) *p = v; return o; }

template <class F> inline void emu_launch(dim3 grid, dim3 block, F kernel_call)
{
	gridDim = grid; blockDim = block;
	std::function<void()> fn = kernel_call;
	emu::body = &fn;
	std::vector<emu::Fiber> fb(block.x);
	for (emu::Fiber &f : fb) f.stack.resize(256 * 1024);
	for (unsigned b = 0; b < grid.x; ++b) {
		blockIdx = dim3(b);
		for (emu::Fiber &f : fb) {
			getcontext(&f.ctx);
			f.ctx.uc_stack.ss_sp = f.stack.data(); f.ctx.uc_stack.ss_size = f.stack.size(); f.ctx.uc_link = nullptr;
			makecontext(&f.ctx, (void (*)())emu::entry, 0);
			f.done = false;
		}
		for (unsigned alive = block.x; alive;) {
			unsigned waiting = 0, left = 0;
			for (unsigned t = 0; t < block.x; ++t) {
				emu::Fiber &f = fb[t];
				if (f.done) continue;
				threadIdx = dim3(t); emu::cur = &f; f.at_barrier = false;
				swapcontext(&emu::main_ctx, &f.ctx);
				if (f.done) ++left; else ++waiting;
			}
			if (waiting && left) { fprintf(stderr, "hip_emu: workgroup %u: %u threads left the kernel while %u wait at a barrier\n", b, left, waiting); abort(); }
			alive = waiting;
		}
	}
	emu::body = nullptr; emu::cur = nullptr;
}
