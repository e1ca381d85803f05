// micro-benchmarks of the per-iteration costs that bound the latency-bound kernels (one workgroup, NT threads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void k(int n, int *out, int dummy)
{
	__shared__ int box[2][16];
	__shared__ int arr[4096];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	int acc = tid, acc2 = dummy;
	for (int i = tid; i < 4096; i += blockDim.x) arr[i] = i;
	__syncthreads();
	for (int r = 0; r < n; ++r) {
		if (MODE == 0) {                    // barrier only
			asm volatile("s_barrier" ::: "memory");
		} else if (MODE == 1) {             // LDS write -> barrier -> LDS read of another wave's value
			if (lane == 0) box[r & 1][wave] = acc;
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
			acc += box[r & 1][(wave + 1) & 3];
		} else if (MODE == 2) {             // 200 dependent VALU adds
#pragma unroll
			for (int k2 = 0; k2 < 200; ++k2) acc = acc * 3 + k2;
		} else if (MODE == 3) {             // 200 independent-ish VALU (two chains)
#pragma unroll
			for (int k2 = 0; k2 < 100; ++k2) { acc = acc * 3 + k2; acc2 = acc2 * 5 + k2; }
		} else if (MODE == 4) {             // dependent LDS reads (pointer chase), 8 per iteration
#pragma unroll
			for (int k2 = 0; k2 < 8; ++k2) acc = arr[(acc + k2) & 4095];
		} else if (MODE == 5) {             // 20 divergent exec branches that no lane takes
#pragma unroll
			for (int k2 = 0; k2 < 20; ++k2) if (acc == -12345 - k2) { arr[k2] = acc; acc += 7; }
			acc += r;
		} else if (MODE == 6) {             // 20 uniform scalar branches, alternately taken
			int s = __builtin_amdgcn_readfirstlane(r);
#pragma unroll
			for (int k2 = 0; k2 < 20; ++k2) { if ((s >> (k2 & 3)) & 1) acc = acc * 3 + 1; else acc = acc + 5; s += 3; }
		} else if (MODE == 7) {             // 200 packed 16-bit dependent ops
			typedef short s2 __attribute__((ext_vector_type(2)));
			s2 a = __builtin_bit_cast(s2, acc), b = __builtin_bit_cast(s2, acc2);
#pragma unroll
			for (int k2 = 0; k2 < 100; ++k2) { a = a + b; b = __builtin_elementwise_max(a, b); }
			acc = __builtin_bit_cast(int, a); acc2 = __builtin_bit_cast(int, b);
		} else if (MODE == 8) {             // global store fire-and-forget + barrier
			out[1024 + ((r * 256 + tid) & 0xfffff)] = acc;
			asm volatile("s_barrier" ::: "memory");
		} else if (MODE == 9) {             // DPP wave reduction (6 steps) x 4
#pragma unroll
			for (int k2 = 0; k2 < 4; ++k2) {
				int v = acc;
				v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
				v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
				v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
				v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
				v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
				v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
				acc += __builtin_amdgcn_readlane(v, 63);
			}
		}
	}
	out[tid] = acc + acc2;
}

template <int MODE> void run(const char *name, int nt, int n, int *d)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(nt), 0, 0, 100, d, 1);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(nt), 0, 0, n, d, 1);
	CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	printf("%-60s nt=%4d  %8.1f ns per iteration\n", name, nt, ms * 1e6 / n);
}

int main()
{
	int *d; CK(hipMalloc(&d, (1 << 22) + 8192));
	const int n = 200000;
	for (int nt : {64, 256, 1024}) {
		run<0>("s_barrier", nt, n, d);
		run<1>("LDS write -> barrier -> LDS read", nt, n, d);
		run<2>("200 dependent VALU (mad)", nt, n, d);
		run<3>("200 VALU in two chains", nt, n, d);
		run<4>("8 dependent LDS reads", nt, n, d);
		run<5>("20 exec branches, none taken", nt, n, d);
		run<6>("20 uniform branches", nt, n, d);
		run<7>("200 dependent packed 16-bit ops", nt, n, d);
		run<8>("global store + barrier", nt, n, d);
		run<9>("4 DPP wave max reductions", nt, n, d);
	}
	return 0;
}
