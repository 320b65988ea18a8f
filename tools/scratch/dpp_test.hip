#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int wave_scan_max_i32(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x111, 0xF, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x112, 0xF, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x114, 0xF, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x118, 0xF, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x142, 0xA, 0xF, false));
	v = max(v, __builtin_amdgcn_update_dpp((-2147483647 - 1), v, 0x143, 0xC, 0xF, false));
	return v;
}
__global__ void k(const int *in, int *out) { out[threadIdx.x] = wave_scan_max_i32(in[threadIdx.x]); }
int main() {
	int h[64], o[64], *di, *dq;
	for (int i = 0; i < 64; i++) h[i] = (i % 7 == 0) ? i : -1;
	hipMalloc(&di, 256); hipMalloc(&dq, 256);
	hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dq);
	hipMemcpy(o, dq, 256, hipMemcpyDeviceToHost);
	int bad = 0, run = -2147483648;
	for (int i = 0; i < 64; i++) { run = h[i] > run ? h[i] : run; if (o[i] != run) { bad++; printf("lane %d got %d want %d\n", i, o[i], run); } }
	printf("bad=%d\n", bad);
	return 0;
}
