// micro-benchmark: why do the 16 B/element table kernels only reach ~1.2 TB/s?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void add_quad(unsigned long long *cs, unsigned *ct, unsigned *cg, long n)
{
	const long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 1024 + 4 * threadIdx.x;
	if (i >= n) return;
	ulonglong2 a = *(ulonglong2 *)(cs + i), b = *(ulonglong2 *)(cs + i + 2);
	uint4 t = *(uint4 *)(ct + i), g = *(uint4 *)(cg + i);
	a.x += 1; a.y += 1; b.x += 1; b.y += 1; t.x += 1; t.y += 1; t.z += 1; t.w += 1; g.x += 1; g.y += 1; g.z += 1; g.w += 1;
	*(ulonglong2 *)(cs + i) = a; *(ulonglong2 *)(cs + i + 2) = b; *(uint4 *)(ct + i) = t; *(uint4 *)(cg + i) = g;
}
__global__ __launch_bounds__(256) void add_stride(unsigned long long *cs, unsigned *ct, unsigned *cg, long n)
{
	for (long i = 4 * ((long)blockIdx.x * 256 + threadIdx.x); i < n; i += 4l * gridDim.x * 256) {
		ulonglong2 a = *(ulonglong2 *)(cs + i), b = *(ulonglong2 *)(cs + i + 2);
		uint4 t = *(uint4 *)(ct + i), g = *(uint4 *)(cg + i);
		a.x += 1; a.y += 1; b.x += 1; b.y += 1; t.x += 1; t.y += 1; t.z += 1; t.w += 1; g.x += 1; g.y += 1; g.z += 1; g.w += 1;
		*(ulonglong2 *)(cs + i) = a; *(ulonglong2 *)(cs + i + 2) = b; *(uint4 *)(ct + i) = t; *(uint4 *)(cg + i) = g;
	}
}
__global__ __launch_bounds__(256) void add_one(uint4 *p, long n16)
{
	for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
		uint4 v = p[i]; v.x += 1; v.y += 1; v.z += 1; v.w += 1; p[i] = v;
	}
}
__global__ __launch_bounds__(256) void read_only(const uint4 *p, long n16, unsigned *sink)
{
	unsigned acc = 0;
	for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
		uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345) *sink = acc;
}
int main()
{
	const long n = 16l * 2 * 524292;   // elements
	unsigned long long *cs; unsigned *ct, *cg, *sink;
	CK(hipMalloc(&cs, n * 8)); CK(hipMalloc(&ct, n * 4)); CK(hipMalloc(&cg, n * 4)); CK(hipMalloc(&sink, 4));
	CK(hipMemset(cs, 0, n * 8)); CK(hipMemset(ct, 0, n * 4)); CK(hipMemset(cg, 0, n * 4));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto time = [&](const char *name, auto launch, double bytes) {
		launch(); hipDeviceSynchronize();
		hipEventRecord(e0); for (int r = 0; r < 10; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
		printf("%-28s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
		return 0;
	};
	const double b = 32.0 * n;
	time("add_quad grid(513,32)", [&] { hipLaunchKernelGGL(add_quad, dim3(513, 32), dim3(256), 0, 0, cs, ct, cg, n); }, b);
	time("add_quad grid(16416,1)", [&] { hipLaunchKernelGGL(add_quad, dim3(16416, 1), dim3(256), 0, 0, cs, ct, cg, n); }, b);
	for (int blocks : {1024, 2048, 4096, 8192})
		time(("add_stride blocks=" + std::to_string(blocks)).c_str(), [&] { hipLaunchKernelGGL(add_stride, dim3(blocks), dim3(256), 0, 0, cs, ct, cg, n); }, b);
	for (int blocks : {2048, 8192})
		time(("add_one(cs) blocks=" + std::to_string(blocks)).c_str(), [&] { hipLaunchKernelGGL(add_one, dim3(blocks), dim3(256), 0, 0, (uint4 *)cs, n / 2); }, 16.0 * n);
	time("read_only(cs) 8192", [&] { hipLaunchKernelGGL(read_only, dim3(8192), dim3(256), 0, 0, (const uint4 *)cs, n / 2, sink); }, 8.0 * n);
	time("memset cs", [&] { hipMemsetAsync(cs, 0, n * 8, 0); }, 8.0 * n);
	return 0;
}
