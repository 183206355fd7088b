// micro-benchmark: practical HBM ceiling for an out-of-place streaming copy (what one lifting level is)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
__global__ __launch_bounds__(256) void copy16(const uint4 *__restrict__ a, uint4 *__restrict__ b, long n16)
{
	for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256)
		b[i] = a[i];
}
// the same copy with 8-byte and 4-byte accesses per lane (calibration of FETCH_SIZE / WRITE_SIZE for narrower accesses:
// the inverse lifting kernel loads its subbands 8 bytes per lane)
__global__ __launch_bounds__(256) void copy8(const uint2 *__restrict__ a, uint2 *__restrict__ b, long n8)
{
	for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256)
		b[i] = a[i];
}
__global__ __launch_bounds__(256) void copy4(const unsigned *__restrict__ a, unsigned *__restrict__ b, long n4)
{
	for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
		b[i] = a[i];
}
// row-strip pattern of the lifting kernels: a wave walks down `rows` rows of a pitch-`pitch16` image, 1 KB per row
__global__ __launch_bounds__(256) void copy_strips(const uint4 *__restrict__ a, uint4 *__restrict__ b, int pitch16, int rows, int h)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int x = blockIdx.x * 64 + lane;
	const int y0 = (blockIdx.y * 4 + wv) * rows;
	const long plane = (long)blockIdx.z * pitch16 * h;
	uint4 n0 = a[plane + (long)y0 * pitch16 + x], n1 = a[plane + (long)(y0 + 1) * pitch16 + x];
	uint4 m0 = a[plane + (long)(y0 + 2) * pitch16 + x], m1 = a[plane + (long)(y0 + 3) * pitch16 + x];
	for (int y = y0; y < y0 + rows; y += 2) {
		const uint4 c0 = n0, c1 = n1;
		n0 = m0; n1 = m1;
		if (y + 4 < y0 + rows) {
			m0 = a[plane + (long)(y + 4) * pitch16 + x];
			m1 = a[plane + (long)(y + 5) * pitch16 + x];
		}
		b[plane + (long)y * pitch16 + x] = c0;
		b[plane + (long)(y + 1) * pitch16 + x] = c1;
	}
}
int main()
{
	const long bytes = 1l << 30;   // 16 planes of 4096x4096 int32
	uint4 *a, *b;
	hipMalloc(&a, bytes); hipMalloc(&b, bytes);
	hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	auto time = [&](const std::string &name, auto launch) {
		launch(); hipDeviceSynchronize();
		hipEventRecord(e0); for (int r = 0; r < 10; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
		printf("%-34s %8.1f us  %6.2f TB/s (read+write)\n", name.c_str(), ms * 1e3, 2.0 * bytes / (ms * 1e-3) / 1e12);
	};
	for (int blocks : {2048, 8192, 32768})
		time("copy16 grid-stride blocks=" + std::to_string(blocks), [&] { hipLaunchKernelGGL(copy16, dim3(blocks), dim3(256), 0, 0, a, b, bytes / 16); });
	time("copy8 grid-stride blocks=8192", [&] { hipLaunchKernelGGL(copy8, dim3(8192), dim3(256), 0, 0, (const uint2 *)a, (uint2 *)b, bytes / 8); });
	time("copy4 grid-stride blocks=8192", [&] { hipLaunchKernelGGL(copy4, dim3(8192), dim3(256), 0, 0, (const unsigned *)a, (unsigned *)b, bytes / 4); });
	for (int rows : {16, 64, 256})
		time("copy_strips rows/wave=" + std::to_string(rows), [&] { hipLaunchKernelGGL(copy_strips, dim3(1024 / 64, 4096 / (4 * rows), 16), dim3(256), 0, 0, a, b, 1024, rows, 4096); });
	time("hipMemcpyAsync D2D", [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
	return 0;
}
