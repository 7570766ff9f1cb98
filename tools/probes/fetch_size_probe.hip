// Calibration of rocprofv3's FETCH_SIZE on gfx950 by read width: three kernels that each read the same 512 MiB exactly once
// (4, 16 and 2 x 16 bytes per lane per iteration -- the last is k_wgrad_mfma's operand access: eight consecutive floats per
// lane as two dwordx4 loads) and write 4 bytes per thread.  Run under  rocprofv3 --pmc FETCH_SIZE --kernel-trace  and compare
// the counter (KiB) with 524288 KiB.     hipcc --offload-arch=gfx950 -O3 -o fetch_size_probe fetch_size_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void read_b4(const float *p, size_t n, float *out)
{
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = s;
}
__global__ void read_b16(const f4 *p, size_t n4, float *out)
{
    f4 s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void read_b32(const f4 *p, size_t n8, float *out) // lane i: floats 8 i .. 8 i + 7
{
    f4 s = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) s += p[2 * i] + p[2 * i + 1];
    out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main()
{
    const size_t bytes = (size_t)512 << 20, n = bytes / 4;
    float *p, *out;
    hipMalloc(&p, bytes);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMemset(p, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_b4, dim3(4096), dim3(256), 0, 0, p, n, out);
        hipLaunchKernelGGL(read_b16, dim3(4096), dim3(256), 0, 0, (const f4 *)p, n / 4, out);
        hipLaunchKernelGGL(read_b32, dim3(4096), dim3(256), 0, 0, (const f4 *)p, n / 8, out);
    }
    hipDeviceSynchronize();
    printf("read 3 x 3 x %zu KiB\n", bytes / 1024);
    return 0;
}
