// Micro-probe: throughput of ds_read_b128 per CU for the scan's fragment addressing (8 waves reading).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(floatx4* out, unsigned long long* t, int iters){
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4, wave = tid >> 6;
  for (int i = tid * 16; i < 65536; i += 512 * 16) *(floatx4*)(lds + i) = floatx4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  unsigned base;
  if (MODE == 0) base = lane * 16;                         // linear: conflict-free by construction
  else if (MODE == 1) base = g * 256 + n * 16;              // scan ring: plane = k-group, row%16 inside
  else if (MODE == 2) base = n * 272 + g * 16;              // old [row][channel] layout with 16 B pad
  else base = (n * 2064) + g * 64;                          // xs-like rows
  base += (wave & 1) * 4096;
  floatx4 acc = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    floatx4 v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + base + (j % 4) * 1024 + (j / 4) * 8192;
      asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(a));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 12; ++j) acc += v[j];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[tid] = acc;
  if (tid == 0) t[0] = t1 - t0;
}
template <int MODE> void run(const char* name, floatx4* out, unsigned long long* t, int waves){
  int iters = 2000; unsigned long long h;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 65536, 0, out, t, iters);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 65536, 0, out, t, iters);
  (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-28s waves=%d  ticks per ds_read_b128 per CU = %.2f   (burst of 12 + wait: %.0f ticks)\n", name, waves,
         (double)h / (iters * 12.0 * waves), (double)h / iters);
}
int main(){
  floatx4* out; unsigned long long* t;
  (void)hipMalloc(&out, 512 * 16); (void)hipMalloc(&t, 8);
  for (int waves : {1, 4, 8}) {
    run<0>("linear lane*16", out, t, waves);
    run<1>("scan ring (g*256+n*16)", out, t, waves);
    run<2>("old [row][ch] n*272+g*16", out, t, waves);
    run<3>("xs rows n*2064+g*64", out, t, waves);
  }
  return 0;
}
