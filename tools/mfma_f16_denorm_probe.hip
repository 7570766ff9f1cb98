#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float aval, float bval){
  int l = threadIdx.x;
  half8 a, b;
  for (int j=0;j<8;++j){ a[j] = (_Float16)0.f; b[j]=(_Float16)0.f; }
  // A[m=l&15][k=8*(l>>4)+j], B[k][n=l&15]; set A[m][0]=aval for all m (lanes 0..15, j=0), B[0][n]=bval
  if ((l>>4)==0){ a[0] = (_Float16)aval; b[0] = (_Float16)bval; }
  floatx4 acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  out[l*4+0]=acc[0]; out[l*4+1]=acc[1]; out[l*4+2]=acc[2]; out[l*4+3]=acc[3];
}
int main(){
  float* d; hipMalloc(&d, 256*4); float h[256];
  float tests[][2] = {{1.0f, 2.0f}, {9.5367431640625e-07f /*2^-20 subnormal*/, 1024.f}, {1024.f, 9.5367431640625e-07f}, {5.9604645e-08f /*2^-24 min subnormal*/, 1.f}, {3.0517578125e-05f/*2^-15 subnormal*/, 3.0517578125e-05f}};
  for (auto& t : tests){
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
    hipMemcpy(h, d, 256*4, hipMemcpyDeviceToHost);
    printf("a=%g b=%g -> D[0][0]=%g expected %g\n", t[0], t[1], h[0], (double)(float)(_Float16)t[0]*(double)(float)(_Float16)t[1]);
  }
  return 0;
}
