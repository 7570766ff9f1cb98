// Micro-probe: operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 (used by the weight fold).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double doublex4 __attribute__((ext_vector_type(4)));
__global__ void k(double *out)
{
    const int l = threadIdx.x;
    // assume A: lane l holds A[i = l%16][k = l/16], B: lane l holds B[k = l/16][j = l%16]
    const double a = 1000.0 * (l % 16) + (l / 16);       // A[i][k] = 1000 i + k
    const double b = ((l / 16) == 2) ? 1.0 + (l % 16) : 0.0; // B[k][j] = (k == 2) ? 1 + j : 0
    doublex4 c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    // expected D[i][j] = A[i][2] * (1 + j) = (1000 i + 2)(1 + j)
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main()
{
    double *d, h[256];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok_a = 1;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const double x = h[l * 4 + v];
            // decode (i, j) assuming the value pattern
            int found = 0;
            for (int i = 0; i < 16 && !found; ++i)
                for (int j = 0; j < 16 && !found; ++j)
                    if (x == (1000.0 * i + 2) * (1 + j)) {
                        found = 1;
                        if (l < 20 || l % 16 == 0) printf("lane %2d v %d -> D[%2d][%2d]\n", l, v, i, j);
                        if (!(i == 4 * (l / 16) + v && j == l % 16)) ok_a = 0;
                    }
            if (!found) { printf("lane %d v %d: value %g not decodable\n", l, v, x); ok_a = 0; }
        }
    printf("layout D[i = 4*(l/16)+v][j = l%%16] with A[l%%16][l/16], B[l/16][l%%16]: %s\n", ok_a ? "CONFIRMED" : "NO");
    fflush(stdout);
    return 0;
}
