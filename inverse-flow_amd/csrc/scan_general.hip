// General wavefront back-substitution scan (any C, any KH x KW, any H x W): fp32 VALU.
//
// One workgroup owns one image for all H+W-1 anti-diagonals (the reference launches one kernel
// + one cudaDeviceSynchronize per (diagonal, channel): inv_conv_with_bp_kernel_general.cu:97-126).
// With the folded taps (prep.hip) every (pixel, channel) of a diagonal is independent:
//     z_p = Wf[0] x_p - sum_{t>0} Wf[t] z_{p-t}
// LDS holds a ring of the last KH+KW-1 diagonals of z ([slot][h][c], c innermost, odd row
// pitch) and a double-buffered staging tile of x for the current/next diagonal.
// This is the fallback for shapes the MFMA scan does not cover; it is exact fp32.
#include "ifl_common.h"
#include "scan_general_body.h"

namespace ifl {

__global__ __launch_bounds__(256) void k_scan_general(const float *__restrict__ xin, const float *__restrict__ wf,
                                                      float *__restrict__ zout, Geom g, int rh, int rw,
                                                      const int *__restrict__ gate, int rf)
{
    if (gate && gate[blockIdx.x] == 0) return;
    extern __shared__ float smem[];
    scan_general_body<256>(xin, wf, zout, g, rh, rw, rf, smem, blockIdx.x, threadIdx.x);
}

size_t scan_general_lds_bytes(const Geom &g)
{
    const size_t Cp = (size_t)(g.C | 1);
    const size_t R = (size_t)(g.KH + g.KW - 1);
    return ((R + 2) * g.H * Cp + Cp) * sizeof(float);
}

int launch_scan_general(const float *x, const float *wf, float *z, const Geom &g, int rh, int rw, hipStream_t s,
                        const int *gate, int rf)
{
    const size_t lds = scan_general_lds_bytes(g);
    if (lds > 160 * 1024)
        IFL_FAIL(IFL_EUNSUPPORTED, "inverse scan: C=%d H=%d K=%dx%d needs %zu B of LDS (> 160 KiB)", g.C, g.H, g.KH,
                 g.KW, lds);
    static LdsOptIn opt_in;
    if (int rc = lds_opt_in(opt_in, (const void *)k_scan_general, 160 * 1024)) return rc;
    hipLaunchKernelGGL(k_scan_general, dim3(g.B), dim3(256), lds, s, x, wf, z, g, rh, rw, gate, rf);
    IFL_HIP(hipGetLastError());
    return IFL_OK;
}

} // namespace ifl
