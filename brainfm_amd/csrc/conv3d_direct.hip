// Direct 3x3x3 convolution, exact fp32 FMA, any channel count.
//
// SingleConv 'gcl' body after the statistics: folded GroupNorm affine on load,
// zero padding after the affine, conv (no bias), LeakyReLU
// (Trainer/models/unet3d/buildingblocks.py:31-60).  Used for the Cin=1 stem
// (HBM/L2-bound: K = 27 is not a GEMM) and for widths the MFMA family does not
// take; also serves as the exact-fp32 device reference for the MFMA kernel.
//
// One thread = one output voxel x 8 output channels; blockIdx.y = channel
// group, so the weight addresses are wave-uniform (scalar loads).
#include "bfm_common.h"

namespace {

constexpr int TPB = 256;
constexpr int COT = 8;

__global__ void pack_direct(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ out) {
    // out[tap][ci][co] = w[co][ci][tap]
    int64_t n = (int64_t)27 * Cin * Cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int co = (int)(i % Cout);
        int64_t r = i / Cout;
        int ci = (int)(r % Cin);
        int tap = (int)(r / Cin);
        out[i] = w[((int64_t)co * Cin + ci) * 27 + tap];
    }
}

__global__ void __launch_bounds__(TPB) conv_direct(const float* __restrict__ A, int CA, const float* __restrict__ B,
                                                   int CB, int D, int H, int W, UpView up,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const float* __restrict__ wp, int Cout, float slope,
                                                   float* __restrict__ out) {
    const int64_t nvox = (int64_t)D * H * W;
    const int64_t v = (int64_t)blockIdx.x * TPB + threadIdx.x;
    const int co0 = blockIdx.y * COT;
    const int Cin = CA + CB;
    const bool live = v < nvox;
    const int64_t vv = live ? v : 0;
    const int x = (int)(vv % W);
    const int64_t t = vv / W;
    const int y = (int)(t % H);
    const int z = (int)(t / H);

    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = 0.f;

    for (int kd = 0; kd < 3; ++kd) {
        for (int kh = 0; kh < 3; ++kh) {
            for (int kw = 0; kw < 3; ++kw) {
                const int zz = z + kd - 1, yy = y + kh - 1, xx = x + kw - 1;
                const bool inb = live && zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
                const int tap = (kd * 3 + kh) * 3 + kw;
                const float* wt = wp + (size_t)tap * Cin * Cout + co0;
                int64_t ia = 0, ib = 0;
                if (inb) {
                    ia = (((int64_t)zz * H + yy) * W + xx) * CA;
                    if (CB > 0) ib = (((int64_t)up.mapD[zz] * up.h + up.mapH[yy]) * up.w + up.mapW[xx]) * CB;
                }
                for (int ci = 0; ci < Cin; ++ci) {
                    float xv = 0.f;
                    if (inb) {
                        float raw = ci < CA ? A[ia + ci] : B[ib + (ci - CA)];
                        xv = fmaf(raw, scale[ci], shift[ci]);
                    }
                    const float* wr = wt + (size_t)ci * Cout;
#pragma unroll
                    for (int j = 0; j < COT; ++j) {
                        float wv = (co0 + j < Cout) ? wr[j] : 0.f;
                        acc[j] = fmaf(xv, wv, acc[j]);
                    }
                }
            }
        }
    }
    if (!live) return;
    float* o = out + v * Cout + co0;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        if (co0 + j < Cout) {
            float r = acc[j];
            o[j] = r >= 0.f ? r : r * slope;
        }
    }
}

}  // namespace

extern "C" size_t bfm_pack_conv_weights_direct_bytes(int Cin, int Cout) {
    return (size_t)27 * Cin * Cout * sizeof(float);
}

extern "C" int bfm_pack_conv_weights_direct(const float* w, int Cin, int Cout, float* wpacked, bfm_stream_t stream) {
    if (!w || !wpacked || Cin <= 0 || Cout <= 0) return BFM_E_ARG;
    int64_t n = (int64_t)27 * Cin * Cout;
    int nb = (int)std::min<int64_t>(2048, bfm_cdiv64(n, 256));
    hipLaunchKernelGGL(pack_direct, dim3(nb), dim3(256), 0, bfm_s(stream), w, Cin, Cout, wpacked);
    return bfm_launch_status();
}

extern "C" int bfm_conv3x3x3_direct(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                                    const bfm_upsample_t* up, const float* scale, const float* shift,
                                    const float* wpacked, int Cout, float slope, float* out, bfm_stream_t stream) {
    if (!A || CA <= 0 || D <= 0 || H <= 0 || W <= 0 || !scale || !shift || !wpacked || Cout <= 0 || !out)
        return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!B || !up || !up->mapD || !up->mapH || !up->mapW))) return BFM_E_ARG;
    int64_t nvox = (int64_t)D * H * W;
    dim3 grid((unsigned)bfm_cdiv64(nvox, TPB), (unsigned)bfm_cdiv(Cout, COT));
    hipLaunchKernelGGL(conv_direct, grid, dim3(TPB), 0, bfm_s(stream), A, CA, B, CB, D, H, W, make_upview(up), scale,
                       shift, wpacked, Cout, slope, out);
    return bfm_launch_status();
}
