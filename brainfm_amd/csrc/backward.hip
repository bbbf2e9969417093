// Backward pass of the fused SingleConv block  y = LeakyReLU(conv3x3x3(GroupNorm(cat(skip, up(x)))))  and of the
// pooling / upsampling around it (first correct version: exact-fp32 matrix cores for the weight gradient,
// the forward conv kernels reused for the data gradient).  References: the reference trains through torch autograd
// over Trainer/models/unet3d/buildingblocks.py:31-60 (SingleConv 'gcl'), :185-186 (MaxPool3d), :265-276, 361-363
// (nearest upsample + concat).
//
//   dP  = dY * (Y > 0 ? 1 : slope)                                   lrelu_bwd
//   dW[co][ci][tap] = sum_v dP[v][co] * Xn[v + tap][ci]              conv_wgrad   (Xn = GN-applied input, 0 outside)
//   dXn = conv3x3x3(dP, W^T mirrored)                                forward conv kernel on transposed weights (host)
//   GroupNorm:  dbeta_c = sum_v dXn,  dgamma_c = sum_v dXn*xhat,
//               dx = rstd_g * (gamma_c*dXn - m1_g - xhat*m2_g),   m1 = mean_g(gamma*dXn), m2 = mean_g(gamma*dXn*xhat)
//   nearest upsample: the gradient of a low-res voxel is the sum over the box of its replicas.
#include "bfm_common.h"

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)(n); i += (int64_t)gridDim.x * blockDim.x)

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ----------------------------------------------------------------------------- LeakyReLU
__global__ void lrelu_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y, int64_t n4, float slope,
                                 float* __restrict__ dP, float* __restrict__ absmax) {
    const float4* a = reinterpret_cast<const float4*>(dY);
    const float4* y = reinterpret_cast<const float4*>(Y);
    float4* o = reinterpret_cast<float4*>(dP);
    float m = 0.f;
    GRID_STRIDE(i, n4) {
        const float4 g = a[i], v = y[i];
        const float4 r = make_float4(v.x > 0.f ? g.x : g.x * slope, v.y > 0.f ? g.y : g.y * slope,
                                     v.z > 0.f ? g.z : g.z * slope, v.w > 0.f ? g.w : g.w * slope);
        o[i] = r;
        m = fmaxf(fmaxf(m, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    }
    if (absmax) {                                         // max |dP| for the kernels that rescale dP (order independent)
        __shared__ float wm[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
            // one atomic per block, and only when it can raise the value (non-negative floats order like ints)
            if (m > 0.f && __float_as_int(m) > *reinterpret_cast<volatile int*>(absmax))
                atomicMax(reinterpret_cast<int*>(absmax), __float_as_int(m));
        }
    }
}

// ----------------------------------------------------------------------------- weight gradient
struct WgParams {
    const float* dP;
    int Cout;
    const float *A, *B;
    int CA, CB;
    UpView up;
    const float *scale, *shift;
    int D, H, W;
    float* part;                      // [S][Cout][Cin][27]
    int S;
    int rows_per_split;               // (z,y) rows per split
};

// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain): A[i=co][k] = dP[voxel k][co], B[k][j=ci] = Xn[voxel k + tap][ci],
// K = two x-neighbouring voxels per step.
// LDS-tiled persistent version for Cout % 64 == 0, CA % 32 == 0, Cin % 32 == 0 (every layer of the full-width net but
// the stem).  A workgroup (8 waves) owns a 64 (co) x 32 (ci) x 27 (tap) block of dW and walks 4x4x16-voxel tiles of
// the volume: dP tile [256][64] and the GroupNorm-applied input tile with its halo [6*6*18][32] are staged once in LDS
// (145 KB), then wave (cob, tg) runs 128 K-steps (two x-neighbouring voxels each) x 7 taps of v_mfma_f32_32x32x2_f32
// with both fragments read conflict-free from LDS (32 consecutive floats per half wave).  Staging is ~2 % of a tile's
// MFMA time, so a single buffer with two barriers per tile is enough.  Accumulators stay in registers across all of a
// workgroup's tiles; one partial per workgroup column (fixed order -> bit-reproducible).
constexpr int WT_Z = 4, WT_Y = 4, WT_X = 16;
constexpr int WH_Z = WT_Z + 2, WH_Y = WT_Y + 2, WH_X = WT_X + 2;
constexpr int WT_VOX = WT_Z * WT_Y * WT_X;                 // 256
constexpr int WH_VOX = WH_Z * WH_Y * WH_X;                 // 648
constexpr int WG2_THREADS = 512;
constexpr int WG2_LDS = (WT_VOX * 64 + WH_VOX * 32) * (int)sizeof(float);

struct Wg2Params {
    WgParams b;
    int nbz, nby, nbx, ntiles;
};

__global__ void __launch_bounds__(WG2_THREADS) conv_wgrad_tiled_kernel(const Wg2Params q) {
    extern __shared__ float wg_lds[];
    const WgParams& p = q.b;
    float* dPs = wg_lds;                                   // [256][64]
    float* Xs = wg_lds + WT_VOX * 64;                      // [648][32]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, lh = lane >> 5;
    const int cob = wave & 1, tg = wave >> 1;              // taps [7 tg, 7 tg + 7) (the last group has 6)
    const int t0 = tg * 7, nt = tg == 3 ? 6 : 7;
    const int Cin = p.CA + p.CB;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 64;
    const bool fromB = ci0 >= p.CA;
    int toff[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int t = min(t0 + j, 26);
        const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
        toff[j] = ((kd * WH_Y + kh) * WH_X + kw) * 32;
    }
    floatx16 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    // staging roles (512 threads): dP quad (tid & 15), input quad (tid & 7) are fixed per thread
    const int xq = tid & 7;
    const float4 sc4 = *reinterpret_cast<const float4*>(p.scale + ci0 + xq * 4);
    const float4 sh4 = *reinterpret_cast<const float4*>(p.shift + ci0 + xq * 4);

    for (int tile = blockIdx.x; tile < q.ntiles; tile += gridDim.x) {
        const int bx = tile % q.nbx;
        const int t2 = tile / q.nbx;
        const int by = t2 % q.nby, bz = t2 / q.nby;
        const int z0 = bz * WT_Z, y0 = by * WT_Y, x0 = bx * WT_X;
        __syncthreads();                                   // the previous tile's reads are done
        for (int i = tid; i < WT_VOX * 16; i += WG2_THREADS) {
            const int v = i >> 4, c4 = i & 15;
            const int zl = v >> 6, yl = (v >> 4) & 3, xl = v & 15;
            const int z = z0 + zl, y = y0 + yl, x = x0 + xl;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (z < p.D && y < p.H && x < p.W)
                val = *reinterpret_cast<const float4*>(p.dP + ((int64_t)(z * p.H + y) * p.W + x) * p.Cout + co0 + c4 * 4);
            *reinterpret_cast<float4*>(dPs + v * 64 + c4 * 4) = val;
        }
        for (int i = tid; i < WH_VOX * 8; i += WG2_THREADS) {
            const int hv = i >> 3;
            const int hz = hv / (WH_Y * WH_X);
            const int r = hv - hz * (WH_Y * WH_X);
            const int hy = r / WH_X, hx = r - hy * WH_X;
            const int zz = z0 + hz - 1, yy = y0 + hy - 1, xx = x0 + hx - 1;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (zz >= 0 && zz < p.D && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
                const float* src = fromB
                    ? p.B + ((int64_t)(p.up.mapD[zz] * p.up.h + p.up.mapH[yy]) * p.up.w + p.up.mapW[xx]) * p.CB + (ci0 - p.CA)
                    : p.A + ((int64_t)(zz * p.H + yy) * p.W + xx) * p.CA + ci0;
                const float4 x4 = *reinterpret_cast<const float4*>(src + xq * 4);
                val = make_float4(fmaf(x4.x, sc4.x, sh4.x), fmaf(x4.y, sc4.y, sh4.y), fmaf(x4.z, sc4.z, sh4.z),
                                  fmaf(x4.w, sc4.w, sh4.w));
            }
            *reinterpret_cast<float4*>(Xs + hv * 32 + xq * 4) = val;
        }
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < WT_Z * WT_Y; ++r) {
            const int zl = r >> 2, yl = r & 3;
            const float* arow = dPs + (r * WT_X + lh) * 64 + cob * 32 + l32;
            const float* brow = Xs + ((zl * WH_Y + yl) * WH_X + lh) * 32 + l32;
#pragma unroll
            for (int xp = 0; xp < WT_X / 2; ++xp) {
                const float a = arow[xp * 2 * 64];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const float b = brow[xp * 2 * 32 + toff[j]];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                }
            }
        }
    }
    float* out = p.part + (int64_t)blockIdx.x * p.Cout * Cin * 27;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (j >= nt) continue;                             // (a break here keeps the loop rolled and acc[] in scratch)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;
            out[((int64_t)(co0 + cob * 32 + row) * Cin + ci0 + l32) * 27 + t0 + j] = acc[j][i];
        }
    }
}

// Split-fp16 version of the tiled weight gradient (three v_mfma_f32_32x32x16_f16 per product: hi*hi + hi*lo + lo*hi,
// fp32 accumulation; same scheme and accuracy class as the forward conv kernels, ~2.5x the fp32 matrix-core rate).
// The reduction index of dW is the voxel, and a 16-deep MFMA wants 8 consecutive K values per lane: K = the 16 x
// positions of one tile row, so the tiles live TRANSPOSED in LDS ([row][channel][x], fp16 hi and lo planes, x pairs
// packed per dword at staging time with v_cvt_pkrtz).  The three kw taps of an input row come from ONE 5-dword read
// per plane: kw=0 -> dwords 0..3, kw=2 -> dwords 1..4, kw=1 -> v_alignbit of neighbouring dwords (the 2-byte shift).
// Workgroup = 8 waves = 2 co-blocks x 4 tap groups (7,7,7,6) on a 64 (co) x 32 (ci) x 27 block of dW, persistent over
// 2x4x16-voxel tiles; the next tile's global loads are in flight during the MFMA phase (prefetch registers), the
// convert + LDS store sits between two barriers.  Row pitches are padded by 8 dwords so that the staging writes (lanes =
// x-pair x row) hit 32 distinct banks.
typedef _Float16 wg_half8 __attribute__((ext_vector_type(8)));
typedef __fp16 wg_fp16x2 __attribute__((ext_vector_type(2)));
constexpr int HT_Z = 2, HT_Y = 4, HT_X = 16;
constexpr int HR = HT_Z * HT_Y;                            // 8 tile rows
constexpr int HHY = HT_Y + 2;
constexpr int HHR = (HT_Z + 2) * HHY;                      // 24 halo rows
constexpr int DP_ROW = 64 * 8 + 8;                         // dwords per dP row: 64 channels x 8 x-pairs, padded
constexpr int DP_PLANE = HR * DP_ROW;
constexpr int X_CH = 12;                                   // dwords per channel of an input row: hx 0..17 (+pad), 48 B
constexpr int X_ROW = 32 * X_CH + 8;
constexpr int X_PLANE = HHR * X_ROW;
constexpr int WG3_LDS = (2 * DP_PLANE + 2 * X_PLANE) * 4;  // 108.5 KB
constexpr int WG3_THREADS = 512;

struct Wg3Params {
    WgParams b;
    int nbz, nby, nbx, ntiles;
    const float* dp_bound;            // [1]  max |dP|
    const float* x_bound;             // [G]  max |GroupNorm-applied input| per group
    int G;
    int part_cin;                     // input channels per row of `part` (Cin; CA when only the skip channels run here)
};

__device__ __forceinline__ int pow2_exp_for(float bmax) {  // e with bmax * 2^e in [2^12, 2^13)
    int e = 0;
    if (bmax > 0.f && bmax < INFINITY) {
        int ex;
        (void)frexpf(bmax, &ex);
        e = 13 - ex;
        e = e > 60 ? 60 : (e < -60 ? -60 : e);
    }
    return e;
}

__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    const wg_fp16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const wg_fp16x2 l = __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// One tile's MFMAs of wave (cob, g).  Every wave runs the SAME code (specialising per tap group with if / else around the
// accumulators made the allocator keep two copies of them: 36 spills with nothing else live): group g owns the input rows
// (kd,kh) = 2g and 2g+1 with all three kw (compile-time fragment shapes, run-time LDS row offsets) and, for g < 3, tap
// (2,2,kw=g) with a wave-uniform select of the fragment shape.
__device__ __forceinline__ void wg3_frags(const uint32_t* xb, int kw, wg_half8& b_hi, wg_half8& b_lo, const uint4& h,
                                          uint32_t h4, const uint4& l, uint32_t l4) {
    uint4 bh, bl;
    if (kw == 0) {
        bh = h; bl = l;
    } else if (kw == 2) {
        bh = make_uint4(h.y, h.z, h.w, h4);
        bl = make_uint4(l.y, l.z, l.w, l4);
    } else {
        bh = make_uint4(__builtin_amdgcn_alignbit(h.y, h.x, 16), __builtin_amdgcn_alignbit(h.z, h.y, 16),
                        __builtin_amdgcn_alignbit(h.w, h.z, 16), __builtin_amdgcn_alignbit(h4, h.w, 16));
        bl = make_uint4(__builtin_amdgcn_alignbit(l.y, l.x, 16), __builtin_amdgcn_alignbit(l.z, l.y, 16),
                        __builtin_amdgcn_alignbit(l.w, l.z, 16), __builtin_amdgcn_alignbit(l4, l.w, 16));
    }
    b_hi = __builtin_bit_cast(wg_half8, bh);
    b_lo = __builtin_bit_cast(wg_half8, bl);
    (void)xb;
}

__device__ __forceinline__ void wg3_mfma_phase(const uint32_t* __restrict__ dPs, const uint32_t* __restrict__ Xs,
                                               floatx16 (&acc)[7], int cob, int g, int l32, int lh) {
    int offR[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 2 * g + i;
        const int kd = row / 3, kh = row - kd * 3;
        offR[i] = (kd * HHY + kh) * X_ROW;
    }
    offR[2] = (2 * HHY + 2) * X_ROW;
#pragma unroll 1
    for (int r = 0; r < HR; ++r) {
        const int zl = r / HT_Y, yl = r - zl * HT_Y;
        const uint32_t* ap = dPs + r * DP_ROW + (cob * 32 + l32) * 8 + 4 * lh;
        const wg_half8 a_hi = __builtin_bit_cast(wg_half8, *reinterpret_cast<const uint4*>(ap));
        const wg_half8 a_lo = __builtin_bit_cast(wg_half8, *reinterpret_cast<const uint4*>(ap + DP_PLANE));
        const uint32_t* base = Xs + (zl * HHY + yl) * X_ROW + l32 * X_CH + 4 * lh;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t* xb = base + offR[i];
            const uint4 h = *reinterpret_cast<const uint4*>(xb);
            const uint32_t h4 = xb[4];
            const uint4 l = *reinterpret_cast<const uint4*>(xb + X_PLANE);
            const uint32_t l4 = xb[X_PLANE + 4];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                wg_half8 b_hi, b_lo;
                wg3_frags(xb, kw, b_hi, b_lo, h, h4, l, l4);
                const int j = i * 3 + kw;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[j], 0, 0, 0);
            }
        }
        if (g < 3) {                                               // wave-uniform
            const uint32_t* xb = base + offR[2];
            const uint4 h = *reinterpret_cast<const uint4*>(xb);
            const uint32_t h4 = xb[4];
            const uint4 l = *reinterpret_cast<const uint4*>(xb + X_PLANE);
            const uint32_t l4 = xb[X_PLANE + 4];
            wg_half8 b_hi, b_lo;
            if (g == 0) wg3_frags(xb, 0, b_hi, b_lo, h, h4, l, l4);
            else if (g == 1) wg3_frags(xb, 1, b_hi, b_lo, h, h4, l, l4);
            else wg3_frags(xb, 2, b_hi, b_lo, h, h4, l, l4);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[6], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[6], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[6], 0, 0, 0);
        }
    }
}

__global__ void __launch_bounds__(WG3_THREADS) conv_wgrad_f16_kernel(const Wg3Params q) {
    extern __shared__ uint32_t wg3_lds[];
    const WgParams& p = q.b;
    uint32_t* dPs = wg3_lds;
    uint32_t* Xs = wg3_lds + 2 * DP_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cob = wave & 1, tg = wave >> 1;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 64;
    const bool fromB = ci0 >= p.CA;

    float xb = 0.f;
    for (int g = 0; g < q.G; ++g) xb = fmaxf(xb, q.x_bound[g]);
    const int ea = pow2_exp_for(q.dp_bound[0]), ex = pow2_exp_for(xb);
    const float sa = ldexpf(1.0f, ea), sx = ldexpf(1.0f, ex), dq = ldexpf(1.0f, -(ea + ex));

    floatx16 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    // ---- staging roles.  A thread keeps ONE channel quad for all its items (one scale/shift quad in registers):
    //   lane = q4 + 4 * sub: four lanes read 64 contiguous bytes of a voxel; within a wave instruction the LDS dword
    //   address is row*ROW + (4q + k')*CH + xp with k' = (k + q4) & 3 (channel order rotated per lane), which spreads the
    //   four quads over distinct banks (channel pitches are multiples of 4 dwords, so un-rotated they would collide).
    //   dP:    quad = 4 (wave & 3) + q4, rows 4 (wave >> 2) + 2 s + (sub >> 3), x pair = sub & 7          (2 items)
    //   input: quad = 4 (wave & 1) + q4, position = 16 (wave >> 1) + sub + 64 s -> (halo row, hx pair)    (4 items)
    const int q4 = lane & 3, sub = lane >> 2;
    const int d_q = 4 * (wave & 3) + q4, d_xp = sub & 7, d_r0 = 4 * (wave >> 2) + (sub >> 3);
    const int x_qd = 4 * (wave & 1) + q4, x_pos0 = 16 * (wave >> 1) + sub;
    // rotate a 4-vector by q4 with two select stages (a chain of ?: on a lane-dependent index becomes branches)
    const bool rot1 = q4 & 1, rot2 = q4 & 2;
    auto rot4 = [&](const float (&v)[4], float (&u)[4]) __attribute__((always_inline)) {
        float w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = rot1 ? v[(k + 1) & 3] : v[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = rot2 ? w[(k + 2) & 3] : w[k];
    };
    float sc[4], sh[4];                                    // already rotated: sc[k] belongs to channel (k + q4) & 3
    {
        const float4 a = *reinterpret_cast<const float4*>(p.scale + ci0 + x_qd * 4);
        const float4 b = *reinterpret_cast<const float4*>(p.shift + ci0 + x_qd * 4);
        const float a4[4] = {a.x * sx, a.y * sx, a.z * sx, a.w * sx}, b4[4] = {b.x * sx, b.y * sx, b.z * sx, b.w * sx};
        rot4(a4, sc);
        rot4(b4, sh);
    }

    // All loads are unconditional (coordinates clamped into the volume, validity kept as mask bits): straight-line code,
    // twelve 16-byte loads in flight per thread.
    float4 pd[2][2], px[4][2];
    unsigned okmask = 0;                                   // bits 0..7: input item s element e (2 s + e); 8..11: dP
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        const int bx = tile % q.nbx;
        const int t2 = tile / q.nbx;
        const int by = t2 % q.nby, bz = t2 / q.nby;
        const int z0 = bz * HT_Z, y0 = by * HT_Y, x0 = bx * HT_X;
        okmask = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int r = d_r0 + 2 * s;
            const int z = z0 + r / HT_Y, y = y0 + r % HT_Y, x = x0 + 2 * d_xp;
            const bool rok = z < p.D && y < p.H;
            const int zc = min(z, p.D - 1), yc = min(y, p.H - 1);
            const float* row = p.dP + ((int64_t)(zc * p.H + yc) * p.W) * p.Cout + co0 + d_q * 4;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int xe = x + e;
                pd[s][e] = *reinterpret_cast<const float4*>(row + (int64_t)min(xe, p.W - 1) * p.Cout);
                okmask |= (rok && xe < p.W) ? (1u << (8 + 2 * s + e)) : 0u;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pos = min(x_pos0 + 64 * s, HHR * 9 - 1);         // slots past the end redo the last item (not stored)
            const int hrow = pos / 9, hxp = pos - hrow * 9;
            const int zz = z0 + hrow / HHY - 1, yy = y0 + hrow % HHY - 1, xx = x0 + 2 * hxp - 1;
            const bool rok = zz >= 0 && zz < p.D && yy >= 0 && yy < p.H;
            const int zc = min(max(zz, 0), p.D - 1), yc = min(max(yy, 0), p.H - 1);
            const float* row;
            int stride;
            if (fromB) {                                               // wave-uniform
                row = p.B + ((int64_t)(p.up.mapD[zc] * p.up.h + p.up.mapH[yc]) * p.up.w) * p.CB + (ci0 - p.CA) + x_qd * 4;
                stride = p.CB;
            } else {
                row = p.A + ((int64_t)(zc * p.H + yc) * p.W) * p.CA + ci0 + x_qd * 4;
                stride = p.CA;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int xe = xx + e;
                int xc = min(max(xe, 0), p.W - 1);
                if (fromB) xc = p.up.mapW[xc];
                px[s][e] = *reinterpret_cast<const float4*>(row + (int64_t)xc * stride);
                okmask |= (rok && xe >= 0 && xe < p.W) ? (1u << (2 * s + e)) : 0u;
            }
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float m0 = (okmask >> (8 + 2 * s)) & 1u ? sa : 0.f, m1 = (okmask >> (9 + 2 * s)) & 1u ? sa : 0.f;
            const float v0[4] = {pd[s][0].x, pd[s][0].y, pd[s][0].z, pd[s][0].w};
            const float v1[4] = {pd[s][1].x, pd[s][1].y, pd[s][1].z, pd[s][1].w};
            float u0[4], u1[4];
            rot4(v0, u0);
            rot4(v1, u1);
            uint32_t* dst = dPs + (d_r0 + 2 * s) * DP_ROW + (d_q * 4) * 8 + d_xp;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t hi, lo;
                split_pair(u0[k] * m0, u1[k] * m1, hi, lo);
                uint32_t* d = dst + ((k + q4) & 3) * 8;               // the channel this rotated slot belongs to
                d[0] = hi;
                d[DP_PLANE] = lo;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int pos = x_pos0 + 64 * s;
            const bool live = pos < HHR * 9;
            const int posc = min(pos, HHR * 9 - 1);
            const int hrow = posc / 9, hxp = posc - hrow * 9;
            const float v0[4] = {px[s][0].x, px[s][0].y, px[s][0].z, px[s][0].w};
            const float v1[4] = {px[s][1].x, px[s][1].y, px[s][1].z, px[s][1].w};
            float u0[4], u1[4];
            rot4(v0, u0);
            rot4(v1, u1);
            const bool ok0 = (okmask >> (2 * s)) & 1u, ok1 = (okmask >> (2 * s + 1)) & 1u;
            uint32_t* dst = Xs + hrow * X_ROW + (x_qd * 4) * X_CH + hxp;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = ok0 ? fmaf(u0[k], sc[k], sh[k]) : 0.f;       // zero padding comes after the affine
                const float b = ok1 ? fmaf(u1[k], sc[k], sh[k]) : 0.f;
                uint32_t hi, lo;
                split_pair(a, b, hi, lo);
                uint32_t* d = dst + ((k + q4) & 3) * X_CH;
                if (live) {
                    d[0] = hi;
                    d[X_PLANE] = lo;
                }
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < q.ntiles) load_tile(tile);
    for (; tile < q.ntiles; tile += gridDim.x) {
        __syncthreads();                                   // the previous tile's fragment reads are done
        store_tile();
        __syncthreads();
        const int nxt = tile + gridDim.x;
        if (nxt < q.ntiles) load_tile(nxt);                // in flight while the matrix core works
        __builtin_amdgcn_sched_barrier(0);
        wg3_mfma_phase(dPs, Xs, acc, cob, tg, l32, lh);
    }
    float* out = p.part + (int64_t)blockIdx.x * p.Cout * q.part_cin * 27;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (j == 6 && tg == 3) continue;                              // group 3 has no seventh tap
        const int tap = j < 6 ? (2 * tg + j / 3) * 3 + j % 3 : 24 + tg;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;
            out[((int64_t)(co0 + cob * 32 + row) * q.part_cin + ci0 + l32) * 27 + tap] = acc[j][i] * dq;
        }
    }
}

// ----------------------------------------------------------------------------- wave-specialised weight gradient (round 4)
// conv_wgrad_f16_kernel above stages a tile (loads -> convert -> LDS) and multiplies it in turn: measured on 64 -> 64 at
// 128^3 the matrix core works 1.2 of the kernel's 1.9 ms (profiles/r04_wgrad_phases.txt), and the same eight waves cannot
// overlap the two (254 registers, a 108 KB single buffer).  Here the roles are split: waves 0-3 (one per SIMD) only read
// LDS and issue MFMAs -- a 32 (co) x 32 (ci) x 27 block, seven taps per wave -- and waves 4-11 only stage: global loads
// run TWO tiles ahead (two register sets), convert + LDS stores one tile ahead into the other half of a double-buffered
// tile (1 x 4 x 16 voxels: 65 KB per buffer).  One s_barrier per tile.  Skip / plain inputs only (A); nearest-upsampled
// channels keep the kernels above.  Measured (profiles/r04_wgrad_phases.txt): the two roles do NOT simply overlap -- on
// 64 -> 64 at 128^3 the staging waves alone take 1.23 ms, the MFMA waves alone 1.28, together 2.03 (the 8-wave kernel:
// 2.12): under its power limit the chip pays for the staging instructions whether or not they run beside the MFMAs.  The
// gain is on the narrower levels (twice the workgroups per layer, no idle matrix core behind a barrier): the plain layers
// of the full-width net 7.87 -> 6.90 ms, the training iteration -2.6 ms.
constexpr int ST_Y = 4, ST_X = 16;
constexpr int SHY = ST_Y + 2, SHR = 3 * SHY;               // 18 halo rows
constexpr int SDP_ROW = 32 * 8 + 8;
constexpr int SDP_PLANE = ST_Y * SDP_ROW;
constexpr int SX_PLANE = SHR * X_ROW;
constexpr int WS_BUF = 2 * SDP_PLANE + 2 * SX_PLANE;       // dwords per buffer (hi + lo planes of both operands)
constexpr int WGS_LDS = 2 * WS_BUF * 4;                    // 129.8 KB
constexpr int WGS_THREADS = 768;                        // 4 MFMA waves + 8 staging waves

struct WgWsParams {
    const float* dP;                  // [D][H][W][Cout]
    const float* A;                   // [D][H][W][CA]
    const float *scale, *shift;
    int Cout, CA, D, H, W;
    float* part;                      // [S][Cout][part_cin][27]
    int part_cin;
    int nby, nbx, ntiles;
    const float* dp_bound;
    const float* x_bound;
    int G;
};

__global__ void __launch_bounds__(WGS_THREADS) conv_wgrad_ws_kernel(const WgWsParams p) {
    extern __shared__ uint32_t wgs_lds[];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    float xb = 0.f;
    for (int g = 0; g < p.G; ++g) xb = fmaxf(xb, p.x_bound[g]);
    const int ea = pow2_exp_for(p.dp_bound[0]), ex = pow2_exp_for(xb);
    const int n_my = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // >= 1
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    if (wave >= 4) {
        // ------------------------------------------------------------------ loaders
        // 16 bytes per lane and load (the texture path takes a wave's load in ~16 cycles whatever its width: one 4-byte
        // load per element, the row-per-wave form, made the four loaders the bottleneck -- 2.56 instead of 2.22 ms on
        // 64 -> 64 at 128^3); four lanes read 64 contiguous bytes of a voxel and the channel order is rotated per lane so
        // that the four quads land in distinct LDS banks (conv_wgrad_f16_kernel's staging; its instruction count does not
        // matter here, the matrix core runs beside it).
        //   dP:    quad = 4 (lw & 1) + q4, row = 2 (lw >> 1) + (sub >> 3), x pair = sub & 7        (1 item, waves lw < 4)
        //   input: quad = 4 (lw & 1) + q4, position = 16 (lw >> 1) + sub + 64 s -> (halo row, hx pair)        (3 items)
        // Eight staging waves: a single wave per SIMD runs this dependent convert chain at one instruction per ~14 cycles
        // (1.5 ms on its own for 64 -> 64 at 128^3, longer than the matrix core's 1.3).
        const int lw = wave - 4;
        const float sa = ldexpf(1.0f, ea), sx = ldexpf(1.0f, ex);
        const int q4 = lane & 3, sub = lane >> 2;
        const int quad = 4 * (lw & 1) + q4, d_xp = sub & 7, d_r = 2 * ((lw >> 1) & 1) + (sub >> 3);
        const bool has_dp = lw < 4;                            // wave-uniform: the dP tile is one item for four of the eight waves
        const int x_pos0 = 16 * (lw >> 1) + sub;
        const bool rot1 = q4 & 1, rot2 = q4 & 2;
        auto rot4 = [&](const float (&v)[4], float (&u)[4]) __attribute__((always_inline)) {
            float w_[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w_[k] = rot1 ? v[(k + 1) & 3] : v[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = rot2 ? w_[(k + 2) & 3] : w_[k];
        };
        float sc[4], sh[4];                                    // already rotated: sc[k] belongs to channel (k + q4) & 3
        {
            const float4 a = *reinterpret_cast<const float4*>(p.scale + ci0 + quad * 4);
            const float4 b = *reinterpret_cast<const float4*>(p.shift + ci0 + quad * 4);
            const float a4[4] = {a.x * sx, a.y * sx, a.z * sx, a.w * sx}, b4[4] = {b.x * sx, b.y * sx, b.z * sx, b.w * sx};
            rot4(a4, sc);
            rot4(b4, sh);
        }
        constexpr int NXI = 3, XSTEP = 64;
        auto coords = [&](int k, int& z0, int& y0, int& x0) __attribute__((always_inline)) {
            const int tile = blockIdx.x + k * gridDim.x;
            const int bx = tile % p.nbx;
            const int t2 = tile / p.nbx;
            z0 = t2 / p.nby; y0 = (t2 % p.nby) * ST_Y; x0 = bx * ST_X;
        };
        // register set of one tile; okm: bits 0..11 input item s element e (2 s + e), 12..13 dP
        auto load = [&](int k, float4 (&pd)[2], float4 (&px)[NXI][2], unsigned& okm) __attribute__((always_inline)) {
            int z0, y0, x0;
            coords(k, z0, y0, x0);
            okm = 0;
            if (has_dp) {
                const int y = y0 + d_r, x = x0 + 2 * d_xp;
                const bool rok = y < p.H;
                const float* row = p.dP + ((int64_t)(z0 * p.H + min(y, p.H - 1)) * p.W) * p.Cout + co0 + quad * 4;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    pd[e] = *reinterpret_cast<const float4*>(row + (int64_t)min(x + e, p.W - 1) * p.Cout);
                    okm |= (rok && x + e < p.W) ? (1u << (12 + e)) : 0u;
                }
            }
#pragma unroll
            for (int s_ = 0; s_ < NXI; ++s_) {
                const int pos = min(x_pos0 + XSTEP * s_, SHR * 9 - 1);        // slots past the end redo the last item (not stored)
                const int hrow = pos / 9, hxp = pos - hrow * 9;
                const int zz = z0 + hrow / SHY - 1, yy = y0 + hrow % SHY - 1, xx = x0 + 2 * hxp - 1;
                const bool rok = zz >= 0 && zz < p.D && yy >= 0 && yy < p.H;
                const int zc = min(max(zz, 0), p.D - 1), yc = min(max(yy, 0), p.H - 1);
                const float* row = p.A + ((int64_t)(zc * p.H + yc) * p.W) * p.CA + ci0 + quad * 4;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int xe = xx + e;
                    px[s_][e] = *reinterpret_cast<const float4*>(row + (int64_t)min(max(xe, 0), p.W - 1) * p.CA);
                    okm |= (rok && xe >= 0 && xe < p.W) ? (1u << (2 * s_ + e)) : 0u;
                }
            }
        };
        auto convert = [&](const float4 (&pd)[2], const float4 (&px)[NXI][2], unsigned okm, uint32_t* buf)
                           __attribute__((always_inline)) {
            uint32_t* dPs = buf;
            uint32_t* Xs = buf + 2 * SDP_PLANE;
            if (has_dp) {
                const float m0 = (okm >> 12) & 1u ? sa : 0.f, m1 = (okm >> 13) & 1u ? sa : 0.f;
                const float v0[4] = {pd[0].x, pd[0].y, pd[0].z, pd[0].w};
                const float v1[4] = {pd[1].x, pd[1].y, pd[1].z, pd[1].w};
                float u0[4], u1[4];
                rot4(v0, u0);
                rot4(v1, u1);
                uint32_t* dst = dPs + d_r * SDP_ROW + (quad * 4) * 8 + d_xp;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t hi, lo;
                    split_pair(u0[k] * m0, u1[k] * m1, hi, lo);
                    uint32_t* d_ = dst + ((k + q4) & 3) * 8;               // the channel this rotated slot belongs to
                    d_[0] = hi;
                    d_[SDP_PLANE] = lo;
                }
            }
#pragma unroll
            for (int s_ = 0; s_ < NXI; ++s_) {
                const int pos = x_pos0 + XSTEP * s_;
                const bool live = pos < SHR * 9;
                const int posc = min(pos, SHR * 9 - 1);
                const int hrow = posc / 9, hxp = posc - hrow * 9;
                const float v0[4] = {px[s_][0].x, px[s_][0].y, px[s_][0].z, px[s_][0].w};
                const float v1[4] = {px[s_][1].x, px[s_][1].y, px[s_][1].z, px[s_][1].w};
                float u0[4], u1[4];
                rot4(v0, u0);
                rot4(v1, u1);
                const bool ok0 = (okm >> (2 * s_)) & 1u, ok1 = (okm >> (2 * s_ + 1)) & 1u;
                uint32_t* dst = Xs + hrow * X_ROW + (quad * 4) * X_CH + hxp;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = ok0 ? fmaf(u0[k], sc[k], sh[k]) : 0.f;       // zero padding comes after the affine
                    const float b = ok1 ? fmaf(u1[k], sc[k], sh[k]) : 0.f;
                    uint32_t hi, lo;
                    split_pair(a, b, hi, lo);
                    uint32_t* d_ = dst + ((k + q4) & 3) * X_CH;
                    if (live) {
                        d_[0] = hi;
                        d_[SX_PLANE] = lo;
                    }
                }
            }
        };
        float4 pd0[2], px0[NXI][2], pd1[2], px1[NXI][2];
        unsigned ok0_ = 0, ok1_ = 0;
        uint32_t* buf0 = wgs_lds;
        uint32_t* buf1 = wgs_lds + WS_BUF;
        load(0, pd0, px0, ok0_);
        if (n_my > 1) load(1, pd1, px1, ok1_);
        convert(pd0, px0, ok0_, buf0);
        if (n_my > 2) load(2, pd0, px0, ok0_);
        barrier();                                                     // buffer 0 holds tile 0
        // even k: set 0 holds the loads of tile k + 2, set 1 those of tile k + 1
        for (int k = 0; k < n_my; k += 2) {
            if (k + 1 < n_my) convert(pd1, px1, ok1_, buf1);           // tile k + 1, beside the MFMAs of tile k (buffer 0)
            if (k + 3 < n_my) load(k + 3, pd1, px1, ok1_);
            barrier();
            if (k + 1 >= n_my) break;
            if (k + 2 < n_my) convert(pd0, px0, ok0_, buf0);           // tile k + 2, beside the MFMAs of tile k + 1 (buffer 1)
            if (k + 4 < n_my) load(k + 4, pd0, px0, ok0_);
            barrier();
        }
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves: tap group tg = wave
    const int tg = wave;
    const float dq = ldexpf(1.0f, -(ea + ex));
    floatx16 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    int offR[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 2 * tg + i;
        const int kd = row / 3, kh = row - kd * 3;
        offR[i] = (kd * SHY + kh) * X_ROW;
    }
    offR[2] = (2 * SHY + 2) * X_ROW;
    auto mfma_tile = [&](const uint32_t* buf) __attribute__((always_inline)) {
        const uint32_t* dPs = buf;
        const uint32_t* Xs = buf + 2 * SDP_PLANE;
#pragma unroll 1
        for (int yl = 0; yl < ST_Y; ++yl) {
            const uint32_t* ap = dPs + yl * SDP_ROW + l32 * 8 + 4 * lh;
            const wg_half8 a_hi = __builtin_bit_cast(wg_half8, *reinterpret_cast<const uint4*>(ap));
            const wg_half8 a_lo = __builtin_bit_cast(wg_half8, *reinterpret_cast<const uint4*>(ap + SDP_PLANE));
            const uint32_t* base = Xs + yl * X_ROW + l32 * X_CH + 4 * lh;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t* xb_ = base + offR[i];
                const uint4 h = *reinterpret_cast<const uint4*>(xb_);
                const uint32_t h4 = xb_[4];
                const uint4 l = *reinterpret_cast<const uint4*>(xb_ + SX_PLANE);
                const uint32_t l4 = xb_[SX_PLANE + 4];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    wg_half8 b_hi, b_lo;
                    wg3_frags(xb_, kw, b_hi, b_lo, h, h4, l, l4);
                    const int j = i * 3 + kw;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[j], 0, 0, 0);
                }
            }
            if (tg < 3) {                                              // wave-uniform
                const uint32_t* xb_ = base + offR[2];
                const uint4 h = *reinterpret_cast<const uint4*>(xb_);
                const uint32_t h4 = xb_[4];
                const uint4 l = *reinterpret_cast<const uint4*>(xb_ + SX_PLANE);
                const uint32_t l4 = xb_[SX_PLANE + 4];
                wg_half8 b_hi, b_lo;
                if (tg == 0) wg3_frags(xb_, 0, b_hi, b_lo, h, h4, l, l4);
                else if (tg == 1) wg3_frags(xb_, 1, b_hi, b_lo, h, h4, l, l4);
                else wg3_frags(xb_, 2, b_hi, b_lo, h, h4, l, l4);
                acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc[6], 0, 0, 0);
                acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc[6], 0, 0, 0);
                acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc[6], 0, 0, 0);
            }
        }
    };
    barrier();                                                         // buffer 0 holds tile 0
#pragma unroll 1
    for (int k = 0; k < n_my; ++k) {                                   // (one copy of the MFMA block: see wg3_mfma_phase)
        mfma_tile(wgs_lds + (k & 1) * WS_BUF);
        barrier();
    }
    float* out = p.part + (int64_t)blockIdx.x * p.Cout * p.part_cin * 27;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if (j == 6 && tg == 3) continue;                              // group 3 has no seventh tap
        const int tap = j < 6 ? (2 * tg + j / 3) * 3 + j % 3 : 24 + tg;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;
            out[((int64_t)(co0 + row) * p.part_cin + ci0 + l32) * 27 + tap] = acc[j][i] * dq;
        }
    }
}

// ----------------------------------------------------------------------------- weight gradient of the upsampled channels
// Decoder first convs read cat(skip, nearest_up2(L)).  For the L channels the 27-tap correlation over the high-res grid
//   dW[co][ci][t] = sum_v dP[v][co] * U[v + t - 1][ci],   U[v] = Ln[v >> 1]   (Ln = GroupNorm-applied L, zero outside)
// folds, exactly as conv3d_upfold.hip folds the forward pass: an output voxel v = 2q + r (parity r in {0,1}^3) sees only
// TWO distinct low-res positions per axis -- r = 0: q-1 (t = 0), q (t = 1, 2);  r = 1: q (t = 0, 1), q+1 (t = 2) -- so
//   F[r][f][co][ci] = sum_q dP[2q + r][co] * Ln[q + r + f - 1][ci],      f in {0,1}^3,
//   dW[..][t]       = sum over the 2^3 (r, f) with f = f(r, t) per axis: t=0: (0,0),(1,0); t=1: (0,1),(1,0); t=2: (0,1),(1,1)
// 64 products per low-res voxel instead of 216 (3.4x fewer), the input tile is the low-res tensor (8x fewer voxels to
// stage), and the padding is the same zero ring in low-res coordinates (2q+r+t-1 leaves the volume exactly when q+r+f-1
// does, for dims = 2 x low-res dims).  Same split-fp16 scheme, LDS layouts and fragment tricks as conv_wgrad_f16_kernel:
// a workgroup owns 32 (co) x 32 (ci) x 64 (r, f), WAVE = PARITY r (its A fragments are the dP rows of its own parity,
// all eight waves share the low-res halo tile), eight accumulators (f) per wave, persistent over 1x4x16 low-res tiles.
// The eighth of dP a parity owns is gathered with stride 2 along x (two 128-byte runs per voxel pair).
constexpr int UT_Y = 4, UT_X = 16;
constexpr int UHY = UT_Y + 2, UHR = 3 * UHY;               // 18 halo rows (z-1, z, z+1)
constexpr int UDP_ROW = 32 * 8 + 8;                        // dwords per dP row: 32 channels x 8 x-pairs, padded
constexpr int UDP_ROWS = 8 * UT_Y;                         // (parity, y) rows
constexpr int UDP_PLANE = UDP_ROWS * UDP_ROW;
constexpr int UX_PLANE = UHR * X_ROW;
constexpr int WGU_LDS = (2 * UDP_PLANE + 2 * UX_PLANE) * 4;  // 121 KB
constexpr int WGU_THREADS = 256;

struct WgUpParams {
    const float* dP;                  // [D][H][W][Cout], D = 2 d ...
    const float* L;                   // [d][h][w][CB]
    const float *scale, *shift;       // of the CB channels
    int Cout, CB, d, h, w;
    float* part;                      // [S][Cout][CB][64]
    int nby, nbx, ntiles;
    const float* dp_bound;
    const float* x_bound;
    int G;
};

// FOUR waves, one per SIMD with the whole register file (256 accumulator + 256 vector registers): wave w owns the two
// parities (rz, ry) = (w >> 1, w & 1), rx = 0 and 1 -- sixteen 32x32 accumulators.  Both x parities read the same halo
// row: x shifts 0, 1 (rx = 0) and 1, 2 (rx = 1) are three views of ONE 5-dword LDS read, so a row costs two A reads and
// four halo reads for 48 MFMAs.  (The first version ran eight waves of 256 registers, one parity each: 19 spills and a
// strictly serial read -> wait -> 3 MFMA schedule, 1.2x over the 27-tap kernel instead of the 2.4x the product count allows.)
__global__ void __launch_bounds__(WGU_THREADS) conv_wgrad_up_f16_kernel(const WgUpParams p) {
    extern __shared__ uint32_t wgu_lds[];
    uint32_t* dPs = wgu_lds;
    uint32_t* Xs = wgu_lds + 2 * UDP_PLANE;
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.z * 32;
    const int H = 2 * p.h, W = 2 * p.w;

    float xb = 0.f;
    for (int g = 0; g < p.G; ++g) xb = fmaxf(xb, p.x_bound[g]);
    const int ea = pow2_exp_for(p.dp_bound[0]), ex = pow2_exp_for(xb);
    const float sa = ldexpf(1.0f, ea), sx = ldexpf(1.0f, ex), dq = ldexpf(1.0f, -(ea + ex));

    floatx16 acc[16];                                      // [rx][fz][fy][fx]
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    // Staging, one LDS row per wave instruction: lane = (channel c = lane >> 1, x half xh = lane & 1) owns the eight x of its
    // half of a row for ONE channel -- eight (ten) 4-byte loads whose addresses differ by a wave-uniform stride (32 lanes x 4
    // bytes contiguous per load), four x pairs converted to one 16-byte LDS store per plane.  No per-lane channel rotation
    // (the float4-per-voxel form of conv_wgrad_f16_kernel spends half of its staging instructions on v_cndmask), no
    // per-element masks: every tile is a full 4 x 16 one (h >= 4, w >= 16 host-checked; the last tile of a ragged row / column is
    // shifted back inside and the voxels it shares with its neighbour count as zero in dP), the x ends of the halo are the
    // only padding.
    // Measured on the 64 + 128 -> 64 join at 128^3 (kernel 1.17 ms: MFMA phase 0.49, LDS stores 0.23, loads 0.43): the
    // float4 form cost 0.55 (stores) + 0.32 (loads); 8-byte loads of channel pairs (62 per thread instead of 114, under
    // the 64 a wave keeps in flight) 0.17 + 0.56 -- four 128-byte segments per load are slower than two; issuing the
    // halo rows half way through the MFMA phase brought spills and no gain.
    //   dP rows 8 wave + s (s = 0..7): parity (rz, ry, rx = s >> 2), y = s & 3 -- each wave stages the rows it multiplies
    //   halo rows wave + 4 s (s = 0..4, < 18)
    const int rz = wave >> 1, ry = wave & 1;               // this wave's parities: (rz, ry, 0) and (rz, ry, 1)
    const int c = lane >> 1, xh = lane & 1;
    const float scx = p.scale[ci0 + c] * sx, shx = p.shift[ci0 + c] * sx;
    constexpr int ND = 8, NX = 5;
    float pd[ND][8], px[NX][10];
    // wave-uniform: bits 0..4 halo row s inside, 5: x0 > 0, 6: x0 + 16 < w, 8..11 / 12..13: leading columns / rows of dP
    // that an earlier tile already covered -- the last tile of a row (column) that does not divide by 16 (4) is shifted
    // back inside the volume and those voxels count as zero
    unsigned xmask = 0;
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        const int bx = tile % p.nbx;
        const int t2 = tile / p.nbx;
        const int by = t2 % p.nby, z0 = t2 / p.nby;
        const int y0 = min(by * UT_Y, p.h - UT_Y), x0 = min(bx * UT_X, p.w - UT_X);
#pragma unroll
        for (int s = 0; s < ND; ++s) {
            const int rx = s >> 2, yl = s & 3;
            const float* rp = p.dP + ((int64_t)((2 * z0 + rz) * H + 2 * (y0 + yl) + ry) * W + 2 * x0 + rx) * p.Cout + co0 +
                              (16 * xh) * p.Cout + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) pd[s][j] = rp[(int64_t)(2 * j) * p.Cout];
        }
        xmask = (x0 > 0 ? 32u : 0u) | (x0 + UT_X < p.w ? 64u : 0u) | ((unsigned)(bx * UT_X - x0) << 8) |
                ((unsigned)(by * UT_Y - y0) << 12);
#pragma unroll
        for (int s = 0; s < NX; ++s) {
            const int hrow = wave + 4 * s;
            if (hrow < UHR) {                                          // wave-uniform
                const int zz = z0 + hrow / UHY - 1, yy = y0 + hrow % UHY - 1;
                if (zz >= 0 && zz < p.d && yy >= 0 && yy < p.h) xmask |= 1u << s;
                const int zc = min(max(zz, 0), p.d - 1), yc = min(max(yy, 0), p.h - 1);
                const float* rp = p.L + ((int64_t)(zc * p.h + yc) * p.w) * p.CB + ci0 + c;
                const int xb_ = x0 - 1 + 8 * xh;
#pragma unroll
                for (int j = 0; j < 8; ++j) px[s][j] = rp[(int64_t)(j == 0 ? max(xb_, 0) : xb_ + j) * p.CB];
                px[s][8] = rp[(int64_t)(x0 + UT_X - 1) * p.CB];                      // hx 16, 17: kept by the xh = 0 lanes
                px[s][9] = rp[(int64_t)min(x0 + UT_X, p.w - 1) * p.CB];
            }
        }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
        if (xmask >> 8) {                                      // wave-uniform: a shifted tile
            const int xskip = (int)((xmask >> 8) & 15u) - 8 * xh, yskip = (int)(xmask >> 12);
#pragma unroll
            for (int s = 0; s < ND; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < xskip || (s & 3) < yskip) pd[s][j] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < ND; ++s) {
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) split_pair(pd[s][2 * k] * sa, pd[s][2 * k + 1] * sa, hi[k], lo[k]);
            uint32_t* dst = dPs + (8 * wave + s) * UDP_ROW + c * 8 + 4 * xh;
            *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4*>(dst + UDP_PLANE) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
        const bool ok_first = xh == 1 || (xmask & 32u), ok_last = (xmask & 64u) != 0;
#pragma unroll
        for (int s = 0; s < NX; ++s) {
            const int hrow = wave + 4 * s;
            if (hrow < UHR) {
                const bool rok = (xmask >> s) & 1u;
                float v[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const bool ok = rok && (j == 0 ? ok_first : (j == 9 ? ok_last : true));
                    v[j] = ok ? fmaf(px[s][j], scx, shx) : 0.f;              // zero padding comes after the affine
                }
                uint32_t hi[5], lo[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) split_pair(v[2 * k], v[2 * k + 1], hi[k], lo[k]);
                uint32_t* dst = Xs + hrow * X_ROW + c * X_CH + 4 * xh;
                *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                *reinterpret_cast<uint4*>(dst + UX_PLANE) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                if (xh == 0) {
                    dst[8] = hi[4];
                    dst[UX_PLANE + 8] = lo[4];
                }
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < p.ntiles) load_tile(tile);
    for (; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();
        store_tile();
        __syncthreads();
        const int nxt = tile + gridDim.x;
        if (nxt < p.ntiles) load_tile(nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 2
        for (int yl = 0; yl < UT_Y; ++yl) {
            const uint32_t* ap = dPs + (wave * 2 * UT_Y + yl) * UDP_ROW + l32 * 8 + 4 * lh;
            const wg_half8 a0_hi = __builtin_bit_cast(wg_half8, *reinterpret_cast<const uint4*>(ap));
            const wg_half8 a0_lo = __builtin_bit_cast(wg_half8, *reinterpret_cast<const uint4*>(ap + UDP_PLANE));
            const wg_half8 a1_hi = __builtin_bit_cast(wg_half8, *reinterpret_cast<const uint4*>(ap + UT_Y * UDP_ROW));
            const wg_half8 a1_lo = __builtin_bit_cast(wg_half8, *reinterpret_cast<const uint4*>(ap + UT_Y * UDP_ROW + UDP_PLANE));
#pragma unroll
            for (int fzy = 0; fzy < 4; ++fzy) {
                const int fz = fzy >> 1, fy = fzy & 1;
                const uint32_t* xb_ = Xs + ((rz + fz) * UHY + yl + ry + fy) * X_ROW + l32 * X_CH + 4 * lh;
                const uint4 h = *reinterpret_cast<const uint4*>(xb_);
                const uint32_t h4 = xb_[4];
                const uint4 l = *reinterpret_cast<const uint4*>(xb_ + UX_PLANE);
                const uint32_t l4 = xb_[UX_PLANE + 4];
                wg_half8 s_hi[3], s_lo[3];                             // x shifts 0, 1, 2 of this halo row
#pragma unroll
                for (int k = 0; k < 3; ++k) wg3_frags(xb_, k, s_hi[k], s_lo[k], h, h4, l, l4);
                // rx = 0: fx = 0, 1 <-> shifts 0, 1;   rx = 1: fx = 0, 1 <-> shifts 1, 2
#pragma unroll
                for (int fx = 0; fx < 2; ++fx) {
                    const int j0 = fzy * 2 + fx, j1 = 8 + fzy * 2 + fx;
                    acc[j0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0_lo, s_hi[fx], acc[j0], 0, 0, 0);
                    acc[j1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1_lo, s_hi[fx + 1], acc[j1], 0, 0, 0);
                    acc[j0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0_hi, s_lo[fx], acc[j0], 0, 0, 0);
                    acc[j1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1_hi, s_lo[fx + 1], acc[j1], 0, 0, 0);
                    acc[j0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0_hi, s_hi[fx], acc[j0], 0, 0, 0);
                    acc[j1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1_hi, s_hi[fx + 1], acc[j1], 0, 0, 0);
                }
            }
        }
    }
    float* out = p.part + (int64_t)blockIdx.x * p.Cout * p.CB * 64;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int par = wave * 2 + (j >> 3);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;
            out[((int64_t)(co0 + row) * p.CB + ci0 + l32) * 64 + par * 8 + (j & 7)] = acc[j][i] * dq;
        }
    }
}

// dW[co][CA + ci][t] = sum over splits and over the eight (parity, folded tap) pairs of tap t (fixed order).
// A block folds 16 consecutive (co, ci) pairs: 16 lanes per pair sum their float4 of the 64 folded entries over the splits
// (4 KB contiguous per split and block), the 16 x 27 taps are then put together from LDS and leave as one contiguous run.
__global__ void __launch_bounds__(256) wgrad_up_fold_kernel(const float* __restrict__ part, int S, int Cout, int CB, int Cin,
                                                            int CA, float* __restrict__ dW) {
    __shared__ float f[16][65];
    const int64_t n = (int64_t)Cout * CB;                            // pairs; CB % 32 == 0: a block stays inside one co
    const int t = threadIdx.x, pl = t >> 4, q = t & 15;
    for (int64_t p0 = (int64_t)blockIdx.x * 16; p0 < n; p0 += (int64_t)gridDim.x * 16) {
        const float4* src = reinterpret_cast<const float4*>(part + (p0 + pl) * 64) + q;
        float4 a = src[0];
        for (int s_ = 1; s_ < S; ++s_) {
            const float4 v = src[(int64_t)s_ * n * 16];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        __syncthreads();
        f[pl][4 * q] = a.x; f[pl][4 * q + 1] = a.y; f[pl][4 * q + 2] = a.z; f[pl][4 * q + 3] = a.w;
        __syncthreads();
        const int co = (int)(p0 / CB), ci = (int)(p0 - (int64_t)co * CB);
        float* dst = dW + ((int64_t)co * Cin + CA + ci) * 27;
        for (int o = t; o < 16 * 27; o += 256) {
            const int pr = o / 27, tp = o - pr * 27;
            const int kd = tp / 9, kh = (tp / 3) % 3, kw = tp % 3;
            float sum = 0.f;
#pragma unroll
            for (int az = 0; az < 2; ++az)                            // parity along z, its folded tap for kd
#pragma unroll
                for (int ay = 0; ay < 2; ++ay)
#pragma unroll
                    for (int ax = 0; ax < 2; ++ax) {
                        const int fz = az == 0 ? (kd >= 1) : (kd == 2), fy = ay == 0 ? (kh >= 1) : (kh == 2),
                                  fx = ax == 0 ? (kw >= 1) : (kw == 2);
                        sum += f[pr][(az * 4 + ay * 2 + ax) * 8 + fz * 4 + fy * 2 + fx];
                    }
            dst[o] = sum;
        }
    }
}

// Narrow layers (Cin or Cout not a multiple of 32: the stem of the full net, every layer of the 8/16-wide test nets) on
// the same fp32 matrix core: dW as a [Cout] x [Cin*27] matrix, one wave per (voxel-row split, 32 rows, 32 columns);
// the B fragment column of a lane is its own (ci, tap) pair, gathered straight from the (cached) input.
__global__ void __launch_bounds__(64) conv_wgrad_cols_kernel(const WgParams p) {
    const int lane = threadIdx.x, l32 = lane & 31, lh = lane >> 5;
    const int Cin = p.CA + p.CB;
    const int ncol = Cin * 27;
    const int co = blockIdx.y * 32 + l32;
    const int col = blockIdx.z * 32 + l32;
    const bool col_ok = col < ncol;
    const int ci = col_ok ? col / 27 : 0;
    const int tap = col_ok ? col - ci * 27 : 0;
    const int kd = tap / 9 - 1, kh = (tap / 3) % 3 - 1, kw = tap % 3 - 1;
    const bool fromB = ci >= p.CA;
    const float sc = p.scale[ci], sh = p.shift[ci];
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nrows = p.D * p.H;
    const int r0 = blockIdx.x * p.rows_per_split, r1 = min(nrows, r0 + p.rows_per_split);
    for (int r = r0; r < r1; ++r) {
        const int z = r / p.H, y = r - z * p.H;
        const int zz = z + kd, yy = y + kh;
        const bool row_ok = col_ok && zz >= 0 && zz < p.D && yy >= 0 && yy < p.H;
        const float* dprow = p.dP + ((int64_t)(z * p.H + y) * p.W) * p.Cout + co;
        const float* xrow = nullptr;
        int xstride = 0;
        if (row_ok) {
            if (!fromB) {
                xrow = p.A + ((int64_t)(zz * p.H + yy) * p.W) * p.CA + ci;
                xstride = p.CA;
            } else {
                xrow = p.B + ((int64_t)(p.up.mapD[zz] * p.up.h + p.up.mapH[yy]) * p.up.w) * p.CB + (ci - p.CA);
                xstride = p.CB;
            }
        }
        for (int x = 0; x < p.W; x += 2) {
            const int xv = x + lh;
            const float a = (xv < p.W && co < p.Cout) ? dprow[(int64_t)xv * p.Cout] : 0.f;
            const int xx = xv + kw;
            float b = 0.f;
            if (row_ok && xv < p.W && xx >= 0 && xx < p.W) {
                const int xs = fromB ? p.up.mapW[xx] : xx;
                b = fmaf(xrow[(int64_t)xs * xstride], sc, sh);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    float* out = p.part + (int64_t)blockIdx.x * p.Cout * ncol;
    if (col_ok)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = blockIdx.y * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (row < p.Cout) out[(int64_t)row * ncol + col] = acc[i];
        }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int S, int64_t n, float* __restrict__ dW) {
    GRID_STRIDE(i, n) {
        float s = part[i];
        for (int k = 1; k < S; ++k) s += part[i + (int64_t)k * n];          // split order: deterministic
        dW[i] = s;
    }
}

// part [S][Cout][Cp][27] (the first Cp input channels only) -> dW [Cout][Cin][27]
__global__ void wgrad_reduce_cols_kernel(const float* __restrict__ part, int S, int Cout, int Cp, int Cin,
                                         float* __restrict__ dW) {
    const int64_t n = (int64_t)Cout * Cp * 27;
    GRID_STRIDE(i, n) {
        float s = part[i];
        for (int k = 1; k < S; ++k) s += part[i + (int64_t)k * n];          // split order: deterministic
        const int64_t co = i / ((int64_t)Cp * 27), r = i - co * ((int64_t)Cp * 27);
        dW[co * (int64_t)Cin * 27 + r] = s;
    }
}

// ----------------------------------------------------------------------------- GroupNorm backward
struct GnbParams {
    const float* dXn;                 // [D][H][W][Cin]
    const float *A, *B;
    int CA, CB;
    UpView up;
    int D, H, W, G;
    const float *mean, *rstd, *gamma;
};

// per-block partial sums over a run of voxels: part[blk][c][0] = sum dXn, [1] = sum dXn * xhat   (fp64)
__global__ void __launch_bounds__(256) gn_bwd_partial_kernel(const GnbParams p, int64_t vox_per_block,
                                                             double* __restrict__ part) {
    extern __shared__ double sm[];                        // [RP][C][2]
    const int C = p.CA + p.CB;
    const int t = threadIdx.x;
    const int CP = C < 256 ? C : 256;
    const int RP = 256 / CP;
    const int64_t nvox = (int64_t)p.D * p.H * p.W;
    const int64_t v0 = (int64_t)blockIdx.x * vox_per_block, v1 = min(nvox, v0 + vox_per_block);
    const int cpg = C / p.G;
    for (int cb = 0; cb < C; cb += CP) {
        const int c = cb + t % CP, part_i = t / CP;
        double s1 = 0.0, s2 = 0.0;
        if (part_i < RP && c < C) {
            const float mu = p.mean[c / cpg], rs = p.rstd[c / cpg];
            for (int64_t v = v0 + part_i; v < v1; v += RP) {
                float xv;
                if (c < p.CA) {
                    xv = p.A[v * p.CA + c];
                } else {
                    const int x = (int)(v % p.W);
                    const int64_t t2 = v / p.W;
                    const int y = (int)(t2 % p.H), z = (int)(t2 / p.H);
                    xv = p.B[((int64_t)(p.up.mapD[z] * p.up.h + p.up.mapH[y]) * p.up.w + p.up.mapW[x]) * p.CB + (c - p.CA)];
                }
                const float g = p.dXn[v * C + c];
                s1 += (double)g;
                s2 += (double)g * (double)((xv - mu) * rs);
            }
        }
        sm[(t * 2)] = s1; sm[t * 2 + 1] = s2;
        __syncthreads();
        if (part_i == 0 && c < C) {
            for (int q = 1; q < RP; ++q) { s1 += sm[(q * CP + t % CP) * 2]; s2 += sm[(q * CP + t % CP) * 2 + 1]; }
            part[((int64_t)blockIdx.x * C + c) * 2] = s1;
            part[((int64_t)blockIdx.x * C + c) * 2 + 1] = s2;
        }
        __syncthreads();
    }
}

// the same partial sums with four channels per thread (C, CA, CB multiples of 4, C/4 <= 256) and four voxels in
// flight: the scalar version above ran at 150 GB/s (one dependent fp64 chain and one 4-byte load per thread at a time)
__global__ void __launch_bounds__(256) gn_bwd_partial4_kernel(const GnbParams p, int64_t vox_per_block,
                                                              double* __restrict__ part) {
    extern __shared__ double sm[];                        // [256][8]
    const int C = p.CA + p.CB;
    const int CQall = C >> 2;
    const int t = threadIdx.x;
    const int64_t nvox = (int64_t)p.D * p.H * p.W;
    const int64_t v0 = (int64_t)blockIdx.x * vox_per_block, v1 = min(nvox, v0 + vox_per_block);
    const int cpg = C / p.G;
    for (int q0 = 0; q0 < CQall; q0 += 256) {             // channel chunks of 1024 (the deep decoder layers have 3072)
        const int CQ = min(256, CQall - q0);
        const int RP = 256 / CQ;
        const int q = t % CQ, lane_v = t / CQ;
        const int c = (q0 + q) * 4;
        double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
        if (lane_v < RP) {
            float mu[4], rs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { mu[k] = p.mean[(c + k) / cpg]; rs[k] = p.rstd[(c + k) / cpg]; }
            const bool fromB = c >= p.CA;
            auto src = [&](int64_t v) -> const float* {
                if (!fromB) return p.A + v * p.CA + c;
                const int x = (int)(v % p.W);
                const int64_t t2 = v / p.W;
                const int y = (int)(t2 % p.H), z = (int)(t2 / p.H);
                return p.B + ((int64_t)(p.up.mapD[z] * p.up.h + p.up.mapH[y]) * p.up.w + p.up.mapW[x]) * p.CB + (c - p.CA);
            };
            int64_t v = v0 + lane_v;
            for (; v + 3 * (int64_t)RP < v1; v += 4 * (int64_t)RP) {
                float4 xv[4], g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = *reinterpret_cast<const float4*>(src(v + (int64_t)u * RP));
                    g[u] = *reinterpret_cast<const float4*>(p.dXn + (v + (int64_t)u * RP) * C + c);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, gs[4] = {g[u].x, g[u].y, g[u].z, g[u].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s1[k] += (double)gs[k];
                        s2[k] += (double)gs[k] * (double)((xs[k] - mu[k]) * rs[k]);
                    }
                }
            }
            for (; v < v1; v += RP) {
                const float4 xv = *reinterpret_cast<const float4*>(src(v));
                const float4 g = *reinterpret_cast<const float4*>(p.dXn + v * C + c);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1[k] += (double)gs[k];
                    s2[k] += (double)gs[k] * (double)((xs[k] - mu[k]) * rs[k]);
                }
            }
        }
        __syncthreads();                                  // the previous chunk's reads of sm are done
#pragma unroll
        for (int k = 0; k < 4; ++k) { sm[t * 8 + 2 * k] = s1[k]; sm[t * 8 + 2 * k + 1] = s2[k]; }
        __syncthreads();
        if (t < CQ) {
            for (int r = 1; r < RP; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1[k] += sm[(r * CQ + t) * 8 + 2 * k]; s2[k] += sm[(r * CQ + t) * 8 + 2 * k + 1]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                part[((int64_t)blockIdx.x * C + c + k) * 2] = s1[k];
                part[((int64_t)blockIdx.x * C + c + k) * 2 + 1] = s2[k];
            }
        }
    }
}

// one block: channel totals (rows in order) -> dgamma, dbeta; group means m1, m2
__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(const double* __restrict__ part, int nb, int C, int G,
                                                              double count_per_channel, const float* __restrict__ gamma,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ m1, float* __restrict__ m2) {
    // one block per group: threads = (channel of the group) x (row lane); rows folded in a fixed order
    extern __shared__ double sm[];                        // [256][2] partials, then [cpg][2]: gamma*s1, gamma*s2
    const int t = threadIdx.x;
    const int g = blockIdx.x;
    const int cpg = C / G;
    double ga = 0.0, gb = 0.0;                            // thread 0: group sums over channel chunks
    for (int c0 = 0; c0 < cpg; c0 += 256) {
        const int CP = min(256, cpg - c0);
        const int RP = 256 / CP;
        const int ci = t % CP, rl = t / CP;
        const int c = g * cpg + c0 + ci;
        double s1 = 0.0, s2 = 0.0;
        if (rl < RP)
            for (int b = rl; b < nb; b += RP) { s1 += part[((int64_t)b * C + c) * 2]; s2 += part[((int64_t)b * C + c) * 2 + 1]; }
        __syncthreads();
        sm[t * 2] = s1; sm[t * 2 + 1] = s2;
        __syncthreads();
        if (t < CP) {
            for (int r = 1; r < RP; ++r) { s1 += sm[(r * CP + t) * 2]; s2 += sm[(r * CP + t) * 2 + 1]; }
            dbeta[c] = (float)s1;
            dgamma[c] = (float)s2;
        }
        __syncthreads();
        if (t < CP) { sm[t * 2] = (double)gamma[c] * s1; sm[t * 2 + 1] = (double)gamma[c] * s2; }
        __syncthreads();
        if (t == 0)
            for (int k = 0; k < CP; ++k) { ga += sm[k * 2]; gb += sm[k * 2 + 1]; }
    }
    if (t == 0) {
        const double n = count_per_channel * (double)cpg;
        m1[g] = (float)(ga / n);
        m2[g] = (float)(gb / n);
    }
}

// dA[v][c] for the skip channels (c < CA)
__global__ void gn_bwd_apply_a_kernel(const GnbParams p, const float* __restrict__ m1, const float* __restrict__ m2,
                                      float* __restrict__ dA) {
    const int C = p.CA + p.CB;
    const int cpg = C / p.G;
    const int64_t n = (int64_t)p.D * p.H * p.W * p.CA;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % p.CA);
        const int64_t v = i / p.CA;
        const int g = c / cpg;
        const float xh = (p.A[i] - p.mean[g]) * p.rstd[g];
        dA[i] = p.rstd[g] * ((p.gamma[c] * p.dXn[v * C + c] - m1[g]) - xh * m2[g]);
    }
}

// dB[u][c] for the upsampled channels: sum over the box of replicas of low-res voxel u
// startD/H/W[u] = first full-res index mapped to u (exclusive prefix sum of the replication counts)
__global__ void gn_bwd_apply_b_kernel(const GnbParams p, const float* __restrict__ m1, const float* __restrict__ m2,
                                      const int32_t* __restrict__ startD, const int32_t* __restrict__ startH,
                                      const int32_t* __restrict__ startW, const int32_t* __restrict__ repD,
                                      const int32_t* __restrict__ repH, const int32_t* __restrict__ repW,
                                      float* __restrict__ dB) {
    const int C = p.CA + p.CB;
    const int cpg = C / p.G;
    const int64_t n = (int64_t)p.up.d * p.up.h * p.up.w * p.CB;
    GRID_STRIDE(i, n) {
        const int cb = (int)(i % p.CB);
        int64_t u = i / p.CB;
        const int ux = (int)(u % p.up.w); u /= p.up.w;
        const int uy = (int)(u % p.up.h);
        const int uz = (int)(u / p.up.h);
        const int c = p.CA + cb;
        const int g = c / cpg;
        float s = 0.f;
        for (int z = startD[uz]; z < startD[uz] + repD[uz]; ++z)
            for (int y = startH[uy]; y < startH[uy] + repH[uy]; ++y)
                for (int x = startW[ux]; x < startW[ux] + repW[ux]; ++x)
                    s += p.dXn[((int64_t)(z * p.H + y) * p.W + x) * C + c];
        const float cnt = (float)(repD[uz] * repH[uy] * repW[ux]);
        const float xh = (p.B[i] - p.mean[g]) * p.rstd[g];
        dB[i] = p.rstd[g] * ((p.gamma[c] * s - cnt * m1[g]) - cnt * (xh * m2[g]));
    }
}

// ----------------------------------------------------------------------------- MaxPool3d(2) backward
// dIn (zero-filled by this kernel) gets dOut at the first maximum of each 2x2x2 window in (dz,dy,dx) scan order
__global__ void maxpool2_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dOut, int C, int D, int H,
                                    int W, float* __restrict__ dIn) {
    const int d = D / 2, h = H / 2, w = W / 2;
    const int64_t n = (int64_t)D * H * W * C;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C);
        int64_t v = i / C;
        const int x = (int)(v % W); v /= W;
        const int y = (int)(v % H);
        const int z = (int)(v / H);
        const int uz = z >> 1, uy = y >> 1, ux = x >> 1;
        float r = 0.f;
        if (uz < d && uy < h && ux < w) {
            // is (z,y,x) the first maximum of its window?
            const float me = in[i];
            bool first = true;
            for (int q = 0; q < 8 && first; ++q) {
                const int zz = 2 * uz + (q >> 2), yy = 2 * uy + ((q >> 1) & 1), xx = 2 * ux + (q & 1);
                const float o = in[((int64_t)(zz * H + yy) * W + xx) * C + c];
                const bool before = (zz < z) || (zz == z && (yy < y || (yy == y && xx < x)));
                if (o > me || (before && o == me)) first = false;
            }
            if (first) r = dOut[((int64_t)(uz * h + uy) * w + ux) * C + c];
        }
        dIn[i] = r;
    }
}

// window-centric form for C % 4 == 0: one thread per (2x2x2 window, channel quad) reads the window once (8 float4), finds
// the first maximum per channel in scan order and writes the 8 outputs; the element-centric kernel above re-reads the
// window from every one of its 8 voxels (0.8 TB/s at 128^3).  Odd trailing slices (outside every window) are zero-filled
// by the threads of the last window along that axis.
__global__ void maxpool2_bwd4_kernel(const float* __restrict__ in, const float* __restrict__ dOut, int C, int D, int H,
                                     int W, float* __restrict__ dIn) {
    const int d = D / 2, h = H / 2, w = W / 2, CQ = C >> 2;
    const int64_t n = (int64_t)d * h * w * CQ;
    GRID_STRIDE(i, n) {
        const int q = (int)(i % CQ);
        int64_t u = i / CQ;
        const int ux = (int)(u % w); u /= w;
        const int uy = (int)(u % h);
        const int uz = (int)(u / h);
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int zz = 2 * uz + (k >> 2), yy = 2 * uy + ((k >> 1) & 1), xx = 2 * ux + (k & 1);
            v[k] = *reinterpret_cast<const float4*>(in + ((int64_t)(zz * H + yy) * W + xx) * C + q * 4);
        }
        const float4 g = *reinterpret_cast<const float4*>(dOut + ((int64_t)(uz * h + uy) * w + ux) * C + q * 4);
        int am[4] = {0, 0, 0, 0};
        float mx[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (e[c] > mx[c]) { mx[c] = e[c]; am[c] = k; }            // strict: the first maximum wins
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int zz = 2 * uz + (k >> 2), yy = 2 * uy + ((k >> 1) & 1), xx = 2 * ux + (k & 1);
            const float4 o = make_float4(am[0] == k ? g.x : 0.f, am[1] == k ? g.y : 0.f, am[2] == k ? g.z : 0.f,
                                         am[3] == k ? g.w : 0.f);
            *reinterpret_cast<float4*>(dIn + ((int64_t)(zz * H + yy) * W + xx) * C + q * 4) = o;
        }
        // odd trailing slices
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool lx = (ux == w - 1) && (W & 1), ly = (uy == h - 1) && (H & 1), lz = (uz == d - 1) && (D & 1);
        const int x1 = lx ? 3 : 2, y1 = ly ? 3 : 2, z1 = lz ? 3 : 2;
        if (lx || ly || lz)
            for (int dz = 0; dz < z1; ++dz)
                for (int dy = 0; dy < y1; ++dy)
                    for (int dx = 0; dx < x1; ++dx)
                        if (dz == 2 || dy == 2 || dx == 2)
                            *reinterpret_cast<float4*>(dIn + ((int64_t)((2 * uz + dz) * H + 2 * uy + dy) * W + 2 * ux + dx) * C + q * 4) = z4;
    }
}

int grid_for(int64_t n) { return (int)std::min<int64_t>(4096, bfm_cdiv64(n, 256)); }

}  // namespace

extern "C" int bfm_lrelu_bwd_ex(const float* dY, const float* Y, int64_t n, float slope, float* dP, float* absmax,
                                bfm_stream_t stream) {
    if (!dY || !Y || !dP || n <= 0) return BFM_E_ARG;
    if (n % 4) return BFM_E_SHAPE;
    if (absmax && hipMemsetAsync(absmax, 0, sizeof(float), bfm_s(stream)) != hipSuccess) return BFM_E_LAUNCH;
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, bfm_s(stream), dY, Y, n / 4, slope, dP, absmax);
    return bfm_launch_status();
}

extern "C" int bfm_lrelu_bwd(const float* dY, const float* Y, int64_t n, float slope, float* dP, bfm_stream_t stream) {
    return bfm_lrelu_bwd_ex(dY, Y, n, slope, dP, nullptr, stream);
}

namespace {
struct WgPlan { int kind; int S; int rows_per_split; int nbz, nby, nbx, ntiles; };    // kind 0 tiled (fp32 or f16 tiles), 1 columns
WgPlan wgrad_plan(int CA, int Cin, int Cout, int D, int H, int W, bool f16 = false) {
    WgPlan pl{};
    if (Cout % 64 == 0 && Cin % 32 == 0 && CA % 32 == 0) {
        pl.kind = 0;
        pl.nbz = bfm_cdiv(D, f16 ? HT_Z : WT_Z); pl.nby = bfm_cdiv(H, f16 ? HT_Y : WT_Y); pl.nbx = bfm_cdiv(W, f16 ? HT_X : WT_X);
        pl.ntiles = pl.nbz * pl.nby * pl.nbx;
        const int cols = (Cin / 32) * (Cout / 64);
        int S = 512 / cols;                               // two full rounds of one workgroup per CU, never a third
        if (S > pl.ntiles) S = pl.ntiles;
        if (S < 1) S = 1;
        const int per = bfm_cdiv(pl.ntiles, S);
        pl.S = bfm_cdiv(pl.ntiles, per);                  // every workgroup gets `per` tiles, the last maybe fewer
    } else {
        pl.kind = 1;
        const int nrows = D * H;
        const int blocks = bfm_cdiv(Cout, 32) * bfm_cdiv(Cin * 27, 32);
        int S = bfm_cdiv(2048, blocks);
        if (S > nrows) S = nrows;
        if (S < 1) S = 1;
        pl.rows_per_split = bfm_cdiv(nrows, S);
        pl.S = bfm_cdiv(nrows, pl.rows_per_split);
    }
    return pl;
}
}  // namespace

namespace {
// conv_wgrad_ws_kernel needs 130 KB of dynamic LDS and 768 threads: asked for once per process; where the device refuses
// (a 64 KB-LDS part), the callers keep the 8-wave conv_wgrad_f16_kernel plan instead of failing the step
bool wgrad_ws_available() {
    static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_ws_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, WGS_LDS) == hipSuccess;
    return ok;
}

// skip / plain channels [0, CA) of a layer through conv_wgrad_ws_kernel + the fold of its partials into dW [Cout][Cin][27]
int launch_wgrad_ws(const float* dP, int Cout, const float* A, int CA, int Cin, int D, int H, int W, const float* scale,
                    const float* shift, const float* dp_bound, const float* x_bound, int G, float* part, float* dW,
                    hipStream_t st) {
    WgWsParams w{};
    w.dP = dP; w.A = A; w.scale = scale; w.shift = shift; w.Cout = Cout; w.CA = CA; w.D = D; w.H = H; w.W = W;
    w.part = part; w.part_cin = CA;
    w.nby = bfm_cdiv(H, ST_Y); w.nbx = bfm_cdiv(W, ST_X); w.ntiles = D * w.nby * w.nbx;
    w.dp_bound = dp_bound; w.x_bound = x_bound; w.G = G;
    const int cols = (CA / 32) * (Cout / 32);
    int S = 512 / cols;                                   // two rounds of one workgroup per CU (130 KB of LDS each)
    if (S > w.ntiles) S = w.ntiles;
    if (S < 1) S = 1;
    const int per = bfm_cdiv(w.ntiles, S);
    S = bfm_cdiv(w.ntiles, per);
    hipLaunchKernelGGL(conv_wgrad_ws_kernel, dim3(S, CA / 32, Cout / 32), dim3(WGS_THREADS), WGS_LDS, st, w);
    hipLaunchKernelGGL(wgrad_reduce_cols_kernel, dim3(grid_for((int64_t)Cout * CA * 27)), dim3(256), 0, st, part, S, Cout, CA,
                       Cin, dW);
    return BFM_OK;
}
size_t wgrad_ws_part_bytes(int Cout, int CA, int D, int H, int W) {
    const int ntiles = D * bfm_cdiv(H, ST_Y) * bfm_cdiv(W, ST_X);
    const int cols = (CA / 32) * (Cout / 32);
    int S = cols > 0 ? 512 / cols : 1;
    if (S > ntiles) S = ntiles;
    if (S < 1) S = 1;
    return (((size_t)S * Cout * CA * 27 * sizeof(float)) + 255) & ~(size_t)255;
}
}  // namespace

extern "C" size_t bfm_conv3x3x3_wgrad_workspace(int Cin, int Cout, int D, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    // the split count does not depend on CA (only the kernel choice does): take the larger of the two plans
    const WgPlan a = wgrad_plan(Cin, Cin, Cout, D, H, W), b = wgrad_plan(1, Cin, Cout, D, H, W),
                 c = wgrad_plan(Cin, Cin, Cout, D, H, W, true);
    int S = a.S > b.S ? a.S : b.S;
    if (c.S > S) S = c.S;
    // + the folded partials of the upsampled channels (conv_wgrad_up_f16_kernel): S_U x Cout x CB x 64 floats with
    //   S_U x CB <= max(CB, 512 x 32 x 32 / Cout), CB < Cin
    const size_t up = (size_t)256 * std::max<size_t>((size_t)Cout * Cin, (size_t)512 * 1024);
    // (the skip channels alone split finer: S_A x CA <= 512 x 32 x 64 / Cout)
    const size_t reg = std::max((size_t)S * Cout * Cin * 27 * sizeof(float), (size_t)512 * 32 * 64 * 27 * sizeof(float));
    return reg + 256 + up;
}

extern "C" int bfm_conv3x3x3_wgrad_ex(const float* dP, int Cout, const float* A, int CA, const float* B, int CB, int D,
                                      int H, int W, const bfm_upsample_t* up, const float* scale, const float* shift,
                                      const float* dp_bound, const float* x_bound, int G, int passes, float* dW,
                                      void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (passes != 0 && passes != 3) return BFM_E_ARG;
    if (passes == 3 && (!dp_bound || !x_bound || G <= 0)) return BFM_E_ARG;
    if (!dP || !A || !scale || !shift || !dW || !workspace || Cout <= 0 || CA <= 0 || D <= 0 || H <= 0 || W <= 0)
        return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!B || !up || !up->mapD || !up->mapH || !up->mapW))) return BFM_E_ARG;
    const int Cin = CA + CB;
    if (workspace_bytes < bfm_conv3x3x3_wgrad_workspace(Cin, Cout, D, H, W)) return BFM_E_WORKSPACE;
    WgParams p{};
    p.dP = dP; p.Cout = Cout; p.A = A; p.B = B; p.CA = CA; p.CB = CB; p.up = make_upview(up);
    p.scale = scale; p.shift = shift; p.D = D; p.H = H; p.W = W;
    p.part = static_cast<float*>(workspace);
    hipStream_t st = bfm_s(stream);
    const int64_t n = (int64_t)Cout * Cin * 27;
    const bool f16 = passes == 3 && Cout % 64 == 0 && Cin % 32 == 0 && CA % 32 == 0;
    // upsampled channels of an exact 2x decoder join: folded form on the low-res tensor, the skip channels alone below
    static const bool upfold_on = []() { const char* e = getenv("BFM_WGRAD_UPFOLD"); return !(e && e[0] == '0'); }();
    static const bool ws_on = []() { const char* e = getenv("BFM_WGRAD_WS"); return !(e && e[0] == '0'); }() && wgrad_ws_available();
    if (f16 && ws_on && CB == 0) {                         // plain layer: the wave-specialised kernel on all channels
        if (wgrad_ws_part_bytes(Cout, CA, D, H, W) > workspace_bytes) return BFM_E_WORKSPACE;
        const int rc = launch_wgrad_ws(dP, Cout, A, CA, Cin, D, H, W, scale, shift, dp_bound, x_bound, G, p.part, dW, st);
        return rc != BFM_OK ? rc : bfm_launch_status();
    }
    if (f16 && upfold_on && CB > 0 && CB % 32 == 0 && 2 * up->d == D && 2 * up->h == H && 2 * up->w == W &&
        up->h >= UT_Y && up->w >= UT_X) {
        hipStream_t st2 = st;
        // (1) skip channels alone, compact partials
        static bool attr3a = false, attrU = false;
        size_t offA;
        if (ws_on) {
            const int rc = launch_wgrad_ws(dP, Cout, A, CA, Cin, D, H, W, scale, shift, dp_bound, x_bound, G, p.part, dW, st2);
            if (rc != BFM_OK) return rc;
            offA = wgrad_ws_part_bytes(Cout, CA, D, H, W);
        } else {
            const WgPlan pa = wgrad_plan(CA, CA, Cout, D, H, W, true);
            Wg3Params q{};
            q.b = p; q.b.S = pa.S; q.nbz = pa.nbz; q.nby = pa.nby; q.nbx = pa.nbx; q.ntiles = pa.ntiles;
            q.dp_bound = dp_bound; q.x_bound = x_bound; q.G = G; q.part_cin = CA;
            if (!attr3a) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_f16_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, WG3_LDS) != hipSuccess)
                    return BFM_E_LAUNCH;
                attr3a = true;
            }
            hipLaunchKernelGGL(conv_wgrad_f16_kernel, dim3(pa.S, CA / 32, Cout / 64), dim3(WG3_THREADS), WG3_LDS, st2, q);
            hipLaunchKernelGGL(wgrad_reduce_cols_kernel, dim3(grid_for((int64_t)Cout * CA * 27)), dim3(256), 0, st2, p.part, pa.S,
                               Cout, CA, Cin, dW);
            offA = (((size_t)pa.S * Cout * CA * 27 * sizeof(float)) + 255) & ~(size_t)255;
        }
        // (2) upsampled channels
        WgUpParams u{};
        u.dP = dP; u.L = B; u.scale = scale + CA; u.shift = shift + CA; u.Cout = Cout; u.CB = CB;
        u.d = up->d; u.h = up->h; u.w = up->w;
        u.nby = bfm_cdiv(up->h, UT_Y); u.nbx = bfm_cdiv(up->w, UT_X); u.ntiles = up->d * u.nby * u.nbx;
        u.dp_bound = dp_bound; u.x_bound = x_bound; u.G = G;
        const int colsU = (CB / 32) * (Cout / 32);
        int SU = 512 / colsU;
        if (SU > u.ntiles) SU = u.ntiles;
        if (SU < 1) SU = 1;
        const int per = bfm_cdiv(u.ntiles, SU);
        SU = bfm_cdiv(u.ntiles, per);
        if (offA + (size_t)SU * Cout * CB * 64 * sizeof(float) > workspace_bytes) return BFM_E_WORKSPACE;
        u.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + offA);
        if (!attrU) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_up_f16_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, WGU_LDS) != hipSuccess)
                return BFM_E_LAUNCH;
            attrU = true;
        }
        hipLaunchKernelGGL(conv_wgrad_up_f16_kernel, dim3(SU, CB / 32, Cout / 32), dim3(WGU_THREADS), WGU_LDS, st2, u);
        hipLaunchKernelGGL(wgrad_up_fold_kernel, dim3((unsigned)std::min<int64_t>(8192, (int64_t)Cout * CB / 16)), dim3(256), 0,
                           st2, u.part, SU, Cout, CB, Cin, CA, dW);
        return bfm_launch_status();
    }
    const WgPlan pl = wgrad_plan(CA, Cin, Cout, D, H, W, f16);
    p.S = pl.S;
    if (f16) {
        Wg3Params q{};
        q.b = p; q.nbz = pl.nbz; q.nby = pl.nby; q.nbx = pl.nbx; q.ntiles = pl.ntiles;
        q.dp_bound = dp_bound; q.x_bound = x_bound; q.G = G; q.part_cin = Cin;
        static bool attr3 = false;
        if (!attr3) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_f16_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, WG3_LDS) != hipSuccess)
                return BFM_E_LAUNCH;
            attr3 = true;
        }
        hipLaunchKernelGGL(conv_wgrad_f16_kernel, dim3(pl.S, Cin / 32, Cout / 64), dim3(WG3_THREADS), WG3_LDS, st, q);
    } else if (pl.kind == 0) {
        Wg2Params q{};
        q.b = p; q.nbz = pl.nbz; q.nby = pl.nby; q.nbx = pl.nbx; q.ntiles = pl.ntiles;
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tiled_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, WG2_LDS) != hipSuccess)
                return BFM_E_LAUNCH;
            attr = true;
        }
        hipLaunchKernelGGL(conv_wgrad_tiled_kernel, dim3(pl.S, Cin / 32, Cout / 64), dim3(WG2_THREADS), WG2_LDS, st, q);
    } else {
        p.rows_per_split = pl.rows_per_split;
        hipLaunchKernelGGL(conv_wgrad_cols_kernel, dim3(pl.S, bfm_cdiv(Cout, 32), bfm_cdiv(Cin * 27, 32)), dim3(64), 0, st, p);
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(n)), dim3(256), 0, st, p.part, p.S, n, dW);
    return bfm_launch_status();
}

extern "C" int bfm_conv3x3x3_wgrad(const float* dP, int Cout, const float* A, int CA, const float* B, int CB, int D,
                                   int H, int W, const bfm_upsample_t* up, const float* scale, const float* shift,
                                   float* dW, void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    return bfm_conv3x3x3_wgrad_ex(dP, Cout, A, CA, B, CB, D, H, W, up, scale, shift, nullptr, nullptr, 0, 0, dW, workspace,
                                  workspace_bytes, stream);
}

// partial blocks of the GroupNorm backward sums: 16 voxels or more each, at most 1024 (the deep levels have 64..4096
// voxels with up to 3072 channels: one or two blocks of 2048 voxels took 3.4 / 7.2 ms there)
static int64_t gnb_blocks(int64_t nvox) {
    int64_t nb = bfm_cdiv64(nvox, 16);
    if (nb > 1024) nb = 1024;
    return nb < 1 ? 1 : nb;
}

extern "C" size_t bfm_gn_bwd_workspace(int C, int D, int H, int W) {
    return (size_t)gnb_blocks((int64_t)D * H * W) * C * 2 * sizeof(double) + 256;
}

// dXn [D][H][W][CA+CB] -> dA [D][H][W][CA], dB [d][h][w][CB] (when CB > 0), dgamma/dbeta [CA+CB].
// start*/rep* : per low-res index, the first full-res index and the number of full-res indices mapped to it.
extern "C" int bfm_gn_bwd(const float* dXn, const float* A, int CA, const float* B, int CB, int D, int H, int W,
                          const bfm_upsample_t* up, const int32_t* startD, const int32_t* startH, const int32_t* startW,
                          const float* mean, const float* rstd, const float* gamma, int G, float* dA, float* dB,
                          float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, bfm_stream_t stream) {
    if (!dXn || !A || CA <= 0 || D <= 0 || H <= 0 || W <= 0 || !mean || !rstd || !gamma || G <= 0 || !dA || !dgamma ||
        !dbeta || !workspace)
        return BFM_E_ARG;
    if (CB < 0 || (CB > 0 && (!B || !up || !up->mapD || !up->repD || !startD || !startH || !startW || !dB)))
        return BFM_E_ARG;
    const int C = CA + CB;
    if (C % G) return BFM_E_SHAPE;
    if (workspace_bytes < bfm_gn_bwd_workspace(C, D, H, W)) return BFM_E_WORKSPACE;
    GnbParams p{};
    p.dXn = dXn; p.A = A; p.B = B; p.CA = CA; p.CB = CB; p.up = make_upview(up);
    p.D = D; p.H = H; p.W = W; p.G = G; p.mean = mean; p.rstd = rstd; p.gamma = gamma;
    hipStream_t st = bfm_s(stream);
    const int64_t nvox = (int64_t)D * H * W;
    int64_t nb = gnb_blocks(nvox);
    const int64_t vpb = bfm_cdiv64(nvox, nb);
    nb = bfm_cdiv64(nvox, vpb);
    char* ws = static_cast<char*>(workspace);
    double* part = reinterpret_cast<double*>(ws);
    float* m1 = reinterpret_cast<float*>(ws + (size_t)nb * C * 16);
    float* m2 = m1 + 32;
    if (G > 32) return BFM_E_SHAPE;
    if (CA % 4 == 0 && CB % 4 == 0)
        hipLaunchKernelGGL(gn_bwd_partial4_kernel, dim3((unsigned)nb), dim3(256), 256 * 64, st, p, vpb, part);
    else
        hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3((unsigned)nb), dim3(256), 256 * 16, st, p, vpb, part);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(G), dim3(256), 256 * 16, st, part, (int)nb, C, G, (double)nvox,
                       gamma, dgamma, dbeta, m1, m2);
    hipLaunchKernelGGL(gn_bwd_apply_a_kernel, dim3(grid_for(nvox * CA)), dim3(256), 0, st, p, m1, m2, dA);
    if (CB > 0) {
        const int64_t nlo = (int64_t)up->d * up->h * up->w * CB;
        hipLaunchKernelGGL(gn_bwd_apply_b_kernel, dim3(grid_for(nlo)), dim3(256), 0, st, p, m1, m2, startD, startH, startW,
                           up->repD, up->repH, up->repW, dB);
    }
    return bfm_launch_status();
}

extern "C" int bfm_maxpool2_bwd(const float* in, const float* dOut, int C, int D, int H, int W, float* dIn,
                                bfm_stream_t stream) {
    if (!in || !dOut || !dIn || C <= 0 || D < 2 || H < 2 || W < 2) return BFM_E_ARG;
    const int64_t n = (int64_t)D * H * W * C;
    if (C % 4 == 0) {
        const int64_t nq = (int64_t)(D / 2) * (H / 2) * (W / 2) * (C / 4);
        hipLaunchKernelGGL(maxpool2_bwd4_kernel, dim3(grid_for(nq)), dim3(256), 0, bfm_s(stream), in, dOut, C, D, H, W, dIn);
    } else {
        hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, bfm_s(stream), in, dOut, C, D, H, W, dIn);
    }
    return bfm_launch_status();
}

// dW'[ci][co][26 - t] = w[co][ci][t] for ci < Cin, 0 for Cin <= ci < CinPad: the weights of the data-gradient conv (transposed,
// tap-mirrored, output channels padded for the matrix-core kernels) in one pass; torch needed permute + flip + pad +
// contiguous + copy_ (five passes over 264 M parameters every training iteration).
namespace {
__global__ void transpose_mirror_kernel(const float* __restrict__ w, int Cout, int Cin, int CinPad, float* __restrict__ out) {
    const int64_t n = (int64_t)CinPad * Cout * 27;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % 27);
        const int64_t r = i / 27;
        const int co = (int)(r % Cout);
        const int ci = (int)(r / Cout);
        out[i] = ci < Cin ? w[((int64_t)co * Cin + ci) * 27 + (26 - t)] : 0.f;
    }
}

// The same through LDS for Cout % 32 == 0 and Cin % 16 == 0: a block moves a 32 (co) x 16 (ci) x 27 tile -- 32 runs of 432
// contiguous floats in, 16 runs of 864 contiguous floats out (the gather above reads 108-byte runs a whole input row apart).
constexpr int TM_ROW = 16 * 27 + 1;
__global__ void __launch_bounds__(256) transpose_mirror_tiled(const float* __restrict__ w, int Cout, int Cin,
                                                              float* __restrict__ out) {
    __shared__ float tm_lds[32 * TM_ROW];
    const int nco = Cout / 32;
    const int cob = blockIdx.x % nco, cib = blockIdx.x / nco;
    const float* src0 = w + ((int64_t)(cob * 32) * Cin + cib * 16) * 27;
    bfm_stage_rows<32, 16 * 27, TM_ROW, 256>(src0, (int64_t)Cin * 27, tm_lds, 1.0f,
                                             ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && (Cin & 3) == 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 32 * 27; i += 256) {   // out[ci][co][t]: 32 * 27 contiguous floats per ci
        const int ci = i / (32 * 27), r = i - ci * (32 * 27);
        const int co = r / 27, t = r - co * 27;
        out[((int64_t)(cib * 16 + ci) * Cout + cob * 32) * 27 + r] = tm_lds[co * TM_ROW + ci * 27 + (26 - t)];
    }
}
__global__ void zero_rows_kernel(float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = 0.f;
}
}  // namespace

extern "C" int bfm_transpose_mirror_weights(const float* w, int Cout, int Cin, int CinPad, float* out, bfm_stream_t stream) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || CinPad < Cin) return BFM_E_ARG;
    if (Cout % 32 == 0 && Cin % 16 == 0) {
        hipLaunchKernelGGL(transpose_mirror_tiled, dim3((unsigned)((Cout / 32) * (Cin / 16))), dim3(256), 0, bfm_s(stream), w,
                           Cout, Cin, out);
        if (CinPad > Cin) {                                   // the zero rows of the padded output channels
            const int64_t nz = (int64_t)(CinPad - Cin) * Cout * 27;
            hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)std::min<int64_t>(2048, bfm_cdiv64(nz, 256))), dim3(256), 0,
                               bfm_s(stream), out + (int64_t)Cin * Cout * 27, nz);
        }
        return bfm_launch_status();
    }
    const int64_t n = (int64_t)CinPad * Cout * 27;
    const int nb = (int)std::min<int64_t>(8192, bfm_cdiv64(n, 256));
    hipLaunchKernelGGL(transpose_mirror_kernel, dim3(nb), dim3(256), 0, bfm_s(stream), w, Cout, Cin, CinPad, out);
    return bfm_launch_status();
}
