// The upsampled half of a decoder's first convolution with the nearest 2x upsample folded into the weights (as
// conv3d_upfold.hip: Decoder._joining, Trainer/models/unet3d/buildingblocks.py:265-276, 361-363) and, along x, three
// products for the two outputs over a low-res voxel instead of four.
//
// Along an axis the two outputs over low-res voxel i are (conv3d_upfold.hip)
//      o0 = w0 x(i-1) + (w1 + w2) x(i)            o1 = (w0 + w1) x(i) + w2 x(i+1)
// and with P = (w0 + w1 + w2) x(i):
//      o0 = P + w0 (x(i-1) - x(i))                o1 = P + w2 (x(i+1) - x(i))
// -- a transform with coefficients +-1 only: V- = x(i-1) - x(i), V0 = x(i), V+ = x(i+1) - x(i) on the input side,
// U- = w0, U0 = w0 + w1 + w2, U+ = w2 on the weight side, one addition per output.  Applied along x (z and y keep the
// folded 2-tap form), a low-res voxel's 8 outputs take 4 (pz,py) x 3 (x position) x 4 (kd,kh) = 48 products per
// (ci, co) instead of 64: 25 % fewer MFMAs for the same fp32-grade result (no amplification: each output still
// depends, numerically too, on its own three inputs along x only).
//
// GEMM view: M = low-res voxels (a box of 64 per workgroup), N = 12 classes x Cout, K = 4 taps x CB.  8 waves; the
// twelve classes (pz, py, x position) go to the waves the way conv3d_wino4.hip's positions do: wave w owns class w
// (both 32-row blocks) and one row block of class 8 + (w >> 1) -- three units, 18 MFMAs per tap and wave, 96
// accumulator registers, two classes' weight fragments streamed L2 -> VGPR one tap ahead.  The transformed, split box
// lives in LDS ([chunk of three][x position][k-half][hi|lo][row][x][8 ch], 84 KB).  Epilogue, per (pz,py) and column block: the three
// classes meet in LDS and one thread per (voxel, cout) writes o0 = (P + P-) dq, o1 = (P + P+) dq to the interleaved
// full-res positions of `out` (no activation: the skip half accumulates onto it and applies LeakyReLU).
// No split-K, no batch: the full-resolution decoders (where conv_upfold spends 18 of its 24 ms per 256^3 volume).
#include "bfm_common.h"
#include <cstdlib>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));

constexpr int KC = 16;
constexpr int NCLS = 12;
constexpr int NTHR = 512;
constexpr int NRG = NTHR / 32;     // row groups of the epilogue (16)
constexpr int MLD = 33;            // epilogue LDS row stride in floats (odd: conflict-free)

struct UdParams {
    const float* B;
    int CB, d, h, w;                 // low-res tensor (channels-last)
    const float *scale, *shift;      // GroupNorm affine of the B channels
    const float* bound;
    int G;
    const uint4* wp;
    int wexp, Cout;
    float* out;                      // [2d][2h][2w][Cout]
    int BD, BH, BW, HT;              // low-res box (64 voxels) and its halo'd rows per slice
    int bw_shift, bhw_shift;         // log2(BW), log2(BH*BW)
    int nTy, nTx, nMt, NT, KCB;
    int npos_lds, plane_stride;      // (BD+2)*HT*BW positions; bytes per plane
};

__device__ __forceinline__ int row_perm(int l) {        // as conv_mfma: each 16-lane b128 group reads 16 consecutive positions
    if (l < 4) return l;
    if (l < 12) return l + 12;
    if (l < 16) return l - 8;
    if (l < 20) return l + 8;
    if (l < 28) return l - 12;
    return l;
}

__device__ __forceinline__ int row_unperm(int q) {      // inverse of row_perm
    if (q < 4) return q;
    if (q < 8) return q + 8;
    if (q < 16) return q + 12;
    if (q < 24) return q - 12;
    if (q < 28) return q - 8;
    return q;
}

__device__ __forceinline__ void box_coords(const UdParams& p, int q, int& bd, int& bh, int& bw) {
    bd = q >> p.bhw_shift;
    const int rem = q & ((1 << p.bhw_shift) - 1);
    bh = rem >> p.bw_shift;
    bw = rem & ((1 << p.bw_shift) - 1);
}

// x = hi + lo in fp16 for four values, two per instruction (truncation: x - hi is exact in fp32)
template <bool LO>
__device__ __forceinline__ void split_store4(const float (&t)[4], unsigned char* dp, int plane_stride) {
    const fp16x2_t h01 = __builtin_amdgcn_cvt_pkrtz(t[0], t[1]);
    const fp16x2_t h23 = __builtin_amdgcn_cvt_pkrtz(t[2], t[3]);
    uint2 hv;
    hv.x = __builtin_bit_cast(unsigned, h01);
    hv.y = __builtin_bit_cast(unsigned, h23);
    *reinterpret_cast<uint2*>(dp) = hv;
    if constexpr (LO) {
        const fp16x2_t l01 = __builtin_amdgcn_cvt_pkrtz(t[0] - (float)h01[0], t[1] - (float)h01[1]);
        const fp16x2_t l23 = __builtin_amdgcn_cvt_pkrtz(t[2] - (float)h23[0], t[3] - (float)h23[1]);
        uint2 lv;
        lv.x = __builtin_bit_cast(unsigned, l01);
        lv.y = __builtin_bit_cast(unsigned, l23);
        *reinterpret_cast<uint2*>(dp + plane_stride) = lv;
    }
}

template <int NPASS>
__global__ void __launch_bounds__(NTHR, 1) conv_updiff(const UdParams p) {
    constexpr int NPL = (NPASS == 3) ? 2 : 1;
    constexpr int NF = 2 * NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave 0..7
    const int clsA = wv;                                           // class owned entirely (units 0, 1 = row blocks 0, 1)
    const int clsB = 8 + (wv >> 1), mbB = wv & 1;                  // unit 2: row block mbB of class clsB
    const int l32 = lane & 31, khalf = lane >> 5;
    int item;
    {   // one workgroup per (box, cout tile), XCD-aware bijective remap
        const int nblk = p.nMt * p.NT;
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = item / p.NT;
    const int nt = item % p.NT;
    const int tx = mt % p.nTx;
    const int ty = (mt / p.nTx) % p.nTy;
    const int tz = mt / (p.nTx * p.nTy);
    const int z0 = tz * p.BD, y0 = ty * p.BH, x0 = tx * p.BW;      // low-res box origin

    float bmax = 0.f;
    for (int g = 0; g < p.G; ++g) bmax = fmaxf(bmax, p.bound[g]);
    int aexp = 0;
    if (bmax > 0.f && bmax < INFINITY) {
        int ex;
        (void)frexpf(bmax, &ex);
        aexp = 13 - ex;                                            // |V| <= 2 * bound
        aexp = aexp > 60 ? 60 : (aexp < -60 ? -60 : aexp);
    }
    const float a_scale = ldexpf(1.0f, aexp);
    const float dq = ldexpf(1.0f, -(aexp + p.wexp));

    // class c = 3 * (2 pz + py) + x position (0: -, 1: 0, 2: +).  A base offsets of the three units: low-res voxel
    // (bd,bh,bw) of class (pz,py,xp), tap (0,0) reads halo row (bd + pz, bh + py), column bw of plane xp
    int a_off[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int cls = u < 2 ? clsA : clsB, mb = u < 2 ? u : mbB;
        const int pp = cls / 3, xp = cls - pp * 3;
        const int pz = pp >> 1, py = pp & 1;
        int bd, bh, bw;
        box_coords(p, mb * 32 + row_perm(l32), bd, bh, bw);
        a_off[u] = ((xp * 2 + khalf) * NPL) * p.plane_stride + (((bd + pz) * p.HT + (bh + py)) * p.BW + bw) * 16;
    }

    // staging items: e = tid + it*NTHR -> (halo row, x, channel quad); off0 = element offset of low-res voxel
    // (gz, gy, x0 + bw - 1) channel 0 (may point outside the row: the mask says which of the 3 x positions exist)
    constexpr int MAX_IT = 2;
    const int n_el = p.npos_lds * 4;
    const int q4 = tid & 3;
    int off0[MAX_IT];
    int msk[MAX_IT];                                               // bit i: x position i inside the volume; -1: no item
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int e = tid + it * NTHR;
        off0[it] = 0;
        msk[it] = -1;
        if (e < n_el) {
            const int ps = e >> 2;
            const int bw = ps & ((1 << p.bw_shift) - 1);
            const int r = ps >> p.bw_shift;
            const int hz = r / p.HT, hy = r - hz * p.HT;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + bw - 1;
            int m = 0;
            if (gz >= 0 && gz < p.d && gy >= 0 && gy < p.h) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (gx + i >= 0 && gx + i < p.w) m |= 1 << i;
            }
            msk[it] = m;
            off0[it] = ((gz * p.h + gy) * p.w + gx) * p.CB;
        }
    }
    const int st_plane = ((q4 >> 1) * NPL) * p.plane_stride + (q4 & 1) * 8;

    floatx16 acc[3][2];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][nb][i] = 0.f;

    // this wave's two weight streams (classes clsA, clsB): S = KCB*4 steps (chunk-major, (kd,kh)-minor), NF fragments of
    // 64 x uint4 per step and class; three register sets, two steps ahead (12 steps per stage of three chunks: the set
    // index is the step index mod 3)
    const int S = p.KCB * 4;
    const uint4* wbA = p.wp + (size_t)(nt * NCLS + clsA) * S * (NF * 64) + lane;
    const uint4* wbB = p.wp + (size_t)(nt * NCLS + clsB) * S * (NF * 64) + lane;
    uint4 wq[3][2][NF];
    auto fetch = [&](int s, uint4 (&dst)[2][NF]) __attribute__((always_inline)) {
        const int sc = s < S ? s : S - 1;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            dst[0][f] = wbA[(size_t)sc * (NF * 64) + f * 64];
            dst[1][f] = wbB[(size_t)sc * (NF * 64) + f * 64];
        }
    };
    fetch(0, wq[0]);
    fetch(1, wq[1]);

    // three K chunks (48 channels) per stage: a chunk of this kernel is only 4 taps x 18 MFMAs per wave, too little to
    // carry a pair of barriers and a staging round trip of its own, and 12 steps keep the three-set weight ring in phase
    const int set_stride = 6 * NPL * p.plane_stride;
    for (int kc0 = 0; kc0 < p.KCB; kc0 += 3) {
        const int nch = min(3, p.KCB - kc0);
        __syncthreads();                                 // previous stage's readers are done
#pragma unroll
        for (int cs = 0; cs < 3; ++cs) {
            if (cs < nch) {
                const int c0 = (kc0 + cs) * KC;
                const float* src = p.B + c0 + q4 * 4;
                const float4 sc4 = *reinterpret_cast<const float4*>(p.scale + c0 + q4 * 4);
                const float4 sh4 = *reinterpret_cast<const float4*>(p.shift + c0 + q4 * 4);
                const float sc[4] = {sc4.x * a_scale, sc4.y * a_scale, sc4.z * a_scale, sc4.w * a_scale};
                const float sh[4] = {sh4.x * a_scale, sh4.y * a_scale, sh4.z * a_scale, sh4.w * a_scale};
#pragma unroll
                for (int it = 0; it < MAX_IT; ++it) {
                    if (msk[it] < 0) continue;
                    float4 v[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (msk[it] & (1 << i)) v[i] = *reinterpret_cast<const float4*>(src + off0[it] + i * p.CB);
                    }
                    float dd[3][4];                      // [x position][channel]: affine, zero padding after it
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const bool ok = msk[it] & (1 << i);
                        const float y[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) dd[i][c] = ok ? fmaf(y[c], sc[c], sh[c]) : 0.f;
                    }
                    const int e = tid + it * NTHR;
                    unsigned char* dst = lds + cs * set_stride + st_plane + (e >> 2) * 16;
                    float t[3][4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        t[0][c] = dd[0][c] - dd[1][c];   // V-
                        t[1][c] = dd[1][c];              // V0
                        t[2][c] = dd[2][c] - dd[1][c];   // V+
                    }
#pragma unroll
                    for (int xp = 0; xp < 3; ++xp)
                        split_store4<NPASS == 3>(t[xp], dst + (xp * 2 * NPL) * p.plane_stride, p.plane_stride);
                }
            }
        }
        __syncthreads();

#pragma unroll
        for (int st = 0; st < 12; ++st) {
            const int cs = st >> 2, t = st & 3;
            if (cs < nch) {
                const int s = (kc0 + cs) * 4 + t;
                const int kd = t >> 1, kh = t & 1;
                const int toff = cs * set_stride + (kd * p.HT + kh) * p.BW * 16;
                const int cur = st % 3;
                fetch(s + 2, wq[(cur + 2) % 3]);         // pinned here: two taps of L2 latency ahead of their use
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    half8 a[NPL];
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl)
                        a[hl] = *reinterpret_cast<const half8*>(lds + a_off[u] + hl * p.plane_stride + toff);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const half8 bhi = __builtin_bit_cast(half8, wq[cur][u < 2 ? 0 : 1][nb * NPL]);
                        if constexpr (NPASS == 3) {
                            const half8 blo = __builtin_bit_cast(half8, wq[cur][u < 2 ? 0 : 1][nb * NPL + 1]);
                            acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bhi, acc[u][nb], 0, 0, 0);
                            acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], blo, acc[u][nb], 0, 0, 0);
                        }
                        acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bhi, acc[u][nb], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ================= epilogue: per (pz,py) and column block the three x classes meet in LDS =================
    float* m = reinterpret_cast<float*>(lds);                      // [3 x positions][64 accumulator rows][MLD]
    const int col = tid & 31, rg = tid >> 5;
#pragma unroll 1
    for (int pp = 0; pp < 4; ++pp) {
        const int pz = pp >> 1, py = pp & 1;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            __syncthreads();                                       // A planes (or the previous round) fully consumed
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int cls = u < 2 ? clsA : clsB, mb = u < 2 ? u : mbB;
                if (cls / 3 == pp) {                               // wave-uniform
                    const int xp = cls - pp * 3;
                    float* mw = m + (xp * 64 + mb * 32 + khalf * 4) * MLD + l32;
#pragma unroll
                    for (int i = 0; i < 16; ++i)                    // accumulator row order; the reader undoes row_perm
                        mw[((i >> 2) * 8 + (i & 3)) * MLD] = acc[u][nb][i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 64 / NRG; ++it) {
                const int q = rg + NRG * it;
                int bd, bh, bw;
                box_coords(p, q, bd, bh, bw);
                const int gz = z0 + bd, gy = y0 + bh, gx = x0 + bw;
                if (gz >= p.d || gy >= p.h || gx >= p.w) continue;
                const int qr = (q & ~31) + row_unperm(q & 31);     // accumulator row holding voxel q
                const float* mr = m + qr * MLD + col;
                const float mm = mr[0 * 64 * MLD], m0 = mr[1 * 64 * MLD], mp = mr[2 * 64 * MLD];
                float* o = p.out + ((((int64_t)(2 * gz + pz) * (2 * p.h) + (2 * gy + py)) * (2 * p.w) + 2 * gx) * p.Cout +
                                    nt * 64 + nb * 32 + col);
                o[0] = (m0 + mm) * dq;
                o[p.Cout] = (m0 + mp) * dq;
            }
        }
    }
}

// packed[ntile64][class 12][kc][(kd,kh) 4][nb 2][hl][lane] (uint4 = 8 halfs): lane l holds
// B[k = 8*(l>>5)+j][n = l&31] = U_class[co = ntile*64 + nb*32 + (l&31)][ci = CA + kc*16 + 8*(l>>5) + j][kd'][kh'] * 2^wexp:
// along z / y the folded 2-tap weights of parity p (p = 0: (w[0], w[1] + w[2]), p = 1: (w[0] + w[1], w[2])), along x
// (w[0], w[0] + w[1] + w[2], w[2]) for x position (-, 0, +).  Sums in double: the order of the adds does not matter.
__global__ void pack_updiff(const float* __restrict__ w, int Cin, int CA, int CB, int Cout, int wexp, int npl,
                            uint4* __restrict__ out) {
    const int KCB = CB / KC;
    const int nf = 2 * npl;
    const int64_t n = (int64_t)(Cout / 64) * NCLS * KCB * 4 * nf * 64;
    const double s = ldexp(1.0, wexp);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int f = (int)(r % nf); r /= nf;
        const int t = (int)(r & 3); r >>= 2;
        const int kc = (int)(r % KCB); r /= KCB;
        const int cls = (int)(r % NCLS); r /= NCLS;
        const int ntile = (int)r;
        const int nb = f / npl, hl = f - nb * npl;
        const int co = ntile * 64 + nb * 32 + (lane & 31);
        const int ci0 = CA + kc * KC + 8 * (lane >> 5);
        const int pp = cls / 3, xp = cls - pp * 3, pz = pp >> 1, py = pp & 1;
        const int kd = t >> 1, kh = t & 1;
        // original taps folded into (kd', kh', x position)
        const int zlo = pz == 0 ? (kd == 0 ? 0 : 1) : (kd == 0 ? 0 : 2), zhi = pz == 0 ? (kd == 0 ? 0 : 2) : (kd == 0 ? 1 : 2);
        const int ylo = py == 0 ? (kh == 0 ? 0 : 1) : (kh == 0 ? 0 : 2), yhi = py == 0 ? (kh == 0 ? 0 : 2) : (kh == 0 ? 1 : 2);
        const int xlo = xp == 0 ? 0 : (xp == 1 ? 0 : 2), xhi = xp == 0 ? 0 : (xp == 1 ? 2 : 2);
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* g = w + ((size_t)co * Cin + ci0 + j) * 27;
            double u = 0.0;
            for (int a = zlo; a <= zhi; ++a)
                for (int b = ylo; b <= yhi; ++b)
                    for (int c = xlo; c <= xhi; ++c) u += (double)g[(a * 3 + b) * 3 + c];
            const float x = (float)(u * s);
            const _Float16 hh = (_Float16)x;
            v[j] = hl == 0 ? hh : (_Float16)(x - (float)hh);
        }
        out[i] = __builtin_bit_cast(uint4, v);
    }
}

int ilog2i(int v) { int r = 0; while ((1 << r) < v) ++r; return r; }

// low-res boxes of exactly 64 voxels (two 32-row blocks per class), power-of-two sides; the cheapest cover wins
bool choose_box(int d, int h, int w, int npl, int& BD, int& BH, int& BW) {
    static const int opts[][3] = {{4, 4, 4}, {2, 4, 8}, {4, 2, 8}, {2, 2, 16}, {8, 4, 2}, {4, 8, 2}, {8, 2, 4}, {2, 8, 4},
                                  {1, 4, 16}, {4, 1, 16}, {1, 8, 8}, {8, 1, 8}, {8, 8, 1}};
    int64_t best = -1;
    for (auto& o : opts) {
        const int64_t npos = (int64_t)(o[0] + 2) * (o[1] + 2) * o[2];
        if (npos * 4 > 2 * NTHR) continue;
        const int64_t plane = ((npos * 16 + 255) / 256) * 256 + 16;
        if (3 * 6 * npl * plane > 96 * 1024) continue;         // three chunks' planes
        int64_t cost = (int64_t)bfm_cdiv(d, o[0]) * bfm_cdiv(h, o[1]) * bfm_cdiv(w, o[2]);
        cost = cost * 64 - o[2];
        if (best < 0 || cost < best) { best = cost; BD = o[0]; BH = o[1]; BW = o[2]; }
    }
    return best >= 0;
}

}  // namespace

extern "C" size_t bfm_pack_conv_weights_updiff_bytes(int CB, int Cout, int passes) {
    if (CB <= 0 || Cout <= 0 || CB % KC || Cout % 64) return 0;
    const int npl = passes == 3 ? 2 : 1;
    return (size_t)(Cout / 64) * NCLS * (CB / KC) * 4 * 2 * npl * 64 * sizeof(uint4);
}

extern "C" int bfm_pack_conv_weights_updiff(const float* w_oidhw, int CA, int CB, int Cout, float wmax_abs_host, int passes,
                                            void* wpacked, int* wexp_host, bfm_stream_t stream) {
    if (!w_oidhw || !wpacked || !wexp_host || CA < 0 || CB <= 0 || Cout <= 0) return BFM_E_ARG;
    if (CB % KC || Cout % 64 || (passes != 1 && passes != 3)) return BFM_E_SHAPE;
    int wexp = 0;
    if (wmax_abs_host > 0.f && wmax_abs_host < INFINITY) {
        int ex;
        (void)frexpf(12.f * wmax_abs_host, &ex);              // a folded weight sums up to 2 x 2 x 3 taps
        wexp = 14 - ex;
        wexp = wexp > 60 ? 60 : (wexp < -60 ? -60 : wexp);
    }
    *wexp_host = wexp;
    const int npl = passes == 3 ? 2 : 1;
    const int64_t n = (int64_t)(Cout / 64) * NCLS * (CB / KC) * 4 * 2 * npl * 64;
    const int nb = (int)std::min<int64_t>(8192, bfm_cdiv64(n, 256));
    hipLaunchKernelGGL(pack_updiff, dim3(nb), dim3(256), 0, bfm_s(stream), w_oidhw, CA + CB, CA, CB, Cout, wexp, npl,
                       static_cast<uint4*>(wpacked));
    return bfm_launch_status();
}

// 1 when bfm_conv3x3x3_updiff can run this shape (a 64-voxel box fits, 32-bit offsets), else 0
extern "C" int bfm_conv3x3x3_updiff_ok(int CB, int d, int h, int w, int Cout, int passes) {
    int BD, BH, BW;
    if (CB <= 0 || d <= 0 || h <= 0 || w <= 0 || Cout <= 0 || CB % KC || Cout % 64) return 0;
    if ((int64_t)d * h * w * CB > 0x7fffffffLL) return 0;
    return choose_box(d, h, w, passes == 3 ? 2 : 1, BD, BH, BW) ? 1 : 0;
}

extern "C" int bfm_conv3x3x3_updiff(const float* B, int CB, int d, int h, int w, const float* scale_b, const float* shift_b,
                                    const float* bound, int G, const void* wpacked, int wexp, int Cout, int passes,
                                    float* out, bfm_stream_t stream) {
    if (!B || CB <= 0 || d <= 0 || h <= 0 || w <= 0 || !scale_b || !shift_b || !bound || G <= 0 || !wpacked || !out)
        return BFM_E_ARG;
    if (CB % KC || Cout % 64 || Cout <= 0) return BFM_E_SHAPE;
    if (passes != 1 && passes != 3) return BFM_E_ARG;
    if ((reinterpret_cast<uintptr_t>(B) & 15) || (reinterpret_cast<uintptr_t>(scale_b) & 15) ||
        (reinterpret_cast<uintptr_t>(shift_b) & 15) || (reinterpret_cast<uintptr_t>(wpacked) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return BFM_E_ARG;
    if ((int64_t)d * h * w * CB > 0x7fffffffLL) return BFM_E_SHAPE;       // 32-bit staging offsets
    const int npl = passes == 3 ? 2 : 1;
    UdParams p{};
    p.B = B; p.CB = CB; p.d = d; p.h = h; p.w = w;
    p.scale = scale_b; p.shift = shift_b; p.bound = bound; p.G = G;
    p.wp = static_cast<const uint4*>(wpacked);
    p.wexp = wexp; p.Cout = Cout; p.out = out;
    if (!choose_box(d, h, w, npl, p.BD, p.BH, p.BW)) return BFM_E_SHAPE;
    p.HT = p.BH + 2;
    p.bw_shift = ilog2i(p.BW); p.bhw_shift = ilog2i(p.BH * p.BW);
    const int nTz = bfm_cdiv(d, p.BD);
    p.nTy = bfm_cdiv(h, p.BH); p.nTx = bfm_cdiv(w, p.BW);
    p.nMt = nTz * p.nTy * p.nTx;
    p.NT = Cout / 64;
    p.KCB = CB / KC;
    p.npos_lds = (p.BD + 2) * p.HT * p.BW;
    p.plane_stride = ((p.npos_lds * 16 + 255) / 256) * 256 + 16;
    size_t smem = (size_t)3 * 6 * npl * p.plane_stride;           // three K chunks per stage
    const size_t epi = (size_t)3 * 64 * MLD * sizeof(float);
    if (smem < epi) smem = epi;
    if (smem > 96 * 1024) return BFM_E_SHAPE;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_updiff<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_updiff<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_done = true;
    }
    if ((int64_t)p.nMt * p.NT > 0x7fffffff) return BFM_E_SHAPE;
    dim3 grid((unsigned)(p.nMt * p.NT));
    if (passes == 3) hipLaunchKernelGGL(conv_updiff<3>, grid, dim3(NTHR), smem, bfm_s(stream), p);
    else hipLaunchKernelGGL(conv_updiff<1>, grid, dim3(NTHR), smem, bfm_s(stream), p);
    return bfm_launch_status();
}
