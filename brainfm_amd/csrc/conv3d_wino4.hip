// 3x3x3 convolution with a Winograd F(4,3) transform along x: 2x fewer matrix-core FLOPs along x than the direct
// form (6 products per 4 outputs instead of 12), 1.33x fewer than conv3d_wino.hip's F(2,3).
//
//      d = in[x0-1 .. x0+4]                 V = B^T d   (6 values: rows of B^T below)
//      g = w[.., kw=0..2]                   U = G g     (6 values, packed offline in float64)
//      m_p = sum over (kd,kh,ci) of V_p * U_p           y = A^T m   (4 outputs x0 .. x0+3)
//
//      B^T = [4  0 -5  0 1 0]    G = [ 1/4    0     0  ]    A^T = [1 1  1 1  1 0]
//            [0 -4 -4  1 1 0]        [-1/6  -1/6  -1/6 ]          [0 1 -1 2 -2 0]
//            [0  4 -4 -1 1 0]        [-1/6   1/6  -1/6 ]          [0 1  1 4  4 0]
//            [0 -2 -1  2 1 0]        [ 1/24  1/12  1/6 ]          [0 1 -1 8 -8 1]
//            [0  2 -1 -2 1 0]        [ 1/24 -1/12  1/6 ]
//            [0  4  0 -5 0 1]        [ 0     0     1   ]
//
// The GEMM is 6 "positions" x (M/4 output quads) x K = 9 (kd,kh) taps x Cin: 13.5 tap-rows per output voxel instead
// of 18 (F(2,3)) or 27 (direct), and 1.5 transformed values per voxel to split and store instead of 2.  The price is
// rounding: measured through the C ABI against a float64 convolution the kernel sits at 1-2.5e-6 of max|y| per layer,
// about twice F(2,3)'s 1e-6 (tests/test_gpu_infer.py::test_winograd_f43_kernel_vs_float64_convolution; the 7e-6 of the
// first NumPy model, scripts/micro/wino_f43_accuracy.py, was pessimistic); the products keep the split-fp16 three-pass
// scheme.
//
// Workgroup = 4 waves for one box of 256 output voxels (64 quads = two 32-row blocks per position) x 64 couts.  Six
// positions do not divide over four SIMDs as waves (a 6-wave workgroup sits 2,2,1,1 on the SIMDs and a second one does
// not fit beside it: profiles/r03_conv_wino4_experiment.txt), so the twelve (position, row block) units go three to a
// wave: wave w owns position w entirely (both row blocks) and one row block of position 4 + (w >> 1) -- equal matrix work
// per wave, 96 accumulator registers, two positions' weights streamed L2 -> VGPR one tap ahead.  The transformed, split
// halo'd box lives in LDS ([pos][k-half][hi|lo][row][quad][8 ch], 55 KB); epilogue: the six m_p meet in LDS, one thread
// per (quad, cout) forms y0..y3, dequantises, optionally adds what `out` holds (accumulate mode), applies LeakyReLU and
// stores.  Single-source inputs only, no split-K; the dense form and the masked form over the boxes the tile mask keeps
// (conv_wino4_masked, the tile loop's last convolution); the uniform-box pair stays with F(2,3) (engine._needs_f23).
#include "bfm_common.h"
#include "wino_shared.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));

constexpr int KC = 16;
constexpr int NPOS = 6;
constexpr int NTHR = 256;          // four waves: one per SIMD, twice per CU
constexpr int NRG = NTHR / 32;     // row groups of the epilogue (8)
constexpr int MLD = 33;            // epilogue LDS row stride in floats (odd: conflict-free)

struct W4Params {
    const float* A;
    int CA, D, H, W;
    const float *scale, *shift, *bound;
    int G;
    const uint4* wp;
    int wexp, Cout;
    float slope;
    float* out;
    int accum;
    int TD, TH, TW, HT, QW;          // box, halo'd rows per slice, quads per row
    int qw_shift, thq_shift;         // log2(QW), log2(TH*QW)
    int nTy, nTx, nMt, NT, KCN;
    int npos_lds, plane_stride;      // (TD+2)*HT*QW positions; bytes per plane
    double *rsum, *rsq;              // optional output-moment rows [nMt][Cout] (see conv3d_mfma.hip)
    float *rmn, *rmx;
    const int* list;                 // sparse forms: the boxes to compute, ascending
    const int* list_n;               // ... their number, on the device
    const unsigned char* uni_flags;  // conv_wino4d_rest / _uniform: [nMt] 0, or 1 + the class of a box whose operands equal its class mates'
    float* uni_acc;                  // [27][NT][2][8][NTHR] float4: the class representatives' output-transformed sums
    // batch (dense form only): nMt = S * nMtS boxes, box mt belongs to sample mt / nMtS; A, out advance by sA, sO elements
    // per sample, scale / shift by saff, bound by G; moment rows are [S * nMtS][Cout].  nMtS == 0: one sample.  A sample's
    // workgroups do exactly what they do in a launch of that sample alone.
    int nMtS, saff;
    int64_t sA, sO;
    int dbg_sleep;                   // diagnostics builds: the first round's odd wave slots start this many kilocycles late
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

__device__ __forceinline__ void quad_coords(const W4Params& p, int q, int& d, int& h, int& j) {
    d = q >> p.thq_shift;
    const int rem = q & ((1 << p.thq_shift) - 1);
    h = rem >> p.qw_shift;
    j = rem & ((1 << p.qw_shift) - 1);
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

// MODE 0: every box in full.  1 (conv_wino4_masked): the boxes of p.list (those that hold input).
// There is no uniform-box pair of this kernel, on purpose: conv_wino's pair rests on class mates multiplying the same
// operands, which needs a layer's NUMERICAL support to be its mathematical one.  F(2,3) has that property (y0 = m0 + m1 + m2
// touches d0, d1, d2 only); F(4,3) does not -- y0 = m0 + .. + m4 rounds differently when d3, d4 change although they
// cancel exactly -- so a box at the edge of the constant background would need a reach of 4 voxels along x per layer
// instead of 1.  Built and measured (tests/diag/diag_wino4_uniform*.py: the pair is exact kernel by kernel, the network
// is not); the layers that can take that shortcut stay with conv_wino (engine._needs_f23).
// ABL (diagnostics builds only, -DBFM_W4_ABLATE, tests/diag/diag_wino4_ablate.py): phases compiled out to see what the
// launch time is made of (wrong results).  1: no weight loads in the tap loop, 2: no LDS operand reads in the tap loop,
// 4: no staging (barriers kept), 16: no epilogue, 64: no MFMAs.
template <int NPASS, int MODE, int ABL = 0>
__device__ __forceinline__ void conv_wino4_body(const W4Params& pin) {
    W4Params p = pin;
    constexpr bool LIST = MODE != 0;
    constexpr int NPL = (NPASS == 3) ? 2 : 1;
    constexpr int NF = 2 * NPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave 0..3
    const int posA = wv;                                           // position owned entirely (units 0, 1 = row blocks 0, 1)
    const int posB = 4 + (wv >> 1), mbB = wv & 1;                  // unit 2: row block mbB of position posB
    const int l32 = lane & 31, khalf = lane >> 5;
    int item;
    {   // one workgroup per (box, cout tile), XCD-aware bijective remap (neighbouring boxes share halo lines in one L2);
        // the sparse forms take their boxes from a device-built list: a workgroup beyond it reads one scalar and ends
        const int nblk = LIST ? p.list_n[0] * p.NT : p.nMt * p.NT;
        const int bid = blockIdx.x;
        if (LIST && bid >= nblk) return;                     // workgroup-uniform, before any barrier
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = LIST ? p.list[item / p.NT] : item / p.NT;
    const int nt = item % p.NT;
    int mtl = mt;                                                  // box inside its sample
    if (!LIST && p.nMtS > 0) {
        const int smp = mt / p.nMtS;
        mtl = mt - smp * p.nMtS;
        p.A += smp * p.sA;
        p.out += smp * p.sO;
        p.scale += smp * p.saff;
        p.shift += smp * p.saff;
        p.bound += smp * p.G;
    }
    const int tx = mtl % p.nTx;
    const int ty = (mtl / p.nTx) % p.nTy;
    const int tz = mtl / (p.nTx * p.nTy);
    const int z0 = tz * p.TD, y0 = ty * p.TH, x0 = tx * p.TW;

    float bmax = 0.f;
    for (int g = 0; g < p.G; ++g) bmax = fmaxf(bmax, p.bound[g]);
    int aexp = 0;
    if (bmax > 0.f && bmax < INFINITY) {
        int ex;
        (void)frexpf(bmax, &ex);
        aexp = 10 - ex;                                            // |V| <= 10 * bound < 16 * 2^ex
        aexp = aexp > 60 ? 60 : (aexp < -60 ? -60 : aexp);
    }
    const float a_scale = ldexpf(1.0f, aexp);
    const float dq = ldexpf(1.0f, -(aexp + p.wexp));

    // A base offsets of the three units: quad (d,h,j), tap (kd,kh)=(0,0) reads halo row (d, h), quad j
    int a_off[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int pos = u < 2 ? posA : posB, mb = u < 2 ? u : mbB;
        int d, h, j;
        quad_coords(p, mb * 32 + row_perm(l32), d, h, j);
        a_off[u] = ((pos * 2 + khalf) * NPL) * p.plane_stride + ((d * p.HT + h) * p.QW + j) * 16;
    }

    // staging items: e = tid + it*NTHR -> (halo row, quad, channel quad); off0 = element offset of voxel
    // (gz, gy, x0 + 4j - 1) channel 0 (may point outside the row: the mask says which of the 6 x positions exist)
    constexpr int MAX_IT = 3;
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
            const int j = ps & ((1 << p.qw_shift) - 1);
            const int r = ps >> p.qw_shift;
            const int hz = r / p.HT, hy = r - hz * p.HT;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + 4 * j - 1;
            int m = 0;
            if (gz >= 0 && gz < p.D && gy >= 0 && gy < p.H) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (gx + i >= 0 && gx + i < p.W) m |= 1 << i;
            }
            msk[it] = m;
            off0[it] = ((gz * p.H + gy) * p.W + gx) * p.CA;
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

    // this wave's two weight streams (positions posA, posB): S = KCN*9 steps (chunk-major, (kd,kh)-minor), NF fragments of
    // 64 x uint4 per step and position; two register sets, one step ahead
    const int S = p.KCN * 9;
    const uint4* wbA = p.wp + (size_t)(nt * NPOS + posA) * S * (NF * 64) + lane;
    const uint4* wbB = p.wp + (size_t)(nt * NPOS + posB) * S * (NF * 64) + lane;
    uint4 wq[2][2][NF];
    auto fetch = [&](int s, uint4 (&dst)[2][NF]) __attribute__((always_inline)) {
        if constexpr ((ABL & 1) != 0) { if (s > 1) return; }
        const int sc = s < S ? s : S - 1;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            dst[0][f] = wbA[(size_t)sc * (NF * 64) + f * 64];
            dst[1][f] = wbB[(size_t)sc * (NF * 64) + f * 64];
        }
    };
    fetch(0, wq[0]);
    if constexpr ((ABL & 1) != 0) fetch(1, wq[1]);
    half8 a_abl[3][NPL];
    if constexpr ((ABL & 2) != 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int hl = 0; hl < NPL; ++hl)
#pragma unroll
                for (int j = 0; j < 8; ++j) a_abl[u][hl][j] = (_Float16)(0.01f * (float)((lane * 7 + u * 3 + hl * 5 + j) % 61) - 0.3f);
    }

    auto do_chunk = [&](int kc, auto par_tag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;      // parity of the chunk's first weight step (9 steps per chunk)
        const int c0 = kc * KC;
        const float* src = p.A + c0 + q4 * 4;
        const float4 sc4 = *reinterpret_cast<const float4*>(p.scale + c0 + q4 * 4);
        const float4 sh4 = *reinterpret_cast<const float4*>(p.shift + c0 + q4 * 4);
        const float sc[4] = {sc4.x * a_scale, sc4.y * a_scale, sc4.z * a_scale, sc4.w * a_scale};
        const float sh[4] = {sh4.x * a_scale, sh4.y * a_scale, sh4.z * a_scale, sh4.w * a_scale};
        __syncthreads();                                 // previous chunk's readers are done
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
            if (msk[it] < 0) continue;
            if constexpr ((ABL & 4) != 0) continue;
            float4 v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (msk[it] & (1 << i)) v[i] = *reinterpret_cast<const float4*>(src + off0[it] + i * p.CA);
            }
            float dd[6][4];                              // [x position][channel]: affine, zero padding after it
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const bool ok = msk[it] & (1 << i);
                const float y[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) dd[i][c] = ok ? fmaf(y[c], sc[c], sh[c]) : 0.f;
            }
            const int e = tid + it * NTHR;
            unsigned char* dst = lds + st_plane + (e >> 2) * 16;
            float t[6][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d0 = dd[0][c], d1 = dd[1][c], d2 = dd[2][c], d3 = dd[3][c], d4 = dd[4][c], d5 = dd[5][c];
                const float a = fmaf(-4.f, d2, d4);      // d4 - 4 d2
                const float b = fmaf(-4.f, d1, d3);      // d3 - 4 d1
                const float cc = d4 - d2;
                const float ee = 2.f * (d3 - d1);
                t[0][c] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                t[1][c] = a + b;
                t[2][c] = a - b;
                t[3][c] = cc + ee;
                t[4][c] = cc - ee;
                t[5][c] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
            }
#pragma unroll
            for (int ps = 0; ps < NPOS; ++ps)
                split_store4<NPASS == 3>(t[ps], dst + (ps * 2 * NPL) * p.plane_stride, p.plane_stride);
            __builtin_amdgcn_sched_barrier(0);           // keep the next item's loads from being hoisted (registers)
        }
        __syncthreads();

#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int s = kc * 9 + t;
            const int kd = t / 3, kh = t - kd * 3;
            const int toff = (kd * p.HT + kh) * p.QW * 16;
            const int cur = (PAR + t) & 1;
            fetch(s + 1, wq[cur ^ 1]);                   // pinned here: one tap of L2 latency ahead of its use
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                half8 a[NPL];
#pragma unroll
                for (int hl = 0; hl < NPL; ++hl) {
                    if constexpr ((ABL & 2) != 0) a[hl] = a_abl[u][hl];
                    else a[hl] = *reinterpret_cast<const half8*>(lds + a_off[u] + hl * p.plane_stride + toff);
                }
                if constexpr ((ABL & 64) != 0) {
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl) acc[u][0][hl] += (float)a[hl][0] + (float)a[hl][7];
                    acc[u][1][0] += __builtin_bit_cast(float, wq[cur][u < 2 ? 0 : 1][0].x) + __builtin_bit_cast(float, wq[cur][u < 2 ? 0 : 1][NF - 1].w);
                    continue;
                }
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
    };
    for (int kc = 0; kc < p.KCN; kc += 2) {              // 9 steps per chunk: the parity flips every chunk
        do_chunk(kc, std::integral_constant<int, 0>{});
        if (kc + 1 < p.KCN) do_chunk(kc + 1, std::integral_constant<int, 1>{});
    }

    if constexpr ((ABL & 16) != 0) {
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) sm += acc[u][nb][i];
        if (sm == 12345.678f) p.out[tid] = sm;
        return;
    }
    // ================= epilogue: output transform through LDS =================
    float* m = reinterpret_cast<float*>(lds);                      // [6 positions][64 accumulator rows][MLD]
    const int col = tid & 31, rg = tid >> 5;
    const bool interior = z0 + p.TD <= p.D && y0 + p.TH <= p.H && x0 + p.TW <= p.W;      // wave-uniform
    unsigned off_t = 0;
    int un[4] = {0, 0, 0, 0};
    if (interior) {
        int d_t, h_t, j_t;
        quad_coords(p, rg, d_t, h_t, j_t);
        off_t = (unsigned)(((d_t * p.H + h_t) * p.W + 4 * j_t) * p.Cout + col);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) un[s2] = row_unperm(rg + 8 * s2);
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        __syncthreads();                                           // A planes (or the previous round) fully consumed
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int pos = u < 2 ? posA : posB, mb = u < 2 ? u : mbB;
            float* mw = m + (pos * 64 + mb * 32 + khalf * 4) * MLD + l32;
#pragma unroll
            for (int i = 0; i < 16; ++i)                            // accumulator row order; the reader undoes row_perm
                mw[((i >> 2) * 8 + (i & 3)) * MLD] = acc[u][nb][i];
        }
        __syncthreads();
        float fs = 0.f, fq = 0.f, fmn = INFINITY, fmx = -INFINITY;   // this thread's column, its 8 quads (<= 32 values)
        if (interior) {
            // Interior boxes (all of a 160^3 tile but its last slabs): the thread's quads are q = rg + 8 it, and the bit
            // fields (d, h, j) of q split into a per-thread part (the three bits of rg, computed once: off_t) and a
            // wave-uniform part (8 it, on the scalar unit) -- no index arithmetic, no bounds tests per value
            float* ob = p.out + nt * 64 + nb * 32;
            float prev[8][4];
            if (p.accum) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    int d_u, h_u, j_u;
                    quad_coords(p, 8 * it, d_u, h_u, j_u);
                    const float* o = ob + (((int64_t)(z0 + d_u) * p.H + (y0 + h_u)) * p.W + x0 + 4 * j_u) * p.Cout;
#pragma unroll
                    for (int k = 0; k < 4; ++k) prev[it][k] = o[off_t + (unsigned)(k * p.Cout)];
                }
            }
            const float* mc = m + col;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                int d_u, h_u, j_u;
                quad_coords(p, 8 * it, d_u, h_u, j_u);
                float* o = ob + (((int64_t)(z0 + d_u) * p.H + (y0 + h_u)) * p.W + x0 + 4 * j_u) * p.Cout;
                const int qr = 32 * (it >> 2) + un[it & 3];              // accumulator row holding quad rg + 8 it
                const float* mr = mc + qr * MLD;
                const float m0 = mr[0 * 64 * MLD], m1 = mr[1 * 64 * MLD], m2 = mr[2 * 64 * MLD];
                const float m3 = mr[3 * 64 * MLD], m4 = mr[4 * 64 * MLD], m5 = mr[5 * 64 * MLD];
                const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                float y[4];
                y[0] = ((m0 + s1) + s2) * dq;
                y[1] = fmaf(2.f, d2, d1) * dq;
                y[2] = fmaf(4.f, s2, s1) * dq;
                y[3] = (fmaf(8.f, d2, d1) + m5) * dq;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float r = y[k];
                    if (p.accum) r = r + prev[it][k];
                    r = r >= 0.f ? r : r * p.slope;
                    o[off_t + (unsigned)(k * p.Cout)] = r;
                    fs += r; fq = fmaf(r, r, fq); fmn = fminf(fmn, r); fmx = fmaxf(fmx, r);
                }
            }
        } else
#pragma unroll 2
        for (int q = rg; q < 64; q += NRG) {
            int d, h, j;
            quad_coords(p, q, d, h, j);
            const int gz = z0 + d, gy = y0 + h, gx = x0 + 4 * j;
            if (gz >= p.D || gy >= p.H || gx >= p.W) continue;
            const int qr = (q & ~31) + row_unperm(q & 31);           // accumulator row holding quad q
            const float* mr = m + qr * MLD + col;
            const float m0 = mr[0 * 64 * MLD], m1 = mr[1 * 64 * MLD], m2 = mr[2 * 64 * MLD];
            const float m3 = mr[3 * 64 * MLD], m4 = mr[4 * 64 * MLD], m5 = mr[5 * 64 * MLD];
            const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            float y[4];
            y[0] = ((m0 + s1) + s2) * dq;
            y[1] = fmaf(2.f, d2, d1) * dq;
            y[2] = fmaf(4.f, s2, s1) * dq;
            y[3] = (fmaf(8.f, d2, d1) + m5) * dq;
            float* o = p.out + (((int64_t)gz * p.H + gy) * p.W + gx) * p.Cout + nt * 64 + nb * 32 + col;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (gx + k < p.W) {
                    float r = y[k];
                    if (p.accum) r = r + o[(size_t)k * p.Cout];
                    r = r >= 0.f ? r : r * p.slope;
                    o[(size_t)k * p.Cout] = r;
                    fs += r; fq = fmaf(r, r, fq); fmn = fminf(fmn, r); fmx = fmaxf(fmx, r);
                }
            }
        }
        if (p.rsum != nullptr) {
            // moment row of this box: fold the 8 row groups of every column in fixed order (scratch behind m)
            double* ls = reinterpret_cast<double*>(lds + (size_t)NPOS * 64 * MLD * sizeof(float));   // [8][32]
            double* lq = ls + NRG * 32;
            float* lmn = reinterpret_cast<float*>(lq + NRG * 32);
            float* lmx = lmn + NRG * 32;
            ls[rg * 32 + col] = (double)fs; lq[rg * 32 + col] = (double)fq;
            lmn[rg * 32 + col] = fmn; lmx[rg * 32 + col] = fmx;
            __syncthreads();
            if (tid < 32) {
                double Ssum = 0.0, Q = 0.0;
                float MN = INFINITY, MX = -INFINITY;
#pragma unroll
                for (int r = 0; r < NRG; ++r) {
                    Ssum += ls[r * 32 + tid]; Q += lq[r * 32 + tid];
                    MN = fminf(MN, lmn[r * 32 + tid]); MX = fmaxf(MX, lmx[r * 32 + tid]);
                }
                const size_t o = (size_t)mt * p.Cout + nt * 64 + nb * 32 + tid;
                p.rsum[o] = Ssum; p.rsq[o] = Q; p.rmn[o] = MN; p.rmx[o] = MX;
            }
        }
    }
}

template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino4(const W4Params p) { conv_wino4_body<NPASS, 0>(p); }
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino4_masked(const W4Params p) { conv_wino4_body<NPASS, 1>(p); }

// ---------------------------------------------------------------------------------------------------------------------
// conv_wino4d: the same arithmetic with the staging off the critical path (round 5).
//
// What the ablation of conv_wino4 showed (tests/diag/diag_wino4_ablate.py, profiles/r05_wino4_ablation.txt; 64->64 @160^3,
// 1.75 ms): without its MFMAs the launch still takes 1.34 ms, the MFMAs alone 0.85 ms, and the staging is 0.66 ms of the
// 1.75 although its instructions would fit in a tenth of that -- a chain of three dependent rounds of (6 global loads ->
// affine, transform, split -> LDS stores) per 16-channel chunk with one round in flight per thread: latency, paid four
// times per box, and two workgroups per CU hide only part of it.  Here the raw fp32 chunk of the NEXT 16 channels is
// brought into LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no instructions waiting on it) while the nine taps
// of the current chunk multiply, spread one or two wave instructions per tap so that each piece has two taps to land before
// the in-order vmcnt wait of a later weight fetch reaches it; the transform then reads LDS, not memory.
//
// To make room for the raw chunk beside the planes at two workgroups per CU the box is 8 x 8 x 4 voxels -- ONE quad per
// (z, y) row: 100 halo rows = 100 transformed positions per plane (the 4 x 4 x 16 box has 144: 30 % less to transform,
// split and store per box as well), planes 38 400 B + raw 600 voxels x 64 B = 38 400 B -> 76 800 B per workgroup.
// A 16-lane ds_read_b128 group reads two runs of eight h-neighbours at d and d + 4 (halo rows 40 positions apart: the runs
// land in complementary halves of the 64 banks), so MFMA row block mb holds d in {2mb, 2mb+1, 2mb+4, 2mb+5}.
// Every output voxel gets the bits of conv_wino4 (same products, same order); the moment rows are per box and therefore
// differ in their partition, not in what they sum.
constexpr int DB_TD = 8, DB_TH = 8, DB_HT = 10, DB_NROW = 100;
constexpr int DB_PLANE = DB_NROW * 16;             // bytes per (position, k-half, hi | lo) plane
constexpr int DB_RAWROW = 6 * 64;                  // one halo row of the raw chunk: 6 voxels x 16 channels fp32
constexpr int DB_RAW = DB_NROW * DB_RAWROW;        // 38 400
constexpr int DB_NDMA = DB_NROW / 2;               // wave instructions per chunk: two halo rows (12 voxels, 48 lanes) each
constexpr int DB_NDMA_WAVE = (DB_NDMA + 3) / 4;    // 13 per wave (the last one on waves 0 and 1 only)

// x = hi + lo in fp16 for four values: hi by truncation (v_cvt_pkrtz_f16_f32, two values per instruction; x - hi is then
// exact in fp32), lo = fp16(x - hi) by one v_fma_mixlo/hi_f16 per value (fp32 = fp16 * -1 + fp32, rounded to nearest) in
// place of convert-back, subtract and re-pack: 36 instead of 72 instructions per item
template <bool LO>
__device__ __forceinline__ void split_store4_mix(const float (&t)[4], unsigned char* dp, int plane_stride) {
    const fp16x2_t h01 = __builtin_amdgcn_cvt_pkrtz(t[0], t[1]);
    const fp16x2_t h23 = __builtin_amdgcn_cvt_pkrtz(t[2], t[3]);
    uint2 hv;
    hv.x = __builtin_bit_cast(unsigned, h01);
    hv.y = __builtin_bit_cast(unsigned, h23);
    *reinterpret_cast<uint2*>(dp) = hv;
    if constexpr (LO) {
        uint2 lv;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lv.x) : "v"(hv.x), "v"(t[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lv.x) : "v"(hv.x), "v"(t[1]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lv.y) : "v"(hv.y), "v"(t[2]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lv.y) : "v"(hv.y), "v"(t[3]));
        *reinterpret_cast<uint2*>(dp + plane_stride) = lv;
    }
}

// Work of a wave: position wv for the whole box (row blocks 0, 1 x cout halves 0, 1: 12 MFMAs per tap) and, of position
// 4 + (wv >> 1), cout half wv & 1 of both row blocks (6 MFMAs per tap): 96 accumulator registers and SIX weight fragments
// per tap (conv_wino4 splits positions 4 / 5 by row block and streams eight), which makes room for a three-deep ring:
// weights are requested TWO taps ahead of their use.
// ABL: diagnostics builds only (-DBFM_W4_ABLATE).  1: no weight loads in the tap loop, 2: no LDS operand reads, 4: no
// transform stage, 8: no LDS-DMA, 16: no epilogue, 64: no MFMAs.
// MODE 0: every box.  1 (conv_wino4d_masked): the boxes of p.list (those that hold input).  2 / 3 (conv_wino4d_rest /
// conv_wino4d_uniform, launched as a pair like conv_wino's): boxes flagged uniform by bfm_uniform_boxes -- the layer's input
// around the box is a function of the distances to the tile's faces alone -- share the output-transformed sums of the first
// flagged box of their class.  ONLY for layers whose output feeds no other layer that takes this shortcut (the skip halves
// of the last two decoders' first convs): a box is a whole number of quads, so what a box's sums depend on NUMERICALLY is its
// mathematical halo, as with F(2,3); a voxel of the OUTPUT, though, carries the rounding of its whole quad (y0 = m0 + .. + m4
// involves d3, d4), which a consumer that assumes a reach of one voxel per layer would not see (conv_wino4 header;
// engine._uniform_variant).
template <int NPASS, int MODE, int ABL = 0>
__device__ __forceinline__ void conv_wino4d_body(const W4Params& pin) {
    W4Params p = pin;
    constexpr bool LIST = MODE != 0;
    constexpr bool UNI = MODE == 3;                                // stream a class's sums to its mates: no staging, no products
    constexpr bool BY_FLAG = MODE == 2 || MODE == 3;
    constexpr int NPL = (NPASS == 3) ? 2 : 1;
    constexpr int NF = 2 * NPL;
    constexpr int NWF = 3 * NPL;                                   // weight fragments per tap and wave
    constexpr int RAW_OFF = 2 * NPOS * NPL * DB_PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int posA = wv;
    const int posB = 4 + (wv >> 1), nbB = wv & 1;
    const int l32 = lane & 31, khalf = lane >> 5;
#ifdef BFM_W4_ABLATE
    if (p.dbg_sleep > 0 && blockIdx.x < 512) {                     // de-phase the two workgroups of a CU (experiment)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (hwid & 1)
            for (int i = 0; i < p.dbg_sleep; ++i) __builtin_amdgcn_s_sleep(16);
    }
#endif
    int item;
    {
        const int nblk = LIST ? p.list_n[0] * p.NT : p.nMt * p.NT;
        const int bid = blockIdx.x;
        if (LIST && bid >= nblk) return;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = LIST ? p.list[item / p.NT] : item / p.NT;
    const int nt = item % p.NT;
    int mtl = mt;
    if (!LIST && p.nMtS > 0) {
        const int smp = mt / p.nMtS;
        mtl = mt - smp * p.nMtS;
        p.A += smp * p.sA;
        p.out += smp * p.sO;
        p.scale += smp * p.saff;
        p.shift += smp * p.saff;
        p.bound += smp * p.G;
    }
    const int tx = mtl % p.nTx;
    const int ty = (mtl / p.nTx) % p.nTy;
    const int tz = mtl / (p.nTx * p.nTy);
    const int z0 = tz * DB_TD, y0 = ty * DB_TH, x0 = tx * 4;
    int cls = 0;                                                   // 1 + class of a flagged box
    if constexpr (BY_FLAG) cls = __builtin_amdgcn_readfirstlane((int)p.uni_flags[mt]);
    const bool rep = MODE == 2 && cls != 0;                        // conv_wino4d_rest's flagged boxes: the class representatives
    float4* const ybuf = BY_FLAG && cls ? reinterpret_cast<float4*>(p.uni_acc) + ((size_t)(cls - 1) * p.NT + nt) * (2 * 8 * NTHR) + tid
                                        : nullptr;

    float bmax = 0.f;
    for (int g = 0; g < p.G; ++g) bmax = fmaxf(bmax, p.bound[g]);
    int aexp = 0;
    if (bmax > 0.f && bmax < INFINITY) {
        int ex;
        (void)frexpf(bmax, &ex);
        aexp = 10 - ex;
        aexp = aexp > 60 ? 60 : (aexp < -60 ? -60 : aexp);
    }
    const float a_scale = ldexpf(1.0f, aexp);
    const float dq = ldexpf(1.0f, -(aexp + p.wexp));

    floatx16 acc[2][2];                                            // position posA: [row block][cout half]
    floatx16 accB[2];                                              // position posB, cout half nbB: [row block]
    if constexpr (!UNI) {
    // A operand offsets: MFMA row l32 of row block mb is the quad (d, h) = (2 mb + g + 4 (k >> 3), k & 7), where
    // g * 16 + k = row_perm(l32) numbers the lanes of the two 16-lane ds_read_b128 groups
    int a_off[4];                                                  // (posA, mb 0), (posA, mb 1), (posB, mb 0), (posB, mb 1)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int pos = u < 2 ? posA : posB, mb = u & 1;
        const int pr = row_perm(l32);
        const int d = mb * 2 + (pr >> 4) + 4 * ((pr & 15) >> 3), h = pr & 7;
        a_off[u] = ((pos * 2 + khalf) * NPL) * DB_PLANE + (d * DB_HT + h) * 16;
    }

    // transform items: (halo row r = (tid >> 2) + 64 it, channel quad q4); bit i of msk: voxel x0 - 1 + i of the row exists
    const int q4 = tid & 3;
    int msk[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = (tid >> 2) + 64 * it;
        msk[it] = -1;
        if (r < DB_NROW) {
            const int hz = r / DB_HT, hy = r - hz * DB_HT;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1;
            int m = 0;
            if (gz >= 0 && gz < p.D && gy >= 0 && gy < p.H) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (x0 - 1 + i >= 0 && x0 - 1 + i < p.W) m |= 1 << i;
            }
            msk[it] = m;
        }
    }
    // the halo'd box lies inside the volume (77 % of the boxes of a 160^3 tile): no padding to apply.  Workgroup-uniform
    const bool inner = z0 >= 1 && z0 + DB_TD < p.D && y0 >= 1 && y0 + DB_TH < p.H && x0 >= 1 && x0 + 4 < p.W;
    const int st_plane = ((q4 >> 1) * NPL) * DB_PLANE + (q4 & 1) * 8;

    // LDS-DMA: wave instruction i brings halo rows 2i and 2i + 1 (lanes 0..47: slot = lane >> 2 = 6 (row & 1) + x, 16 bytes
    // = 4 channels each); voxels outside the volume are read from a clamped address and masked at transform time
    const int slot = lane >> 2;
    const int sb = slot >= 6 ? 1 : 0;
    int gxc = x0 + (slot - 6 * sb) - 1;
    gxc = gxc < 0 ? 0 : (gxc > p.W - 1 ? p.W - 1 : gxc);
    const int lane_el = gxc * p.CA + (lane & 3) * 4;
    const bool dma_lane = slot < 12;
    // the whole raw chunk kc: 13 (12) asynchronous wave instructions.  Inline assembly on purpose: the compiler's wait
    // insertion takes an LDS-DMA it knows about for a store that any later ds_read may alias and drains vmcnt to 0 in front
    // of the tap loop's operand reads (seen in the ISA of the builtin form); unknown to it, the pieces are simply the oldest
    // entries of the in-order vmcnt queue -- older than every weight fetch issued after them, so the fetches' counted waits
    // are still sufficient for the fetches and reach the pieces three taps after their issue (see do_chunk)
    const unsigned raw_lds = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) unsigned char*)(lds + RAW_OFF));
    const int zs = z0 > 0 ? z0 - 1 : 0;                        // first slice of the halo'd box inside the volume
    const float* slab = p.A + (int64_t)zs * p.H * p.W * p.CA;  // 32-bit byte offsets from here (host: 10 slices < 4 GB)
    auto issue_dma = [&](int kc) __attribute__((always_inline)) {
        if constexpr ((ABL & 8) != 0) return;
        if (dma_lane) {
            int wvo = wv;                                      // opaque per call: the 13 row offsets are re-made on the scalar
            asm volatile("" : "+s"(wvo));                      // unit each chunk instead of living in 13 vector registers
#pragma unroll
            for (int n = 0; n < DB_NDMA_WAVE; ++n) {
                const int i = wvo + 4 * n;                     // wave-uniform
                if (n < DB_NDMA_WAVE - 1 || i < DB_NDMA) {
                    const int hz = i / 5, hp = i - 5 * hz;
                    int gz = z0 + hz - 1;
                    gz = gz < 0 ? 0 : (gz > p.D - 1 ? p.D - 1 : gz);
                    int gy0 = y0 + 2 * hp - 1, gy1 = gy0 + 1;
                    gy0 = gy0 < 0 ? 0 : (gy0 > p.H - 1 ? p.H - 1 : gy0);
                    gy1 = gy1 > p.H - 1 ? p.H - 1 : gy1;
                    const unsigned ro0 = (unsigned)((((gz - zs) * p.H + gy0) * p.W) * p.CA);
                    const unsigned ro1 = (unsigned)((((gz - zs) * p.H + gy1) * p.W) * p.CA);
                    const unsigned vo = ((sb ? ro1 : ro0) + (unsigned)lane_el) * 4u;       // bytes from slab + kc * KC
                    const unsigned la = raw_lds + (unsigned)(i * (2 * DB_RAWROW));
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(slab + kc * KC), "s"(la)
                                 : "memory");      // M0 is written here; it cannot be named as a clobber (a reserved register:
                    // clang warns "may lead to undefined behaviour").  Nothing else in these kernels uses M0 (no movrel, no
                    // builtin LDS-DMA, no sendmsg): tests/test_host_cpu.py::test_m0_is_written_only_for_the_lds_dma_of_conv_wino4d
                }
            }
        }
    };

#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[u][0][i] = 0.f; acc[u][1][i] = 0.f; accB[u][i] = 0.f; }
    }

    // this wave's weight streams: S = KCN * 9 steps (chunk-major, (kd,kh)-minor); per step the four fragments of posA
    // ([cout half][hi | lo]) and the two of (posB, nbB); three register sets, step s in set s % 3 = tap % 3
    const int S = p.KCN * 9;
    const uint4* wbA = p.wp + (size_t)(nt * NPOS + posA) * S * (NF * 64) + lane;
    const uint4* wbB = p.wp + (size_t)(nt * NPOS + posB) * S * (NF * 64) + nbB * (NPL * 64) + lane;
    uint4 wq[3][NWF];
    auto fetch = [&](int s, uint4 (&dst)[NWF]) __attribute__((always_inline)) {
        if constexpr ((ABL & 1) != 0) { if (s > 2) return; }
        const int sc = s < S ? s : S - 1;
#pragma unroll
        for (int f = 0; f < NF; ++f) dst[f] = wbA[(size_t)sc * (NF * 64) + f * 64];
#pragma unroll
        for (int f = 0; f < NPL; ++f) dst[NF + f] = wbB[(size_t)sc * (NF * 64) + f * 64];
    };
    issue_dma(0);
    fetch(0, wq[0]);
    fetch(1, wq[1]);
    half8 a_abl[4][NPL];
    if constexpr ((ABL & 2) != 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int hl = 0; hl < NPL; ++hl)
#pragma unroll
                for (int j = 0; j < 8; ++j) a_abl[u][hl][j] = (_Float16)(0.01f * (float)((lane * 7 + u * 3 + hl * 5 + j) % 61) - 0.3f);
    }

    for (int kc = 0; kc < p.KCN; ++kc) {
        const int c0 = kc * KC;
        const float4 sc4 = *reinterpret_cast<const float4*>(p.scale + c0 + q4 * 4);
        const float4 sh4 = *reinterpret_cast<const float4*>(p.shift + c0 + q4 * 4);
        const float sc[4] = {sc4.x * a_scale, sc4.y * a_scale, sc4.z * a_scale, sc4.w * a_scale};
        const float sh[4] = {sh4.x * a_scale, sh4.y * a_scale, sh4.z * a_scale, sh4.w * a_scale};
        // this wave's pieces of the raw chunk have landed (and its weight fetches: the builtin, not inline assembly, so
        // that the compiler's own vmcnt bookkeeping starts from zero here and does not wait for them again later, which
        // would drag the pieces issued below into that wait)
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), expcnt and lgkmcnt left alone
        __syncthreads();                                   // ... everyone's; the previous chunk's plane readers are done
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (msk[it] < 0) continue;
            if constexpr ((ABL & 4) != 0) continue;
            const int r = (tid >> 2) + 64 * it;
            const float4* rp = reinterpret_cast<const float4*>(
                __builtin_assume_aligned(lds + RAW_OFF + r * DB_RAWROW + q4 * 16, 16));
            float4 v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = rp[i * 4];   // six 16-byte reads in flight, one wait
            float dd[6][4];                                // [x position][channel]: affine, zero padding after it
            if (inner) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float y[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) dd[i][c] = fmaf(y[c], sc[c], sh[c]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const bool ok = msk[it] & (1 << i);
                    const float y[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float a = fmaf(y[c], sc[c], sh[c]);
                        dd[i][c] = ok ? a : 0.f;
                    }
                }
            }
            unsigned char* dst = lds + st_plane + r * 16;
            float t[6][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float d0 = dd[0][c], d1 = dd[1][c], d2 = dd[2][c], d3 = dd[3][c], d4 = dd[4][c], d5 = dd[5][c];
                const float a = fmaf(-4.f, d2, d4);
                const float b = fmaf(-4.f, d1, d3);
                const float cc = d4 - d2;
                const float ee = 2.f * (d3 - d1);
                t[0][c] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
                t[1][c] = a + b;
                t[2][c] = a - b;
                t[3][c] = cc + ee;
                t[4][c] = cc - ee;
                t[5][c] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
            }
#pragma unroll
            for (int ps = 0; ps < NPOS; ++ps)
                split_store4_mix<NPASS == 3>(t[ps], dst + (ps * 2 * NPL) * DB_PLANE, DB_PLANE);
        }
        __syncthreads();                                   // planes complete; the raw buffer is free again
        // Tap 2's weights (taps 0 and 1 have theirs since the previous chunk; only two sets are live across the transform,
        // which has no registers to spare), then the next chunk's raw rows.  vmcnt is in order: the pieces complete when
        // tap 2 waits for its weights with the fetches of taps 1 and 2 behind them -- two taps (>= 1.2 k cycles of MFMA)
        // to land, off every wave's critical path
        fetch(kc * 9 + 2, wq[2]);
        if (kc + 1 < p.KCN) issue_dma(kc + 1);

#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int s = kc * 9 + t;
            const int kd = t / 3, kh = t - kd * 3;
            const int toff = (kd * DB_HT + kh) * 16;
            constexpr int dummy = 0; (void)dummy;
            const int cur = t % 3;
            if (t >= 1) fetch(s + 2, wq[(t + 2) % 3]);     // into the set tap t - 1 used
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                half8 a[NPL];
#pragma unroll
                for (int hl = 0; hl < NPL; ++hl) {
                    if constexpr ((ABL & 2) != 0) a[hl] = a_abl[u][hl];
                    else a[hl] = *reinterpret_cast<const half8*>(lds + a_off[u] + hl * DB_PLANE + toff);
                }
                if constexpr ((ABL & 64) != 0) {
                    float ss = 0.f;
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl) ss += (float)a[hl][0] + (float)a[hl][7];
                    ss += __builtin_bit_cast(float, wq[cur][0].x) + __builtin_bit_cast(float, wq[cur][NWF - 1].w);
                    if (u < 2) acc[u][0][0] += ss; else accB[u - 2][0] += ss;
                    continue;
                }
                if (u < 2) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const half8 bhi = __builtin_bit_cast(half8, wq[cur][nb * NPL]);
                        if constexpr (NPASS == 3) {
                            const half8 blo = __builtin_bit_cast(half8, wq[cur][nb * NPL + 1]);
                            acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bhi, acc[u][nb], 0, 0, 0);
                            acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], blo, acc[u][nb], 0, 0, 0);
                        }
                        acc[u][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bhi, acc[u][nb], 0, 0, 0);
                    }
                } else {
                    const half8 bhi = __builtin_bit_cast(half8, wq[cur][NF]);
                    if constexpr (NPASS == 3) {
                        const half8 blo = __builtin_bit_cast(half8, wq[cur][NF + 1]);
                        accB[u - 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bhi, accB[u - 2], 0, 0, 0);
                        accB[u - 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], blo, accB[u - 2], 0, 0, 0);
                    }
                    accB[u - 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bhi, accB[u - 2], 0, 0, 0);
                }
            }
        }
    }

    }   // !UNI

    if constexpr ((ABL & 16) != 0) {
        float sm = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) sm += acc[u][0][i] + acc[u][1][i] + accB[u][i];
        if (sm == 12345.678f) p.out[tid] = sm;
        return;
    }
    // ================= epilogue: output transform through LDS =================
    // thread (col, rg): column col of the 32-cout block, the quads (d, h) = (it, rg), it = 0..7; quad (d, h) sits in
    // accumulator row 32 ((d & 3) >> 1) + row_unperm(16 (d & 1) + 8 (d >> 2) + h)
    float* m = reinterpret_cast<float*>(lds);                      // [6 positions][64 accumulator rows][MLD]
    const int col = tid & 31, rg = tid >> 5;
    const bool interior = z0 + DB_TD <= p.D && y0 + DB_TH <= p.H && x0 + 4 <= p.W;      // wave-uniform
    int un[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) un[c] = row_unperm(16 * (c >> 1) + 8 * (c & 1) + rg);    // c = 2 (d & 1) + (d >> 2)
    const unsigned off_t = (unsigned)((rg * p.W) * p.Cout + col);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        if constexpr (!UNI && (ABL & 256) == 0) {
        __syncthreads();                                           // A planes (or the previous round) fully consumed
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float* mw = m + (posA * 64 + u * 32 + khalf * 4) * MLD + l32;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                mw[((i >> 2) * 8 + (i & 3)) * MLD] = acc[u][nb][i];
        }
        if (nbB == nb) {                                           // wave-uniform: this cout half of posB is this wave's
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float* mw = m + (posB * 64 + u * 32 + khalf * 4) * MLD + l32;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    mw[((i >> 2) * 8 + (i & 3)) * MLD] = accB[u][i];
            }
        }
        __syncthreads();
        } else if (nb == 1 && p.rsum != nullptr) __syncthreads();  // the previous round's moment fold is done with its scratch
        float fs = 0.f, fq = 0.f, fmn = INFINITY, fmx = -INFINITY;
        float* ob = p.out + nt * 64 + nb * 32 + ((int64_t)(z0 * p.H + y0) * p.W + x0) * p.Cout;
        if constexpr ((ABL & 512) != 0) ob = p.out + nt * 64 + nb * 32 + (mt & 127) * 16384;    // the same stores into 8 MB
        const float* mc = m + col;
        if (interior) {
            float prev[8][4];
            if (p.accum) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float* o = ob + (int64_t)it * p.H * p.W * p.Cout;
#pragma unroll
                    for (int k = 0; k < 4; ++k) prev[it][k] = o[off_t + (unsigned)(k * p.Cout)];
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                float* o = ob + (int64_t)it * p.H * p.W * p.Cout;
                if constexpr ((ABL & 512) != 0) o = ob + it * 2048 - (int)off_t + (rg * 256 + col);
                float y[4];
                if constexpr (UNI) {
                    const float4 yy = ybuf[(nb * 8 + it) * NTHR];
                    y[0] = yy.x; y[1] = yy.y; y[2] = yy.z; y[3] = yy.w;
                } else {
                    const int qr = 32 * ((it & 3) >> 1) + un[2 * (it & 1) + (it >> 2)];
                    const float* mr = mc + qr * MLD;
                    float m0, m1, m2, m3, m4, m5;
                    if constexpr ((ABL & 256) != 0) {
                        m0 = acc[0][nb][it]; m1 = acc[1][nb][it]; m2 = accB[0][it]; m3 = accB[1][it];
                        m4 = acc[0][nb][it + 8]; m5 = acc[1][nb][it + 8];
                    } else {
                        m0 = mr[0 * 64 * MLD]; m1 = mr[1 * 64 * MLD]; m2 = mr[2 * 64 * MLD];
                        m3 = mr[3 * 64 * MLD]; m4 = mr[4 * 64 * MLD]; m5 = mr[5 * 64 * MLD];
                    }
                    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                    y[0] = ((m0 + s1) + s2) * dq;
                    y[1] = fmaf(2.f, d2, d1) * dq;
                    y[2] = fmaf(4.f, s2, s1) * dq;
                    y[3] = (fmaf(8.f, d2, d1) + m5) * dq;
                    if (MODE == 2 && rep) ybuf[(nb * 8 + it) * NTHR] = make_float4(y[0], y[1], y[2], y[3]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float r = y[k];
                    if (p.accum) r = r + prev[it][k];
                    r = r >= 0.f ? r : r * p.slope;
                    if constexpr ((ABL & 128) == 0) o[off_t + (unsigned)(k * p.Cout)] = r;
                    fs += r; fq = fmaf(r, r, fq); fmn = fminf(fmn, r); fmx = fmaxf(fmx, r);
                }
            }
            if constexpr ((ABL & 128) != 0) { if (fs == 12345.678f) p.out[tid] = fs + fq + fmn + fmx; }
        } else if (y0 + rg < p.H) {
#pragma unroll 2
            for (int it = 0; it < 8; ++it) {
                if (z0 + it >= p.D) break;
                float* o = ob + (int64_t)it * p.H * p.W * p.Cout;
                float y[4];
                if constexpr (UNI) {
                    const float4 yy = ybuf[(nb * 8 + it) * NTHR];
                    y[0] = yy.x; y[1] = yy.y; y[2] = yy.z; y[3] = yy.w;
                } else {
                    const int qr = 32 * ((it & 3) >> 1) + row_unperm(16 * (it & 1) + 8 * (it >> 2) + rg);
                    const float* mr = mc + qr * MLD;
                    const float m0 = mr[0 * 64 * MLD], m1 = mr[1 * 64 * MLD], m2 = mr[2 * 64 * MLD];
                    const float m3 = mr[3 * 64 * MLD], m4 = mr[4 * 64 * MLD], m5 = mr[5 * 64 * MLD];
                    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                    y[0] = ((m0 + s1) + s2) * dq;
                    y[1] = fmaf(2.f, d2, d1) * dq;
                    y[2] = fmaf(4.f, s2, s1) * dq;
                    y[3] = (fmaf(8.f, d2, d1) + m5) * dq;
                    if (MODE == 2 && rep) ybuf[(nb * 8 + it) * NTHR] = make_float4(y[0], y[1], y[2], y[3]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (x0 + k < p.W) {
                        float r = y[k];
                        if (p.accum) r = r + o[off_t + (unsigned)(k * p.Cout)];
                        r = r >= 0.f ? r : r * p.slope;
                        o[off_t + (unsigned)(k * p.Cout)] = r;
                        fs += r; fq = fmaf(r, r, fq); fmn = fminf(fmn, r); fmx = fmaxf(fmx, r);
                    }
                }
            }
        }
        if (p.rsum != nullptr) {
            double* ls = reinterpret_cast<double*>(lds + (UNI ? (size_t)0 : (size_t)NPOS * 64 * MLD * sizeof(float)));   // [8][32]
            double* lq = ls + NRG * 32;
            float* lmn = reinterpret_cast<float*>(lq + NRG * 32);
            float* lmx = lmn + NRG * 32;
            ls[rg * 32 + col] = (double)fs; lq[rg * 32 + col] = (double)fq;
            lmn[rg * 32 + col] = fmn; lmx[rg * 32 + col] = fmx;
            __syncthreads();
            if (tid < 32) {
                double Ssum = 0.0, Q = 0.0;
                float MN = INFINITY, MX = -INFINITY;
#pragma unroll
                for (int r = 0; r < NRG; ++r) {
                    Ssum += ls[r * 32 + tid]; Q += lq[r * 32 + tid];
                    MN = fminf(MN, lmn[r * 32 + tid]); MX = fmaxf(MX, lmx[r * 32 + tid]);
                }
                const size_t o = (size_t)mt * p.Cout + nt * 64 + nb * 32 + tid;
                p.rsum[o] = Ssum; p.rsq[o] = Q; p.rmn[o] = MN; p.rmx[o] = MX;
            }
        }
    }
}

template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino4d(const W4Params p) { conv_wino4d_body<NPASS, 0>(p); }
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino4d_masked(const W4Params p) { conv_wino4d_body<NPASS, 1>(p); }
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino4d_rest(const W4Params p) { conv_wino4d_body<NPASS, 2>(p); }
template <int NPASS>
__global__ void __launch_bounds__(NTHR, 2) conv_wino4d_uniform(const W4Params p) { conv_wino4d_body<NPASS, 3>(p); }
#ifdef BFM_W4_ABLATE
template <int ABL>
__global__ void __launch_bounds__(NTHR, 2) conv_wino4d_abl(const W4Params p) { conv_wino4d_body<3, 0, ABL>(p); }
#endif

// packed[ntile64][pos 6][kc][(kd,kh) 9][nb 2][hl][lane] (uint4 = 8 halfs): lane l holds
// B[k = 8*(l>>5)+j][n = l&31] = U_pos[co = ntile*64 + nb*32 + (l&31)][ci = kc*16 + 8*(l>>5) + j][kd][kh] * 2^wexp, U = G g along kw
__global__ void pack_wino4(const float* __restrict__ w, int Cin, int Cout, int wexp, int npl, uint4* __restrict__ out) {
    const int KCN = Cin / KC;
    const int nf = 2 * npl;
    const int64_t n = (int64_t)(Cout / 64) * NPOS * KCN * 9 * nf * 64;
    const double s = ldexp(1.0, wexp);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int f = (int)(r % nf); r /= nf;
        const int t = (int)(r % 9); r /= 9;
        const int kc = (int)(r % KCN); r /= KCN;
        const int ps = (int)(r % NPOS); r /= NPOS;
        const int ntile = (int)r;
        const int nb = f / npl, hl = f - nb * npl;
        const int co = ntile * 64 + nb * 32 + (lane & 31);
        const int ci0 = kc * KC + 8 * (lane >> 5);
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* g = w + ((size_t)co * Cin + ci0 + j) * 27 + t * 3;
            const double g0 = g[0], g1 = g[1], g2 = g[2];
            const double u = ps == 0 ? g0 / 4.0
                           : ps == 1 ? -((g0 + g1) + g2) / 6.0
                           : ps == 2 ? -((g0 - g1) + g2) / 6.0
                           : ps == 3 ? g0 / 24.0 + g1 / 12.0 + g2 / 6.0
                           : ps == 4 ? g0 / 24.0 - g1 / 12.0 + g2 / 6.0
                                     : g2;
            const float x = (float)(u * s);
            const _Float16 hh = (_Float16)x;
            v[j] = hl == 0 ? hh : (_Float16)(x - (float)hh);
        }
        out[i] = __builtin_bit_cast(uint4, v);
    }
}

int ilog2i(int v) { int r = 0; while ((1 << r) < v) ++r; return r; }

// The box is conv_wino's (bfm_wino_choose_box: the flags and lists of the sparse forms are per box and are shared by the
// two kernels); it always has 64 quads' worth of voxels, what is left to check is that its x side is a whole number of
// quads and that the six positions' planes fit the LDS of two workgroups per CU.
bool choose_box4(int D, int H, int W, int npl, int& TD, int& TH, int& TW) {
    if (!bfm_wino_choose_box(D, H, W, npl, TD, TH, TW)) return false;
    if (TW % 4 || TD * TH * (TW / 4) != 64) return false;
    const int64_t npos = (int64_t)(TD + 2) * (TH + 2) * (TW / 4);
    if (npos * 4 > 3 * NTHR) return false;
    const int64_t plane = ((npos * 16 + 255) / 256) * 256 + 16;
    return 2 * NPOS * npl * plane <= 62 * 1024;
}

// conv_wino4d (the 8 x 8 x 4 box with the LDS-DMA staging) takes every volume with at least one full box in z and y;
// thinner ones keep conv_wino4 and conv_wino's box
bool use_dma_kernel(int D, int H, int W) {
#ifdef BFM_W4_ABLATE
    if (getenv("BFM_W4_OLD")) return false;                    // diagnostics builds: time conv_wino4 beside conv_wino4d
#endif
    return D >= DB_TD && H >= DB_TH && W >= 4;
}

bool choose_box_any(int D, int H, int W, int npl, int& TD, int& TH, int& TW) {
    if (use_dma_kernel(D, H, W)) { TD = DB_TD; TH = DB_TH; TW = 4; return true; }
    return choose_box4(D, H, W, npl, TD, TH, TW);
}

}  // namespace

extern "C" size_t bfm_pack_conv_weights_wino4_bytes(int Cin, int Cout, int passes) {
    if (Cin <= 0 || Cout <= 0 || Cin % KC || Cout % 64) return 0;
    const int npl = passes == 3 ? 2 : 1;
    return (size_t)(Cout / 64) * NPOS * (Cin / KC) * 9 * 2 * npl * 64 * sizeof(uint4);
}

extern "C" int bfm_pack_conv_weights_wino4(const float* w_oidhw, int Cin, int Cout, float wmax_abs_host, int passes,
                                           void* wpacked, int* wexp_host, bfm_stream_t stream) {
    if (!w_oidhw || !wpacked || !wexp_host || Cin <= 0 || Cout <= 0) return BFM_E_ARG;
    if (Cin % KC || Cout % 64 || (passes != 1 && passes != 3)) return BFM_E_SHAPE;
    int wexp = 0;
    if (wmax_abs_host > 0.f && wmax_abs_host < INFINITY) {
        int ex;
        (void)frexpf(wmax_abs_host, &ex);                        // |U| <= max|g| (the rows of G sum to at most 1)
        wexp = 14 - ex;
        wexp = wexp > 60 ? 60 : (wexp < -60 ? -60 : wexp);
    }
    *wexp_host = wexp;
    const int npl = passes == 3 ? 2 : 1;
    const int64_t n = (int64_t)(Cout / 64) * NPOS * (Cin / KC) * 9 * 2 * npl * 64;
    const int nb = (int)std::min<int64_t>(8192, bfm_cdiv64(n, 256));
    hipLaunchKernelGGL(pack_wino4, dim3(nb), dim3(256), 0, bfm_s(stream), w_oidhw, Cin, Cout, wexp, npl,
                       static_cast<uint4*>(wpacked));
    return bfm_launch_status();
}

// rows of the output-moment table the kernel writes for this volume (its own box choice), 0 if it cannot run
extern "C" int bfm_conv3x3x3_wino4_rows(int D, int H, int W, int passes) {
    int TD, TH, TW;
    if (D <= 0 || H <= 0 || W <= 0 || !choose_box_any(D, H, W, passes == 3 ? 2 : 1, TD, TH, TW)) return 0;
    return bfm_cdiv(D, TD) * bfm_cdiv(H, TH) * bfm_cdiv(W, TW);
}

// the box of output voxels per workgroup for this volume (what the masked form's "boxes that hold input" are)
extern "C" int bfm_conv3x3x3_wino4_box(int D, int H, int W, int passes, int* box) {
    int TD, TH, TW;
    if (!box || D <= 0 || H <= 0 || W <= 0 || !choose_box_any(D, H, W, passes == 3 ? 2 : 1, TD, TH, TW)) return BFM_E_ARG;
    box[0] = TD; box[1] = TH; box[2] = TW;
    return BFM_OK;
}

// bytes of the masked form's workspace: box activity (padded to 4), the count, the list
extern "C" size_t bfm_conv3x3x3_wino4_masked_workspace(int D, int H, int W, int passes) {
    const int n = bfm_conv3x3x3_wino4_rows(D, H, W, passes);
    if (n <= 0) return 0;
    return (((size_t)n + 3) & ~(size_t)3) + 4 + (size_t)n * 4;
}

static int w4_launch(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift, const float* bound,
                     int G, const void* wpacked, int wexp, int Cout, float slope, int passes, int flags, float* out,
                     void* moment_rows, const float* mask_img, void* mask_ws, bfm_stream_t stream, int S = 1,
                     int affine_stride = 0, const unsigned char* uni_flags = nullptr, float* uni_acc = nullptr) {
    const int accumulate = flags & 1;
    if (flags & ~1) return BFM_E_ARG;                           // bit 0 = accumulate; nothing else is defined
    if (!A || CA <= 0 || D <= 0 || H <= 0 || W <= 0 || !scale || !shift || !bound || G <= 0 || !wpacked || !out)
        return BFM_E_ARG;
    if (CA % KC || Cout % 64 || Cout <= 0) return BFM_E_SHAPE;
    if (passes != 1 && passes != 3) return BFM_E_ARG;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(scale) & 15) ||
        (reinterpret_cast<uintptr_t>(shift) & 15) || (reinterpret_cast<uintptr_t>(wpacked) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return BFM_E_ARG;
    if ((int64_t)D * H * W * CA > 0x7fffffffLL) return BFM_E_SHAPE;       // 32-bit staging offsets
    const int npl = passes == 3 ? 2 : 1;
    W4Params p{};
    p.A = A; p.CA = CA; p.D = D; p.H = H; p.W = W;
    p.scale = scale; p.shift = shift; p.bound = bound; p.G = G;
    p.wp = static_cast<const uint4*>(wpacked);
    p.wexp = wexp; p.Cout = Cout; p.slope = slope; p.out = out; p.accum = accumulate ? 1 : 0;
    if (!choose_box_any(D, H, W, npl, p.TD, p.TH, p.TW)) return BFM_E_SHAPE;
    const bool dma = use_dma_kernel(D, H, W);
    if (dma && (int64_t)(DB_TD + 2) * H * W * CA >= (int64_t)1 << 30) return BFM_E_SHAPE;   // 32-bit byte offsets in a slab
    p.HT = p.TH + 2; p.QW = p.TW / 4;
    p.qw_shift = ilog2i(p.QW); p.thq_shift = ilog2i(p.TH * p.QW);
    const int nTz = bfm_cdiv(D, p.TD);
    p.nTy = bfm_cdiv(H, p.TH); p.nTx = bfm_cdiv(W, p.TW);
    p.nMt = nTz * p.nTy * p.nTx;
    if (S > 1 || affine_stride > 0) {                           // batch of S same-shape samples
        if (mask_img || S < 1 || (affine_stride != 0 && (affine_stride < CA || (affine_stride & 3)))) return BFM_E_ARG;
        p.nMtS = p.nMt;
        p.nMt = S * p.nMtS;
        p.saff = affine_stride > 0 ? affine_stride : CA;
        p.sA = (int64_t)D * H * W * CA;
        p.sO = (int64_t)D * H * W * Cout;
    }
    p.NT = Cout / 64;
    p.KCN = CA / KC;
    p.npos_lds = (p.TD + 2) * p.HT * p.QW;
    p.plane_stride = ((p.npos_lds * 16 + 255) / 256) * 256 + 16;
    size_t smem = dma ? (size_t)2 * NPOS * npl * DB_PLANE + DB_RAW : (size_t)2 * NPOS * npl * p.plane_stride;
    const size_t epi = (size_t)NPOS * 64 * MLD * sizeof(float) + (size_t)NRG * 32 * 24;   // output-transform scratch + moment fold
    if (smem < epi) smem = epi;
    if (smem > (dma ? 80u : 64u) * 1024) return BFM_E_SHAPE;
    if (uni_flags) {                                            // the pair: conv_wino4d only, on conv_wino's box grid
        int bd, bh, bw;
        if (!dma || mask_img || S > 1 || affine_stride > 0 || !uni_acc) return BFM_E_ARG;
        if (!bfm_wino_choose_box(D, H, W, npl, bd, bh, bw) || bd != p.TD || bh != p.TH || bw != p.TW) return BFM_E_SHAPE;
        p.uni_flags = uni_flags;
        p.uni_acc = uni_acc;
    }
    if (dma) {                                                  // more than 64 KB of dynamic LDS: once per process
        static bool attr = false;
        if (!attr) {
            const int lim = 80 * 1024;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4d_rest<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4d_rest<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess)
                return BFM_E_LAUNCH;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4d<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4d<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4d_masked<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4d_masked<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lim) != hipSuccess)
                return BFM_E_LAUNCH;
            attr = true;
        }
    }
    if (moment_rows) {
        if (reinterpret_cast<uintptr_t>(moment_rows) & 7) return BFM_E_ARG;
        char* rb = static_cast<char*>(moment_rows);
        const size_t n = (size_t)p.nMt * Cout;
        p.rsum = reinterpret_cast<double*>(rb);
        p.rsq = reinterpret_cast<double*>(rb + n * 8);
        p.rmn = reinterpret_cast<float*>(rb + n * 16);
        p.rmx = reinterpret_cast<float*>(rb + n * 20);
    }
    if ((int64_t)p.nMt * p.NT > 0x7fffffff) return BFM_E_SHAPE;
    dim3 grid((unsigned)(p.nMt * p.NT));
    hipStream_t st = bfm_s(stream);
    // the sparse forms take their boxes from a list built on the device; workgroups beyond the list end at once
    if (mask_img) {
        if (!mask_ws) return BFM_E_ARG;
        const int rc = bfm_wino_mask_list(mask_img, D, H, W, p.TD, p.TH, p.TW, p.nTy, p.nTx, p.nMt, mask_ws, stream);
        if (rc != BFM_OK) return rc;
        int* cnt = reinterpret_cast<int*>(static_cast<unsigned char*>(mask_ws) + (((size_t)p.nMt + 3) & ~(size_t)3));
        p.list = cnt + 1; p.list_n = cnt;
        if (dma) {
            if (passes == 3) hipLaunchKernelGGL(conv_wino4d_masked<3>, grid, dim3(NTHR), smem, st, p);
            else hipLaunchKernelGGL(conv_wino4d_masked<1>, grid, dim3(NTHR), smem, st, p);
        } else if (passes == 3) hipLaunchKernelGGL(conv_wino4_masked<3>, grid, dim3(NTHR), smem, st, p);
        else hipLaunchKernelGGL(conv_wino4_masked<1>, grid, dim3(NTHR), smem, st, p);
        return bfm_launch_status();
    }
    if (uni_flags) {                                            // disjoint boxes: the two launches may overlap
        // flags [nMt pad 4] | first box of each class [27] | the two counts | conv_wino4d_rest's list | conv_wino4d_uniform's
        const int* cnt = reinterpret_cast<const int*>(uni_flags + (((size_t)p.nMt + 3) & ~(size_t)3)) + 27;
        p.list = cnt + 2; p.list_n = cnt;
        if (passes == 3) hipLaunchKernelGGL(conv_wino4d_rest<3>, grid, dim3(NTHR), smem, st, p);
        else hipLaunchKernelGGL(conv_wino4d_rest<1>, grid, dim3(NTHR), smem, st, p);
        p.list = cnt + 2 + p.nMt; p.list_n = cnt + 1;
        if (passes == 3) hipLaunchKernelGGL(conv_wino4d_uniform<3>, grid, dim3(NTHR), 6144, st, p);
        else hipLaunchKernelGGL(conv_wino4d_uniform<1>, grid, dim3(NTHR), 6144, st, p);
        return bfm_launch_status();
    }
#ifdef BFM_W4_ABLATE
    if (const char* e = getenv("BFM_W4_SLEEP")) p.dbg_sleep = atoi(e);
    if (const char* e = getenv("BFM_W4_ABL")) {                 // diagnostics: phases compiled out, the old kernel by "old"
        const int abl = atoi(e);
        if (dma) {
#define W4D_ATTR(N) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino4d_abl<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)
            W4D_ATTR(0); W4D_ATTR(1); W4D_ATTR(2); W4D_ATTR(3); W4D_ATTR(4); W4D_ATTR(8); W4D_ATTR(12); W4D_ATTR(15); W4D_ATTR(16); W4D_ATTR(31); W4D_ATTR(64); W4D_ATTR(28); W4D_ATTR(128); W4D_ATTR(256); W4D_ATTR(384); W4D_ATTR(512);
            switch (abl) {
#define W4D_CASE(N) case N: hipLaunchKernelGGL(conv_wino4d_abl<N>, grid, dim3(NTHR), smem, st, p); return bfm_launch_status()
                W4D_CASE(0); W4D_CASE(1); W4D_CASE(2); W4D_CASE(3); W4D_CASE(4); W4D_CASE(8); W4D_CASE(12); W4D_CASE(15); W4D_CASE(16); W4D_CASE(31); W4D_CASE(64); W4D_CASE(28); W4D_CASE(128); W4D_CASE(256); W4D_CASE(384); W4D_CASE(512);
                default: return BFM_E_ARG;
            }
        }
    }
#endif
    if (dma) {
        if (passes == 3) hipLaunchKernelGGL(conv_wino4d<3>, grid, dim3(NTHR), smem, st, p);
        else hipLaunchKernelGGL(conv_wino4d<1>, grid, dim3(NTHR), smem, st, p);
    } else if (passes == 3) hipLaunchKernelGGL(conv_wino4<3>, grid, dim3(NTHR), smem, st, p);
    else hipLaunchKernelGGL(conv_wino4<1>, grid, dim3(NTHR), smem, st, p);
    return bfm_launch_status();
}

extern "C" int bfm_conv3x3x3_wino4(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                                   const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                   int passes, int flags, float* out, void* moment_rows, bfm_stream_t stream) {
    return w4_launch(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out, moment_rows,
                     nullptr, nullptr, stream);
}

// A (S,D,H,W,CA) -> out (S,D,H,W,Cout): S same-shape samples in one launch, per-sample scale / shift (rows affine_stride
// apart, 0 = CA) and bound [S][G]; moment_rows [S * bfm_conv3x3x3_wino4_rows(...)][Cout].  Per sample the bits of
// bfm_conv3x3x3_wino4 (the deep levels of same-shape tiles, engine.batch_conv).
extern "C" int bfm_conv3x3x3_wino4_batch(const float* A, int CA, int S, int D, int H, int W, const float* scale,
                                         const float* shift, const float* bound, int G, const void* wpacked, int wexp,
                                         int Cout, float slope, int passes, int flags, float* out, void* moment_rows,
                                         int affine_stride, bfm_stream_t stream) {
    if (S < 1) return BFM_E_ARG;
    return w4_launch(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out, moment_rows,
                     nullptr, nullptr, stream, S, affine_stride > 0 ? affine_stride : CA);
}

// the tile loop's last convolution (boxes that hold input only); workspace: bfm_conv3x3x3_wino_masked_workspace() bytes
extern "C" int bfm_conv3x3x3_wino4_masked(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                                          const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                          int passes, int flags, float* out, const float* mask_image, void* workspace,
                                          size_t workspace_bytes, bfm_stream_t stream) {
    if (!mask_image || (flags & ~1)) return BFM_E_ARG;         // no moment rows (boxes are left out)
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 3) ||
        workspace_bytes < bfm_conv3x3x3_wino4_masked_workspace(D, H, W, passes))
        return BFM_E_ARG;
    return w4_launch(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out, nullptr,
                     mask_image, workspace, stream);
}

// The uniform-box pair of the same kernel (bfm_conv3x3x3_wino_uniform's counterpart: same flags from bfm_uniform_boxes,
// same bits with the flags and without).  For layers whose output no other uniform-box layer reads (conv_wino4d_body);
// BFM_E_SHAPE where the volume's box is not conv_wino's.  scratch: bfm_conv3x3x3_wino4_uniform_scratch(Cout) bytes.
extern "C" size_t bfm_conv3x3x3_wino4_uniform_scratch(int Cout) {
    if (Cout <= 0 || Cout % 64) return 0;
    return (size_t)27 * (Cout / 64) * 2 * 8 * NTHR * sizeof(float4);
}

extern "C" int bfm_conv3x3x3_wino4_uniform(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                                           const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                           int passes, int flags, float* out, void* moment_rows,
                                           const unsigned char* uniform_flags, void* scratch, bfm_stream_t stream) {
    if (!uniform_flags || !scratch || (reinterpret_cast<uintptr_t>(uniform_flags) & 3) || (reinterpret_cast<uintptr_t>(scratch) & 15))
        return BFM_E_ARG;
    return w4_launch(A, CA, D, H, W, scale, shift, bound, G, wpacked, wexp, Cout, slope, passes, flags, out, moment_rows,
                     nullptr, nullptr, stream, 1, 0, uniform_flags, static_cast<float*>(scratch));
}
