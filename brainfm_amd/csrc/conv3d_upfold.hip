// The upsampled half of a decoder's first convolution, with the nearest 2x upsample folded into the weights.
//
// Decoder._joining (Trainer/models/unet3d/buildingblocks.py:265-276, 361-363) feeds
// cat((skip, nearest_up_2x(x))) to a 3x3x3 conv.  On the upsampled channels every low-res voxel is seen 8 times,
// so along each axis the three taps of an output voxel of parity p land on only two low-res voxels:
//      p = 0:  (i-1, i, i)   -> weights ( w[-1],        w[0] + w[+1] )
//      p = 1:  (i, i, i+1)   -> weights ( w[-1] + w[0], w[+1]        )
// (zero padding outside [0, 2n) maps exactly onto low-res indices -1 and n).  Each of the 8 output parity classes
// is therefore a 2x2x2 convolution over the low-res tensor with pre-summed weights: 8 taps instead of 27 on these
// channels -- 3.375x fewer matrix-core FLOPs on two thirds of the decoder's input channels -- and the 8x smaller
// low-res box is what gets staged.  Summing weights first changes rounding at the fp32 epsilon level only.
//
// GEMM view: M = low-res voxels (a BDxBHxBW box of 128 per workgroup), N = 8 classes x Cout, K = 8 taps x CB.
// 8 waves per workgroup, wave = parity class (4x2 blocks of 32x32: 128 rows x 64 couts); the class's weight
// fragments are private to the wave and stream L2 -> VGPR two taps ahead.  Same split-fp16 numerics as
// conv3d_mfma.hip (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate).
// The result (no activation) is written to the interleaved full-res positions of `out`; the skip half of the
// convolution then runs through bfm_conv3x3x3_mfma with the accumulate flag (cfg[7] bit 0) and applies LeakyReLU.
#include "bfm_common.h"
#include <type_traits>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16;
constexpr int NTHR = 512;

struct UpParams {
    const float* B;
    int CB, d, h, w;                 // low-res tensor (channels-last)
    const float *scale, *shift;      // GroupNorm affine of the B channels (already offset by CA)
    int saff;                        // elements between two samples' scale / shift rows
    const float* bound;
    int G;
    const uint4* wp;
    int wexp, Cout;
    float* out;                      // [2d][2h][2w][Cout]
    int BD, BH, BW, HT, WT;          // low-res box and its halo'd extents
    int BHW, BVOX;                   // BH * BW, BD * BH * BW (<= 128; rows past it are padding)
    int nTy, nTx, nMt, NT, KCB;
    int nvox_lds, plane_stride;
    int rowoff_lds;                  // byte offset of the epilogue's row table (128 ints) in LDS
    int kc_per_split;                // K chunks per blockIdx.y slab (a multiple of 3: the weight ring's phase period)
    float* slab;                     // split-K: [sample][gridDim.y][2d][2h][2w][Cout] partial sums, summed by upfold_reduce
    int64_t slab_stride;
    // batch: blockIdx.z = sample; B, out advance by sB, sO elements per sample, scale / shift by saff (CB when the tables are the B half alone), bound by G.  A
    // sample's workgroups do exactly what they do in a launch of that sample alone.
    int64_t sB, sO;
    int dbg_sleep;                   // diagnostics builds: odd CUs' first workgroups start this many kilocycles late
};

__device__ __forceinline__ int row_perm(int l) {        // same lane -> row order as conv_mfma (conflict-free b128)
    if (l < 4) return l;
    if (l < 12) return l + 12;
    if (l < 16) return l - 8;
    if (l < 20) return l + 8;
    if (l < 28) return l - 12;
    return l;
}

__device__ __forceinline__ void box_coords(const UpParams& p, int q, int& bd, int& bh, int& bw) {
    bd = q / p.BHW;
    const int rem = q - bd * p.BHW;
    bh = rem / p.BW;
    bw = rem - bh * p.BW;
}

// ABL (diagnostics builds only, -DBFM_UP_ABLATE, tests/diag/diag_upfold_ablate.py): phases compiled out to see what the
// launch time is made of.  1: no weight loads in the tap loop, 2: no LDS operand reads, 4: no staging at all, 8: staging
// without its global loads, 16: no epilogue, 64: no MFMAs.  Ablated launches compute wrong results.
// NW (round 6): waves per workgroup.  8 = one workgroup holds all eight parity classes of a box (512 threads, one workgroup
// per CU: its staging and epilogue phases leave the matrix pipe idle -- 60 % busy, profiles/r06_conv_upfold_pmc.txt).
// 4 = a workgroup holds the classes of ONE z parity (256 threads, two workgroups per CU): the two halves of a box stage the
// same halo'd box independently and drift apart, so one half's staging / epilogue runs under the other's products.  A wave
// does exactly what it does in the 8-wave form (same products, same order, same stores): the bits do not change.
template <int NPASS, int ABL = 0, int NW = 8>
__device__ __forceinline__ void conv_upfold_body(const UpParams& p) {
    constexpr int NTHRV = NW * 64;
    constexpr int NPL = (NPASS == 3) ? 2 : 1;
    constexpr int NF = 2 * NPL;                       // weight fragments per tap: 2 column blocks x (hi[, lo])
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    int xcd_ = blockIdx.x & 7, idx_ = blockIdx.x >> 3, half_ = 0;
    if constexpr (NW == 4) {                                       // id -> (xcd, half, idx): both halves of a box on one XCD
        half_ = idx_ & 1;
        idx_ >>= 1;
        const int nblk = p.nMt * p.NT;
        if (idx_ >= (nblk >> 3) + (xcd_ < (nblk & 7) ? 1 : 0)) return;     // the grid is padded to whole octets
    }
    const int cls = __builtin_amdgcn_readfirstlane(tid >> 6) + half_ * 4;  // wave = output parity class (pz,py,px)
    const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
    const int l32 = lane & 31, khalf = lane >> 5;
#ifdef BFM_UP_ABLATE
    if (p.dbg_sleep > 0 && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1))   // de-phase neighbouring CUs (experiment)
        for (int i = 0; i < p.dbg_sleep; ++i) __builtin_amdgcn_s_sleep(16);
#endif

    int bid;
    {
        const int nblk = p.nMt * p.NT;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = xcd_, idx = idx_;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / p.NT, nt = bid - mt * p.NT;
    const int tx = mt % p.nTx;
    const int ty = (mt / p.nTx) % p.nTy;
    const int tz = mt / (p.nTx * p.nTy);
    const int z0 = tz * p.BD, y0 = ty * p.BH, x0 = tx * p.BW;      // low-res box origin
    const int smp = blockIdx.z;
    const float* const pB = p.B + smp * p.sB;
    const float* const pscale = p.scale + smp * p.saff;
    const float* const pshift = p.shift + smp * p.saff;
    const float* const pbound = p.bound + smp * p.G;

    float bmax = 0.f;
    for (int g = 0; g < p.G; ++g) bmax = fmaxf(bmax, pbound[g]);
    int aexp = 0;
    if (bmax > 0.f && bmax < INFINITY) {
        int ex;
        (void)frexpf(bmax, &ex);
        aexp = 14 - ex;
        aexp = aexp > 60 ? 60 : (aexp < -60 ? -60 : aexp);
    }
    const float a_scale = ldexpf(1.0f, aexp);
    const float dq = ldexpf(1.0f, -(aexp + p.wexp));

    // per-lane A base offsets of the four 32-row blocks: row (bd,bh,bw) of class (pz,py,px), tap (0,0,0) reads the
    // halo'd box at (bd+pz, bh+py, bw+px)
    int a_off[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        int bd, bh, bw;
        const int q = mb * 32 + row_perm(l32);
        box_coords(p, q < p.BVOX ? q : 0, bd, bh, bw);        // padding rows read row 0 (never stored)
        a_off[mb] = (khalf * NPL) * p.plane_stride + (((bd + pz) * p.HT + (bh + py)) * p.WT + (bw + px)) * 16;
    }

    // the epilogue's row table: element offset of row q's full-res voxel (class (0,0,0)) from the box's, -1 for rows that
    // are padding or outside the tensor; published by the chunk loop's first barrier
    int* const rowoff = reinterpret_cast<int*>(lds + p.rowoff_lds);
    if (tid < 128) {
        int bd, bh, bw;
        box_coords(p, tid < p.BVOX ? tid : 0, bd, bh, bw);
        const bool ok = tid < p.BVOX && z0 + bd < p.d && y0 + bh < p.h && x0 + bw < p.w;
        rowoff[tid] = ok ? (((2 * bd) * (2 * p.h) + 2 * bh) * (2 * p.w) + 2 * bw) * p.Cout : -1;
    }

    // staging bookkeeping: element e = tid + it*NTHR -> (halo voxel, channel quad)
    constexpr int MAX_IT = 2048 / NTHRV;
    const int n_el = p.nvox_lds * 4;
    const int q4 = tid & 3;
    int off[MAX_IT];
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int e = tid + it * NTHRV;
        off[it] = -2;
        if (e < n_el) {
            const int vox = e >> 2;
            const int hz = vox / (p.HT * p.WT);
            const int rem = vox - hz * (p.HT * p.WT);
            const int hy = rem / p.WT;
            const int hx = rem - hy * p.WT;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            off[it] = -1;
            if (gz >= 0 && gz < p.d && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w)
                off[it] = ((gz * p.h + gy) * p.w + gx) * p.CB;
        }
    }
    const int st_plane = ((q4 >> 1) * NPL) * p.plane_stride + (q4 & 1) * 8;

    floatx16 acc[4][2];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

    // this wave's weight stream: S = KCB*8 steps (chunk-major, tap-minor), NF fragments of 64 x uint4 per step
    const int S = p.KCB * 8;
    const uint4* wbase = p.wp + (size_t)(nt * 8 + cls) * S * (NF * 64) + lane;
    uint4 wq[3][NF];
    const int kc0_ = blockIdx.y * p.kc_per_split * 8;
    half8 a_abl[4][NPL];
    if constexpr ((ABL & 2) != 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int hl = 0; hl < NPL; ++hl)
#pragma unroll
                for (int j = 0; j < 8; ++j) a_abl[u][hl][j] = (_Float16)(0.01f * (float)((lane * 7 + u * 3 + hl * 5 + j) % 61) - 0.3f);
    }
    auto fetch = [&](int s, uint4 (&dst)[NF]) __attribute__((always_inline)) {
        if constexpr ((ABL & 1) != 0) { if (s > kc0_ + 1) return; }
        const int sc = s < S ? s : S - 1;
#pragma unroll
        for (int f = 0; f < NF; ++f) dst[f] = wbase[(size_t)sc * (NF * 64) + f * 64];
    };
    const int kc0 = blockIdx.y * p.kc_per_split;                  // kc0 % 3 == 0: (kc0 * 8) % 3 == 0, the ring starts at set 0
    const int kc1 = min(p.KCB, kc0 + p.kc_per_split);
    fetch(kc0 * 8, wq[0]);
    fetch(kc0 * 8 + 1, wq[1]);

    // one K-chunk; PH = (kc*8) % 3 is the ring phase, passed as a compile-time constant so that every register-set
    // index below is static (the chunk loop is unrolled by three: 8 % 3 == 2 advances the phase by two per chunk)
    auto do_chunk = [&](int kc, auto ph_tag) __attribute__((always_inline)) {
        constexpr int PH = decltype(ph_tag)::value;
        const int c0 = kc * KC;
        const float* src = pB + c0 + q4 * 4;
        const float4 sc4 = *reinterpret_cast<const float4*>(pscale + c0 + q4 * 4);
        const float4 sh4 = *reinterpret_cast<const float4*>(pshift + c0 + q4 * 4);
        const float sc[4] = {sc4.x * a_scale, sc4.y * a_scale, sc4.z * a_scale, sc4.w * a_scale};
        const float sh[4] = {sh4.x * a_scale, sh4.y * a_scale, sh4.z * a_scale, sh4.w * a_scale};
        __syncthreads();                                 // previous chunk's readers are done
#pragma unroll
        for (int u0 = 0; u0 < MAX_IT; u0 += 2) {
            if constexpr ((ABL & 4) != 0) continue;
            float4 v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr ((ABL & 8) != 0) v[u] = make_float4(0.1f * lane, 0.2f, 0.3f * kc, 0.4f);
                else
                if (off[u0 + u] >= 0) v[u] = *reinterpret_cast<const float4*>(src + off[u0 + u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = tid + (u0 + u) * NTHRV;
                if (off[u0 + u] != -2) {
                    const bool ok = off[u0 + u] >= 0;
                    float y[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    half4 hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t = ok ? fmaf(y[i], sc[i], sh[i]) : 0.f;      // zero padding AFTER the affine
                        _Float16 hh = (_Float16)t;
                        hi[i] = hh;
                        lo[i] = (_Float16)(t - (float)hh);
                    }
                    unsigned char* dst = lds + st_plane + (e >> 2) * 16;
                    *reinterpret_cast<half4*>(dst) = hi;
                    if constexpr (NPASS == 3) *reinterpret_cast<half4*>(dst + p.plane_stride) = lo;
                }
            }
        }
        __syncthreads();

#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // ring of three register sets: step s = kc*8 + t uses set (s % 3) and refills the set used one step ago
            const int s = kc * 8 + t;
            const int ta = t >> 2, tb = (t >> 1) & 1, tc = t & 1;
            const int toff = ((ta * p.HT + tb) * p.WT + tc) * 16;
            const int cur = (PH + t) % 3;                     // compile-time after unrolling
            uint4 bw[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) bw[f] = wq[cur][f];
            fetch(s + 2, wq[(cur + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);           // pin the prefetch: the scheduler would sink it to its first use
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                half8 a[NPL];
#pragma unroll
                for (int hl = 0; hl < NPL; ++hl) {
                    if constexpr ((ABL & 2) != 0) a[hl] = a_abl[mb][hl];
                    else a[hl] = *reinterpret_cast<const half8*>(lds + a_off[mb] + hl * p.plane_stride + toff);
                }
                if constexpr ((ABL & 64) != 0) {
                    float ss = 0.f;
#pragma unroll
                    for (int hl = 0; hl < NPL; ++hl) ss += (float)a[hl][0] + (float)a[hl][7];
                    ss += __builtin_bit_cast(float, bw[0].x) + __builtin_bit_cast(float, bw[NF - 1].w);
                    acc[mb][0][0] += ss;
                    continue;
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const half8 bhi = __builtin_bit_cast(half8, bw[nb * NPL]);
                    if constexpr (NPASS == 3) {
                        const half8 blo = __builtin_bit_cast(half8, bw[nb * NPL + 1]);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi, a[1], acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo, a[0], acc[mb][nb], 0, 0, 0);
                    }
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi, a[0], acc[mb][nb], 0, 0, 0);
                }
            }
        }
    };
    for (int kc = kc0; kc < kc1; kc += 3) {
        do_chunk(kc, std::integral_constant<int, 0>{});
        if (kc + 1 < kc1) do_chunk(kc + 1, std::integral_constant<int, 2>{});
        if (kc + 2 < kc1) do_chunk(kc + 2, std::integral_constant<int, 1>{});
    }

    if constexpr ((ABL & 16) != 0) {
        float sm = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int i = 0; i < 16; ++i) sm += acc[mb][0][i] + acc[mb][1][i];
        if (sm == 12345.678f) p.out[tid] = sm;
        return;
    }
    // epilogue: class (pz,py,px) of low-res voxel (zl,yl,xl) is full-res voxel (2zl+pz, 2yl+py, 2xl+px).  The products run
    // with the operands swapped (weights as the MFMA's row operand, voxels as its column operand: the two register layouts
    // are the same, so this costs nothing), which leaves lane l with VOXEL mb*32 + row_perm(l32) and registers
    // i -> cout 8 (i >> 2) + 4 khalf + (i & 3): four consecutive couts per register quad, one 16-byte store each.  Round 5
    // measured what the dword stores of the other orientation cost: 128 store instructions per thread were 0.12 of a 1.33 ms
    // launch, and the same stores into an L2-resident window cost the same -- the instructions, not the bytes.  The element
    // offset of a row's voxel relative to the box's comes from the table written at the top (the divisions of box_coords per
    // stored row were 4 900 instructions per wave before that, a sixth of the launch).
    const int H2 = 2 * p.h, W2 = 2 * p.w;
    float* const obase = p.slab ? p.slab + ((int64_t)smp * gridDim.y + blockIdx.y) * p.slab_stride : p.out + smp * p.sO;
    float* const ob = obase + (((int64_t)(2 * z0 + pz) * H2 + (2 * y0 + py)) * W2 + (2 * x0 + px)) * p.Cout + nt * 64 + khalf * 4;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int rel = rowoff[mb * 32 + row_perm(l32)];
        if (rel < 0) continue;                                // padding row, or outside the tensor
        float* const o = ob + (unsigned)rel;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(o + nb * 32 + g * 8) =
                    make_float4(acc[mb][nb][4 * g] * dq, acc[mb][nb][4 * g + 1] * dq, acc[mb][nb][4 * g + 2] * dq,
                                acc[mb][nb][4 * g + 3] * dq);
    }
}

template <int NPASS>
__global__ void __launch_bounds__(NTHR, 1) conv_upfold(const UpParams p) { conv_upfold_body<NPASS, 0>(p); }
template <int NPASS>
__global__ void __launch_bounds__(NTHR / 2, 2) conv_upfold_h(const UpParams p) { conv_upfold_body<NPASS, 0, 4>(p); }
#ifdef BFM_UP_ABLATE
template <int ABL>
__global__ void __launch_bounds__(NTHR, 1) conv_upfold_abl(const UpParams p) { conv_upfold_body<3, ABL>(p); }
#endif

// split-K: out = sum of the slabs in slab order (deterministic)
__global__ void upfold_reduce(const float4* __restrict__ slab, int nsplit, int64_t n4, float4* __restrict__ out) {
    slab += (int64_t)blockIdx.y * nsplit * n4;                  // blockIdx.y = sample
    out += (int64_t)blockIdx.y * n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = slab[i];
        for (int k = 1; k < nsplit; ++k) {
            const float4 b = slab[i + (int64_t)k * n4];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        out[i] = a;
    }
}

// packed[ntile64][class 8][kc][tap 8][nb 2][hl][lane] (uint4 = 8 halfs): lane l holds
// B[k = 8*(l>>5)+j][n = l&31] = Wfold_class[co = ntile*64 + nb*32 + (l&31)][ci = CA + kc*16 + 8*(l>>5) + j][tap] * 2^wexp
// where Wfold sums the original taps that land on the same low-res voxel (fixed order: kd, kh, kw ascending).
__global__ void pack_upfold(const float* __restrict__ w, int Cin, int CA, int CB, int Cout, int wexp, int npl,
                            uint4* __restrict__ out) {
    const int KCB = CB / KC;
    const int nf = 2 * npl;
    const int64_t n = (int64_t)(Cout / 64) * 8 * KCB * 8 * nf * 64;
    const float s = ldexpf(1.0f, wexp);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int f = (int)(r % nf); r /= nf;
        const int t = (int)(r & 7); r >>= 3;
        const int kc = (int)(r % KCB); r /= KCB;
        const int cls = (int)(r & 7); r >>= 3;
        const int ntile = (int)r;
        const int nb = f / npl, hl = f - nb * npl;
        const int co = ntile * 64 + nb * 32 + (lane & 31);
        const int ci0 = CA + kc * KC + 8 * (lane >> 5);
        const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
        const int ta = t >> 2, tb = (t >> 1) & 1, tc = t & 1;
        // original taps along one axis for (parity, folded tap): p=0: {0} | {1,2};  p=1: {0,1} | {2}
        const int zlo = pz == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2), zhi = pz == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
        const int ylo = py == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2), yhi = py == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
        const int xlo = px == 0 ? (tc == 0 ? 0 : 1) : (tc == 0 ? 0 : 2), xhi = px == 0 ? (tc == 0 ? 0 : 2) : (tc == 0 ? 1 : 2);
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* wc = w + ((size_t)co * Cin + ci0 + j) * 27;
            float sum = 0.f;
            for (int kd = zlo; kd <= zhi; ++kd)
                for (int kh = ylo; kh <= yhi; ++kh)
                    for (int kw = xlo; kw <= xhi; ++kw) sum += wc[kd * 9 + kh * 3 + kw];
            const float x = sum * s;
            const _Float16 hh = (_Float16)x;
            v[j] = hl == 0 ? hh : (_Float16)(x - (float)hh);
        }
        out[i] = __builtin_bit_cast(uint4, v);
    }
}

// low-res box of 128 voxels (powers of two) wasting the fewest rows on the volume's edges
// The same packing with the 64 (co) x 16 (ci) x 27 block of one (N tile, K chunk) staged through LDS: coalesced reads,
// every weight read once instead of once per class / folded tap / hi-lo plane that sums it (training re-packs every
// iteration: 1.5 ms for the 1024 + 512 -> 512 decoder layer with the gather version above).
constexpr int PKU_ROW = KC * 27 + 1;
__global__ void __launch_bounds__(256) pack_upfold_tiled(const float* __restrict__ w, int Cin, int CA, int CB, int Cout,
                                                         int wexp, int npl, uint4* __restrict__ out) {
    extern __shared__ float pku_lds[];                      // [64][PKU_ROW]
    const int KCB = CB / KC;
    const int nf = 2 * npl;
    const int kc = blockIdx.x % KCB, ntile = blockIdx.x / KCB;
    const float s = ldexpf(1.0f, wexp);
    const float* src0 = w + ((int64_t)(ntile * 64) * Cin + CA + kc * KC) * 27;
    bfm_stage_rows<64, KC * 27, PKU_ROW, 256>(src0, (int64_t)Cin * 27, pku_lds, 1.0f,
                                              ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && (Cin & 3) == 0 && (CA & 3) == 0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = wave; q < 8 * 8 * 2; q += 4) {              // (class, folded tap, column block): hi and lo from one sum
        const int nb = q & 1;
        const int t = (q >> 1) & 7;
        const int cls = q >> 4;
        const int pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
        const int ta = t >> 2, tb = (t >> 1) & 1, tc = t & 1;
        const int zlo = pz == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2), zhi = pz == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
        const int ylo = py == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2), yhi = py == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
        const int xlo = px == 0 ? (tc == 0 ? 0 : 1) : (tc == 0 ? 0 : 2), xhi = px == 0 ? (tc == 0 ? 0 : 2) : (tc == 0 ? 1 : 2);
        const float* src = pku_lds + (nb * 32 + (lane & 31)) * PKU_ROW + (8 * (lane >> 5)) * 27;
        half8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* wc = src + j * 27;
            float sum = 0.f;
            for (int kd = zlo; kd <= zhi; ++kd)
                for (int kh = ylo; kh <= yhi; ++kh)
                    for (int kw = xlo; kw <= xhi; ++kw) sum += wc[kd * 9 + kh * 3 + kw];
            const float x = sum * s;
            const _Float16 hh = (_Float16)x;
            vh[j] = hh;
            vl[j] = (_Float16)(x - (float)hh);
        }
        uint4* dst = out + ((((int64_t)(ntile * 8 + cls) * KCB + kc) * 8 + t) * nf + nb * npl) * 64 + lane;
        dst[0] = __builtin_bit_cast(uint4, vh);
        if (npl == 2) dst[64] = __builtin_bit_cast(uint4, vl);
    }
}

void choose_box(int d, int h, int w, int& BD, int& BH, int& BW) {
    // the last three: 125 / 100 of the 128 rows used, for the small decoder levels (5, 10, 20 voxels per axis), where
    // the power-of-two boxes waste up to three quarters of their rows on the edges; ties keep the earlier entries
    static const int opts[][3] = {{2, 4, 16}, {4, 2, 16}, {4, 4, 8}, {2, 8, 8}, {8, 2, 8}, {8, 4, 4}, {4, 8, 4},
                                  {1, 8, 16}, {8, 1, 16}, {2, 2, 32}, {1, 4, 32}, {4, 1, 32},
                                  {5, 5, 5}, {2, 5, 10}, {5, 2, 10}};
    int64_t best = -1;
    for (auto& o : opts) {
        if ((o[0] + 2) * (o[1] + 2) * (o[2] + 2) * 4 > 4 * NTHR) continue;
        int64_t cost = (int64_t)bfm_cdiv(d, o[0]) * bfm_cdiv(h, o[1]) * bfm_cdiv(w, o[2]);
        // tie-break towards long x-runs (coalesced staging, conflict-free A reads)
        cost = cost * 64 - (o[2] > 32 ? 32 : o[2]);
        if (best < 0 || cost < best) { best = cost; BD = o[0]; BH = o[1]; BW = o[2]; }
    }
}

}  // namespace

extern "C" size_t bfm_pack_conv_weights_upfold_bytes(int CB, int Cout, int passes) {
    if (CB <= 0 || Cout <= 0 || CB % KC || Cout % 64) return 0;
    const int npl = passes == 3 ? 2 : 1;
    return (size_t)(Cout / 64) * 8 * (CB / KC) * 8 * 2 * npl * 64 * sizeof(uint4);
}

extern "C" int bfm_pack_conv_weights_upfold(const float* w_oidhw, int CA, int CB, int Cout, float wmax_abs_host,
                                            int passes, void* wpacked, int* wexp_host, bfm_stream_t stream) {
    if (!w_oidhw || !wpacked || !wexp_host || CA < 0 || CB <= 0 || Cout <= 0) return BFM_E_ARG;
    if (CB % KC || Cout % 64 || (passes != 1 && passes != 3)) return BFM_E_SHAPE;
    int wexp = 0;
    if (wmax_abs_host > 0.f && wmax_abs_host < INFINITY) {
        int ex;
        (void)frexpf(8.f * wmax_abs_host, &ex);               // a folded weight sums up to 8 taps
        wexp = 14 - ex;
        wexp = wexp > 60 ? 60 : (wexp < -60 ? -60 : wexp);
    }
    *wexp_host = wexp;
    const int npl = passes == 3 ? 2 : 1;
    const int64_t nblk = (int64_t)(Cout / 64) * (CB / KC);
    const size_t smem = (size_t)64 * PKU_ROW * sizeof(float);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pack_upfold_tiled), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return BFM_E_LAUNCH;
        attr = true;
    }
    if (nblk <= 0x7fffffff) {
        hipLaunchKernelGGL(pack_upfold_tiled, dim3((unsigned)nblk), dim3(256), smem, bfm_s(stream), w_oidhw, CA + CB, CA, CB,
                           Cout, wexp, npl, static_cast<uint4*>(wpacked));
    } else {
        const int64_t n = nblk * 8 * 8 * 2 * npl * 64;
        int nb = (int)std::min<int64_t>(4096, bfm_cdiv64(n, 256));
        hipLaunchKernelGGL(pack_upfold, dim3(nb), dim3(256), 0, bfm_s(stream), w_oidhw, CA + CB, CA, CB, Cout, wexp, npl,
                           static_cast<uint4*>(wpacked));
    }
    return bfm_launch_status();
}

namespace {
// split K when the low-res level has too few boxes to fill the chip (the small decoder levels: 5^3 .. 20^3 voxels with
// 512-2048 low-res channels): chunks per split a multiple of 3 (the weight ring's phase period)
void upfold_split(int CB, int d, int h, int w, int Cout, int& nsplit, int& per) {
    int BD, BH, BW;
    choose_box(d, h, w, BD, BH, BW);
    const int64_t wgs = (int64_t)bfm_cdiv(d, BD) * bfm_cdiv(h, BH) * bfm_cdiv(w, BW) * (Cout / 64);
    const int KCB = CB / KC;
    nsplit = 1;
    per = KCB;
    if (wgs >= 192 || KCB < 6) return;
    int want = (int)std::min<int64_t>(KCB / 3, bfm_cdiv64(384, wgs));
    if (want < 2) return;
    per = bfm_cdiv(bfm_cdiv(KCB, want), 3) * 3;
    nsplit = bfm_cdiv(KCB, per);
    if (nsplit < 2) { nsplit = 1; per = KCB; }
}
}  // namespace

extern "C" size_t bfm_conv3x3x3_upfold_workspace(int CB, int d, int h, int w, int Cout) {
    if (CB <= 0 || d <= 0 || h <= 0 || w <= 0 || Cout <= 0 || CB % KC || Cout % 64) return 0;
    int nsplit, per;
    upfold_split(CB, d, h, w, Cout, nsplit, per);
    return nsplit > 1 ? (size_t)nsplit * 8 * d * h * w * Cout * sizeof(float) : 0;
}

static int upfold_launch(const float* B, int CB, int S, int d, int h, int w, const float* scale_b, const float* shift_b,
                         const float* bound, int G, const void* wpacked, int wexp, int Cout, int passes, float* out,
                         void* workspace, size_t workspace_bytes, bool strict, int affine_stride, bfm_stream_t stream);

extern "C" int bfm_conv3x3x3_upfold_ex(const float* B, int CB, int d, int h, int w, const float* scale_b,
                                       const float* shift_b, const float* bound, int G, const void* wpacked, int wexp,
                                       int Cout, int passes, float* out, void* workspace, size_t workspace_bytes,
                                       bfm_stream_t stream) {
    return upfold_launch(B, CB, 1, d, h, w, scale_b, shift_b, bound, G, wpacked, wexp, Cout, passes, out, workspace,
                         workspace_bytes, false, 0, stream);
}

extern "C" size_t bfm_conv3x3x3_upfold_batch_workspace(int CB, int S, int d, int h, int w, int Cout) {
    return S > 0 ? (size_t)S * bfm_conv3x3x3_upfold_workspace(CB, d, h, w, Cout) : 0;
}

extern "C" int bfm_conv3x3x3_upfold_batch(const float* B, int CB, int S, int d, int h, int w, const float* scale_b,
                                          const float* shift_b, const float* bound, int G, const void* wpacked, int wexp,
                                          int Cout, int passes, float* out, void* workspace, size_t workspace_bytes,
                                          int affine_stride, bfm_stream_t stream) {
    if (S <= 0 || S > 65535) return BFM_E_ARG;
    if (affine_stride != 0 && (affine_stride < CB || (affine_stride & 3))) return BFM_E_ARG;
    return upfold_launch(B, CB, S, d, h, w, scale_b, shift_b, bound, G, wpacked, wexp, Cout, passes, out, workspace,
                         workspace_bytes, true, affine_stride, stream);
}

// strict (the batch entry point): the split-K plan is a function of the per-sample shape alone, so a workspace too small
// for it is an error instead of a silent fall-back to one slab -- a sample's bits must not depend on the batch it is in
static int upfold_launch(const float* B, int CB, int S, int d, int h, int w, const float* scale_b, const float* shift_b,
                         const float* bound, int G, const void* wpacked, int wexp, int Cout, int passes, float* out,
                         void* workspace, size_t workspace_bytes, bool strict, int affine_stride, bfm_stream_t stream) {
    if (!B || CB <= 0 || d <= 0 || h <= 0 || w <= 0 || !scale_b || !shift_b || !bound || G <= 0 || !wpacked || !out)
        return BFM_E_ARG;
    if (CB % KC || Cout % 64 || Cout <= 0) return BFM_E_SHAPE;
    if (passes != 1 && passes != 3) return BFM_E_ARG;
    if ((reinterpret_cast<uintptr_t>(B) & 15) || (reinterpret_cast<uintptr_t>(scale_b) & 15) ||
        (reinterpret_cast<uintptr_t>(shift_b) & 15) || (reinterpret_cast<uintptr_t>(wpacked) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15))
        return BFM_E_ARG;
    if ((int64_t)d * h * w * CB > 0x7fffffffLL) return BFM_E_SHAPE;       // 32-bit staging offsets
    UpParams p{};
    p.B = B; p.CB = CB; p.d = d; p.h = h; p.w = w;
    p.scale = scale_b; p.shift = shift_b; p.bound = bound; p.G = G;
    p.saff = affine_stride > 0 ? affine_stride : CB;
    p.wp = static_cast<const uint4*>(wpacked);
    p.wexp = wexp; p.Cout = Cout; p.out = out;
    choose_box(d, h, w, p.BD, p.BH, p.BW);
    p.HT = p.BH + 2; p.WT = p.BW + 2;
    p.BHW = p.BH * p.BW; p.BVOX = p.BD * p.BHW;
    const int nTz = bfm_cdiv(d, p.BD);
    p.nTy = bfm_cdiv(h, p.BH); p.nTx = bfm_cdiv(w, p.BW);
    p.nMt = nTz * p.nTy * p.nTx;
    p.NT = Cout / 64;
    p.KCB = CB / KC;
    p.nvox_lds = (p.BD + 2) * p.HT * p.WT;
    if (p.nvox_lds * 4 > 4 * NTHR) return BFM_E_SHAPE;
    p.plane_stride = ((p.nvox_lds * 16 + 255) / 256) * 256 + 16;          // +16: planes start on different banks
    const int npl = passes == 3 ? 2 : 1;
    p.rowoff_lds = ((2 * npl * p.plane_stride + 64 + 15) / 16) * 16;
    const size_t smem = (size_t)p.rowoff_lds + 128 * sizeof(int);
    if ((int64_t)(2 * p.BD) * (2 * h) * (2 * w) * Cout > 0x7fffffffLL) return BFM_E_SHAPE;   // 32-bit row offsets inside a box
    if ((int64_t)p.nMt * p.NT > 0x7fffffff) return BFM_E_SHAPE;
    int nsplit = 1, per = p.KCB;
    const int64_t nout = (int64_t)8 * d * h * w * Cout;
    p.sB = (int64_t)d * h * w * CB;
    p.sO = nout;
    if (workspace || strict) {
        upfold_split(CB, d, h, w, Cout, nsplit, per);
        if (nsplit > 1 && (!workspace || workspace_bytes < (size_t)S * nsplit * nout * sizeof(float) ||
                           (reinterpret_cast<uintptr_t>(workspace) & 15))) {
            if (strict) return BFM_E_ARG;
            nsplit = 1;
            per = p.KCB;
        }
    }
    p.kc_per_split = per;
    p.slab = nsplit > 1 ? static_cast<float*>(workspace) : nullptr;
    p.slab_stride = nout;
    dim3 grid((unsigned)(p.nMt * p.NT), (unsigned)nsplit, (unsigned)S);
#ifdef BFM_UP_ABLATE
    if (const char* e = getenv("BFM_UP_SLEEP")) p.dbg_sleep = atoi(e);
    if (const char* e = getenv("BFM_UP_ABL")) {
        const int a = atoi(e);
        hipStream_t st = bfm_s(stream);
#define UPA(n) case n: hipLaunchKernelGGL(conv_upfold_abl<n>, grid, dim3(NTHR), smem, st, p); break;
        switch (a) {
            UPA(0) UPA(1) UPA(2) UPA(4) UPA(8) UPA(16) UPA(64) UPA(3) UPA(5) UPA(7) UPA(23) UPA(68) UPA(71) UPA(87) UPA(80) UPA(20) UPA(12)
            default: return BFM_E_ARG;
        }
#undef UPA
        return bfm_launch_status();
    }
#endif
    // Two half-box workgroups per CU (see conv_upfold_body) where full workgroups would leave half the CUs idle (<= 128 of
    // them) or where K is split (short workgroups: staging and epilogue are a large share of each): 7-12 % faster there
    // (20^3 256 -> 128, 126 workgroups: 0.136 -> 0.126 ms; 10^3 1024 -> 512, split 6: 0.277 -> 0.246; 5^3 2048 -> 1024, split
    // 22: 0.175 -> 0.159).  Slower where the chip is full and K is long -- 2 % at >= 500 workgroups, 5 % at 250 (in the flow:
    // decoders.3.1up on the 160 x 80 x 80 tiles 158 -> 167 us): the matrix pipe is power-limited, not phase-limited
    // (profiles/r06_conv_upfold_pmc.txt).  Same bits either way; BFM_UPFOLD_WAVES = 4 | 8 forces one form (diagnostics).
    static const int waves_env = [] { const char* e = getenv("BFM_UPFOLD_WAVES"); return e ? atoi(e) : 0; }();
    const int waves = waves_env ? waves_env : ((nsplit > 1 || (int64_t)p.nMt * p.NT * S <= 128) ? 4 : 8);
    if (waves == 4) {
        const int64_t nblk = (int64_t)p.nMt * p.NT;
        const int64_t gx = 16 * bfm_cdiv64(nblk, 8);
        if (gx > 0x7fffffff) return BFM_E_SHAPE;
        dim3 gridh((unsigned)gx, (unsigned)nsplit, (unsigned)S);
        if (passes == 3) hipLaunchKernelGGL(conv_upfold_h<3>, gridh, dim3(NTHR / 2), smem, bfm_s(stream), p);
        else hipLaunchKernelGGL(conv_upfold_h<1>, gridh, dim3(NTHR / 2), smem, bfm_s(stream), p);
    } else if (passes == 3) hipLaunchKernelGGL(conv_upfold<3>, grid, dim3(NTHR), smem, bfm_s(stream), p);
    else hipLaunchKernelGGL(conv_upfold<1>, grid, dim3(NTHR), smem, bfm_s(stream), p);
    if (nsplit > 1) {
        const int64_t n4 = nout / 4;
        const int nb = (int)std::min<int64_t>(4096, bfm_cdiv64(n4, 256));
        hipLaunchKernelGGL(upfold_reduce, dim3(nb, S), dim3(256), 0, bfm_s(stream), reinterpret_cast<const float4*>(p.slab), nsplit,
                           n4, reinterpret_cast<float4*>(out));
    }
    return bfm_launch_status();
}

extern "C" int bfm_conv3x3x3_upfold(const float* B, int CB, int d, int h, int w, const float* scale_b,
                                    const float* shift_b, const float* bound, int G, const void* wpacked, int wexp,
                                    int Cout, int passes, float* out, bfm_stream_t stream) {
    return bfm_conv3x3x3_upfold_ex(B, CB, d, h, w, scale_b, shift_b, bound, G, wpacked, wexp, Cout, passes, out, nullptr, 0,
                                   stream);
}
