// Fused network tail: one pass over the last decoder feature map.
//
//   F.normalize(feat, dim=1)                      unet3d/model.py:207-208
//   TaskHead: n_out x c_feat 1x1x1 conv + bias    head.py:52-59
//   SegProcessor softmax, DistProcessor clamp     joiner.py:69-77,149-157
//   get_postprocessor                             Trainer/models/__init__.py:272-354
//
// The reference reads the 64-channel feature map once per head (9x) plus once
// for normalize; here each voxel's 64 features are read once, and only what the
// caller asks for is written (15 fp32 maps + int64 label = 68 B/voxel; the
// 56-channel softmax and the normalised features are optional).
//
// 128 voxels per tile, persistent workgroups.  The [128][c_feat] feature tile
// is loaded with coalesced float4 reads into LDS (odd row stride: conflict-free
// column access); the head GEMM runs on the fp32 matrix core
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 FMA chain) with the weights held
// in registers as B fragments; the logits return to the LDS rows and one thread
// per voxel applies softmax / argmax / exp / tanh / clamp.
// raw_out != NULL turns the call into TaskHead.forward alone (raw logits,
// [nvox][n_out]) for callers that run the reference's processors separately.
#include "bfm_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int VPB = 128;     // voxels per tile
constexpr int TPBT = 128;    // threads per block: 2 waves, 64 voxels (two 32-row MFMA blocks) each
constexpr int CMAX = 64;     // c_feat upper bound
constexpr int OMAX = 96;     // n_out upper bound (3 MFMA column blocks, LDS row)
constexpr int LD = OMAX + 1; // LDS row stride in floats (odd: conflict-free column access)
constexpr int NMAPS = 64;    // upper bound on output maps

struct TailParams {
    const float* feat;
    const float* input;
    int64_t nvox;
    bfm_tail_desc_t d;
    float* feat_norm;
    float* const* maps;
    float* seg_prob;
    int64_t* label;
    float* raw_out;
    int n_maps;
};

// exp on the hardware exp2 unit: |rel err| ~ 1e-6 for the argument ranges below (softmax <= 0, tanh via exp)
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = fast_exp(2.f * fminf(fmaxf(x, -15.f), 15.f));
    return (e - 1.f) * __builtin_amdgcn_rcpf(e + 1.f);
}

__device__ __forceinline__ float fake_term(float w_or_p, float add, float gain) {
    // gain * (1 - (tanh(2*(v+add)) + 1) / 2)      (__init__.py:329-336, a = 2)
    return gain * (1.f - (fast_tanh(2.f * (w_or_p + add)) + 1.f) / 2.f);
}

// Persistent: each block walks tiles of 128 voxels.  Per tile:
//   1. coalesced float4 load of the [128][C] feature tile into LDS
//   2. one thread per voxel: L2 norm, normalise the row in place (F.normalize)
//   3. head GEMM [128 x C] x [C x n_out] on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain); the
//      weights live in registers as MFMA B fragments for the whole kernel
//   4. logits (+bias) back into the LDS rows, one thread per voxel applies the roles
__global__ void __launch_bounds__(TPBT) tail_kernel(TailParams p) {
    extern __shared__ float tile[];                 // [VPB][LD] then the per-head tables
    // roles / slots / output pointers are read once per block into LDS: fetching them from global memory inside
    // the per-head loop is a chain of dependent loads per output row (it dominated the first version)
    int* s_role = reinterpret_cast<int*>(tile + VPB * LD);            // [OMAX]
    int* s_slot = s_role + OMAX;                                       // [OMAX]
    float** s_map = reinterpret_cast<float**>(s_slot + OMAX);          // [NMAPS]
    int* s_lut = reinterpret_cast<int*>(s_map + NMAPS);                // [OMAX]
    const int C = p.d.c_feat;
    const int NO = p.d.n_out;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, lh = lane >> 5;
    const int nkk = C >> 1;                         // MFMA k-steps (2 channels each)
    const int nnb = (NO + 31) >> 5;                 // 32-wide output column blocks

    // ---- head weights as B fragments: B[k = lh][n = l32] of step kk, block nb = W[nb*32+l32][2kk+lh]
    float wfrag[3][CMAX / 2];
    float bias[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int o = nb * 32 + l32;
        bias[nb] = (nb < nnb && o < NO) ? p.d.head_b[o] : 0.f;
#pragma unroll
        for (int kk = 0; kk < CMAX / 2; ++kk)
            wfrag[nb][kk] = (nb < nnb && o < NO && kk < nkk) ? p.d.head_w[(size_t)o * C + 2 * kk + lh] : 0.f;
    }

    for (int i = threadIdx.x; i < p.d.n_out; i += TPBT) { s_role[i] = p.d.roles[i]; s_slot[i] = p.d.out_slot[i]; }
    for (int i = threadIdx.x; i < p.n_maps; i += TPBT) s_map[i] = p.maps[i];
    for (int i = threadIdx.x; i < p.d.n_seg; i += TPBT) s_lut[i] = p.d.seg_lut[i];
    __syncthreads();

    const int64_t ntiles = (p.nvox + VPB - 1) / VPB;
    for (int64_t tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
        const int64_t v0 = tix * VPB;
        const int nv = (int)min<int64_t>(VPB, p.nvox - v0);
        __syncthreads();                            // previous tile fully consumed
        {
            const int C4 = C >> 2;
            const float4* src = reinterpret_cast<const float4*>(p.feat + v0 * C);
            const int n4 = nv * C4;
            for (int i = t; i < n4; i += TPBT) {
                float4 q = src[i];
                int r = i / C4, c = (i - r * C4) * 4;
                float* dst = tile + r * LD + c;
                dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
            }
            // rows beyond nv: zero so the MFMA reads defined values
            for (int i = t + nv * C; i < VPB * C; i += TPBT) tile[(i / C) * LD + (i % C)] = 0.f;
        }
        __syncthreads();

        const bool live = t < nv;
        float* row = tile + t * LD;
        if (p.d.unit_feat) {
            float ss = 0.f;
            for (int c = 0; c < C; ++c) ss = fmaf(row[c], row[c], ss);
            const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps; one division per voxel
            for (int c = 0; c < C; ++c) row[c] = row[c] * inv;
        }
        __syncthreads();

        if (p.feat_norm) {                                    // optional: normalised features, coalesced
            const int C4 = C >> 2;
            float4* dst = reinterpret_cast<float4*>(p.feat_norm + v0 * C);
            const int n4 = nv * C4;
            for (int i = t; i < n4; i += TPBT) {
                int r = i / C4, c = (i - r * C4) * 4;
                const float* s = tile + r * LD + c;
                dst[i] = make_float4(s[0], s[1], s[2], s[3]);
            }
        }

        // ---- head GEMM: this wave owns voxel rows [64*wave, 64*wave+64)
        if (NO > 0) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int r0 = wave * 64 + mb * 32;
                const float* arow = tile + (r0 + l32) * LD + lh;
                floatx16 acc[3];
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
#pragma unroll
                for (int kk = 0; kk < CMAX / 2; ++kk) {
                    if (kk < nkk) {
                        const float a = arow[2 * kk];
#pragma unroll
                        for (int nb = 0; nb < 3; ++nb)
                            if (nb < nnb)
                                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wfrag[nb][kk], acc[nb], 0, 0, 0);
                    }
                }
                // the wave's own rows: all A reads above are consumed before these writes (data dependence)
#pragma unroll
                for (int nb = 0; nb < 3; ++nb) {
                    if (nb < nnb) {
                        const int o = nb * 32 + l32;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int rr = (i & 3) + 8 * (i >> 2) + 4 * lh;
                            if (o < NO) tile[(r0 + rr) * LD + o] = acc[nb][i] + bias[nb];
                        }
                    }
                }
            }
        }
        __syncthreads();

        if (p.raw_out) {                                      // TaskHead.forward only: raw logits, channels-last
            const int n = nv * NO;
            float* dst = p.raw_out + v0 * NO;
            for (int i = t; i < n; i += TPBT) {
                int r = i / NO, o = i - r * NO;
                dst[i] = tile[r * LD + o];
            }
            continue;
        }

        const int64_t v = v0 + t;
        float dist[4] = {0.f, 0.f, 0.f, 0.f};
        // ---- processors + post-processor per role
        for (int o = 0; o < NO; ++o) {
            const int role = s_role[o];
            if (role == BFM_ROLE_SEG) continue;
            const int slot = s_slot[o];
            const float a = row[o];
            float r = a;
            if (role == BFM_ROLE_CT) r = a * 1000.f;
            else if (role == BFM_ROLE_BIAS_LOG) r = expf(a);
            else if (role == BFM_ROLE_PATHOL) r = 1.f / (1.f + expf(-a));
            else if (role == BFM_ROLE_DIST) {
                r = fminf(fmaxf(a, -p.d.max_dist), p.d.max_dist);
                const int k = o - p.d.dist_first;
                if (k == 0) dist[0] = r; else if (k == 1) dist[1] = r; else if (k == 2) dist[2] = r; else dist[3] = r;
            }
            if (live && slot >= 0) s_map[slot][v] = r;
            if (role == BFM_ROLE_SR && p.d.slot_high_res >= 0 && live && p.input)
                s_map[p.d.slot_high_res][v] = a + p.input[v];
        }

        if (p.d.n_dist > 0 && p.d.slot_fake_cortical >= 0 && live) {
            // order lp, lw[, rp, rw]  (__init__.py:321-337)
            float fake = fake_term(dist[1], 0.3f, 70.f) + fake_term(dist[0], 0.f, 40.f);
            if (p.d.n_dist == 4) fake = fake + (fake_term(dist[3], 0.3f, 70.f) + fake_term(dist[2], 0.f, 40.f));
            s_map[p.d.slot_fake_cortical][v] = fake;
        }

        if (p.d.n_seg > 0) {
            const int ns = p.d.n_seg;
            float* sl = row + p.d.seg_first;
            int best = 0;
            if (live) {
                float m = -INFINITY;
                for (int s = 0; s < ns; ++s) m = fmaxf(m, sl[s]);
                float sum = 0.f;
                for (int s = 0; s < ns; ++s) { float e = fast_exp(sl[s] - m); sl[s] = e; sum += e; }
                const float rs = 1.f / sum;
                float bp = -1.f;
                for (int s = 0; s < ns; ++s) {
                    float pr = sl[s] * rs;
                    sl[s] = pr;
                    if (pr > bp) { bp = pr; best = s; }      // first maximum wins (torch.argmax)
                }
                if (p.label) p.label[v] = (int64_t)s_lut[best];
            }
            if (p.seg_prob) {
                __syncthreads();
                const int n = nv * ns;
                float* dst = p.seg_prob + v0 * ns;
                for (int i = t; i < n; i += TPBT) {
                    int r = i / ns, s = i - r * ns;
                    dst[i] = tile[r * LD + p.d.seg_first + s];
                }
            }
        }
    }
}

}  // namespace

extern "C" int bfm_tail_heads(const float* feat, const float* input, int64_t nvox, const bfm_tail_desc_t* desc,
                              float* feat_norm, float* const* maps, float* seg_prob, int64_t* label,
                              float* raw_out, bfm_stream_t stream) {
    if (!feat || !desc || nvox <= 0 || !desc->head_w || !desc->head_b || !desc->roles || !desc->out_slot)
        return BFM_E_ARG;
    if (!maps && !raw_out && !feat_norm) return BFM_E_ARG;
    if (desc->n_out < 0 || desc->n_out > OMAX || (desc->n_out > 0 && !maps && !raw_out)) return BFM_E_SHAPE;
    if (desc->n_seg > 0 && (desc->seg_first < 0 || desc->seg_first + desc->n_seg > desc->n_out)) return BFM_E_SHAPE;
    if (desc->c_feat <= 0 || desc->c_feat > CMAX || desc->c_feat % 4 != 0) return BFM_E_SHAPE;
    if (desc->n_seg < 0 || (desc->n_seg > 0 && !desc->seg_lut)) return BFM_E_SHAPE;
    if (desc->n_dist != 0 && desc->n_dist != 2 && desc->n_dist != 4) return BFM_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(feat) & 15 || (feat_norm && (reinterpret_cast<uintptr_t>(feat_norm) & 15)))
        return BFM_E_ARG;
    int n_maps = maps ? desc->n_maps : 0;
    if (n_maps < 0 || n_maps > NMAPS) return BFM_E_SHAPE;
    for (int o = 0; o < desc->n_out; ++o) (void)o;
    if (maps && (desc->slot_high_res >= n_maps || desc->slot_fake_cortical >= n_maps)) return BFM_E_SHAPE;
    TailParams p{feat, input, nvox, *desc, feat_norm, maps, seg_prob, label, raw_out, n_maps};
    int64_t nb = bfm_cdiv64(nvox, VPB);
    if (nb > 256 * 6) nb = 256 * 6;                  // persistent: the weight fragments are loaded once per block
    const size_t smem = (size_t)VPB * LD * sizeof(float) + (size_t)3 * OMAX * sizeof(int) + (size_t)NMAPS * sizeof(float*);
    hipLaunchKernelGGL(tail_kernel, dim3((unsigned)nb), dim3(TPBT), smem, bfm_s(stream), p);
    return bfm_launch_status();
}
