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
// 128 voxels per workgroup.  The [128][c_feat] feature tile is loaded with
// coalesced float4 reads into LDS (row stride c_feat+1 floats: conflict-free
// column access), each thread then owns one voxel.  Head weights are
// wave-uniform (scalar loads).  The LDS row is reused for the head logits.
// raw_out != NULL turns the call into TaskHead.forward alone (raw logits,
// [nvox][n_out]) for callers that run the reference's processors separately.
#include "bfm_common.h"

namespace {

constexpr int VPB = 128;     // voxels (= threads) per block
constexpr int CMAX = 64;     // c_feat upper bound (registers)
constexpr int OMAX = 96;     // n_out upper bound (LDS row)

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
};

__device__ __forceinline__ float fake_term(float w_or_p, float add, float gain) {
    // gain * (1 - (tanh(2*(v+add)) + 1) / 2)      (__init__.py:329-336, a = 2)
    return gain * (1.f - (tanhf(2.f * (w_or_p + add)) + 1.f) / 2.f);
}

__global__ void __launch_bounds__(VPB) tail_kernel(TailParams p) {
    extern __shared__ float tile[];                 // [VPB][LD]
    const int C = p.d.c_feat;
    const int LD = OMAX + 1;
    const int t = threadIdx.x;
    const int64_t v0 = (int64_t)blockIdx.x * VPB;
    const int nv = (int)min<int64_t>(VPB, p.nvox - v0);

    // ---- coalesced load of the [nv][C] tile
    {
        const int C4 = C >> 2;
        const float4* src = reinterpret_cast<const float4*>(p.feat + v0 * C);
        const int n4 = nv * C4;
        for (int i = t; i < n4; i += VPB) {
            float4 q = src[i];
            int r = i / C4, c = (i - r * C4) * 4;
            float* dst = tile + r * LD + c;
            dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
        }
    }
    __syncthreads();

    const bool live = t < nv;
    float f[CMAX];
    float* row = tile + t * LD;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) f[c] = (live && c < C) ? row[c] : 0.f;

    if (p.d.unit_feat) {
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) ss = fmaf(f[c], f[c], ss);
        const float denom = fmaxf(sqrtf(ss), 1e-12f);     // F.normalize eps
#pragma unroll
        for (int c = 0; c < CMAX; ++c) f[c] = f[c] / denom;
    }

    if (p.feat_norm) {                                    // optional: write normalised features, coalesced
        if (live) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c) if (c < C) row[c] = f[c];
        }
        __syncthreads();
        const int C4 = C >> 2;
        float4* dst = reinterpret_cast<float4*>(p.feat_norm + v0 * C);
        const int n4 = nv * C4;
        for (int i = t; i < n4; i += VPB) {
            int r = i / C4, c = (i - r * C4) * 4;
            const float* s = tile + r * LD + c;
            dst[i] = make_float4(s[0], s[1], s[2], s[3]);
        }
        __syncthreads();
    }

    const int64_t v = v0 + t;
    float dist[4] = {0.f, 0.f, 0.f, 0.f};

    // ---- phase 1: all head logits into this thread's LDS row (features now live in registers)
    for (int o = 0; o < p.d.n_out; ++o) {
        const float* w = p.d.head_w + (size_t)o * C;      // wave-uniform -> scalar loads
        float a = p.d.head_b[o];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) a = fmaf(w[c], f[c], a);
        row[o] = a;
    }

    if (p.raw_out) {                                      // TaskHead.forward only: raw logits, channels-last
        __syncthreads();
        const int no = p.d.n_out;
        const int n = nv * no;
        float* dst = p.raw_out + v0 * no;
        for (int i = t; i < n; i += VPB) {
            int r = i / no, o = i - r * no;
            dst[i] = tile[r * LD + o];
        }
        return;
    }

    // ---- phase 2: processors + post-processor per role
    for (int o = 0; o < p.d.n_out; ++o) {
        const int role = p.d.roles[o];
        if (role == BFM_ROLE_SEG) continue;
        const int slot = p.d.out_slot[o];
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
        if (live && slot >= 0) p.maps[slot][v] = r;
        if (role == BFM_ROLE_SR && p.d.slot_high_res >= 0 && live && p.input)
            p.maps[p.d.slot_high_res][v] = a + p.input[v];
    }

    if (p.d.n_dist > 0 && p.d.slot_fake_cortical >= 0 && live) {
        // order lp, lw[, rp, rw]  (__init__.py:321-337)
        float fake = fake_term(dist[1], 0.3f, 70.f) + fake_term(dist[0], 0.f, 40.f);
        if (p.d.n_dist == 4) fake = fake + (fake_term(dist[3], 0.3f, 70.f) + fake_term(dist[2], 0.f, 40.f));
        p.maps[p.d.slot_fake_cortical][v] = fake;
    }

    if (p.d.n_seg > 0) {
        const int ns = p.d.n_seg;
        float* sl = row + p.d.seg_first;
        int best = 0;
        if (live) {
            float m = -INFINITY;
            for (int s = 0; s < ns; ++s) m = fmaxf(m, sl[s]);
            float sum = 0.f;
            for (int s = 0; s < ns; ++s) { float e = expf(sl[s] - m); sl[s] = e; sum += e; }
            float bp = -1.f;
            for (int s = 0; s < ns; ++s) {
                float pr = sl[s] / sum;
                sl[s] = pr;
                if (pr > bp) { bp = pr; best = s; }      // first maximum wins (torch.argmax)
            }
            if (p.label) p.label[v] = (int64_t)p.d.seg_lut[best];
        }
        if (p.seg_prob) {
            __syncthreads();
            const int n = nv * ns;
            float* dst = p.seg_prob + v0 * ns;
            for (int i = t; i < n; i += VPB) {
                int r = i / ns, s = i - r * ns;
                dst[i] = tile[r * LD + p.d.seg_first + s];
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
    TailParams p{feat, input, nvox, *desc, feat_norm, maps, seg_prob, label, raw_out};
    const int64_t nb = bfm_cdiv64(nvox, VPB);
    if (nb > 0x7fffffff) return BFM_E_SHAPE;
    const size_t smem = (size_t)VPB * (OMAX + 1) * sizeof(float);
    hipLaunchKernelGGL(tail_kernel, dim3((unsigned)nb), dim3(VPB), smem, bfm_s(stream), p);
    return bfm_launch_status();
}
