/*
 * brainfm_hip.h -- C ABI of libbrainfm_hip.so (gfx950 / MI355X).
 *
 * The reference (jhuldr/BrainFM) is pure Python on stock torch ops: it has no
 * FFI of its own.  This header is therefore the boundary a maintainer would
 * bind with ctypes; every entry point cites the reference
 * call chain it replaces.  Conventions:
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - activations are fp32, channels-last-3D: index = ((z*H + y)*W + x)*C + c
 *     (the reference's NCDHW tensors viewed with torch.channels_last_3d strides);
 *   - no allocation, no global state, no host synchronisation inside a call:
 *     the caller owns every buffer (including workspaces) and the stream;
 *   - return value: 0 = launched, <0 = rejected before launch (BFM_E_*).
 */
#ifndef BRAINFM_HIP_H
#define BRAINFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bfm_stream_t; /* hipStream_t */

#define BFM_OK 0
#define BFM_E_ARG (-1)      /* null pointer / non-positive size */
#define BFM_E_SHAPE (-2)    /* shape not supported by this kernel family */
#define BFM_E_WORKSPACE (-3)/* workspace too small */
#define BFM_E_LAUNCH (-4)   /* hipGetLastError() != hipSuccess after launch */

/* Nearest-neighbour upsampling source (F.interpolate(mode='nearest'),
 * Trainer/models/unet3d/buildingblocks.py:361-363): low-res dims and per-axis
 * index maps map?[dst] = min(floor(dst*in/out), in-1) plus replication counts
 * rep?[src] = #{dst : map[dst]==src}.  All six arrays live on the device. */
typedef struct {
    int d, h, w;
    const int32_t* mapD; const int32_t* mapH; const int32_t* mapW; /* len D,H,W of the hi-res grid */
    const int32_t* repD; const int32_t* repH; const int32_t* repW; /* len d,h,w */
} bfm_upsample_t;

/* Library / build info: "brainfm_hip <version> gfx950". */
const char* bfm_version(void);

/* ---------------------------------------------------------------- GroupNorm
 * nn.GroupNorm(G, C, eps) statistics (buildingblocks.py:48-60) over the
 * virtual concatenation cat((A, nearest_up(B)), channel) of Decoder._joining
 * (buildingblocks.py:265-276).  CB==0 / B==NULL for a plain tensor.
 * Outputs the folded affine y = x*scale[c] + shift[c]
 * (scale = gamma*rstd, shift = beta - mean*scale) and, per group, the largest
 * |y| the folded affine can produce (bound[g]) which the MFMA convolution
 * uses to pick a power-of-two operand scale.
 * workspace: bfm_gn_stats_workspace(...) bytes. */
size_t bfm_gn_stats_workspace(int CA, int CB, int D, int H, int W, const bfm_upsample_t* up);
int bfm_gn_stats(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                 const bfm_upsample_t* up, const float* gamma, const float* beta, int G, float eps,
                 float* scale, float* shift, float* bound, void* workspace, size_t workspace_bytes,
                 bfm_stream_t stream);
/* same, also returning the per-group mean and 1/sqrt(var+eps) (G floats each) that the backward pass needs */
int bfm_gn_stats_train(const float* A, int CA, const float* B, int CB, int D, int H, int W, const bfm_upsample_t* up,
                       const float* gamma, const float* beta, int G, float eps, float* scale, float* shift,
                       float* bound, float* mean_out, float* rstd_out, void* workspace, size_t workspace_bytes,
                       bfm_stream_t stream);

/* ------------------------------------------------------ 3x3x3 convolution
 * SingleConv 'gcl' minus the statistics: y = LeakyReLU_slope( conv3d_p1(
 *   x*scale + shift ) )  with x = cat((A, nearest_up(B))).  Zero padding is
 * applied AFTER the affine, as nn.Conv3d pads GroupNorm's output
 * (buildingblocks.py:31-60).  Weights: see bfm_pack_conv_weights_*.
 *
 * _direct: exact fp32 FMA, any Cin/Cout (stem Cin=1, odd widths).
 *   wpacked layout [27][Cin][Cout] fp32.
 * _mfma: implicit GEMM on v_mfma_f32_32x32x16_f16.  Requires CA%16==0,
 *   CB%16==0, Cout%64==0.  passes=3 splits both operands into f16 hi+lo
 *   (hi*hi + hi*lo + lo*hi, fp32 accumulate: ~2^-22 relative product error);
 *   passes=1 uses hi only (fast mode, ~2^-11).  `bound` = G per-group bounds
 *   from bfm_gn_stats (operand scale is derived from them on the device).
 *   splitk>1 needs workspace of splitk*nvox*Cout*4 bytes. */
size_t bfm_pack_conv_weights_direct_bytes(int Cin, int Cout);
int bfm_pack_conv_weights_direct(const float* w_oidhw, int Cin, int Cout, float* wpacked, bfm_stream_t stream);
size_t bfm_pack_conv_weights_mfma_bytes(int Cin, int Cout);
/* wmax_abs_host = max|w| (host scalar); returns the power-of-two weight scale exponent in *wexp_host */
int bfm_pack_conv_weights_mfma(const float* w_oidhw, int Cin, int Cout, float wmax_abs_host, void* wpacked,
                               int* wexp_host, bfm_stream_t stream);

/* Layout for the v_mfma_f32_16x16x32_f16 kernel variant (plan cfg[6] == 2): K = 32 is a pair of taps x 16
 * channels.  A layer planned with cfg[6] == 2 must be given weights packed by this function; cfg[6] in {0,1}
 * takes bfm_pack_conv_weights_mfma's layout. */
size_t bfm_pack_conv_weights_mfma16_bytes(int Cin, int Cout);
int bfm_pack_conv_weights_mfma16(const float* w_oidhw, int Cin, int Cout, float wmax_abs_host, void* wpacked,
                                 int* wexp_host, bfm_stream_t stream);

int bfm_conv3x3x3_direct(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                         const bfm_upsample_t* up, const float* scale, const float* shift,
                         const float* wpacked, int Cout, float slope, float* out, bfm_stream_t stream);

/* Cin == 1 stem (enc0.1) as a K = 32 (27 taps + padding) GEMM on the matrix core, same split-fp16 numerics as
 * _mfma; Cout in {32, 64}; weights in the _direct layout [27][Cout]; bound = the single GroupNorm bound. */
int bfm_conv3x3x3_stem(const float* A, int D, int H, int W, const float* scale, const float* shift,
                       const float* bound, const float* wpacked_direct, int Cout, float slope, float* out,
                       bfm_stream_t stream);

/* Winograd F(2,3)-along-x variant of the single-source 3x3x3 conv (SingleConv 'gcl', buildingblocks.py:31-60):
 * 4 transformed positions x 9 (kd,kh) taps for every pair of x-neighbouring outputs = 1.5x fewer matrix-core FLOPs
 * than bfm_conv3x3x3_mfma, same split-fp16 products, fp32 transforms (fp32-grade, not bit-identical to _mfma).
 * No second source, no split-K; `accumulate` as cfg[7] bit 0 of _mfma.  Weights packed by _pack_conv_weights_wino. */
size_t bfm_pack_conv_weights_wino_bytes(int Cin, int Cout, int passes);
int bfm_pack_conv_weights_wino(const float* w_oidhw, int Cin, int Cout, float wmax_abs_host, int passes, void* wpacked,
                               int* wexp_host, bfm_stream_t stream);
int bfm_conv3x3x3_wino(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                       const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                       int flags /* bit 0: accumulate onto out; bit 1: wave-specialised persistent kernel; bit 2: 8-wave
                                    software-pipelined kernel */,
                       float* out, bfm_stream_t stream);

int bfm_conv3x3x3_wino_rows(int D, int H, int W, int passes);   /* moment rows of the 4-wave kernel (0: cannot run) */
int bfm_conv3x3x3_wino_ex(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                          const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                          int flags, float* out, void* moment_rows /*or NULL;
 4-wave kernel only*/, bfm_stream_t stream);

/* The last convolution of a tile inside the tile loop (scripts/demo_test.py:88-100 keeps `v * (tile_input != 0)` of every
 * output): a box of output voxels whose tile-input voxels are all zero feeds nothing that survives the mask, so it is not
 * computed (`out` keeps whatever it held there).  mask_image = the tile's input (D,H,W).  The 4-wave kernel, no moment rows. */
/* Boxes that multiply the same operands as other boxes.  Where the network's one-channel input image is bitwise constant
 * (the zero background of a skull-stripped head, utils/test_utils.py:235-284 min-max normalises it to exact zeros) the
 * activations of the first layers depend on the distances to the tile's faces only (zero padding): stem output where the
 * image is constant within 1 voxel, encoders.0 second conv within 2, and so on.  bfm_uniform_boxes flags the boxes of the
 * 4-wave Winograd kernel's grid over which `image` is constant within `radius` voxels (clipped to the volume) with
 * 1 + class, class = 9 cz + 3 cy + cx, c = 0 / 1 / 2 for the first / a middle / the last box along the axis: all boxes of
 * a class see the faces the same way (a box side must be >= radius: BFM_E_SHAPE otherwise).  flags:
 * bfm_uniform_boxes_bytes() bytes, 4-byte aligned: one byte per box, then the first box of each of the 27 classes.
 * bfm_conv3x3x3_wino_uniform is bfm_conv3x3x3_wino_ex that computes the unflagged boxes and the first flagged box of each
 * class in full, and gives every other flagged box its class's accumulators (the bits its own main loop would produce: same
 * operands, same order) before the normal epilogue -- no staging, weights or matrix products there.  scratch:
 * bfm_conv3x3x3_wino_uniform_scratch(Cout) bytes, 16-byte aligned.  The caller passes radius = (number of 3x3x3
 * convolutions between the image and this layer's OUTPUT): 2 for encoders.0's second conv, 3 for the skip half of the last
 * decoder's first conv. */
size_t bfm_uniform_boxes_bytes(int D, int H, int W, int passes);
int bfm_uniform_boxes(const float* image, int D, int H, int W, int radius, int passes, unsigned char* flags,
                      bfm_stream_t stream);
/* the same for a layer `level` MaxPool3d(2) steps down: the box grid is that of the (D >> level, ...) tensor, a box covers
 * image voxels [z0 << level, (z0 + TD) << level), and radius is in IMAGE voxels: 2 * 2^0 for the two level-0 convolutions
 * plus 2^level per convolution at the layer's own level up to its output (4, 6, 8 for the three level-1 layers that read the
 * first activations: encoders.1 conv 1 / conv 2 and the skip half of the second-to-last decoder's conv 1).  flags:
 * bfm_uniform_boxes_bytes(D >> level, H >> level, W >> level, passes) bytes. */
int bfm_uniform_boxes_level(const float* image, int D, int H, int W, int level, int radius, int passes, unsigned char* flags,
                            bfm_stream_t stream);
size_t bfm_conv3x3x3_wino_uniform_scratch(int Cout);
int bfm_conv3x3x3_wino_uniform(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                               const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                               int flags, float* out, void* moment_rows, const unsigned char* uniform_flags, void* scratch,
                               bfm_stream_t stream);
/* SingleConv followed by nn.MaxPool3d(2) (Encoder.forward, buildingblocks.py:185-186, 211-214: the pooling of the next
 * encoder level reads this layer's output) in one pass: out and moment_rows as bfm_conv3x3x3_wino_ex / _wino_uniform write
 * them (the skip connection keeps the full-resolution tensor), pooled (D/2,H/2,W/2,Cout) = the bits of bfm_maxpool2(out), and
 * pooled_rows (or NULL) = that tensor's moment rows, [bfm_conv3x3x3_wino_rows()][Cout].  The 2 x 2 x 2 windows must be whole
 * inside one box of the kernel: bfm_conv3x3x3_wino_pool_ok() says whether a (D,H,W) tensor qualifies (its box is 8 x 8 x 4
 * and tiles it exactly -- the 80- and 160-voxel tiles of the reference tiling at levels 0 and 1 do); BFM_E_SHAPE otherwise,
 * nothing launched: pool with bfm_maxpool2_ex then. */
int bfm_conv3x3x3_wino_pool_ok(int D, int H, int W, int passes);
int bfm_conv3x3x3_wino_pool(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                            const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                            int flags, float* out, void* moment_rows, float* pooled, void* pooled_rows /*or NULL*/,
                            bfm_stream_t stream);
int bfm_conv3x3x3_wino_uniform_pool(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                                    const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope,
                                    int passes, int flags, float* out, void* moment_rows,
                                    const unsigned char* uniform_flags, void* scratch, float* pooled,
                                    void* pooled_rows /*or NULL*/, bfm_stream_t stream);
int bfm_conv3x3x3_wino_box(int D, int H, int W, int passes, int* box /* [3]: the (d,h,w) box of output voxels per workgroup */);
/* workspace: bfm_conv3x3x3_wino_masked_workspace(D, H, W, passes) bytes (4-byte aligned) for the per-box activity, the
 * number of boxes that hold input and their list, built on the device ahead of the persistent launch that walks it. */
size_t bfm_conv3x3x3_wino_masked_workspace(int D, int H, int W, int passes);
int bfm_conv3x3x3_wino_masked(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                              const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                              int flags, float* out, const float* mask_image, void* workspace, size_t workspace_bytes,
                              bfm_stream_t stream);

/* Output-moment rows.  A producer (conv3x3x3_mfma_ex / conv3x3x3_stem_ex) can write, next to its output, one row
 * per tile of per-channel {sum, sumsq} (fp64) and {min, max} (fp32) of the values it stored: buffer of
 * bfm_moment_rows_bytes(nrows, Cout) bytes laid out sum[nrows][C] | sumsq[nrows][C] | min[nrows][C] | max[nrows][C].
 * The consumer's GroupNorm (nn.GroupNorm in the next SingleConv, buildingblocks.py:48-60) then reduces the rows with
 * bfm_gn_stats_rows instead of re-reading the activation.  Deterministic: fixed reduction order, no atomics.
 * bfm_conv3x3x3_mfma_rows returns 0 when the plan cannot emit rows (split-K or the persistent variant). */
size_t bfm_moment_rows_bytes(int nrows, int C);
int bfm_conv3x3x3_mfma_rows(int Cin, int Cout, int D, int H, int W, const int* cfg);
int bfm_conv3x3x3_mfma_ex(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                          const bfm_upsample_t* up, const float* scale, const float* shift, const float* bound,
                          int G, const void* wpacked, int wexp, int Cout, float slope, int passes, const int* cfg,
                          float* out, void* workspace, size_t workspace_bytes, void* moment_rows /*or NULL*/,
                          bfm_stream_t stream);
/* A batch of S same-shape samples (the tiles of one volume that share a shape) through ONE launch: A [S][D,H,W,CA],
 * B [S][d,h,w,CB], out [S][D,H,W,Cout], scale / shift [S][CA+CB], bound [S][G] -- GroupNorm statistics stay per
 * sample (buildingblocks.py:48-60).  Every workgroup does what it does in the one-sample launch (same box, K order and
 * split-K), so the result is bit-identical to S calls of bfm_conv3x3x3_mfma_ex; what changes is that the S workgroups
 * needing the same packed-weight fragments run side by side and read them from HBM once (the deep levels are bound by
 * their 1 GB of weights per tile).  Plan variants 0 and 2 only; moment_rows: [S * rows][Cout].
 * affine_stride: floats between two samples' scale / shift rows (0 = CA + CB; wider when the caller hands a column
 * window of a wider [S][C] table, e.g. the skip half of a decoder concat). */
size_t bfm_conv3x3x3_mfma_batch_workspace(int Cin, int Cout, int S, int D, int H, int W, int splitk);
int bfm_conv3x3x3_mfma_batch(const float* A, int CA, const float* B, int CB, int S, int D, int H, int W,
                             const bfm_upsample_t* up, const float* scale, const float* shift, const float* bound, int G,
                             const void* wpacked, int wexp, int Cout, float slope, int passes, const int* cfg, float* out,
                             void* workspace, size_t workspace_bytes, void* moment_rows, int affine_stride,
                             bfm_stream_t stream);
/* GroupNorm statistics of such a batch: per sample exactly bfm_gn_stats / bfm_gn_stats_rows (same block split, same
 * summation order), outputs [S][C] / [S][G]. */
size_t bfm_gn_stats_batch_workspace(int CA, int CB, int S, int D, int H, int W, const bfm_upsample_t* up);
int bfm_gn_stats_batch(const float* A, int CA, const float* B, int CB, int S, int D, int H, int W,
                       const bfm_upsample_t* up, const float* gamma, const float* beta, int G, float eps, float* scale,
                       float* shift, float* bound, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
int bfm_gn_stats_rows_batch(const void* rowsA, int nrowsA_per_sample, int CA, const void* rowsB, int nrowsB_per_sample,
                            int CB, double weightB, int64_t nvox, int S, const float* gamma, const float* beta, int G,
                            float eps, float* scale, float* shift, float* bound, bfm_stream_t stream);

int bfm_maxpool2_rows(int C, int D, int H, int W);
int bfm_maxpool2_ex(const float* in, int C, int D, int H, int W, float* out, void* moment_rows /*or NULL*/,
                    bfm_stream_t stream);
/* nn.MaxPool3d(2) (buildingblocks.py:185-186) of S same-shape samples in one launch (the batched levels); moment rows
 * [S * bfm_maxpool2_batch_rows()][C] (at most 128 per sample: bfm_gn_stats_rows_batch reads them directly).  A sample's
 * output and rows do not depend on S.  C % 4 == 0. */
int bfm_maxpool2_batch_rows(int C, int D, int H, int W);
int bfm_maxpool2_batch(const float* in, int C, int S, int D, int H, int W, float* out, void* moment_rows /*or NULL*/,
                       bfm_stream_t stream);
int bfm_conv3x3x3_stem_rows(int D, int H, int W);
int bfm_conv3x3x3_stem_ex(const float* A, int D, int H, int W, const float* scale, const float* shift,
                          const float* bound, const float* wpacked_direct, int Cout, float slope, float* out,
                          void* moment_rows /*or NULL*/, bfm_stream_t stream);
/* GroupNorm scale/shift/bound from moment rows.  Source A: rowsA [nrowsA][CA]; optional source B (the low-res half of
 * a decoder concat, every voxel replicated weightB times by the nearest upsample: 8 for an exact 2x): rowsB.
 * nvox = voxels per channel of the normalised tensor (D*H*W of the full-res grid). */
size_t bfm_gn_stats_rows_workspace(int nrowsA, int CA, int nrowsB, int CB);
/* ticket: NULL, or BFM_GN_TICKETS int32 in device memory that are ZERO before the first call and are left zero by every
 * call (a buffer of its own: nothing else may write it; one per stream that may run these calls concurrently).  With
 * tickets, a table of more than 128 rows and G <= BFM_GN_TICKETS the whole thing is ONE launch: workgroup (group, row
 * slice) folds its slice for the group's channels, the last workgroup of a group to finish runs the group's finalize.
 * Without, it is up to three launches (a fold per large table, then the finalize).  Both forms are deterministic (fixed
 * summation orders); they differ from each other in the last bits of the fp64 sums. */
#define BFM_GN_TICKETS 16
int bfm_gn_stats_rows(const void* rowsA, int nrowsA, int CA, const void* rowsB, int nrowsB, int CB, double weightB,
                      int64_t nvox, const float* gamma, const float* beta, int G, float eps, float* scale,
                      float* shift, float* bound, void* workspace, size_t workspace_bytes, void* ticket,
                      bfm_stream_t stream);
/* the same with the per-group mean / rstd kept for the backward pass (NULL: not wanted) */
int bfm_gn_stats_rows_train(const void* rowsA, int nrowsA, int CA, const void* rowsB, int nrowsB, int CB, double weightB,
                            int64_t nvox, const float* gamma, const float* beta, int G, float eps, float* scale,
                            float* shift, float* bound, float* mean_out, float* rstd_out, void* workspace,
                            size_t workspace_bytes, void* ticket, bfm_stream_t stream);

/* the same with each table given as rows [first, first + nrows) of a larger table of `total` rows: one sample's rows of a
 * batched producer ([S * nrows][C] planes, bfm_conv3x3x3_mfma_batch), read where they lie -- the first decoder above the
 * batched levels (buildingblocks.py:265-276: its upsampled source is one sample of the batch's output) */
int bfm_gn_stats_rows_sliced(const void* rowsA, int totalA, int firstA, int nrowsA, int CA, const void* rowsB, int totalB,
                             int firstB, int nrowsB, int CB, double weightB, int64_t nvox, const float* gamma,
                             const float* beta, int G, float eps, float* scale, float* shift, float* bound,
                             void* workspace, size_t workspace_bytes, void* ticket, bfm_stream_t stream);

/* Winograd F(4,3) along x (conv3d_wino4.hip): the same SingleConv body (buildingblocks.py:31-60) with 13.5 tap-rows
 * per output voxel instead of F(2,3)'s 18 -- dense boxes, one source, no split-K; rounding ~2x F(2,3)'s per layer
 * (2e-6 of max|y| against a float64 convolution), so it is a choice of the tune table, never the planner's default.
 * Weights packed by bfm_pack_conv_weights_wino4 (U = G g in float64); flags bit 0 = accumulate onto `out`;
 * moment_rows as elsewhere ([bfm_conv3x3x3_wino4_rows()][Cout]). */
size_t bfm_pack_conv_weights_wino4_bytes(int Cin, int Cout, int passes);
int bfm_pack_conv_weights_wino4(const float* w_oidhw, int Cin, int Cout, float wmax_abs_host, int passes, void* wpacked,
                                int* wexp_host, bfm_stream_t stream);
int bfm_conv3x3x3_wino4_rows(int D, int H, int W, int passes);
/* the (d,h,w) box of output voxels per workgroup for this volume: 8 x 8 x 4 (conv_wino4d: raw chunks by LDS-DMA one chunk
 * ahead, round 5) wherever the volume holds a full box in z and y, bfm_conv3x3x3_wino_box()'s otherwise */
int bfm_conv3x3x3_wino4_box(int D, int H, int W, int passes, int* box /* [3] */);
int bfm_conv3x3x3_wino4(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                        const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                        int flags, float* out, void* moment_rows /*or NULL*/, bfm_stream_t stream);
/* S same-shape samples in one launch (A (S,D,H,W,CA) -> out (S,D,H,W,Cout)): per-sample scale / shift rows affine_stride
 * apart (0 = CA), bound [S][G], moment_rows [S * bfm_conv3x3x3_wino4_rows()][Cout]; per sample the bits of
 * bfm_conv3x3x3_wino4.  The deep levels of same-shape tiles (buildingblocks.py:31-60 on a batch). */
int bfm_conv3x3x3_wino4_batch(const float* A, int CA, int S, int D, int H, int W, const float* scale, const float* shift,
                              const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                              int flags, float* out, void* moment_rows, int affine_stride, bfm_stream_t stream);
/* The masked form of the same kernel: the boxes (bfm_conv3x3x3_wino4_box) that hold a non-zero voxel of mask_image;
 * workspace bfm_conv3x3x3_wino4_masked_workspace(D, H, W, passes) bytes (4-byte aligned).  There
 * is no uniform-box pair: F(4,3)'s rounding reaches 4 voxels along x where F(2,3)'s numerical support is its
 * mathematical one (conv3d_wino4.hip), so the layers that take that shortcut stay with bfm_conv3x3x3_wino_uniform. */
size_t bfm_conv3x3x3_wino4_masked_workspace(int D, int H, int W, int passes);
/* The uniform-box pair of the F(4,3) kernel (round 5): bfm_conv3x3x3_wino_uniform's arguments and contract (flags from
 * bfm_uniform_boxes on the same box grid, same bits with and without them), for the layers whose output no other layer
 * that takes the shortcut reads -- the skip halves of the last two decoders' first convs (buildingblocks.py:265-276):
 * there a box's numerical support is its mathematical halo (a box is a whole number of quads).  BFM_E_SHAPE where the
 * volume's box differs from bfm_conv3x3x3_wino_box()'s.  scratch: bfm_conv3x3x3_wino4_uniform_scratch(Cout) bytes, 16-byte
 * aligned. */
size_t bfm_conv3x3x3_wino4_uniform_scratch(int Cout);
int bfm_conv3x3x3_wino4_uniform(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                                const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                                int flags, float* out, void* moment_rows, const unsigned char* uniform_flags, void* scratch,
                                bfm_stream_t stream);
int bfm_conv3x3x3_wino4_masked(const float* A, int CA, int D, int H, int W, const float* scale, const float* shift,
                               const float* bound, int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                               int flags, float* out, const float* mask_image, void* workspace, size_t workspace_bytes,
                               bfm_stream_t stream);

size_t bfm_conv3x3x3_mfma_workspace(int Cin, int Cout, int D, int H, int W, int splitk);
int bfm_conv3x3x3_mfma_plan(int Cin, int Cout, int D, int H, int W, int* cfg_out /*[8]*/);
/* cfg[7] bit 0 ("accumulate"): add what `out` already holds before the LeakyReLU -- used for the skip half of a
 * decoder's first conv after bfm_conv3x3x3_upfold wrote the upsampled half. */

/* Decoder._joining + SingleConv (buildingblocks.py:265-276, 361-363), upsampled half only, for exact 2x nearest
 * upsampling (full-res dims == 2 x low-res dims): the 27 taps over the 8-fold replicated low-res tensor collapse to
 * 8 taps per output parity class with pre-summed weights (3.375x fewer FLOPs, the low-res box is what gets staged).
 * B [d][h][w][CB] low-res; scale_b/shift_b = the GroupNorm affine of channels CA.. of the concatenation; bound/G as
 * for _mfma; writes sum over (taps, B channels) WITHOUT activation to out [2d][2h][2w][Cout].  Follow with
 * bfm_conv3x3x3_mfma(A = skip, CB = 0, cfg[7] |= 1) on the skip channels' weights. */
size_t bfm_pack_conv_weights_upfold_bytes(int CB, int Cout, int passes);
int bfm_pack_conv_weights_upfold(const float* w_oidhw /*[Cout][CA+CB][27]*/, int CA, int CB, int Cout,
                                 float wmax_abs_host, int passes, void* wpacked, int* wexp_host, bfm_stream_t stream);
int bfm_conv3x3x3_upfold(const float* B, int CB, int d, int h, int w, const float* scale_b, const float* shift_b,
                         const float* bound, int G, const void* wpacked, int wexp, int Cout, int passes, float* out,
                         bfm_stream_t stream);
/* the same with split-K for low-res levels too small to fill the chip (workspace from _workspace(); 0 bytes = not needed):
 * partial sums per K slab, reduced in slab order */
size_t bfm_conv3x3x3_upfold_workspace(int CB, int d, int h, int w, int Cout);
int bfm_conv3x3x3_upfold_ex(const float* B, int CB, int d, int h, int w, const float* scale_b, const float* shift_b,
                            const float* bound, int G, const void* wpacked, int wexp, int Cout, int passes, float* out,
                            void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* S same-shape samples in one launch (the batched deep levels): B [S][d][h][w][CB], scale_b / shift_b [S][CB] (rows
 * affine_stride floats apart, 0 = CB: the upsampled half's window of the concat's [S][CA + CB] table), bound [S][G], out [S][2d][2h][2w][Cout].  A sample's result is bit-identical to its S = 1 launch: the
 * split-K plan depends on the per-sample shape alone, and a workspace smaller than _batch_workspace() is an error. */
size_t bfm_conv3x3x3_upfold_batch_workspace(int CB, int S, int d, int h, int w, int Cout);
int bfm_conv3x3x3_upfold_batch(const float* B, int CB, int S, int d, int h, int w, const float* scale_b,
                               const float* shift_b, const float* bound, int G, const void* wpacked, int wexp, int Cout,
                               int passes, float* out, void* workspace, size_t workspace_bytes, int affine_stride,
                               bfm_stream_t stream);

int bfm_conv3x3x3_mfma(const float* A, int CA, const float* B, int CB, int D, int H, int W,
                       const bfm_upsample_t* up, const float* scale, const float* shift, const float* bound,
                       int G, const void* wpacked, int wexp, int Cout, float slope, int passes,
                       const int* cfg /*from _plan, or NULL*/, float* out, void* workspace,
                       size_t workspace_bytes, bfm_stream_t stream);

/* ------------------------------------------------------------- MaxPool3d(2)
 * nn.MaxPool3d(kernel_size=2): stride 2, floor, no padding
 * (buildingblocks.py:185-186).  in (D,H,W,C) -> out (D/2,H/2,W/2,C). */
int bfm_maxpool2(const float* in, int C, int D, int H, int W, float* out, bfm_stream_t stream);

/* ------------------------------------------------------------------- tail
 * Everything after the last decoder, one pass over the 64-ch feature map:
 * F.normalize(dim=1) (unet3d/model.py:207-208) -> TaskHead 1x1x1 convs + bias
 * (head.py:52-59) -> SegProcessor softmax / DistProcessor clamp
 * (joiner.py:69-77,149-157) -> get_postprocessor (Trainer/models/__init__.py:
 * 272-354): exp, tanh cortical formula, LUT[argmax], CT*1000, residual+input.
 * The head set is data: `roles[o]` gives the meaning of head output row o. */
enum {
    BFM_ROLE_PLAIN = 0,     /* write as is (T1, T2, FLAIR, regx/y/z, high_res_residual, *_sigma) */
    BFM_ROLE_CT = 1,        /* x1000 */
    BFM_ROLE_BIAS_LOG = 2,  /* exp */
    BFM_ROLE_SEG = 3,       /* softmax member (contiguous run of n_seg rows) */
    BFM_ROLE_DIST = 4,      /* clamp(+-max_dist); order lp, lw[, rp, rw] */
    BFM_ROLE_SR = 5,        /* high_res_residual: also emits high_res = v + input */
    BFM_ROLE_PATHOL = 6     /* sigmoid */
};
typedef struct {
    int n_out;                  /* rows of head_w (sum of head channels) */
    int c_feat;                 /* 64 */
    const float* head_w;        /* [n_out][c_feat] */
    const float* head_b;        /* [n_out] */
    const int32_t* roles;       /* [n_out] BFM_ROLE_* */
    const int32_t* out_slot;    /* [n_out] index into `maps` for this row, -1 = none */
    int seg_first, n_seg;       /* rows of the segmentation head; n_seg==0 -> none */
    const int32_t* seg_lut;     /* [n_seg] label list */
    int n_dist;                 /* 0, 2 (left hemi) or 4 */
    int dist_first;
    float max_dist;
    int unit_feat;              /* apply F.normalize before the heads */
    int slot_high_res;          /* maps slot for residual+input, -1 = none; a residual head with c channels (rows with
                                   BFM_ROLE_SR, contiguous) writes channel j to slot_high_res + j */
    int slot_fake_cortical;     /* maps slot, -1 = none */
    int n_maps;                 /* length of the `maps` pointer array (every out_slot / slot_* is < n_maps) */
    float head_wmax;            /* max |head_w| (host knows it): > 0 with unit_feat selects the split-f16 matrix-core
                                   path (fp32-grade, like conv3x3x3_mfma); 0 keeps the exact fp32 MFMA chain */
    int skip_zero_input;        /* 1: a run of 64 voxels whose `input` is all zero is not evaluated (its maps / label stay
                                   unwritten).  For the tile loop only, which keeps outputs where the tile's input is
                                   non-zero (scripts/demo_test.py:88-100); evaluate_image always passes 0 */
} bfm_tail_desc_t;

int bfm_tail_heads(const float* feat, const float* input /*[nvox], may be NULL*/, int64_t nvox,
                   const bfm_tail_desc_t* desc, float* feat_norm /*[nvox][c_feat] or NULL*/,
                   float* const* maps /*device array of device pointers, each [nvox]*/,
                   float* seg_prob /*[nvox][n_seg] or NULL*/, int64_t* label /*[nvox] or NULL*/,
                   float* raw_out /*[nvox][n_out]: if set, write raw head logits only (TaskHead.forward)*/,
                   bfm_stream_t stream);
/* The same pass writing its maps into the rows of ONE buffer: map i = maps_rows + i * row_stride (row_stride >= nvox
 * floats), so a caller that allocates the maps together needs no device-side pointer table (the tile loop inside a
 * hipGraph: no table-building kernels).  flags bit 0 = skip_zero_input for this call (see bfm_tail_desc_t; the
 * descriptor itself stays the one evaluate_image uses).  Replaces the same reference lines as bfm_tail_heads. */
int bfm_tail_heads_rows(const float* feat, const float* input /*[nvox], may be NULL*/, int64_t nvox,
                        const bfm_tail_desc_t* desc, float* feat_norm /*or NULL*/, float* maps_rows, int64_t row_stride,
                        float* seg_prob /*or NULL*/, int64_t* label /*or NULL*/, int flags, bfm_stream_t stream);

/* ----------------------------------------------------------------- stitch
 * scripts/demo_test.py:88-119 without the NIfTI round trip:
 *   full[k][range] += tile[k] * (tile_input != 0)   then   full[k] /= cnt.
 * tile maps are [td*th*tw] fp32 (or int64 labels, summed as float like the
 * reference, quirk Q5). */
int bfm_stitch_accumulate(const float* tile, const int64_t* tile_label, const float* tile_input,
                          int td, int th, int tw, float* full, int D, int H, int W,
                          int z0, int y0, int x0, bfm_stream_t stream);
/* All K stitched keys of one tile in one launch: maps [n_maps][td*th*tw] (row stride map_stride), sel[k] = map
 * row feeding key k or -1 for the int64 label, full [K][D*H*W].  tile_input == NULL: the rows are already masked
 * and float typed (what bfm_pack_tile_multi produced on a peer rank); sel must then be >= 0. */
int bfm_stitch_accumulate_multi(const float* maps, int64_t map_stride, const int32_t* sel, int K,
                                const int64_t* tile_label, const float* tile_input, int td, int th, int tw, float* full,
                                int D, int H, int W, int z0, int y0, int x0, bfm_stream_t stream);
/* tile_input == NULL in bfm_stitch_accumulate means "already masked".  bfm_mask_tile produces the
 * masked, float-typed tile a rank ships to rank 0 in the multi-GPU path. */
int bfm_mask_tile(const float* tile, const int64_t* tile_label, const float* tile_input, int64_t n, float* out,
                  bfm_stream_t stream);
int bfm_tile_count_add(float* cnt, int D, int H, int W, int z0, int z1, int y0, int y1, int x0, int x1,
                       bfm_stream_t stream);
int bfm_divide_by_count(float* full, const float* cnt, int64_t n, bfm_stream_t stream);
/* the K keys of one tile, masked (x tile_input != 0) and float typed, packed [K][n]: the multi-GPU shipping form */
int bfm_pack_tile_multi(const float* maps, int64_t map_stride, const int32_t* sel, int K, const int64_t* tile_label,
                        const float* tile_input, int64_t n, float* out, bfm_stream_t stream);
/* full [K][vol] /= cnt [vol], one launch */
int bfm_divide_by_count_multi(float* full, const float* cnt, int64_t vol, int K, bfm_stream_t stream);
/* The whole stitch in one launch once every tile's packed rows ([K][td*th*tw], bfm_pack_tile_multi) are resident on
 * one device -- rank 0 of the multi-GPU path, scripts/demo_test.py:108-119: full[k][v] = (0 + sum over the tiles
 * covering v, in table order) / (their number), each voxel written once.  tiles: device table [T][8] int64 =
 * {device pointer to the tile's rows, z0, y0, x0, td, th, tw, 0} in the reference's tile order; T <= 2048.
 * Bit-identical to T x bfm_stitch_accumulate_multi on a zeroed volume + bfm_divide_by_count_multi. */
int bfm_stitch_gather_multi(const int64_t* tiles, int T, int K, float* full, int D, int H, int W, bfm_stream_t stream);

/* The compact shipping form of the same flow.  scripts/demo_test.py:88-100 keeps tile_output * (tile input != 0), and a
 * tile's input is a window of the volume: which voxels of a tile survive is known from the volume alone, on every rank,
 * before any tile is computed.  A tile's K rows then hold its surviving voxels only, in the tile's raster order:
 * rows [K][row_stride], voxel i of the tile at column pos[i] = the number of non-zero input voxels before i in the tile.
 *   bfm_tile_mask_index: pos[] of every tile and nnz[t] = its number of surviving voxels, three launches per volume.
 *     tiles: device table [T][8] int64 = {offset of the tile's first entry in pos[], z0, y0, x0, td, th, tw, first block},
 *     first block = sum of bfm_tile_mask_blocks(td*th*tw) over the tiles before it, total_blocks = that sum over all
 *     tiles; block_ws: total_blocks int32 of scratch.
 *   bfm_pack_tile_compact: bfm_pack_tile_multi writing the surviving voxels only (row_stride >= the tile's nnz).
 *   bfm_stitch_gather_compact: bfm_stitch_gather_multi reading such rows; tiles [T][10] int64 = {rows, the tile's pos,
 *     z0, y0, x0, td, th, tw, row_stride, 0}, volume = the (D,H,W) input the masks come from; T <= 1024.
 * Results are bit-identical to the dense form (a masked-out voxel adds +0 in every tile). */
int bfm_tile_mask_blocks(int64_t tile_voxels);
int bfm_tile_mask_index(const float* volume, int D, int H, int W, const int64_t* tiles, int T, int total_blocks,
                        int32_t* pos, int32_t* nnz, int32_t* block_ws, bfm_stream_t stream);
int bfm_pack_tile_compact(const float* maps, int64_t map_stride, const int32_t* sel, int K, const int64_t* tile_label,
                          const float* tile_input, int64_t n, const int32_t* pos, int64_t row_stride, float* out,
                          bfm_stream_t stream);
int bfm_stitch_gather_compact(const int64_t* tiles, int T, int K, const float* volume, float* full, int D, int H, int W,
                              bfm_stream_t stream);

/* ------------------------------------------------------------ elementwise
 * Per-voxel helpers for the stand-alone processors / post-processor
 * (joiner.py:69-77,149-157; Trainer/models/__init__.py:272-354) and the
 * synthesis augmentations (Generator/utils.py:568-638).  Strides are in
 * elements so channel slices of channels-last buffers are read in place. */
enum {
    BFM_EW_EXP = 0, BFM_EW_AFFINE = 1 /* x*a+b */, BFM_EW_CLAMP = 2 /* [a,b] */, BFM_EW_CLAMP_MIN = 3,
    BFM_EW_GAMMA = 4 /* a*(x/a)^b, add_gamma_transform utils.py:568-572 */, BFM_EW_SIGMOID = 5,
    BFM_EW_DIV = 6 /* x/a */, BFM_EW_NONZERO = 7 /* x!=0 ? 1:0 */, BFM_EW_SUB_DIV = 8 /* (x-a)/b */,
    BFM_EW_GE = 9 /* x>=a ? 1:0, binarize utils.py:65-72 */,
    BFM_EW_NAN_TO_NUM = 10 /* torch.nan_to_num defaults, utils/test_utils.py:239 */
};
enum {
    BFM_EW_ADD = 0, BFM_EW_MUL = 1, BFM_EW_MUL_EXP = 2 /* x*exp(y), add_bias_field utils.py:585-587 */,
    BFM_EW_AXPY_CLAMP0 = 3 /* max(x+a*y,0), add_noise utils.py:633-638 */, BFM_EW_AXPY = 4, BFM_EW_DIV2 = 5,
    BFM_EW_ZERO_WHERE_ZERO = 6 /* y==0 ? 0 : x, target['pathology'][SYN_cerebral == 0] = 0, datasets.py:398-399 */
};
int bfm_ew_unary(int op, const float* in, int64_t in_stride, float* out, int64_t out_stride, int64_t n,
                 float a, float b, bfm_stream_t stream);
int bfm_ew_binary(int op, const float* x, int64_t x_stride, const float* y, int64_t y_stride /*0 = broadcast*/,
                  float* out, int64_t out_stride, int64_t n, float a, bfm_stream_t stream);
/* max |x| over `rows` runs of `len` floats `row_stride` apart, folded into *out_zeroed (device, +0 on entry) by an integer
 * atomic maximum of the bit pattern: the max |w| the weight packers scale by, one launch per layer without a host round
 * trip each (training re-packs every layer after AdamW: round 3 used four torch kernels per layer for it). */
int bfm_absmax_f32(const float* x, int64_t rows, int64_t len, int64_t row_stride, float* out_zeroed, bfm_stream_t stream);
int bfm_softmax_cl(const float* x, int64_t x_row_stride, int C, float* y, int64_t y_row_stride, int64_t n,
                   bfm_stream_t stream);
int bfm_argmax_lut_cl(const float* p, int64_t row_stride, int C, const int32_t* lut, int64_t* out, int64_t n,
                      bfm_stream_t stream);
/* encode_pathology -- Generator/datasets.py:496-518 */
int bfm_pathology_encode(const float* I, const float* P, const float* Pprob, const float* randn, float mu0, float mu1,
                         float s0, float s1, int64_t n, float* out, bfm_stream_t stream);
int bfm_fake_cortical(const float* dist, int64_t row_stride, int n_dist, float* out, int64_t n,
                      bfm_stream_t stream);

/* ---------------------------------------------------------------- synthesis
 * Gather / resample kernels of Generator/utils.py and utils/interpol (fp32, results bit-identical to the
 * reference's CPU path where it is a fixed sequence of IEEE operations).  Volumes are (nx,ny,nz[,C])
 * row-major with channels last, exactly the reference's layout. */
/* fast_3D_interp_torch(X, II, JJ, KK, 'linear', default) -- Generator/utils.py:140-192: valid iff
 * II>0 && II<=nx-1 (strict lower bound), upper corner clamped, invalid -> default_value. */
int bfm_interp3d_linear(const float* X, int nx, int ny, int nz, int C, const float* II, const float* JJ,
                        const float* KK, int64_t n, float default_value, float* out, bfm_stream_t stream);
/* ... 'nearest' -- :124-138: round half to even, clamp; 4-byte elements copied bitwise (int32 or fp32). */
int bfm_interp3d_nearest(const void* X, int nx, int ny, int nz, int C, const float* II, const float* JJ,
                         const float* KK, int64_t n, void* out, bfm_stream_t stream);
/* get_deformed_atlas -- utils/test_utils.py:45-57, fused: where mask>0, sample the atlas trilinearly at
 * A(3x4) applied to 100*(regx,regy,regz); 0 elsewhere. */
int bfm_deformed_atlas(const float* mask, const float* regx, const float* regy, const float* regz, const float* atlas,
                       int nx, int ny, int nz, const float* A_host, int64_t n, float* out, bfm_stream_t stream);
/* The same inside the tile loop -- scripts/demo_test.py:88-89,102-104: the mask operand is the tile's input image and
 * M = (tile_in != 0), i.e. the 0/1 mask the script builds before it calls get_deformed_atlas.  Both entry points read
 * the atlas with L1-bypassing (sc1) loads: see HISTORY.md section 3.3. */
int bfm_deformed_atlas_tile(const float* tile_in, const float* regx, const float* regy, const float* regz,
                            const float* atlas, int nx, int ny, int nz, const float* A_host, int64_t n, float* out,
                            bfm_stream_t stream);
/* myzoom_torch -- Generator/utils.py:200-257.  Per-axis tables (floor index, ceil index, weights) are built
 * by the host exactly as the reference builds them (torch.arange in fp32); the three passes are fused. */
typedef struct { const int32_t* f; const int32_t* c; const float* wf; const float* wc; } bfm_zoom_axis_t;
int bfm_zoom_linear(const float* X, int nx, int ny, int nz, int C, const bfm_zoom_axis_t* axes /*[3]*/, int ox, int oy,
                    int oz, float* out, bfm_stream_t stream);
/* one axis of gaussian_blur_3d -- Generator/utils.py:84-94: zero-padded 1-D correlation, odd kernel. */
int bfm_conv1d_axis(const float* in, int nx, int ny, int nz, int axis, const float* kern, int klen, float* out,
                    bfm_stream_t stream);
/* interpol.grid_push / grid_grad (order 1, 3-D) -- utils/interpol/iso1.py:136-387; with grid_pull they make
 * grid_pull / grid_push differentiable to first order (autograd.py:125-190, pushpull.py:262-310).
 * push: inp (Bi,C,ix,iy,iz) scattered through grid (Bg,ix,iy,iz,3) into out (B,C,nx,ny,nz), which MUST be zero on
 * entry (fp32 atomics).  grad: out (B,C,ox,oy,oz,3). */
int bfm_grid_push3d_linear(const float* inp, int Bi, int C, int ix, int iy, int iz, const float* grid, int Bg, int nx,
                           int ny, int nz, const int* bound, int extrapolate, float* out_zeroed, bfm_stream_t stream);
int bfm_grid_grad3d_linear(const float* inp, int Bi, int C, int nx, int ny, int nz, const float* grid, int Bg, int ox,
                           int oy, int oz, const int* bound, int extrapolate, float* out, bfm_stream_t stream);

/* The window a tile takes of the volume (scripts/demo_test.py:84-86, `full_im[:, :, x0:x1, y0:y1, z0:z1]`), copied into
 * a contiguous [d][h][w] buffer: vol [D][H][W] fp32, the window starts at (z0, y0, x0) in that index order. */
int bfm_crop3d(const float* vol, int D, int H, int W, int z0, int y0, int x0, int d, int h, int w, float* out,
               bfm_stream_t stream);

/* Pre-processing around the inference path (utils/test_utils.py:235-284 prepare_image).
 * permute_flip3d: align_volume_to_ref's swapaxes + flips (utils/misc.py:1207-1247) in one gather:
 *   out dims = (n[perm[0]], n[perm[1]], n[perm[2]]); out[i0,i1,i2] = in[j], j[perm[a]] = flip[a] ? n[perm[a]]-1-ia : ia.
 * bbox_nonzero: zero_crop's torch.argwhere(x > tol) min/max (utils/test_utils.py:60-72): box[6] = {x0,y0,z0,x1,y1,z1}
 *   (x1 exclusive); all-background volumes give x0 > x1.
 * mean_lastdim: im.mean(dim=-1) for multi-frame inputs (utils/test_utils.py:243). */
int bfm_permute_flip3d(const float* in, int nx, int ny, int nz, const int* perm /*[3] host*/,
                       const int* flip /*[3] host*/, float* out, bfm_stream_t stream);
int bfm_bbox_nonzero(const float* in, int nx, int ny, int nz, float tol, int32_t* box /*[6] device*/,
                     bfm_stream_t stream);
int bfm_mean_lastdim(const float* in, int64_t n, int c, float* out, bfm_stream_t stream);

/* interpol.resize(interpolation=3, prefilter=True) -- utils/interpol/resize.py:13-119, coeff.py:254-344, nd.py:36-142
 * (Generator/datasets.py:337-338, `bspline_zooming`).  One axis at a time, in place for the prefilter.
 * prefilter: bound 1 ('nearest') or 3 ('dct2') -> DCT-II conditions; the host passes the scalars the reference derives
 * from the pole z = sqrt(3)-2 (gain (1-z)(1-1/z), pole_last, init_scale z/(1-z^2n), final_scale z/(z-1)) and the
 * fp32 table init_w[i-1] = z^i + z^(2n-1-i), i = 1..n-2.  resample: out[o] = sum of 4 cubic B-spline taps around
 * coord[o] (fp32 source coordinates of the output samples along `axis`), indices reflected (3) or clamped (1). */
int bfm_bspline3_prefilter_axis(float* vol, int nx, int ny, int nz, int axis, int bound, float pole, float gain,
                                const float* init_w, float pole_last, float init_scale, float final_scale,
                                bfm_stream_t stream);
int bfm_bspline3_resample_axis(const float* in, int nx, int ny, int nz, int axis, const float* coord, int n_out,
                               int bound, float* out, bfm_stream_t stream);

/* interpol.grid_pull(interpolation='linear') -> iso1.pull3d -- utils/interpol/iso1.py:28-133.
 * inp (Bi,C,nx,ny,nz), grid (Bg,ox,oy,oz,3), out (max(Bi,Bg),C,ox,oy,oz); bound[3] in 0..6
 * (zero, replicate, dct1, dct2, dst1, dst2, dft -- bounds.py:8-15); extrapolate 0 no / 1 yes / 2 hist. */
int bfm_grid_pull3d_linear(const float* inp, int Bi, int C, int nx, int ny, int nz, const float* grid, int Bg, int ox,
                           int oy, int oz, const int* bound, int extrapolate, float* out, bfm_stream_t stream);
/* BaseGen.deform_grid -- Generator/datasets.py:264-303: (xc+F) -> A*. + c2, clamp to the source shape, and the
 * six global min/max (minmax = {min x,y,z, max x,y,z}, device).  F is [n][3] or NULL. */
size_t bfm_deform_grid_workspace(int sx, int sy, int sz);
int bfm_deform_grid(const float* F, int sx, int sy, int sz, const float* A_host /*[9]*/, const float* c2_host /*[3]*/,
                    const int* shp_host /*[3]*/, float* xx, float* yy, float* zz, float* minmax /*[6]*/,
                    void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* generate_sample core -- Generator/datasets.py:366-372: mus[round(G)] + sigmas[round(G)]*randn, clamp >= 0
 * (label 77 merged into 2). */
/* generate_sample's pathology branch -- Generator/datasets.py:388-396: cerebral[i] = (round(G[i]) == 0) ? 0 : syn[i]
 * (G == 77 counts as 2, :368) and the four sums behind wm_mean / gm_mean: stats[0..3] = sum(syn | label in {2,41}),
 * count of those, sum(syn | label not in {0,2,41}), count (fp64, written whole).  partials: workspace of
 * 4 * BFM_CLASS_STATS_BLOCKS doubles -- the blocks' sums, added in block order, so the result does not depend on the
 * run (the reference's torch sums are deterministic on CPU; pathol_direction = gm_mean > wm_mean hangs on them). */
#define BFM_CLASS_STATS_BLOCKS 1024
int bfm_label_class_stats(const float* G, const float* syn, int64_t n, float* cerebral, double* stats /*[4]*/,
                          double* partials, bfm_stream_t stream);
int bfm_label_gauss(const float* G, const float* mus, const float* sigmas, const float* randn, int64_t n, int ntab,
                    float* out, bfm_stream_t stream);
/* onehotmatrix[lut[S]] -- Generator/utils.py:408-411: out [n][n_labels]. */
int bfm_onehot_lut(const int32_t* S, const int32_t* lut, int nlut, int n_labels, int64_t n, float* out,
                   bfm_stream_t stream);

/* Perlin noise on a lattice of host-drawn gradients -- ShapeID/perlin3d.py:38-83 (fp64, NumPy semantics).
 * grad: [(rx+1)(ry+1)(rz+1)][3]. */
int bfm_perlin3d(const double* grad, int sx, int sy, int sz, int rx, int ry, int rz, double* out, bfm_stream_t stream);
/* np.percentile support (perlin3d.py:84-90): 16-bit digit histogram of the order-preserving key of each
 * double among keys whose higher bits equal `prefix`; four passes select an order statistic. */
int bfm_radix_hist_f64(const double* x, int64_t n, uint64_t prefix, int shift, uint32_t* hist65536, bfm_stream_t stream);
int bfm_threshold_mask_f64(const double* x, int64_t n, double thr, double* masked, double* mask, bfm_stream_t stream);
/* stream_3D(gradient_c(a), gradient_c(b), gradient_c(c)) * mult -- ShapeID/misc.py:66-80,198-259. */
int bfm_curl3d(const double* a, const double* b, const double* c, int sx, int sy, int sz, float mult, float* Vx,
               float* Vy, float* Vz, bfm_stream_t stream);
/* AdvDiffPDE.forward, perf_pattern='adv', V_type='vector_div_free' -- ShapeID/DiffEqs/pde.py:616-640:
 * out = -(Vx*dxC + Vy*dyC + Vz*dzC) with upwind differences of the (optionally Neumann-conditioned) field. */
int bfm_advect_upwind_rhs(const void* C, int c_is_f64, const float* Vx, const float* Vy, const float* Vz, int sx,
                          int sy, int sz, int neumann_bc, float* out, bfm_stream_t stream);
/* Runge-Kutta tensor arithmetic of Dopri5Solver (ShapeID/DiffEqs/rk_common.py:22-61, misc.py:22-25,84-170,
 * interp.py:5-65).  k[j] are fp32 stage derivatives, coef[j] the fp32 value of (dt*c_j). */
typedef struct { const float* k[7]; float coef[7]; int nk; } bfm_kset_t;
int bfm_rk_combine(const void* y0 /*NULL: out = sum*/, int is_f64, const bfm_kset_t* ks, void* out, int64_t n,
                   bfm_stream_t stream);
size_t bfm_reduce_workspace(void);
int bfm_rk_error_sumsq(const bfm_kset_t* ks, const void* y0, const void* y1, int is_f64, double atol, double rtol,
                       int64_t n, double* out, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
int bfm_scaled_sumsq(const void* a, const void* b /*may be NULL*/, int ab_is_f64, const void* y0, int y_is_f64,
                     double atol, double rtol, int64_t n, double* out, void* workspace, size_t workspace_bytes,
                     bfm_stream_t stream);
int bfm_dopri5_dense_eval(const void* y0, const void* y1, int is_f64, const bfm_kset_t* mid, double dt, double x,
                          void* out, int64_t n, bfm_stream_t stream);
/* op: 0 min, 1 max, 2 sum(x), 3 sum(x*y); result in fp64 (deterministic two-stage reduction). */
int bfm_reduce_f32(int op, const float* x, const float* y, int64_t n, double* out, void* workspace,
                   size_t workspace_bytes, bfm_stream_t stream);
int bfm_reduce_f64(int op, const double* x, const double* y, int64_t n, double* out, void* workspace,
                   size_t workspace_bytes, bfm_stream_t stream);

/* ---- the generator item without host round trips (round 4; brainfm_amd/csrc/synth_item.hip) --------------------------
 * BrainIDGen.__getitem__ (Generator/datasets.py:700-757) re-reads every NIfTI volume of the case and crops it on the
 * host per item (Generator/utils.py:296-305); here the case's volumes stay resident in HBM, the crop is a box inside
 * them, scalar operands (min / max / sums / order statistics) stay in device memory between kernels, and short chains
 * between two reductions are one kernel.  The arithmetic and its order are those of the unfused entry points above. */

/* torch.randn(shape, device) of the generator: Philox4x32-10 counter (element index / 4, offset) under `seed`,
 * Box-Muller, out = scale * N(0,1).  A different stream from torch's (RNG-stream parity across devices is not a goal
 * of the reference either); the same (seed, offset) gives the same field on every run and device. */
int bfm_randn_philox(float* out, int64_t n, uint64_t seed, uint64_t offset, float scale, bfm_stream_t stream);

/* BaseGen.generate_deformation / deform_grid (datasets.py:187-303) with myzoom_torch(Fsmall, size / small)
 * (utils.py:200-257) folded in: the zoomed field is evaluated per voxel from Fsmall [fnx][fny][fnz][3] and the zoom
 * tables instead of being written and read back (Fsmall NULL: affine only).  _minmax writes the six extrema of the
 * clamped coordinates (device, {min x,y,z, max x,y,z}) and nothing else; the host reads them (the reference
 * synchronises there too, datasets.py:296-301), and _write stores the coordinates minus lo (+ F [n][3] on request).
 * photo_zero_y: F[..., 1] = 0 (photo mode, datasets.py:243). */
size_t bfm_deform_zoom_workspace(void);
int bfm_deform_zoom_minmax(const float* Fsmall, int fnx, int fny, int fnz, const bfm_zoom_axis_t* ax, int photo_zero_y,
                           int sx, int sy, int sz, const float* A_host, const float* c2_host, const int* shp_host,
                           float* minmax6, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
int bfm_deform_zoom_write(const float* Fsmall, int fnx, int fny, int fnz, const bfm_zoom_axis_t* ax, int photo_zero_y,
                          int sx, int sy, int sz, const float* A_host, const float* c2_host, const int* shp_host,
                          const float* lo_host /*[3]*/, float* xx, float* yy, float* zz, float* F_out /*or NULL*/,
                          bfm_stream_t stream);

/* read_and_deform and its callers (Generator/utils.py:296-322; _image :331-345, _distance :376-400,
 * _registration :462-473) for up to BFM_GATHER_MAX_JOBS volumes that share one coordinate field, straight from the
 * resident full volumes [nx][ny][nz]: box6 = {x1,y1,z1,x2,y2,z2} is the crop the reference takes (upper ends clipped
 * to the volume like a NumPy slice); validity and corner clamps of fast_3D_interp_torch are evaluated in CROP space.
 * Per job: texel -> pre (0 none, 1 nan_to_num, 2 nan_to_num then (x - mean) / scale) -> trilinear; invalid
 * coordinates take 0 or, with default_max, the crop's maximum after `pre` (read_and_deform's default_value_linear);
 * then r / post_div (0 = off), clamp, r * sign (0 = off), written at the voxel (mirrored along axis 0 with flip0).
 * want_minmax leaves min / max of the job's output in scalars[BFM_GATHER_MAX_JOBS + 2j (+1)]; scalars[j] holds the
 * default value of job j.  scalars: 3 * BFM_GATHER_MAX_JOBS doubles (device). */
#define BFM_GATHER_MAX_JOBS 12
typedef struct {
    const float* src; float* out;
    float mean, scale; int pre; int default_max;
    float post_div; int clamp; float clamp_lo, clamp_hi; float sign; int want_minmax;
} bfm_gather_job_t;
size_t bfm_gather_targets_workspace(void);
int bfm_gather_targets(const bfm_gather_job_t* jobs, int njobs, int nx, int ny, int nz, const int* box6_host,
                       const float* II, const float* JJ, const float* KK, int sx, int sy, int sz, int flip0,
                       double* scalars, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* I -= min(I); I /= max(I) (read_and_deform_image, utils.py:340-342) with {min, max} of I in device memory. */
int bfm_minmax_normalise(float* x, int64_t n, const double* minmax_dev, bfm_stream_t stream);
/* read_and_deform_segmentation (utils.py:402-425): nearest gather from the resident int32 label volume (crop-space
 * rounding and clamps), lut, one-hot rows out [sx][sy][sz][n_labels]; flip0: torch.flip(., [0])[..., vflip]. */
int bfm_gather_onehot(const int32_t* S, int nx, int ny, int nz, const int* box6_host, const float* II, const float* JJ,
                      const float* KK, int sx, int sy, int sz, int flip0, const int32_t* lut, int nlut, int n_labels,
                      const int32_t* vflip /*[n_labels] or NULL*/, float* out, bfm_stream_t stream);
/* the same, out as [n_labels][sx][sy][sz]: the element order the reference's .permute([3, 0, 1, 2]) view is read in
 * (Generator/utils.py:423-425), so that no consumer has to make it contiguous */
int bfm_gather_onehot_rows(const int32_t* S, int nx, int ny, int nz, const int* box6_host, const float* II, const float* JJ,
                      const float* KK, int sx, int sy, int sz, int flip0, const int32_t* lut, int nlut, int n_labels,
                      const int32_t* vflip /*[n_labels] or NULL*/, float* out, bfm_stream_t stream);

/* np.percentile(x, q) (method 'linear', ShapeID/perlin3d.py:84-90) without leaving the device: radix select of the
 * order statistic k_lo (six passes over 11 / 9-bit digits of the order-preserving key, block histograms in LDS), the
 * next one where needed, NumPy's _lerp with weight t.  out3 (device) = {percentile, x_(k_lo), x_(k_lo+1)}. */
size_t bfm_percentile_workspace(void);
int bfm_percentile_f64(const double* x, int64_t n, int64_t k_lo, int need_next, double t, double* out3, void* workspace,
                       size_t workspace_bytes, bfm_stream_t stream);
/* generate_perlin_noise_3d's mask (perlin3d.py:86-90) with the threshold in device memory: masked = x * (x >= thr),
 * mask (optional) = (x >= thr); max_out = max(masked).  binarize (Generator/utils.py:65-72): P = (p >= thres * max)
 * in p's dtype, sum_out = sum(P). */
size_t bfm_shape_workspace(void);
int bfm_shape_threshold_f64(const double* noise, int64_t n, const double* thr_dev, double* masked, double* mask,
                            double* max_out, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
int bfm_shape_binarize(const void* p, int is_f64, int64_t n, const double* max_dev, double thres, void* P,
                       double* sum_out, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* generate_sample's pathology branch (datasets.py:398-399): target['pathology'][cer == 0] = 0 and the same for
 * pathology_prob, in place in their own dtype; sum_out = sum of the masked pathology (the test of :322 / :387). */
int bfm_pathology_mask(void* P, void* Pprob, int is_f64, const float* cerebral, int64_t n, double* sum_out,
                       void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* encode_pathology (datasets.py:496-518) with I_mu = sum(I*P) / sum(P) kept on the device (dotsum_out[2]); u4 = the
 * uniform draws behind pth_mus[0], pth_mus[1], pth_sigmas[0], pth_sigmas[1]; direction 1 / 0, or -1 to take
 * gm_mean > wm_mean from bfm_label_class_stats' sums (datasets.py:392-404) without reading them back. */
size_t bfm_pathology_encode_workspace(void);
int bfm_pathology_encode_dev(const float* I, const void* P, const void* Pprob, int is_f64, const float* randn,
                             const float* u4_host, int direction, const double* class_stats_dev, int64_t n, float* out,
                             double* dotsum_out, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* resample_resolution's sample (utils.py:600-606): fast_3D_interp_torch on meshgrid(ax, ay, az), the grid never
 * materialised. */
int bfm_interp3d_linear_axes(const float* X, int nx, int ny, int nz, const float* ax, const float* ay, const float* az,
                             int ox, int oy, int oz, float default_value, float* out, bfm_stream_t stream);
/* augment_sample's tail (datasets.py:340-352): input = I / max, residual = high_res / max - input (optional), both
 * mirrored along axis 0 with flip0; max in device memory. */
int bfm_sample_finalize(const float* I, const float* high_res, int sx, int sy, int sz, const double* max_dev, int flip0,
                        float* input_out, float* residual_out, bfm_stream_t stream);
/* elementwise with the scalar in device memory: op 0: x / s, op 1: x >= a * s ? 1 : 0 (fp32). */
int bfm_ew_dev(int op, const float* x, int64_t n, const double* scalar_dev, float a, float* out, bfm_stream_t stream);

/* Dormand-Prince integration of AdvDiffPDE ('adv', div-free V) with the step controller on the device
 * (ShapeID/DiffEqs/dopri5.py:58-172, rk_common.py:22-61, misc.py:145-170, interp.py:5-65, pde.py:616-640): one kernel per
 * stage (stage state evaluated at the stencil points from y and the k_j, never stored), error norm, accept / reject with
 * the reference's forced-accept clamps, next step size and the dense outputs at t_out all computed from / into `state`
 * (device, bfm_dopri5_advect_state_bytes()).  The host calls _init, copies y0 into y[0] and sol[0], f(t0, y0) into f[0],
 * then enqueues steps in chunks with _steps and reads the state block back between chunks: its int fields at byte offset
 * 40 are {cur, done, next_out, nsteps, naccept, accepted, err}; steps enqueued past the end do nothing.  y / sol are in
 * the state dtype (fp64 or fp32), stages fp32.  workspace: bfm_dopri5_advect_workspace(sx, sy, sz) bytes. */
typedef struct {
    void* y[2]; float* f[2]; float* k[5];
    const float *Vx, *Vy, *Vz;
    int sx, sy, sz, neumann_bc, is_f64;
    double atol, rtol, tol_min_dt, dt_max, safety, ifactor, dfactor;
    const double* t_out /*device [nt]*/; int nt; void* sol /*[nt][n]*/;
    void* state; void* workspace;
} bfm_dopri5_advect_t;
size_t bfm_dopri5_advect_state_bytes(void);
size_t bfm_dopri5_advect_workspace(int sx, int sy, int sz);
int bfm_dopri5_advect_init(const bfm_dopri5_advect_t* d, double t0, double dt0, bfm_stream_t stream);
int bfm_dopri5_advect_steps(const bfm_dopri5_advect_t* d, int nsteps, bfm_stream_t stream);

/* ---- backward pass of the SingleConv block and its neighbours (first correct version) -------------------
 * The reference trains through torch autograd over buildingblocks.py:31-60 (GroupNorm -> Conv3d -> LeakyReLU),
 * :185-186 (MaxPool3d(2)), :265-276,361-363 (nearest upsample + concat).
 *   lrelu_bwd    dP = dY * (Y > 0 ? 1 : slope)                 (n % 4 == 0)
 *   wgrad        dW[co][ci][27] = sum_v dP[v][co] * GN(x)[v+tap][ci]   (exact fp32 matrix cores, fixed split order)
 *   data grad    = bfm_conv3x3x3_mfma / _direct on the transposed, tap-mirrored weights with identity affine, slope 1
 *   gn_bwd       dXn -> dA (skip channels), dB (low-res channels: sum over the replica box), dgamma, dbeta
 *   maxpool2_bwd gradient to the first maximum of each 2x2x2 window (scan order dz,dy,dx), zeros elsewhere */
int bfm_lrelu_bwd(const float* dY, const float* Y, int64_t n, float slope, float* dP, bfm_stream_t stream);
/* the same, also writing max |dP| (device float; the split-fp16 weight / data gradient kernels scale dP by it) */
int bfm_lrelu_bwd_ex(const float* dY, const float* Y, int64_t n, float slope, float* dP, float* absmax,
                     bfm_stream_t stream);
size_t bfm_conv3x3x3_wgrad_workspace(int Cin, int Cout, int D, int H, int W);
int bfm_conv3x3x3_wgrad(const float* dP, int Cout, const float* A, int CA, const float* B, int CB, int D, int H, int W,
                        const bfm_upsample_t* up, const float* scale, const float* shift, float* dW /*[Cout][Cin][27]*/,
                        void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* passes = 0: exact fp32 matrix core (= bfm_conv3x3x3_wgrad).  passes = 3: split-fp16 (hi*hi + hi*lo + lo*hi, fp32
 * accumulate -- the forward kernels' accuracy class) on layers with Cout % 64 == 0, CA % 32 == 0, Cin % 32 == 0, the fp32
 * kernel elsewhere; needs dp_bound [1] = max |dP| and x_bound [G] = max |GroupNorm-applied input| per group (device). */
int bfm_conv3x3x3_wgrad_ex(const float* dP, int Cout, const float* A, int CA, const float* B, int CB, int D, int H, int W,
                           const bfm_upsample_t* up, const float* scale, const float* shift, const float* dp_bound,
                           const float* x_bound, int G, int passes, float* dW, void* workspace, size_t workspace_bytes,
                           bfm_stream_t stream);
size_t bfm_gn_bwd_workspace(int C, int D, int H, int W);
int bfm_gn_bwd(const float* dXn, const float* A, int CA, const float* B, int CB, int D, int H, int W,
               const bfm_upsample_t* up, const int32_t* startD, const int32_t* startH, const int32_t* startW,
               const float* mean, const float* rstd, const float* gamma, int G, float* dA, float* dB, float* dgamma,
               float* dbeta, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
int bfm_maxpool2_bwd(const float* in, const float* dOut, int C, int D, int H, int W, float* dIn, bfm_stream_t stream);
/* weights of the data-gradient conv: out[ci][co][26-t] = w[co][ci][t], zero rows for Cin <= ci < CinPad */
int bfm_transpose_mirror_weights(const float* w_oidhw, int Cout, int Cin, int CinPad, float* out, bfm_stream_t stream);

/* ---- losses, task-head backward and optimiser step (the rest of one training iteration) -------------------
 * Replaces, for the supervised heads, Trainer/models/criterion.py:111-124,178-186,215-294 + losses.py:10-74 (loss values
 * and, through autograd, their gradients), head.py:52-59 + model.py:207 (1x1x1 heads over F.normalize'd features),
 * torch.optim.AdamW and the clip/unscale reductions of Trainer/engine.py:128-147.
 * raw / dRaw are the channels-last head outputs [nvox][n_out] of bfm_tail (raw mode); targets and weights keep the
 * reference's NCDHW layout (one channel = nvox contiguous floats; seg target [ns][nvox]). Every loss ADDS
 * coef * dL/draw into its columns of dRaw (NULL: value only) and writes the loss value(s) in fp64 (device memory).
 *   l1       mean(|o*m - t*m| * w); o clamped to +-clampv first when clampv > 0 (DistProcessor); weight, mask_mul optional
 *   grad_l1  mean|dx(o) - dx(t)|w + mean|dy ..|w + mean|dz ..|w, forward differences, zero on the last slice
 *   seg      p = softmax(raw[c0:c0+ns]);  loss_out[0] = sum_v CE_v, loss_out[1..ns] = sum_v p t, loss_out[1+ns..] =
 *            sum_v (p+t); gradient of coef_ce * mean_v CE + coef_dice * Dice. P [nvox][ns] receives the probabilities. */
size_t bfm_loss_workspace(int ns);
int bfm_loss_l1(const float* raw, int n_out, int co, const float* target, const float* weight, const float* mask_mul,
                int64_t nvox, float clampv, int l2 /* 1: mean squared error (bias_field_log_type 'l2') */, float coef,
                float* dRaw, double* loss_out, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* n <= 32 l1 / l2 entries of one sample in one pass over raw (host arrays of n columns, l2 flags, clamps, coefficients and
 * device pointers; weights / masks arrays or their elements may be NULL); loss_out [n] (device, fp64 means) */
size_t bfm_loss_l1_multi_workspace(void);
int bfm_loss_l1_multi(const float* raw, int n_out, int64_t nvox, int n, const int32_t* cols, const int32_t* l2,
                      const float* clampv, const float* coef, const float* const* targets, const float* const* weights,
                      const float* const* masks, float* dRaw, double* loss_out, void* workspace, size_t workspace_bytes,
                      bfm_stream_t stream);
int bfm_loss_grad_l1(const float* raw, int n_out, int co, const float* target, const float* weight, int D, int H, int W,
                     float coef, float* dRaw, double* loss_out, void* workspace, size_t workspace_bytes,
                     bfm_stream_t stream);
/* every gradient-L1 entry of a sample in one launch: entry k is bfm_loss_grad_l1 on column cols[k] (distinct columns) with
 * coef[k], targets[k], weights[k] (or NULL); loss_out[k] as there.  n <= 32; workspace: bfm_loss_l1_multi_workspace(). */
int bfm_loss_grad_l1_multi(const float* raw, int n_out, int n, const int32_t* cols, const float* coef,
                           const float* const* targets, const float* const* weights, int D, int H, int W, float* dRaw,
                           double* loss_out, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
int bfm_loss_seg(const float* raw, int n_out, int c0, int ns, const float* target, const float* wce, const float* wdice,
                 int64_t nvox, float coef_ce, float coef_dice, float* P, float* dRaw, double* loss_out /*[1+2ns]*/,
                 void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* Trainer/models/criterion.py:193-212 loss_pathol_ce / loss_pathol_dice on p = sigmoid(raw) (PatholProcessor,
 * Trainer/models/joiner.py:79-87), the one-channel pathology head: ce = mean_v(-log(max(p, 1e-5)) t), dice = 1 - 2 sum(p t) /
 * max(sum(p + t), 1e-5).  The head output of voxel v is raw[col_offset + v * voxel_stride] (channels-last: column, n_out;
 * rows: column * row_stride, 1); dRaw (same addressing, may be NULL) += coef_ce d ce + coef_dice d dice; loss_ce / loss_dice
 * (device fp64; either may be NULL: that loss and its gradient are left out).  Fixed-order fp64 reductions. */
size_t bfm_loss_pathol_workspace(void);
int bfm_loss_pathol(const float* raw, int64_t col_offset, int64_t voxel_stride, const float* target, int64_t nvox,
                    float coef_ce, float coef_dice, float* dRaw, double* loss_ce, double* loss_dice, void* workspace,
                    size_t workspace_bytes, bfm_stream_t stream);
/* dW [n_out][C], db [n_out], dFn [nvox][C] from dRaw [nvox][n_out] and the normalised features Fn [nvox][C] */
size_t bfm_head_bwd_workspace(int n_out, int C, int64_t nvox);
int bfm_head_bwd(const float* dRaw, const float* Fn, const float* head_w, int n_out, int C, int64_t nvox, float* dW,
                 float* db, float* dFn, void* workspace, size_t workspace_bytes, bfm_stream_t stream);
/* ---- the same five steps with the head outputs as [n_out] ROWS of nvox values (row pitch row_stride >= nvox) instead of
 * channels-last [nvox][n_out].  criterion.py walks one channel of the outputs per loss (outputs[key][:, c], NCDHW in the
 * reference): in rows every access is a contiguous run, the kernels need no LDS staging and the per-voxel expressions are
 * those of the channels-last entries above.  P of bfm_loss_seg_rows is [ns][nvox]; ns <= 64.  bfm_tail_raw_rows is
 * TaskHead.forward alone (head.py:52-59) writing that layout; bfm_head_bwd_rows needs C == 64 and n_out <= 96. */
int bfm_tail_raw_rows(const float* feat, int64_t nvox, const bfm_tail_desc_t* desc, float* feat_norm, float* raw_rows,
                      int64_t row_stride, bfm_stream_t stream);
int bfm_loss_l1_multi_rows(const float* raw_rows, int64_t row_stride, int n_out, int64_t nvox, int n, const int32_t* cols,
                           const int32_t* l2, const float* clampv, const float* coef, const float* const* targets,
                           const float* const* weights, const float* const* masks, float* dRaw_rows, double* loss_out,
                           void* workspace, size_t workspace_bytes, bfm_stream_t stream);
int bfm_loss_grad_l1_multi_rows(const float* raw_rows, int64_t row_stride, int n_out, int n, const int32_t* cols,
                                const float* coef, const float* const* targets, const float* const* weights, int D, int H,
                                int W, float* dRaw_rows, double* loss_out, void* workspace, size_t workspace_bytes,
                                bfm_stream_t stream);
int bfm_loss_seg_rows(const float* raw_rows, int64_t row_stride, int n_out, int c0, int ns, const float* target,
                      const float* wce, const float* wdice, int64_t nvox, float coef_ce, float coef_dice, float* P,
                      float* dRaw_rows, double* loss_out /*[1+2ns]*/, void* workspace, size_t workspace_bytes,
                      bfm_stream_t stream);
int bfm_head_bwd_rows(const float* dRaw_rows, int64_t row_stride, const float* Fn, const float* head_w, int n_out, int C,
                      int64_t nvox, float* dW, float* db, float* dFn, void* workspace, size_t workspace_bytes,
                      bfm_stream_t stream);
int bfm_normalize_bwd(const float* feat, const float* dFn, int C, int64_t nvox, float eps, float* dfeat,
                      bfm_stream_t stream);
/* One descriptor per parameter tensor for the multi-tensor forms below (device memory, 64 bytes).  bias1 = 1 - beta1^t,
 * bias2_sqrt = sqrt(1 - beta2^t) of the tensor's own step count t (torch.optim.AdamW keeps one per parameter); first_chunk =
 * index of the tensor's first chunk in chunk_tensor. */
typedef struct bfm_adam_tensor {
    float* p; const float* g; float* m; float* v;
    int64_t n;
    float grad_scale, bias1, bias2_sqrt;
    int32_t first_chunk;
    int64_t reserved;
} bfm_adam_tensor_t;
/* utils/misc.py:1329-1338 clip_gradients' per-parameter norms and Trainer/models/__init__.py:362-366's AdamW over ALL
 * parameters in one launch each: a block owns one chunk (chunk_elems elements, a multiple of 4) of one tensor;
 * chunk_tensor[c] = tensor of chunk c (device).  sums_out[t] = sum of squares of tensors[t].g (fp64, fixed order);
 * workspace >= nchunks * 8 bytes. */
int bfm_grad_sumsq_multi(const bfm_adam_tensor_t* tensors, int ntensors, const int32_t* chunk_tensor, int nchunks,
                         int64_t chunk_elems, double* sums_out, int32_t* nonfinite, void* workspace, size_t workspace_bytes,
                         bfm_stream_t stream);
int bfm_adamw_step_multi(const bfm_adam_tensor_t* tensors, int ntensors, const int32_t* chunk_tensor, int nchunks,
                         int64_t chunk_elems, float lr, float beta1, float beta2, float eps, float weight_decay,
                         bfm_stream_t stream);
/* torch.optim.AdamW step t (1-based) on one flat parameter; g is multiplied by grad_scale first (unscale * clip) */
int bfm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, bfm_stream_t stream);
/* out[0] = sum g^2 (fp64); nonfinite[0] |= 1 when g holds inf/nan (GradScaler.unscale_ / clip_grad_norm_) */
int bfm_grad_sumsq(const float* g, int64_t n, double* out, int32_t* nonfinite, void* workspace, size_t workspace_bytes,
                   bfm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BRAINFM_HIP_H */
