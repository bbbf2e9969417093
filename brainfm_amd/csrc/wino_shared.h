// What conv3d_wino.hip (F(2,3)) and conv3d_wino4.hip (F(4,3)) share: the box of a volume and the device-built list of
// the boxes a masked launch computes -- the flags of bfm_uniform_boxes and the lists of both kernels are per box, so
// both kernels must cut a volume into the same boxes.
#pragma once
#include "bfm_common.h"

// the 256-voxel box (TD x TH x TW) conv_wino uses for this volume; false: none fits
bool bfm_wino_choose_box(int D, int H, int W, int npl, int& TD, int& TH, int& TW);
// act[nMt pad 4] | count | list[nMt] in `ws` (bfm_conv3x3x3_wino_masked_workspace bytes): the boxes that hold a non-zero
// voxel of the (D,H,W) image, ascending; returns the launch status
int bfm_wino_mask_list(const float* mask_img, int D, int H, int W, int TD, int TH, int TW, int nTy, int nTx, int nMt, void* ws,
                       bfm_stream_t stream);
