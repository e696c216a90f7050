/* Experimental entry points of libspeecht_hip.so: the opt-in "bf16x6" convolution path
 * (ST_CONV_MODE=bf16x6).  Not part of the drop-in boundary -- signatures may change between
 * rounds -- but declared here so that every exported symbol of the library has a prototype.
 *
 * bf16x6: an fp32 tensor is carried as THREE bf16 planes (h | m | l, an exact split: x = h + m + l) with the
 * geometry of the fp32 tensor; a product is evaluated as the six largest cross terms on the bf16 matrix pipe
 * with fp32 accumulation, which is at least as accurate as an fp32 FMA chain.  `*_planes` arguments point to
 * 3 * plane_elems bf16 values.  Same conventions as speecht_hip.h (status codes, st_last_error, streams). */
#ifndef SPEECHT_HIP_EXPERIMENTAL_H_
#define SPEECHT_HIP_EXPERIMENTAL_H_

#include "speecht_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* fp32 [n] -> planes 3 x [n] (n % 4 == 0) */
int st_exp_split3_bf16(const float* src, size_t n, void* planes, void* stream);
/* packed filters [k_pad][n_pad] fp32 -> planes 3 x [n_pad][k_pad] (transposed: reduction-contiguous) */
int st_exp_split3_transpose_bf16(const float* packed, int k_pad, int n_pad, void* planes, void* stream);
/* y = relu?(conv(x) + bias): fp32 y->base and, when y_planes != NULL, the planes of y for the next layer */
int st_exp_conv1d_fwd_bf16x6(const st_tensor3* x, const void* x_planes, const void* w_planes, const float* bias,
                             int width, int stride, int pad_left, int relu, const st_tensor3* y, void* y_planes,
                             void* stream);
/* dx = conv^T(dz) * [act > 0] (stride-1 layers); wt_planes = planes of st_filters_flip_transpose_f32's output;
 * the optional workspace (st_exp_conv1d_bwd_data_bf16x6_ws bytes) enables the split reduction */
size_t st_exp_conv1d_bwd_data_bf16x6_ws(const st_tensor3* dz, const st_tensor3* dx, int width);
int st_exp_conv1d_bwd_data_bf16x6(const st_tensor3* dz, const void* dz_planes, const void* wt_planes, int width,
                                  int pad_left, const st_tensor3* act, const st_tensor3* dx, void* dx_planes,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* reduction-major planes [c_pitch][batch * tq] of rows [row0, row0 + rows) of every utterance of t */
int st_exp_transpose_split3_bf16(const st_tensor3* t, int row0, int rows, int tq, size_t plane_elems, void* planes,
                                 void* stream);
/* filter gradient of a stride-1 layer from the reduction-major planes of its input and of dz */
int st_exp_conv1d_bwd_filter_bf16x6(const void* xt_planes, const void* dzt_planes, int batch, int tq, int width,
                                    int cin_pitch, int x_first_row, int cout, float* dpacked, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHT_HIP_EXPERIMENTAL_H_ */
