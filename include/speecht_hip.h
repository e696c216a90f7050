/* speecht_hip.h -- C ABI of libspeecht_hip.so: the MI355X (gfx950) kernels behind the
 * speechT Wav2Letter training / inference step.
 *
 * The reference (louiskirsch/speechT) has no FFI of its own: its hot path is a list of
 * TensorFlow-1 / librosa op call sites.  Each entry point below replaces one of those call
 * sites (file:line relative to the reference root) and is what a reference-side ctypes
 * binding would bind (INTEGRATION.md shows the stub).  Conventions:
 *   - plain `extern "C"`, pointers + sizes only, no torch / C++ types;
 *   - every pointer is a caller-owned DEVICE pointer unless the name says `host_`;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue
 *     work, they never synchronise and never allocate;
 *   - return 0 on success, a negative ST_E* code otherwise; st_last_error() gives the text
 *     (thread-local);
 *   - workspaces are explicit: query the size, pass the buffer.
 *
 * Activation layout ("padded NWC"): the reference's (batch, time, channel) tensors
 * (speech_model.py:44) are stored with zero halo rows around every utterance and the channel
 * pitch rounded up to 16 floats, so that tf.nn.conv1d's SAME padding (speech_model.py:155)
 * becomes plain in-bounds reads and every im2col row is one contiguous, 64-byte aligned span.
 */
#ifndef SPEECHT_HIP_H
#define SPEECHT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ST_OK 0
#define ST_EINVAL (-1)   /* bad argument / layout precondition violated */
#define ST_ELAUNCH (-2)  /* HIP launch error */
#define ST_EWORKSPACE (-3)
#define ST_ECOMM (-4)    /* RCCL unavailable or a collective failed */

/* element (b, t, c) lives at base[((int64)b * t_pitch + halo + t) * c_pitch + c];
 * rows outside [0, frames) and channels in [channels, c_pitch) must hold zeros. */
typedef struct st_tensor3 {
  float* base;
  int32_t batch;
  int32_t frames;
  int32_t channels;
  int32_t halo;     /* zero rows in front of frame 0 */
  int32_t t_pitch;  /* rows per utterance, halo + frames + trailing zero rows */
  int32_t c_pitch;  /* floats per row, multiple of 16 */
} st_tensor3;

int st_version(void);
const char* st_last_error(void);

/* ---- diagnostics (no reference counterpart) ----------------------------------------------
 * Launch trace: between st_trace_begin() and st_trace_end() every entry point appends one text line per
 * kernel launch naming the variant and split policy it chose (e.g. "gemm_nn<128,128,2,2,fast> epi=1
 * splits=2 M=16032 Np=256 Kp=64512").  st_trace_end copies the text to a HOST buffer (truncating) and
 * returns the bytes needed including the terminator.  Used by the parity tests to assert which kernels a
 * full-size step really ran.  Every line of a matrix-pipe launch carries "gflop=<executed GFLOP, padding included>".
 * st_trace_begin_timed() additionally launches every traced kernel through hipExtLaunchKernel with a start and a stop
 * event (the dispatch's own begin / end time stamps, as a profiler's kernel trace shows them; no marker between
 * launches); st_trace_end then waits for them and appends " ms=<duration>" to each line -- per-launch times INSIDE the
 * real launch sequence of a step, side streams and all (bench.py's in-step roofline).
 * Tuning overrides for performance experiments ("gemm_tile", "gemm_splits", "fwd_splits", "xcd_gm",
 * "no_fast", "bf16_tile", "bf16_wgrad_splits", "bf16_sched", "streamk", "transform_wgs", "bf16_wgrad_target", "streamk_slots"); value 0 restores the library's own policy.
 * The launch path never reads the environment. */
int st_trace_begin(void);
int st_trace_begin_timed(void);
/* Timed mode only times the launches whose trace line contains `text` (NULL or "": all).  A timed launch costs the stream
 * ~7 us (hipExtLaunchKernel with events gives up the back-to-back dispatch), so timing ONE kernel's ten launches per step
 * leaves the step as it is (+1 %), timing all ~100 stretches it by 10 %. */
int st_trace_timed_filter(const char* text);
size_t st_trace_end(char* host_buf, size_t capacity);
int st_set_tuning(const char* name, int value);
/* CRC-32C of a HOST buffer, continuing from `crc` (0 to start): the checksum TensorFlow's checkpoint bundles carry
 * (speech_model.py:122 tf.train.Saver; read and written by speecht_amd/tf_checkpoint.py). */
uint32_t st_host_crc32c(const void* host_data, size_t n, uint32_t crc);

/* ---- filter packing -------------------------------------------------------------------
 * Reference filters are [W, Cin, Cout] (speech_model.py:148-151; the `export --weights`
 * layout, exporting.py:30-40).  The kernels use the row-major GEMM operand
 * packed[k_pad][n_pad], k = w * cin_pitch + c, zero in all padding. */
int st_packed_dims(int width, int cin_pitch, int cout, int* k_valid, int* k_pad, int* n_pad);
int st_pack_filters_f32(const float* filters, int width, int cin, int cout, int cin_pitch,
                        float* packed, void* stream);
int st_unpack_filters_f32(const float* packed, int width, int cin, int cout, int cin_pitch,
                          float* filters, void* stream);
/* packedT[(w' * cout_pitch + o)][c] = packed[((W-1-w') * cin_pitch + c)][o]: the operand that
 * turns back-prop to the layer input into the same implicit-GEMM convolution. */
int st_filters_flip_transpose_f32(const float* packed, int width, int cin, int cout,
                                  int cin_pitch, int cout_pitch, float* packed_t, void* stream);

/* ---- K5-K7: tf.nn.conv1d('SAME') + bias_add + relu (speech_model.py:155,173,177) -------
 * y[b,t,o] = act(bias[o] + sum_{w,c} x[b, t*stride + w - pad_left, c] * F[w,c,o]).
 * Requires x->halo >= pad_left and enough trailing halo; bias has n_pad floats. */
int st_conv1d_nwc_fwd_f32(const st_tensor3* x, const float* packed, const float* bias, int width,
                          int stride, int pad_left, int relu, const st_tensor3* y, void* stream);
/* Same, with a workspace (st_conv1d_fwd_ws bytes; 0 when the shape does not split).  With few output rows
 * (live / single-utterance inference) the reduction is split over the idle CUs and summed in a fixed
 * order; without a large enough workspace the call is identical to st_conv1d_nwc_fwd_f32. */
size_t st_conv1d_fwd_ws(const st_tensor3* x, const st_tensor3* y, int width);
int st_conv1d_nwc_fwd_ws_f32(const st_tensor3* x, const float* packed, const float* bias, int width,
                             int stride, int pad_left, int relu, const st_tensor3* y, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- frequency-domain form of the same three operations (csrc/conv_fft.hip) ---------------------------------
 * Time is cut into blocks of 64 frames, every block is taken to the frequency domain by a length-(63 + W) DFT, the W
 * taps become one complex channel-contraction per frequency bin -- run as real GEMMs on the exact-fp32 MFMA kernel --
 * and the result comes back by an inverse DFT fused with the bias / ReLU / mask epilogue: for the model's 32-tap
 * 250 -> 2000 layer (speech_model.py:285; 66 % of the step's MACs) ~10x fewer multiplications than the W-tap form,
 * same fp32 arithmetic class (tests/test_gpu_fft_conv.py).  stride 1, W <= 33; the output channels must pack to a
 * multiple of 128 (and the input channels too for back-prop to the input).  A stride-2 layer of even-ish width runs
 * through the same entry points on its polyphase view: the input read as [B][T/2][2 * c_pitch] (frame pairs as
 * channels), W/2 + 1 taps, the packed filters shifted by one c_pitch block (engine.py does this for
 * the model's first layer).
 *   tables    st_conv1d_fft_table_floats() floats, filled once per (width, pad_left) by st_conv1d_fft_tables_f32
 *   gfwd      the filter spectra as a GEMM operand (st_conv1d_fft_filter_floats floats), rebuilt by
 *             st_conv1d_fft_filters_f32 whenever the weights change; the forward pass multiplies by it, back-prop to the
 *             input by its transpose (read in place: no second set of spectra, no flipped copy of the weights)
 *   sf        spectra of the layer input (st_conv1d_fft_sf_floats floats), written by the forward call and read by
 *             the filter-gradient call
 *   zf        spectra of the gradient wrt the layer output (st_conv1d_fft_zf_floats floats), written by
 *             st_conv1d_fft_dz_spectra_f32, read by both gradient calls
 *   workspace st_conv1d_fft_ws bytes, scratch of one call, headed by st_gemm_nn_batched_ws_f32's area (one workspace per stream) */
int st_conv1d_fft_plan(int width, int frames, int batch, int* n, int* valid, int* blocks, int* bins, int* rows_pad);
/* the per-bin products themselves: `batches` independent row-major fp32 GEMMs C[i] = A[i] * B[i] (A [m][lda], B [k][n],
 * C [m][ldc]; k a multiple of 32, n of 128; strides in floats) on the convolution MFMA kernel, bin i on XCD i % 8 */
int st_gemm_nn_batched_f32(const float* a, int64_t lda, int64_t a_batch, const float* b, int64_t b_batch, float* c,
                           int64_t ldc, int64_t c_batch, int m, int k, int n, int batches, void* stream);
/* The same with st_gemm_nn_batched_ws_bytes() bytes of scratch: a launch whose 64 x 128 tiles would load the CUs unevenly
 * (the 7-tap layers: 36 bins x 16 tiles = 576 workgroups, three on a quarter of the CUs, two on the rest) then runs as ONE
 * persistent launch that deals the (bin, tile, k-tile) list in equal runs to 512 (or 768) workgroups; a tile cut into pieces
 * is summed head + next + ... by the workgroup that holds its start (bit-reproducible, no atomics on data).  The scratch may
 * hold any content (its first st_gemm_nn_batched_ctrl_bytes() bytes are flags tagged with a per-launch epoch and reset by
 * their reader); one scratch per stream.  The frequency-domain entry points below carry this area at the head of their
 * workspace. */
size_t st_gemm_nn_batched_ws_bytes(void);
size_t st_gemm_nn_batched_ctrl_bytes(void);
/* Lost hand-offs of the persistent launches since the library was loaded.  A reader whose bounded poll runs out (a producer
 * workgroup that never published its partial tile) writes NaN into its tile -- never a sum with an unpublished partial -- and
 * counts here.  st_streamk_lost_ptr: the device address of the 32-bit count; st_streamk_lost_count: a synchronous read (default
 * stream); st_streamk_lost_fetch_async: for a caller that already copies status words back after a step (engine.fetch_losses). */
int st_streamk_lost_ptr(void** device_word);
int st_streamk_lost_count(unsigned* count);
/* the count copied to `host_word` (pinned host memory) behind everything enqueued on `stream` so far */
int st_streamk_lost_fetch_async(unsigned* host_word, void* stream);
int st_gemm_nn_batched_ws_f32(const float* a, int64_t lda, int64_t a_batch, const float* b, int64_t b_batch, float* c,
                              int64_t ldc, int64_t c_batch, int m, int k, int n, int batches, void* workspace,
                              size_t workspace_bytes, void* stream);
/* C[i] = A[i] * Bt[i]^T with the second operand given TRANSPOSED, Bt [n][k] (k contiguous): the products of back-prop to the
 * input, which read the forward filter spectra in place (X = Z gfwd^T) */
int st_gemm_nn_batched_bt_ws_f32(const float* a, int64_t lda, int64_t a_batch, const float* bt, int64_t bt_batch, float* c,
                                 int64_t ldc, int64_t c_batch, int m, int k, int n, int batches, void* workspace,
                                 size_t workspace_bytes, void* stream);
/* out[i] = A[i]^T * Z[i]: A [m][lda] (k columns), Z [m][ldz] (n columns), out [k][n]; the reduction runs over the m rows
 * (m a multiple of 32, k and n of 128) -- the lag products of the filter gradient */
int st_gemm_tn_batched_f32(const float* a, int64_t lda, int64_t a_batch, const float* z, int64_t ldz, int64_t z_batch,
                           float* out, int64_t out_batch, int m, int k, int n, int batches, void* stream);
/* ... with the second operand SHARED by groups of 2^z_batch_shift consecutive products (product b reads Z of batch
 * b >> z_batch_shift): the real and the imaginary lag products of a bin -- over the input spectra and their rotated copy,
 * rows read at half length -- both against that bin's gradient spectra */
int st_gemm_tn_batched_shared_f32(const float* a, int64_t lda, int64_t a_batch, const float* z, int64_t ldz, int64_t z_batch,
                                  float* out, int64_t out_batch, int m, int k, int n, int batches, int z_batch_shift, void* stream);
/* A bin's COMPLEX product in three real products instead of four (Gauss: (a + ib)(c + id) = (k1 - k3) + i (k1 + k2), k1 = c (a + b),
 * k2 = a (d - c), k3 = b (c + d)), the shared product computed once per output tile (round 6; the 32-tap layer's per-bin products,
 * 25 % fewer multiplications).  Operand planes A_p = a + a_off[p], B_p = b + b_off[p] (p = 0, 1, 2; offsets in floats; b rows ldb
 * apart, or with b_transposed != 0 the planes hold B_p^T, [n][ldb]):
 *     c[i][row][col]          = A_0 B_0 + A_1 B_1          c[i][row][c_off2 + col] = A_0 B_0 + A_2 B_2
 * k = reduction length of ONE product (a multiple of 64), n a multiple of 128.  The sums and signs live in the planes: the
 * transforms write [S_r + S_i | S_i | S_r | S_i - S_r] / [Z_r + Z_i | Z_r | Z_i] rows, the filter spectra are the planes
 * G_r, G_i - G_r, -(G_r + G_i) (st_conv1d_fft_three_products says which layers: conv_fft.hip g3_form). */
int st_gemm_nn_g3_batched_f32(const float* a, int64_t lda, int64_t a_batch, const int64_t* a_off, const float* b, int64_t ldb,
                              int64_t b_batch, const int64_t* b_off, int b_transposed, float* c, int64_t ldc, int64_t c_batch,
                              int64_t c_off2, int m, int k, int n, int batches, void* stream);
/* ... and the lag products of the filter gradient: out[i] = A_0^T Z_0 + A_1^T Z_1 and, out_part floats behind it,
 * A_2^T Z_2 - A_0^T Z_0 ([k][n] each), the reduction over the m rows (m a multiple of 32, k and n of 128) */
int st_gemm_tn_g3_batched_f32(const float* a, int64_t lda, int64_t a_batch, const int64_t* a_off, const float* z, int64_t ldz,
                              int64_t z_batch, const int64_t* z_off, float* out, int64_t out_batch, int64_t out_part, int m, int k,
                              int n, int batches, void* stream);
/* which form a frequency-domain layer's per-bin products take: 0 four real products; 2 three (its sf / zf / gfwd buffers then
 * hold the layouts above -- same sizes from the st_conv1d_fft_*_floats functions); 1 three-part gradient spectra rows read by the
 * four-product kernels (an input whose spectra do not tile the three-product kernels).  st_set_tuning("no_g3", 1): always 0. */
int st_conv1d_fft_three_products(int width, int cin_pitch, int cout);
size_t st_conv1d_fft_table_floats(void);
int st_conv1d_fft_tables_f32(int width, int pad_left, float* tables, size_t table_floats, void* stream);
size_t st_conv1d_fft_filter_floats(int width, int cin_pitch, int cout);
int st_conv1d_fft_filters_f32(const float* packed, int width, int cin, int cout, int cin_pitch, const float* tables, float* gfwd,
                              void* stream);
size_t st_conv1d_fft_sf_floats(const st_tensor3* x, const st_tensor3* y, int width);
size_t st_conv1d_fft_zf_floats(const st_tensor3* dz, int width);
size_t st_conv1d_fft_ws(const st_tensor3* x, const st_tensor3* y, int width);
int st_conv1d_nwc_fwd_fft_f32(const st_tensor3* x, const float* gfwd, const float* bias, int width, int pad_left, int relu,
                              const st_tensor3* y, const float* tables, float* sf, void* workspace,
                              size_t workspace_bytes, void* stream);
/* The same for a CHAIN of frequency-domain layers (round 4): sf_ready != 0 -- `sf` already holds this layer's input spectra (the
 * previous layer's call wrote them); next_tables / next_sf / next_width / next_pad_left -- the next layer of the chain: when the
 * shapes allow (at most 8 blocks of 64 frames per utterance, batch x blocks a multiple of 128, the next layer's window reaching at
 * most 4 frames into either neighbour block: the model's 7-tap layers at utterances of up to 10.2 s) the inverse transform hands its
 * frames to the next layer's forward transform in registers and writes next_sf; *next_sf_written says whether it did. */
int st_conv1d_nwc_fwd_fft_chain_f32(const st_tensor3* x, const float* gfwd, const float* bias, int width, int pad_left, int relu,
                                    const st_tensor3* y, const float* tables, float* sf, int sf_ready, const float* next_tables,
                                    float* next_sf, int next_width, int next_pad_left, int* next_sf_written, void* workspace,
                                    size_t workspace_bytes, void* stream);
int st_conv1d_fft_dz_spectra_f32(const st_tensor3* dz, int width, const float* tables, float* zf, void* stream);
/* dbias[o] = sum_{b,t} dz[b,t,o] read off bin 0 of the spectra zf (npad floats written, pads zero): the bias gradient of a
 * frequency-domain layer without another pass over dz */
int st_conv1d_fft_bias_grad_f32(const st_tensor3* dz, int width, const float* zf, float* dbias, void* stream);
int st_conv1d_nwc_bwd_data_fft_f32(const st_tensor3* dz, const float* zf, const float* gfwd, int width, int pad_left,
                                   const st_tensor3* act, const st_tensor3* dx, const float* tables, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* The same for a CHAIN (round 4): below_tables / below_zf / below_width -- the frequency-domain layer below, whose output gradient
 * dx is.  When the shapes allow (at most 8 blocks of 64 frames per utterance, batch x blocks a multiple of 128, at most 4 frames
 * of a block's window in either neighbour block) ONE launch inverts every block at its whole window, overlap-adds through LDS,
 * masks, stores dx and -- the frames still in registers -- writes the layer below's dz spectra to below_zf (*below_zf_written = 1:
 * skip st_conv1d_fft_dz_spectra_f32 for that layer).  below_* may be NULL / 0: the window-form inverse alone. */
int st_conv1d_nwc_bwd_data_fft_chain_f32(const st_tensor3* dz, const float* zf, const float* gfwd, int width, int pad_left,
                                         const st_tensor3* act, const st_tensor3* dx, const float* tables, const float* below_tables,
                                         float* below_zf, int below_width, int* below_zf_written, void* workspace,
                                         size_t workspace_bytes, void* stream);
int st_conv1d_nwc_bwd_filter_fft_f32(const st_tensor3* x, const st_tensor3* dz, const float* sf, const float* zf, int width,
                                     const float* tables, float* dpacked, void* workspace, size_t workspace_bytes,
                                     void* stream);

/* ---- the same three operations with the per-bin products on the bf16 matrix pipe (round 4) --------------------------------
 * planes = 1: BASELINE configs[3] arithmetic -- the tensors are read / written in their bf16 form (x_bf16, y_bf16, ...: same
 *             padded NWC geometry as the descriptor, 2-byte elements), spectra and filter spectra are one bf16 plane; the
 *             DFTs, the products' accumulation and the inverse DFTs are fp32.
 * planes = 3: fp32 tensors (the *_bf16 arguments NULL); every spectrum value is split exactly into three bf16 planes and a
 *             product taken as the six largest cross terms with fp32 accumulation: at least fp32-FMA accuracy at the bf16
 *             pipe's rate.
 * Buffers (elements of 2 bytes unless noted): sf_planes / zf_planes = planes x the fp32 form's float count
 * (st_conv1d_fft_sf_floats / _zf_floats); g_planes, gt_planes = planes x st_conv1d_fft_filter_plane_elems each (the filter
 * spectra and their per-bin transposes, rebuilt by st_conv1d_fft_filters_planes whenever the weights change); dc = fp32
 * [rows_pad][n_pad]: the blocks' frame sums (bin 0) kept in fp32 for the bias gradient (st_conv1d_fft_bias_grad_dc_f32 with
 * rows = batch x blocks); workspace st_conv1d_fft_planes_ws bytes. */
size_t st_conv1d_fft_filter_plane_elems(int width, int cin_pitch, int cout);
int st_conv1d_fft_filters_planes(const float* packed, int width, int cin, int cout, int cin_pitch, const float* tables, void* g_planes,
                                 void* gt_planes, int planes, void* stream);
size_t st_conv1d_fft_planes_ws(const st_tensor3* x, const st_tensor3* y, int width, int planes);
int st_conv1d_nwc_fwd_fft_planes(const st_tensor3* x, const void* x_bf16, const void* gt_planes, const float* bias, int width,
                                 int pad_left, int relu, const st_tensor3* y, void* y_bf16, const float* tables, void* sf_planes,
                                 int planes, void* workspace, size_t workspace_bytes, void* stream);
int st_conv1d_fft_dz_spectra_planes(const st_tensor3* dz, const void* dz_bf16, int width, const float* tables, void* zf_planes,
                                    int planes, float* dc, void* stream);
int st_conv1d_fft_bias_grad_dc_f32(const float* dc, int rows, int channels, int n_pad, float* dbias, void* stream);
int st_conv1d_nwc_bwd_data_fft_planes(const st_tensor3* dz, const void* zf_planes, const void* g_planes, int width, int pad_left,
                                      const st_tensor3* act, const void* act_bf16, const st_tensor3* dx, void* dx_bf16,
                                      const float* tables, int planes, void* workspace, size_t workspace_bytes, void* stream);
int st_conv1d_nwc_bwd_filter_fft_planes(const st_tensor3* x, const st_tensor3* dz, const void* sf_planes, const void* zf_planes, int width,
                                        const float* tables, float* dpacked, int planes, void* workspace, size_t workspace_bytes,
                                        void* stream);

/* ---- K11: back-prop (optimizer.compute_gradients, speech_model.py:78) -------------------
 * bwd_data: dx[b,t,c] = mask * sum_{w,o} dz[b, t + pad_left - w, o] * F[w,c,o]   (stride 1),
 *   mask = (act[b,t,c] > 0) when act != NULL (tf.nn.relu's gradient of the producing layer).
 *   packed_t comes from st_filters_flip_transpose_f32; dz->halo >= width-1-pad_left.  The workspace
 *   (st_conv1d_bwd_data_ws bytes, may be 0 / NULL) enables split-K for long reductions.
 * bwd_filter: dF[w,c,o] = sum_{b,t} x[b, t*stride + w - pad_left, c] * dz[b,t,o] in packed
 *   layout [k_pad][n_pad]; dbias[o] = sum_{b,t} dz[b,t,o].  Workspace: st_conv1d_bwd_filter_ws. */
size_t st_conv1d_bwd_data_ws(const st_tensor3* dz, const st_tensor3* dx, int width);
int st_conv1d_nwc_bwd_data_f32(const st_tensor3* dz, const float* packed_t, int width,
                               int pad_left, const st_tensor3* act, const st_tensor3* dx,
                               void* workspace, size_t workspace_bytes, void* stream);
/* Same, and additionally dbias_dx[n_pad of dx->channels] = column sums of the dx just written: the bias gradient
 * of the layer BELOW (its pre-activation gradient is dx), collected in the epilogue of the kernel that writes dx
 * instead of by a second pass over up to 129 MB.  Workspace: st_conv1d_bwd_data_bias_ws bytes. */
size_t st_conv1d_bwd_data_bias_ws(const st_tensor3* dz, const st_tensor3* dx, int width);
int st_conv1d_nwc_bwd_data_bias_f32(const st_tensor3* dz, const float* packed_t, int width, int pad_left,
                                    const st_tensor3* act, const st_tensor3* dx, float* dbias_dx,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* The same for a ONE-TAP layer (speech_model.py:288,292: the two 1 x 1 layers on top) without any derived operand:
 * dx = dz W^T with `packed` the layer's own forward filters [cin_pitch][n_pad(cout)], read as a transposed operand.
 * Needs dz->c_pitch % 32 == 0 and an input width that packs to a multiple of 128; dbias_dx may be NULL.  Workspace as above
 * with width = 1. */
int st_conv1d_1tap_bwd_data_bias_f32(const st_tensor3* dz, const float* packed, const st_tensor3* act, const st_tensor3* dx,
                                     float* dbias_dx, void* workspace, size_t workspace_bytes, void* stream);
size_t st_conv1d_bwd_filter_ws(const st_tensor3* x, const st_tensor3* dz, int width);
/* bias gradient alone: dbias[o] = sum_{b,t} dz[b,t,o]  (n_pad floats) */
size_t st_bias_grad_ws(const st_tensor3* dz);
int st_bias_grad_f32(const st_tensor3* dz, float* dbias, void* workspace, size_t workspace_bytes, void* stream);
int st_conv1d_nwc_bwd_filter_f32(const st_tensor3* x, const st_tensor3* dz, int width, int stride,
                                 int pad_left, float* dpacked, float* dbias, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* ---- K9-K10: tf.nn.ctc_loss + reduce_mean gradient (speech_model.py:74-75) --------------
 * logits: [B, T', C] padded NWC (halo 0), blank = C-1 (C <= 32).  labels CSR: label_offsets
 * [B+1], label_ids [label_offsets[B]].  seq_lens[b] = frames to use (reference passes
 * sequence_lengths // 2).  Outputs: loss[b] = -log p(l|x); grad = grad_scale * dloss/dlogits
 * (0 for t >= seq_lens[b]); status[b] != 0 when the label does not fit ("Not enough time for
 * target transition sequence") -- then loss = +inf and grad = 0.  The forward-backward lattice is kept as mantissa * 2^exponent
 * with an integer exponent per state (not in log space): exact range, no transcendental on the sequential chain; a class more than
 * 2^-30000 below its frame's best counts as impossible.  Workspace: st_ctc_ws bytes (log-softmax, emission factors, two lattices). */
size_t st_ctc_ws(int batch, int frames, int max_label_len);
int st_ctc_loss_grad_f32(const st_tensor3* logits, const int32_t* label_ids,
                         const int32_t* label_offsets, const int32_t* seq_lens, int max_label_len,
                         float grad_scale, float* loss, const st_tensor3* grad, int32_t* status,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The same with the loss as a (hi, lo) float pair: loss[b] is the fp32 value above (what TF's fp32 op returns), loss_lo[b]
 * (may be null) the part of -log p that fp32 cannot hold at that magnitude -- the kernel knows log p in double (the lattice
 * rounds 6e-8 relative per step, one fp32 ulp of a loss of 1 239 is 1.2e-4); (double)loss[b] + loss_lo[b] is the loss to
 * ~1e-6 of the float64 oracle.  speech_model.py:74-75: avg_loss = mean of these. */
int st_ctc_loss_grad_hilo_f32(const st_tensor3* logits, const int32_t* label_ids,
                              const int32_t* label_offsets, const int32_t* seq_lens, int max_label_len,
                              float grad_scale, float* loss, float* loss_lo, const st_tensor3* grad, int32_t* status,
                              void* workspace, size_t workspace_bytes, void* stream);

/* ---- K14: tf.nn.ctc_greedy_decoder(merge_repeated) (speech_model.py:113-115) ------------
 * ids [B][max_out] int32 (max_out >= frames), out_lens [B], neg_sum_logits [B]. */
int st_ctc_greedy_decode(const st_tensor3* logits, const int32_t* seq_lens, int merge_repeated,
                         int32_t* ids, int max_out, int32_t* out_lens, float* neg_sum_logits,
                         void* stream);

/* ---- LM-free CTC prefix beam search, top path (BASELINE config 5) -----
 * The reference only reaches a beam search through its KenLM TensorFlow fork
 * (speech_model.py:101-111: beam_width=100, merge_repeated=False, top_paths=1); this is the stock
 * tf.nn.ctc_beam_search_decoder recursion without a scorer.  blank = C-1, C <= 32, beam <= 128 (one wavefront per
 * utterance; beams above 64 run a two-entries-per-lane instantiation with an exact W-th-largest selection bound).
 * ids [B][max_out] int32 (labels beyond max_out are dropped, out_lens still reports the true
 * length), log_prob [B] = ln p(top prefix) under the per-frame softmax.  The labels are the top prefix itself, i.e.
 * merge_repeated=False as the reference asks (speech_model.py:110); collapsing repeats is a host-side option. */
size_t st_ctc_beam_ws(int batch, int frames, int beam_width);
int st_ctc_beam_search_decode(const st_tensor3* logits, const int32_t* seq_lens, int beam_width,
                              int32_t* ids, int max_out, int32_t* out_lens, float* log_prob,
                              void* workspace, size_t workspace_bytes, void* stream);
/* ... with the reference's input transform: input_transform = 1 searches on log10(softmax(logits) + 1e-8), what
 * speech_model.py:102 hands the decoder (which normalises it per frame like any input); 0 = the logits themselves. */
int st_ctc_beam_search_decode_ex(const st_tensor3* logits, const int32_t* seq_lens, int beam_width, int input_transform,
                                 int32_t* ids, int max_out, int32_t* out_lens, float* log_prob,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- K12-K13: clip_by_global_norm + AdamOptimizer(epsilon outside) (speech_model.py:77-82)
 * Flat fp32 buffers of n floats.  stats (device, 2 floats) receives {global_norm, scale}.
 * lr_t = lr * sqrt(1-beta2^t)/(1-beta1^t) is computed by the caller (host, double).
 * p -= lr_t * m / (sqrt(v) + eps) after m,v updates with g * scale. */
size_t st_global_norm_ws(size_t n);
int st_global_norm_clip_adam_f32(float* params, const float* grads, float* m, float* v, size_t n,
                                 float clip_norm, float lr_t, float beta1, float beta2, float eps,
                                 float* stats, void* workspace, size_t workspace_bytes,
                                 void* stream);
/* Gated form.  tf.nn.ctc_loss rejects a batch with an utterance that cannot be aligned (InvalidArgument,
 * speech_model.py:74) and the failing sess.run then touches no variable (speech_model.py:82).  Here the CTC
 * kernel reports such utterances in `status`; st_ctc_status_gate_f32 folds the status words into one device float
 * (the number of bad utterances; data-parallel training all-reduces it with the gradients), and the update
 * becomes a no-op -- params, m and v untouched, stats still written -- when *gate != 0.  gate == NULL: ungated.
 * The update is also skipped when the global norm is not finite (stats[0] then holds the NaN / Inf): a poisoned gradient never
 * reaches params, m or v. */
int st_ctc_status_gate_f32(const int32_t* status, int batch, float* gate, void* stream);
/* ... and this rank's share of the GLOBAL mean loss (speech_model.py:75 reduce_mean) in gate[1] = sum_b (loss_hi[b] + loss_lo[b]) *
 * loss_scale (summed in double; loss_lo may be NULL; loss_scale = 1 / global batch): gate[0..1] sit inside the first gradient
 * bucket, so data-parallel training gets the mean loss out of the gradient exchange itself -- no scalar all-reduce per step. */
int st_ctc_status_gate_loss_f32(const int32_t* status, int batch, const float* loss_hi, const float* loss_lo, float loss_scale,
                                float* gate, void* stream);
int st_global_norm_clip_adam_gated_f32(float* params, const float* grads, float* m, float* v, size_t n,
                                       float clip_norm, float lr_t, float beta1, float beta2, float eps,
                                       float* stats, const float* gate, void* workspace,
                                       size_t workspace_bytes, void* stream);
/* norm only (stats[0] = ||g||, stats[1] = clip/max(norm,clip)); used for reporting */
int st_global_norm_f32(const float* grads, size_t n, float clip_norm, float* stats,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- K1-K3: calc_power_spectrogram (preprocessing.py:36-58) -----------------------------
 * audio: concatenated float samples, sample_offsets [n_utts+1].  For every utterance:
 * reflect-pad, Hann-512 STFT with hop, |.|^2, mel_basis [n_mels][n_fft/2+1], power_to_db with
 * ref = max, top_db 80, then (x-mean)/std over the whole matrix, written transposed as
 * out[frame_offsets[u] + t][n_mels] (frames = 1 + len/hop).  max_samples = longest utterance
 * (each must exceed n_fft/2 samples, numpy reflect padding); total_frames = frame_offsets[n_utts]. */
/* Planned form: the sparse mel filterbank is compiled once per (sample rate, n_mels) into a device-side plan
 * (st_melspec_plan_bytes() bytes, 16-byte aligned) and st_melspec_planned_f32 runs without touching mel_basis;
 * st_melspec_f32 is the one-call form that rebuilds the plan inside its workspace on every call. */
size_t st_melspec_plan_bytes(void);
int st_melspec_plan_f32(const float* mel_basis, int n_mels, int n_fft, void* plan, size_t plan_bytes, void* stream);
int st_melspec_planned_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                           const void* plan, int n_mels, int n_fft, int hop, const int64_t* frame_offsets,
                           int64_t total_frames, float* out, void* workspace, size_t workspace_bytes, void* stream);
size_t st_melspec_ws(int n_utts, int64_t total_frames, int n_mels);
int st_melspec_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                   const float* mel_basis, int n_mels, int n_fft, int hop,
                   const int64_t* frame_offsets, int64_t total_frames, float* out, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---- bf16 activations (BASELINE config 4: "bf16 activations / fp32 CTC") ------------------
 * Additional symbols, not a second code path for fp32.  Activations and activation gradients are
 * stored in HBM as bf16 (round-to-nearest-even) with the SAME padded NWC geometry as the fp32
 * tensors: every entry point takes the st_tensor3 for the geometry (its `base` is ignored) and a
 * separate pointer to the bf16 data.  Filters stay fp32 masters (packed layout above) with a
 * transposed bf16 copy [n_pad][k_pad] made by st_filters_bf16 (k_pad, n_pad from st_packed_dims;
 * the back-prop operand comes from st_filters_bwd_bf16).
 * Accumulation, bias, ReLU, the logits of the last layer, CTC, clip and Adam are fp32.
 *   forward : y = relu?(conv(x) + bias) -> y_bf16 and/or y_f32 (either may be NULL)
 *   bwd-data: dx = conv^T(dz) * [act > 0]  (stride-1 layers; act = the layer input, NULL = no mask);
 *             workspace of st_conv1d_bwd_data_bf16_ws bytes (0 for most shapes: only long reductions
 *             on few output tiles are split into fp32 partial sums)
 *   bwd-filt: dpacked [k_pad][n_pad] (rows < width*c_pitch written) and dbias [n_pad], fp32;
 *             stride 1 or 2; workspace from st_conv1d_bwd_filter_bf16_ws. */
int st_cast_bf16(const float* src, size_t n, void* dst, void* stream);
int st_filters_bf16(const float* packed, int k_pad, int n_pad, void* wt, void* stream);
/* back-prop operand [n_pad(cin)][k_pad(width*cout_pitch)] (zero-initialised by the caller) straight from
 * the packed filters: out[c][(W-1-w)*cout_pitch + o] = packed[w*cin_pitch + c][o] */
int st_filters_bwd_bf16(const float* packed, int width, int cin, int cout, int cin_pitch, int cout_pitch,
                        void* wtt, void* stream);
int st_conv1d_nwc_fwd_bf16(const st_tensor3* x, const void* x_bf16, const void* wt_bf16,
                           const float* bias, int width, int stride, int pad_left, int relu,
                           const st_tensor3* y, void* y_bf16, float* y_f32, void* stream);
/* same with a workspace (st_conv1d_fwd_bf16_ws bytes): few output rows split the reduction, as in
 * st_conv1d_nwc_fwd_ws_f32 */
size_t st_conv1d_fwd_bf16_ws(const st_tensor3* x, const st_tensor3* y, int width);
int st_conv1d_nwc_fwd_ws_bf16(const st_tensor3* x, const void* x_bf16, const void* wt_bf16,
                              const float* bias, int width, int stride, int pad_left, int relu,
                              const st_tensor3* y, void* y_bf16, float* y_f32, void* workspace,
                              size_t workspace_bytes, void* stream);
size_t st_conv1d_bwd_data_bf16_ws(const st_tensor3* dz, const st_tensor3* dx, int width);
int st_conv1d_nwc_bwd_data_bf16(const st_tensor3* dz, const void* dz_bf16, const void* wtt_bf16,
                                int width, int pad_left, const st_tensor3* act, const void* act_bf16,
                                const st_tensor3* dx, void* dx_bf16, void* workspace,
                                size_t workspace_bytes, void* stream);
size_t st_conv1d_bwd_filter_bf16_ws(const st_tensor3* x, const st_tensor3* dz, int width, int stride,
                                    int pad_left);
int st_conv1d_nwc_bwd_filter_bf16(const st_tensor3* x, const void* x_bf16, const st_tensor3* dz,
                                  const void* dz_bf16, int width, int stride, int pad_left,
                                  float* dpacked, float* dbias, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* The same for STRIDE-1 layers without the reduction-major copies: both planes are staged as they lie in HBM and transposed on
 * their way from LDS into the matrix registers (ds_read_b64_tr_b16, csrc/wgrad_tr_bf16.hip).  Preconditions beyond the entry
 * point above: stride 1, x->t_pitch == dz->t_pitch (always true for a 'SAME' layer laid out as the header describes), and both
 * planes READABLE AND ZERO for st_conv1d_bwd_filter_tr_bf16_slack_rows() rows behind their last row.  _ws returns 0 for a
 * geometry the entry point does not take. */
int st_conv1d_bwd_filter_tr_bf16_slack_rows(void);
size_t st_conv1d_bwd_filter_tr_bf16_ws(const st_tensor3* x, const st_tensor3* dz, int width, int stride, int pad_left);
int st_conv1d_nwc_bwd_filter_tr_bf16(const st_tensor3* x, const void* x_bf16, const st_tensor3* dz, const void* dz_bf16,
                                     int width, int stride, int pad_left, float* dpacked, float* dbias, void* workspace,
                                     size_t workspace_bytes, void* stream);

/* ---- gradient exchange (RCCL over xGMI) ---------------------------------
 * The reference is single-replica (training.py:46); data parallelism follows from
 * speech_model.py:75-82 (mean loss over the batch, clip and Adam on the mean gradient).  One
 * communicator per process.  Rank 0 creates the id, the host moves its bytes to the other ranks,
 * every rank calls st_comm_init (collective).  st_allreduce_f32 SUMS in place, asynchronously on
 * `stream`; the bucket form groups several slices of one flat buffer into a single RCCL launch. */
int st_comm_unique_id_bytes(void);
int st_comm_unique_id(void* id_out, size_t id_bytes);
int st_comm_init(const void* id, size_t id_bytes, int rank, int world, void** comm_out);
int st_comm_destroy(void* comm);
/* ranks in the communicator as RCCL itself counts them (ncclCommCount): bench.py prints it beside torch.distributed's world size */
int st_comm_count(void* comm, int* count);
int st_allreduce_f32(void* comm, float* buf, size_t n, void* stream);
int st_allreduce_buckets_f32(void* comm, float* base, const size_t* starts, const size_t* counts, int n_buckets,
                             void* stream);

/* ---- calc_mfccs (preprocessing.py:61-84): librosa.feature.mfcc + delta + delta-delta ----------
 * Same input conventions as st_melspec_f32 (mel_basis is the n_mels = 128 filterbank librosa's
 * mfcc uses by default).  out [total_frames][3 * n_mfcc]: mfcc | delta | delta2, each block
 * z-normalised per utterance.  delta follows the librosa 0.5.x FIR implementation (see oracle). */
size_t st_mfcc_ws(int n_utts, int64_t total_frames, int n_mels, int n_mfcc);
int st_mfcc_f32(const float* audio, const int64_t* sample_offsets, int n_utts, int64_t max_samples,
                const float* mel_basis, int n_mels, int n_mfcc, int n_fft, int hop,
                const int64_t* frame_offsets, int64_t total_frames, float* out, void* workspace,
                size_t workspace_bytes, void* stream);

/* ---- helpers -------------------------------------------------------------------------- */
int st_fill_f32(float* dst, float value, size_t n, void* stream);
/* zero the halo rows of a padded NWC tensor (needed when a buffer is re-described for a new shape) */
int st_zero_halos_f32(const st_tensor3* t, void* stream);
/* ... and the same for every buffer of a shape in ONE launch: `regions_device` is a DEVICE table of n_regions pairs
 * {uint64 address, uint64 bytes} (addresses and sizes multiples of 16; one workgroup zeroes one range, so long ranges are best cut
 * into pieces of ~1 MB).  The engine keeps one table per (batch, frames) shape it has seen: re-entering a shape -- the reference pads
 * every batch to its own longest member (speech_input.py:37-45), so that is nearly every step -- costs one launch, not one per tensor. */
int st_zero_regions(const void* regions_device, int n_regions, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHT_HIP_H */
