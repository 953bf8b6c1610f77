/*
 * gennet_hip.h -- C ABI of libgennet_hip.so, the MI355X (gfx950) kernel library behind gennet_amd.
 *
 * The reference (hagabbar/GenNet, BBH_version/) has no native boundary of its own: every FLOP of its hot
 * path runs inside Keras 2.2.4 / TensorFlow 1.12 ops and numpy.  Each entry point below therefore cites
 * the reference call site whose arithmetic it replaces (file:line in /root/reference/BBH_version/), i.e.
 * the TF kernel that `model.add(<Layer>)` / `train_on_batch` / `predict` would have dispatched.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless the name ends in _host; buffers are caller-owned
 *   - activations are channels-last fp32: (B, L, C) row-major, exactly Keras' layout
 *   - Conv1D kernels (k, Cin, Cout), Dense kernels (in, out), Conv2D kernels (kh, kw, Cin, Cout)
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no hidden syncs, no allocation
 *   - return 0 on success, a negative GN_E* code on bad arguments or launch failure (no exceptions)
 */
#ifndef GENNET_HIP_H
#define GENNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GN_OK 0
#define GN_EINVAL (-1)   /* bad shape / unsupported configuration */
#define GN_ELAUNCH (-2)  /* hipLaunch / runtime error (see gn_last_error) */
#define GN_EWORKSPACE (-3) /* workspace too small */

/* activation kinds (Activation('relu'|'tanh'|'sigmoid'|'linear'), LeakyReLU(alpha), ReLU(max_value)):
 * bbhMahoGANy.py:238,254,...,293 (tanh/linear), :363-400 (relu, ReLU(max_value=1.0)), :440,448 (LeakyReLU 0.2), :495 (sigmoid) */
enum gn_act { GN_ACT_LINEAR = 0, GN_ACT_RELU = 1, GN_ACT_RELU_MAX = 2, GN_ACT_LEAKY = 3, GN_ACT_TANH = 4, GN_ACT_SIGMOID = 5 };

const char* gn_last_error(void);
int gn_version(void);

/* ---- Conv1D (bbhMahoGANy.py:250,259,267,275,283,292 generator; :362-371, :382-394 point-estimator; the
 *      width-2 Conv2D of :439,:447 after gn_conv2d_w2_fold) ----------------------------------------------
 * y[b,t,co] = act(bias[co] + sum_{k,ci} x[b, stride*t + k - pad_left, ci] * w[k,ci,co]), zero outside [0,L).
 * Lout is the caller-computed output length (TF SAME/VALID rule).  bias may be NULL.
 * Dispatches to the MFMA implicit-GEMM kernel (Cin >= 16), the small-Cin streaming kernel (Cin <= 4) or the
 * small-Cout reduction kernel (Cout <= 4). */
int gn_conv1d_fwd(const float* x, const float* w, const float* bias, float* y,
                  int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout,
                  int act, float act_param, void* stream);

/* EXPERIMENTAL, opt-in: the same forward convolution on the bf16 matrix cores with every fp32 operand split into three bf16
 * pieces (six v_mfma_f32_32x32x16_bf16 products, fp32 accumulation; error of the order of one fp32 rounding per product --
 * see gennet_amd/csrc/conv_bf16x3.hip).  Needs Cin % 16 == 0, Cout % 64 == 0, k <= 5, stride 1.  `ws` holds the split
 * operands (gn_conv1d_bf16x3_workspace bytes); resplit = 0 reuses the planes of the previous call (benchmarks only). */
size_t gn_conv1d_bf16x3_workspace(int B, int L, int Cin, int Cout, int k);
int gn_conv1d_fwd_bf16x3(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                         int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout,
                         int act, float act_param, int resplit, void* stream);
/* Transform-domain fp32 convolution (Cook-Toom / Winograd F(2,5), gennet_amd/csrc/conv_wino.hip) for the unit-stride 5-tap layers --
 * generator Conv1D(128..1024, 5, padding='same') bbhMahoGANy.py:259-283, PE q branch Conv1D(128, 5) / Conv1D(256, 5) :382-384: six fp32
 * multiplies per two outputs instead of ten, every product still an exact fp32 fma on v_mfma_f32_32x32x2_f32.  Direct entry (tests, layer
 * benchmarks); `ws` receives the transformed kernel of this launch (gn_conv1d_wino_workspace bytes).  Needs k == 5, stride 1, Cin % 8 == 0,
 * Cout % 64 == 0. */
size_t gn_conv1d_wino_workspace(int Cin, int Cout);
int gn_conv1d_fwd_wino(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                       int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout,
                       int act, float act_param, void* stream);
/* Process-wide opt-in: mode 1 routes every Conv1D forward / data-gradient launch that the bf16x3 kernel supports and that is
 * large enough to gain from it (Cin >= 256 and Cout >= 256, unit input stride) through it, operands split per launch into the
 * caller-owned device `workspace` (launches whose planes do not fit stay on the fp32 kernel); mode 0 (the default) restores
 * the exact-fp32 MFMA path everywhere.  The workspace must stay alive while mode 1 is set. */
int gn_set_conv_math(int mode, void* workspace, size_t workspace_bytes);

/* Same with a following Dropout fused into the epilogue (discriminator: Conv2D -> LeakyReLU -> Dropout(0.4), bbhMahoGANy.py:439-443,
 * :447-452): y = mask ? act(.)/(1-rate) : 0, mask = uint8 keep-mask of y's shape (gn_dropout_mask).  Cout > 4. */
int gn_conv1d_fwd_dropout(const float* x, const float* w, const float* bias, const uint8_t* mask, float* y,
                          int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout,
                          int act, float act_param, float rate, void* stream);
/* Linear Conv1D forward that also returns the BatchNorm statistics of its output, sums[0:Cout] = sum y, sums[Cout:2*Cout] = sum y^2
 * (fp64) -- the Conv1D -> BatchNormalization pairs of the generator (bbhMahoGANy.py:250-251, ..., :283-284).  On the pipelined MFMA
 * kernel the sums are accumulated in the epilogue (per-block fp64 partials, fixed-order final reduction): the separate statistics
 * pass over the output tensor disappears; other shapes run the convolution and then gn_bn_stats' pass, with the same result. */
size_t gn_conv1d_fwd_stats_workspace(int B, int Lout, int Cout);
int gn_conv1d_fwd_stats(const float* x, const float* w, const float* bias, float* y, double* sums, void* ws, size_t ws_bytes,
                        int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout, void* stream);

/* wt[k, co, ci] = w[k, ci, co]  (operand layout the dgrad GEMM consumes) */
int gn_conv1d_transpose_w(const float* w, float* wt, int k, int Cin, int Cout, void* stream);

/* dx[b,tau,ci] = sum_{k,co} dy[b,t,co] * w[k,ci,co] over stride*t + k - pad_left == tau.  wt from
 * gn_conv1d_transpose_w.  (Backward of the call sites above; TF Conv2DBackpropInput.) */
int gn_conv1d_dgrad(const float* dy, const float* wt, float* dx,
                    int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout, void* stream);

/* Same, with the backward of the PRODUCER layer's [activation -> dropout] fused into the epilogue, so the gradient leaves the kernel
 * already multiplied by act'(.): dx = mask_prev ? dx/(1-rate) * act'(y_prev*(1-rate)) : 0 (mask_prev NULL: dx *= act'(y_prev)).
 * y_prev (B, L, Cin) is the producer's output (= this layer's input).  MFMA path only (Cin > 4 and Cout > 4). */
int gn_conv1d_dgrad_fused(const float* dy, const float* wt, float* dx,
                          int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout,
                          const float* y_prev, const uint8_t* mask_prev, int act_prev, float act_param_prev, float rate_prev, void* stream);

/* dw[k,ci,co] = sum_{b,t} x[b, stride*t + k - pad_left, ci] * dy[b,t,co];  db[co] = sum_{b,t} dy[b,t,co] (db may be NULL).
 * ws: workspace of at least gn_conv1d_wgrad_workspace(...) bytes (split-K partial slabs, summed in a fixed order,
 * so the result is bitwise reproducible).  (TF Conv2DBackpropFilter + BiasAddGrad.) */
size_t gn_conv1d_wgrad_workspace(int B, int L, int Cin, int Cout, int k, int stride, int Lout);
int gn_conv1d_wgrad(const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes,
                    int B, int L, int Cin, int Cout, int k, int stride, int pad_left, int Lout, void* stream);

/* ---- width-2 Conv2D fold (bbhMahoGANy.py:439,:447: Conv2D(C,(5,5),strides=(2,1),padding='same') on (n,2,Cin)) ----
 * wf[kh, w*Cin+c, w2*Cout+c2] = w[kh, w-w2+2, c, c2];   biasf = [bias, bias]
 * unfold_grad: dw[kh,kw,c,c2] = sum over the (w,w2) blocks with w-w2+2 == kw (0 for kw in {0,4}); db = dbf[:Cout]+dbf[Cout:] */
int gn_conv2d_w2_fold(const float* w, const float* bias, float* wf, float* biasf, int kh, int Cin, int Cout, void* stream);
int gn_conv2d_w2_unfold_grad(const float* dwf, const float* dbf, float* dw, float* db, int kh, int Cin, int Cout, void* stream);
/* Conv1D with 6 <= k <= 40 taps (`filtsize = 5 # 10 is best`, bbhMahoGANy.py:228, the Conv1D(.., filtsize, ..) of :250-292; the 16-tap layers of the
 * reference's saved Keras models) as an h-tap convolution over G*Cin channels with pad_left 0 (csrc/tap_fold.hip), G = ceil(k/5) groups of h = ceil(k/G)
 * taps (gn_conv1d_tap_groups): every further tap group becomes a further group of input channels over the input shifted by g*h rows, so the <= 5-tap
 * kernels above run it and accumulate the groups in their own K loop.
 *   gn_conv1d_tapfold_x:    x (B, L, Cin) -> x2 (B, L + pad_left, G*Cin), x2[b, j, g*Cin:(g+1)*Cin] = x[b, j - pad_left + g*h] (0 outside the input)
 *   gn_conv1d_tapfold_w:    w (k, Cin, Cout) -> w2 (h, G*Cin, Cout), w2[t, g*Cin:(g+1)*Cin] = w[t + g*h] (zero taps where t + g*h >= k)
 *   gn_conv1d_tapunfold_dw: the inverse, for the weight gradient: dw2 (h, G*Cin, Cout) -> dw (k, Cin, Cout)
 *   gn_conv1d_tapunfold_dx: dx2 (B, L + pad_left, G*Cin) -> dx (B, L, Cin), dx[b, l] = sum_g dx2[b, l + pad_left - g*h, g*Cin:(g+1)*Cin]
 * Then y = gn_conv1d_fwd*(x2, w2, bias, k' = h, stride, pad_left' = 0, L' = L + pad_left, Cin' = G*Cin) with the layer's own Lout. */
int gn_conv1d_tap_groups(int k, int* groups, int* taps);
int gn_conv1d_tapfold_x(const float* x, float* x2, int B, int L, int Cin, int k, int pad_left, void* stream);
int gn_conv1d_tapunfold_dx(const float* dx2, float* dx, int B, int L, int Cin, int k, int pad_left, void* stream);
int gn_conv1d_tapfold_w(const float* w, float* w2, int k, int Cin, int Cout, void* stream);
int gn_conv1d_tapunfold_dw(const float* dw2, float* dw, int k, int Cin, int Cout, void* stream);

/* ---- UpSampling1D(2) -> Conv1D(C, 5, strides=s, padding='same') fold (bbhMahoGANy.py:249-250 s=2, :258-259 s=1) ----
 * The pair equals a 3-tap stride-1 'same' conv (pad_left 1) on the UN-upsampled input x (B, L, Cin), so the upsampled tensor is
 * never written and 2 of 5 taps' multiplies go away:
 *   s=2:  y[t]    = W0 x[t-1] + (W1+W2) x[t] + (W3+W4) x[t+1]                     wf (3, Cin, Cout),   biasf = bias
 *   s=1:  y[2u]   = (W0+W1) x[u-1] + (W2+W3) x[u] + W4 x[u+1]     columns [0,Cout)
 *         y[2u+1] = W0 x[u-1] + (W1+W2) x[u] + (W3+W4) x[u+1]     columns [Cout,2Cout)  wf (3, Cin, 2*Cout), biasf = [bias, bias];
 *         the folded conv's (B, L, 2*Cout) output is the layer's (B, 2L, Cout) output in memory.
 * unfold_grad maps the folded conv's weight / bias gradient (gn_conv1d_wgrad on x and the same dy memory) back onto the 5 taps:
 * dW[k] = sum of the folded taps W[k] went into; db = dbf (s=2) or dbf[:Cout] + dbf[Cout:] (s=1). */
int gn_conv1d_up2_fold(const float* w, const float* bias, float* wf, float* biasf, int Cin, int Cout, int stride, void* stream);
int gn_conv1d_up2_unfold_grad(const float* dwf, const float* dbf, float* dw, float* db, int Cin, int Cout, int stride, void* stream);

/* ---- Dense (bbhMahoGANy.py:234 generator 100 -> 256*n_pix/2; :377,:399,:494 flatten -> 1 heads) -------------
 * y[b,o] = act(bias[o] + sum_i x[b,i] * w[i,o]).  Large `out` goes through the MFMA GEMM, out <= 4 through
 * the streaming dot-product kernel. */
int gn_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int in, int out,
                 int act, float act_param, void* stream);
/* dw[i,o] = sum_b x[b,i]*dy[b,o]; db[o] = sum_b dy[b,o]; dx[b,i] = sum_o dy[b,o]*w[i,o] (dx may be NULL). */
size_t gn_dense_bwd_workspace(int B, int in, int out);
int gn_dense_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                 void* ws, size_t ws_bytes, int B, int in, int out, void* stream);

/* flatten -> Dense(out <= 4) backward with the producer's [activation -> dropout] backward fused (x IS the producer's output) */
int gn_dense_bwd_fused(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int in, int out,
                       const uint8_t* mask_prev, int act_prev, float act_param_prev, float rate_prev, void* stream);

/* ---- elementwise ----------------------------------------------------------------------------------------- */
/* y = act(x) (Activation / LeakyReLU / ReLU layers when not fused into the producing kernel) */
int gn_act_fwd(const float* x, float* y, size_t n, int act, float act_param, void* stream);
/* dx = dy * act'(.) expressed through the activation OUTPUT y; in-place (dx == dy) allowed */
int gn_act_bwd(const float* dy, const float* y, float* dx, size_t n, int act, float act_param, void* stream);
/* backward of [activation -> dropout] in one pass through the post-dropout output y: dx = mask ? dy/(1-rate) * act'(y*(1-rate)) : 0 */
int gn_act_dropout_bwd(const float* dy, const float* y, const uint8_t* mask, float* dx, size_t n, int act, float act_param, float rate, void* stream);
/* gn_bn_apply with the Dropout keep-mask GENERATED in the same pass (the draw of gn_dropout_mask, bit for bit) and written to
 * mask_out for the backward pass: y = mask ? act(x * scale + shift) / (1 - rate) : 0.  C % 4 == 0. */
int gn_bn_apply_dropgen(const float* x, const float* scale, const float* shift, uint8_t* mask_out, float* y, size_t rows, int C,
                        int act, float act_param, float rate, uint64_t seed, uint64_t offset, void* stream);
/* Inference-phase BatchNormalization folded into the preceding convolution (generator.predict, bbhMahoGANy.py:1248, :1330):
 * w_out[r, n] = w[r, n] * scale[n] (r over taps * Cin rows), bias_out[n] = bias[n] * scale[n] + shift[n], with scale / shift
 * from gn_bn_infer_coeffs; conv(x; w_out, bias_out) then equals BN_infer(conv(x; w, bias)).  Cout % 4 == 0; bias may be NULL. */
int gn_conv_fold_bn(const float* w, const float* bias, const float* scale, const float* shift, float* w_out, float* bias_out,
                    size_t rows, int Cout, void* stream);
/* PReLU (bbhMahoGANy.py:39; the act = 'prelu' branches at :237-286, :315-325): y[b,f] = x > 0 ? x : alpha[f] * x with one
 * alpha per feature of a sample (F features, F % 4 == 0).  Backward: dx = dy * (x > 0 ? 1 : x < 0 ? alpha : 0) (Keras'
 * relu(x) - alpha * relu(-x) has zero gradient at x == 0), dalpha[f] = sum_b dy[b,f] * min(x[b,f], 0); dx or dalpha may be NULL. */
int gn_prelu_fwd(const float* x, const float* alpha, float* y, int B, size_t F, void* stream);
int gn_prelu_bwd(const float* dy, const float* x, const float* alpha, float* dx, float* dalpha, int B, size_t F, void* stream);
/* Dropout (bbhMahoGANy.py:239,255,...,288 rate 0.2; :443,:452 rate 0.4): keep-mask generation (Philox4x32-10,
 * element i draws counter (offset + i/4), lane i%4; keep iff u >= rate) and application y = x*mask/(1-rate). */
int gn_dropout_mask(uint8_t* mask, size_t n, float rate, uint64_t seed, uint64_t offset, void* stream);
int gn_dropout_apply(const float* x, const uint8_t* mask, float* y, size_t n, float rate, void* stream);
/* UpSampling1D(size=2) (bbhMahoGANy.py:249,:258) and its adjoint */
int gn_upsample2_fwd(const float* x, float* y, int B, int L, int C, void* stream);
int gn_upsample2_bwd(const float* dy, float* dx, int B, int L, int C, void* stream);
/* MaxPooling2D(pool_size=(2,1)) (the discriminator's `maxpool = True` configuration, bbhMahoGANy.py:426, :444-490): x (B, H, R) -> y (B, H/2, R), the
 * maximum over row pairs along H, R = W * C floats per row; an odd last row is dropped ('valid').  Backward: dy to the row that held the maximum (a tie:
 * the first row of the pair, as TensorFlow's max-pool gradient), zero elsewhere. */
int gn_maxpool_h2_fwd(const float* x, float* y, int B, int H, int R, void* stream);
int gn_maxpool_h2_bwd(const float* dy, const float* x, float* dx, int B, int H, int R, void* stream);
/* MyLayer (bbhMahoGANy.py:180-184): img[b,t,0] = x[b,t]; img[b,t,1] = event[t] - x[b,t];  adjoint dx = d0 - d1 */
int gn_subtract_stack_fwd(const float* x, const float* event, float* img, int B, int n, void* stream);
int gn_subtract_stack_bwd(const float* dimg, float* dx, int B, int n, void* stream);
/* A user-defined keras Layer whose call() is K.stack([a0*x + b0, a1*x + b1], axis=2) on x (B, n, 1) -- the form of the script's own
 * MyLayer.call (bbhMahoGANy.py:180-184: diff = self.const - x; K.stack([x, diff], axis=2) is a0 = 1, b0 = NULL, a1 = -1, b1 = const):
 *   img[b, t, 0] = a0*x[b,t] + b0[t],  img[b, t, 1] = a1*x[b,t] + b1[t]   (b0 / b1 may be NULL = 0);   dx = a0*dimg[...,0] + a1*dimg[...,1]. */
int gn_affine_stack_fwd(const float* x, const float* b0, const float* b1, float a0, float a1, float* img, int B, int n, void* stream);
int gn_affine_stack_bwd(const float* dimg, float a0, float a1, float* dx, int B, int n, void* stream);
/* discriminator batch assembly (bbhMahoGANy.py:1268-1289): sX (2B, n, 2, 1): rows [0,B) = [real[b,t], noise[b,t]], rows [B,2B) =
 * [fake[j,t], event[t]-fake[j,t]] with j = 2B-1-row (the reference's np.append prepends, so the fake half is in reversed order) */
int gn_assemble_d_batch(const float* real, const float* noise, const float* fake, const float* event, float* sX, int B, int n, void* stream);
/* uniform(lo,hi) and normal(mean,std) fills from Philox (host RNG replacement for bbhMahoGANy.py:1161,1247,1277,1295) */
int gn_fill_uniform(float* out, size_t n, float lo, float hi, uint64_t seed, uint64_t offset, void* stream);
int gn_fill_normal(float* out, size_t n, float mean, float std, uint64_t seed, uint64_t offset, void* stream);
/* out[i,:] = src[idx[i],:] (random.sample batch gather, bbhMahoGANy.py:1156-1157, :1244) */
int gn_gather_rows(const float* src, const int64_t* idx, float* out, int rows, int width, void* stream);
/* y[i] += a * x[i] */
int gn_axpy(float* y, const float* x, float a, size_t n, void* stream);

/* ---- BatchNormalization(momentum=0.99) (bbhMahoGANy.py:235 over 256*n_pix/2 features; :251,...,:284 over channels) ----
 * x is viewed as (rows, C).  Statistics are accumulated in fp64: sums[0:C] = sum x, sums[C:2C] = sum x^2.
 * Between gn_bn_stats and gn_bn_finalize a data-parallel caller all-reduces `sums` (SyncBN). */
size_t gn_bn_stats_workspace(size_t rows, int C);
int gn_bn_stats(const float* x, size_t rows, int C, double* sums, void* ws, size_t ws_bytes, void* stream);
/* mean = S1/n, var = S2/n - mean^2 (biased); scale = gamma/sqrt(var+eps), shift = beta - mean*scale;
 * moving statistics as a plain exponential average, TF's assign_moving_average(zero_debias=False):
 *   moving_mean -= (moving_mean - mean)*(1-m); moving_var -= (moving_var - var*n/(n-(1+eps)))*(1-m)   (keras normalization.py variance factor).
 * save_mean/save_invstd are kept for the backward pass. */
int gn_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum,
                   float* moving_mean, float* moving_var, float* scale, float* shift,
                   float* save_mean, float* save_invstd, int C, void* stream);
/* the same with the moving statistics updated as keras 2.2.4's TF backend does (K.moving_average_update ->
 * tf moving_averages.assign_moving_average(x, value, momentum, zero_debias=True); bbhMahoGANy.py:223 momentum, :1248 the
 * generator.predict that consumes them): biased_* are shadow accumulators that start at ZERO,
 *   biased -= (biased - value)*(1-m);  moving -= moving - biased/(1 - m^local_step)
 * with local_step the ALREADY incremented update count (1 for the first update): the moving statistic is the debiased average
 * of the batch values and forgets its 0/1 initial value at the first update. */
int gn_bn_finalize_zero_debias(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum,
                               float* moving_mean, float* moving_var, float* biased_mean, float* biased_var, int local_step,
                               float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* stream);
/* inference phase: scale/shift from the moving statistics */
int gn_bn_infer_coeffs(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                       float eps, float* scale, float* shift, int C, void* stream);
/* y = dropout(act(x*scale[c] + shift[c])) in one pass; mask may be NULL (no dropout / inference) */
int gn_bn_apply(const float* x, const float* scale, const float* shift, const uint8_t* mask, float* y,
                size_t rows, int C, int act, float act_param, float rate, void* stream);
/* backward pass 1: g = dy * mask/(1-rate) * act'(a), xhat = (x-mean)*invstd, with a = the activation output: taken from the stored
 * layer output (a = y/(mask scale)) or, when scale and shift (the forward pass' gn_bn_finalize outputs) are given, RECOMPUTED from
 * the pre-BN tensor as act(fma(x, scale, shift)) -- bit-identical to the forward and one 4-byte read per element less (y may then
 * be NULL).  dsums[0:C] = sum g, dsums[C:2C] = sum g*xhat (fp64).  A data-parallel caller all-reduces dsums. */
int gn_bn_bwd_stats(const float* dy, const float* y, const float* x, const uint8_t* mask,
                    const float* save_mean, const float* save_invstd, double* dsums, void* ws, size_t ws_bytes,
                    size_t rows, int C, int act, float act_param, float rate, const float* scale, const float* shift, void* stream);
/* backward pass 2: dx = gamma*invstd*(g - dsum/n - xhat*dsum_xhat/n); dgamma = dsums_local[C:2C], dbeta = dsums_local[0:C];
 * scale / shift as above. */
int gn_bn_bwd_apply(const float* dy, const float* y, const float* x, const uint8_t* mask,
                    const float* gamma, const float* save_mean, const float* save_invstd,
                    const double* dsums_global, double count, const double* dsums_local, float* dx, float* dgamma, float* dbeta,
                    size_t rows, int C, int act, float act_param, float rate, const float* scale, const float* shift, void* stream);

/* BatchNormalization backward when the layer's output feeds ONLY a Conv1D(1 filter, k <= 5 taps, stride 1) -- the generator's last
 * BatchNormalization -> tanh -> Dropout -> Conv1D(1, 5, padding='same') (bbhMahoGANy.py:284-292).  The gradient arriving at the BN
 * output is then that conv's data gradient  dz[b,t,c] = sum_j g[b, t - j + pad_left] * w[j,c]  (g (B, Lout): the conv's output
 * gradient, w (k, C): its kernel; terms with t - j + pad_left outside [0, Lout) are zero), rank-k in (t, c).  These two calls take
 * (g, w, L, Lout, k, pad_left) in place of dy and form dz where it is consumed, so gn_conv1d_dgrad never writes the (B, L, C) tensor
 * and the two passes never read it (12 bytes per element less: 4.3 GB x 3 per generator step at the BASELINE size).  rows = B * L.
 * Same results as gn_bn_bwd_stats / gn_bn_bwd_apply on the materialised dz up to the rounding of the k-term sum.
 * Needs C % 4 == 0 and scale / shift (the activation output is recomputed from x). */
int gn_bn_bwd_stats_conv1(const float* g, const float* w, int L, int Lout, int k, int pad_left, const float* x, const uint8_t* mask,
                          const float* save_mean, const float* save_invstd, double* dsums, void* ws, size_t ws_bytes, size_t rows, int C,
                          int act, float act_param, float rate, const float* scale, const float* shift, void* stream);
int gn_bn_bwd_apply_conv1(const float* g, const float* w, int L, int Lout, int k, int pad_left, const float* x, const uint8_t* mask,
                          const float* gamma, const float* save_mean, const float* save_invstd, const double* dsums_global, double count,
                          const double* dsums_local, float* dx, float* dgamma, float* dbeta, size_t rows, int C, int act, float act_param,
                          float rate, const float* scale, const float* shift, void* stream);

/* ---- losses + metric (compile(loss='binary_crossentropy'|'mean_squared_error', metrics=['accuracy']),
 *      bbhMahoGANy.py:1101-1119) ---------------------------------------------------------------------------
 * p, y: (B, 1).  out[0] = loss (mean over the local B rows scaled by B/Bglobal), out[1] = #rows with round(p)==y;
 * dp = dLoss/dp with the mean taken over Bglobal rows (data-parallel ranks pass the global batch size). */
int gn_bce_loss(const float* p, const float* y, float* dp, float* out, int B, int Bglobal, void* stream);
int gn_mse_loss(const float* p, const float* y, float* dp, float* out, int B, int Bglobal, void* stream);

/* ---- Adam, keras form (bbhMahoGANy.py:1101,1107,1115,1119: Adam(lr=9e-5, beta_1=0.5)) ---------------------
 * m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (host). */
int gn_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps, void* stream);

/* ---- hipGraph capture of a whole train step (bbhMahoGANy.py:1153-1168, :1241-1299 at the script's own batch size 8, where the loop is
 * launch-bound): every launch of the library is stream-ordered and allocation-free, so train_on_batch can be captured on the launch stream and
 * replayed.  Scalars that change from step to step would be frozen into the graph as by-value kernel arguments; these entry points read them
 * from device memory instead (the host refreshes one small block before each replay).
 * gn_set_rng_base: from now on (this host thread) every Philox kernel of the library -- gn_dropout_mask, gn_fill_uniform, gn_fill_normal(_dyn),
 * gn_bn_apply_dropgen -- adds *base_dev to its counter offset at run time; NULL switches it off.  With base = (stream position at replay) -
 * (stream position at capture) a replay draws exactly what the un-captured step would have drawn. */
int gn_set_rng_base(const uint64_t* base_dev);
/* gn_adam_step with lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) read from device memory */
int gn_adam_step_dyn(float* p, const float* g, float* m, float* v, size_t n, const float* lr_t_dev, float b1, float b2, float eps, void* stream);
/* gn_fill_normal with the standard deviation read from device memory (the CNN loop's per-batch sigma ~ U(0, 5), bbhMahoGANy.py:1161) */
int gn_fill_normal_dyn(float* out, size_t n, float mean, const float* sd_dev, uint64_t seed, uint64_t offset, void* stream);
/* gn_bn_finalize_zero_debias with the (already incremented) local_step read from device memory */
int gn_bn_finalize_zero_debias_dyn(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum,
                                   float* moving_mean, float* moving_var, float* biased_mean, float* biased_var, const int32_t* local_step_dev,
                                   float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* stream);

/* ---- profiling hooks used by bench.py: accumulate HIP-event time of every MFMA conv / wgrad launch ---------- */
int gn_prof_enable(int on);
int gn_prof_reset(void);
/* sums over launches since reset of one kernel family (kind 0 = conv_mfma_kernel: forward + data gradient,
 * 1 = wgrad_mfma_kernel, -1 = both): out[0] = launches, out[1] = total ms, out[2] = total algorithmic FLOP,
 * out[3] = total algorithmic bytes (each operand read once, the result written once) */
int gn_prof_collect(int kind, double* out4_host);

/* ---- template synthesiser (gw_template_maker.py) --------------------------------------------------------- */
/* Closed-form frequency-domain IMR chirp (this library's own model, NOT LAL IMRPhenomPv2; replaces the call at
 * gw_template_maker.py:507-516) fused with whiten_data(...,'fd') (:243-286, :518-519):
 *   out_hp/out_hc[(b, f)] complex128 interleaved (re,im), Nf = N/2+1 bins, bin f <-> f*df; zero below f_low and at DC.
 *   scale[f] = sqrt(2*invpsd[f]/fs) precomputed by the caller from the PSD (0 where psd <= 0). */
int gn_chirp_fd_whitened(const double* m1, const double* m2, const double* scale, double* out_hp, double* out_hc,
                         int nb, int Nf, double df, double f_low, double dist_mpc, double iota, double phi0, void* stream);
/* Batched real FFTs with numpy semantics (np.fft.irfft at gw_template_maker.py:191,:283,:521-522,:775-777; np.fft.rfft
 * at :268).  irfft: X (nb, N/2+1) complex128 -> out (nb, N) float64, 1/N normalisation, imaginary parts of DC/Nyquist
 * ignored.  rfft: x (nb, N) -> X (nb, N/2+1).  N a power of two, 16 <= N <= 16384.
 * twiddle: N/2 complex128 values exp(+2*pi*i*k/N), k = 0..N/2-1, computed once by the caller (fp64 on the host). */
int gn_irfft_f64(const double* X, double* out, const double* twiddle, int nb, int N, void* stream);
int gn_rfft_f64(const double* x, double* X, const double* twiddle, int nb, int N, void* stream);
/* gen_bbh alignment (gw_template_maker.py:521-565) for a batch: given hp, hc = irfft(...) (nb, N):
 *   ref = argmax_n (hp^2+hc^2)[(n - fs) mod N-rolled]  (first maximum, as numpy.argmax after np.roll(.,-fs));
 *   out[b, n] = g * (Fp*hp + Fc*hc)_rolled[ref - idx[b] - peak_off + crop0 + n], n in [0, crop_len), 0 past the end.
 *   ref_out[b] receives ref. */
int gn_align_crop(const double* hp, const double* hc, const int32_t* idx, double* out, int32_t* ref_out,
                  int nb, int N, int roll, int crop0, int crop_len, int peak_off, double Fp, double Fc, double g, void* stream);
/* gen_bbh + the crop of sim_data for a batch, FUSED (gw_template_maker.py:507-565, :695): one workgroup per template evaluates the
 * whitened chirp spectrum once (the same closed form as gn_chirp_fd_whitened), obtains both polarisations from M = N/2-point
 * complex inverse FFTs resident in LDS, finds ref = argmax(hp^2 + hc^2) of the rolled series (first maximum) and writes
 *   out[b, n] = g * (Fp*hp + Fc*hc)_rolled[ref - idx[b] - peak_off + crop0 + n],  n in [0, crop_len), 0 past the end
 * to out_f64 and/or out_f32 (either may be NULL; not both) and ref to ref_idx (may be NULL).  No intermediate spectrum or time
 * series ever reaches HBM.  scale: (N/2+1,) whitening scale sqrt(2/(psd*fs)) (0 where psd <= 0; gw_template_maker.py:273-281);
 * twiddle as for gn_irfft_f64.  N in {1024, 2048, 4096, 8192, 16384}. */
int gn_synth_templates(const double* m1, const double* m2, const int32_t* idx, const double* scale, const double* twiddle,
                       double* out_f64, float* out_f32, int32_t* ref_idx, int nb, int N, int roll, int crop0, int crop_len,
                       int peak_off, double df, double f_low, double dist_mpc, double iota, double phi0, double Fp, double Fc, double g,
                       void* stream);
/* The same kernel with the PRIOR drawn inside it (BASELINE configs[4]: synthesis fused into the training loop): every template draws
 * (m1, m2) from gen_masses('hunt_constrain') (gw_template_maker.py:327-339: log-uniform component masses in [m_min, M_max - m_min]
 * subject to m1 + m2 < M_max, m1 >= m2, m2/m1 >= 0.5, 20 <= mc <= 35, by rejection) and idx uniformly from [idx_lo, idx_hi)
 * (gen_par :422-426) from a counter-based Philox stream: template b uses counters counter + 1024*b .. counter + 1024*b + 1023 of
 * stream `seed`, the lowest accepted trial wins, so a batch is a pure function of (seed, counter) and ranks / steps take disjoint
 * counter ranges (advance by 1024 per template).  Not the reference's MT19937 stream (statistical parity only).
 * labels (nb, 2) = [mc, m2/m1] fp32, m_out (nb, 2) = [m1, m2], idx_out (nb,): each may be NULL. */
int gn_synth_templates_prior(const double* scale, const double* twiddle, double* out_f64, float* out_f32, float* labels, double* m_out,
                             int32_t* idx_out, int32_t* ref_idx, int nb, int N, int roll, int crop0, int crop_len, int peak_off,
                             double df, double f_low, double dist_mpc, double iota, double phi0, double Fp, double Fc, double g,
                             uint64_t seed, uint64_t counter, int idx_lo, int idx_hi, double m_min, double M_max, void* stream);
/* PSD-coloured Gaussian noise generated AND whitened in one launch, one workgroup per row, nothing but the crop written (BASELINE
 * configs[4]): replaces gen_noise (gw_template_maker.py:161-193: amp = sqrt(0.25 T psd), 0 where psd == 0; re, im = amp * N(0,1), Nf draws
 * each, re block first; DC = 0; x = N * irfft(re + i im) * df) followed by whiten_data(x, flag='td') (:243-286: rfft(tukey(N, 1/8) * x) *
 * sqrt(2 invpsd / fs), DC = 0, irfft) and the crop [crop0, crop0 + crop_len) of sim_data (:695), all fp64 in LDS.
 * amp, wscale: (N/2+1) tables; window: (N) tukey(N, 1/8); twiddle as for gn_irfft_f64.  normals_in (nb, 2 (N/2+1)) = numpy's draws per row
 * [re block | im block], or NULL: Philox (bin k of row b = counter + b (N/2+1) + k of stream `seed`, Box-Muller -> (re, im)); normals_out
 * (same layout) or NULL receives the normals used.  add_f64 (nb, crop_len) or NULL: rows the noise is added to (template crops).
 * out_f64 / out_f32 (nb, crop_len): at least one.  N in {1024, 2048, 4096, 8192, 16384}. */
int gn_noise_whitened(const double* amp, const double* wscale, const double* window, const double* twiddle, const double* normals_in,
                      double* normals_out, const double* add_f64, double* out_f64, float* out_f32, int nb, int N, int crop0, int crop_len,
                      double df, uint64_t seed, uint64_t counter, void* stream);
/* gn_synth_templates / gn_synth_templates_prior (m1 == m2 == idx == NULL: prior mode) with that noise chain run by the SAME workgroup after
 * the template's crop is formed: out = template * g + whitened coloured noise, one launch per batch (BASELINE configs[4]: "template + noise
 * synth fused into the train loop").  scale doubles as the whitening scale of the noise; noise_amp, window as for gn_noise_whitened; the
 * noise stream is (noise_seed, noise_counter), row b bin k = noise_counter + b (N/2+1) + k.  crop_len <= N/4. */
int gn_synth_templates_noise(const double* m1, const double* m2, const int32_t* idx, const double* scale, const double* twiddle,
                             const double* noise_amp, const double* window, double* out_f64, float* out_f32, float* labels, double* m_out,
                             int32_t* idx_out, int32_t* ref_idx, int nb, int N, int roll, int crop0, int crop_len, int peak_off, double df,
                             double f_low, double dist_mpc, double iota, double phi0, double Fp, double Fc, double g, uint64_t seed,
                             uint64_t counter, int idx_lo, int idx_hi, double m_min, double M_max, uint64_t noise_seed,
                             uint64_t noise_counter, double* normals_out, void* stream);
/* gen_noise (gw_template_maker.py:161-193) spectrum: X[b,f] = amp[f]*(xi_re + i xi_im), DC = 0 (Philox normals, re block then im block) */
int gn_noise_fd(const double* amp, double* X, int nb, int Nf, uint64_t seed, uint64_t offset, void* stream);
/* x *= s (fp64), used for N*df and gw_norm_constant scalings; and fp64 -> fp32 narrowing with scale */
int gn_scale_f64(double* x, double s, size_t n, void* stream);
/* x[i] *= w[e % period], e = i (real x) or i/2 (complex_x: interleaved re/im): whiten_data's xf *= sqrt(2*invpsd/fs)
 * (:276) and the Tukey window of the 'td' path (:267-268) over a batch */
int gn_mul_f64(double* x, const double* w, size_t n, size_t period, int complex_x, void* stream);
int gn_f64_to_f32(const double* x, float* y, double s, size_t n, void* stream);

/* ---- posterior read-out score (bbhMahoGANy.py:811-873 overlap_tests; kernel.pdf(positions) at :861,:866) -------------
 * 2-D Gaussian KDE: out[p] = norm * sum_i exp(-0.5 * d^T Sinv d), d = pts[:,p] - data[:,i]; data (2,n), pts (2,m) row-major fp64;
 * Sinv = [[inv00, inv01],[inv01, inv11]] and norm = 1/(n*sqrt(det(2*pi*Sigma))) as scipy.stats.gaussian_kde defines them. */
int gn_kde2d_pdf(const double* data, int n, const double* pts, int m, double inv00, double inv01, double inv11, double norm,
                 double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GENNET_HIP_H */
