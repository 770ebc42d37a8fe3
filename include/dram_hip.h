/*
 * dram_hip.h -- C ABI of libdram_hip.so: the MI355X (gfx950) kernels under the
 * DRAM DC3D forward/backward hot path.
 *
 * The reference (DIAGNijmegen/bodyct-dram) has no FFI: its hot path is the set
 * of ATen ops that dram/parts.py and dram/models.py dispatch.  Each entry point
 * below replaces one of those dispatches; the comment above it cites the
 * reference line that issues the op.  The Python host side
 * (bodyct-dram_amd/dram_amd/) binds these with ctypes; INTEGRATION.md shows the
 * stub.
 *
 * Conventions
 *   - every tensor is fp32, NCDHW, contiguous, in device (HBM) memory;
 *   - S = D*H*W; "row" = one (n, c) pair = S contiguous floats;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on
 *     it and performs no allocation and no host synchronisation;
 *   - `ws` is caller-provided device scratch of at least the size returned by
 *     the matching *_ws_bytes() query (may be NULL when that size is 0);
 *   - return value: 0 on success, a negative DRAM_E* code otherwise (nothing is
 *     launched on an argument error); dram_last_error() gives the message of
 *     the calling thread's last failure.
 */
#ifndef DRAM_HIP_H
#define DRAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRAM_OK 0
#define DRAM_EINVAL (-1)  /* bad shape / null pointer / unsupported argument */
#define DRAM_EWS (-2)     /* workspace too small */
#define DRAM_EHIP (-3)    /* HIP runtime error at launch */

/* norm kinds for dram_norm_* (reference parts.py:17-35 normal_wrapper) */
#define DRAM_NORM_BATCH 0 /* statistics per channel over (N, D, H, W): "bn", "bnt", "bntna", "sbn" */
#define DRAM_NORM_GROUP 1 /* statistics per (sample, group): "ln"/"lnna" (G=1), "in" (G=C) */

const char* dram_last_error(void);
int dram_abi_version(void);

/* ---- 3x3x3 convolution, stride 1, zero padding 1 (nn.Conv3d at parts.py:95,105,133,142,177,185) ----
 *
 * The kernels take the filter in a packed, GEMM-friendly form of dram_conv3d_k3_packed_floats(Cout, Cin)
 * floats that dram_conv3d_k3_pack_weights builds from the reference's [Cout][Cin][3][3][3] parameter:
 *   wt[27][Cin][Cout] (tap-major, Cout fastest) for the direct kernel, followed by
 *   wz[9*4][Cin][Cout], the Winograd F(2,3)-along-z transformed filters ((ky,kx) column x 4 transformed taps).
 *   mode 0 (forward):  wt[t][ci][co] = w[co][ci][t]
 *   mode 1 (backward-data): wt[t][co][ci] = w[co][ci][26-t], i.e. the filter
 *          of the transposed convolution -- run dram_conv3d_k3_fwd on dy with
 *          Cin/Cout swapped to obtain dx.
 * Which kernel runs is the library's choice (Winograd-z for Cin >= 8; exact fp32 arithmetic either way).
 */
size_t dram_conv3d_k3_packed_floats(int Cout, int Cin);
int dram_conv3d_k3_pack_weights(const float* w, float* wt, int Cout, int Cin, int mode, void* stream);

/* y[N,Cout,D,H,W] = conv3d(x[N,Cin,D,H,W], w) (+ bias[Cout] when bias != NULL). */
int dram_conv3d_k3_fwd(const float* x, const float* wt, const float* bias, float* y,
                       int N, int Cin, int Cout, int D, int H, int W, void* stream);

/* Same, with the input given as the channel concatenation of two tensors that
 * is never materialised: channels [0,C1) come from x1[N,C1,D,H,W]; channels
 * [C1,C1+C2) from x2[N,C2,D2,H2,W2] centre-cropped with start offsets
 * (oz,oy,ox) -- crop_concat_5d of parts.py:37-46 fused into the consumer conv
 * (parts.py:153-154). */
int dram_conv3d_k3_fwd_cat(const float* x1, int C1, const float* x2, int C2, int D2, int H2, int W2,
                           int oz, int oy, int ox, const float* wt, const float* bias, float* y,
                           int N, int Cout, int D, int H, int W, void* stream);

/* General form of the two calls above, also used for backward-data: the INPUT is the (virtual)
 * concatenation x1[N,C1,D,H,W] ++ crop(x2[N,C2,D2,H2,W2]) (x2 == NULL: x1 only) and the OUTPUT
 * channels are split the same way over y1[N,Co1,D,H,W] and, when y2 != NULL, the crop window
 * (yoz,yoy,yox) of y2[N,Co2,yD2,yH2,yW2] (elements of y2 outside the window are not written:
 * zero them first when the window is smaller than y2).  wt is [27][C1+C2][Co1+Co2]. */
int dram_conv3d_k3_fwd_ex(const float* x1, int C1, const float* x2, int C2, int D2, int H2, int W2,
                          int oz, int oy, int ox, const float* wt, const float* bias,
                          float* y1, int Co1, float* y2, int Co2, int yD2, int yH2, int yW2,
                          int yoz, int yoy, int yox, int N, int D, int H, int W, void* stream);

/* dw[Cout,Cin,3,3,3] = sum over (n,z,y,x) of dy[n,co,z,y,x] * xpad[n,ci,z+dz,y+dy,x+dx].
 * Deterministic: per-block partial slabs in `ws`, summed in a fixed order. */
size_t dram_conv3d_k3_wgrad_ws_bytes(int N, int Cin, int Cout, int D, int H, int W);
int dram_conv3d_k3_wgrad(const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes,
                         int N, int Cin, int Cout, int D, int H, int W, void* stream);

/* Same with x given as a virtual concatenation (Cin = C1 + C2; size the workspace for that Cin). */
int dram_conv3d_k3_wgrad_ex(const float* x1, int C1, const float* x2, int C2, int D2, int H2, int W2,
                            int oz, int oy, int ox, const float* dy, float* dw, void* ws, size_t ws_bytes,
                            int N, int Cout, int D, int H, int W, void* stream);

/* dbias[Cout] = sum over (n,z,y,x) of dy (conv_bias=True when norm_method is None, models.py:78). */
size_t dram_channel_sum_ws_bytes(int N, int C, int64_t S);
int dram_channel_sum(const float* dy, float* dbias, void* ws, size_t ws_bytes, int N, int C, int64_t S, void* stream);

/* ---- normalisation (+ fused ReLU) (normal_wrapper parts.py:17-35, act_wrapper parts.py:48-54) ----
 *
 * Training-mode forward: computes statistics of x (two-level Chan/Welford
 * combination, fp64 finalise), writes y = act(gamma * (x - mean) * rstd + beta).
 *   kind = DRAM_NORM_BATCH: one statistic per channel; save_mean/save_rstd have
 *          C entries; when running_mean != NULL the running statistics are
 *          updated in place with `momentum` (unbiased variance), as
 *          nn.BatchNorm3d does.
 *   kind = DRAM_NORM_GROUP: one statistic per (n, group); save_* have N*G entries.
 * gamma/beta may be NULL (affine=False).  relu != 0 fuses nn.ReLU.
 * Also fills rowcoef[2*N*C] = {a,b} with y_pre = a*x + b per row, which the
 * backward entry point reuses.
 */
size_t dram_norm_ws_bytes(int N, int C, int64_t S);
int dram_norm_fwd_train(const float* x, const float* gamma, const float* beta, float* y,
                        float* save_mean, float* save_rstd, float* rowcoef,
                        float* running_mean, float* running_var, float momentum, float eps,
                        int kind, int G, int relu, int N, int C, int64_t S,
                        void* ws, size_t ws_bytes, void* stream);

/* Eval-mode BatchNorm: y = act(gamma * (x - running_mean) / sqrt(running_var + eps) + beta);
 * fills save_mean/save_rstd/rowcoef like the training entry so backward works. */
int dram_bn_fwd_eval(const float* x, const float* gamma, const float* beta,
                     const float* running_mean, const float* running_var, float* y,
                     float* save_mean, float* save_rstd, float* rowcoef,
                     float eps, int relu, int N, int C, int64_t S, void* stream);

/* Backward of the above.  dy is the gradient w.r.t. the (activated) output; x is
 * the saved input; the ReLU mask is recomputed from rowcoef.  Outputs dx (may
 * alias dy), dgamma[C], dbeta[C] (either may be NULL).  `batch_stats` = 1 when
 * the forward used the statistics of x itself (training), 0 for eval-mode BN. */
int dram_norm_bwd(const float* dy, const float* x, const float* gamma,
                  const float* save_mean, const float* save_rstd, const float* rowcoef,
                  float* dx, float* dgamma, float* dbeta,
                  int kind, int G, int relu, int batch_stats, int N, int C, int64_t S,
                  void* ws, size_t ws_bytes, void* stream);

/* ---- cross-rank BatchNorm (normal_wrapper "sbn" = nn.SyncBatchNorm, dram/parts.py:32-33, under data parallelism) ----
 * The statistics and the backward sums leave the device between two stages so that the host can exchange them
 * (all-gather / all-reduce over RCCL); every stage is the same row kernels as above.
 *   forward:  dram_bn_stats -> {mean, M2} per channel (fp64) of the local batch; combine over ranks on the host
 *             side (Chan); then dram_bn_fwd_eval with the global mean / biased variance.
 *   backward: dram_bn_bwd_sums -> {sum dy', sum dy'*xhat} per channel (= dbeta, dgamma of the local batch);
 *             all-reduce; dram_bn_bwd_apply_sums with the global sums and element count.
 * Workspace: dram_norm_ws_bytes(N, C, S). */
int dram_bn_stats(const float* x, double* mean_m2, int N, int C, int64_t S, void* ws, size_t ws_bytes, void* stream);
int dram_bn_bwd_sums(const float* dy, const float* x, const float* save_mean, const float* save_rstd,
                     const float* rowcoef, double* sums, int relu, int N, int C, int64_t S, void* ws,
                     size_t ws_bytes, void* stream);
int dram_bn_bwd_apply_sums(const float* dy, const float* x, const float* gamma, const float* save_mean,
                           const float* save_rstd, const float* rowcoef, const double* sums, double count,
                           float* dx, int relu, int N, int C, int64_t S, void* ws, size_t ws_bytes, void* stream);

/* Stand-alone activations (act_wrapper parts.py:48-54 when no norm precedes them). */
int dram_relu_fwd(const float* x, float* y, int64_t n, void* stream);
int dram_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);

/* ---- nn.MaxPool3d(2, 2, 0) (parts.py:191) ----
 * out[N,C,D/2,H/2,W/2]; idx (uint8, 0..7 = dz*4+dy*2+dx) records the first
 * maximum in (z,y,x) scan order -- the element ATen routes the gradient to. */
int dram_maxpool3d_2_fwd(const float* x, float* out, uint8_t* idx, int N, int C, int D, int H, int W, void* stream);
int dram_maxpool3d_2_bwd(const float* dout, const uint8_t* idx, float* dx, int N, int C, int D, int H, int W, void* stream);

/* ---- nn.Upsample(mode='trilinear', align_corners=True) (parts.py:149 scale 2; models.py:146 to input size) ----
 * Arbitrary (D,H,W) -> (Do,Ho,Wo); source index = dst * (in-1)/(out-1) as ATen computes it.
 * The backward is the exact adjoint in gather form (deterministic, no atomics). */
int dram_upsample_trilinear_ac_fwd(const float* x, float* y, int N, int C, int D, int H, int W,
                                   int Do, int Ho, int Wo, void* stream);
int dram_upsample_trilinear_ac_bwd(const float* dy, float* dx, int N, int C, int D, int H, int W,
                                   int Do, int Ho, int Wo, void* stream);
/* The same adjoint in two stages (reduce z with 16-byte loads, then y/x) through a workspace: any size >= 8
 * planes of D*Ho*Wo floats works (planes are processed in groups that fit); _ws_bytes returns the size for all
 * planes at once, 0 when the shape does not qualify (then, or with too small a workspace, this is _bwd). */
size_t dram_upsample_trilinear_ac_bwd_ws_bytes(int N, int C, int D, int H, int W, int Do, int Ho, int Wo);
int dram_upsample_trilinear_ac_bwd_ws(const float* dy, float* dx, void* ws, size_t ws_bytes, int N, int C,
                                      int D, int H, int W, int Do, int Ho, int Wo, void* stream);

/* ---- crop_concat_5d (parts.py:37-46) ----
 * out[N,C1+C2,D,H,W] = cat(t1[N,C1,D,H,W], t2[N,C2,D2,H2,W2][..., oz:oz+D, oy:oy+H, ox:ox+W]). */
int dram_crop_concat_fwd(const float* t1, const float* t2, float* out, int N, int C1, int C2,
                         int D, int H, int W, int D2, int H2, int W2, int oz, int oy, int ox, void* stream);
/* dt1 = dout[:, :C1]; dt2 = 0 outside the crop window, dout[:, C1:] inside.  Either output may be NULL. */
int dram_crop_concat_bwd(const float* dout, float* dt1, float* dt2, int N, int C1, int C2,
                         int D, int H, int W, int D2, int H2, int W2, int oz, int oy, int ox, void* stream);

/* ---- 1x1x1 convolution with bias: the regression head top_layer (models.py:109-110, 145) ---- */
int dram_conv3d_k1_fwd(const float* x, const float* w, const float* bias, float* y,
                       int N, int Cin, int Cout, int64_t S, void* stream);
size_t dram_conv3d_k1_bwd_ws_bytes(int N, int Cin, int Cout, int64_t S);
/* dx[N,Cin,S] (may be NULL), dw[Cout,Cin], dbias[Cout] (may be NULL). */
int dram_conv3d_k1_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias,
                       void* ws, size_t ws_bytes, int N, int Cin, int Cout, int64_t S, void* stream);

/* ---- lobe-masked mean: pooling_dense_features (models.py:37-49, default branch) ----
 * out[n,c] = sum_s x[n,c,s]*m[n,s] / sum_s m[n,s];  msum[n] = sum_s m[n,s] is also returned. */
size_t dram_masked_mean_ws_bytes(int N, int C, int64_t S);
int dram_masked_mean_fwd(const float* x, const float* mask, float* out, float* msum,
                         void* ws, size_t ws_bytes, int N, int C, int64_t S, void* stream);
/* dx[n,c,s] = dout[n,c] * m[n,s] / msum[n]. */
int dram_masked_mean_bwd(const float* dout, const float* mask, const float* msum, float* dx,
                         int N, int C, int64_t S, void* stream);

/* ---- per-lobe whole-scan inference helpers: LesionSegChunkTrain.evaluate_scan,
 *      dram/job_runner.py:729-770; thresholding of LesionSegTest.run, job_runner.py:1003-1005 ----
 * scan: int16 [D,H,W] HU, lobe: uint8 [D,H,W] label map, both resident in HBM.
 * `chunks` is a HOST array of L x {z0,y0,x0,dz,dy,dx,label} (L <= 8). */

/* boxes[(label-1)*6 + {0,1,2}] = min z,y,x; {3,4,5} = max z,y,x (inclusive); min > max if the label is absent
 * (find_crops without the border, dram/utils.py:244-254). */
int dram_label_bboxes(const uint8_t* lobe, int* boxes, int nlabels, int D, int H, int W, void* stream);

/* out[L,1,R,R,R]: crop, set voxels outside the lobe to the window minimum (-2048 HU clips to it), window
 * [wmin,wmax] -> [0,1] (data_transforms.py:37-54), resample to R^3 (trilinear, align_corners=True). */
int dram_lobe_chunks(const int16_t* scan, const uint8_t* lobe, float* out, const int* chunks, int L,
                     int D, int H, int W, int R, float wmin, float wmax, void* stream);

/* htp[v] = resize(sigmoid(dense[l]))(v) for every voxel of chunk l with lobe == label (job_runner.py:765-770). */
int dram_lobe_paste(const float* dense, const uint8_t* lobe, float* htp, const int* chunks, int L,
                    int D, int H, int W, int R, void* stream);

/* 256-bin histogram of uint8(clip(htp,0,1)*255) over voxels with lobe > 0 (input of Otsu, binary_cam
 * dram/utils.py:226-242) and the fp64 sum of htp over the same voxels.  hist: 256 x uint64, sum: 1 x double. */
int dram_lung_hist256(const float* htp, const uint8_t* lobe, unsigned long long* hist, double* sum,
                      int64_t n, void* stream);

/* mask[v] = htp[v] > th. */
int dram_threshold_mask(const float* htp, uint8_t* mask, float th, int64_t n, void* stream);

/* ---- the post-processing tail of LesionSegTest.run (dram/job_runner.py:1003-1012, 1033-1037) ----
 * w_scan = windowing(scan, from_span=(wmin, wmax), to_span=(0, 1)) (utils.py:189-198; the reference calls it with the default
 * span (-1150, 350), job_runner.py:1006), in fp64 exactly as numpy evaluates it.
 * dram_scan_hist256: 256-bin histogram of binary_cam's 8-bit view of w_scan over voxels with lobe > 0 (the input of the
 *   brightness gate's Otsu threshold, `binary_cam(w_scan[lobe > 0], 0.75)`, job_runner.py:1007).  hist: 256 x uint64.
 * dram_lesion_post: pred = htp > th (job_runner.py:1004; pred may be NULL), post = pred & (w_scan > th_scan) & ~(vessel > 0)
 *   (job_runner.py:1008-1010; vessel may be NULL = no vessel mask).
 * dram_mask_overlap: counts[4] (uint64) = {|a & b|, |a | b|, |a|, |b|} of two uint8 masks (non-zero = set): IOU =
 *   (counts[0] + s) / (counts[1] + s), Dice = (2 counts[0] + s) / (counts[2] + counts[3] + s) (utils.py:437-446). */
int dram_scan_hist256(const int16_t* scan, const uint8_t* lobe, unsigned long long* hist, int wmin, int wmax, int64_t n,
                      void* stream);
int dram_lesion_post(const float* htp, const int16_t* scan, const uint8_t* vessel, uint8_t* pred, uint8_t* post, float th,
                     int wmin, int wmax, double th_scan, int64_t n, void* stream);
int dram_mask_overlap(const uint8_t* a, const uint8_t* b, unsigned long long* counts, int64_t n, void* stream);

/* ---- utils.resample(narray, spacing, required_spacing=, new_size=, interpolator=) (dram/utils.py:414-434 -> 299-381), the way
 *      LesionSegTest.run takes its masks (nearest), the scan and the heat map (linear) back to the scan's original grid
 *      (dram/job_runner.py:1016-1032): sitk.ResampleImageFilter, identity transform, same origin / direction, default value 0,
 *      output pixel type = input pixel type.  Restated from ITK's published semantics (SimpleITK absent: parity unpinned) --
 *      output voxel o samples continuous index o * spacing_out / spacing_in; zero beyond size_in - 0.5; nearest = round half up;
 *      linear in double with the upper neighbour clamped, integer pixels by clamp + truncation.
 *      kind: 0 uint8, 1 int16, 2 float32; linear: 0 / 1; in: [Di][Hi][Wi], out: [Do][Ho][Wo]; spacing_*: 3 doubles (z, y, x),
 *      HOST pointers. ---- */
int dram_resample_volume(const void* in, void* out, int kind, int linear, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                         const double* spacing_in, const double* spacing_out, void* stream);

/* ---- nn.PReLU (act_wrapper "prelu", dram/parts.py:51-52): y = x > 0 ? x : a*x, a[nparam], nparam in {1, C};
 *      x: [N,C,S].  bwd: dx (may be NULL), da[nparam] = sum dy*x over x <= 0 (deterministic) ---- */
int dram_prelu_fwd(const float* x, const float* a, float* y, int N, int C, int nparam, int64_t S, void* stream);
size_t dram_prelu_bwd_ws_bytes(int N, int C, int64_t S);
int dram_prelu_bwd(const float* dy, const float* x, const float* a, float* dx, float* da, void* ws,
                   size_t ws_bytes, int N, int C, int nparam, int64_t S, void* stream);

/* ---- F.adaptive_max_pool3d(x, 1) (pooling_dense_features 'global_max', dram/models.py:41-42): per (n,c) row
 *      the maximum and the index of its first occurrence; bwd routes dout to that element ---- */
int dram_global_max_fwd(const float* x, float* out, int64_t* idx, int NC, int64_t S, void* stream);
int dram_global_max_bwd(const float* dout, const int64_t* idx, float* dx, int NC, int64_t S, void* stream);

/* ---- on-device OneShot transforms (dram/data_transforms.py:1140-1239, used by the affine-consistency
 *      losses dram/metrics.py:213-310 on [N,C,D,H,W] device tensors) ----
 * Rescale3DOneShot: F.interpolate(mode='trilinear') with align_corners=False on "#image" tensors (+ its adjoint, the
 * probabilities are resized inside the loss) and mode='nearest' on "#reference" tensors.  scale_* <= 0: the default
 * in/out; > 0: the 1/scale_factor ATen uses when the caller passed scale factors. */
int dram_resize_trilinear_fwd(const float* x, float* y, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                              float scale_z, float scale_y, float scale_x, void* stream);
int dram_resize_trilinear_bwd(const float* dy, float* dx, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                              float scale_z, float scale_y, float scale_x, void* stream);
int dram_resize_nearest(const float* x, float* y, int N, int C, int D, int H, int W, int Do, int Ho, int Wo,
                        float scale_z, float scale_y, float scale_x, void* stream);
/* Flip3DOneShot / Rotate903DOneShot (torch.flip, torch.rot90 over spatial axes): out[o] = in[i] with
 * i[perm[k]] = flip[k] ? n-1-o[k] : o[k]; perm, flip: HOST arrays of 3 ints; (D,H,W) = input extents,
 * output extents = (in[perm[0]], in[perm[1]], in[perm[2]]).  The adjoint is the inverse signed permutation. */
int dram_spatial_permute_flip(const float* x, float* y, int N, int C, int D, int H, int W, const int* perm,
                              const int* flip, void* stream);

/* ---- IntRegRefineLoss, fused and device resident: dram/metrics.py:158-177 (interval hinge on
 *      the lobe-mean probability), 331-358 + 17-51 (pseudo label + BootBinCrossEntropy), 360-373 ----
 * dense, refined, lobes, lesions: [N,1,D,H,W] (S = D*H*W).  refined = the model's second output
 * (DC3DATGeneric); NULL or == dense for DC3D, whose outputs are one tensor.  The regression term and the
 * pseudo label use dense, the segmentation term uses refined (metrics.py:333-357, 362-364).
 * keep[N] = 0 where the CT severity score is 0 (pseudo label forced to background, metrics.py:326-327),
 * 1 otherwise; targets[N][2] = regression band (get_labels, metrics.py:122-138); weight[N] =
 * clamp(frequency, 0.2, 0.8) (metrics.py:172-174).
 * out[2] = {reg_loss, seg_loss}; state (dram_intreg_loss_state_floats(N) floats) feeds the backward. */
size_t dram_intreg_loss_ws_bytes(int N, int64_t S);
int dram_intreg_loss_state_floats(int N);
int dram_intreg_loss_fwd(const float* dense, const float* refined, const float* lobes, const float* lesions,
                         const float* keep, const float* targets, const float* weight, float smoothing,
                         float* out, float* state, void* ws, size_t ws_bytes, int N, int64_t S, void* stream);
/* d(gout[0]*reg + gout[1]*seg) / d dense -> ddense and / d refined -> drefined (NULL when refined is
 * NULL or == dense: the sum goes to ddense); gout is a DEVICE array of 2 floats. */
int dram_intreg_loss_bwd(const float* dense, const float* refined, const float* lobes, const float* lesions,
                         const float* keep, const float* targets, const float* weight, const float* state,
                         const float* gout, float smoothing, float* ddense, float* drefined, int N, int64_t S,
                         void* stream);

/* ---- PCM local attention on the voxel grid: dram/models.py PCM.init_graph 221-258 (the
 *      neighbour graph becomes E stencil offsets), merge_func 259-331 (dot-product and geo families), compute_cross_x
 *      365-397, forward / update_all 333-363.
 * theta, phi: [B,F,D,H,W]; offsets: HOST array of E*3 ints (dz,dy,dx), E <= 128; node i receives from
 * i + offset (in-grid offsets only).  attn: [B,E,D,H,W] = softmax over the node's edges of
 * scale * normalise(act(theta_i . phi_j)); flags: DRAM_PCM_RELU | DRAM_PCM_L2NORM;
 * scale_mode: 0 none, 1 1/sqrt(#edges of the node), 2 1/0.01. */
#define DRAM_PCM_RELU 1
#define DRAM_PCM_L2NORM 2
int dram_pcm_attention_fwd(const float* theta, const float* phi, const int* offsets, int E, int flags,
                           int scale_mode, float* attn, int B, int F, int D, int H, int W, void* stream);
/* ds: scratch [B,E,D,H,W] (receives d loss / d raw dot products); dtheta, dphi: [B,F,D,H,W]. */
int dram_pcm_attention_bwd(const float* theta, const float* phi, const float* attn, const float* dattn,
                           const int* offsets, int E, int flags, int scale_mode, float* ds, float* dtheta,
                           float* dphi, int B, int F, int D, int H, int W, void* stream);
/* The geo variants of merge_func (models.py:287-299: scaled_dot_product_geo, scaled_dot_product_geo_relu, att_is_all) add a
 * positional-encoding term to the appearance term.  On feature planes concatenated as [appearance ; positional] they are
 * the kernels above with the activation confined to the first F_relu planes:
 *     logit = act(sum_{f < F_relu} theta_f phi_f) + sum_{f >= F_relu} theta_f phi_f       (F_relu == F: the plain forms).
 * ds2: second scratch [B,E,D,H,W], needed when F_relu < F and DRAM_PCM_RELU is set (NULL otherwise). */
int dram_pcm_attention_split_fwd(const float* theta, const float* phi, const int* offsets, int E, int flags,
                                 int scale_mode, int F_relu, float* attn, int B, int F, int D, int H, int W, void* stream);
int dram_pcm_attention_split_bwd(const float* theta, const float* phi, const float* attn, const float* dattn,
                                 const int* offsets, int E, int flags, int scale_mode, int F_relu, float* ds, float* ds2,
                                 float* dtheta, float* dphi, int B, int F, int D, int H, int W, void* stream);
/* The merge types without a softmax (models.py:300-302 cosine, 307-320 heu1 / heu2): a per-edge similarity v_e of
 * (theta_i, phi_j), normalised by its sum over the node's edges: attn_e = v_e / (eps + sum_k v_k).
 * mode 0 cosine: v = F.cosine_similarity (eps 1e-8; sum eps 0); 1 heu1: u = theta.phi / (1 + |theta - phi|_1), v = u if
 * u >= 0.03 else 0; 2 heu2: v = relu(u); sum eps 1e-7.  F <= 64.  ds: scratch [B,E,D,H,W].  (The reference forms heu1's masked
 * similarities under torch.no_grad(), models.py:311-314: its attention is a constant of the graph, and the host side does not
 * call the backward entry for mode 1; the entry itself differentiates through the unmasked pairs.) */
int dram_pcm_attention_sum_fwd(const float* theta, const float* phi, const int* offsets, int E, int mode, float* attn,
                               int B, int F, int D, int H, int W, void* stream);
int dram_pcm_attention_sum_bwd(const float* theta, const float* phi, const float* attn, const float* dattn,
                               const int* offsets, int E, int mode, float* ds, float* dtheta, float* dphi, int B, int F,
                               int D, int H, int W, void* stream);
/* out[b,c,i] = sum_e attn[b,e,i] * v[b,c,i+offset_e]   (torch.matmul(f_sm, x_g), models.py:394) */
int dram_pcm_aggregate_fwd(const float* attn, const float* v, const int* offsets, int E, float* out,
                           int B, int C, int D, int H, int W, void* stream);
int dram_pcm_aggregate_bwd(const float* attn, const float* v, const float* dout, const int* offsets, int E,
                           float* dattn, float* dv, int B, int C, int D, int H, int W, void* stream);

/* ---- fused conv -> norm -> ReLU -> conv chains (parts.py:177-187 conv -> norm -> act) ----
 *
 * "Lazy" tensors: between two convolutions the reference materialises norm(y) and relu_(norm(y)) (parts.py:19-31,
 * 49-50).  Here a conv writes its RAW output y once, its epilogue leaves the BatchNorm / GroupNorm moments of y
 * as partials, dram_norm_finalize_parts turns them into the per-row coefficients {a, b} (y_norm = a*y + b), and
 * every consumer of the activated tensor -- the next conv (forward and backward-weights), the 2x2x2 max-pool,
 * the trilinear upsample, the 1x1x1 head -- takes (y, coef, relu) and applies max(a*y + b, relu ? 0 : -inf)
 * while loading.  The activated tensor never exists in HBM; values are bit-identical to dram_row_affine_act's.
 *   coefK: [N*CK][2] floats, NULL = that source is an ordinary tensor;  reluK: apply ReLU after the affine map.
 */

/* number of statistics partials per (n, c) row that dram_conv3d_k3_fwd_fused writes for this shape (0: bad shape) */
int dram_conv3d_k3_stats_parts(int Cin, int Cout, int D, int H, int W);

/* y = conv3d(act1(x1) ++ crop(act2(x2)), w) as dram_conv3d_k3_fwd_cat, each source lazily normalised (above), and --
 * when stats != NULL -- per (row of y, part) {mean, M2, count} of the outputs into stats[N*Cout][nparts][3]
 * (nparts = dram_conv3d_k3_stats_parts; two-pass moments per 64 outputs, no E[x^2]-E[x]^2).  stats and bias
 * exclude each other (a conv followed by a norm has no bias, models.py:78). */
int dram_conv3d_k3_fwd_fused(const float* x1, int C1, const float* coef1, int relu1, const float* x2, int C2,
                             const float* coef2, int relu2, int D2, int H2, int W2, int oz, int oy, int ox,
                             const float* wt, const float* bias, float* y, float* stats, int nparts, int N,
                             int Cout, int D, int H, int W, void* stream);

/* 1 if backward-weights of this shape has the lazy-operand path (the Winograd kernel runs), else materialise x */
int dram_conv3d_k3_wgrad_lazy_ok(int N, int C1, int C2, int Cout, int D, int H, int W);

/* dram_conv3d_k3_wgrad_ex with lazily normalised x sources */
int dram_conv3d_k3_wgrad_fused(const float* x1, int C1, const float* coef1, int relu1, const float* x2, int C2,
                               const float* coef2, int relu2, int D2, int H2, int W2, int oz, int oy, int ox,
                               const float* dy, float* dw, void* ws, size_t ws_bytes, int N, int Cout, int D,
                               int H, int W, void* stream);

/* ---- which 3x3x3 conv kernel the library launches (the reference leaves this choice to cuDNN's heuristics behind
 * nn.Conv3d, parts.py:95..185; here it is a pure function of the shape, exposed so that benchmarks attribute time to
 * the right kernel and tests assert that the kernel they mean to verify is the one that ran) ----
 * Kernel families; the *_choice queries return one of these and write the instantiation's name as rocprofv3 prints
 * it (without namespace and argument list, e.g. "conv3d_k3_wgrad_wz_kernel<16, 2, 4, 2, true>") to name[cap]. */
#define DRAM_K3_FWD_DIRECT 0    /* conv3d_k3_fwd_kernel: direct 27-tap form (first layer, DRAM_CONV_DIRECT=1) */
#define DRAM_K3_FWD_WZ 1        /* conv3d_k3_fwd_wz_kernel: Winograd F(2,3) along z */
#define DRAM_K3_FWD_WZY 2       /* conv3d_k3_fwd_wzy_kernel: Winograd F(2x2,3x3) over (z,y) */
#define DRAM_K3_WGRAD_DIRECT 3  /* conv3d_k3_wgrad_kernel */
#define DRAM_K3_WGRAD_VEC 4     /* conv3d_k3_wgrad_vec_kernel: direct form, 16-byte staging */
#define DRAM_K3_WGRAD_WZ 5      /* conv3d_k3_wgrad_wz_kernel<.., false>: transposed F(2,3) along z, plain x operand */
#define DRAM_K3_WGRAD_WZ_LAZY 6 /* conv3d_k3_wgrad_wz_kernel<.., true>: x normalised + rectified on load */
#define DRAM_K3_WGRAD_C1 7      /* conv3d_k3_wgrad_c1_kernel: first layer (Cin = 1) */
#define DRAM_K3_FWD_C1 8        /* conv3d_k3_fwd_c1_kernel: first layer (Cin = 1), HBM-bound vector kernel */
#define DRAM_K3_WGRAD_WZY 9     /* conv3d_k3_wgrad_wzy_kernel: transposed F(2x2,3x3) over (z,y) */
#define DRAM_K3_KINDS 10

/* forward / backward-data of [N,Cin,D,H,W] -> Cout channels; the destination may be split over two tensors as in
 * dram_conv3d_k3_fwd_ex (dstC2 = 0: one tensor); fused != 0: the dram_conv3d_k3_fwd_fused variant. */
int dram_conv3d_k3_fwd_choice(int Cin, int Cout, int D, int H, int W, int dstC1, int dstC2, int dstD2, int dstH2,
                              int dstW2, int fused, char* name, size_t cap);
/* ... given what the SOURCE of the launch looks like as well: a cropped second source tensor [., srcC2, srcD2, srcH2, srcW2]
 * whose window starts at x offset srcox, and whether a source base pointer is off 16-byte alignment.  The (z,y) kernel
 * fetches rows as aligned 16-byte pieces: such a launch runs the z-only kernel on the same 32x4x2 boxes instead (the count of
 * statistics partials, dram_conv3d_k3_stats_parts, does not depend on it). */
int dram_conv3d_k3_fwd_choice_src(int Cin, int Cout, int D, int H, int W, int dstC1, int dstC2, int dstD2, int dstH2,
                                  int dstW2, int fused, int srcC2, int srcD2, int srcH2, int srcW2, int srcox,
                                  int src_misaligned, char* name, size_t cap);
/* backward-weights with x = x1[.,C1,..] ++ crop(x2[.,C2,..]) (C2 = 0: one tensor); lazy != 0: the *_fused variant
 * with at least one lazily normalised source. */
int dram_conv3d_k3_wgrad_choice(int N, int C1, int C2, int Cout, int D, int H, int W, int lazy, char* name, size_t cap);
/* launches per kernel family (index = DRAM_K3_*) since the library was loaded, into counts[0..n) */
int dram_conv3d_k3_launch_counts(unsigned long long* counts, int n);

/* Training-mode BatchNorm / GroupNorm statistics (parts.py:19-31) from the conv epilogue's partials: save_mean,
 * save_rstd per statistic, rowcoef[N*C][2], running statistics updated as dram_norm_fwd_train does.  Chan's
 * combine in fp64; a total count that differs from the statistic's population poisons it with NaN. */
size_t dram_norm_parts_ws_bytes(int N, int C, int nparts);
int dram_norm_finalize_parts(const float* parts, int nparts, const float* gamma, const float* beta,
                             float* save_mean, float* save_rstd, float* rowcoef, float* running_mean,
                             float* running_var, float momentum, float eps, int kind, int G, int N, int C,
                             int64_t S, void* ws, size_t ws_bytes, void* stream);

/* this rank's per-channel {mean, M2} in fp64 (mean_m2[2c], [2c+1]: the layout of dram_bn_stats) from the same partials: what
 * nn.SyncBatchNorm (parts.py:32-33) exchanges between ranks; ws as dram_norm_parts_ws_bytes */
int dram_bn_parts_stats(const float* parts, int nparts, double* mean_m2, int N, int C, int64_t S, void* ws,
                        size_t ws_bytes, void* stream);

/* eval-mode BatchNorm: rowcoef (and save_mean / save_rstd) from the running statistics, no pass over a tensor */
int dram_bn_eval_coef(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float* save_mean, float* save_rstd, float* rowcoef, float eps, int N, int C, void* stream);

/* y = act(rowcoef[row][0] * x + rowcoef[row][1]): materialise a lazy tensor (rows = N*C) */
int dram_row_affine_act(const float* x, const float* rowcoef, float* y, int relu, int64_t rows, int64_t S,
                        void* stream);

/* nn.MaxPool3d(2,2,0) (parts.py:191) of a lazy tensor; and its backward ADDED onto dx (the skip branch's gradient) */
int dram_maxpool3d_2_fwd_lazy(const float* x, const float* coef, int relu, float* out, uint8_t* idx, int N, int C,
                              int D, int H, int W, void* stream);
int dram_maxpool3d_2_bwd_acc(const float* dout, const uint8_t* idx, float* dx, int N, int C, int D, int H,
                             int W, void* stream);

/* nn.Upsample(trilinear, align_corners=True) (parts.py:149) of a lazy tensor */
int dram_upsample_trilinear_ac_fwd_lazy(const float* x, const float* coef, int relu, float* y, int N, int C, int D,
                                        int H, int W, int Do, int Ho, int Wo, void* stream);

/* top_layer 1x1x1 conv (models.py:109,145) of a lazy tensor, and its backward (dx = gradient w.r.t. act(x)) */
int dram_conv3d_k1_fwd_lazy(const float* x, const float* coef, int relu, const float* w, const float* bias, float* y,
                            int N, int Cin, int Cout, int64_t S, void* stream);
int dram_conv3d_k1_bwd_lazy(const float* dy, const float* x, const float* coef, int relu, const float* w, float* dx,
                            float* dw, float* dbias, void* ws, size_t ws_bytes, int N, int Cin, int Cout, int64_t S,
                            void* stream);

/* ---- affine-consistency losses (IntRegAffRefineLoss, dram/metrics.py:376-462): the two ops they add ----
 * F.sigmoid (metrics.py:434,445) as a differentiable op: y = sigmoid(x); dx = dy * p * (1 - p) recomputed from x. */
int dram_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream);
int dram_sigmoid_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
/* F.smooth_l1_loss(a[m > 0], b[m > 0]) (metrics.py:449,451-452: beta 1, mean) for a, b [N,C,S] and the mask m[N,1,S]
 * expanded over C.  out[0] = loss, out[1] = number of selected elements; deterministic (fp64, fixed order).
 * Backward: da (and/or db = -da) for the upstream gradient gout[0]. */
size_t dram_masked_smooth_l1_ws_bytes(int N, int C, int64_t S);
int dram_masked_smooth_l1_fwd(const float* a, const float* b, const float* mask, float* out, void* ws,
                              size_t ws_bytes, int N, int C, int64_t S, void* stream);
int dram_masked_smooth_l1_bwd(const float* a, const float* b, const float* mask, const float* out,
                              const float* gout, float* da, float* db, int N, int C, int64_t S, void* stream);

/* Rotate3DXOneShot (data_transforms.py:1186-1208): F.grid_sample(x, F.affine_grid(theta, x.size())) with the defaults
 * (trilinear, zeros padding, align_corners=False); theta12 = HOST pointer to the row-major [3][4] matrix shared by all
 * samples.  Backward = the scatter adjoint (float atomics: summation order not fixed, like ATen's). */
int dram_affine_sample_fwd(const float* x, float* y, const float* theta12, int N, int C, int D, int H, int W,
                           void* stream);
int dram_affine_sample_bwd(const float* dy, float* dx, const float* theta12, int N, int C, int D, int H, int W,
                           void* stream);

/* ---- measured ceilings of the device ("calibrate both peaks on the box with a copy kernel and an FMA
 * loop"); no reference counterpart.  Both only launch: the caller brackets them with HIP events on `stream`.
 * dram_calibrate_hbm_copy: dst = src, 16 bytes per lane (2 x nbytes of HBM traffic).
 * dram_calibrate_mfma_f32: a register-only v_mfma_f32_32x32x2_f32 loop, `blocks` x 8 waves x iters x 32 MFMAs; *flops (may be
 *   NULL) receives the launch's FLOP count; sink: blocks * 512 floats. ---- */
int dram_calibrate_hbm_copy(const void* src, void* dst, size_t nbytes, void* stream);
int dram_calibrate_mfma_f32(float* sink, int blocks, int iters, double* flops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DRAM_HIP_H */
