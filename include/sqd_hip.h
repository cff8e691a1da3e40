/* sqd_hip.h -- C ABI of libsqdhip.so: the MI355X (gfx950) SqueezeDet hot path.
 *
 * The reference (hazenai/SqueezeDet-PyTorch) is pure Python over torch.nn; it has no FFI of its own.
 * The seam this library plugs into is the nn.Module surface; every entry point
 * below names the reference code it replaces.  Conventions:
 *   - plain pointers to DEVICE memory, explicit sizes, a hipStream_t passed as void*; no torch types;
 *   - every function only enqueues work on `stream` (never synchronises, never allocates), so it can
 *     be captured into a hipGraph; the library keeps no mutable global state and is re-entrant;
 *   - return value: 0 ok, 1 bad argument, 2 unsupported configuration, 3 launch failure;
 *   - activations are NHWC fp32; a tensor argument is described by (pitch, coff, C): the layer
 *     touches channels [coff, coff+C) of a buffer whose pixel stride is `pitch` floats.  All channel
 *     counts / pitches / offsets are multiples of 4 and all base pointers 16-byte aligned.
 */
#ifndef SQD_HIP_H
#define SQD_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tile configurations of the implicit-GEMM convolution -------------------------------------- */
int sqd_conv_num_cfgs(void);
/* taps (1 or 9), K-chunk, pixels per workgroup tile, output channels per workgroup slice */
int sqd_conv_cfg_info(int cfg_id, int* taps, int* kc, int* tile_px, int* bn);
/* 0: register-staged, 4 waves; 1: LDS-DMA double buffering, 4 waves; 2: LDS-DMA, 8 waves (xmask unsupported
 * when != 0); -1: bad id */
int sqd_conv_cfg_is_dma(int cfg_id);

/* Convolution 1x1 or 3x3/pad 1, stride 1, on the fp32 matrix cores, with fused bias / ReLU /
 * accumulate and an optional ReLU mask applied to the input (x * (xmask > 0)).
 * Replaces nn.Conv2d + nn.ReLU(inplace) of Fire.squeeze / expand1x1 / expand3x3
 * (src/model/squeezedet.py:12-14,18-22), torch.cat (:19-22, via y_coff), ConvDet (:73-75,:83) and,
 * with transposed+flipped weights, their autograd data-gradients (src/engine/trainer.py:47).
 * w_packed: [ceil(C/kc)][taps][Npad][kc] fp32, zero padded, Npad = ceil(N/bn)*bn (kc, bn from cfg). */
/* Epilogue order: acc (+bias) (+= y if accumulate) (*= ymul) (zero where ymask <= 0) (ReLU) -> y.
 * ymask / ymul address the same pixels as y with their own (pitch, coff); they serve the backward pass
 * (ReLU mask of the layer that produced this gradient's forward activation; dropout mask). */
int sqd_conv_fwd(const float* x, const float* w_packed, const float* bias, float* y, const float* xmask,
                 const float* ymask, const float* ymul, int B, int H, int W, int C, int x_pitch, int x_coff,
                 int N, int Npad, int y_pitch, int y_coff, int relu, int accumulate, int xmask_pitch,
                 int xmask_coff, int ymask_pitch, int ymask_coff, int ymul_pitch, int ymul_coff, int cfg_id,
                 void* stream);

/* Canonical OIHW parameter [No][Ci][k][k] -> packed layout of sqd_conv_fwd for the forward conv
 * (dgrad=0: N=No, C=Ci) or for its data-gradient conv (dgrad=1: N=Ci, C=No, taps flipped). */
int sqd_pack_conv_weight(const float* w_oihw, float* w_packed, int No, int Ci, int taps, int kc, int Npad,
                         int dgrad, void* stream);

/* The same for n weights in ONE launch (training: every packed copy is refreshed after each optimizer step).
 * descs_dev: device array of n records of ten int64 {w ptr, out ptr, No, Ci, taps, kc, Npad, ceil(C/kc), dgrad,
 * total output floats}; grid = blocks_per_desc x n. */
int sqd_pack_conv_weights_batched(const void* descs_dev, int n, int blocks_per_desc, void* stream);

/* Weight + bias gradient of a 1x1 / 3x3 conv (autograd of nn.Conv2d, src/engine/trainer.py:47).
 * dy window [dy_coff, dy_coff+N) must already carry its ReLU mask; slab = workspace of
 * S*(N*taps*C + N) floats (S pixel-splits, summed in fixed order => bitwise reproducible);
 * dw is OIHW [N][C][k][k], db [N] (may be NULL). */
int sqd_conv_wgrad(const float* dy, const float* x, float* slab, float* dw, float* db, int B, int H, int W,
                   int N, int dy_pitch, int dy_coff, int C, int x_pitch, int x_coff, int taps, int S,
                   void* stream);
/* A Fire squeeze's whole backward in ONE launch (autograd of Fire.squeeze + squeeze_activation, src/model/squeezedet.py:12,19,
 * as triggered by loss.backward(), src/engine/trainer.py:47): the weight / bias gradient slabs of sqd_conv_wgrad (taps = 1,
 * partial slabs only: reduce with sqd_wgrad_reduce_batched) and the data gradient
 * dx[p][dx_coff + c] = sum_n dy[p][dy_coff + n] * w[n][c], zeroed where relu_mask != 0 and x[p][x_coff + c] <= 0 (x = the
 * squeeze's forward input when that is a ReLU output).  w_oihw: the layer's own [N][C][1][1] parameter.  N <= 128.  Also serves
 * Fire.expand1x1 (dy = the expand1x1 window of the Fire output's gradient, x = the squeeze output, relu_mask = 0).
 * S pixel-splits as in sqd_conv_wgrad with 64-channel in-tiles: slab = S * (N*C + N) floats. */
int sqd_squeeze_bwd(const float* dy, const float* x, const float* w_oihw, float* slab, float* dx, int B, int H, int W,
                    int N, int dy_pitch, int dy_coff, int C, int x_pitch, int x_coff, int dx_pitch, int dx_coff,
                    int relu_mask, int S, void* stream);
/* Winograd F(2x2,3x3) form of sqd_conv_wgrad for the 3x3 layers (Fire expand3x3, src/model/squeezedet.py:14; autograd of
 * nn.Conv2d as triggered by loss.backward(), src/engine/trainer.py:47): same arguments (taps is 9), slab layout and
 * dw == NULL convention; executes 2.25x fewer multiply-adds.  Supported: (N % 64 == 0 or N <= 80) and S <= B * ceil(H/4) * ceil(W/16);
 * SQD_ERR_UNSUPPORTED otherwise (use sqd_conv_wgrad).  tc = input-channel blocks of 16 per workgroup (1 or 2; N % 64 != 0
 * always runs 1). */
int sqd_conv_wgrad_wino(const float* dy, const float* x, float* slab, float* dw, float* db, int B, int H, int W,
                        int N, int dy_pitch, int dy_coff, int C, int x_pitch, int x_coff, int S, int tc, void* stream);
/* The slabs of up to 6 such layers that share B, H, W (the expand3x3 layers of one backward stage: the runs of Fire modules between two
 * pools, src/model/squeezedet.py:33-49) in ONE launch, every layer cut into the same S splits: S then only has to fill the chip once over
 * ALL the layers, which divides the slab bytes and the per-workgroup epilogues by the number of layers.  layers = host array of n records of
 * 9 int64 {dy, x, slab (device pointers), N, dy_pitch, dy_coff, C, x_pitch, x_coff}; slab i holds S * (N_i*9*C_i + N_i) floats.  Every
 * layer needs N % 64 == 0 and must select tile form tc (tc = 1: C < 32 or C % 32 == 16; tc = 2 otherwise); SQD_ERR_UNSUPPORTED otherwise.
 * Bitwise the slabs of n sqd_conv_wgrad_wino(dw = NULL) calls with the same S and tc. */
int sqd_conv_wgrad_wino_group(const long long* layers, int n, int B, int H, int W, int S, int tc, void* stream);
/* The same for 1x1 layers (the expand1x1 convolutions of a stage whose width rules out the fused squeeze backward, Fire.expand1x1,
 * src/model/squeezedet.py:13,20): records as above, slab i holds S * (N_i*C_i + N_i) floats.  Every layer must select the same tile form
 * sqd_conv_wgrad would give it (N >= 64 and not 64 < N <= 96; in-channel tile 16 * ceil(C / 16) up to 64, 128 from C = 256) and
 * S <= the number of its pixel blocks; SQD_ERR_UNSUPPORTED otherwise.  Bitwise the slabs of n sqd_conv_wgrad(dw = NULL, taps = 1) calls
 * with the same S. */
int sqd_conv_wgrad_group(const long long* layers, int n, int B, int H, int W, int S, void* stream);

/* dw == NULL in sqd_conv_wgrad: write the S partial slabs only; the caller then reduces many layers with ONE launch:
 * descs_dev = device array of n records of 9 int64 {slab offset, dw offset, db offset (floats from slab_base / grad_base;
 * db offset < 0: no bias gradient), S, slab stride (= N*taps*C + N), N, C, taps, first workgroup of the record}
 * (a record takes ceil((N*taps*C + N) / 64) workgroups; total_blocks = their sum; slab_base 16-byte aligned, slab offsets and strides
 * multiples of 4 floats -- they are, with the N % 4 == 0, C % 4 == 0 the slab writers require).  Results are bitwise those of the
 * per-layer reduction, times `scale` (1 = exactly the sum; the data-parallel exchange passes the rank's image count: the
 * weighting of src/utils/data_parallel.py's gathered loss mean, src/engine/trainer.py:43). */
int sqd_wgrad_reduce_batched(const void* descs_dev, int n, int total_blocks, const float* slab_base, float* grad_base, float scale,
                             void* stream);
/* The same for a contiguous range of the table: descs_dev = address of the range's first record, n = records in the
 * range, block_first = that record's "first workgroup" field, nblocks = workgroups of the range.  Lets the backward of
 * src/engine/trainer.py:47 hand each finished stage's gradient bucket to the all-reduce early. */
int sqd_wgrad_reduce_batched_range(const void* descs_dev, int n, int block_first, int nblocks, const float* slab_base,
                                   float* grad_base, float scale, void* stream);
/* Element-wise steps of the gradient exchange that replaces the reference's DataParallel (src/engine/trainer.py:82-92,
 * src/utils/data_parallel.py:93-113): g[0..n) = g * mul, divided by *div_by when div_by != NULL (a DEVICE scalar: the image count
 * summed by the same all-reduce); fill_ptr (or NULL, outside [g, g+n)) receives fill_value (this rank's image count).  n may be 0
 * with fill_ptr given. */
int sqd_grad_scale(float* g, long long n, float mul, const float* div_by, float* fill_ptr, float fill_value, void* stream);


/* Stem weight + bias gradient (the image needs no data gradient).  slab: S*(N*3*k*k + N) floats. */
int sqd_stem_wgrad(const float* dy_nhwc, const float* img_nchw, float* slab, float* dw_oihw, float* db,
                   int B, int Hin, int Win, int N, int ksize, int S, void* stream);

/* The same when the forward ran fused (sqd_stem_conv_relu_pool_fwd): the backward of ReLU + MaxPool is folded
 * into the staging.  dpool: NHWC [B][Hp][Wp][N]; argmax: the uint8 tensor written by the forward, whose codes carry the
 * ReLU mask (15 = pooled value not > 0); pooled (NHWC like dpool) may be NULL -- when given, `pooled > 0` is ANDed in. */
int sqd_stem_wgrad_pooled(const float* dpool, const float* pooled, const unsigned char* argmax, const float* img_nchw,
                          float* slab, float* dw_oihw, float* db, int B, int Hin, int Win, int N, int ksize, int S,
                          void* stream);

/* Stem: Conv2d(3, N, k, stride 2, pad k/2) + ReLU, NCHW image -> NHWC features.
 * (k,N) = (3,64) squeezedet (src/model/squeezedet.py:34-35) or (7,96) squeezedetplus (:52-53).
 * w is the checkpoint tensor itself (OIHW). */
int sqd_stem_conv_relu_fwd(const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc,
                           int B, int Hin, int Win, int N, int ksize, void* stream);
/* The same with the activation optional (relu = 0: the bare Conv2d of features[0], src/model/squeezedet.py:34, for
 * callers that walk the module list layer by layer). */
int sqd_stem_conv_fwd(const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc,
                      int B, int Hin, int Win, int N, int ksize, int relu, void* stream);

/* Fused features[0..2]: conv + ReLU + MaxPool2d(3,2,ceil_mode=True) (src/model/squeezedet.py:34-36 / :52-54)
 * without materialising the conv output.  y: NHWC [B][Hp][Wp][N]; argmax (uint8, may be NULL): window position 0..8 of
 * the maximum, or 15 where the pooled value is 0 (the ReLU mask the backward needs, as sqd_maxpool3x3s2_ceil_fwd_relu). */
int sqd_stem_conv_relu_pool_fwd(const float* x_nchw, const float* w_oihw, const float* bias, float* y_nhwc,
                                unsigned char* argmax, int B, int Hin, int Win, int N, int ksize, void* stream);

/* features[0..2] AND the first Fire's squeeze (src/model/squeezedet.py:34-37 with :17-18: Fire.squeeze + squeeze_activation) in one
 * launch, inference only: y NHWC [B][Hp][Wp][nsq] = ReLU(conv1x1(MaxPool(ReLU(conv(x))))); the pooled 64-channel tensor is never
 * written.  w_sq: the squeeze's OIHW weight [nsq][N][1][1], b_sq its bias.  Supported: ksize 3, N 64, nsq 16, Win % 4 == 0, x
 * 16-byte aligned; anything else returns SQD_ERR_UNSUPPORTED and the caller keeps the two launches. */
int sqd_stem_pool_squeeze_fwd(const float* x_nchw, const float* w_oihw, const float* bias, const float* w_sq, const float* b_sq,
                              float* y_nhwc, int B, int Hin, int Win, int N, int ksize, int nsq, void* stream);

/* Training form of sqd_stem_pool_squeeze_fwd (the forward of src/engine/trainer.py:42 through src/model/squeezedet.py:34-37, 17-18): the
 * pooled tensor y_pooled [B][Hp][Wp][N] and its codes (argmax, as sqd_stem_conv_relu_pool_fwd) are stored -- the backward reads both --
 * and the first Fire's squeeze output y_sq [B][Hp][Wp][nsq] leaves the same launch.  Same limits as sqd_stem_pool_squeeze_fwd. */
int sqd_stem_pool_squeeze_train_fwd(const float* x_nchw, const float* w_oihw, const float* bias, const float* w_sq, const float* b_sq,
                                    float* y_pooled, unsigned char* argmax, float* y_sq, int B, int Hin, int Win, int N, int ksize,
                                    int nsq, void* stream);

/* nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True) (src/model/squeezedet.py:36,39,42), NHWC.
 * argmax (uint8, same shape as y, may be NULL) records the window position 0..8 for the backward. */
int sqd_maxpool3x3s2_ceil_fwd(const float* x, float* y, unsigned char* argmax, int B, int H, int W, int C,
                              void* stream);
/* The same pool when its input is a ReLU output and a backward will follow (the training forward: every pool of
 * src/model/squeezedet.py:33-49 sits behind a ReLU, :36,:39,:42): argmax (required) additionally carries the ReLU mask of the
 * input -- code 15 where the pooled value is not > 0 -- so the backward below needs no relu_src. */
int sqd_maxpool3x3s2_ceil_fwd_relu(const float* x, float* y, unsigned char* argmax, int B, int H, int W, int C,
                                   void* stream);
/* relu_src (may be NULL): the pool's forward input when that was a ReLU output; its mask is folded in (not needed -- pass NULL --
 * with the codes of sqd_maxpool3x3s2_ceil_fwd_relu). */
int sqd_maxpool3x3s2_ceil_bwd(const float* dy, const unsigned char* argmax, float* dx, const float* relu_src,
                              int B, int H, int W, int C, void* stream);

/* Dense decode: PredictionResolver.forward + the argmax/max of SqueezeDet.forward
 * (src/model/squeezedet.py:109-120,199-202; src/model/modules.py:17-45,66-68).
 * pred [B][A][num_classes+5], anchors [A][4] (cx,cy,w,h) -> class_ids int64 [B][A], scores [B][A],
 * boxes [B][A][4] xyxy clamped to the input. */
int sqd_decode_fwd(const float* pred, const float* anchors, long long* class_ids, float* scores, float* boxes,
                   int B, int A, int num_classes, int input_h, int input_w, void* stream);

/* PredictionResolver.forward itself (src/model/squeezedet.py:109-120): the reference's five dense outputs.
 * probs / logp [B][A][C] (logp may be NULL = log_softmax=False), scores [B][A], deltas / boxes [B][A][4]. */
int sqd_resolve_fwd(const float* pred, const float* anchors, float* probs, float* logp, float* scores, float* deltas,
                    float* boxes, int B, int A, int num_classes, int input_h, int input_w, void* stream);

/* Fused decode + Detector.filter for a whole batch (src/engine/detector.py:87-122 + torchvision nms +
 * the scale division of boxes_postprocess, src/utils/boxes.py:145-147): top keep_top_k (<= 64) by score,
 * class-wise NMS, score threshold, compacted in class order.  Fixed-capacity outputs [B][keep_top_k];
 * det_count[b] rows are valid.  det_anchor = anchor index of every kept detection (not returned by the
 * reference; this is what "box indices bit-exact" is asserted on).  scales [B][2]=(sy,sx) or NULL.
 * ONE kernel launch (one workgroup per image; scores, selection and NMS never leave the LDS).  keys_ws: unused, may be
 * NULL (kept so that callers of the earlier two-launch version need not change). */
int sqd_detect_fwd(const float* pred, const float* anchors, const float* scales, unsigned* keys_ws, int* det_count,
                   long long* det_class, float* det_score, float* det_box, int* det_anchor, int B, int A,
                   int num_classes, int input_h, int input_w, int keep_top_k, float nms_thresh,
                   float score_thresh, void* stream);

/* The same with shifts ([B][2] = (dy, dx), may be NULL) added to the boxes after the scale division: the padding / crops terms of
 * boxes_postprocess (src/utils/boxes.py:149-155) for images pre-processed by sqd_preprocess_u8_padcrop_fwd (the reference's
 * cfg.forbid_resize branch); per axis only one of padding / crops is non-zero, so one add of (crops - padding) is bit-exact.
 * keys_ws: NULL (one workgroup per image, as sqd_detect_fwd) or a workspace of B * ceil4(A) + B uint32 whose last B words are zero
 * before the first launch (every launch leaves them zero): the anchors of an image are then scored by eight workgroups and the
 * image's last arriver selects and suppresses -- same results bit for bit. */
int sqd_detect_shift_fwd(const float* pred, const float* anchors, const float* scales, const float* shifts, unsigned* keys_ws,
                         int* det_count, long long* det_class, float* det_score, float* det_box, int* det_anchor, int B, int A,
                         int num_classes, int input_h, int input_w, int keep_top_k, float nms_thresh, float score_thresh,
                         void* stream);

/* Detector.filter on already decoded dense tensors (class_ids int64 [B][A], scores [B][A], boxes [B][A][4]). */
int sqd_filter_fwd(const long long* class_ids, const float* scores, const float* boxes, unsigned* keys_ws, int* det_count,
                   long long* det_class, float* det_score, float* det_box, int* det_anchor, int B, int A,
                   int num_classes, int keep_top_k, float nms_thresh, float score_thresh, void* stream);

/* GPU-side input pipeline: whiten + cv2.resize(INTER_LINEAR) + HWC->CHW of
 * DataWrapper.__getitem__ / BaseDataset.preprocess / whiten / resize (src/engine/detector.py:132-142,
 * src/datasets/base.py:43-59, src/utils/image.py:9-19,77-88) for a batch of uint8 RGB images of arbitrary sizes.
 * src: device buffer with the images back to back (HWC); offsets [B] (bytes); sizes [B][2] = (H0, W0);
 * out: NCHW fp32 [B][3][H][W]; scales [B][2] = (H/H0, W/W0) or NULL; mean3 / std3: HOST pointers to 3 floats. */
int sqd_preprocess_u8_fwd(const unsigned char* src, const long long* offsets, const int* sizes, float* out,
                          float* scales, const float* mean3, const float* std3, int B, int H, int W, void* stream);

/* The input pipeline's OTHER branch, cfg.forbid_resize (src/datasets/base.py:53-54): whiten (src/utils/image.py:9-19), then
 * crop_or_pad (:91-124: per axis zero-pad a smaller image / centre-crop a larger one to the target, floor half in front),
 * HWC->CHW.  Integer index arithmetic + whiten's one float32 subtract and divide: bit-exact against the reference.
 * Arguments as sqd_preprocess_u8_fwd; shifts [B][2] fp32 = (crops[0] - padding[0], crops[2] - padding[2]) for
 * sqd_detect_shift_fwd, or NULL; padcrop [B][8] int32 = padding (top, bottom, left, right), crops (top, bottom, left, right),
 * or NULL. */
int sqd_preprocess_u8_padcrop_fwd(const unsigned char* src, const long long* offsets, const int* sizes, float* out, float* shifts,
                                  int* padcrop, const float* mean3, const float* std3, int B, int H, int W, void* stream);

/* Fused Fire expand (Fire.forward, src/model/squeezedet.py:18-22: expand1x1 and expand3x3 of the squeeze output,
 * concatenated): y[..., y_coff : y_coff+E] = ReLU(conv1x1(x)), y[..., y_coff+E : y_coff+2E] = ReLU(conv3x3(x)) in ONE
 * launch.  w_packed / bias: the 2E output channels in alternating 16-channel groups (group 2i = expand1x1 channels
 * 16i..16i+15 written as a 3x3 kernel whose only non-zero tap is the centre, group 2i+1 = expand3x3 channels
 * 16i..16i+15), packed by sqd_pack_conv_weight for cfg_id; cfg_id must be a 3x3 LDS-DMA configuration with an even
 * number of 16-channel groups per slice whose slice width divides 2E (+ 1000 * k = workgroups-per-CU cap). */
int sqd_fire_expand_fwd(const float* x, const float* w_packed, const float* bias, float* y, int B, int H, int W, int C,
                        int x_pitch, int x_coff, int E, int Npad, int y_pitch, int y_coff, int cfg_id, void* stream);

/* Winograd F(2x2,3x3) form of the 3x3 / pad 1 convolution (same layers as sqd_conv_fwd with a 3x3 configuration: Fire
 * expand3x3, src/model/squeezedet.py:14,20-22; ConvDet, :73-75,83): y[..., y_coff : y_coff+N] = (ReLU)(conv3x3(x[...,
 * x_coff : x_coff+C]) + bias), fp32, within fp32 rounding of the direct form.  u_packed = sqd_pack_wino_weight output
 * (C/8 * 16 * Npad * 8 floats, Npad = N rounded up to the configuration's slice width); C % 8 == 0.  Epilogue options as
 * in sqd_conv_fwd, for the data-gradient use: accumulate != 0 adds to y; ymul / ymask (NULL or NHWC tensors with the SAME
 * pixel pitch and channel offset as y) multiply the result (dropout) / zero it where ymask <= 0 (ReLU backward).  cfg_id in
 * [0, sqd_wino_num_cfgs()) (+ 1000 * k = workgroups-per-CU cap); sqd_wino_cfg_info reports slice width and waves. */
int sqd_wino_num_cfgs(void);
int sqd_wino_cfg_info(int cfg_id, int* bn, int* waves);
int sqd_conv_wino_fwd(const float* x, const float* u_packed, const float* bias, float* y, const float* ymask, const float* ymul,
                      int B, int H, int W, int C, int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff, int relu,
                      int accumulate, int cfg_id, void* stream);
/* The same convolution for NARROW outputs and long reductions -- ConvDet (Conv2d(768 -> 72, 3, padding 1), src/model/squeezedet.py:
 * 73-75,83; squeezedetplus 512 -> 72): N <= 80, u_packed = sqd_pack_wino_weight output with Npad = 80.  Twelve-wave workgroups of
 * (4x16-pixel group, 16-channel block) units whose waves share the group's transformed input through LDS (csrc/conv_wino_vs.hip);
 * bias + optional ReLU epilogue only.  Results equal sqd_conv_wino_fwd's bit for bit. */
int sqd_conv_wino_vs_fwd(const float* x, const float* u_packed, const float* bias, float* y, int B, int H, int W, int C, int x_pitch,
                         int x_coff, int N, int Npad, int y_pitch, int y_coff, int relu, void* stream);
/* Transformed weights U = G g G^T of an OIHW [No][Ci][3][3] parameter into the layout above; dgrad != 0 packs the
 * data-gradient orientation (in/out channels swapped, taps flipped). */
int sqd_pack_wino_weight(const float* w_oihw, float* u_packed, int No, int Ci, int Npad, int dgrad, void* stream);
/* The same for many plans in ONE launch (after an optimizer step).  descs_dev: device array of n records of 7 int64
 * {w_oihw ptr, u_packed ptr, No, Ci, Npad, dgrad, C/8 * Npad * 8}; blocks_per_desc workgroups walk each record. */
int sqd_pack_wino_weights_batched(const void* descs_dev, int n, int blocks_per_desc, void* stream);

/* The same convolution with a BALANCED (stream-K) work split (csrc/conv_wino_sk.hip): same layers (Fire expand3x3,
 * src/model/squeezedet.py:14,20-22; ConvDet, :73-75,83; their data gradients, src/engine/trainer.py:47), same operands
 * (u_packed from sqd_pack_wino_weight with Npad a multiple of 32) and epilogue order (+= y, * ymul, * yscale, mask, ReLU).
 * Every resident workgroup gets an equally long run of (4- or 8-group super-group, slice, K chunk) stages; a slice whose upper 16
 * channels do not exist (N = 72) runs as 16-channel stages over two groups per wave; units cut by a run boundary are summed in
 * part order through `ws` by the last arriver (bitwise reproducible).
 *   sqd_wino_sk_grid(): workgroups per launch on the current device (two per CU);
 *   sqd_wino_sk_schedule (HOST arrays, no GPU needed): ngroups = B * ceil(H/4) * ceil(W/16); minseg = shortest part in stages;
 *     h_bias_pm = per-mille correction of the 16-channel class's share of the grid (1000 = proportional); ksplit = 0: contiguous
 *     ("stream-K") runs, k >= 1: every unit cut at the same K boundaries into k parts dealt round-robin; writes seg_off [G + 1],
 *     segs [<= max_segs][8] = {super-group, first channel, first stage, end stage, class, parts, part, first slab}, the record
 *     count and the number of partial slabs;
 *   sqd_conv_wino_sk_fwd: seg_off / segs = DEVICE copies of that schedule; ws = nslabs * 8192 floats; cnt = nslabs * 4 unsigned,
 *     zero before the first launch (every launch leaves them zero). */
int sqd_wino_sk_grid(void);
int sqd_wino_sk_schedule(int ngroups, int N, int C, int G, int minseg, int h_bias_pm, int ksplit, int* seg_off, int* segs, int max_segs,
                         int* nsegs_out, int* nslabs_out);
int sqd_conv_wino_sk_fwd(const float* x, const float* u_packed, const float* bias, float* y, const float* ymask, const float* ymul,
                         float yscale, int B, int H, int W, int C, int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff,
                         int relu, int accumulate, const int* seg_off, const int* segs, int G, int nslabs, float* ws, unsigned* cnt,
                         const unsigned long long* drop_state, int drop_keep16, float drop_scale, unsigned long long* drop_advance,
                         void* stream);

/* Counter-based dropout in front of ConvDet (nn.Dropout(p, inplace=True), src/model/squeezedet.py:71-72,81-82; csrc/sqd_common.h):
 * the keep decision of element e of the dropped NHWC tensor is a pure function of (seed, step, e) -- one 64-bit hash per four
 * consecutive elements, keep where a 16-bit field < keep16 = round((1 - p) * 65536), kept values scaled by scale = 1 / (1 - p).
 * state: DEVICE uint64[2] = {seed, step}.  The last Fire's expand launches apply it in their epilogues (sqd_conv_drop_fwd: the
 * weight-stationary 1x1 configurations, sqd_conv_cfg_is_dma >= 3; sqd_conv_wino_sk_fwd's drop_state) with e = element index in
 * the OUTPUT BUFFER (pixel * y_pitch + y_coff + channel); sqd_dropout_mask_fwd writes the same multipliers (scale or 0) of
 * elements [0, 4 * n4) as a tensor (layer configurations without a fused epilogue; witness of the fused ones);
 * sqd_dropout_advance adds 1 to step (once per forward; sqd_conv_wino_sk_fwd's drop_advance does the same inside a launch). */
int sqd_conv_drop_fwd(const float* x, const float* w_packed, const float* bias, float* y, int B, int H, int W, int C, int x_pitch,
                      int x_coff, int N, int Npad, int y_pitch, int y_coff, int relu, const unsigned long long* drop_state, int keep16,
                      float scale, int cfg_id, void* stream);
int sqd_dropout_mask_fwd(const unsigned long long* state, int keep16, float scale, float* mask, long long n4, void* stream);
int sqd_dropout_advance(unsigned long long* state, void* stream);

/* One wave that idles for `us` microseconds (0 <= us <= 20000) on `stream`.  No reference counterpart: the lane executor of the
 * inference driver (Detector.detect_dataset, src/engine/detector.py:52-85; lanes.py here) uses it once to detect which of its
 * streams share a hardware queue (work queued behind the spin on another stream = aliased). */
int sqd_spin_us(int us, void* stream);

/* Fused MaxPool2d(3, 2, ceil_mode) + Fire squeeze 1x1 + ReLU, inference forward (src/model/squeezedet.py:39,42 followed
 * by :12,18): y[..., y_coff : y_coff+N] = ReLU(conv1x1(pool(x[..., x_coff : x_coff+C])) + bias); the pooled tensor is never
 * materialised.  x NHWC [B][H][W][x_pitch], y NHWC [B][Ho][Wo][y_pitch]; w_packed = sqd_pack_conv_weight output for a 1x1
 * configuration whose KC divides C with Npad = N rounded up to 16 (Npad <= 96). */
int sqd_pool_squeeze_fwd(const float* x, const float* w_packed, const float* bias, float* y, int B, int H, int W, int C,
                         int x_pitch, int x_coff, int N, int Npad, int y_pitch, int y_coff, void* stream);

/* On-device GT encoding: compute_deltas (src/utils/boxes.py:84-135: greedy unique
 * anchor assignment by free-anchor IoU, nearest free anchor by squared (cx,cy,w,h) distance when no free anchor
 * overlaps) + BaseDataset.prepare_annotations (src/datasets/base.py:61-76: dense gt row = mask, xyxy, deltas,
 * one-hot).  boxes [total][4] xyxy fp32, class_ids [total] int32, box_offsets [B+1] int32, anchors [A][4] FLOAT64
 * (cx,cy,w,h; the reference keeps them in float64 and the overlaps are float64 arithmetic).  Outputs, each may be
 * NULL: gt [B][A][C+9] (fully overwritten), anchor_idx [total] int32 (A = unassigned), deltas [total][4] fp32.
 * Ties in overlap / distance -> lowest anchor index (the reference leaves them to numpy's unstable argsort).
 * workspace: 16 * total_boxes bytes (device) or NULL: enables the parallel first-choice pass (same results). */
int sqd_encode_gt_fwd(const float* boxes, const int* class_ids, const int* box_offsets, const double* anchors,
                      float* gt, int* anchor_idx, float* deltas, void* workspace, int total_boxes, int B, int A,
                      int num_classes, void* stream);

/* KITTI 2D detection AP, host code: what KITTI.evaluate (src/datasets/kitti.py:99-124)
 * obtains from the evaluate_object binary (src/utils/kitti-eval/cpp/evaluate_object.cpp:281-571).  n_images frames;
 * gt_off / det_off [n+1]; gt_kind: 0 car, 1 pedestrian, 2 cyclist, 3 van, 4 person_sitting, 5 DontCare, 6 other;
 * boxes x1,y1,x2,y2 (float64); det_cls 0..2 (else not evaluated).  ap [3][3] = class x (easy, moderate, hard);
 * precision [3][3][41] or NULL; evaluated [3].  All pointers are HOST pointers. */
int sqd_kitti_ap(int n_images, const int* gt_off, const int* gt_kind, const double* gt_box, const double* gt_trunc,
                 const int* gt_occ, const int* det_off, const int* det_cls, const double* det_box,
                 const double* det_score, double* ap, double* precision, int* evaluated);

/* Multi-task loss (Loss.forward, src/model/squeezedet.py:133-174; compute_overlaps, modules.py:48-63).
 * pred [B][A][C+5], gt [B][A][C+9] = (mask, x1,y1,x2,y2, dx,dy,dw,dh, onehot[C]), anchors [A][4].
 * workspace: B*16*5 floats.  losses: [4][B] = (class, score = pos+neg, bbox, total); nobj: [B]. */
int sqd_loss_fwd(const float* pred, const float* gt, const float* anchors, float* workspace, float* losses,
                 float* nobj, int B, int A, int num_classes, int input_h, int input_w, float w_class,
                 float w_pos, float w_neg, float w_bbox, void* stream);
/* Analytic backward incl. the un-detached IoU path.  coef [3][B]: upstream gradient of (class, score, bbox)
 * per image (the gradient of `total` already added to each).  dpred [B][A][C+5]. */
int sqd_loss_bwd(const float* pred, const float* gt, const float* anchors, const float* nobj, const float* coef,
                 float* dpred, int B, int A, int num_classes, int input_h, int input_w, float w_class,
                 float w_pos, float w_neg, float w_bbox, void* stream);
/* `loss.mean()` + its backward (src/engine/trainer.py:43-47) inside the loss launches: sqd_loss_mean_fwd also writes mean4 [4] =
 * batch means of (class, score, bbox, total); sqd_loss_mean_bwd takes gmean (DEVICE float: gradient arriving at mean(total), 1 for
 * loss.mean().backward()) and gives every image's components gmean / B.  No torch reduction / elementwise kernel in the step. */
int sqd_loss_mean_fwd(const float* pred, const float* gt, const float* anchors, float* workspace, float* losses, float* nobj,
                      float* mean4, int B, int A, int num_classes, int input_h, int input_w, float w_class, float w_pos,
                      float w_neg, float w_bbox, void* stream);
int sqd_loss_mean_bwd(const float* pred, const float* gt, const float* anchors, const float* nobj, const float* gmean, float* dpred,
                      int B, int A, int num_classes, int input_h, int input_w, float w_class, float w_pos, float w_neg,
                      float w_bbox, void* stream);


/* Fire.forward's two expand convolutions + torch.cat (src/model/squeezedet.py:18-22) in ONE Winograd launch (inference):
 * y[..., y_coff3 : +N3] = ReLU(conv3x3(x, w3) + b3), y[..., y_coff1 : +N1] = ReLU(conv1x1(x, w1) + b1).  A 1x1 convolution only
 * occupies the four inner Winograd positions, so expand1x1 rides along as extra workgroup slices (128 channels x 4 positions
 * each) that reuse the squeeze tile's staging instead of a second launch re-reading it.
 * sqd_pack_wino_fire: w3 OIHW [N3][C][3][3] + w1 OIHW [N1][C][1][1] -> u_packed of (C/8)*16*Npad_total*8 floats,
 * Npad_total = ceil32(N3) + 32*ceil(N1/128).  cfg_id of sqd_fire_wino_fwd: 4 / 6 (streamed U, 8 / 4 waves), 8 / 10
 * (U-stationary: whole U in LDS, needs (C/8)*16 KB + patch ring <= 160 KB), 12 (C <= 16: one workgroup stream runs ALL channel
 * passes of a pixel group from a transformed input held in registers; whole U in LDS), + 1000*k = workgroups-per-CU cap. */
int sqd_pack_wino_fire(const float* w3_oihw, const float* w1_oihw, float* u_packed, int N3, int N1, int C, int Npad_total, void* stream);
int sqd_fire_wino_fwd(const float* x, const float* u_packed, const float* bias3, const float* bias1, float* y, int B, int H, int W,
                      int C, int x_pitch, int x_coff, int N3, int y_coff3, int N1, int y_coff1, int Npad_total, int y_pitch,
                      int cfg_id, void* stream);

/* Fire k's expand pair + torch.cat + Fire k+1's squeeze (src/model/squeezedet.py:18-22 applied twice; features[k] -> the
 * squeeze of features[k+1] in squeezedet.py:43-61) in ONE launch (inference): y[..., y_coff : +Nsq] =
 * ReLU(Wsq . cat(ReLU(conv1x1(x) + b1), ReLU(conv3x3(x) + b3)) + bsq); the concatenated expand output never reaches HBM.
 * u_packed from sqd_pack_wino_fire.  bias_tab [Npad_total/32][8][16] floats: per 32-wide slice of the packed axis the biases of
 * its 16-channel blocks (expand3x3 slice s: blocks 0, 1 = b3[32 s + 16 j + n]; expand1x1 slice s: 8 blocks = b1[128 s + 16 blk + n];
 * 0 where padded).  sq_ops [blocks][4][ceil(Nsq/16)][64] floats, blocks in the same order (2 per expand3x3 slice, then 8 per
 * expand1x1 slice): value at (block, t, q, lane = 16 g + lr) = Wsq[16 q + lr][cat channel of (block, 4 g + t)], 0 where padded.
 * sq_bias [Nsq].  Nsq <= 32.  cfg_id 10: U resident in LDS (needs (Npad_total/32)*(C/8)*16 KB + 32 KB + operands <= 160 KB), 6: streamed;
 * + 1000*k = workgroups-per-CU cap.  cfg_id 12 (C <= 16; the transformed input of a group stays in registers, 16-wide passes):
 * bias_tab [passes][4][16], passes = 2*ceil(N3/32) expand3x3 passes (block 0 = b3[16 p + n]) then 2*ceil(N1/128) expand1x1 passes s1
 * (block r = b1[128 (s1 >> 1) + (2 r + (s1 & 1)) 16 + n]); sq_ops blocks in that order (1 per expand3x3 pass, 4 per expand1x1 pass). */
int sqd_fire_bridge_fwd(const float* x, const float* u_packed, const float* bias_tab, const float* sq_ops, const float* sq_bias,
                        float* y, int B, int H, int W, int C, int x_pitch, int x_coff, int N3, int N1, int Npad_total, int Nsq,
                        int y_pitch, int y_coff, int cfg_id, void* stream);

/* Training form of sqd_fire_bridge_fwd cfg 12 (C <= 16; the forward of src/engine/trainer.py:42 through src/model/squeezedet.py:18-22
 * twice): the same launch ALSO stores the concatenated expand output, save [B][H][W][save_pitch] with expand1x1 at save_coff1 and
 * expand3x3 at save_coff3 -- the tensor the backward reads (input of the next squeeze's weight gradient, ReLU masks).  The next
 * Fire's squeeze has no launch of its own and does not read the 128-channel tensor back. */
int sqd_fire_bridge_save_fwd(const float* x, const float* u_packed, const float* bias_tab, const float* sq_ops, const float* sq_bias,
                             float* y, float* save, int B, int H, int W, int C, int x_pitch, int x_coff, int N3, int N1,
                             int Npad_total, int Nsq, int y_pitch, int y_coff, int save_pitch, int save_coff3, int save_coff1,
                             void* stream);

/* In-place refresh of the Fire bridges' operands after an optimizer step (src/engine/trainer.py:50 changes every parameter), ONE launch
 * for all of them: descs_dev = n records of 7 int64 {dst, idx (int32 map), count, src0, src1, src2, src3}; dst[i] = 0 where idx[i] == -1,
 * untouched where -2, else scale * src[(idx >> 26) & 3][idx & 0x3ffffff] with scale 1 / +0.25 / -0.25 for (idx >> 28) & 3 = 0 / 1 / 2. */
int sqd_gather_pack_batched(const void* descs_dev, int n, int blocks_per_desc, void* stream);

/* Fire k's expand pair + torch.cat + MaxPool2d(3, 2, ceil_mode=True) + Fire k+1's squeeze (src/model/squeezedet.py:18-22 and the
 * features[...] sequence at 47-52: Fire, MaxPool2d, Fire) in ONE launch (inference): y [B][Hp][Wp] window [y_coff, +Nsq) =
 * ReLU(Wsq . maxpool(cat(ReLU(conv1x1(x) + b1), ReLU(conv3x3(x) + b3))) + bsq).  Neither the expand output nor the pooled tensor
 * is written.  C <= 16, N1 <= 64, N3 <= 64, Nsq <= 32.  u_packed / bias_tab as for sqd_fire_bridge_fwd cfg 12; sq_ops
 * [blocks][4][ceil(Nsq/16)][64] with one block per expand3x3 pass and TWO per expand1x1 pass s1 (channels 128 (s1 >> 1) +
 * (2 r + (s1 & 1)) 16 + n, r = 0, 1).  Hp x Wp = the pool's output size.  nseg >= 1: vertical segments per column strip. */
int sqd_fire_pool_bridge_fwd(const float* x, const float* u_packed, const float* bias_tab, const float* sq_ops, const float* sq_bias,
                             float* y, int B, int H, int W, int C, int x_pitch, int x_coff, int N3, int N1, int Npad_total, int Nsq,
                             int Hp, int Wp, int y_pitch, int y_coff, int nseg, void* stream);

/* Training form of sqd_fire_pool_bridge_fwd (the forward of src/engine/trainer.py:42 through src/model/squeezedet.py:18-22, 47-52): the
 * same launch ALSO stores the POOLED expand output -- save [B][Hp][Wp][save_pitch], expand1x1 at save_coff1, expand3x3 at save_coff3 --
 * and the pool's arg-max / ReLU codes (codes: one byte per element of save's geometry; first window position 3 dy + dx holding the
 * pooled value, 15 where it is not > 0: the codes of sqd_maxpool3x3s2_ceil_fwd_relu).  That is all the backward reads of this stage:
 * the unpooled expand output is never written, the max pool and the next squeeze have no launches of their own. */
int sqd_fire_pool_bridge_save_fwd(const float* x, const float* u_packed, const float* bias_tab, const float* sq_ops, const float* sq_bias,
                                  float* y, float* save, unsigned char* codes, int B, int H, int W, int C, int x_pitch, int x_coff,
                                  int N3, int N1, int Npad_total, int Nsq, int Hp, int Wp, int y_pitch, int y_coff, int save_pitch,
                                  int save_coff3, int save_coff1, int nseg, void* stream);

/* clip_grad_norm_ + torch.optim.SGD(momentum, weight_decay).step() (src/engine/trainer.py:47-50) for every parameter tensor in ONE
 * launch.  descs_dev: n records of 4 int64 {param ptr, grad, momentum-buffer ptr, elements}, grad = element offset into grad_base (the
 * backward's flat gradient buffer) or the gradient's address when grad_base is NULL; total_norm: device float, the L2
 * norm of all gradients (read on the device: no host sync; NULL allowed when max_norm <= 0 = no clipping).  Per element, in torch's
 * order: g = grad * min(1, max_norm / (total_norm + 1e-6)); g += wd * p; buf = momentum * buf + g; p -= lr * buf. */
int sqd_sgd_clip_step(const void* descs_dev, int n, const float* grad_base, const float* total_norm, float max_norm, float lr,
                      float momentum, float weight_decay, int blocks_per_desc, void* stream);
/* The gradient norm of clip_grad_norm_ (src/engine/trainer.py:49) without a torch reduction kernel: sqd_grad_sumsq writes
 * sqd_grad_sumsq_parts() partial sums of squares of the flat gradient (fixed tree: bitwise reproducible); sqd_sgd_clip_step_parts is
 * sqd_sgd_clip_step with total_norm = sqrt(sum of the partials in index order), every workgroup evaluating it for itself; norm_out
 * (DEVICE float or NULL) receives the norm. */
int sqd_grad_sumsq(const float* grad_flat, long long n, float* parts, void* stream);
int sqd_grad_sumsq_parts(void);
int sqd_sgd_clip_step_parts(const void* descs_dev, int n, const float* grad_base, const float* sumsq_parts, float* norm_out,
                            float max_norm, float lr, float momentum, float weight_decay, int blocks_per_desc, void* stream);
/* The same step with the elements dealt in equal chunks (one workgroup per sqd_sgd_chunk_elems() elements of one tensor) instead of a
 * fixed number of workgroups per tensor: chunks_dev = [nchunks][2] int64 {tensor index, first element}. */
int sqd_sgd_chunk_elems(void);
int sqd_sgd_clip_step_chunked(const void* descs_dev, const void* chunks_dev, int nchunks, const float* grad_base,
                              const float* sumsq_parts, float* norm_out, float max_norm, float lr, float momentum, float weight_decay,
                              void* stream);


#ifdef __cplusplus
}
#endif
#endif /* SQD_HIP_H */
