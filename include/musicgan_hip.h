/*
 * musicgan_hip.h -- C ABI of libmusicgan_hip.so, the MI355X (gfx950) kernels behind the MusicGAN hot path.
 *
 * The reference (Ipsedo/MusicGAN) has no FFI/plugin layer: its hot path is stock torch.nn modules
 * (music_gan/networks/) and torchaudio/torch.stft calls (music_gan/audio/functions.py).  Each entry point below
 * names the reference module/ATen op it replaces (file:line relative to /root/reference/music_gan).  A maintainer binds
 * these with ctypes; musicgan_amd/_lib.py is that binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32, NCHW contiguous, 16-byte aligned (torch allocations are);
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); every call is asynchronous on it;
 *   - functions return 0 on success, a negative MG_E* code otherwise, never throw; mg_last_error() gives the text of the
 *     calling thread's last failure; no hidden global state apart from one-time kernel attribute set-up;
 *   - callable concurrently from several host threads on distinct streams (autograd backward threads do).
 */
#ifndef MUSICGAN_HIP_H
#define MUSICGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mg_stream_t;

#define MG_OK 0
#define MG_EINVAL (-1)  /* bad shape / flag combination */
#define MG_ELAUNCH (-2) /* hip launch error */
#define MG_EWORKSPACE (-3)
#define MG_EIO (-4)      /* a host file operation failed (mg_pt_write_samples): errno text in mg_last_error() */

int mg_version(void);
const char* mg_last_error(void);

/* ------------------------------------------------------------------ 3x3 convolution, stride 1, pad 1
 * Replaces nn.Conv2d(k=3,s=1,p=1) [generator.py:16-22,31-37; discriminator.py:15-21,26-32] fused with what follows it:
 * LeakyReLU(0.2) [generator.py:23,38; discriminator.py:22,33], PixelNorm [layers.py:11-17], and with what precedes it
 * in the generator, nn.Upsample(x2, nearest) [generator.py:26-29].  The same kernel evaluates the data gradient
 * (aten::convolution_backward, input grad) when given weights packed with dgrad=1.
 */
#define MG_CONV_UPS_IN 1   /* logical input = nearest-upsample x2 of x (x is N,Cin,H/2,W/2) */
#define MG_CONV_LRELU 2    /* y = leaky_relu(acc + bias, slope) */
#define MG_CONV_MASK_AUX 4 /* y = acc * (aux > 0 ? 1 : slope): LeakyReLU backward fused on the output (aux: N,Cout,H,W) */
#define MG_CONV_PIXNORM 8  /* also emit p = y * rn and rn = 1/sqrt(mean_c(y^2)+1e-8) (needs MG_CONV_LRELU) */
#define MG_CONV_POOL_OUT 16 /* also emit p = AvgPool2d(2,2)(y) (N,Cout,H/2,W/2) [discriminator.py:24]; excludes PIXNORM */
/* Tile masks (mg_wino3x3 only): what the backward pass of conv -> LeakyReLU -> AvgPool2d [discriminator.py:15-24] needs of the
 * full-resolution activation is its sign, so the forward pass can keep one BYTE per 2x2 tile and out-channel instead of four
 * floats: bit 2i+j set <=> y[2Y+i][2X+j] > 0, tensor (N,Cout,H/2,W/2) of uint8. */
#define MG_CONV_MASK_OUT 32   /* with POOL_OUT|LRELU: y (cast to uint8_t*) receives the tile mask; the fp32 y is not written */
#define MG_CONV_MASK_BYTES 64 /* with MASK_AUX|POOL_OUT: aux (cast to const uint8_t*) is the tile mask of THIS conv's output; only p is written */
#define MG_CONV_UNPOOL 128    /* alone: aux = tile mask with one byte per OUTPUT PIXEL (N,Cout,H,W) of this conv; y is (N,Cout,2H,2W): \
                                 y[2Y+i][2X+j] = 0.25 * acc[Y][X] * (bit 2i+j ? 1 : slope), i.e. AvgPool2d backward and the LeakyReLU \
                                 backward of the layer below fused on the data-gradient conv that feeds them */

/* number of floats of the packed (LDS-image) weight layout for a Cin->Cout conv */
size_t mg_conv3x3_packed_floats(int Cin, int Cout);
/* w is the module weight [Co][Ci][3][3].  dgrad=0: pack for the forward conv Ci->Co.  dgrad=1: pack for the data-gradient
 * conv Co->Ci (taps flipped, channels transposed). */
int mg_conv3x3_pack(const float* w, float* wp, int Co, int Ci, int dgrad, mg_stream_t stream);
/* y[N,Cout,H,W] = epilogue(conv3x3(x, wp) + bias).  bias/aux/p/rn may be NULL when the flag that uses them is unset. */
int mg_conv3x3(const float* x, const float* wp, const float* bias, const float* aux, float* y, float* p, float* rn,
               int N, int Cin, int Cout, int H, int W, int flags, float slope, mg_stream_t stream);

/* Upsample(x2 nearest) -> Conv2d(3x3) [generator.py:26-37] in sub-pixel form: four 2x2 convolutions on the LOW-resolution
 * input (16 instead of 36 multiply-adds per low-res pixel and channel pair; same result up to fp32 summation order).
 * x: (N,Cin,Hin,Win); y/p: (N,Cout,2Hin,2Win); rn: (N,1,2Hin,2Win).  flags: MG_CONV_LRELU, MG_CONV_PIXNORM.  wp from
 * mg_upconv3x3_pack (effective weights, mg_upconv3x3_packed_floats floats).  Cout <= 96 recommended (register budget). */
/* The same layer -- Upsample(x2 nearest) -> Conv2d(3x3) [generator.py:24-39] (+ LeakyReLU + PixelNorm, layers.py:11-17) -- and its
 * data gradient (aten::convolution_backward input + upsample_nearest2d_backward) in Winograd F(2x2,3x3) form on the up-sampled grid
 * without the 7 of 16 components that vanish there: 9 multiply-adds per output tile and channel pair (sub-pixel form: 16, direct on
 * the up-sampled tensor: 36).  up: MG_PACK_WINOUPS filters of the module weight [Cout][Cin][3][3] (mg_pack_multi, dgrad = 0 / 1;
 * mg_winoups3x3_packed_floats floats).  Forward: x (N,Cin,Hin,Win) -> y / p (N,Cout,2Hin,2Win), rn (N,1,2Hin,2Win); flags
 * MG_CONV_LRELU, MG_CONV_PIXNORM (then p is written, y optional).  Data gradient: gy (N,Cout,2Hin,2Win) -> gx (N,Cin,Hin,Win).
 * mg_winoups3x3_supported: Win a multiple of 16, Hin of 8, 16..64 out-channels of the call in whole tiles, the 9-component filter
 * bank within the LDS. */
int mg_winoups3x3_supported(int N, int Cin, int Cout, int Hin, int Win, int dgrad);
size_t mg_winoups3x3_packed_floats(int Cin, int Cout, int dgrad);
int mg_winoups3x3(const float* x, const float* up, const float* bias, float* y, float* p, float* rn, int N, int Cin, int Cout,
                  int Hin, int Win, int flags, float slope, mg_stream_t stream);
int mg_winoups3x3_dgrad(const float* gy, const float* up, float* gx, int N, int Cin, int Cout, int Hin, int Win, mg_stream_t stream);
/* mg_winoups3x3_dgrad with the PixelNorm + LeakyReLU backward of the layer below [generator.py:31-39, layers.py:11-17] in the epilogue:
 * p (N,Cin,Hin,Win) that layer's normalised output, rn (N,1,Hin,Win) its 1/norm; gpre = lrelu'(p) rn (gx - p mean_c(gx p)). */
int mg_winoups3x3_dgrad_pn(const float* gy, const float* up, const float* p, const float* rn, float* gpre, int N, int Cin, int Cout, int Hin,
                           int Win, float slope, mg_stream_t stream);
/* mg_winoups3x3 (LeakyReLU + PixelNorm) with the generator's 1x1 head [generator.py:118-126 ToMagnPhaseLayer] on the normalised
 * activation in the same epilogue: mp = tanh(hw p + hb), hw (2,Cout), hb (2) or NULL, mp (N,2,2Hin,2Win); p, rn as above, y optional. */
int mg_winoups3x3_head_supported(int N, int Cin, int Cout, int Hin, int Win); /* mg_winoups3x3_supported and at most 48 out-channels */
int mg_winoups3x3_head(const float* x, const float* up, const float* bias, float* y, float* p, float* rn, const float* hw, const float* hb,
                       float* mp, int N, int Cin, int Cout, int Hin, int Win, float slope, mg_stream_t stream);
size_t mg_upconv3x3_packed_floats(int Cin, int Cout);
int mg_upconv3x3_pack(const float* w, float* wp, int Co, int Ci, mg_stream_t stream);
int mg_upconv3x3(const float* x, const float* wp, const float* bias, float* y, float* p, float* rn, int N, int Cin, int Cout,
                 int Hin, int Win, int flags, float slope, mg_stream_t stream);

/* The same convolution (no UPS_IN; even H, W) in Winograd F(2x2,3x3) form: 16 instead of 36 multiplies per 2x2 output tile
 * and channel pair, fp32 throughout (rounding within ~1.2x rms of the direct form).  Same arguments, flags and fused epilogues
 * as mg_conv3x3; PIXNORM needs Cout <= 64.  up from mg_wino3x3_pack (dgrad = 1: filters of the data-gradient convolution). */
size_t mg_wino3x3_packed_floats(int Cin, int Cout);
int mg_wino3x3_pack(const float* w, float* up, int Co, int Ci, int dgrad, mg_stream_t stream);
/* whether MG_CONV_MASK_AUX | MG_CONV_MASK_BYTES WITHOUT MG_CONV_POOL_OUT (y = result x lrelu'(tile-mask bytes), full resolution) is
 * available for this shape: an epilogue of the wave-per-tile-block kernel (csrc/wino_strip.hip) only */
int mg_wino3x3_mask_bytes_y_supported(int N, int Cin, int Cout, int H, int W);
int mg_wino3x3(const float* x, const float* up, const float* bias, const float* aux, float* y, float* p, float* rn, int N,
               int Cin, int Cout, int H, int W, int flags, float slope, mg_stream_t stream);

/* The critic's fade-in blend [discriminator.py:111-116: alpha * conv_blocks[curr](start_block(x)) + (1 - alpha) *
 * last_start_block(x)] fused on the Winograd convolution next to it, so that neither the blend nor its backward is a pass of its
 * own.  coef: {alpha, 1 - alpha} in DEVICE memory (a captured graph stays valid while alpha moves); tile masks as above.
 *   MG_FADE_FWD      y = coef[0] * leaky_relu(conv(x) + bias) + coef[1] * other;   out2 (uint8, N,Cout,H/2,W/2) = tile mask of
 *                    the LeakyReLU output (all the backward passes need of it)
 *   MG_FADE_TANGENT  y = coef[0] * (conv(x) * lrelu'(mask_in)) + coef[1] * other   (tangent pass of the gradient penalty)
 *   MG_FADE_BWD      up = data-gradient filters, x = gradient w.r.t. the conv AFTER the blend:
 *                    y = (coef[0] * conv(x)) * lrelu'(mask_in),  out2 (float, N,Cout,H,W) = (coef[1] * conv(x)) * lrelu'(other > 0)
 * Results are bitwise those of mg_wino3x3 followed by mg_axpby / mg_blend_lrelu_bwd. */
#define MG_FADE_FWD 1
#define MG_FADE_TANGENT 2
#define MG_FADE_BWD 3
int mg_wino3x3_fade(const float* x, const float* up, const float* bias, const unsigned char* mask_in, const float* other,
                    const float* coef, float* y, void* out2, int N, int Cin, int Cout, int H, int W, int mode, float slope,
                    mg_stream_t stream);

/* Weight (+ bias) gradient of the same convolution in Winograd F(3x3,2x2) form (even H, W; flags: MG_CONV_UPS_IN): same result as
 * mg_conv3x3_wgrad within fp32 rounding, 2.25x fewer multiplies, split-K slabs reduced in a fixed order (deterministic).
 * gw[Cout][Cin][3][3] (+)= ..., gb[Cout] (+)= sum of gy over samples n < bias_n (0: all; gb may be NULL). */
/* Which kernel mg_wino3x3_wgrad_partial(_multi) runs for a layer (accounting of executed FLOPs, tests): 0 = chunk-staged forms
 * (16 Winograd products per tile), 1 = row-staged form, 2 = row-staged form for an up-sampled input (MG_CONV_UPS_IN: 9 of the 16
 * products), -1 = shape not supported.  group_max_chunks as passed to mg_wino3x3_wgrad_partial_multi (0: single launches). */
int mg_wino3x3_wgrad_form(int N, int Cin, int Cout, int H, int W, int flags, int group_max_chunks);
size_t mg_wino3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W);
int mg_wino3x3_wgrad(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int N, int Cin, int Cout,
                     int H, int W, int flags, int accumulate, int bias_n, mg_stream_t stream);
/* The same in two halves, so that the slab reductions of several layers run in ONE launch at the end of an update's weight-gradient
 * sweep (each is a ~20 us latency-bound kernel): _partial runs the matrix kernel into `ws` (one workspace PER LAYER, alive until
 * the reduce) and fills `job`; _reduce sums the slabs of n jobs in a fixed order, applies G^T . G and writes gw / gb. */
typedef struct {
  const float* slab;
  const float* slab_b;
  float* gw;
  float* gb;
  int32_t nsplit, Cout, Cin, CoutP, CinP, accumulate;
} mg_wgrad_job_t;
int mg_wino3x3_wgrad_partial(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int N, int Cin,
                             int Cout, int H, int W, int flags, int accumulate, int bias_n, mg_wgrad_job_t* job,
                             mg_stream_t stream);
int mg_wino3x3_wgrad_reduce(const mg_wgrad_job_t* jobs, int n, mg_stream_t stream);
/* _partial for the n layers of a sweep at once (the weight gradients of one backward pass: convolution_backward of every
 * nn.Conv2d(3x3) in generator.py:9-40 / discriminator.py:8-34).  Layers of at most group_max_chunks x (number of CUs) units of work
 * (8-tile chunks x channel blocks) whose channel blocks have the same shape share ONE launch, with their splits sized so that the
 * group -- not every layer -- fills the chip: fewer launches, fewer and smaller slabs for the reduce.  group_max_chunks <= 0:
 * one launch per layer, exactly as n calls of _partial.  jobs[i] belongs to d[i]; results are bitwise independent of the grouping
 * only up to the split count (the slabs are summed in a fixed order either way: deterministic for a given grouping). */
typedef struct {
  const float* x;
  const float* gy;
  float* gw;
  float* gb;
  void* ws;
  size_t ws_bytes;
  int32_t N, Cin, Cout, H, W, flags, accumulate, bias_n;
} mg_wgrad_desc_t;
int mg_wino3x3_wgrad_partial_multi(const mg_wgrad_desc_t* d, int n, int group_max_chunks, mg_wgrad_job_t* jobs, mg_stream_t stream);
/* the same split of mg_conv3x3_wgrad (the direct form's jobs carry CoutP = CinP = 0 and go to their own reduce) */
int mg_conv3x3_wgrad_partial(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int N, int Cin,
                             int Cout, int H, int W, int flags, int accumulate, int bias_n, mg_wgrad_job_t* job,
                             mg_stream_t stream);
int mg_conv3x3_wgrad_reduce(const mg_wgrad_job_t* jobs, int n, mg_stream_t stream);

/* Data gradient of Upsample(x2) -> Conv3x3 w.r.t. the LOW-resolution input: one stride-2 convolution with the 4x4 effective
 * kernel over gy (N,Cout,2Hin,2Win) -> gx (N,Cin,Hin,Win); replaces conv-dgrad at 2Hx2W + the 2x2 block sums of Upsample's
 * backward.  wp from mg_upconv3x3_dgrad_pack(w [Co][Ci][3][3]). */
size_t mg_upconv3x3_dgrad_packed_floats(int Cin, int Cout);
int mg_upconv3x3_dgrad_pack(const float* w, float* wp, int Co, int Ci, mg_stream_t stream);
int mg_upconv3x3_dgrad(const float* gy, const float* wp, float* gx, int N, int Cin, int Cout, int Hin, int Win,
                       mg_stream_t stream);

/* weight/bias gradient of the same conv (aten::convolution_backward, weight+bias grads):
 *   gw[Cout][Cin][3][3] (+)= sum_{n,y,x} gy[n,o,y,x] * xin[n,c,y+ky-1,x+kx-1],  gb[Cout] (+)= sum gy   (gb may be NULL)
 * flags: MG_CONV_UPS_IN as above; accumulate!=0 adds to gw/gb instead of overwriting.  bias_n: only samples n < bias_n feed
 * gb (<= 0: all N) -- lets one launch over a concatenated batch sum weight gradients of all samples but bias gradients of a
 * leading sub-batch (the gradient-penalty part has no bias gradient).  ws: scratch of mg_conv3x3_wgrad_ws_bytes() bytes
 * (split-K partial slabs, reduced in a fixed order => deterministic). */
size_t mg_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W);
int mg_conv3x3_wgrad(const float* x, const float* gy, float* gw, float* gb, void* ws, size_t ws_bytes, int N, int Cin,
                     int Cout, int H, int W, int flags, int accumulate, int bias_n, mg_stream_t stream);

/* the same on 1x1 maps (the last critic conv after the final AvgPool2d, discriminator.py:14-34): x (N,Cin,1,1), gy (N,Cout,1,1);
 * only the centre tap is non-zero.  No workspace, no reduce; float64 accumulation in sample order (deterministic). */
int mg_conv3x3_wgrad_1x1map(const float* x, const float* gy, float* gw, float* gb, int N, int Cin, int Cout, int accumulate,
                            int bias_n, mg_stream_t stream);

/* ------------------------------------------------------------------ 1x1 convolutions (stem 2->C, head C->2)
 * Replaces MagPhaseLayer [discriminator.py:37-50] and ToMagnPhaseLayer [generator.py:43-52].  One of Cin/Cout must be <= 4.
 */
#define MG_C1_LRELU 1      /* y = leaky_relu(.) */
#define MG_C1_TANH 2       /* y = tanh(.) */
#define MG_C1_MASK_AUX 4   /* Cin<=4: y = acc * (aux > 0 ? 1 : slope), aux is N,Cout,HW (output side);
                            * Cout<=4 (Cin>4): x is first multiplied by (aux > 0 ? 1 : slope), aux is N,Cin,HW (input side) */
#define MG_C1_TRANSPOSED 8 /* use w^T: w is [Cin][Cout] in memory (data gradient of the forward conv) */
#define MG_C1_TANH_BWD_IN 16 /* input is gy*(1-aux_in^2): tanh backward fused on the INPUT side (aux_in: N,Cin,HW) */
#define MG_C1_ACCUM 32       /* Cin<=4: y += result (two gradient branches joining: generator.py:118-124 backwards) */
int mg_conv1x1(const float* x, const float* w, const float* bias, const float* aux, float* y, int N, int Cin, int Cout,
               int HW, int flags, float slope, mg_stream_t stream);
/* gw[Cout][Cin] (+)= sum gy*x, gb (+)= sum gy; if tanh_y != NULL gy is first multiplied by (1 - tanh_y^2).
 * bias_n (as for the 3x3 weight gradients): only samples n < bias_n feed gb (0: all) -- the penalty's tangent samples of the
 * fused critic step have no bias gradient. */
size_t mg_conv1x1_wgrad_ws_bytes(int N, int Cin, int Cout, int HW);
int mg_conv1x1_wgrad(const float* x, const float* gy, const float* tanh_y, float* gw, float* gb, void* ws,
                     size_t ws_bytes, int N, int Cin, int Cout, int HW, int accumulate, int bias_n, mg_stream_t stream);

/* The two-channel ends of both networks while a block fades in, one launch each (csrc/fade_ends.hip).
 * mg_stem_pair [discriminator.py:107-113 forward]: h0 = act(ws x + bs) (N,C0,H,W); xp = AvgPool2d(x) (N,2,H/2,W/2; may be NULL);
 *   o = act(wo xp + bo) (N,C1,H/2,W/2).  flags: MG_C1_LRELU, or MG_C1_MASK_AUX = bias-free results times the LeakyReLU derivative of
 *   the activations h0 / o hold, written over them (the penalty's tangent pass).  ws (C0,2), wo (C1,2).  H even, W % 4 == 0.
 *   h0_mask (optional, forward form): the tile mask of h0 in mg_wino3x3's format (N,C0,H/2,W/2 uint8, bit 2i+j <-> h0[2Y+i][2X+j] > 0),
 *   what the data-gradient conv in front of the stem reads instead of h0 itself (MG_CONV_MASK_AUX | MG_CONV_MASK_BYTES).
 * mg_stem_pair_gx [the same lines, backward to x]: gx = ws^T gs + 0.25 * up2(wo^T go), gs (N,C0,H,W), go (N,C1,H/2,W/2), gx (N,2,H,W).
 * mg_head_pair [generator.py:118-126]: mp = tanh(wh x + bh) (N,2,H,W), old = tanh(wo xl + bo) (N,2,H/2,W/2), out = a mp + b up2(old);
 *   (a, b) = coef[0..1] from device memory if coef != NULL, else (ca, cb); mp / old may be NULL (not kept).  wh (2,C), wo (2,Cl).
 * mg_blend_up_bwd: gx = a g, gy = b * (2x2 block sums of g): the blend's backward (g: (NC,H,W), gy: (NC,H/2,W/2)).
 * mg_gen_head_bwd [generator.py:118-126 backward + layers.py:11-17 / generator.py:31-39 backward of the block in front]: from the
 *   gradient g_mp (N,2,H,W) at the head's tanh output mp: t = g_mp (1 - mp^2); gw (2,C) (+)= sum t p, gb (2) (+)= sum t (autograd's
 *   convolution_backward of the 1x1 head, x = p (N,C,H,W) the last block's PixelNorm output); g = w^T t; gpre = lrelu'(p) rn
 *   (g - p mean_c(g p)) (N,C,H,W) = the gradient at the last conv's pre-activation (PixelNorm + LeakyReLU backward, rn (N,1,H,W) the
 *   stored 1/norm).  One read of p, one write of gpre.  g_in (optional, (N,C,H,W)): a second gradient arriving at p, added to g (the old
 *   head of a fading-in level: p is also the input of the last block's first conv, g_in that conv's data gradient).
 *   C in {16, 32, 48, 64}, or any multiple of 4 up to 128 on small maps (N H W <= 32 768); ws: mg_gen_head_bwd_ws_floats(N, C, HW) floats. */
int mg_stem_pair(const float* x, const float* ws, const float* bs, const float* wo, const float* bo, float* h0, float* xp, float* o,
                 unsigned char* h0_mask, int N, int C0, int C1, int H, int W, int flags, float slope, mg_stream_t stream);
int mg_stem_pair_gx(const float* gs, const float* ws, const float* go, const float* wo, float* gx, int N, int C0, int C1, int H, int W,
                    mg_stream_t stream);
int mg_head_pair(const float* x, const float* wh, const float* bh, const float* xl, const float* wo, const float* bo, const float* coef,
                 float ca, float cb, float* mp, float* old, float* out, int N, int C, int Cl, int H, int W, mg_stream_t stream);
int mg_blend_up_bwd(const float* g, const float* coef, float ca, float cb, float* gx, float* gy, int NC, int H, int W,
                    mg_stream_t stream);
/* mg_head_pair with mp given (written by mg_winoups3x3_head): old = tanh(wo xl + bo), out = a mp + b up2(old). */
int mg_head_pair_from_mp(const float* mp, const float* xl, const float* wo, const float* bo, const float* coef, float ca, float cb,
                         float* old, float* out, int N, int Cl, int H, int W, mg_stream_t stream);
int mg_gen_head_bwd_supported(int C, int Cout);                      /* C in {16, 32, 48, 64}: any size */
int mg_gen_head_bwd_supported_at(int C, int Cout, int N, int HW);     /* ... or a multiple of 4 up to 128 on at most 32 768 pixels */
size_t mg_gen_head_bwd_ws_floats(int N, int C, int HW);
int mg_gen_head_bwd(const float* g_mp, const float* mp, const float* w, const float* p, const float* rn, const float* g_in, float* gpre,
                    float* gw, float* gb, float* ws, size_t ws_floats, int N, int C, int HW, float slope, int accumulate, mg_stream_t stream);

/* ------------------------------------------------------------------ element-wise / small ops */
/* PixelNorm forward [layers.py:11-17]: p = y*rn, rn[n,hw] = 1/sqrt(mean_c y^2 + 1e-8) */
int mg_pixelnorm_fwd(const float* y, float* p, float* rn, int N, int C, int HW, mg_stream_t stream);
/* backward through PixelNorm then LeakyReLU: gpre = mask(y) * rn * (gp - p * mean_c(gp * p)), p = y*rn.
 * from_p != 0: the `y` argument is the normalised output p itself (the pre-norm activation need not be kept). */
int mg_pixelnorm_lrelu_bwd(const float* gp, const float* y, const float* rn, float* gpre, int N, int C, int HW,
                           float slope, int from_p, mg_stream_t stream);
/* nn.Upsample(x2 nearest) forward / backward (sum of each 2x2 block) [generator.py:26-29,99-102] */
int mg_upsample2x_fwd(const float* x, float* y, int NC, int Hin, int Win, mg_stream_t stream);
int mg_upsample2x_bwd(const float* gy, float* gx, int NC, int Hin, int Win, mg_stream_t stream);
/* nn.AvgPool2d(2,2) forward [discriminator.py:24,131]; backward fused with the LeakyReLU mask of the layer below:
 * gx = 0.25 * gy[.., h/2, w/2] * (act > 0 ? 1 : slope)   (act NULL => no mask) */
int mg_avgpool2_fwd(const float* x, float* y, int NC, int H, int W, mg_stream_t stream);
int mg_avgpool2_bwd(const float* gy, const float* act, float* gx, int NC, int H, int W, float slope,
                    mg_stream_t stream);
/* the same with the mask given as tile bytes (MG_CONV_MASK_OUT above: (NC,H/2,W/4*2) uint8, bit 2i+j <-> act[2h+i][2w+j] > 0);
 * W must be a multiple of 4 */
int mg_avgpool2_bwd_tilemask(const float* gy, const unsigned char* mask, float* gx, int NC, int H, int W, float slope,
                             mg_stream_t stream);
/* out = g * (act > 0 ? 1 : slope) */
int mg_lrelu_bwd(const float* g, const float* act, float* out, size_t n, float slope, mg_stream_t stream);
/* backward of the critic's fade-in blend [discriminator.py:111-113] and of the two LeakyReLUs feeding it, in one pass:
 * out_a = ca*g*(act_a > 0 ? 1 : slope), out_o = co*g*(act_o > 0 ? 1 : slope)  (rounded as (c*g)*mask, like axpby + lrelu_bwd) */
int mg_blend_lrelu_bwd(const float* g, const float* act_a, const float* act_o, float ca, float co, float* out_a, float* out_o,
                       size_t n, float slope, mg_stream_t stream);
/* out = a*x + b*y (fade-in blend [generator.py:124, discriminator.py:113]); y may be NULL (out = a*x) */
int mg_axpby(float a, const float* x, float b, const float* y, float* out, size_t n, mg_stream_t stream);
/* out[n,c,h,w] = a*x[n,c,h,w] + b*up2(y)[n,c,h,w], y is (NC, H/2, W/2) */
int mg_blend_up(float a, const float* x, float b, const float* y, float* out, int NC, int H, int W,
                mg_stream_t stream);
/* The three fade-in kernels with their coefficients (a, b) = coef[0], coef[1] read from DEVICE memory (y == NULL: coef[0] only):
 * alpha changes every iteration of a fade-in (utils.py:62-68); as launch arguments it would be baked into a captured HIP graph
 * of the update.  Same values, same results as the scalar forms. */
int mg_axpby_dev(const float* coef, const float* x, const float* y, float* out, size_t n, mg_stream_t stream);
int mg_blend_up_dev(const float* coef, const float* x, const float* y, float* out, int NC, int H, int W, mg_stream_t stream);
int mg_blend_lrelu_bwd_dev(const float* g, const float* act_a, const float* act_o, const float* coef, float* out_a, float* out_o,
                           size_t n, float slope, mg_stream_t stream);
/* Linear(K -> 1) [discriminator.py:103-105]: y[n] = b + sum_k w[k] x[n,k] */
int mg_linear1_fwd(const float* x, const float* w, const float* b, float* y, int N, int K, mg_stream_t stream);
/* gx[n,k] = gy[n]*w[k]; gw[k] (+)= sum_n gy[n] x[n,k]; gb (+)= sum_{n < bias_n} gy[n] (0: all n)   (gx/gw/gb may be NULL) */
int mg_linear1_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int N, int K,
                   int accumulate, int bias_n, mg_stream_t stream);
/* gradient-penalty helpers [discriminator.py:166-184] */
int mg_gp_interp(const float* x_real, const float* x_fake, const float* eps, float* out, int N, size_t chw,
                 mg_stream_t stream); /* out = eps[n]*real + (1-eps[n])*fake */
int mg_sumsq_per_sample(const float* g, float* out, int N, size_t chw, mg_stream_t stream);
int mg_scale_per_sample(const float* g, const float* coef, float* out, int N, size_t chw, mg_stream_t stream);
/* tiny: from sumsq[N] compute penalty = factor*mean((sqrt(ss)-1)^2) and coef[n] = upstream*factor*2*(norm-1)/(N*norm) */
int mg_gp_finish(const float* sumsq, float* penalty, float* coef, int N, float factor, float upstream,
                 mg_stream_t stream);
/* mg_gp_finish + mg_scale_per_sample in one launch: out = g * coef[n] with coef as above, penalty (may be NULL) as above */
int mg_gp_apply(const float* g, const float* sumsq, float* penalty, float* out, int N, size_t chw, float factor, float upstream,
                mg_stream_t stream);
/* sum over (n, hw) of a per-channel tensor: out[c] (+)= sum x[n,c,hw] */
int mg_channel_sum(const float* x, float* out, int N, int C, int HW, int accumulate, mg_stream_t stream);

/* Wasserstein losses [criterion.py:12-18] and the score means train.py:180-186 logs, one launch: out[g] = mean of the g-th group of
 * n consecutive critic scores (g < groups <= 8), out[groups] = groups >= 2 ? out[1] - out[0] (= wasserstein_discriminator_loss
 * with group 0 = real, 1 = fake) : -out[0] (= wasserstein_generator_loss).  out has groups + 1 floats. */
int mg_group_means(const float* x, int groups, int n, float* out, mg_stream_t stream);

/* ------------------------------------------------------------------ multi-tensor weight re-packing
 * All of mg_conv3x3_pack / mg_wino3x3_pack / mg_upconv3x3_pack / mg_upconv3x3_dgrad_pack for a list of weights in ONE launch (the
 * reference has no counterpart: these layouts replace what MIOpen / oneDNN re-derive from nn.Conv2d.weight inside every call).
 * descs is a HOST array; `out` buffers are sized by the matching *_packed_floats(). */
enum { MG_PACK_CONV3X3 = 0, MG_PACK_WINO3X3 = 1, MG_PACK_UPCONV3X3 = 2, MG_PACK_UPCONV3X3_DGRAD = 3, MG_PACK_SMALLNET = 4, MG_PACK_WINOUPS = 5 };
typedef struct {
  const float* w; /* module weight [Co][Ci][3][3] */
  float* out;
  int32_t kind, Co, Ci, dgrad; /* dgrad: data-gradient variant (kinds 0, 1, 4 and 5 only) */
} mg_pack_desc_t;
int mg_pack_multi(const mg_pack_desc_t* descs, int n, mg_stream_t stream);

/* ------------------------------------------------------------------ fused Adam [train.py:64-70,175,214]
 * Multi-tensor torch.optim.Adam step (amsgrad off, weight_decay 0).  desc is a HOST array of n_tensors records (device pointers
 * inside); the records are passed to the kernel by value, so the call neither copies nor synchronises. */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
  float step_size;  /* lr / (1 - beta1^step), step = count AFTER this update (>= 1) */
  float bc2_sqrt;   /* sqrt(1 - beta2^step) */
} mg_adam_tensor_t;
int mg_adam_step(const mg_adam_tensor_t* desc, int n_tensors, float beta1, float beta2, float eps, float grad_scale,
                 mg_stream_t stream);
/* Same update with the per-parameter step count in DEVICE memory (int32, count BEFORE this update; advanced by the call): the
 * bias corrections are formed on the device, so the launch carries nothing that changes from step to step and can be captured
 * in a HIP graph and replayed (torch.optim.Adam(capturable=True) semantics). */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
  int32_t* step;
} mg_adam_tensor_dev_t;
int mg_adam_step_dev(const mg_adam_tensor_dev_t* desc, int n_tensors, float lr, float beta1, float beta2, float eps,
                     float grad_scale, mg_stream_t stream);

/* ------------------------------------------------------------------ STFT [audio/functions.py:38-62]
 * wav: mono fp32 [L]; out_re/out_im: [512][T] (freq-major, Nyquist row dropped), T = 1 + L/256.  Periodic Hann(1024),
 * centre reflect padding, hop 256, divided by sqrt(sum w^2).  If out_im == NULL, out_re is interleaved complex64 [512][T][2]. */
int mg_stft_1024(const float* wav, float* out_re, float* out_im, int64_t L, mg_stream_t stream);
/* The same transform straight from a file's PCM frames [functions.py:43-49: th_audio.load (normalised to [-1, 1]) and
 * raw_audio.mean(0)]: pcm holds L frames of `channels` interleaved samples as a WAV file stores them (kind: MG_PCM_*), the
 * scaling (int16: / 32768, int32: / 2^31, uint8: (v - 128) / 128) and the mono mean happen inside the STFT kernel's loads for
 * float32 / int16 with one or two channels; other formats take one conversion pass through `ws` (mg_stft_1024_pcm_ws_bytes). */
enum { MG_PCM_F32 = 0, MG_PCM_I16 = 1, MG_PCM_I32 = 2, MG_PCM_U8 = 3 };
/* the conversion alone: mono[L] float32 = mean over channels of the normalised samples [functions.py:43-49] */
int mg_pcm_to_mono(const void* pcm, int kind, int channels, float* mono, int64_t L, mg_stream_t stream);
/* wav_to_stft's other arguments [functions.py:38-41: nperseg, stride]: the same definition for any power-of-two n_fft in
 * [64, 8192] and any hop (periodic Hann(n_fft), reflect padding n_fft/2, / sqrt(sum w^2), rows 0 .. n_fft/2 - 1);
 * out_c64: interleaved complex64 [n_fft/2][1 + L/hop].  An untuned radix-2 path: the drivers only use 1024 / 256. */
int mg_stft_generic(const float* wav, float* out_c64, int64_t L, int n_fft, int hop, mg_stream_t stream);
size_t mg_stft_1024_pcm_ws_bytes(int64_t L, int channels, int kind);
int mg_stft_1024_pcm(const void* pcm, int kind, int channels, float* out_re, float* out_im, void* ws, size_t ws_bytes, int64_t L,
                     mg_stream_t stream);

/* ------------------------------------------------------------------ multi-layer chains on small maps
 * The <= 4x4 ends of both networks -- the generator's first blocks [generator.py:15-40,67-76: conv3x3 -> LeakyReLU -> PixelNorm
 * -> Upsample -> conv3x3 -> LeakyReLU -> PixelNorm] and the critic's last blocks + classifier [discriminator.py:14-34,60-70,
 * 94-101: conv3x3 -> LeakyReLU -> AvgPool2d -> conv3x3 -> LeakyReLU ... -> Flatten -> Linear(160, 1)] -- are ~0.1 GFLOP of work
 * behind 6-10 dependent launches per pass at one launch per layer.  mg_smallnet runs such a chain as ONE launch: a workgroup
 * carries `imgs_per_wg` images through a list of ops with the activations in LDS ([pixel incl. a zero halo ring][channel]), the
 * 3x3 filters streamed from L2 straight into MFMA operand registers (layout of MG_PACK_SMALLNET / mg_smallnet_packed_floats:
 * [Cin/16][9 taps][Cout/16][64 lanes][4], dgrad = 1: the flipped / transposed filters of the data gradient) and every tensor a
 * later pass needs (activations, masks' sources, masked gradients) written to global memory on the way.  The same op list
 * serves the forward passes, the data-gradient chains and the tangent pass of the gradient penalty; weight gradients stay
 * per-layer launches on the stored tensors.  Ops (global tensors are (N, C, H, W) fp32; `src` / `dst` = LDS buffer 0..2):
 *   MG_SN_LOAD     dst <- in                                             (C, H, W)
 *   MG_SN_STORE    out <- src
 *   MG_SN_CONV     dst <- act(conv3x3(src, w) + bias), also -> out       (C -> C2 at H x W; flags MG_SN_LRELU | MG_SN_MASK: act =
 *                  LeakyReLU, or multiplication by lrelu'(aux) = (aux > 0 ? 1 : slope): the tangent pass / LeakyReLU backward)
 *                  1x1 maps take the centre tap in fp64 accumulation (what mg_conv3x3 does there)
 *   MG_SN_MASK     src <- src * lrelu'(aux), also -> out                 (LeakyReLU backward on its own)
 *   MG_SN_PIXNORM  src <- src * rn, rn = 1/sqrt(mean_c src^2 + 1e-8); p -> out, rn -> out2 (N,1,H,W)        [layers.py:11-17]
 *   MG_SN_PNBWD    src <- lrelu'(p) * rn * (src - p * mean_c(src * p)), also -> out;  p = in, rn = aux (PixelNorm + LeakyReLU backward)
 *   MG_SN_POOL     dst <- AvgPool2d(2,2)(src), also -> out               (H, W = input size)
 *   MG_SN_POOLBWD  dst <- 0.25 * up2(src) * lrelu'(aux), also -> out     (H, W = input size; MG_SN_NOLDS: only -> out)
 *   MG_SN_UP       dst <- nearest-upsample x2 of src                     (H, W = input size)
 *   MG_SN_UPBWD    dst <- 2x2 block sums of src                          (H, W = input size)
 *   MG_SN_LINEAR   out[n] <- sum_c w[c] * src[c] + bias[0]               (1x1 map; w = in, bias = aux; fp64 accumulation)
 *   MG_SN_LINBWD   dst[c] <- in[n] * aux[c]                              (in = upstream (N,1), aux = classifier weight)
 * Limits: H, W <= 8 and imgs_per_wg * H * W <= 64 at every CONV, C <= 192, lds_floats_per_buffer * 12 bytes <= 160 KB
 * (mg_smallnet_buffer_floats gives the size one tensor needs).  descs is a HOST array of at most MG_SN_MAX_OPS records. */
enum { MG_SN_LOAD = 0, MG_SN_STORE, MG_SN_CONV, MG_SN_MASK, MG_SN_PIXNORM, MG_SN_PNBWD, MG_SN_POOL, MG_SN_POOLBWD, MG_SN_UP,
       MG_SN_UPBWD, MG_SN_LINEAR, MG_SN_LINBWD };
#define MG_SN_LRELU 1
#define MG_SN_MASK_AUX 2
#define MG_SN_NOLDS 4
#define MG_SN_MAX_OPS 32
typedef struct {
  int32_t op, src, dst, C, C2, H, W, flags;
  const float* in;  /* LOAD: tensor; CONV: packed filters; PNBWD: p; LINEAR: weight; LINBWD: upstream */
  const float* aux; /* CONV: mask source (MG_SN_MASK_AUX); MASK / POOLBWD: mask source; PNBWD: rn; LINEAR: bias; LINBWD: weight */
  const float* bias; /* CONV */
  float* out;
  float* out2; /* PIXNORM: rn */
} mg_sn_op_t;
size_t mg_smallnet_packed_floats(int Cin, int Cout);
size_t mg_smallnet_buffer_floats(int imgs_per_wg, int C, int H, int W);
int mg_smallnet(const mg_sn_op_t* ops, int nops, int N, int imgs_per_wg, size_t lds_floats_per_buffer, float slope,
                mg_stream_t stream);

/* One 3x3 convolution (the nn.Conv2d(3x3) + LeakyReLU / AvgPool2d / Upsample neighbours of mg_conv3x3 above) on maps of at
 * most 16x16 as a latency-optimised launch: a workgroup takes a few images x ONE 16-out-channel tile, its waves split the input
 * channels and have their whole filter share in flight at once (one memory round trip instead of Cin / 8 dependent ones).
 * wpk: MG_PACK_SMALLNET filters (dgrad = 1 for the data gradient).  y (N,Cout,H,W) unless noted; flags:
 *   MG_CONV_UPS_IN     x is (N,Cin,H/2,W/2), nearest-upsampled on the way in [generator.py:26-29]
 *   MG_CONV_LRELU      y = leaky_relu(conv + bias);   MG_CONV_MASK_AUX   y = conv * lrelu'(aux), aux (N,Cout,H,W) (may alias y)
 *   MG_CONV_POOL_OUT   also p (N,Cout,H/2,W/2) = AvgPool2d(2,2)(y) [discriminator.py:24];  y may be NULL
 *   MG_CONV_UPSUM_OUT  also p = 2x2 block SUMS of y (backward of the nearest upsampling in front of the forward conv)
 *   MG_CONV_UNPOOL     alone: y is (N,Cout,2H,2W) = 0.25 * up2(conv) * lrelu'(aux), aux fp32 of that shape (AvgPool2d backward +
 *                      LeakyReLU backward of the layer below; mg_wino3x3's flag of this name takes tile-mask bytes instead) */
#define MG_CONV_UPSUM_OUT 256
int mg_conv3x3_small_supported(int N, int Cin, int Cout, int H, int W);
int mg_conv3x3_small(const float* x, const float* wpk, const float* bias, const float* aux, float* y, float* p, int N, int Cin,
                     int Cout, int H, int W, int flags, float slope, mg_stream_t stream);
/* The same launch with the PixelNorm of the layer in front [layers.py:11-17; generator.py:22-23,38-39: ... LeakyReLU -> PixelNorm
 * -> (Upsample ->) Conv2d] folded into its input staging: x_raw is the UN-normalised activation, y = act(conv(PixelNorm(x_raw)) +
 * bias); pn_p (N,Cin,Hin,Win) and pn_rn (N,1,Hin,Win) receive the normalised activation and 1 / norm (what the backward pass keeps;
 * may be NULL).  flags: MG_CONV_UPS_IN, MG_CONV_LRELU. */
int mg_conv3x3_small_pn(const float* x_raw, const float* wpk, const float* bias, float* y, float* pn_p, float* pn_rn, int N, int Cin,
                        int Cout, int H, int W, int flags, float slope, mg_stream_t stream);

/* ------------------------------------------------------------------ magnitude/phase codec + inverse STFT
 * mg_codec_fwd: stft_to_phase_magn [audio/functions.py:65-94].  stft_c64: interleaved complex64 [512][T] (mg_stft_1024 output);
 * bark_scale: [512] unit-norm bark vector (functions.py:26-35); outputs [S][512][nb_vec], S = (T-1)/nb_vec, both in [-1,1].
 * mg_codec_inv: magn_phase_to_wav [audio/functions.py:97-139] without the file write.  magn_phase: [N][2][512][W];
 * wav_out: [256*(N*W-1)].  The forward unwrap's cumulative sum [functions.py:23] is torch.cumsum's: a float64 running sum
 * rounded to float32 per frame, evaluated as an exact blocked scan (T <= 2^24); the inverse's cumulative phase
 * [functions.py:117-118] is the reference's sequential float32 loop.
 * mg_codec_fwd_strided: the same with consecutive images `img_stride` floats apart (>= 512*nb_vec): with
 * phase_out = magn_out + 512*nb_vec and img_stride = 2*512*nb_vec the outputs ARE the (S, 2, 512, nb_vec) tensor that
 * create_dataset stacks [create_dataset.py:52-58], written once. */
size_t mg_codec_fwd_ws_bytes(int T);
int mg_codec_fwd(const float* stft_c64, const float* bark_scale, float* magn_out, float* phase_out, void* ws,
                 size_t ws_bytes, int T, int nb_vec, mg_stream_t stream);
int mg_codec_fwd_strided(const float* stft_c64, const float* bark_scale, float* magn_out, float* phase_out, size_t img_stride,
                         void* ws, size_t ws_bytes, int T, int nb_vec, mg_stream_t stream);
size_t mg_codec_inv_ws_bytes(int N, int W);
int mg_codec_inv(const float* magn_phase, const float* bark_scale, float* wav_out, void* ws, size_t ws_bytes, int N, int W,
                 mg_stream_t stream);

/* CRC-32 (zip / zlib) of the float64 widening of float32 samples: crc_out[i] = crc32 of the little-endian bytes of
 * x[i*floats_per_sample ...].astype(float64) -- the checksum the zip container of `th.save(sample.to(th.float64))`
 * [create_dataset.py:52-62] stores for its payload, which the reference's writer computes on one host core per sample.
 * floats_per_sample = 8192 x a power of two <= 256 (a (2,512,512) sample: 524 288).  ws: mg_crc32_f64_ws_bytes. */
size_t mg_crc32_f64_ws_bytes(int n, int64_t floats_per_sample);
int mg_crc32_f64(const float* x, uint32_t* crc_out, void* ws, size_t ws_bytes, int n, int64_t floats_per_sample, mg_stream_t stream);

/* HOST function (no GPU work): the writer threads' per-sample work of create_dataset [create_dataset.py:52-62, th.save(magn_phase.to(
 * th.float64), path)] for n consecutive float32 samples `rows` (row_floats each, e.g. a slice of a pinned chunk): widen to float64 and
 * write file i as  prefix | float64 payload | suffixes[i*suffix_len ...]  (the zip container of th.save around the payload, from
 * musicgan_amd.fast_pt.PtTemplate) to the i-th NUL-terminated path in `paths`; side_fd >= 0: also pwrite the float32 row at
 * side_off + i * row_floats * 4 of that file (the fast loader's side-car).  Thread-safe; bound through ctypes it runs without the
 * interpreter lock. */
int mg_pt_write_samples(const float* rows, int n, int64_t row_floats, const char* paths, const unsigned char* prefix, int64_t prefix_len,
                        const unsigned char* suffixes, int64_t suffix_len, int side_fd, int64_t side_off);
/* Measurement helper (bench.py `host_io_probe`, no GPU work): what one writer thread's sample costs the host as plain system calls
 * -- n x { new file of file_bytes from a zero buffer; pwrite of side_bytes } and n x { row_floats float32 -> float64 } -- timed
 * separately; called from several threads at once.  The files <dir>/probe_raw_<tid>_<i>.bin, probe_rawside_<tid>.bin are left. */
int mg_host_io_probe(const char* dir, int tid, int n, int64_t file_bytes, int64_t side_bytes, const float* src, int64_t row_floats,
                     double* write_seconds, double* widen_seconds);

/* Per-batch input transform of the training loop, fused: ChannelMinMaxNorm -> ChangeRange(-1,1) -> Resize(S) (bilinear with
 * anti-aliasing, align_corners = False: torchvision's tensor path) [audio/transforms.py:4-40, utils.py:70-86, train.py:138-140].
 * x (N,2,H,W) float64 (x_is_f64 != 0: cast to float32 on read, == x.to(th.float)) or float32; out (N,2,S,S) float32, S <= H, W. */
size_t mg_input_transform_ws_bytes(int N, int H, int W, int S);
int mg_input_transform(const void* x, int x_is_f64, float* out, void* ws, size_t ws_bytes, int N, int H, int W, int S, float eps,
                       mg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
