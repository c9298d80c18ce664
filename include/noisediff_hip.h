/*
 * noisediff_hip.h -- C ABI of libnoisediff_hip.so (gfx950 / MI355X).
 *
 * The reference (IVRL/NoiseDiff) has no native layer: its "kernels" are ATen ops
 * dispatched from Python.  This library is the native layer under
 * the two Python plug-in interfaces of the sampling hot path -- the arch class
 * (models/modules.py:32-41 -> models/archs/Diffusion_arch.py:447-646) and
 * GaussianDiffusion (models/denoising_diffusion_pytorch.py:167-451).  Each entry
 * point below names the reference code it replaces.  The Python side that binds
 * them with ctypes is noisediff_amd/_lib.py; INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated; activations are
 *     NHWC ("pixel-major": [B][H][W][C], `ld` = floats between consecutive pixels);
 *   - `stream` is a hipStream_t passed as void*; nothing here allocates, frees,
 *     synchronises or touches the default stream, so every call is legal inside
 *     hipStreamBeginCapture/EndCapture;
 *   - return value: 0 = ok, >0 = hipError_t of the launch, <0 = ND_E_* argument error.
 *     No exceptions cross the boundary; nd_last_error() gives a message.
 */
#ifndef NOISEDIFF_HIP_H
#define NOISEDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ND_E_BADARG   (-1)   /* null pointer / non-positive size            */
#define ND_E_SHAPE    (-2)   /* shape not supported by the kernel's tiling   */
#define ND_E_ALIGN    (-3)   /* pointer or stride not 16-byte aligned        */
#define ND_E_STATE    (-4)   /* graph/plan handle misuse                     */

/* ------------------------------------------------------------------ library */
int         nd_version(void);                 /* 1000*major + minor                     */
const char* nd_last_error(void);              /* thread-local, never NULL               */
int         nd_device_arch(char* buf, int n); /* gcnArchName of the current device      */

/* ------------------------------------------------------------------ operand descriptors */

/* How a GEMM-like kernel reads its activation operand (the "A" side).              */
enum nd_prologue {
    ND_PRO_NONE        = 0,  /* raw values                                              */
    ND_PRO_AFFINE_SILU = 1,  /* silu((x - M[b,c]) * A[b,c] + D[b,c]): GroupNorm + time
                                scale/shift + SiLU of Block.forward, Diffusion_arch.py:137-143 */
    ND_PRO_AFFINE_MAP_SILU = 2, /* same, then *(S[p,c]+1)+Sh[p,c] before SiLU: ResnetBlock2's
                                per-pixel scale/shift, Diffusion_arch.py:188-192           */
    ND_PRO_LAYERNORM   = 3,  /* LayerNorm over C of (x + vec[b,c]): AttnBlock.norm2 on
                                x + CrossAttention(...), Diffusion_arch.py:438-439          */
    ND_PRO_SILU        = 4,  /* silu(x): ResnetBlock2.mlp[0], Diffusion_arch.py:177         */
    ND_PRO_LEAKY       = 5,  /* LeakyReLU(0.2)(x): LSID's nn.LeakyReLU after every conv,
                                models/archs/SID_arch.py:58,108-168 (applied by the consumer)   */
    ND_PRO_LEAKY_SECOND = 6, /* LeakyReLU(0.2) on the p1 channels of a virtual concat only:
                                torch.cat((up(x), conv_k)), SID_arch.py:135,142,150,158          */
    ND_PRO_AFFINE_GENMAP_SILU = 7  /* ND_PRO_AFFINE_MAP_SILU with the maps FORMED IN THE KERNEL (r6; nd_conv3x3_wino4_nhwc_f32 only): scale | shift =
                                ResnetBlock2.mlp[1] (a 1x1 convolution 8 -> 2C, Diffusion_arch.py:177,188) of the ACTIVATED position embedding.  `map` =
                                silu(pos_emb) [B][H][W][8] (32 bytes per pixel instead of 8 C), `gamma` = mlp[1].weight [2C][8] (rows 0..C-1 scale,
                                C..2C-1 shift), `beta` = mlp[1].bias [2C].  One source, C a multiple of 16.                                             */
};

typedef struct nd_src {
    const float* p0;      /* first source                                               */
    const float* p1;      /* second source of a virtual concat (torch.cat dim=1,
                             Diffusion_arch.py:598,628,631,640) or NULL                 */
    int32_t c0, c1;       /* channels taken from p0 / p1 (multiples of 4)               */
    int32_t ld0, ld1;     /* pixel strides in floats                                    */
    int32_t mode;         /* enum nd_prologue                                           */
    int32_t upsample;     /* 1: p0 is (H/2, W/2), read through nearest x2 (nn.Upsample,
                             Diffusion_arch.py:74); conv3x3 only                        */
    int32_t unshuffle;    /* 1: p0 is (2H, 2W, c0/4) read as 'b c (h p1)(w p2) -> b (c p1 p2) h w'
                             (Diffusion_arch.py:80) with K order (p1 p2 c); pointwise only */
    int32_t map_blocked;  /* AFFINE_MAP: 0: `map` is [B][H][W][scale C | shift C] (the layout torch.chunk(2, dim=1) reads,
                             Diffusion_arch.py:188); 1: blocked by 16 channels, [B][H][W][C/16][scale 16 | shift 16] -- one
                             128-byte line per pixel and 16-channel K chunk; nd_conv3x3_wino4_nhwc_f32 only (C % 16 == 0) */
    const float* mad;     /* [B][3][C] M, A, D for the AFFINE modes (from nd_groupnorm_finalize_f32) */
    const float* map;     /* [B][H][W][2C] scale / shift map (AFFINE_MAP), layout per map_blocked */
    const float* vec;     /* [B][C] per-sample vector added before LayerNorm            */
    const float* gamma;   /* [C] LayerNorm weight                                       */
    const float* beta;    /* [C] LayerNorm bias                                         */
    const float* rowstats;/* [B*HW][2] per-pixel {mean, rstd} of (x + vec) from nd_layernorm_stats_f32;
                             required for LAYERNORM when C > 64 (for C <= 64 the GEMM derives them itself) */
} nd_src;

enum nd_act { ND_ACT_NONE = 0, ND_ACT_GELU = 1, ND_ACT_SILU = 2 };

/* ------------------------------------------------------------------ conv 3x3 */

/* nn.Conv2d(cin, cout, 3, padding=1) on NHWC fp32 with exact-fp32 MFMA accumulation
 * (v_mfma_f32_32x32x2_f32 == an fmaf chain).  Replaces Block.proj (Diffusion_arch.py:131,136),
 * the last-stage down/up convs (:533,:547) and Upsample's conv (:75).
 * Weights are pre-packed by nd_pack_conv3x3_weight.  When `stats` != NULL the epilogue
 * also emits per-(sample, slot, channel) partial sums {sum, M2} of the output for the
 * following GroupNorm (Block.norm :132); `slot_count[slot]` receives the number of pixels
 * behind each slot.  Query the slot count with nd_conv3x3_stat_slots. */
typedef struct nd_conv3x3 {
    nd_src   src;
    const float* weight;   /* packed, see nd_pack_conv3x3_weight                         */
    const float* bias;     /* [cout] or NULL                                             */
    float*   out;          /* [B][H][W] x ldo                                            */
    float*   stats;        /* [B][slots][cout][2] or NULL                                */
    float*   slot_count;   /* [slots] or NULL                                            */
    int32_t  B, H, W;      /* OUTPUT spatial size                                        */
    int32_t  cin, cout, ldo;
} nd_conv3x3;

int nd_conv3x3_nhwc_f32(const nd_conv3x3* d, void* stream);
int nd_conv3x3_stat_slots(int H, int W, int cout, int B);
/* which conv3x3_kernel<TW, MB, NB> instance a shape runs on: TW*100 + MB*10 + NB (for profiling reports) */
int nd_conv3x3_tiling_id(int B, int H, int W, int cout);
/* OIHW (cout,cin,3,3) -> [tap][cin/4][coutP][4], coutP = cout rounded up to 64; zero padded. */
int64_t nd_pack_conv3x3_weight_floats(int cin, int cout);
int nd_pack_conv3x3_weight(const float* oihw, float* packed, int cin, int cout, void* stream);
/* training (loss.backward() of p_losses, models/trainer_diffusion.py:187): the same packing of the DATA-GRADIENT operator of a layer
 * (nn.Conv2d(cin, cout, 3, padding=1), Diffusion_arch.py:131), read in place from the forward layer's OIHW weight
 * `oihw_fwd` (cout x cin there = cin x cout here): g'[co][ci][r][s] = oihw_fwd[ci][co][2 - r][2 - s].  `cin`, `cout` are the
 * data-gradient convolution's (cin = the forward layer's cout).  Likewise for the two Winograd packings below. */
int nd_pack_conv3x3_weight_dgrad(const float* oihw_fwd, float* packed, int cin, int cout, void* stream);

/* Same operator, same descriptor, computed with Winograd F(2x2,3x3) (2.25x fewer multiplies; fp32 transforms,
 * ~1e-6 relative difference to the direct form).  `weight` must come from nd_pack_conv3x3_wino_weight
 * (U = G g G^T in 128 KB blocks [cinP/32][coutP/64][16 positions][8 quads][64][4]); statistics slots:
 * nd_conv3x3_wino_stat_slots.  Used for images >= 16x16. */
int nd_conv3x3_wino_nhwc_f32(const nd_conv3x3* d, void* stream);
/* Same contract, one-workgroup-per-CU variant: all 16 Winograd position accumulators stay in registers across the
 * K loop and the halo tile is double-buffered in LDS (conv3x3_wino2.hip). */
int nd_conv3x3_wino2_nhwc_f32(const nd_conv3x3* d, void* stream);
int nd_conv3x3_wino_stat_slots(int H, int W);
int64_t nd_pack_conv3x3_wino_weight_floats(int cin, int cout);
int nd_pack_conv3x3_wino_weight(const float* oihw, float* packed, int cin, int cout, void* stream);
int nd_pack_conv3x3_wino_weight_dgrad(const float* oihw_fwd, float* packed, int cin, int cout, void* stream);

/* The same operator with Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32 (1.78x fewer multiplies than F(2x2,3x3); ~1e-5 relative
 * difference to the direct form): the kernel that carries the sampling path (conv3x3_wino4.hip).  Takes plain / two-source
 * (concat on a 16-channel boundary) inputs, the GroupNorm-affine (+ per-pixel map) + SiLU and LeakyReLU prologues, nearest-x2
 * upsample addressing of a single source and the statistics epilogue; needs cin > 16, cout <= 2048, W <= 2048, sources below
 * 1 GiB / 2^24 pixels (rejected otherwise: the caller picks nd_conv3x3_wino2_nhwc_f32).  `weight` from
 * nd_pack_conv3x3_wino4_weight (U = G g G^T in blocks [cin/8][coutP/16][18 position pairs][64 lanes][4]). */
int nd_conv3x3_wino4_nhwc_f32(const nd_conv3x3* d, void* stream);
/* The same kernel on 16 x 16-pixel regions with TWO co-resident workgroups per CU (two waves per SIMD: one workgroup's LDS round trips,
 * barriers, store queue and transforms run under the other's MFMAs; r4).  Same packed weights, statistics slots and descriptor, and the
 * same bits as nd_conv3x3_wino4_nhwc_f32 (identical arithmetic in identical order).  Plain and GroupNorm-affine + SiLU sources only
 * (Block.proj of ResnetBlock, Diffusion_arch.py:128-170; the resampling convs :75,533,547); also the F(4x4) path of images narrower
 * than 32 pixels (BASELINE config 2's 16 x 16 stage). */
int nd_conv3x3_wino4_16_nhwc_f32(const nd_conv3x3* d, void* stream);
/* its statistics epilogue writes ONE slot per 16 x 16-pixel tile (the F(2x2) kernels: two) */
int nd_conv3x3_wino4_stat_slots(int H, int W);
int64_t nd_pack_conv3x3_wino4_weight_floats(int cin, int cout);
int nd_pack_conv3x3_wino4_weight(const float* oihw, float* packed, int cin, int cout, void* stream);
int nd_pack_conv3x3_wino4_weight_dgrad(const float* oihw_fwd, float* packed, int cin, int cout, void* stream);
/* Many weights in one launch (the training path repacks every Block.proj weight, forward and data-gradient form, once per optimizer step):
 * nd_pack_item records in DEVICE memory (shared with nd_pack_pointwise_weights_batch below); item.w = OIHW weight (the FORWARD layer's when
 * item.transposed != 0, which selects the _dgrad form; cin / cout are then the data-gradient operator's), item.packed = its
 * nd_pack_conv3x3_wino4_weight_floats(cin, cout) floats. */
typedef struct nd_pack_item {
    const float* w;
    float*       packed;
    int32_t      cin, cout, transposed, reserved;
} nd_pack_item;
int nd_pack_conv3x3_wino4_weights_batch(const nd_pack_item* items_dev, int n_items, void* stream);
/* Split-K form of nd_conv3x3_wino4_nhwc_f32 for plain and GroupNorm-affine + SiLU sources -- the forward and data-gradient convolutions of
 * Block.proj under GaussianDiffusion.p_losses (models/archs/Diffusion_arch.py:128-144; models/denoising_diffusion_pytorch.py:481-531) at
 * training batch sizes (512 -> 512 at
 * 32 x 32 with 4 samples is 64 workgroup items for 256 CUs, each walking 32 K chunks): cin is cut into `splits` ranges (2, 4 or 8; whole
 * 16-channel chunks, at least two per range), every (range, sample, region, cout tile) is a workgroup item that writes partial sums to
 * `workspace` ([splits][B][H][W][cout] floats), and a second kernel adds them in range order and the bias into d->out -- and, when d->stats is
 * set, leaves the statistics slots of the summed output as the plain kernel's epilogue does (one slot per 16 x 16 tile).  _plan returns the
 * split count for a shape (1 = use the plain entry) -- a function of the shape alone, so results never depend on the device. */
int nd_conv3x3_wino4_splitk_plan(int B, int H, int W, int cin, int cout);
int64_t nd_conv3x3_wino4_splitk_workspace_floats(int B, int H, int W, int cout, int splits);
int nd_conv3x3_wino4_splitk_nhwc_f32(const nd_conv3x3* d, float* workspace, int splits, void* stream);
/* Split-K on the 16 x 16-region form, for SAMPLING (r4): layers with few (region, cout tile) items per sample -- the 16 x 16 and 32 x 32 stages of
 * BASELINE config 2 (Diffusion_arch.py:533,547 at H/8) -- are cut along cin.  _plan looks at the SAMPLE's geometry only (no batch argument), so a
 * sample's bits never depend on the batch it is sharded into: 2, 4 or 8 ranges so that a sample has about 32 items, at least four chunks per range.
 * Workspace: nd_conv3x3_wino4_splitk_workspace_floats. */
int nd_conv3x3_wino4_16_splitk_plan(int H, int W, int cin, int cout);
int nd_conv3x3_wino4_16_splitk_nhwc_f32(const nd_conv3x3* d, float* workspace, int splits, void* stream);

/* ------------------------------------------------------------------ conv 3x3, training */

/* Weight gradient of nn.Conv2d(cin, cout, 3, padding=1): dw[co][ci][r][s] = sum_{b,y,x} dy[b][y][x][co] * x[b][y+r-1][x+s-1][ci]
 * (zero padding), x and dy NHWC fp32, dw in the torch OIHW layout.  The backward of Block.proj and the resampling convs under
 * GaussianDiffusion.p_losses (models/denoising_diffusion_pytorch.py:481-531; loss.backward() at models/trainer_diffusion.py:187).
 * Exact-fp32 MFMA; the split over pixel tiles depends on the shape only and the partial sums are added in a fixed order, so the
 * result is bitwise repeatable.  `workspace`: nd_conv3x3_wgrad_workspace_floats(...) floats.  `dbias` (may be NULL): the bias
 * gradient db[co] = sum_{b,y,x} dy[b][y][x][co], from the same dy tiles (same fixed order).
 * Two forms, chosen by the shape alone.  H % 4 == 0, W % 16 == 0 and both channel counts multiples of 32: the Winograd-domain F(4x4,3x3)
 * form -- per 4 x 4 pixel tile the 6 x 6 transforms of the input patch and of the dY block, 36 position-wise products (a quarter of the
 * nine-tap form's MFMAs), one back-transform G^T M G at the end; error ~4e-6 of max|dW| against a float64 sum.  Anything else: nine tap
 * GEMMs over the pixels (4e-7).  ND_WGRAD_WINO=0 forces the nine-tap form (A/B); nd_conv3x3_wgrad_form pins any of them.
 * The DATA gradient of the same layer is the forward operator itself: nd_conv3x3_*_nhwc_f32 on weights packed by
 * nd_pack_conv3x3_*_weight_dgrad (taps flipped, channel roles swapped) -- see noisediff_amd/train.py. */
int64_t nd_conv3x3_wgrad_workspace_floats(int B, int H, int W, int cin, int cout);
/* Tests and A/B tools: pin the form -- 0 by the shape (default), 1 nine taps, 2 Winograd domain on four waves, 3 on eight waves (cout % 64 == 0;
 * else four); anything else only queries.  Returns the previous setting.  Process-wide; ask for the workspace size AFTER setting it. */
/* The same for an input that is the virtual concatenation cat((x0, x1), channels) -- the up path's skip connections -- without the concatenated
 * tensor: dw (cout, c0 + c1, 3, 3).  Winograd-domain forms only (H % 4 == 0, W % 16 == 0, c0, c1, cout multiples of 32): ND_E_SHAPE otherwise
 * (concatenate and call nd_conv3x3_wgrad_nhwc_f32).  Workspace: nd_conv3x3_wgrad_workspace_floats(B, H, W, c0 + c1, cout). */
int nd_conv3x3_wgrad_cat_nhwc_f32(const float* x0, int ldx0, int c0, const float* x1, int ldx1, int c1, const float* dy, int ldy, float* dw_oihw,
                                  float* dbias, float* workspace, int B, int H, int W, int cout, void* stream);
int nd_conv3x3_wgrad_form(int form);
int nd_conv3x3_wgrad_nhwc_f32(const float* x, int ldx, const float* dy, int ldy, float* dw_oihw, float* dbias, float* workspace,
                              int B, int H, int W, int cin, int cout, void* stream);

/* nn.GroupNorm(groups, C) forward and backward on NHWC fp32 for training: Block.norm (Diffusion_arch.py:132,138) under
 * GaussianDiffusion.p_losses -> loss.backward().  Statistics per (sample, group) over (C / groups) x HW values, eps inside the
 * square root, affine weight and bias -- torch.nn.functional.group_norm's definition.  Forward: y = (x - mean) rstd gamma + beta and
 * `mean_rstd` [B][groups][2] for the backward.  Backward: dx, dgamma [C], dbeta [C] from dy, x and the saved statistics.
 * Every pass streams NHWC once (partial sums per pixel slot, fp64 finalize, one elementwise pass); fixed summation order.
 * `workspace`: nd_groupnorm_train_workspace_floats(B, HW, C) floats, shared by both directions. */
int64_t nd_groupnorm_train_workspace_floats(int B, int HW, int C);
int nd_groupnorm_train_forward_f32(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* mean_rstd,
                                   float* workspace, int B, int HW, int C, int groups, float eps, void* stream);
int nd_groupnorm_train_backward_f32(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* mean_rstd,
                                    float* dx, int lddx, float* dgamma, float* dbeta, float* workspace, int B, int HW, int C, int groups,
                                    void* stream);

/* Block's tail as one training operator: y = silu(GroupNorm(x) * (scale + 1) + shift) (Block.forward, Diffusion_arch.py:137-143) with the
 * per-(sample, channel) scale / shift of the time embedding (`scale_shift` [B][2C] = scale | shift as ResnetBlock.mlp emits it, :150-152,162-164;
 * NULL: plain GroupNorm + SiLU).  Forward saves `mean_rstd` [B][groups][2] and `mad` [B][3][C]; backward returns dx, dgamma, dbeta and
 * `dscale_shift` [B][2C] (NULL iff scale_shift is NULL).  Two passes over the tensor per direction (the separate ops take four and seven).
 * `workspace`: nd_groupnorm_silu_train_workspace_floats(B, HW, C) floats. */
int64_t nd_groupnorm_silu_train_workspace_floats(int B, int HW, int C);
int nd_groupnorm_silu_train_forward_f32(const float* x, int ldx, const float* gamma, const float* beta, const float* scale_shift, const float* res, int ldr,
                                        float* y, int ldy, float* mean_rstd, float* mad, float* workspace, int B, int HW, int C, int groups, float eps,
                                        void* stream);   /* res (may be NULL): + the ResnetBlock's shortcut, h + res_conv(x) (Diffusion_arch.py:170), in the same pass */
int nd_groupnorm_silu_train_backward_f32(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta,
                                         const float* scale_shift, const float* mean_rstd, const float* mad, float* dx, int lddx,
                                         float* dgamma, float* dbeta, float* dscale_shift, float* workspace, int B, int HW, int C, int groups,
                                         void* stream);

/* ResnetBlock2's modulation + activation for training (Diffusion_arch.py:173-196: scale / shift are per-pixel MAPS from the position
 * embedding): y = silu(n * (scale + 1) + shift), n [N][>= C] the GroupNorm's output, map [N][>= 2C] = scale | shift as ResnetBlock2.mlp emits it.
 * Backward: dn, and dmap [N][2C] = d scale | d shift, one pass each (the separate ops take four and five). */
int nd_modulate_silu_forward_f32(const float* n, int ldn, const float* map, int ldm, float* y, int ldy, int64_t N, int C, void* stream);
int nd_modulate_silu_backward_f32(const float* dy, int lddy, const float* n, int ldn, const float* map, int ldm, float* dn, int lddn, float* dmap, int lddm,
                                  int64_t N, int C, void* stream);

/* out[b][c] = sum over the HW tokens of x[b][p][c]: the gradient of a per-sample vector broadcast over the tokens -- AttnBlock's one-token ISO
 * cross attention adds to_out(to_v(ctx)) to every token (Diffusion_arch.py:435-437).  Fixed order; workspace: nd_token_sum_workspace_floats. */
int64_t nd_token_sum_workspace_floats(int B, int HW, int C);
int nd_token_sum_f32(const float* x, int ldx, float* out, float* workspace, int B, int HW, int C, void* stream);

/* nn.LayerNorm(C) over the channels of NHWC tokens for training (AttnBlock.norm1 / norm2, Diffusion_arch.py:427-428): forward
 * y = (x - mean) rstd gamma + beta with `stats` [N][2] = {mean, rstd} per token saved for the backward; backward dx, dgamma [C],
 * dbeta [C].  C = 64, 128 or a multiple of 256 up to 1024 (a row lives in 16 / 32 / 64 lanes); eps inside the square root, biased
 * variance -- torch.nn.functional.layer_norm's definition.  One streaming pass per direction, fixed summation order.
 * `workspace` (backward): nd_layernorm_train_workspace_floats(N, C) floats. */
int64_t nd_layernorm_train_workspace_floats(int64_t N, int C);
int nd_layernorm_train_forward_f32(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* stats,
                                   int64_t N, int C, float eps, void* stream);
int nd_layernorm_train_backward_f32(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* stats,
                                    float* dx, int lddx, float* dgamma, float* dbeta, float* workspace, int64_t N, int C, void* stream);

/* Weight and bias gradient of a token Linear / 1x1 convolution: dw[co][ci] = sum_p dy[p][co] * x[p][ci], dbias[co] = sum_p dy[p][co]
 * over N tokens (x, dy: [N][ld] fp32, NHWC pixels are tokens; dw in torch's (cout, cin) layout; dbias may be NULL).  The backward of
 * res_conv, Mlp.fc1/fc2, FeedForward, proj_out and the attention projections (Diffusion_arch.py:156,345-347,410-419,432) under
 * GaussianDiffusion.p_losses.  Exact-fp32 MFMA, fixed split and summation order (bitwise repeatable).
 * `workspace`: nd_linear_wgrad_workspace_floats(N, cin, cout) floats.  (The data gradient is a plain GEMM, dy @ W.) */
int64_t nd_linear_wgrad_workspace_floats(int64_t N, int cin, int cout);
int nd_linear_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* workspace,
                        int64_t N, int cin, int cout, void* stream);

/* ------------------------------------------------------------------ pointwise GEMM */

/* out[p, n] = epi( sum_k pro(in[p, k]) * W[k, n] + bias[n] ) per pixel: nn.Conv2d(k=1) and
 * nn.Linear on tokens are the same op in NHWC.  Replaces res_conv (:156), Downsample's conv
 * (:81), Mlp.fc1/fc2 (:345-347), ResnetBlock2.mlp (:178), FeedForward (:410-419),
 * AttnBlock.proj_out (:432), final_conv (:554).
 * Epilogue, in order: +bias, act, +res0, +res1, +vec[b,n], +silu((t - M)*A + D) where t is a
 * raw conv output (ResnetBlock tail `h + res_conv(x)`, :170). */
typedef struct nd_pointwise {
    nd_src   src;
    const float* weight;   /* packed [cinP/4][coutP][4], see nd_pack_pointwise_weight     */
    const float* bias;     /* [cout] or NULL                                             */
    float*   out;
    const float* res0; const float* res1;   /* [B][HW] x ldr0/ldr1 or NULL               */
    const float* vec;      /* [B][cout] or NULL                                          */
    const float* gn_t;     /* tensor t for the fused GroupNorm+SiLU add, or NULL          */
    const float* gn_mad;   /* [B][3][cout]                                               */
    int32_t  B, HW, W;     /* HW = output pixels per sample; W = output width (unshuffle) */
    int32_t  cin, cout, ldo, ldr0, ldr1, ldt;
    int32_t  act;          /* enum nd_act                                                */
    int32_t  shuffle_c;    /* > 0: ConvTranspose2d(k=2, s=2) scatter (SID_arch.py:84-99): cout = 4*shuffle_c ordered
                              (p1 p2 c); row p=(y,x) writes channel c of output pixel (2y+p1, 2x+p2) of a
                              (shuffle_h, shuffle_w) image (rows/cols beyond it are cropped, :135)   */
    int32_t  shuffle_h, shuffle_w;
} nd_pointwise;

int nd_pointwise_gemm_nhwc_f32(const nd_pointwise* d, void* stream);
/* (cout, cin) row-major (Linear / 1x1 conv weight) -> [cinP/4][coutP][4]; `unshuffle_c` > 0
 * permutes K from (c p1 p2) to (p1 p2 c) for a pixel-unshuffled input with c = unshuffle_c. */
int64_t nd_pack_pointwise_weight_floats(int cin, int cout);
int nd_pack_pointwise_weight(const float* w, float* packed, int cin, int cout, int unshuffle_c, void* stream);
/* training: the packing of the DATA-GRADIENT operator dx = dy @ W of a Linear / 1x1 convolution (res_conv, Mlp.fc1 / fc2, FeedForward:
 * Diffusion_arch.py:156,345-347,410-419), read in place from its forward weight
 * `w_t` ((cin, cout) row-major here = torch's (out_features, in_features) of the forward layer; cin = the forward layer's cout). */
int nd_pack_pointwise_weight_t(const float* w_t, float* packed, int cin, int cout, void* stream);
/* Many packings in ONE launch (training: the forward and the data-gradient packing of every Linear / 1x1 weight once per optimizer step;
 * noisediff_amd/train.py keeps the table).  `items_dev`: n_items records in DEVICE memory; item i packs `w` ((cout, cin) row-major, or with
 * `transposed` the forward weight (cin, cout) of the layer whose data gradient this packing serves) into `packed` (nd_pack_pointwise_weight_floats
 * floats), exactly as nd_pack_pointwise_weight / nd_pack_pointwise_weight_t do. */
int nd_pack_pointwise_weights_batch(const nd_pack_item* items_dev, int n_items, void* stream);

/* The same operator (same descriptor, prologues, epilogues, errors) with every product on the bf16 matrix cores at the operands' FULL fp32
 * significand: each fp32 operand is split exactly into three bf16 terms (v = v1 + v2 + v3, round-to-nearest-even remainders), six of the nine
 * term products are kept (the dropped ones lie below 2^-25 of the product -- under the rounding of an fp32 FMA), accumulation in fp32.  Replaces
 * the same reference layers as nd_pointwise_gemm_nhwc_f32 where they are wide: FeedForward's Linears and AttnBlock.proj_out at C >= 128
 * (Diffusion_arch.py:405-443), res_conv of the up path (:156), Attention.to_qkv / to_out (:252-253).  `weight`: nd_pack_pointwise_weight_split
 * ((cout, cin) row-major -> [cin/16][term 3][cout/32][64 lanes][8 bf16]).  Takes: cin % 32 == 0, cin >= 64, cout % 128 == 0, plain pixel
 * addressing, LayerNorm with src.rowstats, a concat on a 32-channel boundary (nd_pointwise_gemm_split_takes); anything else: ND_E_SHAPE. */
int nd_pointwise_gemm_split_nhwc_f32(const nd_pointwise* d, void* stream);
int nd_pointwise_gemm_split_takes(const nd_pointwise* d);
int64_t nd_pack_pointwise_weight_split_floats(int cin, int cout);
int nd_pack_pointwise_weight_split(const float* w, float* packed, int cin, int cout, void* stream);

/* ------------------------------------------------------------------ chained pointwise layers
 * Two or three per-pixel Linear layers in one kernel, the intermediate activations never leaving registers:
 *   Mlp (Diffusion_arch.py:340-356):            out = fc2(act(fc1(x)))
 *   AttnBlock tail (:405-443, with src.vec = v): out = proj_out(ff2(GELU(ff1(LN(x + v)))) + x + v) + x
 * src: p0/p1 virtual concat, mode ND_PRO_NONE or ND_PRO_LAYERNORM (gamma, beta; src.vec[b] is added to the row first).
 * Stage i computes  h = act_i(W_i . h_prev + bias_i + res_i)  with res_i one of enum nd_chain_res.  Weights come from
 * nd_pack_chain_weight (operand order of the transposed MFMA product, K padded to 8 for the first stage and to 32 for
 * the later ones, N to 32).  Only the width combinations NoiseDiffNet uses are instantiated
 * (nd_pointwise_chain_supported); HW must be a multiple of 32. */
enum nd_chain_res { ND_CHAIN_RES_NONE = 0, ND_CHAIN_RES_INPUT = 1 /* + x + v */, ND_CHAIN_RES_INPUT_RAW = 2 /* + x */ };
typedef struct nd_chain_stage {
    const float* weight;
    const float* bias;     /* [cout] or NULL */
    int32_t  cin, cout;
    int32_t  act;          /* enum nd_act, applied after bias and residual */
    int32_t  res;          /* enum nd_chain_res; needs cout == input width */
} nd_chain_stage;
typedef struct nd_chain {
    nd_src   src;
    nd_chain_stage st[3];
    float*   out;
    int32_t  n_stages, B, HW, ldo;
} nd_chain;
int nd_pointwise_chain_nhwc_f32(const nd_chain* d, void* stream);
int nd_pointwise_chain_supported(int cin, int n1, int n2, int n3);   /* n3 = 0: two stages */
int64_t nd_pack_chain_weight_floats(int cin, int cout, int first_stage);
int nd_pack_chain_weight(const float* w, float* packed, int cin, int cout, int first_stage, void* stream);
/* The same chains (same descriptor, widths, prologues, residuals, errors) with the products on the bf16 matrix cores at full fp32 significand
 * (three-term split, six products, fp32 accumulation: see nd_pointwise_gemm_split_nhwc_f32); the stages' `weight` pointers are
 * nd_pack_chain_weight_split packings (three bf16 terms per value in the operand order of v_mfma_f32_32x32x16_bf16, K padded to 16 for the first
 * stage and to 32 for the later ones, N to 32). */
int nd_pointwise_chain_split_nhwc_f32(const nd_chain* d, void* stream);
int64_t nd_pack_chain_weight_split_floats(int cin, int cout, int first_stage);
int nd_pack_chain_weight_split(const float* w, float* packed, int cin, int cout, int first_stage, void* stream);

/* ------------------------------------------------------------------ GroupNorm plumbing */

/* Combine conv partials (Chan's parallel variance, fp64) into per-(b, group) mean / rstd
 * (nn.GroupNorm eps, biased variance) and fold gamma/beta and the optional time
 * scale/shift into M, A, D so that Block.forward's  GN -> x*(scale+1)+shift  (:137-141)
 * becomes (x - M) * A + D.  scale/shift: [B] rows of stride ld_ss, scale at +0, shift at +cout. */
int nd_groupnorm_finalize_f32(const float* stats, const float* slot_count, int slots,
                              const float* gamma, const float* beta,
                              const float* scale_shift, int ld_ss,
                              float* mad, int B, int C, int groups, float eps, void* stream);
/* training (Block.forward under p_losses): the same finalize, which also saves `mean_rstd` [B][groups][2] for the backward pass
 * (nd_groupnorm_silu_train_backward_f32) -- the convolution's statistics epilogue then feeds the norm in training as it does in sampling,
 * and the forward tail is nd_affine_silu_add_f32. */
int nd_groupnorm_finalize_train_f32(const float* stats, const float* slot_count, int slots, const float* gamma, const float* beta,
                                    const float* scale_shift, int ld_ss, float* mad, float* mean_rstd, int B, int C, int groups, float eps,
                                    void* stream);

/* Per-pixel LayerNorm statistics over channels of (x + vec[b]): stats[p] = {mean, 1/sqrt(var + eps)}
 * (nn.LayerNorm: biased variance; AttnBlock.norm2, Diffusion_arch.py:430,439).  vec may be NULL. */
int nd_layernorm_stats_f32(const float* x, int ldx, const float* vec, float* stats,
                           int B, int HW, int C, float eps, void* stream);

/* out = silu((t - M)*A + D) [+ res0] [+ res1]: the tail of ResnetBlock.forward (:168-170)
 * when res_conv is the identity, plus `shot_emb + r` (:603). */
int nd_affine_silu_add_f32(const float* t, int ldt, const float* mad,
                           const float* res0, int ldr0, const float* res1, int ldr1,
                           float* out, int ldo, int B, int HW, int C, void* stream);

/* ------------------------------------------------------------------ small dense ops on rows */

/* out[b, n] = act_out( sum_k act_in(in[b, k]) * W[n, k] + bias[n] ), W in torch (N, K) layout.
 * time_mlp (:502-507), every ResnetBlock.mlp (:149-152, batched as one tall matrix),
 * CrossAttention.to_v / to_out on the 1-token context (:385,:402). */
int nd_linear_rows_f32(const float* in, int ld_in, const float* W, const float* bias,
                       float* out, int ld_out, int B, int K, int N,
                       int act_in, int act_out, void* stream);
/* SinusoidalPosEmb.forward (:100-107): emb[b] = cat(sin(t*f), cos(t*f)), t int64 -> fp32 multiply.
 * freqs[half] = exp(arange(half) * -(ln(theta)/(half-1))) is a constant of the model, computed
 * once on the host so that the angle t*f is bit-identical to the reference's. */
int nd_sinusoidal_time_emb_f32(const int64_t* time, const float* freqs, float* emb, int B, int half, void* stream);
/* The whole time conditioning of one diffusion step in one launch (SURVEY 8b minimum symbol set):
 * out[b, j] = (Wp . silu(W2 . gelu(W1 . emb(time[b]) + b1) + b2) + bp)[j] with emb = SinusoidalPosEmb (:100-107), W1/W2 =
 * time_mlp[1]/[3] (:502-507), Wp/bp = every ResnetBlock.mlp[1] Linear stacked into one (J, 4 dim) matrix (:149-152; the SiLU
 * is ResnetBlock.mlp[0]).  Weights in torch (N, K) layout.  Dynamic LDS: nd_cond_step_lds_bytes(B, dim) <= 160 KB. */
int64_t nd_cond_step_lds_bytes(int B, int dim);
int nd_cond_step_f32(const int64_t* time, const float* freqs, const float* W1, const float* b1, const float* W2, const float* b2,
                     const float* Wp, const float* bp, float* out, int ld_out, int B, int dim, int J, void* stream);
/* The same step with the head (emb -> time_mlp -> SiLU) looked up instead of computed: `table` (table_rows x 4 dim) holds the head's result for
 * timestep = row index, built once per weight set by nd_cond_table_build_f32 (same kernel code: the very same bits).  A timestep outside the
 * table falls back to computing the head.  Replaces the same ATen call sites as nd_cond_step_f32 (Diffusion_arch.py:100-107, 502-507, 149-152). */
int nd_cond_table_build_f32(const float* freqs, const float* W1, const float* b1, const float* W2, const float* b2, float* table, int rows, int dim,
                            void* stream);
int nd_cond_step_table_f32(const int64_t* time, const float* freqs, const float* W1, const float* b1, const float* W2, const float* b2,
                           const float* Wp, const float* bp, float* out, int ld_out, int B, int dim, int J, const float* table, int table_rows,
                           void* stream);
/* ... and with the projection tabulated as well: `ptable` (table_rows x J, row t = the J outputs for timestep t, filled by calling nd_cond_step_table_f32 on the
 * timesteps 0 .. table_rows-1 with ld_out = J).  When every sample's timestep lies in the table the launch copies B rows (the same bits); otherwise it computes. */
int nd_cond_step_ptable_f32(const int64_t* time, const float* freqs, const float* W1, const float* b1, const float* W2, const float* b2,
                            const float* Wp, const float* bp, float* out, int ld_out, int B, int dim, int J, const float* table, int table_rows,
                            const float* ptable, void* stream);
/* nn.Embedding lookup (:591): out[b] = table[idx[b]], idx int64. */
int nd_embedding_rows_f32(const int64_t* idx, const float* table, float* out, int B, int rows, int dim, void* stream);

/* ------------------------------------------------------------------ full-resolution special layers */

/* init_conv: nn.Conv2d(4, cout, 7, padding=3) (:478,:606).  x NHWC with 4 channels;
 * weight packed [7*7*4][cout] by nd_pack_conv7x7_weight. */
int nd_conv7x7_c4_f32(const float* x, const float* wpacked, const float* bias, float* out, int ldo,
                      int B, int H, int W, int cout, void* stream);
/* training: weight and bias gradient of that stem, dw (cout, 4, 7, 7) OIHW = sum over the pixels of dy[p][co] x[p + tap][ci] (zero padding) and
 * db[co] = sum dy[p][co]; x NHWC with 4 channels, dy NHWC with pixel stride ldy.  The patch matrix is never built (GEMM with K = pixels straight from a
 * halo tile in LDS); fixed summation order (bitwise repeatable).  Taken for H % 4 == 0, W % 32 == 0, cout in {32, 48, 64, 96, 128}:
 * nd_conv7x7_c4_wgrad_workspace_floats returns -1 otherwise (unfold the image and use nd_linear_wgrad_f32).  dbias may be NULL. */
int64_t nd_conv7x7_c4_wgrad_workspace_floats(int B, int H, int W, int cout);
int nd_conv7x7_c4_wgrad_f32(const float* x, const float* dy, int ldy, float* dw_oihw, float* dbias, float* workspace, int B, int H, int W, int cout,
                            void* stream);
int nd_pack_conv7x7_weight(const float* oihw, float* packed, int cout, void* stream);
/* The same stem (same arguments, shapes, errors) with the products on the bf16 matrix cores at the operands' full fp32 significand: three bf16 terms per
 * fp32 value, six of nine term products, fp32 accumulation (see nd_pointwise_gemm_split_nhwc_f32).  `wsplit`: nd_pack_conv7x7_weight_split
 * ((cout, 4, 7, 7) OIHW -> [cout/32][13 K steps][term 3][64 lanes][8 bf16], nd_pack_conv7x7_weight_split_floats(cout) floats). */
int nd_conv7x7_c4_split_f32(const float* x, const float* wsplit, const float* bias, float* out, int ldo,
                            int B, int H, int W, int cout, void* stream);
int64_t nd_pack_conv7x7_weight_split_floats(int cout);
int nd_pack_conv7x7_weight_split(const float* oihw, float* packed, int cout, void* stream);
/* LearnedSinusoidalPosEmb (:331-337): position NCHW (B,2,H,W) -> NHWC (B,H,W,3*hid):
 * w = conv1x1(position); cat(w, sin(2 pi w), cos(2 pi w)). */
int nd_pos_enc_f32(const float* position_nchw, const float* w /*[hid][2]*/, const float* bias,
                   float* out, int B, int H, int W, int hid, void* stream);
/* nn.MaxPool2d(2, 2, ceil_mode=True) on NHWC (SID_arch.py:60): out (ceil(H/2), ceil(W/2)). */
int nd_maxpool2x2_nhwc_f32(const float* in, float* out, int B, int H, int W, int C, void* stream);
/* layout plumbing for the 4-channel API tensors (reference tensors are NCHW) */
int nd_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int H, int W, void* stream);
int nd_nhwc_to_nchw_f32(const float* in, float* out, int B, int C, int H, int W, void* stream);
/* NCHW (B,C,H,W) -> NHWC with Cpad >= C channels, the tail zero-filled (a 4-channel image as an 8-channel conv input) */
int nd_nchw_to_nhwc_pad_f32(const float* in, float* out, int B, int C, int H, int W, int Cpad, void* stream);

/* ------------------------------------------------------------------ sampler */

/* Device-resident loop state so that one captured step graph can be replayed T times:
 * step counter, current/next timestep table and the per-timestep coefficient table. */
typedef struct nd_sampler_state {
    int32_t* step;          /* [1] device counter, incremented by nd_sampler_advance       */
    const int32_t* t_cur;   /* [n_steps] timestep fed to the network at each step          */
    const int32_t* t_next;  /* [n_steps] DDIM: next timestep (-1 at the end); DDPM unused  */
    const float* coef;      /* [n_steps][8] per-step scalars, see sampler.hip              */
    int64_t* time_out;      /* [B] int64 buffer the network's time embedding reads         */
    const int64_t* rng;     /* [2] device {seed, first_sample} or NULL.  When set it OVERRIDES the seed /
                               first_sample arguments of nd_sampler_step_*: a captured step graph then
                               serves every seed and every rank shard (no re-capture per sample() call) */
    int32_t n_steps, B;
} nd_sampler_state;

/* time_out[b] = t_cur[*step]  (p_sample's batched_times, denoising_diffusion_pytorch.py:369,419) */
int nd_sampler_begin_step(const nd_sampler_state* s, void* stream);
int nd_sampler_advance(const nd_sampler_state* s, void* stream);

/* One reverse-diffusion update on NHWC (B,H,W,C) tensors, in place on x:
 *  DDPM (p_sample :366-373 + p_mean_variance :356-364 + q_posterior :322-329),
 *  DDIM (ddim_sample :418-439, model_predictions :331-354 with clip + rederived eps).
 * objective: 0 pred_noise, 1 pred_x0, 2 pred_v.  noise: explicit draws (parity mode; the draw of
 * step i is at noise + i * noise_step_stride floats, the i-th torch.randn_like of the reference)
 * or NULL => Philox4x32-10 keyed (seed, first_sample + b, step) (throughput mode). */
int nd_sampler_step_ddpm_f32(float* x, const float* model_out, const float* noise, int64_t noise_step_stride,
                             const nd_sampler_state* s, int objective,
                             uint64_t seed, int64_t first_sample,
                             int B, int HW, int C, void* stream);
int nd_sampler_step_ddim_f32(float* x, const float* model_out, const float* noise, int64_t noise_step_stride,
                             const nd_sampler_state* s, int objective,
                             uint64_t seed, int64_t first_sample,
                             int B, int HW, int C, void* stream);
/* x_T ~ N(0,1): torch.randn(shape) (:381,:413) from the same Philox stream (step = -1). */
int nd_philox_normal_f32(float* out, uint64_t seed, int64_t first_sample, int32_t step,
                         int B, int HW, int C, void* stream);

/* ------------------------------------------------------------------ full attention (config 4) */

/* Attention.forward (:255-266) core on MFMA: out = softmax(q k^T / sqrt(dh)) v per (b, head).
 * qkv NHWC [B][N][3*heads*dh] as produced by the to_qkv 1x1 conv ('b (h c) x y' channel order,
 * q | k | v thirds); out [B][N][heads*dh].  dh must be 32. */
int nd_attention_mfma_f32(const float* qkv, int ld_qkv, float* out, int ld_out,
                          int B, int N, int heads, int dh, void* stream);
/* LinearAttention.forward (:218-235) core, O(N dh^2): q softmax over channels (x dh^-1/2), k softmax over pixels,
 * out = (k v^T)^T q per (b, head).  Same qkv / out layout as nd_attention_mfma_f32; `workspace` holds
 * nd_linear_attention_workspace_floats(B, N, heads) floats.  Defined but not wired in the reference net. */
int64_t nd_linear_attention_workspace_floats(int B, int N, int heads);
int nd_linear_attention_f32(const float* qkv, int ld_qkv, float* out, int ld_out, float* workspace,
                            int B, int N, int heads, int dh, void* stream);
/* RMSNorm.forward (:89-90) over channels: out = x / max(||x||, 1e-12) * g * sqrt(C). */
int nd_rmsnorm_nhwc_f32(const float* x, int ldx, const float* g, float* out, int ldo,
                        int B, int HW, int C, void* stream);
/* out = RMSNorm(x) * g + res: LinearAttention's closing RMSNorm (Diffusion_arch.py:213-216) with the residual of the per-stage wiring */
int nd_rmsnorm_add_nhwc_f32(const float* x, int ldx, const float* g, const float* res, int ldr, float* out, int ldo, int B, int HW, int C, void* stream);

/* ------------------------------------------------------------------ optimizer step of the training path
 * torch.optim.Adam's update (the reference's optimizer: models/trainer_diffusion.py:94; L2 weight decay added to the gradient, no amsgrad)
 * for all parameters of a group in ONE launch -- PyTorch's foreach form makes ~10 passes over the parameters.  `items_dev`: n_items records
 * in DEVICE memory, one per parameter: p, m (exp_avg), v (exp_avg_sq) are updated in place from g; step_size = lr / (1 - beta1^t) and
 * bias2_sqrt = sqrt(1 - beta2^t) with the parameter's own step count t (after the increment), computed by the caller in double precision;
 * vec4 != 0 promises that the four tensors are 16-byte aligned.  `chunks_dev`: n_chunks pairs (item index, chunk index) of int32 in device
 * memory, one per nd_adam_chunk_elements() elements of a parameter (the last chunk of a parameter may be short).  noisediff_amd.train.Adam
 * is the host side (a torch.optim.Adam subclass with the same state dict). */
typedef struct nd_adam_item {
    float*       p;
    const float* g;
    float*       m;
    float*       v;
    int64_t      n;
    float        step_size, bias2_sqrt;
    int32_t      vec4, reserved;
    float*       step;        /* capturable form only: the parameter's step counter (one float in device memory) */
} nd_adam_item;
int nd_adam_chunk_elements(void);
int nd_adam_step_f32(const nd_adam_item* items_dev, int n_items, const int32_t* chunks_dev, int n_chunks, float beta1, float beta2, float eps,
                     float weight_decay, void* stream);
/* The same with nothing computed on the host (a training step captured as one graph): item.step points at the parameter's step counter in device
 * memory (a float, as torch.optim.Adam(capturable=True) keeps it); the call adds one to every counter and derives step_size / bias2_sqrt from it
 * (item.step_size, item.bias2_sqrt are ignored).  Two launches. */
int nd_adam_step_capturable_f32(const nd_adam_item* items_dev, int n_items, const int32_t* chunks_dev, int n_chunks, float lr, float beta1, float beta2,
                                float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------ HIP graph helpers */
int nd_stream_create(void** stream);
int nd_stream_destroy(void* stream);
int nd_stream_sync(void* stream);
int nd_graph_begin(void* stream);                 /* hipStreamBeginCapture (thread-local mode) */
int nd_graph_end(void* stream, void** graph_exec);
int nd_graph_launch(void* graph_exec, void* stream);
int nd_graph_destroy(void* graph_exec);
/* nd_graph_begin / _end / _launch make the STREAM's device current for the call (and restore the caller's): one host thread may drive the
 * step graphs of several GPUs round-robin -- GaussianDiffusion.sample under nn.DataParallel(device_ids=[...]), models/modules.py:73-83.
 * nd_stream_device: the device ordinal a stream was created on (-1: unknown). */
int nd_stream_device(void* stream);
/* timing on the library's own stream: HIP events are only meaningful on the stream they are
 * recorded on (torch.cuda.Event sees torch's current stream only). */
int nd_event_create(void** ev);
int nd_event_record(void* ev, void* stream);
int nd_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises `stop` */
int nd_event_destroy(void* ev);
/* cross-stream dependencies (the fork / join edges of the two-branch step graph: the shot-noise branch of NoiseDiffNet.forward,
 * Diffusion_arch.py:598-604, is independent of the U-Net until the final add :644): an event without timing, and "stream waits for event".
 * Recorded / waited on capturing streams they become edges of the captured graph. */
int nd_event_create_untimed(void** ev);
int nd_stream_wait_event(void* stream, void* ev);

#ifdef __cplusplus
}
#endif
#endif /* NOISEDIFF_HIP_H */
