/*
 * motif_hip.h -- C ABI of libmotif_hip.so, the MI355X (gfx950) device library behind the MoTIF
 * C-STVSR inference hot path.  This is the drop-in boundary: every entry point
 * replaces one native interface the reference binds today (cited per function, paths under
 * /root/reference).  The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions (SURVEY.md §8(b) "Native FFI"):
 *   - extern "C", plain pointers and sizes; all tensors fp32, NCHW contiguous unless stated.
 *   - every pointer is a DEVICE pointer; the caller (PyTorch) owns all memory incl. workspaces;
 *     the library never allocates, frees or synchronises.
 *   - `stream` is the hipStream_t to launch on (torch.cuda.current_stream().cuda_stream), the same
 *     raw-pointer + stream contract the reference uses for cupy (models/softsplat_cp.py:240-249).
 *   - return value: 0 = ok, <0 = argument error (MOTIF_E*), >0 = hipError_t of the failed launch.
 *   - no host threads, no global mutable state; re-entrant across streams; one process per GPU.
 */
#ifndef MOTIF_HIP_H
#define MOTIF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOTIF_OK 0
#define MOTIF_EINVAL (-1)   /* bad size / null pointer */
#define MOTIF_ELIMIT (-2)   /* shape outside what the kernel supports */

/* activation codes for fused epilogues */
#define MOTIF_ACT_NONE 0
#define MOTIF_ACT_RELU 1
#define MOTIF_ACT_LRELU 2   /* negative slope 0.1 */
#define MOTIF_ACT_SIGMOID 3
#define MOTIF_ACT_TANH 4

int motif_abi_version(void);                 /* bumps when any signature below changes */
int motif_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len);

/* Tuning / test switches (no reference counterpart; the reference has no kernel-selection knobs).  The library reads the
 * environment ONCE, at its first call (MOTIF_<NAME> in upper case, e.g. MOTIF_CONV_ENGINE=1), never per launch; after that
 * the values change only through motif_set_option.  Every option defaults to 0 = "let the library choose"; none of them
 * changes a result beyond kernel-selection rounding.  Names: conv_dbg, conv_ck, conv_nospec, conv_engine (1 = always the
 * round-2 two-block 3x3 kernel, 2 / 3 / 4 = the round-3 kernel with 12- / 8-row tiles (one workgroup per CU) / 6-row tiles (two per CU) wherever it applies, 5 = the round-4 Winograd F(2,3) kernel (conv_wino.hip) wherever it applies, 6 = never that one (nor conv_pw.hip / conv_ig16.hip: the fp32 engine for their layers), 7 = conv_ig16.hip (round 6: stride 2, 7x7, dilated, narrow / wide layers on the fp16 matrix cores) wherever a shape FITS it instead of only where it pays, 0 = the library's choice: the Winograd kernel for every eligible 3x3 layer with a plain epilogue, else by tile count), lds_pad, corr81 (1 tiled / 2 small / 3 tiled with nine waves per block),
 * dcn_nowin, dcn_waves (4 / 8), dcn_front_pad, dcn_back_pad, siren_stagger, conv_novec (1 = 4-byte staging in the fp32 engine), conv_nodirect (1 = the narrow layers on large maps stay on the MFMA engine instead of conv_direct.hip), conv_wino_tr (1 = the Winograd kernel's experimental form with transposed accumulators and a register-only epilogue wherever the activation is uniform over the couts: bit-identical, measured no faster; 0 = row-major accumulators, LDS-transposed epilogue), conv_wino_rpre (1 = the Winograd kernel requests all residual quads in the epilogue, as in round 4; 0 = the first two passes' quads under the tile's last chunk), conv_chain_wgs (workgroups of a motif_conv2d_chain_fwd launch; 0 = one per CU), resize_narrow (1 = motif_resize_bilinear keeps the 4-column form for x2 upsampling; 0 = 8 columns x 2 rows per thread, bit-identical), conv_direct_quads (1 = the deep direct form keeps quad-indexed workgroups with loaded edge pixels on small maps; 0 = whole rows per workgroup and shuffled edge pixels, bit-identical).  Returns MOTIF_EINVAL for an unknown name.
 * Not thread-safe against concurrent launches -- a test / tuning aid, not part of the data path. */
int motif_set_option(const char* name, int value);
int motif_get_option(const char* name, int* value);

/* Range status word (no reference counterpart: the reference is fp32 end to end, models/modules/Ours.py:419, and asserts where it
 * matters, models/softsplat_cp.py:25-26).  The two-part fp16 arithmetic (MotifConvDesc.mma = 7, `pre` = 3, DCN mma = 7) has fp16's
 * operand range: an activation, a transformed activation (up to 2 |x| in the Winograd kernel) or 2^8 x a weight of 65520 or more is
 * packed as inf, and inf times any weight part -- zero included -- is non-finite, so EVERY output of the affected pixel comes out
 * inf / NaN.  Downstream stages can launder such values (the fused splat clamps its plane values and drops sources with a
 * non-finite flow), so the kernels that run the form report it at the source instead: `status` is a caller-owned device uint32
 * (NULL = no report) into which they atomically OR bit 0 when an accumulator is non-finite.  The word is sticky -- the caller
 * zeroes it before a clip and reads it after (one 4-byte copy); a set bit means "render this clip again with mma = 6", which has
 * fp32's exponent range.  Small magnitudes need no report: the low activation part is stored times 2^11 (see mma = 7 below), so the
 * form is fp32-equivalent for tensors whose magnitude is anywhere from ~3e-5 to 3e4 (tests/test_kernels_gpu.py sweeps 1 .. 1e-4)
 * and degrades gracefully below that (absolute operand error <= 2^-36). */

/* ------------------------------------------------------------------------------------------------
 * A1-A3  fused soft-splat forward.
 * Replaces the three cupy launches of kernel_Softsplat_updateOutput:
 *   models/softsplat_cp.py:12-52,221-258 (sum of [feat*e^z, e^z], via FunctionSoftsplat :320-347),
 *   models/softsplat_max_cp.py:12-58,254 (max of e^z*w, output initialised to 1),
 *   models/softsplat_count_cp.py:14-52,163-165 (unweighted +1 per touched corner).
 * Plain operator form (reference module surface): inputs NCHW, flow [N,2,H,W], metric z [N,1,H,W]
 * (may be NULL -> only `cnt`/`mx` on `src`), outputs caller-initialised (sum/cnt: zeros, mx: ones),
 * any output pointer may be NULL.  out_sum has C channels, out_norm 1 channel.
 * ---------------------------------------------------------------------------------------------- */
int motif_splat_fwd(const float* src, const float* flow, const float* z,
                    float* out_sum, float* out_norm, float* out_max, float* out_cnt,
                    int N, int C, int H, int W, void* stream);

/* Fused MoTIF form (Ours.py:777-816): sources are never materialised.  Per direction d (0,1),
 * batch b, timestamp n the source stack is [imnet_out(64) | pred flow(2, raw) | feat_low(64 gathered
 * from the LR encoder feature by the nearest tables)], flow = pred[0:2]*flow_scale,
 * z = relu(pred[2])*alpha.  Both directions accumulate into the same accumulator
 *   acc [B*N, 133, HH, WW]: planes 0..129 feature sums, 130 sum of e^z*w, 131 max (init 1), 132 count.
 * imnet_out [2B,64,Q], pred [2B*N,3,Q], feat_lr [2B,64,H,W], iy[HH], ix[WW] int32 tables.
 * row0: 0 for a whole image; when the tensors hold the HR row band [row0, row0+HH) of a taller image (spatial
 * tiling), the image row of local row 0 -- coordinates are then formed with image rows, so corner indices and
 * weights are bit-identical to the untiled call. */
int motif_splat_motif_fwd(const float* imnet_out, const float* pred, const float* feat_lr,
                          const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                          float* acc, int B, int N, int H, int W, int HH, int WW, int row0, void* stream);
/* The same with `accumulate`: 0 = acc is written (as above); 1 = this call's two directions are added to an accumulator an
 * earlier call wrote (sums and count added, max plane max-ed).  The 4-source generator sums four directions
 * (Ours_44.py:713-719): two calls, the second with accumulate = 1. */
int motif_splat_motif_acc_fwd(const float* imnet_out, const float* pred, const float* feat_lr,
                              const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                              float* acc, int B, int N, int H, int W, int HH, int WW, int row0, int accumulate, void* stream);
/* Pre-contracted form (same result, half the accumulator).  The splat is linear in its sources and synth_net's first
 * layer is linear in the normalised splat (Ours.py:811-814, 839-856), so W0[:, 0:130] is applied BEFORE the splat:
 *   u_hr  [2B,64,Q]   = imnet with its head composed with W0[:, 0:64]  (motif_siren_imnet_fwd on composed weights),
 *   g_lr  [2B,64,H,W] = W0[:, 66:130] . encoder feature (a 1x1 convolution at LR; gathered by the nearest tables here),
 *   ab    [2,64]      = W0[:, 64], W0[:, 65] (the raw predicted-flow channels, Ours.py:789),
 * a source carries u + g + a*p0 + b*p1.   acc [B*N, 67, HH, WW]: planes 0..63 sums, 64 sum of e^z*w, 65 max (init 1),
 * 66 count; consumed by motif_siren_synth_pre_fwd.
 * g_lr = NULL: u_hr already holds u + g (motif_siren_imnet_add_fwd added the gathered LR term when it stored u -- the
 * same fp32 sum, so the result is bit-identical); the kernel then reads one value per source and plane, which is the form
 * its plane pipeline is sized for (a third faster than with g_lr).  Plane values are clamped to +-2^17 (see splat.hip). */
int motif_splat_motif_pre_fwd(const float* u_hr, const float* pred, const float* g_lr, const float* ab,
                              const int32_t* iy, const int32_t* ix, const float* alpha, float flow_scale,
                              float* acc, int B, int N, int H, int W, int HH, int WW, int row0, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * B1-B4  space-time local implicit MLPs (SIREN, omega0=30) with the nearest gather fused in.
 * Replaces torch's grid_sample(nearest)+cat+Linear+sin chains at Ours.py:699-737 and 839-858 over
 * models/modules/SIREN.py:44-45,77-79.  Weights are passed PACKED (motif_siren_pack).
 * ---------------------------------------------------------------------------------------------- */
/* n_layers linear layers: dims[0]=in, dims[1..n_layers]=out of each layer (last is the linear head).
 * w[i] is [dims[i+1], dims[i]] row-major (nn.Linear.weight), b[i] is [dims[i+1]].
 * Returns number of floats the packed blob needs when `packed` is NULL. */
long motif_siren_pack(const float* const* w, const float* const* b, const int* dims, int n_layers,
                      float* packed, void* stream);

/* The same three networks (mode 0 imnet 66-64-64-256-64, 1 flow_imnet 67-64-64-256-3, 2 synth_net
 * 198-64-64-64-256-3) packed for the bf16 matrix cores: every weight split into three bf16 parts (fp32-equivalent,
 * see MotifConvDesc.mma = 6), stored as MFMA fragments.  Pass the blob with pre = 2.  Size query with packed = NULL.
 * mode + 8: the two-part fp16 form (every weight x 2^8 split into two fp16 parts, three products per fp32 MAC instead of six,
 * MotifConvDesc.mma = 7; hidden activations are sines, so both operands suit fp16's range); pass that blob with pre = 3. */
long motif_siren_pack_split(int mode, const float* const* w, const float* const* b, float* packed, void* stream);

/* `pre` (all three): 0 = `*_lr` holds the raw LR feature (64 ch) and the whole first layer runs per HR pixel;
 * 1 = `*_lr` holds the LR-resolution partial pre-activation W0[:, gathered 64 channels] . feature + b0 (a 1x1
 * convolution done once per clip -- it depends neither on the HR pixel nor on t), which seeds the accumulator;
 * 2 = as 1, with `packed` from motif_siren_pack_split: the contractions run as 6 bf16 products per fp32 MAC;
 * 3 = as 2, with `packed` from motif_siren_pack_split(mode + 8): 3 fp16 products per fp32 MAC.
 * imnet: in = [feat_lr[d*B+b](64 gathered) | rel_y | rel_x] -> out [2B,64,Q] planar. */
int motif_siren_imnet_fwd(const float* packed, const float* feat_lr, const int32_t* iy, const int32_t* ix,
                          const float* rel_y, const float* rel_x, float* out,
                          int B2, int H, int W, int HH, int WW, int pre, void* stream);
/* motif_siren_imnet_fwd with an LR tensor add_lr [2B,64,H,W] gathered through the same tables and ADDED to the 64 output planes
 * (fp32 add after the head): the pre-contracted splat's G term (see motif_splat_motif_pre_fwd, g_lr = NULL), which costs the
 * splat one load per source and plane instead of two.  pre must be 2 or 3; add_lr = NULL is motif_siren_imnet_fwd. */
int motif_siren_imnet_add_fwd(const float* packed, const float* feat_lr, const float* add_lr, const int32_t* iy, const int32_t* ix,
                              const float* rel_y, const float* rel_x, float* out,
                              int B2, int H, int W, int HH, int WW, int pre, void* stream);
/* flow_imnet: in = [flow_feat_lr(64 gathered) | t | rel_y | rel_x] -> pred [B2*N,3,Q] planar,
 * image index i = b2*N + n, t taken from times[(i) % (B*N)] laid out [B,N] (Ours.py:727-733). */
int motif_siren_flow_fwd(const float* packed, const float* flowfeat_lr, const int32_t* iy, const int32_t* ix,
                         const float* rel_y, const float* rel_x, const float* times, float* pred,
                         int B2, int N, int H, int W, int HH, int WW, int pre, void* stream);
/* synth: post-splat normalise (Ours.py:811-836) + [out(130) | extra(3) | residual_lr(64 gathered) | t]
 * -> synth_net -> clamp(0,1) -> frames [N,B,3,HH,WW].  acc as produced by motif_splat_motif_fwd. */
int motif_siren_synth_fwd(const float* packed, const float* acc, const float* residual_lr,
                          const int32_t* iy, const int32_t* ix, const float* times, float* frames,
                          int B, int N, int H, int W, int HH, int WW, int pre, uint32_t* status, void* stream);
/* status (both synth forms; may be NULL): range status word, see above.  Only the first layer of synth_net reads data of
 * unbounded magnitude (the splat's max plane, its normalised sums); imnet / flow_imnet read coordinates, t and fp32 LR partials,
 * and every hidden activation is a sine -- they cannot leave fp16's range and take no status argument. */
/* synth on the pre-contracted accumulator of motif_splat_motif_pre_fwd: pre-activation of the first layer =
 * residual_l0 (the LR partial W0[:, 133:197] . residual + b0, as with pre = 1/2) + acc[0:64] / warped_z +
 * W0[:, 130:133] . extra + W0[:, 197] t, with the exact-equality patches of Ours.py:811-830 on warped_z / count;
 * `packed` from motif_siren_pack_split(kind = 3 [+ 8]).  Split arithmetic only: pre = 2 (three bf16 parts) or 3 (two fp16 parts). */
int motif_siren_synth_pre_fwd(const float* packed, const float* acc67, const float* residual_l0,
                              const int32_t* iy, const int32_t* ix, const float* times, float* frames,
                              int B, int N, int H, int W, int HH, int WW, int pre, uint32_t* status, void* stream);
/* debugging/parity aid: materialise the 198-channel synth input [B*N,198,HH,WW] (Ours.py:839-844). */
int motif_synth_input_fwd(const float* acc, const float* residual_lr, const int32_t* iy, const int32_t* ix,
                          const float* times, float* out, int B, int N, int H, int W, int HH, int WW, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense convolution engine: fp32 MFMA implicit GEMM (D1, C1, C3, E1 contractions).
 * Replaces cuDNN conv2d as reached through torch.nn.Conv2d in Ours.py / models/core / OpticalFlow.
 * ---------------------------------------------------------------------------------------------- */
typedef struct MotifConvDesc {
    int N, H, W;              /* input batch and spatial size */
    int C0, C1;               /* channels taken from in0 and in1 (in1 may be NULL, C1=0): fused concat */
    int Cout, KH, KW;
    int stride, pad, dil, groups;
    int pad_mode;             /* 0 zeros, 1 reflect */
    int act, act2, act_split; /* channels >= act_split use act2 (act_split<=0: all use act) */
    int res_mode;             /* 0 none; 1 out=act(acc+res); 2 out=act(acc)+res; 3 out=relu(act(acc)+res);
                                 4 out=act(acc)*res */
    long in0_bs, in1_bs, res_bs, out_bs;  /* batch strides in elements (0 -> dense default) */
    int mma;                  /* arithmetic of the contraction: 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32);
                                 6 = fp32-equivalent on the bf16 matrix cores (3-way bf16 split, 6 products, fp32
                                 accumulate); 3 = 2-way split, 3 products (16-bit mantissa); 1 = plain bf16.
                                 Non-zero values apply to 3x3/stride-1 layers with >= 16 input channels per group and
                                 > 32 couts per group (or 17 .. 32 couts when the group has >= 128 input channels: half of
                                 a workgroup's matrix work is then spent on zero weights, still well ahead of mode 0 on
                                 such long reductions); every other layer runs mode 0.  The packed weight format
                                 depends on it: pack and forward must be given the same value.  With mode 6 the blob of
                                 such a layer holds two fragment blocks, direct and Winograd F(2,3)-along-the-rows
                                 (U = G g, formed in fp64 and split from there); which kernel runs is decided per launch
                                 (layout, alignment, epilogue: option conv_engine).  The Winograd form has the same
                                 6-product arithmetic with 2/3 of the matrix instructions; its rounding differs from the
                                 direct form by one fp32 addition per operand and a three-term output sum (measured
                                 against fp64: not larger than the direct form's error).
                                 7 = as 6, except that the Winograd block and kernel use the TWO-PART fp16 form: every
                                 operand = hi + lo, hi = rne_fp16(x), three products per fp32 MAC instead of six.  Weights:
                                 lo = rne_fp16(2^8 w - hi), packed times 2^8 (so that the low parts of everyday weights,
                                 |w| >= 5e-4, are normal numbers); the epilogue multiplies by 2^-8; both exact.  Activations
                                 (since ABI 7): the low part is stored times 2^11, lo_s = rne_fp16((x - hi) * 2^11) -- a
                                 number of hi's own magnitude, hence normal whenever hi is -- and multiplied by 2^-11 x the
                                 high weight part (exact while normal): |x - hi - 2^-11 lo_s| <= 2^-22 |x| for every
                                 |x| >= 2^-14, an absolute 2^-36 below that.  Domain: fp32-equivalent (error at or below an
                                 fp32 MFMA's, tests/test_kernels_gpu.py: activation scales 1 .. 1e-4, weight scales 1/24,
                                 1e-2) for tensors of magnitude ~3e-5 .. 3e4.  Above: |2 x| or 2^8 |1.5 w| >= 65520 gives
                                 inf / NaN in every cout of the pixel, never a clamped value, AND sets `status` (below).
                                 Which layers run the form: 3x3 / stride-1 / zero-pad layers the Winograd kernel takes
                                 (conv_wino.hip: > 16 input channels per group, W % 4 == 0, aligned tensors) and 1x1 /
                                 stride-1 / groups-1 layers with 17 .. 128 couts and >= 8 input channels (conv_pw.hip, same
                                 arithmetic, reads the fp32 blob), and -- round 6, conv_ig16.hip -- the remaining layers
                                 with >= 24 input channels per group that are strided, dilated or wide (the PCD / RAFT /
                                 PWC stride-2 pyramids, PWC-Net's dilated refiner, 64 -> 32 layers ...: an implicit GEMM on
                                 the fp16 cores whose A fragments are a second block of the packed blob); option
                                 conv_engine = 6 disables all three.  Every other layer of a mode-7 network runs as mode 6
                                 (3x3 split-eligible) or mode 0. */
    uint32_t* status;         /* device word (may be NULL), see "Range status word" below: kernels running the two-part fp16
                                 form OR bit 0 into it when an accumulator comes out non-finite (an operand beyond fp16's range) */
} MotifConvDesc;

long motif_conv2d_packed_size(const MotifConvDesc* d);
int motif_conv2d_pack(const MotifConvDesc* d, const float* weight /*[Cout,Cin/groups,KH,KW]*/,
                      float* packed, void* stream);
int motif_conv2d_fwd(const MotifConvDesc* d, const float* in0, const float* in1, const float* packed,
                     const float* bias /*may be NULL*/, const float* res /*may be NULL*/, float* out, void* stream);
/* P (<= 4) independent convolutions of identical shape/hyper-parameters in ONE launch (the two alignment
 * directions of PCD_Align and the h/c branches of the deformable ConvLSTM, Ours.py:107-172,289-290, are such
 * sets): per-problem pointer arrays and batch strides (0 -> dense).  The desc's *_bs fields are ignored. */
int motif_conv2d_fwd_multi(const MotifConvDesc* d, int P, const float* const* in0, const float* const* in1,
                           const float* const* packed, const float* const* bias, const float* const* res,
                           float* const* out, const long* in0_bs, const long* in1_bs, const long* res_bs,
                           const long* out_bs, void* stream);

/* A CHAIN of L dependent 3x3 / stride-1 / zero-pad-1 convolutions C -> C of one shape in ONE persistent launch (ABI 8): what
 * nn.Sequential(ResidualBlock_noBN x 40) -- the reconstruction trunk, models/modules/module_util.py:34-52 as built at
 * models/modules/Ours.py:349 and run at Ours.py:393-409 -- and the 5-block feature extraction (Ours.py:352-356, 368-370)
 * are: x' = x + conv2(relu(conv1(x))).  Same arithmetic and the same bits as L calls of motif_conv2d_fwd (the per-tile
 * computation is that kernel's; tests compare with torch.equal), without L launch boundaries, with one kernel prologue per
 * workgroup instead of one per layer and no idle tail per layer: the tiles of all layers are handed out in layer-major order,
 * a tile starts when the <= 9 tiles of the layer before that it reads have been published (conv_wino.hip, CHAIN).
 *   d        shape and arithmetic of EVERY layer: mma = 7, groups 1, C1 = 0, 49 <= C0 = Cout <= 64, W % 4 == 0; d->act /
 *            res_mode are ignored (per layer below); d->in0_bs / out_bs = batch strides of x / out (0 = dense); d->status as usual,
 *            bit 1 = NO tile of the chain was published for a second (a stalled device; a workgroup that merely waits long while
 *            other workgroups advance does not give up) and the launch was abandoned: results invalid; never seen in operation.
 *            bit 2 = the layer table names a buffer the call does not provide (see `work_floats`), writes x, or has no source:
 *            nothing was read or written.  With d->status = NULL either condition TRAPS the launch (the stream reports a
 *            launch failure at its next synchronisation): an abandoned chain never looks like a finished one.
 *   layers   DEVICE array of L entries.  Buffer ids: 0 = x (never written), 1 = out, 2 + i = scratch buffer i =
 *            work + i * N * C * H * W (dense NCHW), valid for i < work_floats / (N * C * H * W) -- checked on the device
 *            (the table is a device array) by a one-block kernel in front of the launch.  The caller orders the buffers so that a layer overwrites a buffer only
 *            when all its readers are producers-of-producers of the writing tile; the rotation of a residual trunk
 *            (conv1: X_b -> T; conv2: T (+ X_b) -> X_b+1, X alternating between two buffers) satisfies it with three buffers.
 *   ws       device scratch of motif_conv2d_chain_ws_words(d, L) 32-bit words (<= 0: this shape cannot run as a chain:
 *            call the layers one by one); zeroed by the entry on `stream` before the launch.  Word 0 = ticket counter, 1 = abort
 *            word, 2 = tiles published so far, 64.. = completion counters.
 * 16-byte aligned pointers; all on `stream`. */
typedef struct MotifChainLayer {
    const float* packed;      /* motif_conv2d_pack blob of the layer (same desc but for the epilogue fields) */
    const float* bias;        /* [C] or NULL */
    int32_t src, dst, res;    /* buffer ids; res = -1: no residual */
    int32_t act_rm;           /* MOTIF_ACT_* | residual mode << 8 (modes as MotifConvDesc.res_mode) */
} MotifChainLayer;
long motif_conv2d_chain_ws_words(const MotifConvDesc* d, int L);
int motif_conv2d_chain_fwd(const MotifConvDesc* d, int L, const MotifChainLayer* layers /*device*/, const float* x, float* out,
                           float* work, long work_floats, uint32_t* ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * D2  modulated deformable convolution v2, forward.
 * Replaces _ext.dcn_v2_forward (models/modules/DCNv2/src/dcn_v2.h:9-39, cuda/dcn_v2_cuda.cu:42-171,
 * cuda/dcn_v2_im2col_cuda.cu:125-194).  offset [B,2*dg*kh*kw,Ho,Wo] ((dy,dx) interleaved per tap per
 * group), mask [B,dg*kh*kw,Ho,Wo] (already sigmoid'ed), `packed` = motif_conv2d_pack of the weight
 * viewed as a 1x1 conv over C*kh*kw channels, `columns` workspace of B*C*kh*kw*Ho*Wo floats.
 * offset_bs/mask_bs: batch strides (elements) so both can alias one conv_offset_mask output. */
int motif_dcn_v2_fwd(const float* input, const float* offset, const float* mask, const float* packed,
                     const float* bias, float* columns, float* out,
                     int B, int C, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil,
                     int deformable_groups, long offset_bs, long mask_bs, int act, void* stream);
/* P (<= 4) independent DCNs of identical shape in one launch pair; `columns` holds P*B*C*kh*kw*Ho*Wo floats;
 * input_bs: per-problem batch stride of `input` (0 -> dense). */
int motif_dcn_v2_fwd_multi(int P, const float* const* input, const long* input_bs, const float* const* offset,
                           const float* const* mask, const float* const* packed, const float* const* bias,
                           float* columns, float* const* out, int B, int C, int H, int W, int Cout, int kh, int kw,
                           int stride, int pad, int dil, int deformable_groups, long offset_bs, long mask_bs,
                           int act, void* stream);

/* Fused form for the configuration MoTIF uses (3x3, stride 1, pad 1, dilation 1, Ours.py:65-94): the deformable
 * im2col goes straight into LDS and is consumed by the MFMA loop -- no `columns` workspace.  `packed3x3` =
 * motif_conv2d_pack of the [Cout,C,3,3] weight with a 3x3 descriptor.  Returns MOTIF_ELIMIT for other shapes. */
int motif_dcn_v2_fused_fwd_multi(int P, const float* const* input, const long* input_bs, const float* const* offset,
                                 const float* const* mask, const float* const* packed3x3, const float* const* bias,
                                 float* const* out, int B, int C, int H, int W, int Cout, int deformable_groups,
                                 long offset_bs, long mask_bs, int act, int mma, uint32_t* status, void* stream);
/* mma = 0: `packed3x3` as above, fp32 MFMA.  mma = 6: the GEMM on the bf16 matrix cores with the fp32-equivalent 3-way split
 * (MotifConvDesc.mma), `packed3x3` from motif_dcn_split_pack (size query with packed = NULL; returns floats).  mma = 7: the two-part
 * fp16 form (three products, weights x 2^8, low activation part x 2^11: MotifConvDesc.mma = 7) in the window kernel, mma = 6 where that
 * kernel does not apply; the blob of motif_dcn_split_pack holds both fragment blocks.  status: range status word (may be NULL). */
long motif_dcn_split_pack(const float* weight /*[Cout,C,3,3]*/, float* packed, int Cout, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * C2  RAFT windowed correlation lookup.  Replaces alt_cuda_corr.forward (third-party, not vendored;
 * call site models/core/corr.py:78-83).  fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] channels-last,
 * coords [B,2,H1,W1] (x,y planes; the caller passes coords/2^level through `coord_scale`),
 * out channel (ix*(2r+1)+iy) written at out[b, ch_off + ., h, w] of a [B,out_C,H1,W1] tensor,
 * every value divided by `div` (corr.py:87). */
int motif_raft_corr_lookup(const float* fmap1, const float* fmap2, const float* coords, float coord_scale,
                           float* out, int B, int H1, int W1, int H2, int W2, int C, int r,
                           int out_C, int ch_off, float div, void* stream);

/* all (<= 4) pyramid levels of AlternateCorrBlock.__call__ (corr.py:70-87) in one launch: level i samples
 * fmap2[i] [B,H2[i],W2[i],C] at coords/2^i and writes channels [49*i, 49*i+49) of out [B,out_C,H1,W1]. */
int motif_raft_corr_lookup_pyramid(const float* fmap1, const float* const* fmap2, const int* H2, const int* W2, int levels,
                                   const float* coords, float* out, int B, int H1, int W1, int C, int r,
                                   int out_C, float div, const int32_t* index1_host, const int32_t* index2_host, void* stream);
/* index1_host / index2_host (HOST int32 [B], may be NULL = identity): pair b correlates fmap1[index1[b]] with
 * fmap2[.][index2[b]] -- MoTIF feeds the pairs (a,b) and (b,a) of one frame stack (Ours.py:544), so the encoders run once per frame
 * and the pairing is an index, not a gathered copy of five feature maps. */

/* C4  PWC-Net 9x9 cost volume.  Replaces kernel_Correlation_rearrange + kernel_Correlation_updateOutput
 * (OpticalFlow/correlation.py:17-112,294-348): out[b,(dy+4)*9+(dx+4),y,x] = mean_c f1*f2(y+dy,x+dx). */
int motif_corr81_fwd(const float* first, const float* second, float* out, int B, int C, int H, int W,
                     int act, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pointwise / resampling kernels of the path (E1, F1 and RAFT/encoder glue).
 * ---------------------------------------------------------------------------------------------- */
/* F.interpolate(mode='bilinear') (Ours.py:540,548; PCD_Align 123-167; utils.py:80-82), out *= mul.
 * post: 0 = nothing more; 1 = the result is additionally normalised as RAFT's input, 2 * ((v * 255) / 255) - 1 with the roundings of the
 * reference's four element-wise operations (Ours.py:544 `* 255`, raft.py:90-91) -- the HR frames feed nothing else. */
int motif_resize_bilinear(const float* in, float* out, int NC, int H, int W, int Ho, int Wo,
                          int align_corners, float mul, int post, void* stream);
/* BackWarp.forward (Ours.py:899-923): (x/w)*2-1 grid, grid_sample(bilinear, align_corners=True, border);
 * sign multiplies `img` (the -flow case at Ours.py:565). */
int motif_backwarp(const float* img, const float* flow, float* out, int N, int C, int H, int W, float sign, void* stream);
/* PWC Decoder.Backward (PWCNet.py:146-177): linspace grid + flow/((size-1)/2), grid_sample default
 * (align_corners=False, zeros) and the >0.999 validity mask. */
int motif_pwc_backward_warp(const float* img, const float* flow, const float* gx_table /*linspace(-1,1,W)*/,
                            const float* gy_table /*linspace(-1,1,H)*/, float* out, int N, int C, int H, int W, void* stream);
/* reliability maps (Ours.py:562-578) + flow encoder input (Ours.py:614-631):
 * fr0/fr1: the centre pair, each [3,H,W] per batch item with batch stride fr_bs (elements);
 * flow [4B,2,H,W] (pairs 00,01,10,11; 00 and 11 already zeroed) -> psies [4B,3,H,W], flow_feat [2B,14,H,W]. */
int motif_reliability_fwd(const float* fr0, const float* fr1, long fr_bs, const float* flow, const float* g_filter,
                          float* psies, float* flow_feat, int B, int H, int W, void* stream);
/* the same maps for the 4-frame generators (Ours_4.py:514-592: 8 flows from frames 1,2; Ours_44.py:519-593: 16 flows, all
 * ordered pairs), table driven: `table` is a HOST int32 [J][4] = (source frame, target frame, index of the flow in `flow`, index of
 * its reverse flow), `durations` a HOST float [J][2] (ref_start_durations already divided); frames [B][n][3,H,W] with the given
 * element strides; flow [F*B,2,H,W]; S flows per source frame -> psies [J*B,3,H,W], flow_feat [(J/S)*B, S*7, H, W].  J <= 16. */
int motif_reliability_pairs_fwd(const float* frames, long frame_stride, long batch_stride, const float* flow,
                                const float* g_filter, const int32_t* table_host, const float* durations_host, int J, int S,
                                float* psies, float* flow_feat, int B, int H, int W, void* stream);
/* InstanceNorm2d (eps 1e-5, no affine) + optional relu, optional residual: out = relu?(res + relu?(norm(x)))
 * mode 0: norm; 1: relu(norm); 2: relu(res + relu(norm))   (models/core/extractor.py:60-116,246-248) */
int motif_instance_norm(const float* x, const float* res, float* out, int NC, int HW, int mode, void* stream);
/* same, with a caller-owned fp64 workspace of NC*(2+128) doubles: large planes are split over many workgroups, both moments
 * in one pass over the tensor (fp64 sum and sum of squares) */
int motif_instance_norm_ws(const float* x, const float* res, float* out, double* workspace, int NC, int HW, int mode, void* stream);
/* InstanceNorm2d(C, affine=True): y = (x - mean) / sqrt(var + 1e-5) * gamma[c] + beta[c] per (image, channel) plane of x [N,C,HW].
 * Replaces torch.nn.InstanceNorm2d(3, affine=True) at OpticalFlow/PWCNet_light.py:18 (applied at :259-260).  workspace as
 * motif_instance_norm_ws (N*C*(2+128) doubles; NULL = one block per plane).  (ABI 9) */
int motif_instance_norm_affine_ws(const float* x, const float* gamma, const float* beta, float* out, double* workspace,
                                  int N, int C, int HW, void* stream);
/* InstanceNorm whose statistics span several processes (one clip tiled over GPUs, SURVEY.md 8(e) row 3: "all_reduce of per-channel
 * sum x, sum x^2 for each of the 21 InstanceNorm layers of fnet", extractor.py:85-90,206-207): moments over the rows [row_lo,row_hi) a rank
 * owns -> sums [NC][3] doubles (sum, sum of squares, element count); the caller all-reduces them (SUM); apply normalises the whole
 * plane with them (mode as motif_instance_norm). */
int motif_instance_norm_moments(const float* x, double* sums, int NC, int H, int W, int row_lo, int row_hi, void* stream);
int motif_instance_norm_apply(const float* x, const float* res, const double* sums, float* out, int NC, int HW, int mode, void* stream);
int motif_avg_pool2(const float* in, float* out, int NC, int H, int W, void* stream);   /* F.avg_pool2d(x,2,2) */
int motif_nchw_to_nhwc(const float* in, float* out, int N, int C, int HW, void* stream);
/* ConvGRU update (update.py:24-31): h' = (1-z)*h + z*q */
int motif_gru_update(const float* z, const float* q, const float* h, float* out, long n, void* stream);
/* ConvLSTM gates (convlstm.py:49-58): cc [B,4*hid,H,W] (i,f,o,g pre-activations) */
int motif_lstm_gates(const float* cc, const float* c_cur, float* h_next, float* c_next, int B, int hid, int HW, void* stream);
/* out = a*x + b*y (y may be NULL) */
int motif_axpby(const float* x, const float* y, float a, float b, float* out, long n, void* stream);
/* the same for B items of n elements, item i written at out + i * out_bs (a channel slice of a wider tensor) */
int motif_axpby_bs(const float* x, const float* y, float a, float b, float* out, int B, long n, long out_bs, void* stream);
/* The flow LunaTokis.forward returns (Ours.py:794 scales the prediction up, (p*20)*ratio, :858 scales it back, /20 /ratio):
 * pred [N,3,Q] -> out [N,2,Q] = (((p*a)*b)/a)/b with the four roundings of the four torch operations. */
int motif_flow_roundtrip(const float* pred, float* out, int N, long Q, float a, float b, void* stream);
/* ConvTranspose2d(k=4,s=2,p=1) with few output channels (PWCNet.py:102-106) */
int motif_deconv4x4s2(const float* in, const float* weight /*[Cin,Cout,4,4]*/, const float* bias, float* out,
                      int N, int Cin, int Cout, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Frame formats either side of the path.
 * decode: uint8 interleaved [N,H,W,3] (cv2.imread order, BGR) -> fp32 planar [N,3,H,W] in [0,1]; swap_rb = 1 also turns
 *   BGR into RGB -- `img.astype(np.float32) / 255.`, `[:, :, :, [2, 1, 0]]`, HWC->CHW of data/Adobe_test_3.py:171-195.
 * encode: fp32 planar [N,3,H,W] -> uint8 interleaved [N,H,W,3]: clamp(0,1), *255, then round_mode 1 = round half to even
 *   (utils/util.py:105-129 tensor2img, swap_rb = 1 for its RGB->BGR) or 0 = truncate (demo.py:94-99, swap_rb = 0).
 * ---------------------------------------------------------------------------------------------- */
int motif_frames_u8_to_f32(const unsigned char* in, float* out, int N, int H, int W, int swap_rb, void* stream);
int motif_frames_f32_to_u8(const float* in, unsigned char* out, int N, int H, int W, int round_mode, int swap_rb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOTIF_HIP_H */
