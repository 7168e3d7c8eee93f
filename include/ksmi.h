/* libksmi — C-ABI of the MI355X-native (gfx950) Kuro Siwo training hot path.
 *
 * The reference (Orion-AI-Lab/KuroSiwo) is pure Python on torch.nn and has no FFI
 * layer; each entry point below replaces the ATen/cuDNN op(s) behind one row of
 * SURVEY.md §8(a) and cites the reference file:line whose computation it takes over.
 * A reference-side binding (ctypes stub) for every group is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by
 *     the caller (PyTorch allocator); kernels never allocate or free.
 *   - every launch goes to the `stream` argument (hipStream_t passed as void*).
 *   - return 0 on success, a positive hipError_t, or a negative KSMI_E_* code;
 *     ksmi_last_error() returns a thread-local message.  Launches are asynchronous.
 *   - dtype codes: KSMI_F32 = 0 (parity mode), KSMI_BF16 = 1 (performance mode).
 *     Activations are NHWC in `dtype`; parameters, gradients, statistics are fp32.
 *   - boundary tensors keep the reference layout: images NCHW fp32, labels int64
 *     [B,H,W], logits NCHW fp32.
 */
#ifndef KSMI_H
#define KSMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSMI_ABI_VERSION 7   /* 7: round 6 (ksmi_set_knob, ksmi_conv_dispatch_info, ksmi_argmax_confusion_grouped, ksmi_run_list, fused SR-attention block, stream-K token GEMMs); 6: round 5 (ksmi_adam_step_mirror, ksmi_maxpool3x3s2_forward_idx / _backward_idx, ksmi_conv_wgrad_fuses_bias == 2: partial bias rows in ksmi_wgrad_desc.bias_grad, hbm probe window bits); 5: round 4 (BatchNorm statistics finished inside the consuming pass: ksmi_bn_fin_*, ksmi_bn*_bwd_fin_*); 2: round 2 (stats_rows, stochastic layers, bias_grad in ksmi_wgrad_desc, ...); 3: round 3 (gate epilogue, ksmi_desc_size); 4: first conv on raw tiles, tile reader, BIT token path */
#define KSMI_F32 0
#define KSMI_BF16 1
#define KSMI_E_ARG (-1)
#define KSMI_E_UNSUPPORTED (-2)
#define KSMI_MAX_SRC 6
#define KSMI_MAX_CHUNKS 72

int ksmi_abi_version(void);
const char* ksmi_last_error(void);
/* Test / probe hook (ABI 7).  Every launcher switch goes through ONE registry (csrc/api.hip: ksmi_knob_int / _str / _is_set; no other
 * translation unit calls getenv).  The launchers read their switches (KSMI_* environment variables) ONCE: the start-up switches at their first
 * use, the run-time knobs -- KSMI_IGEMM3_CUS, KSMI_IGEMM4_CUS, KSMI_IGEMM4_VAR, KSMI_IG4_PATCH, KSMI_IG3_DBG, KSMI_IG4_DBG, KSMI_WGRAD3_NST,
 * KSMI_WGRAD3_WGS: grid shrinkers, forced tile variants and profiling switches of the tests and probes -- at their first look-up, after
 * which only this call changes them (value NULL: back to the built-in default).  No launch path calls getenv per launch.  Not part of
 * the reference's operator surface; production code never calls it. */
int ksmi_set_knob(const char* name, const char* value);
/* bytes of K per packed k-chunk / elements per chunk for a dtype (32 bf16, 16 fp32) */
int ksmi_chunk_elems(int dtype);

/* ---------------------------------------------------------------------------------
 * Launch-list executor (ABI 7, csrc/runlist.hip).  No reference counterpart: the reference's step is an autograd graph walked by
 * torch; here a step is a static list of prepared entry-point calls (the plan modules of kurosiwo_amd), and one ksmi_run_list call issues a
 * whole segment of it on the streams of the step instead of one host-language call per launch.
 *   ksmi_op.kind CALL: fn(args..., stream) through the typed thunk `sig` (ksmi_thunk_id of its signature string: one letter per
 *     argument without the trailing stream -- p pointer, i int, u unsigned, l int64, z size_t, f float, d double); args = nargs 64-bit
 *     slots (integers / pointers by value, float in the low 32 bits, double by bit pattern).  lane 0 | 1 = compute lane; side 1 | 2 =
 *     the launch goes to that weight-gradient stream behind an event recorded on its lane's stream; tag >= 0 = record an event
 *     behind it that KSMI_OP_WAIT_SIDE(tag) consumes.
 *   ORDER: lane b waits for everything lane a was handed.  WAIT_SIDE: lane waits for the tagged side launch (tag < 0: the whole side stream).
 * `skip` (or NULL): one byte per op, non-zero = leave that CALL out this time.  On failure *failed_at = index of the op. */
#define KSMI_OP_CALL 0
#define KSMI_OP_ORDER 1
#define KSMI_OP_WAIT_SIDE 2
typedef struct ksmi_op {
  int32_t kind, sig, lane, side, tag, a, b, nargs;
  const void* fn;
  const uint64_t* args;
} ksmi_op;
int ksmi_thunk_id(const char* signature);                       /* -1: no thunk for that signature */
void* ksmi_runner_create(void);
int ksmi_runner_destroy(void* runner);
/* lane1 NULL: one compute lane; side NULL: weight gradients stay on their lane's stream */
int ksmi_runner_set_streams(void* runner, void* main_stream, void* lane1, void* side, void* side2);
int ksmi_run_list(void* runner, const ksmi_op* ops, int first, int last, const uint8_t* skip, int32_t* failed_at);
/* end of a step: main waits for every other stream that was handed work; tagged events are forgotten */
int ksmi_runner_join(void* runner);

/* ---------------------------------------------------------------------------------
 * Implicit-GEMM convolution family (MFMA).  Replaces nn.Conv2d / nn.ConvTranspose2d
 * forward + input-gradient + weight-gradient:
 *   models/snunet.py:15-17 (3x3 convs), :41 (ConvTranspose2d k2 s2), :132-146
 *   (torch.cat => "virtual concat": up to 6 NHWC sources read in place).
 * ------------------------------------------------------------------------------- */
typedef struct ksmi_src {
  const void* ptr;     /* NHWC activations, `dtype` */
  const float* scale;  /* optional per-channel affine applied on load: v*scale[c]+shift[c] */
  const float* shift;  /*   (indexed by channel within this source's used range)          */
  int32_t C;           /* channel count of the tensor (pixel stride)                        */
  int32_t c_off;       /* first channel used                                                */
  int32_t c_len;       /* number of channels used (multiple of 8)                           */
  int32_t relu;        /* apply max(0,.) after the affine                                   */
} ksmi_src;

typedef struct ksmi_dst {
  void* ptr;           /* NHWC activations, `dtype` */
  int32_t C;           /* channel count of the tensor                                       */
  int32_t c_off;       /* first channel written                                             */
  int32_t n_begin;     /* first GEMM column mapped to this segment (segments ascending)     */
  int32_t n_len;       /* number of columns                                                 */
  int32_t accumulate;  /* 1: dst += result                                                  */
  int32_t pad_;
} ksmi_dst;

typedef struct ksmi_conv_desc {
  ksmi_src src[KSMI_MAX_SRC];
  ksmi_dst dst[KSMI_MAX_SRC];
  int32_t nsrc, ndst;
  const void* wpk;      /* packed weights (ksmi_pack_weights), `dtype` */
  const float* bias;    /* [N] fp32 or NULL */
  float* stats;         /* NULL or [grid_m][2][Npad] partial (sum, sumsq) of the fp32 results  */
  /* optional ReLU/BatchNorm-backward epilogue (dgrad of conv2 in conv_block_nested,
   * models/snunet.py:19-29): result *= (m*m_scale+m_shift > 0) with m read from mask_src;
   * stats then holds (sum r, sum r*xhat), xhat = (m - m_mean)*m_rstd.                   */
  const void* mask_src; /* NHWC `dtype`, N channels */
  const float* m_mean; const float* m_rstd; const float* m_scale; const float* m_shift;
  int32_t B, Hin, Win, Hout, Wout;
  int32_t KH, KW, stride, pad;   /* supported kernel sizes: 1x1, 2x2, 3x3, 4x4 (any stride/pad that keeps the halo <= 512 px) */
  int32_t TH, TW;                /* output patch per workgroup, TH*TW <= 256 */
  int32_t N, Npad;               /* GEMM columns (output channels), padded to 16 */
  int32_t nchunks;
  int32_t ps_cout;               /* >0: pixel-shuffle store for ConvTranspose2d(k2,s2): N = 4*ps_cout,
                                    column j=(dy*2+dx)*ps_cout+n -> out[b,2y+dy,2x+dx,n] */
  uint16_t chunk_c0[KSMI_MAX_CHUNKS];  /* channel offset (within the source's used range) per k-chunk */
  uint8_t chunk_src[KSMI_MAX_CHUNKS];  /* source index per k-chunk */
  int32_t pad_x;                 /* left padding (pad = top padding); set pad_x = pad for the usual symmetric case */
  /* strided output placement (phase convolutions of ConvTranspose2d k4 s2 p1): result pixel (oy,ox) is written to
   * (oy*out_sy + out_oy, ox*out_sx + out_ox) of a [B, out_H, out_W, C] tensor; out_sy == 0 => dense [B,Hout,Wout,C] */
  int32_t out_sy, out_sx, out_oy, out_ox, out_H, out_W;
  /* single-source descriptors may exceed KSMI_MAX_CHUNKS: uniform_kc = chunk elements (ksmi_chunk_elems) means
   * chunk ch starts at channel ch*uniform_kc of src[0] and the chunk tables are ignored; 0 = use the tables */
  int32_t uniform_kc;
  /* strided input view (input gradient of ConvTranspose2d k4 s2 p1 as four 2x2 convolutions over the parity sub-images of dOut):
   * logical source pixel (iy, ix) of the Hin x Win view lives at (iy*in_sy + in_oy, ix*in_sx + in_ox) of a [B, in_H, in_W, C]
   * tensor; in_sy == 0 => dense [B,Hin,Win,C] */
  int32_t in_sy, in_sx, in_oy, in_ox, in_H, in_W;
  /* epilogue extras (igemm2 path): v = alpha*(acc + bias) [alpha == 0 means 1] ; v += resid[pixel][n] (dense
   * [B,Hout,Wout,residC] `dtype`, ResidualBlock of models/changeformer.py:471-483) ; v = max(v, 0) if relu_out
   * (conv -> ReLU -> BatchNorm ordering of conv_diff / make_prediction, changeformer.py:31-46: the BN statistics in
   * `stats` are those of the ReLU output) */
  float alpha;
  int32_t relu_out;
  int32_t residC;
  const void* resid;
  /* rows of `stats` the caller allocated, as returned by ksmi_conv_stats_rows (0: one row per M-tile = ksmi_conv_grid_m, the only
   * layout of the non-persistent kernels).  Short-K bf16 convolutions run on persistent workgroups (csrc/igemm3.hip) that emit one
   * row per workgroup; that kernel is only chosen for a descriptor with statistics when stats_rows announces its row count. */
  int32_t stats_rows;
  /* BatchNorm2-backward GATE of conv_block_nested's output gradient (models/snunet.py:26-29: out = relu(bn2(z) + identity)).  With
   * gate_src != NULL the epilogue forms the TOTAL gradient v = result (+ the old destination when dst.accumulate), zeroes it where
   * gate_src (the block output `out`, N channels) is <= 0, stores it, and `stats` receives (sum v, sum v * xhat) with
   * xhat = (xhat_src - g_mean) * g_rstd (z and the saved statistics of bn2): the separate reduction pass over (d out, out, z) of the
   * block backward disappears into the launch that writes d out.  Only the persistent long-K kernel implements it
   * (ksmi_conv_gate_supported); mutually exclusive with mask_src. */
  const void* gate_src; const void* xhat_src; const float* g_mean; const float* g_rstd;
  /* profiling tag: 1 = this launch is an INPUT GRADIENT.  No effect on the result: the persistent kernels select an identically
   * compiled instantiation whose name carries the direction, so that a rocprofv3 trace / PMC pass separates forward and
   * input-gradient launches of the same tile shape (profiles/summarize.py, bench.py roofline.traffic). */
  int32_t dir;
  int32_t pad2_;
} ksmi_conv_desc;


/* sizeof of a descriptor struct as the library was compiled (0 conv, 1 wgrad, 2 pack, 3 rowsum, 4 tiff_info): bindings check their mirror */
/* Names of the convolution / GEMM / attention kernels this THREAD's entry points have launched since the previous call, ';'-separated,
 * spelled as rocprofv3 demangles them (template arguments included, "(anonymous namespace)::" dropped, bf16 for the 16-bit storage
 * type); returns their number and resets the list.  Measurement aid: ties a HIP-event-timed launch to its row of a rocprofv3
 * table (bench.py `roofline.kernel`).  Elementwise kernels are not listed. */
int ksmi_last_kernels(char* buf, int cap);
/* Device-memory rate probes (measurement aid: bench.py `roofline.measured_peaks`, SURVEY.md §8(d) "measure ... on the box and quote
 * both"): ONE asynchronous pass over `nbytes` of a (and b, c) on `stream`, timed by the caller with HIP events.  mode 0 read by LDS-DMA
 * (nbytes a multiple of 32 MiB), 1 read by 16-byte non-temporal loads, 2 copy a -> b, 3 fp32 triad c = a + s b, 4 fill a; | 8: non-temporal
 * loads / stores, | 16 * g: 8192 >> g workgroups, | 64: one contiguous chunk per workgroup (the caller keeps the best variant).  Mode 0
 * only: | (w << 8), 16 <= w <= 63: the reads wrap inside a window of 2^w bytes (the same nbytes of LDS-DMA traffic served by the L2 or
 * the memory-side cache instead of HBM: the LDS fill rate of on-chip data, tools/fill_probe.py), | (1 << 14): every workgroup walks the
 * same addresses.  `sink`: two device words the read probes may write. */
int ksmi_hbm_probe(int mode, const void* a, void* b, void* c, size_t nbytes, unsigned* sink, void* stream);
size_t ksmi_desc_size(int which);
/* 1: ksmi_conv_forward(d, dtype) runs on a kernel that implements the gate epilogue (gate_src) for this descriptor */
int ksmi_conv_gate_supported(const ksmi_conv_desc* d, int dtype);
/* number of M-tiles (= rows of `stats` when stats_rows == 0) a descriptor launches */
int ksmi_conv_grid_m(const ksmi_conv_desc* d);
/* rows of `stats` ksmi_conv_forward(d, dtype) will write when d->stats_rows is set to the returned value */
int ksmi_conv_stats_rows(const ksmi_conv_desc* d, int dtype);
int ksmi_conv_forward(const ksmi_conv_desc* d, int dtype, void* stream);
/* (ABI 7, test / tooling hook) which kernel ksmi_conv_forward(d, dtype) starts, decided on the host without launching: info[0] = kernel
 * generation (4 = persistent long-K igemm4, 3 = persistent short-K igemm3, 2 = igemm2, 1 = first generation), info[1] = 1 when the chosen
 * launcher has a compiled instance for the geometry it picked, igemm4: info[2..7] = WM, NF, waves, schedule, gx, gy.  `info`: 8 ints. */
int ksmi_conv_dispatch_info(const ksmi_conv_desc* d, int dtype, int32_t* info);

/* Weight packing fp32 parameter -> `dtype` [nchunks][taps][Npad][chunk_elems].
 * element (chunk, tap, j, kk) = w[ k*sK + (j % n_mod)*sN + (j / n_mod)*sD + tap'*sT ],
 * k = k_off[chunk] + kk (zero if kk >= k_len[chunk]); tap' = taps-1-tap if flip. */
typedef struct ksmi_pack_desc {
  const float* w; void* out;
  int32_t nchunks, taps, N, Npad, n_mod;
  int64_t sK, sN, sD, sT;
  int32_t flip;
  int32_t k_off[KSMI_MAX_CHUNKS];
  int32_t k_len[KSMI_MAX_CHUNKS];
  int32_t use_tap_map;            /* 1: tap' = tap_map[tap] (phase kernels of ConvTranspose2d k4 s2 p1 / strided-conv gradients); < 0: zero tap */
  int32_t tap_map[16];
  int32_t uniform_kc, k_total;    /* uniform_kc != 0: k_off = chunk*uniform_kc, k_len = min(uniform_kc, k_total - k_off); tables ignored */
} ksmi_pack_desc;
int ksmi_pack_weights(const ksmi_pack_desc* d, int dtype, void* stream);
/* n descriptors stored in DEVICE memory, packed by one launch (a model plan re-packs every conv each step) */
int ksmi_pack_weights_batched(const ksmi_pack_desc* descs_device, int n, int dtype, void* stream);

/* Weight gradient: G[tap][k][n] = sum_pixels X[p*stride+tap-pad][k] * dY[p][n], X = virtual
 * concat (with the same optional affine+ReLU on load), reduced over all pixels with a
 * split over M-tiles, then written as fp32 to grad[k*gK + n*gN + tap*gT] (+= if accumulate). */
typedef struct ksmi_wgrad_desc {
  ksmi_src src[KSMI_MAX_SRC];
  int32_t nsrc;
  const void* dy; int32_t dyC; int32_t dy_c_off;   /* NHWC `dtype` */
  int32_t B, Hin, Win, Hout, Wout;
  int32_t KH, KW, stride, pad;
  int32_t TH, TW;
  int32_t N;                 /* columns (channels of dY used), multiple of 8 */
  int32_t nchunks;
  int32_t nsplit;            /* number of partial slabs (workgroups along the pixel axis) */
  float* partial;            /* workspace [nsplit][taps][nchunks*chunk_elems][Npad16] fp32 */
  float* grad; int64_t gK, gN, gT; int32_t accumulate;
  int32_t k_off[KSMI_MAX_CHUNKS];  /* first K index (row of grad) of each chunk */
  int32_t k_len[KSMI_MAX_CHUNKS];
  uint16_t chunk_c0[KSMI_MAX_CHUNKS];
  uint8_t chunk_src[KSMI_MAX_CHUNKS];
  int32_t uniform_kc, k_total;    /* same meaning as in the pack descriptor: single source, more than KSMI_MAX_CHUNKS chunks */
  /* phase weight gradients of ConvTranspose2d k4 s2 p1: strided view of the halo-side source (as ksmi_conv_desc.in_*), separate left
   * padding, and an explicit tap -> gradient offset table (use_tap_off: grad[k*gK + n*gN + tap_off[tap]]) */
  int32_t in_sy, in_sx, in_oy, in_ox, in_H, in_W;
  int32_t pad_x_set, pad_x;
  int32_t use_tap_off;
  int32_t tap_off[16];
  /* optional fused bias gradient of a plain nn.Linear (1x1, one source), by the return value of ksmi_conv_wgrad_fuses_bias():
   *   1: bias_grad[n] (+)= sum over rows of dY[row][n] (the token-GEMM weight gradient that writes the gradient in one split: dY is
   *      already in LDS there);
   *   2 (ABI 6): the split mode of the same path OVERWRITES bias_grad[split * N + n] = sum over the split's rows of dY[row][n] for
   *      split < nsplit -- the caller sums the nsplit rows (bias_accumulate is not used);
   *   0: the field is ignored and the caller runs ksmi_colsum / ksmi_channel_sum. */
  float* bias_grad; int32_t bias_accumulate;
} ksmi_wgrad_desc;
size_t ksmi_conv_wgrad_workspace(const ksmi_wgrad_desc* d, int dtype);
int ksmi_conv_wgrad(const ksmi_wgrad_desc* d, int dtype, void* stream);
int ksmi_conv_wgrad_fuses_bias(const ksmi_wgrad_desc* d, int dtype);

/* First-layer 3x3 conv on the raw image (NCHW fp32, Cin <= 8 -> Cout = 32*k):
 * models/snunet.py:75 conv0_0.conv1.  Forward writes NHWC `dtype` + BN partial stats;
 * wgrad writes fp32 dW (OIHW) and db. */
int ksmi_conv_first_forward(const float* x_nchw, const float* w, const float* bias, void* out, float* stats,
                            int B, int Cin, int H, int W, int Cout, int dtype, void* stream);
/* The same on RAW tiles: the Dataset's per-tile pipeline (dataset/Dataset.py:164-168 clamp to [0, clamp_input] then
 * nan_to_num(nan = clamp_input); :193-198 Normalize(mean, std)) applied in the image load.  mean / std / clamp: [Cin] fp32
 * device arrays, all three or none (none = already normalised); clamp[c] < 0: channel c is not clamped and its NaNs become
 * the mean (DEM / slope).  Bit-identical to ksmi_sar_preprocess followed by ksmi_conv_first_forward.
 * x_tail != NULL: the trainer's torch.cat((image, dem), dim=1) (training/change_detection_trainer.py:117-133) as an address
 * choice: channels [0, c_head) are read from x_nchw [B,c_head,H,W], channels [c_head, Cin) from x_tail [B,Cin-c_head,H,W]. */
int ksmi_conv_first_forward_raw(const float* x_nchw, const float* x_tail, int c_head, const float* w, const float* bias,
                                void* out, float* stats, int B, int Cin, int H, int W, int Cout, const float* mean,
                                const float* stdv, const float* clamp, int dtype, void* stream);
int ksmi_conv_first_stats_rows(int B, int H, int W);
/* im2col of the raw image: out[b,y,x,c*9+t] (NHWC `dtype`, Kpad channels, zero padded); the first conv and its
 * weight gradient then run on the MFMA implicit-GEMM kernels as a 1x1 conv (k = c*9+t = OIHW flattening). */
int ksmi_im2col3x3(const float* x_nchw, void* out, int B, int Cin, int H, int W, int Kpad, int dtype, void* stream);
int ksmi_im2col3x3_raw(const float* x_nchw, const float* x_tail, int c_head, void* out, int B, int Cin, int H, int W,
                       int Kpad, const float* mean, const float* stdv, const float* clamp, int dtype, void* stream);
int ksmi_conv_first_wgrad(const float* x_nchw, const void* dy, float* dw, float* workspace, size_t ws_bytes,
                          int B, int Cin, int H, int W, int Cout, int accumulate, int dtype, void* stream);
size_t ksmi_conv_first_wgrad_workspace(int B, int Cin, int H, int W, int Cout);

/* ---------------------------------------------------------------------------------
 * BatchNorm2d (train: batch statistics) + the conv_block_nested glue,
 * models/snunet.py:16,18,19-29 (nn.BatchNorm2d defaults eps 1e-5, momentum 0.1).
 * ------------------------------------------------------------------------------- */
/* NOTE: `partial` of ksmi_bn_finalize / ksmi_reduce_rows is scratch: long row lists are folded IN PLACE first.
 * partial [rows][2][Cpad] (sum, sumsq) -> mean, rstd, scale=gamma*rstd, shift=beta-mean*scale;
 * updates running_mean/var (unbiased var) and num_batches_tracked (int64) when training.
 * training==0: scale/shift from the running statistics. */
int ksmi_bn_finalize(const float* partial, int rows, int Cpad, int C, double count,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float momentum, float eps, int training,
                     float* mean, float* rstd, float* scale, float* shift, void* stream);
/* out = relu(z*scale + shift + identity)   (snunet.py:27-28) */
int ksmi_bn_add_relu(const void* z, const void* identity, const float* scale, const float* shift, void* out,
                     int64_t npix, int C, int dtype, void* stream);
/* backward of out = relu(bn2(z) + i):  g = dout*(out>0) (written in place of dout);
 * pass 1 accumulates partial[rows][2][C] = (sum g, sum g*zhat). */
int ksmi_bnrelu_bwd_reduce(const void* dout, const void* out, const void* z, const float* mean, const float* rstd,
                           float* partial, int rows, int64_t npix, int C, int dtype, void* stream);
/* sums[K][C] = sum over rows of partial[row][k][Cstride] (fp64 accumulate); optionally
 * dgamma (+)= sums[1], dbeta (+)= sums[0] */
int ksmi_reduce_rows(const float* partial, int rows, int K, int Cstride, int C, float* sums,
                     float* dgamma, float* dbeta, int accumulate, void* stream);
/* Deferred row reductions of a whole backward pass in ONE launch (parameter gradients nobody waits for: conv / deconv bias
 * gradients, models/snunet.py:15-17,41): entry e sums partial[(r*K + k)*Cstride + c] over its rows (fp64, fixed order) into
 * dst[c] (+= if accumulate); rows == 0 contributes nothing (a bias followed by a train-mode BatchNorm gets exact zeros).
 * Entries sharing one dst (a module called twice: the siamese encoder blocks) form a chain: `head` = 1 on the first, `next` =
 * index of the following entry or -1; the head's block sums the whole chain.  n descriptors in DEVICE memory. */
typedef struct ksmi_rowsum_desc {
  const float* partial;
  float* dst;
  int32_t rows, K, k, Cstride, C, accumulate;
  int32_t head, next;
} ksmi_rowsum_desc;
int ksmi_reduce_rows_batched(const ksmi_rowsum_desc* descs_device, int n, void* stream);
/* The same launch for destinations wider than 512 columns (the 9*4C depthwise-convolution weight gradients of the MiT blocks,
 * reference models/changeformer/ChangeFormer.py DWConv): max_c = the widest C of the table. */
int ksmi_reduce_rows_batched_wide(const ksmi_rowsum_desc* descs_device, int n, int max_c, void* stream);
/* pass 2: g = dout*(out>0) -> dout (in place); dz = gamma*rstd*(g - s0/n - zhat*s1/n) */
int ksmi_bnrelu_bwd_apply(void* dout_g, const void* out, const void* z, const float* mean, const float* rstd,
                          const float* gamma, const float* sums, void* dz, double count,
                          int64_t npix, int C, int dtype, void* stream);
/* pass 2 behind a convolution with the gate epilogue (ksmi_conv_desc.gate_src: g is already d out * (out > 0) and `sums` come from
 * that launch's statistics rows): dz = gamma*rstd*(g - s0/n - zhat*s1/n); g is read only */
int ksmi_bn_bwd_apply_gated(const void* g, const void* z, const float* mean, const float* rstd, const float* gamma, const float* sums,
                            void* dz, double count, int64_t npix, int C, int dtype, void* stream);
/* di = g + gamma*rstd*(r - t0/n - xhat*t1/n), xhat=(i-mean)*rstd; written over r; also
 * partial[rows][1][C] = sum di (conv1 bias gradient). */
int ksmi_bn_bwd_apply_add(void* r_di, const void* g, const void* i, const float* mean, const float* rstd,
                          const float* gamma, const float* sums, float* partial, int rows, double count,
                          int64_t npix, int C, int dtype, void* stream);
/* Round 4: the statistics FINISH folded into the pass that needs the finished statistics (csrc/bnfused.hip): one launch where the
 * forward pass had ksmi_bn_finalize + ksmi_bn_add_relu (+ ksmi_maxpool2x2_forward) and the backward pass ksmi_reduce_rows + an apply
 * pass (models/snunet.py:16,18,24-29 forward, their autograd backward).  The pass runs on <= 256 workgroups of 1024 threads; each of
 * them first sums the partial rows itself (fp64, fixed order: every workgroup gets the same bits; lists longer than 512 rows are
 * folded in place first, `partial` is scratch), workgroup 0 publishes mean / rstd / scale / shift + the running statistics (forward)
 * or `sums` + dbeta (+)= sums[0], dgamma (+)= sums[1] (backward).  The streaming arithmetic is that of the passes they replace.
 * ksmi_bn_fused_supported: 1 when (C, row stride, dtype) qualify (C <= 512, multiples of the 16-byte vector). */
int ksmi_bn_fused_supported(int C, int Cstride, int dtype);
int ksmi_bn_fused_max_rows(void);     /* most rows of bias_partial ksmi_bn_bwd_fin_apply_add accepts (= its workgroup count) */
/* out = relu(bn(z) + identity) from the partial rows [rows][2][Cpad] of the convolution that wrote z; pooled != NULL: also
 * pooled[B,H/2,W/2,C] = maxpool2x2(out) (snunet.py:73, 121-130: every encoder block output is pooled right away) */
int ksmi_bn_fin_add_relu(float* partial, int rows, int Cpad, int C, double count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                         float* mean, float* rstd, float* scale, float* shift, const void* z, const void* identity, void* out,
                         void* pooled, int B, int H, int W, int dtype, void* stream);
/* ksmi_reduce_rows(partial [rows][2][Cstride]) + ksmi_bn_bwd_apply_gated */
int ksmi_bn_bwd_fin_apply_gated(float* partial, int rows, int Cstride, float* sums, float* dgamma, float* dbeta, int accumulate,
                                const void* g, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz,
                                double count, int64_t npix, int C, int dtype, void* stream);
/* ksmi_reduce_rows + ksmi_bnrelu_bwd_apply */
int ksmi_bnrelu_bwd_fin_apply(float* partial, int rows, int Cstride, float* sums, float* dgamma, float* dbeta, int accumulate,
                              void* dout_g, const void* out, const void* z, const float* mean, const float* rstd, const float* gamma,
                              void* dz, double count, int64_t npix, int C, int dtype, void* stream);
/* ksmi_reduce_rows + ksmi_bn_bwd_apply_add; bias_partial [bias_rows][C], bias_rows <= ksmi_bn_fused_max_rows() */
int ksmi_bn_bwd_fin_apply_add(float* partial, int rows, int Cstride, float* sums, float* dgamma, float* dbeta, int accumulate,
                              void* r_di, const void* g, const void* i, const float* mean, const float* rstd, const float* gamma,
                              float* bias_partial, int bias_rows, double count, int64_t npix, int C, int dtype, void* stream);
/* partial[rows][1][C] = per-channel sum of x (bias gradients) */
int ksmi_channel_sum(const void* x, float* partial, int rows, int64_t npix, int C, int dtype, void* stream);
/* out[c] (+)= sum_r x[r][c] of a row-major token matrix in one launch (bias gradient of nn.Linear, vision_transformer.py:22-31:
 * a few thousand rows; long pixel axes use ksmi_channel_sum + ksmi_reduce_rows).  Deterministic summation order. */
int ksmi_colsum(const void* x, int64_t rows, int C, float* out, int accumulate, int dtype, void* stream);

/* nn.MaxPool2d(2,2)  models/snunet.py:73 ; backward routes to the first maximum */
int ksmi_maxpool2x2_forward(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
int ksmi_maxpool2x2_backward(const void* x, const void* dy, void* dx, int accumulate,
                             int B, int H, int W, int C, int dtype, void* stream);

/* ---------------------------------------------------------------------------------
 * ECAM head: models/snunet.py:49-62 (ChannelAttention), :146-151 (fusion + conv_final).
 * x[4] = x0_1..x0_4, NHWC `dtype`, C channels each (C = base_channel).
 * ------------------------------------------------------------------------------- */
/* pooled[B][5C] avg, [B][5C] max (first 4C: cat(x0_1..4), last C: their sum), argmax idx */
int ksmi_ecam_pool(const void* const x[4], float* avg, float* mx, int32_t* argmax,
                   float* workspace, int B, int HW, int C, int dtype, void* stream);
size_t ksmi_ecam_pool_workspace(int B, int HW, int C);
/* ca[B][4C] and ca1[B][C] = sigmoid(fc2(relu(fc1(avg))) + fc2(relu(fc1(max)))) */
int ksmi_ecam_mlp(const float* avg, const float* mx, const float* ca_fc1, const float* ca_fc2,
                  const float* ca1_fc1, const float* ca1_fc2, float* ca, float* ca1,
                  float* hidden /* [B][2][(4C/16)+(C/4)] saved for backward */, int B, int C, void* stream);
/* logits[b,k,p] = bias[k] + sum_c Wf[k][c] * ca[b,c]*(x[c/C][b,p,c%C] + ca1[b,c%C])  (NCHW fp32 out) */
int ksmi_ecam_final_forward(const void* const x[4], const float* ca, const float* ca1, const float* wf,
                            const float* bias, float* logits, int B, int HW, int C, int ncls, int dtype, void* stream);
/* backward, phase A: reductions dca[B][4C], dca1[B][C], dWf[ncls][4C], dbias[ncls] (partials in ws) */
int ksmi_ecam_final_backward_reduce(const void* const x[4], const float* dlogits, const float* ca, const float* ca1,
                                    const float* wf, float* dca, float* dca1, float* dwf, float* dbias,
                                    float* workspace, int B, int HW, int C, int ncls, int dtype, void* stream);
size_t ksmi_ecam_bwd_workspace(int B, int HW, int C, int ncls);
/* MLP backward: dca,dca1 -> davg[B][5C], dmax[B][5C], and fc weight grads (+=) */
int ksmi_ecam_mlp_backward(const float* avg, const float* mx, const float* hidden, const float* ca, const float* ca1,
                           const float* dca, const float* dca1, const float* ca_fc1, const float* ca_fc2,
                           const float* ca1_fc1, const float* ca1_fc2, float* davg, float* dmax,
                           float* g_ca_fc1, float* g_ca_fc2, float* g_ca1_fc1, float* g_ca1_fc2,
                           float* workspace, int B, int C, void* stream);
size_t ksmi_ecam_mlp_bwd_workspace(int B, int C);
/* phase B: dx[j][b,p,c] = ca*dout' + davg terms/HW, plus the max-pool scatter at argmax */
int ksmi_ecam_final_backward_dx(void* const dx[4], const float* dlogits, const float* ca, const float* wf,
                                const float* davg, const float* dmax, const int32_t* argmax,
                                int B, int HW, int C, int ncls, int dtype, void* stream);

/* ---------------------------------------------------------------------------------
 * Loss: utilities/bce_and_dice.py:18-24 + utilities/dice.py:93-137 (softmax CE + softmax
 * Dice, ignore_index 3) and nn.CrossEntropyLoss alone (utilities/utilities.py:307-347).
 * logits NCHW fp32 [B,3,H,W], labels int64 [B,H,W].  out[0..2] = total, ce, dice.
 * ------------------------------------------------------------------------------- */
size_t ksmi_loss_workspace(int B, int HW);
int ksmi_ce_dice_forward(const float* logits, const int64_t* labels, const float* class_w, int with_dice,
                         float* out3, float* workspace, int B, int HW, int ignore_index, void* stream);
/* dlogits = grad_scale * d total / d logits ; must follow ksmi_ce_dice_forward on the same workspace */
int ksmi_ce_dice_backward(const float* logits, const int64_t* labels, const float* class_w, int with_dice,
                          const float* workspace, const float* grad_scale /* device scalar or NULL (=1) */,
                          float* dlogits, int B, int HW, int ignore_index, void* stream);

/* Metrics: predictions = argmax(1) (lowest index on ties), cm[4][4] int64 += counts of
 * (target,pred) over target != ignore_index.  training/change_detection_trainer.py:152,184-189;
 * utilities/utilities.py:228-265. */
int ksmi_argmax_confusion(const float* logits, const int64_t* labels, int64_t* pred /* or NULL */, int64_t* cm,
                          int B, int C, int HW, int ignore_index, void* stream);
/* (ABI 7) the same pass with per-sample GROUP tables -- the per-AOI and per-climate-zone metrics of the evaluation loops
 * (training/change_detection_trainer.py:331-337, 437-472: one torchmetrics object per activation id / zone, updated sample by sample):
 * sample b also adds its counts to cms_a[slot_a[b]][4][4] and cms_b[slot_b[b]][4][4] (device int32 slots, < 0: the sample belongs to
 * no group of that table; a NULL table is skipped; cm may be NULL).  One launch per batch whatever the number of groups. */
int ksmi_argmax_confusion_grouped(const float* logits, const int64_t* labels, int64_t* pred /* or NULL */, int64_t* cm /* or NULL */,
                                  const int32_t* slot_a, int64_t* cms_a, const int32_t* slot_b, int64_t* cms_b,
                                  int B, int C, int HW, int ignore_index, void* stream);

/* ---------------------------------------------------------------------------------
 * Optimisers: torch.optim.Adam(lr) / SGD(momentum, weight_decay) as used at
 * training/change_detection_trainer.py:45-66.  Flat fp32 buffers (parameter arena).
 * step_count: device int64 scalar incremented by the kernel (graph-replay safe).
 * ------------------------------------------------------------------------------- */
int ksmi_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t* step_count,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);
/* torch.optim.AdamW (training/change_detection_trainer.py:55-60, the `adamw` branch: betas and weight_decay of the method json):
 * decoupled weight decay p *= 1 - lr * weight_decay, then the Adam update; same state and step counter as ksmi_adam_step */
int ksmi_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t* step_count,
                    float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);
/* ksmi_adam_step / ksmi_adamw_step (decoupled != 0) that ALSO writes the updated parameters as bf16 into mirror_bf16 [n] (8-byte aligned):
 * the bf16 operand copy of the token GEMMs without a separate cast pass (SURVEY.md K8 "optional bf16 shadow write") */
int ksmi_adam_step_mirror(float* p, const float* g, float* m, float* v, int64_t n, int64_t* step_count,
                          float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, int decoupled,
                          void* mirror_bf16, void* stream);
int ksmi_sgd_step(float* p, const float* g, float* mom, int64_t n, int64_t* step_count,
                  float lr, float momentum, float weight_decay, float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------
 * Token-sequence ops of the FloodViT path: models/vision_transformer.py (lucidrains-style ViT) and the
 * FinetunerSegmentation / Decoder head, models/model_utilities.py:22-94.  Activations are [rows][C] `dtype`
 * (= NHWC with H*W tokens); the Linear layers run on ksmi_conv_forward / ksmi_conv_wgrad as 1x1 convolutions.
 * ------------------------------------------------------------------------------- */
/* nn.LayerNorm(C) (vision_transformer.py:22,43,72,124,126; eps 1e-5); mean/rstd [rows] saved for backward */
int ksmi_layernorm_forward(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                           int rows, int C, float eps, int dtype, void* stream);
/* dx (+)= LN backward; partial[blocks][2][C] = (sum dy, sum dy*xhat) per block -> ksmi_reduce_rows gives dbeta, dgamma */
int ksmi_layernorm_bwd_blocks(int rows);
int ksmi_layernorm_backward(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                            void* dx, int accumulate, float* partial, int rows, int C, int dtype, void* stream);
/* nn.GELU() exact-erf (vision_transformer.py:24), elementwise helpers */
int ksmi_gelu_forward(const void* x, void* y, int64_t n, int dtype, void* stream);
int ksmi_gelu_backward(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream);
/* (ABI 7) dx = (dy * Dropout(site; element)) * gelu'(x): the backward of act -> drop (models/changeformer.py:129-130) in one pass */
int ksmi_gelu_backward_drop(const void* dy, const void* x, void* dx, int64_t n, uint32_t thr, float inv_keep, uint32_t site,
                            const uint32_t* rng_state, int dtype, void* stream);
int ksmi_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
int ksmi_relu_backward(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream);
int ksmi_relu_forward(const void* x, void* y, int64_t n, int dtype, void* stream);   /* Decoder.relu, model_utilities.py:44 */
/* x[:, 1:] (vision_transformer.py:150-151): backward=0 [B][N1][C] -> dense [B][N1-1][C]; backward=1 the adjoint (cls rows zero) */
/* ---- MAE pre-training (models/mae.py:54-124) --------------------------------------------------------
 * Token shuffles driven by slices of the per-sample permutation `rand_indices` (:73-78): idx is int64 [B][idx_rs] (row stride in
 * elements), rows are [B*N][C] in `dtype`, tables / fill vectors are fp32 parameters.
 *   gather : dst[b][j] = src[b][idx[b][j]] (+ table[idx[b][j] + table_off])     tokens[batch_range, unmasked_indices] (:78) with
 *            the position rows (:65), masked_patches (:82), decoded mask tokens (:113)
 *   scatter: dst[b][idx[b][j]] = (src ? src[b][j] : fill) + table[idx[b][j]]     decoder_tokens assembly (:94-110); with
 *            table = fill = NULL it is the adjoint of gather (the destination must be zeroed where no row lands)
 *   batch_sum: out[i] (+)= sum_b x[b][i]   gradient of a position table;  mse_loss: F.mse_loss (:122) and its gradient
 *            dpred = 2 (pred - target) grad_scale [* *upstream, a device scalar] / n in one pass (dpred, upstream may be NULL). */
int ksmi_gather_rows(const void* src, const int64_t* idx, int idx_rs, void* dst, const float* table, int table_off, int B, int Ns, int Nd, int C,
                     int dtype, void* stream);
int ksmi_scatter_rows(const void* src, const float* fill, const int64_t* idx, int idx_rs, const float* table, void* dst, int B, int Nsrc, int Nd,
                      int C, int dtype, void* stream);
int ksmi_batch_sum(const void* x, float* out, int B, int64_t n, int accumulate, int dtype, void* stream);
size_t ksmi_mse_workspace(void);
int ksmi_mse_loss(const void* pred, const void* target, void* dpred, float grad_scale, const float* upstream, float* loss, float* workspace,
                  int64_t n, int dtype, void* stream);
int ksmi_drop_cls(const void* x, void* y, int B, int N1, int C, int backward, int dtype, void* stream);
/* head output NHWC [B][HW][Cs] (first C channels real) -> NCHW fp32 logits, and d(logits) back (pad channels zero) */
int ksmi_logits_to_nchw(const void* x, float* y, int B, int C, int Cs, int64_t HW, int dtype, void* stream);
int ksmi_dlogits_to_nhwc(const float* dy, void* dx, int B, int C, int Cs, int64_t HW, int dtype, void* stream);
/* Rearrange("b c (h p1) (w p2) -> b (h w) (p1 p2 c)") of the NCHW fp32 image (vision_transformer.py:122) */
int ksmi_patchify(const float* x_nchw, void* out, int B, int Cin, int H, int W, int P, int dtype, void* stream);
/* x0 = cat(cls, emb) + pos_embedding (vision_transformer.py:143-145) and its backward (dcls, dpos fp32 "=") */
int ksmi_vit_embed_forward(const void* emb, const float* cls, const float* pos, void* x0, int B, int N1, int C, int dtype, void* stream);
int ksmi_vit_embed_backward(const void* dx0, void* demb, float* dcls, float* dpos, int B, int N1, int C, int dtype, void* stream);
/* softmax(q k^T * scale) v per head (vision_transformer.py:52-63); qkv [B*N][3*H*D] in "(3 h d)" order, out [B*N][H*D],
 * lse [B][H][N] saved for backward; dqkv written ("=").  D must be 64.  bf16: MFMA flash-style kernels (csrc/attn_mfma.hip),
 * the backward needs `workspace` (ksmi_attention_bwd_workspace bytes); fp32 (parity mode): VALU kernels. */
int ksmi_attention_forward(const void* qkv, void* out, float* lse, int B, int N, int H, int D, float scale, int dtype, void* stream);
size_t ksmi_attention_bwd_workspace(int B, int N, int H, int D, int dtype);
int ksmi_attention_backward(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, void* workspace,
                            int B, int N, int H, int D, float scale, int dtype, void* stream);
/* [relu ->] nn.Upsample(scale_factor=2) nearest (model_utilities.py:36-41) */
int ksmi_upsample2_forward(const void* x, void* y, int B, int H, int W, int C, int relu, int dtype, void* stream);
int ksmi_upsample2_backward(const void* dy, const void* x_pre, void* dx, int B, int H, int W, int C, int relu, int dtype, void* stream);

/* ---------------------------------------------------------------------------------
 * ChangeFormerV6 glue (models/changeformer.py); the dense contractions run on ksmi_conv_forward / ksmi_conv_wgrad.
 * ------------------------------------------------------------------------------- */
/* out[b,oy,ox, c*KH*KW + ky*KW + kx] (Kpad columns, zero padded) from an NHWC `dtype` activation or (src_nchw_f32) the raw
 * NCHW fp32 image: OverlapPatchEmbed.proj (:267) and Attention.sr (:166) become GEMMs against the OIHW-flattened weight */
/* Channel-fastest variant of the same family (K index = tap * Cin + c, NHWC source only): every (pixel, tap) moves whole 16-byte
 * channel vectors in both directions.  The GEMM weight of the convolution (OIHW, fp32: changeformer.py:262-263, :163) is re-ordered
 * to match by ksmi_weight_to_tc (out[n][tap*Cin + c] = w[n][c][tap], zero K padding) and its gradient is returned to OIHW order by
 * ksmi_grad_from_tc. */
int ksmi_im2col_tc(const void* x, void* out, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int Kpad,
                   int dtype, void* stream);
int ksmi_col2im_tc(const void* dcol, void* dx, int accumulate, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride,
                   int pad, int Kpad, int dtype, void* stream);
int ksmi_weight_to_tc(const float* w, void* out, int N, int Cin, int taps, int Kpad, int dtype, void* stream);
int ksmi_grad_from_tc(const float* g, float* grad, int N, int Cin, int taps, int Kpad, int accumulate, void* stream);
int ksmi_im2col(const void* x, void* out, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad,
                int Kpad, int src_nchw_f32, int dtype, void* stream);
int ksmi_col2im(const void* dcol, void* dx, int accumulate, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad,
                int Kpad, int dtype, void* stream);
/* Mlp.dwconv + act (:85-96,128-129): z = depthwise3x3(x) + b (w fp32 [C][9]), g = gelu(z); adjoint of the conv; weight/bias
 * gradient partials partial[rows][10*C] (columns c*9+t, then 9C+c) for ksmi_reduce_rows */
int ksmi_dwconv3x3_gelu_forward(const void* x, const float* w, const float* bias, void* z, void* g, int B, int H, int W, int C,
                                int dtype, void* stream);
/* (ABI 7) ... with Mlp.drop (:130) behind the activation applied to g as it is stored: the draw of ksmi_dropout_apply(g, site) on element
 * (token * C + c), product formed on the rounded activation (bit-identical to the two passes); thr = 0: the plain forward */
int ksmi_dwconv3x3_gelu_forward_drop(const void* x, const float* w, const float* bias, void* z, void* g, int B, int H, int W, int C,
                                     uint32_t thr, float inv_keep, uint32_t site, const uint32_t* rng_state, int dtype, void* stream);
int ksmi_dwconv3x3_backward_input(const void* dz, const float* w, void* dx, int B, int H, int W, int C, int dtype, void* stream);
int ksmi_dwconv3x3_wgrad(const void* x, const void* dz, float* partial, int rows, int B, int H, int W, int C, int dtype, void* stream);
/* Attention against the spatially reduced keys (:190-207): q [B*Nq][C], kv [B*Nk][2C] "(2 h d)", out [B*Nq][C]; Nk = 49,
 * head dim C/H in {64, 80}.  backward (out = the forward's output) writes dq, dkv ("=") using `workspace`
 * (ksmi_sr_attention_bwd_workspace bytes).  bf16 runs on the MFMA kernels of csrc/attn_mfma.hip. */
int ksmi_sr_attention_forward(const void* q, const void* kv, void* out, int B, int Nq, int Nk, int H, int C, float scale, int dtype,
                              void* stream);
size_t ksmi_sr_attention_bwd_workspace(int B, int Nq, int Nk, int H, int C);
int ksmi_sr_attention_backward(const void* q, const void* kv, const void* out, const void* dout, void* dq, void* dkv, void* workspace, int B,
                               int Nq, int Nk, int H, int C, float scale, int dtype, void* stream);
/* Stochastic layers of ChangeFormerV6 (drop_rate = attn_drop = drop_path_rate = 0.1, changeformer.py:267-275).  The reference draws
 * its Bernoulli masks from torch's global generator; here a draw is a pure function of (seed, step, site, element): the backward
 * pass regenerates the forward's mask and oracle/rng_ref.py reproduces it on the CPU.  rng_state = 2 words in device memory
 * {seed, step}; ksmi_rng_advance does step += 1 on the stream (once per training forward).  thr = round(p * 2^32), an element
 * is kept when its 32-bit draw >= thr and scaled by inv_keep = 1/(1-p); thr = 0 switches a layer off.
 *   ksmi_dropout_apply: y[r][c] = [resid[r][c] +] x[r][c] * Dropout(site; element r*cols+c) * DropPath(dp_site; sample
 *   r / rows_per_sample) -- nn.Dropout (:107,162) fused with DropPath (:236-241, timm drop_path: per-sample mask / keep_prob)
 *   and the residual add of Block.forward; the backward of the branch is the same call on the gradient with resid = NULL.
 *   ksmi_sr_attention_*_drop: attn_drop (:160,203) on the softmax rows, element ((b*H+h)*Nq+q)*Nk+key. */
int ksmi_rng_advance(uint32_t* rng_state, void* stream);
int ksmi_dropout_apply(const void* x, const void* resid, void* y, int64_t rows, int cols, int rows_per_sample, uint32_t thr, float inv_keep,
                       uint32_t site, uint32_t dp_thr, float dp_inv_keep, uint32_t dp_site, const uint32_t* rng_state, int dtype, void* stream);
/* FC-Siam (N2 row): y = relu(z*scale[c] + shift[c]) * Dropout2d -- nn.Dropout2d(p=0.2) after every BN+ReLU (siam_conc.py:20-93,
 * 101-172) zeroes whole (sample, channel) planes: draw index b*C + c of `site`, kept planes scaled by inv_keep; thr = 0: plain BN+ReLU.
 * Backward: the plane scale is a constant on the active set, so ksmi_bnrelu_bwd_reduce + ksmi_reduce_rows_scaled +
 * ksmi_bnrelu_bwd_apply_scaled with alpha = inv_keep (the mask is read back from out > 0) are the adjoint. */
int ksmi_bn_relu_drop2d(const void* z, const float* scale, const float* shift, void* y, int B, int64_t HW, int C, uint32_t thr, float inv_keep,
                        uint32_t site, const uint32_t* rng_state, int dtype, void* stream);
int ksmi_reduce_rows_scaled(const float* partial, int rows, int K, int Cstride, int C, float* sums, float* dgamma, float* dbeta,
                            int accumulate, float alpha, void* stream);
int ksmi_bnrelu_bwd_apply_scaled(void* dout_g, const void* out, const void* z, const float* mean, const float* rstd, const float* gamma,
                                 const float* sums, void* dz, double count, int64_t npix, int C, float alpha, int dtype, void* stream);
/* FC-Siam-diff skips (siam_diff.py:137-161): y = |a - b| ; da (+)= dy*sign(a-b), db (+)= -dy*sign(a-b) */
int ksmi_absdiff_forward(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);
int ksmi_absdiff_backward(const void* a, const void* b, const void* dy, void* da, void* db, int accumulate_a, int accumulate_b, int64_t n,
                          int dtype, void* stream);
int ksmi_sr_attention_forward_drop(const void* q, const void* kv, void* out, int B, int Nq, int Nk, int H, int C, float scale,
                                   uint32_t drop_thr, float drop_inv_keep, uint32_t site, const uint32_t* rng_state, int dtype, void* stream);
int ksmi_sr_attention_backward_drop(const void* q, const void* kv, const void* out, const void* dout, void* dq, void* dkv, void* workspace,
                                    int B, int Nq, int Nk, int H, int C, float scale, uint32_t drop_thr, float drop_inv_keep, uint32_t site,
                                    const uint32_t* rng_state, int dtype, void* stream);
/* F.interpolate(mode="bilinear", align_corners=False) (:581-608): y = [add +] resize(x); adjoint dx (+)= resize^T(dy) (upsampling) */
int ksmi_bilinear_forward(const void* x, const void* add, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int dtype, void* stream);
int ksmi_bilinear_backward(const void* dy, void* dx, int accumulate, int B, int Hi, int Wi, int Ho, int Wo, int C, int dtype, void* stream);
/* input gradient of y = BN(r), r = relu(v) or v (conv -> ReLU -> BN of conv_diff / make_prediction :31-46; linear_fuse :563-567):
 * dv = gamma*rstd*(dy - sums[0]/n - rhat*sums[1]/n) [masked by r > 0] */
/* partial[rows][2][C] = (sum dy, sum dy*xhat) of a plain BatchNorm (ResNet downsample branch of row U1) for ksmi_reduce_rows */
int ksmi_bn_bwd_reduce(const void* dy, const void* x, const float* mean, const float* rstd, float* partial, int rows, int64_t npix, int C,
                       int dtype, void* stream);
int ksmi_bn_bwd_apply(const void* dy, const void* r, const float* mean, const float* rstd, const float* gamma, const float* sums,
                      void* dv, int relu_mask, double count, int64_t npix, int C, int dtype, void* stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the ResNet stem (U1 row: torchvision-style ResNet18 encoder); backward routes
 * to the first maximum in window scan order, dx (+)= */
int ksmi_maxpool3x3s2_forward(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
int ksmi_maxpool3x3s2_backward(const void* x, const void* dy, void* dx, int accumulate, int B, int H, int W, int C, int dtype, void* stream);
/* training pair: the forward also records the window position (ky * 3 + kx, one byte per output element, [B, Ho, Wo, C]) of the first
 * maximum -- the element torch.nn.MaxPool2d routes the gradient to -- and the backward compares codes instead of re-reading the windows */
int ksmi_maxpool3x3s2_forward_idx(const void* x, void* y, void* idx, int B, int H, int W, int C, int dtype, void* stream);
int ksmi_maxpool3x3s2_backward_idx(const void* idx, const void* dy, void* dx, int accumulate, int B, int H, int W, int C, int dtype, void* stream);
/* y = alpha * [relu](x*scale[c] + shift[c]) (scale = shift = NULL: plain scaled copy): the materialised BatchNorm output of
 * linear_fuse (:563-567) and the 0.1 branch scale of ResidualBlock (:479-481) in the backward pass */
int ksmi_affine(const void* x, const float* scale, const float* shift, void* y, int64_t npix, int C, int relu, float alpha, int dtype,
                void* stream);
/* head output NHWC [B][HW][Cs] -> NCHW fp32, act = 1: sigmoid (:635-639), act = 2: softmax over the C channels (siam_conc.py:93,177),
 * act = 3: log-softmax (siam_diff.py:93,173);
 * adjoint (y = the forward's NCHW output) */
int ksmi_out_to_nchw(const void* x, float* y, int B, int C, int Cs, int64_t HW, int act, int dtype, void* stream);
int ksmi_dout_to_nhwc(const float* dy, const float* y, void* dx, int B, int C, int Cs, int64_t HW, int act, int dtype, void* stream);

/* ---------------------------------------------------------------------------------
 * Token GEMMs (bf16) = nn.Linear forward / input gradient on 128x128 MFMA tiles (csrc/gemm.hip).  `w` is the bf16 mirror of the
 * fp32 parameter (ksmi_cast_bf16 of the arena once per step), row-major [N][K] with row stride w_rs; strides in elements.
 * ------------------------------------------------------------------------------- */
int ksmi_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
/* ConvTranspose2d(k = 2, s = 2) of `up` (models/snunet.py:32-46) as token GEMMs (round 4; bf16, C a multiple of 128): the output pixel
 * block (2y + dy, 2x + dx), dy, dx < 2, of input pixel m = (b, y, x) is the "depth row" {(d, n)} of 4C values = two contiguous runs of
 * 2C elements of the [B, 2H, 2W, C] NHWC tensor, which the LDS-DMA loaders / the epilogue of the token GEMM kernels (csrc/gemm2.hip)
 * address directly: no pixel-shuffle pass, no 2 x 2 window gather.
 *   forward          y_depth[m][(d, n)] = sum_c x[m][c] Wt[c][n][d] + bias[n]           (GEMM NT, shuffled store)
 *   input gradient   dx[m][c] (+)= sum_(d, n) dy_depth[m][(d, n)] Wt[c][n][d]           (GEMM NN, depth-row operand)
 *   weight gradient  dWt[c][n][d] (+)= sum_m x[m][c] dy_depth[m][(d, n)]                (GEMM TN over row splits + permuting reducer)
 * wb = the [4C][C] bf16 image of Wt written by ksmi_up_pack_weight each step; x, y, dy, dx NHWC bf16 ([B,H,W,C] / [B,2H,2W,C]). */
int ksmi_up_gemm_supported(int B, int H, int W, int C, int dtype);
int ksmi_up_wgrad_supported(int B, int H, int W, int C, int dtype);      /* the weight gradient alone also takes C = 64 */
int ksmi_up_pack_weight(const float* wt, void* wb, int C, void* stream);
/* the same pack for up to KSMI_UP_PACK_MAX tensors in one launch (host arrays of n pointers / channel counts) */
#define KSMI_UP_PACK_MAX 16
int ksmi_up_pack_weights_batched(const float* const* wt, void* const* wb, const int* C, int n, void* stream);
int ksmi_up_forward(const void* x, const void* wb, const float* bias, void* y, int B, int H, int W, int C, void* stream);
int ksmi_up_dgrad(const void* dy, const void* wb, void* dx, int accumulate, int B, int H, int W, int C, void* stream);
size_t ksmi_up_wgrad_workspace(int B, int H, int W, int C);
int ksmi_up_wgrad(const void* x, const void* dy, float* workspace, float* grad, int accumulate, int B, int H, int W, int C, void* stream);

/* y[rows][N] = x[rows][K] w^T + bias (+ resid) */
int ksmi_gemm_nt(const void* x, int x_rs, const void* w, int w_rs, const float* bias, const void* resid, int r_rs, void* y, int y_rs,
                 int rows, int K, int N, void* stream);
/* dx[rows][K] (+)= dy[rows][N] w */
int ksmi_gemm_nn(const void* dy, int dy_rs, const void* w, int w_rs, void* dx, int dx_rs, int rows, int K, int N, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------
 * BIT-CD BASE_Transformer, the token path (models/bit_cd.py:802-934).  Pixels are [images][N][32] in `dtype` with
 * images = dates * B in date-major order; everything on the token side (2 * token_len tokens per image pair) is fp32.
 * ------------------------------------------------------------------------------- */
/* Strided batched product, fp32: c[b1][b2][m][n] = alpha * sum_k a[b1][b2][m][k] * b[b1][b2][k][n] + bias[n] (+ c if accumulate).
 * a_strides = element strides of a over {b1, b2, m, k}, b_strides over {b1, b2, k, n}, c_strides over {b1, b2, m, n} (a stride of
 * 0 broadcasts; transposes are strides).  The token-side Linear layers (nn.Linear :469-472, to_qkv :534, to_q/k/v :484-486), the
 * per-head products of Attention (:542,:555) and the folded matrices of ksmi_token_cross_forward are all instances. */
int ksmi_bmm_f32(const float* a, const float* b, const float* bias, float* c, int nb1, int nb2, int M, int N, int K,
                 const int64_t* a_strides, const int64_t* b_strides, const int64_t* c_strides, float alpha, int accumulate, void* stream);
/* y = softmax(scale * x) over rows of n <= 64 values (dots.softmax(dim=-1), :552); dx = scale * y * (dy - sum(y * dy)) */
int ksmi_softmax_rows_f32(const float* x, float* y, int64_t rows, int n, float scale, void* stream);
int ksmi_softmax_rows_backward_f32(const float* y, const float* dy, float* dx, int64_t rows, int n, float scale, void* stream);
/* _forward_semantic_tokens (:857-865) of both dates + the position table (:880-881): tokens[b][date*L + l][c] = pos[date*L + l][c] +
 * sum_n softmax_n(x[date*B+b][n] . wa[l]) x[date*B+b][n][c]; wa = conv_a.weight [L][32]; stats[images][L][2] = {max, sum of exp}
 * for the backward, which adds the gradient into dx (accumulate) and writes dwa_partial[images][L*32] (sum the rows for conv_a). */
int ksmi_semantic_tokens_forward(const void* x, const float* wa, const float* pos, float* tokens, float* stats, int B, int dates, int N,
                                 int C, int L, int dtype, void* stream);
int ksmi_semantic_tokens_backward(const void* x, const float* wa, const float* stats, const float* dtokens, void* dx, float* dwa_partial,
                                  int B, int dates, int N, int C, int L, int accumulate, int dtype, void* stream);
/* One TransformerDecoder attention sub-layer on the pixels (Residual2(PreNorm2(Cross_Attention)), :436-459, :476-524):
 *   y = x + to_out(softmax(scale * to_q(LN(x)) . to_k(LN(m))) to_v(LN(m)))
 * with the token side folded into A[b][date][j][hd][c] = sum_d to_q.weight[hd*D+d][c] * k[b][date*L+j][hd*D+d] and
 * Bv[b][date][j][hd][c] = sum_d to_out.weight[c][hd*D+d] * v[b][date*L+j][hd*D+d] (32 floats per (token, head): the head dimension D
 * does not exist on the pixel side).  The backward updates g (gradient of the residual stream) in place and writes dA, dBv ("="),
 * dgamma / dbeta (LayerNorm over the pixels) and dbo (to_out bias) ("=" or "+="); workspace = ksmi_token_cross_bwd_workspace bytes. */
int ksmi_token_cross_forward(const void* x, const float* gamma, const float* beta, const float* A, const float* Bv, const float* bo, void* y,
                             int B, int dates, int N, int C, int heads, int L, float scale, int dtype, void* stream);
size_t ksmi_token_cross_bwd_workspace(int B, int dates, int N);
int ksmi_token_cross_backward(const void* x, const float* gamma, const float* beta, const float* A, const float* Bv, void* g, float* dA,
                              float* dBv, float* dgamma, float* dbeta, float* dbo, int accumulate_ln, int accumulate_bo, float* workspace,
                              int B, int dates, int N, int C, int heads, int L, float scale, int dtype, void* stream);

/* GPU-side input pipeline: the Dataset's per-tile clamp -> nan_to_num -> Normalize (dataset/Dataset.py:164-168,
 * 193-198) on raw backscatter tiles already in HBM; x, y NCHW fp32 (y may alias x).  clamp_input < 0: Normalize only (the SLC class,
 * dataset/Dataset.py:1081-1085, neither clamps nor replaces NaNs) */
int ksmi_sar_preprocess(const float* x, const float* mean, const float* stdv, float* y, int B, int C, int64_t HW, float clamp_input, void* stream);

/* Host half of N4: reader for the archive's GeoTIFF tiles.  The reference decodes each file in a DataLoader worker with
 * cv2.imread(path, cv2.IMREAD_ANYDEPTH) (dataset/Dataset.py:664-728: MS1_IVV/IVH, SL1_*, SL2_*, MK0_MLU, MK0_MNA) and rioxarray
 * for MK0_DEM (:730-737).  Host-only (no GPU needed): TIFF 6.0 + BigTIFF, both byte orders, strips / tiles, chunky / planar,
 * compression none / LZW / Deflate / PackBits, predictor 1 / 2 / 3, 8..64-bit unsigned / signed / IEEE samples. */
typedef struct ksmi_tiff_info {
  int32_t width, height, bands;
  int32_t bits, sample_format;     /* bits per sample; 1 unsigned, 2 signed, 3 IEEE */
  int32_t compression, predictor;  /* TIFF tag values */
  int32_t tiled, big_endian, bigtiff;
  int32_t has_geo;                 /* bit 0: pixel_scale valid (ModelPixelScale 33550); bit 1: origin valid (ModelTiepoint 33922) */
  int32_t has_nodata;              /* GDAL_NODATA (42113) present */
  double pixel_scale[2];           /* x, y size of a pixel in model units */
  double origin[2], tie_pixel[2];  /* model position `origin` of raster position `tie_pixel` */
  double nodata;
} ksmi_tiff_info;
/* header of the first image of the file */
int ksmi_tiff_info_read(const char* path, ksmi_tiff_info* info);
/* the first image as fp32, band-sequential [bands][height][width] (what cv2 returns for a float32 tile; integer masks are
 * converted exactly); cap_elems = capacity of `out` in elements; info may be NULL */
int ksmi_tiff_read_f32(const char* path, float* out, int64_t cap_elems, ksmi_tiff_info* info);
/* the same in the file's own sample type (host byte order) */
int ksmi_tiff_read_native(const char* path, void* out, int64_t cap_elems, ksmi_tiff_info* info);
/* n single-band H x W tiles decoded by `threads` host threads into out[n][H][W] fp32 (one pinned staging buffer -> one copy to the
 * GPU): the body of Dataset.__getitem__'s file loop for a whole batch.  A tile of another size or band count is an error. */
int ksmi_tile_batch_read(const char* const* paths, int n, float* out, int H, int W, int threads);
/* the same for multi-band tiles (the 4-band SLC products, dataset/Dataset.py:1110-1129): out[n][bands][H][W] */
int ksmi_tile_batch_read_bands(const char* const* paths, int n, float* out, int bands, int H, int W, int threads);
/* DEM gaps: every NaN of each of the n H x W tiles takes the value of the nearest valid pixel (Euclidean, pixel grid), in place:
 * rioxarray's interpolate_na(method="nearest") of dataset/Dataset.py:733-735.  Tiles without NaN (or without a valid pixel) are
 * left as they are. */
int ksmi_tiles_fill_nodata(float* tiles, int n, int H, int W, int threads);

/* plumbing */
int ksmi_fill_zero(void* p, size_t bytes, void* stream);
/* NCHW fp32 -> NHWC dtype and back (tests / debugging only) */
int ksmi_nchw_to_nhwc(const float* x, void* y, int B, int C, int HW, int dtype, void* stream);
int ksmi_nhwc_to_nchw(const void* x, float* y, int B, int C, int HW, int dtype, void* stream);
/* self-tests of the MFMA fragment conventions (used by tests/test_gpu_selftest.py) */
int ksmi_selftest_mma(const void* a, const void* b, float* c, int dtype, void* stream);
int ksmi_selftest_tr16(const uint16_t* in256, uint16_t* out256, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KSMI_H */
