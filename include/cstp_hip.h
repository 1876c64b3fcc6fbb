/*
 * cstp_hip.h -- C ABI of libcstp_hip.so: the MI355X (gfx950) kernels behind the CSTP
 * R(2+1)D-BYOL pre-training step.
 *
 * The reference (KT27-A/CSTP) owns no native code and no FFI: every op below is an ATen
 * call-site of its Python hot path, cited per entry point (paths relative to the
 * reference root).  A maintainer binds this library with ctypes;
 * cstp_amd/_lib.py is that binding.
 *
 * Conventions
 *   - all tensors are fp32, contiguous, NCDHW ([N][C][D][H][W]); 2-D [B][F] tensors are the
 *     D=H=W=1 case.  Labels are int64.
 *   - every pointer is a DEVICE pointer on the current HIP device; `stream` is a hipStream_t
 *     passed as void* (NULL = default stream).  Nothing here allocates, frees, copies to the
 *     host or synchronises: calls only enqueue work (graph-capture safe).  Scratch comes from
 *     the caller through (ws, ws_bytes); query the size with the *_workspace_bytes functions.
 *   - return value: 0 on success, non-zero on error; cstp_last_error() returns a message for
 *     the calling thread.
 *   - PROCESS-GLOBAL mutable state behind this interface (all of it; none is per stream or per call):
 *       * the thread-local error string;
 *       * the mutex-guarded table of tuned tile shapes (cstp_conv3d_autotune / cstp_conv3d_set_tile write it, every
 *         convolution call reads it);
 *       * the GEMM arithmetic selector cstp_gemm_set_split_terms (atomic; 0 = the CSTP_GEMM environment default) -- it picks
 *         the kernel family of EVERY later convolution / linear call of the process, on any stream, and is part of the tile
 *         table's key;
 *       * the deterministic-mode switch cstp_set_deterministic (atomic; ordered slabs instead of float atomics in the weight
 *         gradients), likewise process-wide;
 *       * the per-thread pack mode and record list (cstp_pack_mode) and the process-wide set of registered workspaces
 *         (cstp_pack_register; mutex-guarded);
 *       * environment knobs read once: CSTP_GEMM, CSTP_DETERMINISTIC, CSTP_PERSIST_CUS, CSTP_K1P_QUAD, CSTP_K1W, CSTP_LINEAR,
 *         CSTP_BN_SMALL, CSTP_TILE / CSTP_WTILE (developer overrides).
 *     A caller that wants two arithmetics side by side in one process must serialise the switch with its launches
 *     (bench.py does, between timed loops); results never depend on the state of another STREAM.
 */
#ifndef CSTP_HIP_H
#define CSTP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSTP_ABI_VERSION 18

/* Geometry of one nn.Conv3d(bias=False) call-site.
 * models/pace/r21d_byol.py:81-82 (spatial 1xkxk), :91-92 (temporal tx1x1), :125 (1x1x1 shortcut);
 * nn.Linear (r21d_byol.py:235-241, 249-255, 276-291) is the D=H=W=1, k=1 case. */
typedef struct cstp_conv_desc {
  int32_t n, c, d, h, w; /* input  [n][c][d][h][w]            */
  int32_t k;             /* output channels                    */
  int32_t kt, kh, kw;    /* kernel                             */
  int32_t st, sh, sw;    /* stride                             */
  int32_t pt, ph, pw;    /* zero padding                       */
} cstp_conv_desc;

/* Optional input transform fused into a convolution's gather: z = act(x * scale + shift) with one
 * (scale, shift) pair per (BN group, input channel) -- the train-mode BatchNorm (+ReLU) that precedes the
 * convolution in the reference (r21d_byol.py:94-97 `temporal_conv(relu(bn(spatial_conv(x))))`, :142-143
 * `conv2(relu1(bn1(.)))`) applied on the fly, so the normalised tensor is never written to HBM.
 * scale_shift: float[groups][c][2] as produced by cstp_bn_stats_train; zero padding applies to z. */
typedef struct cstp_in_affine {
  const float* scale_shift;
  int32_t groups; /* 1..4, must divide desc->n */
  int32_t relu;   /* non-zero: z = max(z, 0) */
} cstp_in_affine;

/* 1 when BOTH the forward and the weight gradient of this geometry apply a cstp_in_affine over `groups` BatchNorm groups inside
 * their f16-pair gather kernels (igemm_k1s / igemm_k2s <.., AFF>) given the transformed tensor's absmax cell; 0 = such a call
 * would fall back to the native fp32 kernels (correct, much slower): callers then materialise the BatchNorm output instead. */
int32_t cstp_conv3d_in_affine_fused(const cstp_conv_desc* desc, int32_t groups);

int cstp_abi_version(void);
const char* cstp_last_error(void);

/* ---- convolution (F.conv3d / F.linear and their autograd) ------------------------------- */
/* (F.linear is the D = H = W = 1, 1x1x1 case.  On n <= 32 rows -- the heads, r21d_byol.py:159-176, 318-334 -- forward, data gradient
 *  (reduction length a multiple of 64) and weight gradient run as weight-streaming kernels in EXACT fp32 FMAs, slices summed in a
 *  fixed order (csrc/linear.h); CSTP_LINEAR=0, the native-f32 arithmetic and a pinned tile keep them on the convolution kernels.) */
size_t cstp_conv3d_workspace_bytes(const cstp_conv_desc* desc);
/* y[n][k][do][ho][wo] = conv3d(T(x), w) (+ bias[k] when bias != NULL).  w is [k][c][kt][kh][kw].
 * T = identity when in_affine == NULL, else the fused BN(+ReLU) input transform above.  (The _am variants: with an in_affine the
 * absmax cell is that of T(x), cstp_bn_finalize_pre.) */
int cstp_conv3d_forward(void* stream, const cstp_conv_desc* desc, const float* x, const float* w,
                        const float* bias, const cstp_in_affine* in_affine, float* y, void* ws, size_t ws_bytes);
/* dx = conv3d input gradient (aten::convolution_backward, input mask). */
int cstp_conv3d_backward_data(void* stream, const cstp_conv_desc* desc, const float* dy, const float* w,
                              float* dx, void* ws, size_t ws_bytes);
/* dw = conv3d weight gradient w.r.t. T(x), [k][c][kt][kh][kw] (aten::convolution_backward, weight mask);
 * in_affine as in the forward call (NULL = identity). */
int cstp_conv3d_backward_weight(void* stream, const cstp_conv_desc* desc, const float* x,
                                const cstp_in_affine* in_affine, const float* dy, float* dw, void* ws,
                                size_t ws_bytes);

/* The same three calls with the LARGEST MAGNITUDE of the gathered activation operand(s) supplied by the caller: each
 * *_absmax points to one device uint32 holding the fp32 bit pattern (sign cleared) of max |element| of that tensor -- exactly
 * what cstp_bn_forward_train_am / cstp_bn_backward_am leave behind for the tensor they write.  The 2xf16-split kernels
 * (csrc/igemm_split.h: operands scaled by a power of two into f16 range and split into an f16 pair, three f16 MFMA
 * products per fp32 product) need it for their operand scale; with NULL they measure it themselves with one extra read of
 * the tensor.  Every other kernel variant ignores it.  The value must not be smaller than the true maximum (f16 overflow);
 * a larger value costs one bit of the 22-bit operand precision per factor of two. */
int cstp_conv3d_forward_am(void* stream, const cstp_conv_desc* desc, const float* x, const float* w, const float* bias,
                           const cstp_in_affine* in_affine, float* y, void* ws, size_t ws_bytes, const uint32_t* x_absmax);
/* Forward of a bias-free convolution whose output feeds a train-mode BatchNorm3d over `groups` (1 or 2) equal slices of the batch
 * (r21d_byol.py:82-83: spatial_conv -> bn): where the layer runs the LDS-resident-patch kernel, the launch also leaves the
 * per-channel, per-group sums of y and y^2 as fp64 partials part[k][groups][nsplit][2] (cstp_bn_forward_train_pre folds
 * them: the BatchNorm then needs no statistics pass over y).  *nsplit = partials per (channel, group), or 0 when this layer's
 * kernel cannot deliver them -- the call is then exactly cstp_conv3d_forward_am and BatchNorm takes its usual path.
 * cstp_conv3d_bnstats_nsplit: the same answer in advance (part needs k * groups * nsplit * 3 + k doubles).
 * pivot (float[k] on the device, or NULL = zeros): the sums are taken AROUND it -- sum(y - pivot[ch]), sum((y - pivot[ch])^2) --
 * and the launch stores the values it used in the k doubles behind the sums; any finite values are correct, values near the
 * channel means (the BatchNorm's running_mean) keep the variance free of cancellation when |mean| >> std.
 * Behind the pivots the launch leaves every channel's smallest and largest output per group and partial
 * (uint32 keys [k][groups][nsplit][2]): cstp_bn_finalize_pre turns them into the exact largest magnitude of the normalised
 * tensor, so that the convolution consuming it can apply the BatchNorm in its own gather (cstp_in_affine) with the tensor
 * never written.  z_cell (device uint32, or NULL): zeroed by this launch for that purpose (cstp_bn_finalize_pre takes the
 * maximum into it with atomics).  in_affine (or NULL): the input transform of cstp_conv3d_forward_am, for a temporal
 * convolution that consumes one BatchNorm inside its gather and feeds the next (x_absmax is then that of T(x)). */
int32_t cstp_conv3d_bnstats_nsplit(const cstp_conv_desc* desc, int32_t groups);
/* ... the answer for a call that carries an in_affine (such a forward may run another kernel variant than the plain one:
 * the weight-resident temporal kernel igemm_k1w where it applies, csrc/igemm_twres.h). */
int32_t cstp_conv3d_bnstats_nsplit_aff(const cstp_conv_desc* desc, int32_t groups);
int cstp_conv3d_forward_bnstats(void* stream, const cstp_conv_desc* desc, const float* x, const float* w, float* y, void* ws,
                                size_t ws_bytes, const uint32_t* x_absmax, int32_t groups, const float* pivot, double* part,
                                size_t part_bytes, int32_t* nsplit, uint32_t* z_cell, const cstp_in_affine* in_affine);
int cstp_conv3d_backward_data_am(void* stream, const cstp_conv_desc* desc, const float* dy, const float* w, float* dx,
                                 void* ws, size_t ws_bytes, const uint32_t* dy_absmax);
/* ... and with accumulate != 0: dx += the data gradient instead of dx = -- autograd's sum of the gradients of a tensor that
 * feeds two consumers (the residual connection, r21d_byol.py:141-148: x goes into conv1 AND into the addition behind bn2),
 * folded into the convolution's epilogue: one extra read of dx instead of a separate three-tensor add pass. */
int cstp_conv3d_backward_data_acc(void* stream, const cstp_conv_desc* desc, const float* dy, const float* w, float* dx,
                                  void* ws, size_t ws_bytes, const uint32_t* dy_absmax, int32_t accumulate);
int cstp_conv3d_backward_weight_am(void* stream, const cstp_conv_desc* desc, const float* x, const cstp_in_affine* in_affine,
                                   const float* dy, float* dw, void* ws, size_t ws_bytes, const uint32_t* x_absmax,
                                   const uint32_t* dy_absmax);
/* ... and with accumulate != 0: dw += the weight gradient instead of dw = -- autograd's AccumulateGrad (`.grad += `,
 * main_byol.py:87 loss.backward()) folded into the unpacking pass, for a `.grad` that lives in the flat gradient arena. */
int cstp_conv3d_backward_weight_acc(void* stream, const cstp_conv_desc* desc, const float* x, const cstp_in_affine* in_affine,
                                    const float* dy, float* dw, void* ws, size_t ws_bytes, const uint32_t* x_absmax,
                                    const uint32_t* dy_absmax, int32_t accumulate);

/* Deterministic mode, process-wide (also CSTP_DETERMINISTIC=1 in the environment): the weight-gradient kernels reduce their
 * split-K partial sums through per-split slabs added in a fixed order instead of f32 atomics, so two runs on the same
 * inputs give bit-identical gradients (slower; for debugging -- the default's summation order varies from run to run, rel 1e-6).
 * Switch it BEFORE sizing workspaces: cstp_conv3d_workspace_bytes grows by the slab count. */
int cstp_set_deterministic(int32_t on);
int32_t cstp_get_deterministic(void);

/* Arithmetic of the split kernels (csrc/igemm_split.h), process-wide: 2 = every fp32 operand scaled by a power of two and
 * split into an f16 PAIR, three f16 MFMA products per fp32 product (default; 22 operand bits, measured at least as close to
 * fp64 as the native f32 MFMA chain); 3 = split EXACTLY into a bf16 TRIPLE, six bf16 MFMA products (no operand scaling, no
 * dynamic-range caveat; also selected by CSTP_GEMM=bf16x3 in the environment); 1 = no split kernels at all: every GEMM-shaped
 * op on the native f32 MFMA (bit-for-bit an fmaf chain; also CSTP_GEMM=f32), with its own class of tuned tiles;
 * 0 = back to the environment's choice.  All accumulate in f32.  cstp_gemm_get_split_terms returns 1, 2 or 3. */
int cstp_gemm_set_split_terms(int32_t terms);
int32_t cstp_gemm_get_split_terms(void);

/* The weight packs of a whole network pass from ONE launch.  Every convolution / linear call re-lays its weights out for the
 * kernel variant it runs (a launch of ~10 us per call: ~137 per R(2+1)D-18 training step, main_byol.py:60-91 being ONE Python
 * step).  A caller that keeps a PERSISTENT workspace per (weight tensor, direction) can hoist them:
 *   cstp_pack_mode(1); <the call>; cstp_pack_mode(0);  n = cstp_pack_recorded(recs, cap);     -- once: what the call packs
 *   cstp_pack_replay(stream, recs_dev, first_block_dev, n, total_blocks);                     -- per pass: all recorded packs
 *   cstp_pack_mode(2); <the same call, same pointers>; cstp_pack_mode(0);                     -- the call skips its pack
 * A record is the pack launch itself (kind 1: f16-pair split rows, 2: LDS-patch K-tiles, 3: native fp32 re-layout, 4: the
 * bf16-storage path's operand rows; pointers,
 * integer arguments, blocks of 256 threads), so the replay does exactly what the call would have done -- provided descriptor,
 * tile table, weight pointer and workspace pointer are those of the recorded call.  first_block_dev[i] = sum of nblocks of the
 * records before i; total_blocks = the sum over all.  The mode is per calling THREAD (0 = default).  bf16-triple packs are not
 * recorded (they keep packing inside the call in every mode). */
typedef struct cstp_pack_rec {
  int32_t kind, nblocks;
  const float* w;
  void* dst;
  float* inv_a;
  uint32_t* cells;
  int32_t a[10];
} cstp_pack_rec;
int cstp_pack_mode(int32_t mode);
int32_t cstp_pack_recorded(cstp_pack_rec* out, int32_t cap);
int cstp_pack_replay(void* stream, const cstp_pack_rec* recs_dev, const int32_t* first_block_dev, int32_t n, int32_t total_blocks);
/* ... or, instead of mode 2 around every call: REGISTER the workspaces (the records' dst pointers) whose packs the caller replays --
 * a call that receives a registered workspace skips its pack (on = 0: unregister; n = 0 with on = 0: forget all).  Process-wide. */
int cstp_pack_register(const void* const* workspaces, int32_t n, int32_t on);

/* Which kernel variant the next forward (mode 0) / backward_data (mode 1) / backward_weight (mode 2) call with this
 * descriptor will run: out[0] = rows per block tile, out[1] = positions (mode 2: (tap, channel) columns) per block tile,
 * out[2] = 0 for the native f32 MFMA kernel, else the number of terms each operand is split into by the split kernel (2 or
 * 3, see cstp_gemm_set_split_terms), out[3] = K-tiles per barrier (mode 2: split-K block target / 256).  mode 3 = a forward
 * call that carries an in_affine (it may run another variant than the plain forward: cstp_conv3d_bnstats_nsplit_aff).
 * Reporting only (bench.py names the kernels and picks their roofline peaks with it). */
int cstp_conv3d_query_tile(const cstp_conv_desc* desc, int32_t mode, int32_t* out4);

/* The TUNED entry of one geometry and direction in cstp_conv3d_set_tile's encoding (tile4[0] = -1 when the geometry has
 * not been tuned or pinned in the current arithmetic class): lets the host persist a tuning table and replay it with
 * cstp_conv3d_set_tile, so that every rank and every run executes the same kernels (cstp_amd/tuned/). */
int cstp_conv3d_get_tile(const cstp_conv_desc* desc, int32_t mode, int32_t* tile4);

/* Pin the kernel variant of one geometry and direction (what cstp_conv3d_autotune would otherwise decide by timing):
 * mode 0 forward / 1 backward_data: tile[0] = 1 for the split kernel (arithmetic: cstp_gemm_set_split_terms; tile[1] = row tiles of 16: 2,3,4,5,6,8,9;
 * tile[2] = 2 selects the 256-column tile, available with 8 / 9 row tiles) or 0
 * for the native f32 kernel (tile[1] = row tiles of 32: 1..5, or 9 = the 144-row tile of nine 16-row MFMA tiles, tile[2] = waves
 * along rows 1|2|4, tile[3] = K-tiles per barrier 1|2); mode 2 backward_weight: tile[0] = 1 split (tile[1] = 4|8|9 row tiles of 16) or 0 native (tile[1] = 1..5 row tiles of
 * 32, or 9 = the 144-row tile), tile[2] = split-K block target / 256 (4, 8, 16).  Inadmissible requests (3-channel stem, > 27
 * taps, >= 2 GiB tensors for the split kernels) fall back to a native tile at call time.  Used by the parity tests to put every
 * kernel variant against the fp64 reference regardless of which one is fastest on the machine at hand. */
int cstp_conv3d_set_tile(const cstp_conv_desc* desc, int32_t mode, const int32_t* tile4);

/* Optional one-off tuning, OUTSIDE graph capture: times the tile shapes of the forward (mode 0: src = x, w = weights,
 * out = y), data-gradient (mode 1: src = dy, w = weights, out = dx) or weight-gradient (mode 2: src = x, w = dy,
 * out = dw; row-tile height x split-K block count) kernel for this geometry on `stream` (this call DOES synchronise)
 * and remembers the fastest in a process-wide table that later calls with the same descriptor consult.  The native
 * f32 tiles give bit-identical results among themselves; the split tiles (candidates unless the environment says
 * CSTP_GEMM=f32) differ from them in the last bits (both sit ~4e-7 rms from the fp64 result per convolution,
 * tools/split_accuracy.py).  Without tuning an analytic native-f32 choice is used.  `out` is overwritten. */
int cstp_conv3d_autotune(void* stream, const cstp_conv_desc* desc, int32_t mode, const float* src, const float* w,
                         float* out, void* ws, size_t ws_bytes, int32_t iters);

/* ---- BatchNorm3d / BatchNorm1d in TRAIN mode, fused with the residual add and ReLU that follow
 *      it (r21d_byol.py:83-84,133-134,138-139,148,199-200,216; Projector/Predictor/heads BN1d).
 *      x,y,residual: [n][c][s] (s = D*H*W).  y = act(bn(x) + residual), act = relu if relu != 0.
 *      `groups`: the batch holds that many independent BN calls back to back (n = groups * n_per_group,
 *      e.g. the two views of a clip pair, r21d_byol.py:359-360): statistics are per (group, channel),
 *      running stats are updated group after group as successive F.batch_norm(training=True) calls would.
 *      save_mean/save_invstd: [groups][c] outputs for backward.  running_mean/var updated in place with
 *      `momentum` (unbiased variance).  scale_shift (optional, s > 1): float[groups][c][2] affine form of the
 *      normalisation; handing it to cstp_bn_backward lets the ReLU mask be recomputed from x instead of
 *      re-reading y (two fewer HBM passes in the backward). */
size_t cstp_bn_workspace_bytes(int32_t n, int32_t c, int32_t s, int32_t groups);
int cstp_bn_forward_train(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, float* save_mean,
                          float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s, int32_t groups,
                          float eps, float momentum, int32_t relu, void* ws, size_t ws_bytes);
/* ... and, as a by-product of the pass that writes y (s > 1 only), the largest magnitude of y into *y_absmax (fp32 bits,
 * sign cleared; NULL = not wanted) for the consumer convolution (cstp_conv3d_forward_am / _backward_weight_am). */
int cstp_bn_forward_train_am(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float* save_mean,
                             float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s, int32_t groups,
                             float eps, float momentum, int32_t relu, void* ws, size_t ws_bytes, uint32_t* y_absmax);
/* ... with the statistics pass replaced by the partial sums a producing convolution left (cstp_conv3d_forward_bnstats;
 * part[c][groups][nsplit][2] fp64 sums of (x - pivot) and (x - pivot)^2, then pivot[c]; s > 1).  Everything else as
 * cstp_bn_forward_train_am. */
int cstp_bn_forward_train_pre(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float* save_mean,
                              float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s, int32_t groups,
                              float eps, float momentum, int32_t relu, void* ws, size_t ws_bytes, uint32_t* y_absmax,
                              const double* part, int32_t nsplit);
/* The finalize step of cstp_bn_forward_train_pre ALONE (no pass over x, no output tensor): save_mean / save_invstd, the running
 * statistics, the affine table scale_shift[groups][c][2] and -- from the minima / maxima cstp_conv3d_forward_bnstats left behind
 * its sums -- the largest magnitude of act(x * scale + shift) over the whole tensor, bit-identical to what the apply pass of
 * cstp_bn_forward_train_pre would have measured, taken into *z_cell with atomicMax (the cell must hold 0 or a lower bound: the
 * producing launch zeroes the one it was given).  count = n * s / groups values per channel and group. */
int cstp_bn_finalize_pre(void* stream, const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float* save_mean, float* save_invstd, float* scale_shift, int32_t n, int32_t c, int32_t s,
                         int32_t groups, float eps, float momentum, int32_t relu, const double* part, int32_t nsplit,
                         uint32_t* z_cell);
/* Statistics only: save_mean/save_invstd [groups][c], running stats update, and the affine table
 * scale_shift float[groups][c][2] = (invstd*gamma, beta - mean*invstd*gamma) that a consumer convolution
 * applies in its gather (cstp_in_affine) -- the BN output itself is never materialised. */
int cstp_bn_stats_train(void* stream, const float* x, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float* save_mean, float* save_invstd, float* scale_shift, int32_t n,
                        int32_t c, int32_t s, int32_t groups, float eps, float momentum, void* ws, size_t ws_bytes);
/* Backward of the fused op.  y is the forward OUTPUT (its sign is the ReLU mask); when the output was
 * never materialised (cstp_bn_stats_train + cstp_in_affine) pass y = NULL and the scale_shift table and the
 * mask is recomputed from x.  dresidual may be NULL.  dgamma/dbeta: [c], summed over the groups. */
int cstp_bn_backward(void* stream, const float* x, const float* y, const float* dy, const float* gamma,
                     const float* save_mean, const float* save_invstd, const float* scale_shift, float* dx,
                     float* dresidual, float* dgamma, float* dbeta, int32_t n, int32_t c, int32_t s, int32_t groups,
                     int32_t relu, void* ws, size_t ws_bytes);

/* ... with the largest magnitude of dx as a by-product (the dY operand of the producing convolution's data / weight
 * gradient, cstp_conv3d_backward_data_am / _backward_weight_am); s > 1 only, NULL = not wanted.  accumulate != 0: dgamma and
 * dbeta are ADDED to the buffers (the parameters' gradient slices of a flat arena: no separate accumulation kernel). */
int cstp_bn_backward_am(void* stream, const float* x, const float* y, const float* dy, const float* gamma,
                        const float* save_mean, const float* save_invstd, const float* scale_shift, float* dx,
                        float* dresidual, float* dgamma, float* dbeta, int32_t n, int32_t c, int32_t s, int32_t groups,
                        int32_t relu, void* ws, size_t ws_bytes, uint32_t* dx_absmax, int32_t accumulate);

/* EVAL mode (model.eval(): main_ft_mp.py:254-262 validation, test.py:74-76): the running statistics are the
 * statistics -- y = act((x - running_mean) / sqrt(running_var + eps) * gamma + beta + residual); nothing is updated.
 * Forward only (the reference evaluates under torch.no_grad()).  ws: cstp_bn_eval_workspace_bytes(c) when s > 1. */
size_t cstp_bn_eval_workspace_bytes(int32_t c);
int cstp_bn_forward_eval(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                         const float* beta, const float* running_mean, const float* running_var, int32_t n, int32_t c,
                         int32_t s, float eps, int32_t relu, void* ws, size_t ws_bytes);
/* ... with max |y| as a by-product (s > 1; see cstp_bn_forward_train_am); ws then is cstp_bn_workspace_bytes(n, c, s, 1). */
int cstp_bn_forward_eval_am(void* stream, const float* x, const float* residual, float* y, const float* gamma,
                            const float* beta, const float* running_mean, const float* running_var, int32_t n, int32_t c,
                            int32_t s, float eps, int32_t relu, void* ws, size_t ws_bytes, uint32_t* y_absmax);

/* ---- AdaptiveAvgPool3d(1) (r21d_byol.py:210,222-223) and its backward ------------------- */
int cstp_avgpool_forward(void* stream, const float* x, float* y, int32_t rows, int32_t s);
int cstp_avgpool_backward(void* stream, const float* dy, float* dx, int32_t rows, int32_t s);
/* nn.MaxPool3d (models/BE/r3d_byol.py:158,197: kernel 3, stride 2, padding 1 after the stem of the 3D-ResNet backbone; any
 * kernel / stride / padding with 2*pad <= kernel here).  x: [rows = n*c][d][h][w] -> y: [rows][do][ho][wo]; argmax (int32,
 * same shape as y) records the flat in-plane index of the first maximum in (d, h, w) scan order, as aten's kernel keeps it.
 * backward gathers: dx[p] = sum of dy over the windows whose argmax is p (deterministic, no atomics). */
int cstp_maxpool3d_forward(void* stream, const float* x, float* y, int32_t* argmax, int32_t rows, int32_t d, int32_t h,
                           int32_t w, const int32_t* kernel3, const int32_t* stride3, const int32_t* pad3);
int cstp_maxpool3d_backward(void* stream, const float* dy, const int32_t* argmax, float* dx, int32_t rows, int32_t d,
                            int32_t h, int32_t w, const int32_t* kernel3, const int32_t* stride3, const int32_t* pad3);
/* out[c] = sum over n,s of x[n][c][s]  (bias gradient of nn.Linear). */
int cstp_channel_sum(void* stream, const float* x, float* out, int32_t n, int32_t c, int32_t s, void* ws,
                     size_t ws_bytes);

/* ---- loss heads ------------------------------------------------------------------------- */
/* BYOL regression loss r21d_byol.py:346-349: loss[b] = 2 - 2*<x/|x|, y/|y|>, eps 1e-12; x,y [b][f].
 * backward writes dx only (the target branch is detached, r21d_byol.py:368-369). */
int cstp_byol_loss_forward(void* stream, const float* x, const float* y, float* loss, int32_t b, int32_t f);
int cstp_byol_loss_backward(void* stream, const float* x, const float* y, const float* dloss, float* dx,
                            int32_t b, int32_t f);
/* F.normalize(x, p=2, dim=1) of the fine-tune/test head (r21d_byol.py:396): y = x / max(|x|_2, eps), x,y [rows][f];
 * norm[rows] = max(|x|, eps) is kept for the backward: dx = (dy - y <y, dy>) / norm. */
int cstp_l2_normalize_forward(void* stream, const float* x, float* y, float* norm, int32_t rows, int32_t f, float eps);
int cstp_l2_normalize_backward(void* stream, const float* y, const float* norm, const float* dy, float* dx, int32_t rows,
                               int32_t f, float eps);
/* nn.CrossEntropyLoss() (mean) main_byol.py:63-68: logits [b][k], labels int64 [b] -> loss[1].
 * backward: dlogits = dloss[0] * (softmax - onehot) / b. */
int cstp_cross_entropy_forward(void* stream, const float* logits, const int64_t* labels, float* loss, int32_t b,
                               int32_t k);
int cstp_cross_entropy_backward(void* stream, const float* logits, const int64_t* labels, const float* dloss,
                                float* dlogits, int32_t b, int32_t k);
/* NT-Xent loss/NTXent.py:46-62 on reps = cat(zjs, zis) [2n][f] (cosine similarity, eps 1e-8).
 * forward needs ws of cstp_ntxent_workspace_bytes (the 2n x 2n similarity matrix + norms);
 * backward reuses that ws and writes dreps [2n][f]. */
size_t cstp_ntxent_workspace_bytes(int32_t two_n, int32_t f);
int cstp_ntxent_forward(void* stream, const float* reps, float* loss, int32_t two_n, int32_t f, float temperature,
                        void* ws, size_t ws_bytes);
int cstp_ntxent_backward(void* stream, const float* reps, const float* dloss, float* dreps, int32_t two_n,
                         int32_t f, float temperature, void* ws, size_t ws_bytes);

/* ---- GPU clip assembly (data path next to the step: data_process/datasets.py:876-948, preprocess_data.py:479-581,1103-1110) ----
 * frames: decoded video, uint8 [f][h][w][3] in HBM.  For each of the t frames frame_idx[i] (device int32): Image.transpose(rot in
 * {0, 90, 180, 270}, counter-clockwise) -> Image.crop at (box_x0, box_y0) of the ROTATED frame -> Image.resize((size, size),
 * Image.BICUBIC) -> FLIP_LEFT_RIGHT if flip -> ToTensor -> x*2-1 clamped  =>  out fp32 [3][t][size][size].
 * The resize is Pillow's fixed-point algorithm bit for bit; its integer coefficient tables come from the caller (kh/kv int32
 * [size][ks*], bh/bv int32 [size][2] = (first tap, tap count), computed as Resample.c's precompute_coeffs +
 * normalize_coeffs_8bpc do for the crop's width / height -- cstp_amd/clip_ops.py); the horizontal pass covers the crop rows
 * [row_first, row_first + rows) that the vertical pass reads and writes them to tmp (uint8 [t][rows][size][3]). */
int cstp_clip_assemble(void* stream, const uint8_t* frames, int32_t f, int32_t h, int32_t w, const int32_t* frame_idx, int32_t t,
                       int32_t rot, int32_t box_x0, int32_t box_y0, int32_t size, int32_t flip, const int32_t* kh,
                       const int32_t* bh, int32_t ksh, const int32_t* kv, const int32_t* bv, int32_t ksv, int32_t row_first,
                       int32_t rows, uint8_t* tmp, float* out);

/* ... the same up to the resize, result kept as 8-bit RGB [t][size][size][3] (no flip, no normalisation): the input of the
 * base_transform operations below. */
int cstp_clip_assemble_u8(void* stream, const uint8_t* frames, int32_t f, int32_t h, int32_t w, const int32_t* frame_idx,
                          int32_t t, int32_t rot, int32_t box_x0, int32_t box_y0, int32_t size, const int32_t* kh,
                          const int32_t* bh, int32_t ksh, const int32_t* kv, const int32_t* bv, int32_t ksv, int32_t row_first,
                          int32_t rows, uint8_t* tmp, uint8_t* out);
/* ---- the `base_transform` branch of the clip pipeline (data_process/preprocess_data.py:1110-1121; taken with p = 0.3 per clip,
 * TwoClipTransform :713-741) on 8-bit RGB clips [t][h][w][3] in HBM.  Each entry reproduces, bit for bit, the Pillow call the
 * reference makes (directly, or through torchvision's PIL backend):
 *   cstp_clip_rotate    RandomRotation(10) :1060-1100 -> Image.rotate(angle): NEAREST, same size, black fill.  coef6 (HOST
 *                       pointer) = Geometry.c affine_fixed's six 16.16 fixed-point coefficients of Image.rotate's reverse matrix.
 *   cstp_clip_blend     ClipColorJitter :584-672 -> adjust_brightness (mode 0), adjust_contrast (1; ws_means: t int32 on the
 *                       device), adjust_saturation (2) = ImageEnhance.{Brightness, Contrast, Color}.enhance(alpha) = Image.blend.
 *   cstp_clip_hue       adjust_hue: RGB -> HSV, h += shift (= uint8(hue_factor * 255)) modulo 256, HSV -> RGB; mode 1 / 2: the
 *                       RGB -> HSV / HSV -> RGB conversion alone.  In place allowed.
 *   cstp_clip_gray      ClipRandomGray.grayscale :704-709: frame i keeps channel[i] (device int32, < 0 = unchanged) in all three.
 *   cstp_clip_box_blur  ClipGaussianBlur :675-687 -> ImageFilter.GaussianBlur(radius) = `passes` (3) extended box blurs per axis
 *                       (BoxBlur.c); radius / ww / fw = integer box radius and its two 24-bit fixed-point weights, computed by
 *                       the caller as BoxBlur.c does from the float box radius.  In place (tmp: same size scratch).
 *   cstp_clip_finish    [FLIP_LEFT_RIGHT] -> ToTensor -> x * 2 - 1 clamped: uint8 [t][h][w][3] -> fp32 [3][t][h][w]. */
int cstp_clip_rotate(void* stream, const uint8_t* src, uint8_t* dst, int32_t t, int32_t h, int32_t w, const int32_t* coef6);
int cstp_clip_blend(void* stream, const uint8_t* src, uint8_t* dst, int32_t t, int32_t h, int32_t w, int32_t mode, float alpha,
                    int32_t* ws_means);
int cstp_clip_hue(void* stream, const uint8_t* src, uint8_t* dst, size_t npix, int32_t shift, int32_t mode);
int cstp_clip_gray(void* stream, const uint8_t* src, uint8_t* dst, int32_t t, int32_t h, int32_t w, const int32_t* channel);
int cstp_clip_box_blur(void* stream, uint8_t* img, uint8_t* tmp, int32_t t, int32_t h, int32_t w, int32_t radius, uint32_t ww,
                       uint32_t fw, int32_t passes);
int cstp_clip_finish(void* stream, const uint8_t* src, float* out, int32_t t, int32_t h, int32_t w, int32_t flip);

/* ---- per-step utilities over FLAT parameter arenas -------------------------------------- */
/* EMA r21d_byol.py:331-337: target = target*m + online*(1-m) over n floats. */
int cstp_ema_update(void* stream, float* target, const float* online, size_t n, double m);
/* out[0] = sum(g^2) over n floats (for clip_grad_norm_, main_byol.py:88-90); ws >= 8 KiB. */
int cstp_sumsq(void* stream, const float* g, size_t n, float* out, void* ws, size_t ws_bytes);
/* coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)); norm_out[0] = sqrt(sumsq[0]). */
int cstp_clip_coef(void* stream, const float* sumsq, float max_norm, float* coef, float* norm_out);
/* torch.optim.SGD step (main_byol.py:91,228-232; momentum, weight decay, no nesterov, dampening 0):
 *   g' = g*coef[0] (coef may be NULL); if write_back_grad: g = g';  g' += wd*p;
 *   buf = first_step ? g' : momentum*buf + g';  p -= lr[0]*buf.   lr is a DEVICE scalar. */
int cstp_sgd_step(void* stream, float* p, float* g, float* buf, size_t n, const float* lr, float momentum,
                  float weight_decay, const float* coef, int32_t first_step, int32_t write_back_grad);
/* torch.optim.Adam (decoupled = 0: g += wd*p) / AdamW (decoupled = 1: p *= 1 - lr*wd), no amsgrad
 * (main_ft_mp.py:139-147): m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
 * p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).  `step` counts from 1; lr is a DEVICE scalar. */
int cstp_adam_step(void* stream, float* p, const float* g, float* exp_avg, float* exp_avg_sq, size_t n, const float* lr,
                   float beta1, float beta2, float eps, float weight_decay, int32_t decoupled, int32_t step);

/* ---- the bf16-STORAGE path (BASELINE configs[4]: 3D-ResNet-50 backbone swap, bf16) --------------------------------
 * The ATen call-sites of models/BE/r3d_byol.py under bf16 activations (what torch.autocast(bfloat16) makes of them): every
 * 5-D activation tensor and every gradient of one is bf16 in HBM (uint16_t* below = raw bf16 bits, NCDHW, contiguous), all
 * arithmetic between a load and a store is fp32 (fp64 in the BatchNorm reductions), each stored value is rounded to
 * nearest-even once; weights, weight gradients, BatchNorm parameters / statistics stay fp32; convolution operands are the
 * bf16 activations and the fp32 weights rounded to bf16 at use (v_mfma_f32_16x16x32_bf16, fp32 accumulation).
 * Geometry limits: channel counts that are multiples of 16 with at most 27 taps (every layer of the network but the stem), or
 * any channel count with taps * channels <= 1056 (the 3-channel 7x7x7 stem, forward and weight gradient only); every
 * gathered tensor < 2 GiB.  Anything else is refused with an error, not emulated. */
/* x.to(torch.bfloat16) of the clip (r3d_byol.py:193-194 under autocast): n floats -> n bf16, round to nearest even. */
int cstp_b16_cast(void* stream, const float* x, uint16_t* y, size_t n);
size_t cstp_b16_conv3d_workspace_bytes(const cstp_conv_desc* desc);
/* F.conv3d(x, w) of r3d_byol.py:45-53 (conv3x3x3), :104-111 (Bottleneck 1x1x1 / 3x3x3 / 1x1x1), :150-152 (7x7x7 stem),
 * :177-180 (1x1x1 shortcut): y[n][k][do][ho][wo] bf16. */
int cstp_b16_conv3d_forward(void* stream, const cstp_conv_desc* desc, const uint16_t* x, const float* w, uint16_t* y, void* ws,
                            size_t ws_bytes);
/* its autograd backward w.r.t. the input: dx[n][c][d][h][w] bf16 (every element written). */
int cstp_b16_conv3d_backward_data(void* stream, const cstp_conv_desc* desc, const uint16_t* dy, const float* w, uint16_t* dx,
                                  void* ws, size_t ws_bytes);
/* ... accumulate != 0: dx = bf16(bf16(gradient) + dx) -- the second gradient of a tensor that feeds two ops (a residual
 * connection) added in the epilogue instead of by a separate add pass; the rounding points are those of the add it replaces. */
int cstp_b16_conv3d_backward_data_acc(void* stream, const cstp_conv_desc* desc, const uint16_t* dy, const float* w, uint16_t* dx,
                                  void* ws, size_t ws_bytes, int32_t accumulate);
/* ... and w.r.t. the weight: dw[k][c][kt][kh][kw] fp32, accumulate != 0: dw += (position splits through fp32 slabs summed in
 * a fixed order: bit-reproducible). */
int cstp_b16_conv3d_backward_weight(void* stream, const cstp_conv_desc* desc, const uint16_t* x, const uint16_t* dy, float* dw,
                                    void* ws, size_t ws_bytes, int32_t accumulate);
size_t cstp_b16_bn_workspace_bytes(int32_t n, int32_t c, int32_t s, int32_t groups);
/* relu(bn(x) + residual) in train mode (r3d_byol.py:84-95, :117-135, :153-154, :181): as cstp_bn_forward_train on bf16 x /
 * residual / y; scale_shift (float[groups][c][2], required) receives the affine form the apply pass uses. */
int cstp_b16_bn_forward_train(void* stream, const uint16_t* x, const uint16_t* residual, uint16_t* y, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                              float* scale_shift, int32_t n, int32_t c, int32_t s, int32_t groups, float eps, float momentum,
                              int32_t relu, void* ws, size_t ws_bytes);
/* its backward: as cstp_bn_backward_am (y == NULL with relu: the mask is recomputed from x and scale_shift). */
int cstp_b16_bn_backward(void* stream, const uint16_t* x, const uint16_t* y, const uint16_t* dy, const float* gamma,
                         const float* save_mean, const float* save_invstd, const float* scale_shift, uint16_t* dx,
                         uint16_t* dresidual, float* dgamma, float* dbeta, int32_t n, int32_t c, int32_t s, int32_t groups,
                         int32_t relu, void* ws, size_t ws_bytes, int32_t accumulate);
/* every BatchNorm3d under model.eval() (main_ft_mp.py:261-262, test.py:75-81) on bf16: as cstp_bn_forward_eval. */
int cstp_b16_bn_forward_eval(void* stream, const uint16_t* x, const uint16_t* residual, uint16_t* y, const float* gamma,
                             const float* beta, const float* running_mean, const float* running_var, int32_t n, int32_t c,
                             int32_t s, float eps, int32_t relu);
/* nn.MaxPool3d r3d_byol.py:158 on bf16 (same tie rule and argmax as cstp_maxpool3d_forward). */
int cstp_b16_maxpool3d_forward(void* stream, const uint16_t* x, uint16_t* y, int32_t* argmax, int32_t rows, int32_t d, int32_t h,
                               int32_t w, const int32_t* kernel3, const int32_t* stride3, const int32_t* pad3);
int cstp_b16_maxpool3d_backward(void* stream, const uint16_t* dy, const int32_t* argmax, uint16_t* dx, int32_t rows, int32_t d,
                                int32_t h, int32_t w, const int32_t* kernel3, const int32_t* stride3, const int32_t* pad3);
/* nn.AdaptiveAvgPool3d(1) r3d_byol.py:203: bf16 rows -> fp32 means; backward: fp32 dy -> bf16 dx. */
int cstp_b16_avgpool_forward(void* stream, const uint16_t* x, float* y, int32_t rows, int32_t s);
int cstp_b16_avgpool_backward(void* stream, const float* dy, uint16_t* dx, int32_t rows, int32_t s);

#ifdef __cplusplus
}
#endif
#endif /* CSTP_HIP_H */
