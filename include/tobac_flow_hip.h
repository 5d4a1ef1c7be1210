/* tobac_flow_hip.h -- C ABI of the MI355X (gfx950) hot path of tobac-flow.
 *
 * One shared library, libtobac_flow_hip.so, built from the .hip sources in tobac_flow_amd/csrc/.  Every entry
 * point is `extern "C"`, takes plain pointers and sizes, never retains a pointer past return
 * and never allocates device memory behind the caller's back: scratch comes from a caller
 * supplied workspace whose size is given by the matching *_workspace_bytes() query.
 *
 * All data pointers are DEVICE pointers (HBM) unless the name ends in `_host`.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous with respect to
 * the host unless stated otherwise.
 *
 * Return value: 0 on success, a negative TF_E* code on failure; tf_last_error() returns a
 * thread-local message.  The Python host layer (tobac_flow_amd/) validates arguments first and
 * raises the same exception types as the reference (ValueError / AssertionError / ...).
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * /root/reference/).  How a maintainer binds them from the reference is shown in INTEGRATION.md.
 */
#ifndef TOBAC_FLOW_HIP_H
#define TOBAC_FLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TF_OK 0
#define TF_EINVAL (-1)   /* bad argument (shape, enum, null pointer)          */
#define TF_ENOMEM (-2)   /* workspace too small                               */
#define TF_EHIP (-3)     /* a HIP runtime call or kernel launch failed        */
#define TF_ENOCONV (-4)  /* an iterative kernel hit its sweep limit           */
#define TF_EDEPTH (-5)   /* tf_watershed*: ties left at the deepest chain level the workspace allows; output written,
                            but not guaranteed to equal the reference at the reported pixels */
#define TF_ESTARVED (-6) /* tf_farneback*: a row-sum chain of the iteration kernel gave up waiting for its left neighbour
                            (a stalled device); the flow of the launches concerned holds NaN rows -- recompute */

/* interpolation of the semi-Lagrangian gathers: cv2.INTER_NEAREST / _LINEAR / _CUBIC as selected
 * by tobac_flow/convolve.py:47-54 and tobac_flow/utils/flow_utils.py:22-34 */
#define TF_INTERP_NEAREST 0
#define TF_INTERP_LINEAR 1
#define TF_INTERP_CUBIC 2
#define TF_INTERP_LANCZOS 3   /* cv2.INTER_LANCZOS4: 8 x 8 taps (convolve.py:47-54 "lanczos") */

/* element types of `data` / `out` arguments */
#define TF_F32 0
#define TF_F64 1
#define TF_I32 2

/* reductions over the gathered (n_struct, H, W) stack, i.e. the `func` argument of
 * tobac_flow/convolve.py:248-348 for the callables the reference itself passes */
#define TF_FUNC_STACK 0        /* func=None: return the whole stack (n_struct, T, H, W)            */
#define TF_FUNC_SOBEL 1        /* sobel.py:66-86  _sobel_func                                      */
#define TF_FUNC_SOBEL_UPHILL 2 /* sobel.py:32-46  _sobel_func_uphill                               */
#define TF_FUNC_SOBEL_DOWNHILL 3 /* sobel.py:49-63 _sobel_func_downhill                            */
#define TF_FUNC_NANMEAN 4      /* detection.py:53-55,190-195  lambda x: np.nanmean(x, 0)           */
#define TF_FUNC_DIFF 5         /* flow.py:180-184  centred semi-Lagrangian time difference         */
#define TF_FUNC_ANY 6          /* detection.py:313-320  partial(np.any, axis=0) on int32           */
#define TF_FUNC_NANMAX 7       /* detection.py:57-59 (commented-out variant kept for completeness) */

int tf_version(void);
const char *tf_last_error(void);
/* number of visible HIP devices, or a negative TF_E* code; never initialises a context */
int tf_device_count(void);

/* ---- a3: to_8bit(linear_norm(data[i:i+2]), 0, 1) --------------------------------------------
 * replaces tobac_flow/utils/normalisation_utils.py:10-33 (to_8bit) composed with :59-72
 * (linear_norm) exactly as tobac_flow/flow.py:411-414 calls them: joint nanmin/nanmax over the
 * frame pair, scale to [0,1], clip, *255, non-finite -> 127 then patched from the other frame,
 * truncate to uint8.  ws: >= tf_to8bit_workspace_bytes() bytes. */
size_t tf_to8bit_workspace_bytes(int64_t H, int64_t W);
int tf_to8bit_pair(const float *frame0, const float *frame1, int64_t H, int64_t W,
                   uint8_t *out0, uint8_t *out1, void *ws, size_t ws_bytes, void *stream);

/* ---- a5: dense Farnebaeck flow for one frame pair, BOTH directions ---------------------------
 * replaces cv2.optflow.createOptFlow_Farneback().calc(prev, next, None) and
 * .calc(next, prev, None) as issued by tobac_flow/flow.py:511,516 (model factory
 * tobac_flow/utils/flow_utils.py:52-53; OpenCV defaults numLevels=5, pyrScale=0.5, winSize=13,
 * numIters=10, polyN=5, polySigma=1.1, flags=0).  The two directions share the Gaussian pyramid
 * and the polynomial expansion of both images.  flow_fwd / flow_bwd: (H, W, 2) float, (dx, dy).
 * Either output may be NULL to skip that direction. */
typedef struct {
    int num_levels;     /* 5   */
    double pyr_scale;   /* 0.5 */
    int win_size;       /* 13  */
    int num_iters;      /* 10  */
    int poly_n;         /* 5   */
    double poly_sigma;  /* 1.1 */
    int chain_form;     /* TF_FB_CHAIN_*: scheduling of the iteration kernel for THIS call; the flows are the same bits */
    int status_slot;    /* 0 = the device's shared status word; > 0 = a word of this caller's own (tf_farneback_status_acquire) */
} tf_farneback_params;
/* chain_form: the row-sum chains of the iteration kernel as one lane per chain (leaves LDS for the kernels of other streams,
 * e.g. floods of finished windows: what a caller that runs other work beside the flow wants) or in two parts one row group
 * apart (39 KB of LDS per workgroup, a CU is full: 8 % faster when the flow has the GPU to itself).  DEFAULT = one lane.
 * TF_FBI_TWO_PART_CHAIN=0 / 1 in the environment overrides it (development switch). */
#define TF_FB_CHAIN_DEFAULT 0
#define TF_FB_CHAIN_ONE_LANE 1
#define TF_FB_CHAIN_TWO_PART 2
void tf_farneback_default_params(tf_farneback_params *p);
size_t tf_farneback_workspace_bytes(int64_t H, int64_t W, const tf_farneback_params *p);
int tf_farneback_pair(const uint8_t *prev, const uint8_t *next, int64_t H, int64_t W,
                      const tf_farneback_params *p, float *flow_fwd, float *flow_bwd,
                      void *ws, size_t ws_bytes, void *stream);
/* Diagnostic / test entry: the 3 x 3 Gaussian of a uint8 frame and its polynomial expansion exactly as the full-resolution
 * pyramid level computes them (cv2's FarnebackPolyExp on the blurred frame), for stage-by-stage comparison with the oracle.
 * blur_out: H * W floats or NULL; R_out: 5 * H * W floats = H * W float4 {r0, r1, r2, r3} + one plane r4 (OpenCV's five
 * interleaved channels, in its order). */
int tf_farneback_expansion(const uint8_t *img, int64_t H, int64_t W, const tf_farneback_params *p,
                           float *blur_out, float *R_out, void *stream);
/* The same for B independent frame pairs in one set of launches (the loop of tobac_flow/flow.py:411-423):
 * pair b reads prev + b*img_stride / next + b*img_stride (bytes = pixels, uint8) and writes
 * flow_fwd + b*flow_stride / flow_bwd + b*flow_stride (strides in floats), so the results can land
 * directly in forward_flow[i0 + b] and backward_flow[i0 + 1 + b].  Batching keeps the coarse pyramid
 * levels -- a few hundred workgroups per pair -- busy on all 256 CUs. */
size_t tf_farneback_workspace_bytes_batch(int64_t B, int64_t H, int64_t W, const tf_farneback_params *p);
/* How many pairs to hand to tf_farneback_batch at a time.  OpenCV's box filter carries the float rounding of its running
 * column sums down a whole column (optflowgf.cpp FarnebackUpdateFlow_Blur: `vsum[x] += srow1[x] - srow0[x]`), so the
 * iteration kernel cannot split the rows of a column over workgroups: its parallelism is strips x directions x pairs,
 * and a launch costs a whole number of "rounds" of resident workgroups, at every pyramid level.  Returns the batch size
 * <= max_pairs whose workspace fits max_bytes (0 = no limit) with the best ratio of work to rounds x rows summed over
 * the levels -- the largest one within 1 % of the best (42 pairs at 5424 x 5424 on 256 CUs: full rounds at levels 0, 1, 2). */
int64_t tf_farneback_batch_hint(int64_t H, int64_t W, const tf_farneback_params *p, int64_t max_pairs, size_t max_bytes);
/* Launches are asynchronous, so what a launch found out arrives later: tf_farneback_check() -- call it once the stream the
 * batches ran on has been synchronised (or an event after them has completed) -- returns TF_ESTARVED if a row-sum chain of
 * any iteration launch of this device (made with status_slot 0) since the last report gave up waiting for its left
 * neighbour's hand-over words (bounded polls: a device stalled for seconds by a profiler's serialisation or a preempted
 * queue), TF_OK otherwise.  The flow of those launches then holds NaN rows.  Every tf_farneback_batch* call also reports (and
 * clears) on entry what earlier launches left in ITS status word, so a caller that never checks still cannot go on
 * unnoticed.  The reference has no counterpart (cv2's calc is synchronous); error convention of SURVEY section 8(b).
 *
 * Status slots (round 6): the shared word is read and cleared by whoever looks first, which is wrong as soon as two flows are
 * in flight on one device (two host threads; a deferred check beside the next stack's first batch): one flow's entry check
 * would consume the other's report.  tf_farneback_status_acquire() hands out a word of the caller's own (1 .. 255; 0 = none
 * left or no device: share the device's word); put it into tf_farneback_params.status_slot for every call of ONE flow, read
 * it with tf_farneback_status_check(slot) once that flow's launches have finished (TF_ESTARVED / TF_OK; clears), give it back
 * with tf_farneback_status_release(slot).  Calls with a slot of their own neither read nor clear the shared word. */
int tf_farneback_check(void);
int tf_farneback_status_acquire(void);
int tf_farneback_status_check(int slot);
void tf_farneback_status_release(int slot);
/* test hooks: set a status word from the host, as a starved chain would (tests of the host-side path) */
int tf_farneback_debug_set_starved(void);
int tf_farneback_debug_set_starved_slot(int slot);
/* Workgroups of the iteration kernel's full-resolution launch for B pairs, and (resident_out, may be NULL) how many the
 * device holds at once.  A launch costs whole rounds of resident workgroups: cut a batch into parts
 * (tf_farneback_batch_phase) only while a part still fills a round. */
int64_t tf_farneback_iteration_workgroups(int64_t H, int64_t W, const tf_farneback_params *p, int64_t B, int64_t *resident_out);
int tf_farneback_batch(const uint8_t *prev, const uint8_t *next, int64_t B, int64_t img_stride,
                       int64_t H, int64_t W, const tf_farneback_params *p,
                       float *flow_fwd, float *flow_bwd, int64_t flow_stride,
                       void *ws, size_t ws_bytes, void *stream);
/* tf_farneback_batch_split (round 4): the same batch with the pyramid levels >= 2 run for all B pairs at once and the two
 * finest levels in `parts` parts of ceil(B / parts) pairs -- full-size scratch for one part only
 * (tf_farneback_workspace_bytes_split), so a scratch budget that holds 21 full-size pairs still fills the coarse levels'
 * launches with 42.  Same kernels on the same data: results identical to tf_farneback_batch (parts = 1 is that call).
 * Needs pyr_scale = 0.5 (otherwise, and with fewer than three levels, it runs unsplit). */
size_t tf_farneback_workspace_bytes_split(int64_t B, int64_t parts, int64_t H, int64_t W, const tf_farneback_params *p);
int tf_farneback_batch_split(const uint8_t *prev, const uint8_t *next, int64_t B, int64_t parts, int64_t img_stride,
                             int64_t H, int64_t W, const tf_farneback_params *p,
                             float *flow_fwd, float *flow_bwd, int64_t flow_stride,
                             void *ws, size_t ws_bytes, void *stream);
/* The two halves of a split batch as calls of their own (tf_farneback_can_split says whether the geometry allows it):
 * phase 1 = pyramid levels >= 2 for B pairs, their flow left in the output frames; phase 2 = levels 1 and 0 for B pairs
 * whose output frames hold that flow (any part of the pairs of phase 1).  create_flow runs phase 1 for a whole batch, then
 * per part: phase 2, refinement, smoothing, on_frames_ready -- the coarse launches hold twice the pairs, the caller still
 * gets its frames part by part. */
int tf_farneback_can_split(int64_t H, int64_t W, const tf_farneback_params *p);
size_t tf_farneback_workspace_bytes_phase(int64_t B, int64_t H, int64_t W, const tf_farneback_params *p, int phase);
int tf_farneback_batch_phase(const uint8_t *prev, const uint8_t *next, int64_t B, int64_t img_stride,
                             int64_t H, int64_t W, const tf_farneback_params *p,
                             float *flow_fwd, float *flow_bwd, int64_t flow_stride,
                             void *ws, size_t ws_bytes, void *stream, int phase);

/* ---- a5 tail / section 8f-1: variational refinement of one flow field -------------------------------------
 * replaces cv2.VariationalRefinement.create().calc(I0, I1, flow) as tobac_flow/flow.py:359 creates it and
 * tobac_flow/flow.py:513-519 calls it once per direction when vr_steps > 0 (OpenCV defaults: fixedPointIterations 5,
 * sorIterations 5, alpha 20, delta 5, gamma 10, omega 1.6).  I0, I1: (H, W) uint8; flow: (H, W, 2) float (dx, dy),
 * refined IN PLACE.  params NULL = defaults.  Parity with OpenCV itself is unpinned. */
typedef struct {
    int fixed_point_iterations;   /* 5   */
    int sor_iterations;           /* 5   */
    float alpha;                  /* 20  smoothness term weight        */
    float delta;                  /* 5   colour constancy weight       */
    float gamma;                  /* 10  gradient constancy weight     */
    float omega;                  /* 1.6 SOR relaxation factor         */
} tf_varref_params;
void tf_varref_default_params(tf_varref_params *p);
size_t tf_varref_workspace_bytes(int64_t H, int64_t W);
int tf_varref(const uint8_t *I0, const uint8_t *I1, int64_t H, int64_t W, const tf_varref_params *params,
              float *flow, void *ws, size_t ws_bytes, void *stream);
/* tf_varref_ex = tf_varref + `flags`.
 * TF_VR_FAST_DIVIDE (opt-in): the divisions and square roots of the system assembly and of the SOR updates use the
 *   hardware reciprocal / reciprocal square root (1 ulp) and reciprocal-multiply (the SOR update divides by A11 / A22
 *   ten times per fixed-point iteration: one reciprocal, ten products).  The result is no longer bit-identical to the
 *   arithmetic of OpenCV's variational_refinement.cpp as restated in oracle/c/varref.c, but stays within the 1e-4 px the
 *   flow is specified to (given the same input flow).  The default (flags = 0) evaluates every expression as written. */
#define TF_VR_FAST_DIVIDE 1
/* TF_VR_FAST_SOR (round 4): the reciprocal-multiply form in the SOR sweeps only (one hardware reciprocal of A11 / A22 per
 * pixel and fixed-point iteration, a multiplication per update instead of a correctly rounded division); the system
 * assembly keeps its correctly rounded divisions and square roots. */
#define TF_VR_FAST_SOR 2
int tf_varref_ex(const uint8_t *I0, const uint8_t *I1, int64_t H, int64_t W, const tf_varref_params *params,
                 float *flow, int flags, void *ws, size_t ws_bytes, void *stream);
/* tf_varref_batch (round 6): the refinement of B images in ONE set of launches -- the loop of tobac_flow/flow.py:411-423 over
 * the frame pairs, one direction: image b reads I0 + b * img_stride and I1 + b * img_stride (pixels) and refines flow +
 * b * flow_stride (floats, even) in place.  The grid gains an image dimension, the code per tile is tf_varref's: the same
 * bits.  What it buys is launches that fill the chip -- at 1500 x 2500 the fused SOR kernel of ONE image is 282 tiles on
 * 256 CUs (a full round plus one that is 10 % full), of 23 images 6486 tiles in one launch -- and 11 launches per direction
 * and batch instead of 11 per image.  ws: tf_varref_workspace_bytes_batch(B, H, W) (72 B per pixel and image); with less
 * (>= one image's) the images are refined one after the other. */
size_t tf_varref_workspace_bytes_batch(int64_t B, int64_t H, int64_t W);
int tf_varref_batch(const uint8_t *I0, const uint8_t *I1, int64_t B, int64_t img_stride, int64_t H, int64_t W,
                    const tf_varref_params *params, float *flow, int64_t flow_stride, int flags,
                    void *ws, size_t ws_bytes, void *stream);

/* ---- a6: forward/backward consistency smoothing ----------------------------------------------
 * replaces tobac_flow/flow.py:530-568 smooth_flow_step (4 x cv2.remap via
 * tobac_flow/utils/flow_utils.py:80-99 + np.nanmean).  Outputs must not alias inputs. */
int tf_smooth_flow_step(const float *fwd, const float *bwd, int64_t H, int64_t W, int interp,
                        float *fwd_out, float *bwd_out, void *stream);
/* The same step with create_flow's clip (tobac_flow/flow.py:60-63, np.minimum(np.maximum(v, -max), max): NaN propagates)
 * applied to what it stores -- for the LAST smoothing pass of a stack: clipping is elementwise, so it may ride on the
 * store of the stage that produces the final vectors instead of costing another pass over both flow arrays
 * (tf_flow_finalize_ends then only mirrors the two end frames).  max_value = +inf: no clip. */
int tf_smooth_flow_step_clip(const float *fwd, const float *bwd, int64_t H, int64_t W, int interp,
                             float *fwd_out, float *bwd_out, float max_value, void *stream);

/* single-image warp: tobac_flow/utils/flow_utils.py:80-99 warp_flow (BORDER_CONSTANT NaN) */
int tf_warp_flow(const float *img, const float *flow, int64_t H, int64_t W, int interp,
                 float *out, void *stream);

/* ---- a2 tail: end-frame mirroring and clipping -------------------------------------------------
 * tobac_flow/flow.py:425-426 (forward[-1] = -backward[-1]; backward[0] = -forward[0]) and
 * tobac_flow/flow.py:60-61 (clip both to [-max_value, max_value]); in place. */
int tf_flow_finalize(float *fwd, float *bwd, int64_t T, int64_t H, int64_t W, float max_value,
                     void *stream);
/* tobac_flow/flow.py:425-426 alone, for a stack whose interior is already clipped (tf_smooth_flow_step_clip): writes
 * forward[T - 1] = -backward[T - 1] and backward[0] = -forward[0] (clipped like the rest), touches nothing else. */
int tf_flow_finalize_ends(float *fwd, float *bwd, int64_t T, int64_t H, int64_t W, float max_value,
                          void *stream);

/* ---- a8-a13: semi-Lagrangian convolve / sobel / diff ------------------------------------------
 * replaces tobac_flow/convolve.py:248-348 convolve (with :8-86 warp_flow, :89-144
 * convolve_same_step, :147-245 convolve_step fused into one gather), tobac_flow/sobel.py:89-143
 * and tobac_flow/flow.py:159-191.
 *   data      (T, H, W), data_type TF_F32 or TF_I32 (TF_I32 requires TF_INTERP_NEAREST)
 *   fwd, bwd  (T, H, W, 2) float
 *   structure 27 bytes, (3,3,3) C order, non-zero = tap present
 *   fill      value for out-of-image taps and, when func != STACK, for output pixels whose
 *             input is NaN (convolve.py:346-347)
 *   out       func == STACK: (n_struct, T, H, W); otherwise (T, H, W); element type out_type
 *             (TF_F64 reproduces Flow.sobel's dtype=None behaviour, flow.py:193-199)
 *   t0, t1    only frames t0 <= t < t1 are written (frames t0-1 and t1 are still read as
 *             neighbours) so that a time series can be sharded without a halo copy. */
int tf_convolve(const void *data, int data_type, int64_t T, int64_t H, int64_t W,
                const float *fwd, const float *bwd, const uint8_t *structure_host,
                int interp, double fill, int func, void *out, int out_type,
                int64_t t0, int64_t t1, void *stream);

/* ---- a17 piece: combined edge field --------------------------------------------------------------
 * the elementwise tail of tobac_flow/detection.py:620-642 get_combined_edge_field:
 * edges[edges > 0] += 1; edges -= field; edges[isnan(field)] = inf, in float64 like the reference;
 * out_type TF_F64 (the function's own result) or TF_F32 (the cast tobac_flow/watershed.py:64-65 applies). */
int tf_edge_field(const double *sobel, const float *field, int64_t n, void *out, int out_type, void *stream);

/* tf_sobel_edge_field: the same result in ONE kernel -- tobac_flow/detection.py:620-642 get_combined_edge_field:
 *   edges = Flow.sobel(field, direction="uphill", method=interp)     (float64 stack, sobel.py:89-143, fill NaN)
 *   edges[edges > 0] += 1;  edges = edges - field;  edges[isnan(field)] = inf;   stored as out_type
 * (TF_F32 rounds the finished value the way watershed.py:64-65 would).  Bit-identical to
 * tf_convolve(func = TF_FUNC_SOBEL_UPHILL, out_type = TF_F64) followed by tf_edge_field; the float64 Sobel volume
 * (8 B written + 8 B read per voxel) is never stored. */
int tf_sobel_edge_field(const float *field, int64_t T, int64_t H, int64_t W, const float *fwd, const float *bwd,
                        int interp, void *out, int out_type, void *stream);

/* ---- a14/a15: semi-Lagrangian marker-controlled watershed -------------------------------------
 * replaces tobac_flow/watershed.py:17-168 (wrapper) and tobac_flow/_watershed.pyx:222-344
 * (watershed_raveled, the reference's only native kernel; compactness = 0, wsl = False).
 *   field    (T, H, W) float32            image values (watershed.py:64-65 coercion is the caller's); a NaN at a
 *            floodable pixel (or at a seed next to one) is TF_EINVAL: the reference's heap order is undefined for NaN
 *   markers  (T, H, W) int32, non-zero = seed (negative allowed)
 *   mask     (T, H, W) int8 or NULL (= all ones)
 *   fwd, bwd (T, H, W, 2) float; rounded half-to-even to integer pixel offsets (watershed.py:121-141)
 *   nbr_host n_nbr x 3 int8 (dt, dy, dx) neighbour list IN THE REFERENCE'S ORDER
 *            (skimage _offsets_to_raveled_neighbors, watershed.py:114-116)
 *   chain_depth  number of tie-break levels of the pop-order key (>= 1; 3 suffices for tie-free
 *            fields and for the plateau structure of detect_anvils; see DESIGN.md and the contract below)
 *   labels   (T, H, W) int32 out
 * No padding is needed: out-of-volume neighbours are rejected by coordinate tests, which is what
 * the reference's zero-padded mask achieves (watershed.py:111-113).
 * Workspace: the flood keys live in compact arrays over the RELEVANT pixels (floodable pixels +
 * markers that touch one).  tf_watershed_workspace_bytes(..., max_relevant) sizes the workspace for
 * at most that many (0 = worst case T*H*W).  If the volume has more, tf_watershed returns
 * TF_ENOMEM and stats_host[6] holds the exact count to size a retry with.
 * Size limits: voxels are indexed in 64 bits (T < 65536, H, W < 32768, T*H*W <= 2^36) -- a whole long stack can be
 * flooded exactly in one call if it fits the memory; the compact ids are int32: at most 2^30 relevant pixels
 * (TF_EINVAL beyond).  tf_watershed_raveled keeps the reference's int32 strides: at most 2^31 - 1 voxels.
 * stats_host (optional, 8 x int64): [0] sweeps phase A, [1] sweeps root phase (fast path),
 * [2..4] sweeps of the chain phases when the fast path found a label conflict, [5] conflict flag,
 * [6] relevant pixel count.  This call synchronises the stream.
 *
 * EXACTNESS CONTRACT (never a silent wrong label).  After the root phase the library checks, on the device, every
 * pixel whose label was decided by the last-resort rule "smallest root index among candidates that tie on all
 * compared chain levels":
 *   - ties caused only by the cut-off at `chain_depth` levels are removed by computing further levels, up to
 *     `max_depth` (tf_watershed_ex2; tf_watershed / tf_watershed_ex have max_depth = chain_depth).  If some remain:
 *     return TF_EDEPTH (labels are still written);
 *   - ties that remain with COMPLETE chains are ties between EQUAL-VALUED MARKERS.  The reference pushes all markers
 *     with age 0 (_watershed.pyx:278-284), so their pop order is a by-product of its binary heap's array mechanics
 *     (:67-152), which no order-free formulation reproduces.  The library resolves them by push order (raster index
 *     of the marker = the reference's own marker_locations order) and returns TF_WS_AMBIGUOUS (> 0, not an error)
 *     when at least one pixel's LABEL depends on that; tf_watershed_ex2 reports which pixels.
 *   Return TF_OK therefore means: bit-identical to the reference's output for this input, whatever the tie-breaks. */
#define TF_WS_REPLAY_PENDING 2 /* tf_watershed_finish only: exported for a host replay, call tf_watershed_finish again (see there) */
#define TF_WS_AMBIGUOUS 1      /* success; stats[9] pixels carry a label that depends on the order of equal-valued markers */
#define TF_WS_MAX_DEPTH 12     /* largest chain depth (levels of the pop-order key) */
#define TF_WS_NSTATS 16
#define TF_WS_AMB_DEPENDS 1    /* `ambiguous` bits: label depends on a last-resort tie-break ...                        */
#define TF_WS_AMB_MARKER_TIE 2 /* ... the tie arises HERE, between chains that end in equal-valued markers            */
#define TF_WS_AMB_DEPTH 4      /* ... the tie arises HERE because the chains are cut off at the final depth (TF_EDEPTH) */
size_t tf_watershed_workspace_bytes(int64_t T, int64_t H, int64_t W, int n_nbr, int chain_depth,
                                    int64_t max_relevant);
int tf_watershed(const float *field, const int32_t *markers, const int8_t *mask,
                 const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                 const int8_t *nbr_host, int n_nbr, int chain_depth, int32_t *labels,
                 void *ws, size_t ws_bytes, int64_t *stats_host, void *stream);
/* tf_watershed_ex = tf_watershed + `flags`.
 * TF_WS_SKIP_FAST_PATH: after phase A go straight to the chain phases instead of first trying the
 *   K2-only root phase + exactness check.  The labels are IDENTICAL either way (the fast path is only
 *   taken when no tie-break can matter); the flag is a scheduling hint for inputs known to contain
 *   label conflicts (exact plateaus, as in detect_anvils), where the speculative root phase is wasted
 *   work.  With the flag stats_host[1] = 0 and stats_host[5] = -1 (speculative phase not run). */
#define TF_WS_SKIP_FAST_PATH 1
/* TF_WS_REFERENCE_ORDER (tf_watershed_ex2, tf_watershed_begin, tf_watershed_raveled_ex): when labels hang on the order
 *   of EQUAL-VALUED MARKERS (the case otherwise reported as TF_WS_AMBIGUOUS), reproduce the order the reference's binary
 *   heap gives them.  That order is a by-product of the heap's array mechanics (_watershed.pyx:67-152, 278-284) and depends
 *   on every push and pop before it, so it cannot be derived from the tied markers alone: a first-party host routine of
 *   this library replays the heap's push / pop / sift sequence over the compact flood graph the device has built --
 *   keys only, no labels -- up to the largest marker value at which such a tie occurs, and returns each marker's pop
 *   rank; the device then repeats its root phase with the pop rank in place of the raster index as the last component
 *   of the chain comparison, and writes the labels.  The call returns TF_OK: the labels are the reference's bit for
 *   bit.  Only the heap items at or below that value have to be followed: every other item compares larger than all of
 *   them and only matters through the heap position it occupies, so the replay keeps those "small" items alone (an
 *   occupancy bitmap over the positions + a position -> item table), and the device sends only the seeds among them
 *   (heap position, key, id: 16 B each) and the SUB-GRAPH the replay can reach: the pixels it can pop plus their
 *   out-neighbours (key + neighbour row each; ~1 % of the relevant pixels on a detect_anvils window), through pinned
 *   host memory the library keeps between calls.  Cost: sequential, O(small items x log n) bit tests on one host core;
 *   nothing extra when no such tie exists.  (With more small seeds than relevant pixels -- no staging room -- or
 *   TF_WS_REFERENCE_DENSE=1 in the environment: the dense form, every seed sent and pushed.)
 *   No host pass at all in the commonest case (round 4): when every item at or below that value is a SEED OF THAT ONE
 *   VALUE (no smaller seed, no floodable pixel below it, none of those seeds among the last S heap positions), equal keys
 *   never swap and the reference's build and pops have a closed form -- seed k settles at the first node of its
 *   root-to-k chain no earlier such seed holds, and the seeds leave in the pre-order of the tree they form -- which the
 *   device evaluates itself (k_ws_tie_*: one launch per tree level); TF_WS_REFERENCE_HOST=1 keeps the host replay.
 *   stats[13] = items the replay popped, [14] = seeds the replay was given, [15] = microseconds the detour took on the
 *   host clock (export + replay + repeated root phase; 0 if not needed). */
#define TF_WS_REFERENCE_ORDER 2
/* TF_WS_DEFER_SWEEPS (tf_watershed_begin only; round 5): begin returns after the set-up and the export on a guessed tie value
 * -- the only parts that read `fwd` / `bwd`, `field` and `markers` arrays of the window -- and leaves phase A and the chain
 * levels to tf_watershed_sweeps (or to tf_watershed_finish, which runs them if nobody has).  A caller that begins several
 * windows at once starts all their host replays first and sweeps afterwards (parallel.detect_stack_windows).  Same labels. */
#define TF_WS_DEFER_SWEEPS 4
int tf_watershed_ex(const float *field, const int32_t *markers, const int8_t *mask,
                    const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                    const int8_t *nbr_host, int n_nbr, int chain_depth, int flags, int32_t *labels,
                    void *ws, size_t ws_bytes, int64_t *stats_host, void *stream);
/* tf_watershed_ex2: the full form.
 *   chain_depth  depth the chain phases start at (when the speculative phase is skipped or rejected)
 *   max_depth    deepest level the call may go to on its own (chain_depth <= max_depth <= TF_WS_MAX_DEPTH); the
 *                workspace must be sized for it: tf_watershed_workspace_bytes(..., max_depth, ...)
 *   ambiguous    (T, H, W) uint8 out or NULL: TF_WS_AMB_* bits per voxel (0 everywhere when the call returns TF_OK)
 *   stats_host   TF_WS_NSTATS x int64 or NULL: [0..7] as above, [8] chain depth the labels were computed at,
 *                [9] pixels whose label depends on a last-resort tie-break, [10] origins of such ties between
 *                equal-valued markers, [11] origins left by the depth cut-off (> 0 <=> TF_EDEPTH), [12] root phases run */
int tf_watershed_ex2(const float *field, const int32_t *markers, const int8_t *mask,
                     const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                     const int8_t *nbr_host, int n_nbr, int chain_depth, int max_depth, int flags,
                     int32_t *labels, uint8_t *ambiguous, void *ws, size_t ws_bytes,
                     int64_t *stats_host, void *stream);

/* One flood in three parts (round 4): the replay of TF_WS_REFERENCE_ORDER is host work, so the replays of several time
 * windows can run on worker threads while the device floods the next windows.
 *   tf_watershed_begin   set-up, phase A and the chain phases (same arguments as tf_watershed_ex2, no outputs yet).  TF_OK:
 *                        *job_out is a job that MUST be passed to tf_watershed_finish or tf_watershed_abandon; any other
 *                        code: no job.  `ws`, `markers`, `field` and the job belong together until then: the workspace must
 *                        not be used by another call (one workspace per flood in flight).
 *                        guessed_tie_key: -1, or the ordered key (tf_watershed_job_info [7] of an earlier, similar flood) of
 *                        a marker value the caller expects the largest tie between equal-valued markers at.  With
 *                        TF_WS_REFERENCE_ORDER the export of what the replay reads then happens HERE, before the relaxation
 *                        phases (it needs none of their results), the replay can run beside them, and finish runs ONE root
 *                        phase, with the pop ranks.  Any guess is safe: finish compares it with the tie value it finds and
 *                        falls back on export - replay - second root phase when the guess was too low (a guess that is too
 *                        high only makes the replay longer).
 *   tf_watershed_needs_replay   1 if tf_watershed_replay has work to do (a guess was given)
 *   tf_watershed_sweeps  phase A and the chain levels of a job begun with TF_WS_DEFER_SWEEPS (a no-op otherwise); same stream,
 *                        same thread rules as begin / finish; synchronises the stream; stats_host (NULL or TF_WS_NSTATS x int64)
 *                        receives the job's statistics so far ([0] phase A's sweeps, [5] the scheduling probe).  Optional: finish
 *                        runs them if nobody has.  On an error the job stays the caller's to abandon.
 *   tf_watershed_replay  the host replay; no HIP call, any thread, different jobs concurrently.  Optional: finish runs it
 *                        if the caller did not.  MUST have returned before the job is finished or abandoned.
 *   tf_watershed_finish  root phase (with the pop ranks if there are any) + exactness check, labels (and report) written;
 *                        frees the job; return codes and stats of tf_watershed_ex2; info_host: NULL or 12 x int64 as
 *                        tf_watershed_job_info.  Synchronises the stream.
 *                        With TF_WS_REFERENCE_ORDER and labels that hang on the order of equal-valued markers without
 *                        (sufficient) ranks it exports for the tie value it found and returns TF_WS_REPLAY_PENDING (2):
 *                        the job is NOT freed, nothing is written; the caller runs tf_watershed_replay (again: any thread)
 *                        and calls tf_watershed_finish a second time, which uploads the ranks, repeats the root phase and
 *                        completes (it runs the replay itself if the caller did not).
 *   tf_watershed_abandon frees a job without finishing it. */
int tf_watershed_begin(const float *field, const int32_t *markers, const int8_t *mask,
                       const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                       const int8_t *nbr_host, int n_nbr, int chain_depth, int max_depth, int flags,
                       int64_t guessed_tie_key, void *ws, size_t ws_bytes, int64_t *stats_host, void *stream, void **job_out);
int tf_watershed_needs_replay(const void *job);
int tf_watershed_sweeps(void *job, int64_t *stats_host);
int tf_watershed_replay(void *job);
int tf_watershed_finish(void *job, int32_t *labels, uint8_t *ambiguous, int64_t *stats_host, int64_t *info_host);
void tf_watershed_abandon(void *job);
/* the stream tf_watershed_finish works on (default: the one the job was begun on; everything begin enqueued has completed
 * when begin returns, so any stream is safe): a caller that keeps its main stream busy with other work -- the flow of the
 * next frames -- finishes its floods on a second one instead of queueing the root phase behind that work */
int tf_watershed_set_stream(void *job, void *stream);
/* info (12 x int64): [0] replay form (0 none, 1 sparse, 2 dense, 3 none needed: pop ranks computed on the device in closed
 * form -- every heap item at or below the tie value was a seed of that one value), [1] seeds, [2] seeds at or below the tie value,
 * [3] pixels of the exported sub-graph, [4] relevant pixels, [5] microseconds of the export, [6] microseconds of the
 * replay (-1: not run), [7] ordered key of the largest tie value finish found (-1: no label hangs on such a tie, or not
 * finished), [8] 1 if the export ran on a guess, [9] 1 if the guess covered the tie value, [10] the tie key the export
 * used (-1: none), [11] 1 if the pop ranks came from the device's closed form */
int tf_watershed_job_info(const void *job, int64_t *info);

/* tf_watershed_raveled: the reference's only native seam, argument for argument --
 *   tobac_flow/_watershed.pyx:222-233  watershed_raveled(image, marker_locations, structure, forward_offset,
 *   backward_offset, forward_offset_locations, backward_offset_locations, mask, strides, compactness, output, wsl)
 * as tobac_flow/watershed.py:151-164 calls it (flat, padded, C-contiguous arrays; `output` mutated in place).
 *   image, forward_offset, backward_offset, mask, output, marker_locations: DEVICE arrays (n elements; marker_locations
 *   n_markers int64, strictly ascending = np.flatnonzero order); structure / *_offset_locations / strides: HOST arrays
 *   of n_structure (n_strides) entries.  compactness must be 0 and wsl 0 (the reference's call path); neighbour indices
 *   are range-checked in addition to the mask test.  Exactness contract, return codes and stats as tf_watershed_ex2
 *   (chain depth starts at min(3, max_depth)). */
size_t tf_watershed_raveled_workspace_bytes(int64_t n, int n_structure, int max_depth, int64_t max_relevant);
int tf_watershed_raveled(const float *image, int64_t n, const int64_t *marker_locations, int64_t n_markers,
                         const int64_t *structure_host, int n_structure,
                         const int32_t *forward_offset, const int32_t *backward_offset,
                         const int32_t *forward_offset_locations_host, const int32_t *backward_offset_locations_host,
                         const int8_t *mask, const int32_t *strides_host, int n_strides, double compactness,
                         int32_t *output, int wsl, int max_depth, void *ws, size_t ws_bytes,
                         int64_t *stats_host, void *stream);
/* the same with `flags` (TF_WS_REFERENCE_ORDER) */
int tf_watershed_raveled_ex(const float *image, int64_t n, const int64_t *marker_locations, int64_t n_markers,
                         const int64_t *structure_host, int n_structure,
                         const int32_t *forward_offset, const int32_t *backward_offset,
                         const int32_t *forward_offset_locations_host, const int32_t *backward_offset_locations_host,
                         const int8_t *mask, const int32_t *strides_host, int n_strides, double compactness,
                         int32_t *output, int wsl, int max_depth, int flags, void *ws, size_t ws_bytes,
                         int64_t *stats_host, void *stream);

/* ---- section 8f-2: scipy.ndimage glue of the detection recipes (bit-exact with SciPy) ---------------------
 * tf_binary_morph: scipy.ndimage.binary_erosion (op 0) / binary_dilation (op 1) of a (T, H, W) uint8 volume with a
 *   3x3x3 structuring element (27 host bytes, C order), `iterations` >= 1, `border_value` 0/1
 *   (tobac_flow/detection.py:80-93, 509, 551-553, 571-573, 608-616; binary_opening = erosion then dilation).
 *   tmp: scratch of the same size, required when iterations > 1.
 * tf_linearise: tobac_flow/utils/normalisation_utils.py:36-56 linearise_field in float32.
 * tf_label_extent: per label 1..n_labels the first / last leading index (tobac_flow/analysis.py:15-35
 *   find_object_lengths = tmax - tmin + 1) and whether it touches `mask` (analysis.py:38-63 mask_labels);
 *   arrays of n_labels + 1 entries, entry 0 unused; absent labels: tmin = 0x7f7f7f7f, tmax = -1.
 * tf_apply_lut: out = lut[labels] (tobac_flow/utils/label_utils.py:265-309 remap_labels' gather). */
/* tf_correlate1d_sym: one pass of scipy.ndimage.correlate1d along `axis` (0 = t, 1 = y, 2 = x) with a SYMMETRIC kernel
 *   of 2*radius+1 host doubles, mode 'reflect' -- the pass scipy.ndimage.gaussian_filter is made of
 *   (tobac_flow/detection.py:65, 137-138, 150: ndi.gaussian_filter(field, (0, sigma, sigma)) = the y pass then the x pass,
 *   each rounding to the array's dtype).  type TF_F32 / TF_F64 (in and out alike); in != out; radius <= 64.
 *   Bit-exact with SciPy: double accumulation, centre first, then the pairs from the OUTERMOST inwards.
 * tf_grey_morph: scipy.ndimage.grey_erosion (op 0) / grey_dilation (op 1) with a flat, point-symmetric footprint inside a
 *   3x3x3 box (27 host bytes, C order), mode 'reflect' (tobac_flow/detection.py:106-108 ndi.grey_opening = erosion then
 *   dilation).  SciPy's visiting order and strict comparisons are kept, so NaNs land where SciPy puts them. */
int tf_correlate1d_sym(const void *in, int type, int64_t T, int64_t H, int64_t W, int axis,
                       const double *weights_host, int radius, void *out, void *stream);
int tf_grey_morph(const void *in, int type, int64_t T, int64_t H, int64_t W, const uint8_t *footprint_host,
                  int op, void *out, void *stream);
int tf_binary_morph(const uint8_t *in, int64_t T, int64_t H, int64_t W, const uint8_t *structure_host,
                    int op, int iterations, int border_value, uint8_t *out, uint8_t *tmp, void *stream);
int tf_linearise(const float *field, int64_t n, double lower, double upper, float *out, void *stream);
int tf_label_extent(const int32_t *labels, const uint8_t *mask, int64_t T, int64_t H, int64_t W, int n_labels,
                    int *tmin, int *tmax, uint8_t *hit, void *stream);
int tf_apply_lut(const int32_t *labels, int64_t n, const int32_t *lut, int n_lut, int32_t *out, void *stream);
/* the same gather with ids <= 0 passed through (window stitch, tobac_flow/linking.py:153-161: background -1 and 0 keep
 * their value while the positive ids are renumbered) */
int tf_apply_lut_keep_nonpositive(const int32_t *labels, int64_t n, const int32_t *lut, int n_lut, int32_t *out, void *stream);
/* The elementwise glue of detect_anvils' seeds (tobac_flow/detection.py:547-561, 590-617) in two passes:
 * tf_field_masks: ge1 = field >= 1, le0_or_nan = (field <= 0) | isnan(field), isnan_out = isnan(field) as 0 / 1 bytes;
 * tf_merge_seeds: seeds = (bg | isnan) ? -1 : comp. */
int tf_field_masks(const float *field, int64_t n, uint8_t *ge1, uint8_t *le0_or_nan, uint8_t *isnan_out, void *stream);
int tf_merge_seeds(const int32_t *comp, const uint8_t *bg, const uint8_t *isnan_in, int64_t n, int32_t *seeds, void *stream);
/* tf_label: scipy.ndimage.label(input, structure) for a (T, H, W) uint8 volume and a centro-symmetric 3x3x3
 *   structuring element -- used as tobac_flow/utils/label_utils.py:143-180 flat_label (time planes of the structure
 *   zeroed) and for 3-D labelling.  Component numbers follow SciPy (raster order of each component's first pixel).
 *   n_labels_host receives the component count; the call synchronises the stream. */
size_t tf_label_workspace_bytes(int64_t T, int64_t H, int64_t W);
int tf_label(const uint8_t *in, int64_t T, int64_t H, int64_t W, const uint8_t *structure_host,
             int32_t *labels, int *n_labels_host, void *ws, size_t ws_bytes, void *stream);

/* ---- a16: flow-aware labelling (tobac_flow/label.py:84-321) and section 8e/8f-3: cross-window linking ----------
 * tf_pair_counts: for two int32 volumes of n voxels, every distinct pair (a[i], b[i]) with a[i] > 0 and b[i] > 0
 *   (b[i] >= 0 when include_b_zero) and how often it occurs, sorted by (a, b) -- the counting the reference does per
 *   label with np.bincount / np.unique (tobac_flow/utils/label_utils.py:352-376, tobac_flow/linking.py:33-47).
 *   out_a / out_b / out_count: device arrays of max_pairs entries; *n_pairs_host = number of pairs.  The workspace is
 *   sized for at most max_runs runs of equal consecutive pairs (0 = worst case n); with more the call returns
 *   TF_ENOMEM and *n_pairs_host = the run count to size a retry with.  Synchronises the stream.
 * tf_label_sizes: np.bincount(labels) for ids 0 .. n_labels (device int64[n_labels + 1]).
 * tf_flow_link_overlap: tobac_flow/label.py:249-321 -- links per-step labels `flat_labels` (T, H, W) into objects:
 *   nearest-neighbour warps of the labels to t-1 / t+1 (structure * [1, 0, 1]; exactly one tap per outer plane, as the
 *   reference's tuple unpacking requires), overlap counts, the criterion `count > absolute_overlap and count >=
 *   overlap * min(n_locs, size(other))`, and the reference's first-come-first-served grouping in ascending label order
 *   (forward neighbours before backward ones).  out may alias nothing.  *n_objects_host = number of objects.
 * tf_flow_label: tobac_flow/label.py:84-175 with subsegment_shrink = 0: flat_label (utils/label_utils.py:143-180 =
 *   scipy.ndimage.label with the structure's t-planes zeroed) followed by the linking above.  mask: (T, H, W) uint8.
 * tf_window_overlap_pairs: tobac_flow/linking.py:49-93 / :96-140 -- `left` / `right` are the labels two consecutive
 *   time windows give to the frames they share (first and last common frame already dropped, linking.py:55-56);
 *   returns on the HOST the (left id, right id) pairs with count >= atol (count > 0 if atol == 0) and, if rtol > 0,
 *   max(count / pixels(left id), count / pixels(right id)) >= rtol; the reference uses atol = 5, rtol = 0.5.
 *   pairs_host: 2 * max_pairs int32.  Connected components of these pairs are the stitched objects (linking.py:153-161). */
/* tf_pair_rank: tobac_flow/utils/label_utils.py:183-200 (`make_step_labels`, the first step of detection.relabel_anvils
 *   :660-687) -- out[i] = 1 + the index of (a[i], b[i]) in the list of distinct pairs (pairs_a, pairs_b: device arrays as
 *   tf_pair_counts returns them, sorted by (a, b)), 0 where a[i] <= 0, b[i] <= 0 or the pair is not listed.  With a = the
 *   pieces of the non-zero mask connected within a time step (tf_label, t planes of the structure zeroed) and b = the
 *   labels this is the reference's numbering: contiguous from 1, by piece, then by original label.  Volumes that are all
 *   16-byte aligned take wide loads; any other (4-byte aligned) views -- labels[1:] with H * W % 4 != 0 -- scalar ones. */
int tf_pair_rank(const int32_t *a, const int32_t *b, int64_t n, const int32_t *pairs_a, const int32_t *pairs_b,
                 int64_t n_pairs, int32_t *out, void *stream);
size_t tf_pair_counts_workspace_bytes(int64_t n, int64_t max_runs);
int tf_pair_counts(const int32_t *a, const int32_t *b, int64_t n, int include_b_zero,
                   int32_t *out_a, int32_t *out_b, int64_t *out_count, int64_t max_pairs,
                   int64_t *n_pairs_host, void *ws, size_t ws_bytes, void *stream);
int tf_label_sizes(const int32_t *labels, int64_t n, int64_t n_labels, int64_t *sizes, void *stream);
size_t tf_flow_link_workspace_bytes(int64_t T, int64_t H, int64_t W, int64_t max_runs);
int tf_flow_link_overlap(const int32_t *flat_labels, const float *fwd, const float *bwd,
                         int64_t T, int64_t H, int64_t W, const uint8_t *structure_host,
                         double overlap, int64_t absolute_overlap, int32_t *out, int *n_objects_host,
                         void *ws, size_t ws_bytes, void *stream);
size_t tf_flow_label_workspace_bytes(int64_t T, int64_t H, int64_t W, int64_t max_runs);
int tf_flow_label(const uint8_t *mask, const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                  const uint8_t *structure_host, double overlap, int64_t absolute_overlap,
                  int32_t *labels, int *n_objects_host, void *ws, size_t ws_bytes, void *stream);
int tf_window_overlap_pairs(const int32_t *left, const int32_t *right, int64_t n, int64_t atol, double rtol,
                            int32_t *pairs_host, int64_t max_pairs, int64_t *n_pairs_host,
                            void *ws, size_t ws_bytes, void *stream);

/* tf_slice_labels: tobac_flow/utils/label_utils.py:312-349 (`slice_labels`, the `*_step_label` variables of the output
 *   files, tobac_flow/dataset.py:189-229): one id per (label, time step) of an int32 (T, hw) label volume -- the pieces of
 *   a label inside one step stay together -- numbered densely from 1 in ascending (step, label) order; 0 (and anything
 *   negative) stays 0.  (As in the reference, a volume in which NO voxel is 0 loses its smallest shifted id to 0: the
 *   dense LUT is np.arange over the ids that occur.)  *n_step_labels_host = the largest id written.  The workspace holds two int32 per shifted id (sum over
 *   steps of the step's largest label, `id_capacity`); if it is too small the call returns TF_ENOMEM with
 *   *n_step_labels_host = the capacity to size a retry with.  Synchronises the stream. */
/* tf_label_stats: tobac_flow/analysis.py:293-376 (`weighted_statistics_on_labels`) and :204-245 (`get_stats_for_labels`,
 *   weights == NULL: every weight 1) -- per label id 1 .. n_labels of an int32 volume of n voxels, over the voxels whose
 *   `field` value is not NaN: out[(id - 1) * 6 + {0..5}] = { sum of all (non-NaN) weights of the label, sum of the weights
 *   at valued voxels, weighted mean, weighted standard deviation about it (sqrt(sum w (x - mean)^2 / sum w)), max and min
 *   of the values with weight > 0 } as doubles on the device (NaN where undefined).  Sums are accumulated in double. */
size_t tf_label_stats_workspace_bytes(int64_t n_labels);
int tf_label_stats(const int32_t *labels, const float *field, const float *weights, int64_t n, int64_t n_labels,
                   double *out, void *ws, size_t ws_bytes, void *stream);
size_t tf_slice_labels_workspace_bytes(int64_t T, int64_t id_capacity);
int tf_slice_labels(const int32_t *labels, int64_t T, int64_t hw, int32_t *out, int64_t *n_step_labels_host,
                    void *ws, size_t ws_bytes, void *stream);

/* ---- measurement aid (bench.py's roofline figure) ---------------------------------------------
 * tf_profile_enable(1): every kernel launch of the library is bracketed by HIP events on its own
 * stream and tagged with its ALGORITHMIC byte count.  tf_profile_collect() synchronises
 * the recorded events and returns, per kernel id < tf_profile_kernel_count(), the number of
 * launches, their summed duration in ms and their summed algorithmic bytes, then clears them. */
int tf_profile_enable(int on);
int tf_profile_kernel_count(void);
const char *tf_profile_kernel_name(int id);
int tf_profile_collect(int64_t *calls, double *ms, double *bytes);

/* ---- host staging: the containers of the reference's API <-> HBM ------------------------------------------------------
 * The reference's entry points take and return numpy / xarray containers (tobac_flow/decorators.py:21-61,
 * scripts/dcc_detect_goes.py:164-303); a drop-in caller therefore pays PCIe for every field and every label volume, and
 * pageable memory moves at a third of the link's rate.  These entry points are what the Python layer moves host containers
 * with ("pinned-host staging to HBM"); pointers named *_host are HOST pointers.
 *
 * tf_host_alloc / tf_host_free: PINNED host blocks from a pool kept by size class (a freed block is handed out again; the
 *   pool keeps at most TF_PINNED_CACHE_GB = 16 GB of free blocks, tf_host_pool_trim(keep_bytes) returns more of them to the
 *   system).  A new block is mmap'ed with MADV_HUGEPAGE, touched by
 *   the host threads and registered (hipHostRegister): 22 ms for 1.88 GB where hipHostMalloc takes 250 ms
 *   (tools/microbench/pin_cost.hip); hipHostMalloc is the fallback.  A result is
 *   downloaded straight into such a block, which the caller wraps as the array it returns: no second host copy.
 *   tf_host_is_pinned: 1 if [ptr, ptr + bytes) lies inside a block that is handed out.  tf_host_pool_spare: how many free
 *   blocks of a size class the pool holds.  (Allocating blocks AHEAD on a background thread, while the device computes, was
 *   measured in round 6 and dropped: hipHostMalloc holds up the launching thread's runtime calls -- the first pass of the
 *   script sequence got slower, 3.02 s against 2.70 s.)
 * tf_upload(dst, src_host, bytes, hash_out_host, stream): host -> device on `stream`.  A pageable source is copied by host
 *   threads (<= 16, TF_STAGING_THREADS) chunk by chunk (8 MiB, TF_STAGING_CHUNK_MB) into a ring of pinned slots, each chunk's
 *   DMA enqueued as soon as it is staged; returns when the SOURCE has been read completely (the caller may free or change it),
 *   the last DMAs and everything enqueued after the call stay asynchronous.  A source inside a pool block goes out as one DMA
 *   (the block must stay untouched until `stream` has passed it).  hash_out_host != NULL: the 128-bit content checksum of the
 *   source (two uint64), computed by the copying threads on the way.
 * tf_download(dst_host, src, bytes, stream): device -> host, complete on return (DMA at the link's rate into a pool block).
 * tf_hash_host / tf_hash_dev: the same 128-bit checksum of a host buffer (host threads) / of device memory (one kernel; the
 *   two words are returned on the host, the call synchronises `stream`): sums over 16-byte blocks of
 *   fold(mul128(w0 ^ a_i, w1 ^ b_i)), position-keyed, the byte length folded in -- any split over threads or lanes gives the
 *   same words, so the checksum of a device result equals the checksum of its downloaded copy.  The Python layer keys its
 *   cache of device twins with it (tobac_flow_amd/_staging.py): a host array presented again is recognised by CONTENT -- the
 *   same object, or another temporary with the same values (`wvd - swd` of scripts/dcc_detect_goes.py:227,241) -- never
 *   trusted by its address.  Not cryptographic. */
int tf_host_alloc(size_t bytes, void **ptr_out_host);
int tf_host_free(void *ptr_host);
int tf_host_is_pinned(const void *ptr_host, size_t bytes);
int tf_host_pool_stats(int64_t *live_bytes, int64_t *cached_bytes);
int tf_host_pool_spare(size_t bytes);      /* free blocks of the size class of `bytes` in the pool */
int tf_host_pool_trim(size_t keep_bytes);
int tf_upload(void *dst, const void *src_host, size_t bytes, uint64_t *hash_out_host, void *stream);
int tf_download(void *dst_host, const void *src, size_t bytes, void *stream);
int tf_hash_host(const void *src_host, size_t bytes, uint64_t *hash_out_host);
int tf_hash_dev(const void *src, size_t bytes, uint64_t *hash_out_host, void *stream);

/* tf_copy16: dst = src, 16 bytes per lane per access -- the measured practical HBM ceiling of bench.py's roofline.
 * (Round 5's tf_stream_create_cu_mask / tf_stream_create_priority / tf_stream_destroy / tf_debug_cu_histogram served two
 * scheduling experiments that were not adopted; they live in tools/experiments/stream_experiments.hip now, outside the
 * library: DESIGN.md section 7.) */
int tf_copy16(const void *src, void *dst, size_t bytes, void *stream);
int tf_copy16_variant(const void *src, void *dst, size_t bytes, void *stream, int variant);   /* development forms of the same copy */

/* SURVEY.md 8(b) "ownership": the library never retains a caller's pointer past return and owns no device memory (all
 * scratch is the caller's workspace); its only pooled resource is the HIP events of the timing facility above.
 * tf_shutdown() switches timing off and destroys them.  Idempotent; the library remains usable afterwards. */
int tf_shutdown(void);

/* development aid (tests): compares, on `count` pseudo-random operand pairs drawn from the range it is used in, the
 * shared-reciprocal division of the refinement's system kernel with the hardware's correctly rounded division;
 * *mismatches_host must come back 0.  Allocates and frees 8 bytes of device memory. */
int tf_selftest_shared_divide(int64_t count, uint64_t seed, uint64_t *mismatches_host, void *stream);

#ifdef __cplusplus
}
#endif
#endif
