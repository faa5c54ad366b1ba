/*
 * sucre_hip.h -- C ABI of libsucre_hip.so, the MI355X (gfx950) engine for the SUCRe hot path.
 *
 * The reference (clementinboittiaux/sucre) is pure Python and has no FFI layer; its boundary for this path is
 * the Python surface   sfm.Image.match_images (sfm.py:127-138),  loader.MatchesFile.prepare_matches /
 * load_matches (loader.py:78-118),  sucre.SUCRe.__init__/update_J/forward (sucre.py:36-82)  and
 * sucre.adam (sucre.py:124-157).  A drop-in therefore binds these entry points with ctypes (INTEGRATION.md shows
 * the stub); sucre_amd/{sfm,loader,sucre}.py is that binding, mirroring the reference's names and arguments.
 *
 * Conventions
 *   - Plain C: pointers and sizes only.  Every `*_dev` / device pointer is HIP device memory owned by the
 *     caller (e.g. a PyTorch-ROCm allocation); the library allocates no persistent device memory, performs no
 *     host<->device copy of bulk data and never synchronises: every call only enqueues kernels on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream) and returns.
 *   - Return value: 0 = OK, negative = error; sucre_last_error() gives the message for the calling thread.
 *     No C++ exception crosses the ABI.  All arguments are validated before anything is launched.
 *   - Re-entrant; no global mutable state except the thread-local error string.  One host thread / process
 *     per GPU; concurrent calls on different streams or devices are safe.
 *   - All state of one restoration lives in ONE caller-allocated workspace of sucre_workspace_bytes() bytes,
 *     256-byte aligned.  Its internal layout (observation store, Adam state, ...) is private; the few regions a
 *     host needs to read back are located with sucre_ws_offset().
 *
 * Data layout in HBM: the target image is cut into 16x16-pixel tiles.  Matching writes,
 * for every (tile, view) pair, one 1792-byte chunk = 256 float32 ranges z=||cP|| (0 = no observation) + 256 x 3
 * uint8 colours (7 B/observation instead of the reference's 28-byte (u,v,cP,I) records, loader.py:33-53,103-118,
 * and its HDF5 spill); sucre_finalize_matches then compacts them per pixel, sorted by observation count, into the
 * store the fit streams: the sorted pixels are cut into strips of 64 (one pixel per lane of the wave that fits the
 * strip), a strip's observations are stored as chunks of 64 pixels x 4 levels, and finalize also writes every fit
 * wave's list of items to stream (the plan).  J and the Adam moments are float32 planes per strip in that sorted order.
 */
#ifndef SUCRE_HIP_H
#define SUCRE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUCRE_ABI_VERSION 1

/* error codes */
#define SUCRE_OK 0
#define SUCRE_ERR_ARG (-1)      /* NULL / misaligned pointer, bad size            */
#define SUCRE_ERR_RANGE (-2)    /* view index, iteration count ... out of range    */
#define SUCRE_ERR_LAUNCH (-3)   /* HIP reported a launch error                     */

/* fit flags */
#define SUCRE_FIT_CLOSED_FORM 1u /* --use-closed-form: J is a closed-form buffer, sucre.py:66-77,141 */
#define SUCRE_FIT_OBS_U16MM 2u   /* the store was finalised with SUCRE_OBS_U16MM (must match, see below) */
#define SUCRE_FIT_KEEP_J 8u      /* with SUCRE_FIT_CLOSED_FORM: do not append the final update_J (sucre.py:156), i.e. leave
                                    J = J(theta_k) of the last iteration k next to theta_{k+1} -- the pair the reference
                                    plots at a --save-interval stop (sucre.py:141,153-154) */

/*
 * Observation formats of the store the fit streams.  SUCRE_OBS_F32: float32 range + 3 uint8
 * colours = 7 B/observation, lossless for the reference's data (default).  How the library keeps those float32 ranges is
 * its own business as long as every bit comes back: when all ranges of an image lie within 2^24 - 2 float32 bit patterns
 * of the smallest one (a span of about a factor of four) and nothing rides along in extension planes, the finalize step
 * stores them as 24-bit offsets from it (6 B/observation, decided on the device, no host synchronisation); an image whose
 * ranges span more keeps the float32 words.  SUCRE_OBS_F32_Z26 asks for 26-bit offsets instead (6.25 B/observation) whenever
 * the ranges lie within 2^26 - 2 bit patterns (a factor of up to 256 between the nearest and the farthest range) -- an
 * opt-in: measured, the fit reads the words faster than it decodes these.  Results are bit-identical
 * whichever form the store takes.  SUCRE_OBS_F32_PLAIN asks for the float32 words themselves (A/B measurements, tests).
 * SUCRE_OBS_U16MM: the range as uint16 millimetres, rint(1000 z) clamped to [1, 65535] = 5 B/observation (BASELINE
 * config 5) -- lossy by at most 0.5 mm of range; every sum is still accumulated in float32/float64 exactly as with
 * SUCRE_OBS_F32.  A store finalised with SUCRE_OBS_F32 or SUCRE_OBS_F32_PLAIN is fitted WITHOUT SUCRE_FIT_OBS_U16MM.
 */
#define SUCRE_OBS_F32 0
#define SUCRE_OBS_U16MM 1
#define SUCRE_OBS_F32_PLAIN 2
#define SUCRE_OBS_F32_Z26 3   /* float32 ranges as 26-bit offsets when they fit, else as words (never the 24-bit form) */

/*
 * What the three float32 extension planes of the second workspace (`lws`, see the *_light entry points) carry per
 * observation.  SUCRE_EXT_POINTS: the camera-frame point cP (artificial-light model).  SUCRE_EXT_COLOUR: the colour I
 * as float32 -- for inputs that are not k/255 (the reference's --image-scale resizes colours in float64,
 * loader.py:156-163), with the plain water model; the uint8 colours of the main store are then unused.
 */
#define SUCRE_EXT_POINTS 1
#define SUCRE_EXT_COLOUR 2
#define SUCRE_EXT_POINTS_COLOUR 3 /* both, in two sets of planes: the light model on inputs whose colours are not k/255
                                     (--light-model with --image-scale); needs sucre_light_workspace_bytes_ext(.., 3) */
#define SUCRE_FIT_EXT_COLOUR 4u  /* flag of sucre_fit_run_light / sucre_update_J_ext: lws was matched with SUCRE_EXT_COLOUR */
#define SUCRE_FIT_EXT_BOTH 16u   /* ... with SUCRE_EXT_POINTS_COLOUR: the light model on float32 colours (excludes the former) */

/*
 * One view of the scene = the arguments the reference reads from an sfm.Image (sfm.py:81-88): depth map,
 * colour image, camera, pose.  The 3x3 matrices are row-major float32 and must be computed by the caller the
 * way the reference computes them (Kinv = K.inverse(), sfm.py:92; Rinv/tinv = Pose.inverse() = (R.T, -R.T@t),
 * sfm.py:42-47) so that match sets are bit-identical to the reference's.
 */
typedef struct sucre_view {
    const float *depth;   /* device, (H,W) float32 metres, <=0 = invalid   loader.py:166-170 */
    const uint8_t *rgb;   /* device, (H,W,3) uint8                         loader.py:156-163
                             NULL in a NEIGHBOUR view of sucre_match_views / sucre_match_views_light / sucre_match_map:
                             `depth` then points to the view's sucre_pack_view records (depth and colour in one gather) */
    int32_t H, W;
    float K[9];
    float Kinv[9];
    float R[9];           /* world-from-camera                              sfm.py:32-40      */
    float t[3];
    float Rinv[9];
    float tinv[3];
} sucre_view_t;

/* workspace regions a host may read back (all written by the library) */
enum {
    SUCRE_WS_VIEW_COUNT = 0, /* uint64[n_views]  matches per view (before the min_cover rule), sfm.py:136 */
    SUCRE_WS_VIEW_KEEP = 1,  /* uint32[n_views]  1 = view passed min_cover                                */
    SUCRE_WS_N_OBS = 2,      /* uint64[1]        len(matches_data), sucre.py:135,202                       */
    SUCRE_WS_PARAMS = 3,     /* float32[9]       B[3], beta[3], gamma[3], sucre.py:41-43                  */
    SUCRE_WS_SUMS = 4,       /* float64[12]      reduced gradient sums of the last sucre_fit_grad (multi-GPU) */
    SUCRE_WS_N_OBS_TOTAL = 5,/* uint64[1]        n_obs used for the 1/(3 n_obs) scale (shared-water: summed over ranks) */
    SUCRE_WS_STORE_FORMAT = 6/* uint32[4]        how sucre_finalize_matches* laid the observations out: 0 = float32 ranges,
                                                 1 = uint16 millimetres, 2 = 24-bit, 3 = 26-bit offsets of the float32 bit
                                                 patterns; the offset; the smallest and the largest float32 bit pattern among
                                                 the ranges */
};

int sucre_version(void);
const char *sucre_last_error(void);

/* Bytes of the workspace for an HxW target fitted against n_views views (0 on invalid arguments). */
size_t sucre_workspace_bytes(int H, int W, int n_views);
/* Byte offset of a SUCRE_WS_* region inside the workspace (-1 on invalid arguments). */
int64_t sucre_ws_offset(int H, int W, int n_views, int region);

/*
 * Replaces Image.match_images (sfm.py:127-138) + MatchesFile.prepare_matches/load_matches (loader.py:78-118)
 * for views k0 <= k < k1 of `views_dev` (a device array of n_views sucre_view_t): two-way depth-map matching
 * (sfm.py:90-125, 154-175), d = depth2[v2,u2], cP = K2^-1 d [u2+.5, v2+.5, 1], z = ||cP|| (sucre.py:53),
 * I = rgb2[v2,u2]; results go to the observation store of `ws`.  `target` is a HOST struct whose depth/rgb
 * point to device memory.
 */
int sucre_match_views(void *ws, int H, int W, int n_views, const sucre_view_t *target,
                      const sucre_view_t *views_dev, int k0, int k1, void *stream);

/*
 * Image.match_two_way (sfm.py:121-125) of the target against view k in dense form: map_dev[(v1,u1)] = v2*W2 + u2
 * of the matched pixel, or -1 (H*W int32).  The (u1,v1,u2,v2) lists of sfm.Matches (sfm.py:145-152) are its
 * non-negative entries in row-major order.  Needs no workspace.  Used by the sfm.Matches compatibility API,
 * the HDF5 shim of loader.MatchesFile.save_matches (loader.py:68-76) and the parity tests.
 */
int sucre_match_map(int H, int W, int n_views, const sucre_view_t *target, const sucre_view_t *views_dev, int k,
                    int32_t *map_dev, void *stream);

/*
 * Optional, per neighbour view and once per scene: depth map and uint8 colour image interleaved into H*W 8-byte records
 * {float32 depth, r, g, b, 0} at packed_dev (8-byte aligned, H*W*8 bytes).  A view of `views_dev` whose `rgb` is NULL and
 * whose `depth` points to such records is matched with ONE gather per landing pixel instead of two (the match kernel is
 * bound by its gathers); the values are the same, so are the results.  The target keeps the plain form (its depth map and
 * colour image are read as such: sfm.py:129, sucre.py:47).  Not for float32 colour images (SUCRE_EXT_COLOUR modes).
 */
int sucre_pack_view(const float *depth_dev, const uint8_t *rgb_dev, int H, int W, void *packed_dev, void *stream);
/*
 * The same for n views of one size in as few launches as possible (sixteen views per launch): depth_dev[k], rgb_dev[k],
 * packed_dev[k] are HOST arrays of device pointers.  A target's image_list (sfm.py:127-138) is packed with five launches
 * instead of 65.
 */
int sucre_pack_views(const float *const *depth_dev, const uint8_t *const *rgb_dev, void *const *packed_dev, int n, int H, int W,
                     void *stream);

/*
 * Image.project_to_view + the truncation and bound test of Image.match_one_way (sfm.py:103-107, 115-117) for an
 * explicit list of world points: wP_dev is (3, n) float32 row-major (x[n], y[n], z[n], as sfm.py lays points out),
 * pix_dev[i] = v2 * W + u2 of the pixel of `view` that point i truncates into, or -1 when it falls outside the
 * sensor (NaN / inf included; like the reference, nothing tests that the point is in front of the camera).  Same
 * float32 operation order as the fused match kernel, i.e. bit-identical to the reference's torch CPU arithmetic.
 * `view` is a HOST struct (only its camera and pose are read).  This is the building block behind
 * sfm.Image.match_one_way / match_two_way when a caller passes its own point sets.
 */
int sucre_project_points(const sucre_view_t *view, const float *wP_dev, int64_t n, int32_t *pix_dev, void *stream);

/*
 * Alternative to sucre_match_views for one view: loads an explicit match list -- one group of a matches file
 * written by the reference (loader.py:68-76: u1, v1 int16; the ranges z = ||K2^-1 d [u2+.5, v2+.5, 1]|| of
 * loader.py:113 + sucre.py:53 and the colours I*255 as uint8, n x 3) -- into view k of the observation store,
 * replacing whatever the view held.  This is how a kept matches file (--keep-matches, sucre.py:185) is consumed
 * without re-matching.  Follow with sucre_finalize_matches as usual.
 */
int sucre_import_view(void *ws, int H, int W, int n_views, int k, const int16_t *u1_dev, const int16_t *v1_dev,
                      const float *z_dev, const uint8_t *rgb_dev, int64_t n, void *stream);

/*
 * The `len(matches) / (W*H) > min_cover` rule (sfm.py:136) for every view, n_obs, and the count-sorted per-pixel
 * compaction of the kept observations that the fit iterates over.  Call once after all sucre_match_views calls.
 */
int sucre_finalize_matches(void *ws, int H, int W, int n_views, double min_cover, void *stream);
/*
 * Same with an explicit observation format.  A store finalised with SUCRE_OBS_U16MM must be fitted with
 * SUCRE_FIT_OBS_U16MM in the flags of sucre_fit_run / sucre_fit_grad (and sucre_update_J_fmt); a mismatch is
 * detected on the device and poisons the logged cost with NaN instead of reading the store wrongly.
 */
int sucre_finalize_matches_fmt(void *ws, int H, int W, int n_views, double min_cover, int obs_format, void *stream);

/*
 * SUCRe.__init__ (sucre.py:36-50): B, beta, gamma <- params0 (host, 9 floats; the reference uses 0.1),
 * J <- rgb1/255 with NaN where depth1 <= 0, Adam moments <- 0.  If J0_dev != NULL it is an (H,W,3) float32
 * warm start replacing rgb1/255 (--params-path, sucre.py:206-207).
 */
int sucre_fit_init(void *ws, int H, int W, int n_views, const uint8_t *rgb1_dev, const float *depth1_dev,
                   const float *params0, const float *J0_dev, void *stream);

/*
 * sucre.adam (sucre.py:124-157): iterations t0+1 .. t0+T of torch.optim.Adam(lr, betas, eps) on {B,beta,gamma,J}
 * (or {B,beta,gamma} with SUCRE_FIT_CLOSED_FORM), all enqueued without a host sync.  trace_dev (nullable) gets
 * T x 10 float64: the cost sum r^2 the reference logs (sucre.py:146,150) and the nine parameters after the step.
 * With SUCRE_FIT_CLOSED_FORM the final update_J of sucre.py:156 is included (unless SUCRE_FIT_KEEP_J), and a run that
 * starts at t0 = 0 first solves J once from the initial parameters, so that the one-pass kernel measures its
 * residuals from a J that is already close (accuracy only: iteration 0 re-solves J from the observations anyway).
 */
int sucre_fit_run(void *ws, int H, int W, int n_views, int t0, int T, double lr, double beta1, double beta2,
                  double eps, unsigned flags, double *trace_dev, void *stream);

/*
 * The two halves of one iteration, for data-parallel runs where several GPUs share the water parameters:
 * sucre_fit_grad leaves the reduced sums at SUCRE_WS_SUMS (float64[12]) so the host can all-reduce them
 * (RCCL) before sucre_fit_step applies Adam to B, beta, gamma.  sucre_fit_run == T x (grad, step).
 */
int sucre_fit_grad(void *ws, int H, int W, int n_views, int step, double lr, double beta1, double beta2,
                   double eps, unsigned flags, void *stream);
int sucre_fit_step(void *ws, int H, int W, int n_views, int step, double lr, double beta1, double beta2,
                   double eps, double *trace_row_dev, void *stream);
/* Overrides the observation count used in the 1/(3 n_obs) gradient scale (sum over ranks). */
int sucre_set_n_obs_total(void *ws, int H, int W, int n_views, uint64_t n_obs_total, void *stream);

/*
 * Shared water parameters over SEVERAL images (all the images of one rank; BASELINE config 4): one launch and one
 * collective per iteration.  `group_dev`: sucre_group_bytes(n_images) bytes of device memory, 256-byte aligned, set
 * up once by sucre_group_init from the images' workspaces (each matched, finalised and fit_init-ed; `images` is a host
 * array) and the nine start parameters.  sucre_group_iter(step), step = 1, 2, ...: first takes the Adam step on B,
 * beta, gamma that the sums left by the previous call -- all-reduced over the ranks by the host in between -- call
 * for (and logs it in row step-2 of trace_dev, T x 10 float64, nullable), then runs the gradient pass of iteration
 * `step` over every image (J steps stay local, sucre.py:142-148) and leaves this rank's ten float64 sums at
 * sucre_group_sums_offset() inside the buffer.  sucre_group_finish(step) takes the last pending step and copies the
 * final parameters into every image's workspace (SUCRE_WS_PARAMS).  n_obs_total: observations over all images of all
 * ranks, the n_obs of the 1/(3 n_obs) scale.
 */
typedef struct sucre_group_image {
    void *ws;              /* the image's workspace */
    int32_t H, W, n_views; /* its geometry */
    int32_t reserved;
} sucre_group_image_t;
size_t sucre_group_bytes(int n_images);
int64_t sucre_group_sums_offset(void);
int sucre_group_init(void *group_dev, int n_images, const sucre_group_image_t *images, const float *params0, void *stream);
int sucre_group_iter(void *group_dev, int n_images, int step, double lr, double beta1, double beta2, double eps, unsigned flags,
                     uint64_t n_obs_total, double *trace_dev, void *stream);
int sucre_group_finish(void *group_dev, int n_images, int step, double lr, double beta1, double beta2, double eps,
                       uint64_t n_obs_total, double *trace_dev, void *stream);

/*
 * Independent images, ONE launch per iteration: the loop over the images of a scene (sucre.py:243-261, one SUCRe module and one
 * sucre.adam call per image, nothing shared) with the iterations of n_images images advancing together.  Every image keeps
 * its own B, beta, gamma, J, Adam state and log; its results are bit for bit those of sucre_fit_run on it alone -- the
 * launch only spares small images the per-launch cost they are dominated by (BASELINE config 1) and fills the end of one
 * image's pass with the next image's beginning.  All images have one size (H, W: one launch grid); their view counts may
 * differ (n_views[i] = what image i's workspace was laid out for).  They must have been matched / imported, finalized and
 * initialised (sucre_fit_init) and stand at the same step t0.
 * ws / trace_dev / n_views: HOST arrays of n_images entries (trace_dev itself or any of its entries may be NULL; an entry is
 * that image's (T, 10) float64 device log as in sucre_fit_run); batch_dev: sucre_batch_bytes(n_images) bytes of device memory,
 * 256-byte aligned, that the call fills (the image table the launches read) and that must stay untouched until they have run.
 * flags as in sucre_fit_run.
 */
size_t sucre_batch_bytes(int n_images);
int sucre_fit_run_batch(void *batch_dev, int n_images, void *const *ws, double *const *trace_dev, int H, int W, const int *n_views,
                        int t0, int T, double lr, double beta1, double beta2, double eps, unsigned flags, void *stream);

/* SUCRe.update_J(force_update=True) (sucre.py:66-77): closed-form J from the current parameters. */
int sucre_update_J(void *ws, int H, int W, int n_views, void *stream);
int sucre_update_J_fmt(void *ws, int H, int W, int n_views, int obs_format, void *stream);

/* J as the reference lays it out: (H,W,3) float32 (sucre.py:213-215). */
int sucre_export_J(const void *ws, int H, int W, int n_views, float *J_dev, void *stream);

/*
 * Output stage, SUCRe.plot_J (sucre.py:84-95): exact order statistics of the restored image without moving it.  For
 * every channel of J_dev ((H,W,3) float32) the values at the given 0-based ranks among the VALID pixels (no NaN in any
 * channel, sucre.py:87), ascending: out_dev[c * n_ranks + r].  `ranks` is a host array of n_ranks <= 8 entries, each
 * below the number of valid pixels; scratch_dev: sucre_select_scratch_bytes() bytes, 8-byte aligned.  numpy's percentile
 * is a linear interpolation between two such values; the caller does that part (in numpy's arithmetic).
 */
size_t sucre_select_scratch_bytes(void);
int sucre_select_ranks(const float *J_dev, int H, int W, int n_ranks, const uint64_t *ranks, float *out_dev, void *scratch_dev,
                       void *stream);

/* Number of valid pixels of J_dev (no NaN in any channel: `valid`, sucre.py:86-87) -> *count_dev (uint64, device). */
int sucre_count_valid(const float *J_dev, int H, int W, uint64_t *count_dev, void *stream);

/*
 * The remainder of SUCRe.plot_J (sucre.py:88-94) in one pass over J_dev: clip every channel to [lo[c], hi[c]] (the
 * two percentiles; host arrays of 3), subtract the minimum, divide by the maximum, multiply by 255 and truncate to
 * uint8 -> out_dev ((H,W,3) uint8); pixels with a NaN come out black.  After the clip the minimum is lo[c] and the
 * maximum of the shifted values is the float32 difference hi[c] - lo[c], so no reduction is involved and the result
 * equals numpy's bit for bit.
 */
int sucre_plot_stretch(const float *J_dev, int H, int W, const float *lo, const float *hi, uint8_t *out_dev, void *stream);

/*
 * MatchesFile.check_integrity (loader.py:89-101) over the whole store in one launch: verdict_dev[k] (uint32, one per
 * view) gets bit 0 if a stored range of view k is not finite, bit 1 if one is negative, bit 2 if the number of
 * stored ranges > 0 differs from the view's match count; 0 = sound.  scratch_dev: n_views uint64 of scratch.
 */
int sucre_check_store(const void *ws, int H, int W, int n_views, uint32_t *verdict_dev, uint64_t *scratch_dev,
                      void *stream);

/*
 * One view of the observation store as dense planes (inverse of the tiling): z (H,W) float32, 0 = no match,
 * and rgb (H,W,3) uint8.  Either output may be NULL.  Used by tests and by the MatchesData compatibility shim.
 */
int sucre_export_view(const void *ws, int H, int W, int n_views, int k, float *z_dev, uint8_t *rgb_dev,
                      void *stream);

/*
 * ---- artificial-light model: --light-model, SUCRe.compute_l_z with se3.exp (sucre.py:54-61, se3.py:22-27) ------------
 * The model needs the camera-frame point cP of every observation (loader.py:113), which the 7-byte store does not
 * keep, so this mode uses a second caller-owned buffer `lws` of sucre_light_workspace_bytes() bytes (256-byte
 * aligned) next to `ws`: three float32 planes per chunk + the 19 parameters B[3], beta[3], gamma[3],
 * cam2light[6], sigma[4] (row-major 2x2; sucre.py:41-46) and their Adam state.  The *_light entry points replace
 * their namesakes; J, sucre_export_J, sucre_export_view, sucre_match_map and SUCRE_WS_* work unchanged on `ws`.
 * params0: 19 host floats (the reference starts from 0.1 x 9, 0 x 6, identity).  trace_dev: T x 20 float64
 * (cost, then the 19 parameters after the step).  The parameters live at sucre_light_params_offset() in `lws`.
 */
size_t sucre_light_workspace_bytes(int H, int W, int n_views);
int64_t sucre_light_params_offset(int H, int W, int n_views);
int sucre_match_views_light(void *ws, void *lws, int H, int W, int n_views, const sucre_view_t *target,
                            const sucre_view_t *views_dev, int k0, int k1, void *stream);
/* Same with float32 colour images: every view's `rgb` points to (H,W,3) float32 instead of uint8 (SUCRE_EXT_COLOUR). */
int sucre_match_views_fcolour(void *ws, void *lws, int H, int W, int n_views, const sucre_view_t *target,
                              const sucre_view_t *views_dev, int k0, int k1, void *stream);
/*
 * Both at once -- the reference accepts --light-model together with --image-scale (sfm.py:193-199, sucre.py:54-61): the
 * views' `rgb` are (H,W,3) float32 images, the camera points go to the first set of planes and the float32 colours to
 * a second one (SUCRE_EXT_POINTS_COLOUR).  `lws` must then be sucre_light_workspace_bytes_ext(H, W, n_views,
 * SUCRE_EXT_POINTS_COLOUR) bytes (the second set sits behind everything else, so all other offsets are unchanged);
 * finalise with sucre_finalize_matches_ext(.., SUCRE_EXT_POINTS_COLOUR, ..) and pass SUCRE_FIT_EXT_BOTH to
 * sucre_fit_run_light / sucre_update_J_ext.  sucre_finalize_matches_ext with SUCRE_EXT_POINTS or SUCRE_EXT_COLOUR equals
 * sucre_finalize_matches_light.
 */
size_t sucre_light_workspace_bytes_ext(int H, int W, int n_views, int ext_mode);
int sucre_match_views_light_fcolour(void *ws, void *lws, int H, int W, int n_views, const sucre_view_t *target,
                                    const sucre_view_t *views_dev, int k0, int k1, void *stream);
int sucre_finalize_matches_ext(void *ws, void *lws, int H, int W, int n_views, double min_cover, int ext_mode, void *stream);
/*
 * sucre_import_view with extension planes: ext_dev holds three float32 planes [3][n] -- the camera points cP of the
 * list (loader.py:113; ext_mode SUCRE_EXT_POINTS, rgb_dev required) or its colours I (loader.py:87; SUCRE_EXT_COLOUR,
 * for colours that are not k/255; rgb_dev may be NULL), or both as six planes [6][n], cP first (SUCRE_EXT_POINTS_COLOUR;
 * rgb_dev may be NULL).  This is how a caller-built MatchesData (loader.py:36-53)
 * enters the engine; sucre_export_view_ext returns the planes of view k as (3, H, W), zero where nothing was observed.
 * With the light model the range of an observation IS the norm of its camera point (sucre.py:53, compute_l_z): the
 * J-parameter light kernel forms it from the point instead of reading the stored range, and takes a non-zero cP.z (the
 * depth) as the mark of a real observation.
 */
int sucre_import_view_ext(void *ws, void *lws, int H, int W, int n_views, int k, const int16_t *u1_dev, const int16_t *v1_dev,
                          const float *z_dev, const uint8_t *rgb_dev, const float *ext_dev, int64_t n, int ext_mode, void *stream);
int sucre_export_view_ext(const void *ws, const void *lws, int H, int W, int n_views, int k, float *planes_dev, void *stream);
int sucre_finalize_matches_light(void *ws, void *lws, int H, int W, int n_views, double min_cover, void *stream);
int sucre_fit_init_light(void *ws, void *lws, int H, int W, int n_views, const uint8_t *rgb1_dev, const float *depth1_dev,
                         const float *params0, const float *J0_dev, void *stream);
/* closed-form J with the illumination factor in absorption and backscatter (sucre.py:66-77 with light_model) */
int sucre_update_J_light(void *ws, void *lws, int H, int W, int n_views, void *stream);
int sucre_update_J_ext(void *ws, void *lws, int H, int W, int n_views, unsigned flags, void *stream);
int sucre_fit_run_light(void *ws, void *lws, int H, int W, int n_views, int t0, int T, double lr, double beta1,
                        double beta2, double eps, unsigned flags /* SUCRE_FIT_CLOSED_FORM | SUCRE_FIT_KEEP_J | SUCRE_FIT_EXT_COLOUR or SUCRE_FIT_EXT_BOTH */, double *trace_dev,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SUCRE_HIP_H */
