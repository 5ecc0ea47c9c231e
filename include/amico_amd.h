/*
 * amico_amd.h -- C ABI of the MI355X-native AMICO per-voxel fitter (libamico_amd.so).
 *
 * Drop-in boundary for the hot path `model.fit(evaluation)` of daducci/AMICO v2.1.0.
 * The reference has no C ABI for this path: its native boundary is Cython `cdef` calls
 * (amico/lut.pxd:4 `dir_to_lut_idx`; amico/models.pyx:18 `cyspams.interfaces.nnls/lasso`)
 * made from the `_fit` hot loops (models.pyx:816-991 NODDI, 1168-1286 FreeWater,
 * 1509-1627 SANDI).  Each entry point below cites the reference interface it replaces.
 *
 * Conventions: plain pointers + sizes, no C++/torch types; every call returns an int
 * status (0 = ok, negative = error, see AMX_E_*); the library never keeps host pointers
 * after a call returns; device memory is owned by amx_ctx / amx_lut handles.
 * Arrays use the reference's layouts and dtypes (C-order, float64 signals / maps).
 * Threading: the calls on one context, and on the dictionary handles made from it, are serialised by the caller, like the
 * reference's single `model.fit` call per Evaluation; the FreeWater / SANDI handles cache small per-dictionary tables for the
 * last (lambda1, lambda2) they were fitted with and rebuild them when the solver parameters change.
 */
#ifndef AMICO_AMD_H
#define AMICO_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AMX_OK              0
#define AMX_E_BADARG       -1   /* NULL / negative size / unsupported protocol size          */
#define AMX_E_HIP          -2   /* HIP runtime error (message in amx_last_error)             */
#define AMX_E_DIR_OOB      -3   /* lut.pyx:352-354 "index out of bounds" (voxel in last_error) */
#define AMX_E_OVERFLOW     -4   /* a voxel needed more active atoms than the kernels support  */
#define AMX_E_NODEVICE     -5   /* no HIP device / wrong architecture                         */

/* flags of the *_fit calls (Evaluation.get_config(...) switches read by BaseModel.fit,
 * models.pyx:214-217, 797, 1149) */
#define AMX_F_RMSE          1u  /* doComputeRMSE        -> out_rmse   f64[n_vox]      */
#define AMX_F_NRMSE         2u  /* doComputeNRMSE       -> out_nrmse  f64[n_vox]      */
#define AMX_F_MODULATED     4u  /* doSaveModulatedMaps  -> out_mod    f64[n_vox][2]   */
#define AMX_F_CORRECTED     8u  /* doSaveCorrectedDWI   -> out_ycorr  f64[n_vox][nS]  */
#define AMX_F_DEBUG_X      16u  /* solver coefficients  -> the buffer registered with amx_set_debug_x */

typedef struct amx_ctx amx_ctx;   /* one per process+GPU: stream-ordered workspace, error state */
typedef struct amx_lut amx_lut;   /* device-resident dictionary (KERNELS) of one model          */

int  amx_version(void);
/* "amico_amd <version> csrc <16 hex digits>": the digits are sha256 of the library's sources (the .hip and .hpp files of amico_amd/csrc and this
 * header, concatenated in name order) as they were when the library was BUILT -- amico_amd._capi.source_id() takes the same hash of
 * the sources in the tree, so a stale .so, or a profile summary taken from other kernels (profiles/pmc_traffic.json carries the id
 * of the build it measured), is detected instead of trusted.  Static storage. */
const char *amx_build_id(void);
/* HIP devices this process may create contexts on: the largest gfx950 device number + 1 (0: none).  The reference's caller is ONE process
 * (core.py:465-466: results = self.model.fit(self)) whose fit spreads its voxels over `nthreads` host threads in contiguous chunks
 * (models.pyx:204-211); the MI355X counterpart of that is one context per device of the node, driven by one host thread each, on contiguous
 * shards of the caller's arrays -- amico_amd.models does exactly that when AMX_DEVICES=all (or a list) is set: every call of this ABI sets its
 * context's device for the calling thread, contexts share nothing, and calls on DIFFERENT contexts may run concurrently from different threads. */
int  amx_device_count(void);
/* A context that fits ONE SHARD of a larger call (several contexts sharing a fit, see above): the host-buffer calls that follow choose their
 * paths -- seeded chain or not, kernel builds, rescue pass -- by `total`, the size of the whole call, not by the shard's, so that a voxel is
 * settled by the same arithmetic whichever device it lands on and however many there are (bit-identical maps).  0 = off (the default). */
int  amx_set_call_voxels(amx_ctx *ctx, int64_t total);
/* device < 0: current HIP device.  Fails with AMX_E_NODEVICE when no gfx950 GPU is visible. */
int  amx_ctx_create(int device, amx_ctx **out);
void amx_ctx_destroy(amx_ctx *ctx);
/* last error text of this ctx ("" if none); valid until the next call on ctx */
const char *amx_last_error(amx_ctx *ctx);

/* ---- dictionaries: replace the per-thread re-materialisation of KERNELS in _fit
 *      (models.pyx:840-847 NODDI, 1190-1191 FreeWater, 1527-1528 SANDI) by ONE upload.   */

/* NODDI.  KERNELS['wm'] f32[n_wm][ndirs][nS], ['iso'] f32[nS], ['norms'] f64[dwi][n_wm]
 * (rows identical, models.pyx:781-784), ['icvf'],['kappa'] f32[n_wm] (models.pyx:763-789);
 * htable int16[181*181] (lut.pyx:71-91); dwi_idx = scheme.dwi_idx int64[dwi_count].
 * Any shape the reference's loop takes (models.pyx:825-861) up to nS <= 512 volumes and n_wm + 1 (+ 1) <= 256 atoms.  The seeded
 * chain (seed solvers -> Gram-space certificates) covers every protocol of that range with <= 160 atoms; a dictionary tile that does
 * not fit a compute unit's LDS (288 volumes x 145 atoms float32 = 167 KB: an HCP-style acquisition) is read from HBM / L2 by the
 * wavefront-per-voxel kernels instead -- slower for the few per cent of voxels that reach them, same results.                      */
int amx_lut_upload_noddi(amx_ctx *ctx, const float *wm, const float *iso, const double *norms,
                         const float *icvf, const float *kappa, const int16_t *htable,
                         const int64_t *dwi_idx, int n_wm, int ndirs, int nS, int dwi_count,
                         int is_exvivo, amx_lut **out);
/* FreeWater.  KERNELS['D'] f32[n_perp][ndirs][nS], ['CSF'] f32[n_iso][nS] (models.pyx:1122-1123) */
int amx_lut_upload_freewater(amx_ctx *ctx, const float *D, const float *CSF, const int16_t *htable,
                             int n_perp, int n_iso, int ndirs, int nS, amx_lut **out);
/* SANDI.  KERNELS['signal'] f64 column-major [nS][n_atoms], ['norms'] f64[n_atoms]
 * (models.pyx:1456-1482); Rs, d_in, d_isos = model parameters used by the maps (:1540-1542) */
int amx_lut_upload_sandi(amx_ctx *ctx, const double *signal, const double *norms, const double *Rs,
                         const double *d_in, const double *d_isos, int nS, int n_rs, int n_in,
                         int n_iso, amx_lut **out);
/* CylinderZeppelinBall.  KERNELS['wmr'] f32[n_rs][ndirs][nS] (cylinders), ['wmh'] f32[n_perp][ndirs][nS] (zeppelins),
 * ['iso'] f32[n_iso][nS] (balls) (models.pyx:488-520); Rs = model.Rs f64[n_rs] in metres, used by the maps (:627) */
int amx_lut_upload_czb(amx_ctx *ctx, const float *wmr, const float *wmh, const float *iso, const double *Rs,
                       const int16_t *htable, int n_rs, int n_perp, int n_iso, int ndirs, int nS, amx_lut **out);
void amx_lut_destroy(amx_lut *lut);

/* ---- lut.pxd:4  cdef int dir_to_lut_idx(double[::1] direction, short[::1] hash_table)
 * batched; dirs f64[n][3] (host, NOT modified -- the reference flips it in place,
 * lut.pyx:335-338); out_idx int32[n].  AMX_E_DIR_OOB mirrors the RuntimeError.             */
int amx_dir_to_lut_idx(amx_ctx *ctx, const amx_lut *lut, const double *dirs, int64_t n,
                       int32_t *out_idx);

/* ---- model.fit hot loops, HOST buffers in / out (H2D + kernels + D2H, blocking; the copies of large
 * inputs overlap with the solver, in batches -- and the maps of finished batches go home while later batches are still being solved: when a
 * call returns an error (AMX_E_DIR_OOB from a later batch, a HIP error) the output arrays may already hold the results of the batches before it.
 * The reference raises before it returns anything (models.pyx:904 inside the loop that fills `estimates`, which is then lost with the
 * exception); the Python mirror does the same: its wrappers raise and the arrays are unreachable.
 * y f64[n_vox][nS] (evaluation.y, core.py:451-452), dirs f64[n_vox][3] (evaluation.DIRs).
 * lambda1 >= 0, lambda2 >= 0 like cyspams' lasso (lambda2 = 0 runs the QR solver in A-space).  */

/* NODDI._fit models.pyx:816-991: estimates f64[n_vox][3 (+1 ex-vivo)] = NDI, ODI, FWF(, dot) */
int amx_noddi_fit(amx_ctx *ctx, const amx_lut *lut, const double *y, const double *dirs,
                  int64_t n_vox, double lambda1, double lambda2, unsigned flags,
                  double *out_estimates, double *out_rmse, double *out_nrmse, double *out_mod);
/* FreeWater._fit models.pyx:1168-1286: estimates f64[n_vox][2 (Human) | 4 (Mouse)] */
int amx_freewater_fit(amx_ctx *ctx, const amx_lut *lut, const double *y, const double *dirs,
                      int64_t n_vox, double lambda1, double lambda2, int is_mouse, unsigned flags,
                      double *out_estimates, double *out_rmse, double *out_nrmse, double *out_ycorr);
/* SANDI._fit models.pyx:1509-1627: estimates f64[n_vox][6] */
int amx_sandi_fit(amx_ctx *ctx, const amx_lut *lut, const double *y, int64_t n_vox,
                  double lambda1, double lambda2, unsigned flags,
                  double *out_estimates, double *out_rmse, double *out_nrmse);

/* CylinderZeppelinBall._fit models.pyx:526-652: estimates f64[n_vox][3] = v, a, d.  Any lambda2 >= 0 like the reference's
 * lasso (models.pyx:439, 615): the default 4.0 makes the Gram-space solver the right tool, lambda2 < 1e-6 runs the thin-QR
 * solver in A-space.  (The model's `isExvivo`, which the reference never defines, is not a parameter here.) */
int amx_czb_fit(amx_ctx *ctx, const amx_lut *lut, const double *y, const double *dirs, int64_t n_vox,
                double lambda1, double lambda2, unsigned flags,
                double *out_estimates, double *out_rmse, double *out_nrmse);
int amx_czb_fit_f32(amx_ctx *ctx, const amx_lut *lut, const float *y, const double *dirs, int64_t n_vox,
                    double lambda1, double lambda2, unsigned flags,
                    double *out_estimates, double *out_rmse, double *out_nrmse);
int amx_czb_fit_device(amx_ctx *ctx, const amx_lut *lut, const double *d_y, const double *d_dirs,
                       int64_t n_vox, double lambda1, double lambda2, unsigned flags,
                       double *d_estimates, double *d_rmse, double *d_nrmse, void *hip_stream);

/* The same three calls with FLOAT32 signals: the image is float32 in the reference (core.py:136) and only cast to
 * float64 when the masked voxels are gathered (core.py:451), so a float32 `y` carries the same values in half the PCIe
 * bytes; it is widened on the GPU and every result is identical to the float64 call.                           */
int amx_noddi_fit_f32(amx_ctx *ctx, const amx_lut *lut, const float *y, const double *dirs,
                      int64_t n_vox, double lambda1, double lambda2, unsigned flags,
                      double *out_estimates, double *out_rmse, double *out_nrmse, double *out_mod);
int amx_freewater_fit_f32(amx_ctx *ctx, const amx_lut *lut, const float *y, const double *dirs,
                          int64_t n_vox, double lambda1, double lambda2, int is_mouse, unsigned flags,
                          double *out_estimates, double *out_rmse, double *out_nrmse, double *out_ycorr);
int amx_sandi_fit_f32(amx_ctx *ctx, const amx_lut *lut, const float *y, int64_t n_vox,
                      double lambda1, double lambda2, unsigned flags,
                      double *out_estimates, double *out_rmse, double *out_nrmse);

/* Progress of the host-buffer calls: models.pyx:28-43, 981 keep a per-thread voxel counter that ProgressBar polls
 * (util.py); here `callback(done, total, user)` is called from the calling thread as batches of voxels complete (large
 * inputs are fitted in batches: a first one of 131 072 voxels, then equal parts of at most 393 216 voxels) and once with done == total at the end.  NULL unregisters.
 * The device-pointer calls below only ENQUEUE the fit: there the callback is a host function on the stream (hipLaunchHostFunc),
 * i.e. it is called from a HIP runtime thread when the work before it has finished -- after each of NODDI's three stages
 * (done = n/3, 2n/3, n) and at the end of a FreeWater / SANDI / CylinderZeppelinBall fit (done = n).  It must not call back
 * into this library.                                                                                                        */
int amx_set_progress(amx_ctx *ctx, void (*callback)(int64_t done, int64_t total, void *user), void *user);

/* ---- the same with DEVICE buffers (inputs already resident in HBM, e.g. torch tensors'
 * data_ptr()); work is enqueued on `hip_stream` (a hipStream_t, NULL = default stream) and
 * the call returns without synchronising.  amx_sync_status() waits for the stream and
 * returns the status of everything enqueued since the previous amx_sync_status().          */
int amx_noddi_fit_device(amx_ctx *ctx, const amx_lut *lut, const double *d_y, const double *d_dirs,
                         int64_t n_vox, double lambda1, double lambda2, unsigned flags,
                         double *d_estimates, double *d_rmse, double *d_nrmse, double *d_mod,
                         void *hip_stream);
int amx_freewater_fit_device(amx_ctx *ctx, const amx_lut *lut, const double *d_y,
                             const double *d_dirs, int64_t n_vox, double lambda1, double lambda2,
                             int is_mouse, unsigned flags, double *d_estimates, double *d_rmse,
                             double *d_nrmse, double *d_ycorr, void *hip_stream);
int amx_sandi_fit_device(amx_ctx *ctx, const amx_lut *lut, const double *d_y, int64_t n_vox,
                         double lambda1, double lambda2, unsigned flags, double *d_estimates,
                         double *d_rmse, double *d_nrmse, void *hip_stream);
/* The same with float32 signals in HBM -- the dtype the image has before core.py:451-452 widen it (core.py:136: float32), so
 * the maps are bit-identical to the float64 calls on the widened values.  NODDI (A'y GEMM, left-over kernels), every
 * wavefront-per-voxel kernel, CylinderZeppelinBall and FreeWater's matrix-core projection read the float32 rows in place (FreeWater:
 * 260 instead of 520 bytes per voxel); for the remaining lane kernels (FreeWater with error maps / corrected signal, SANDI) a float64
 * copy is made on the device first.                                                                                            */
int amx_noddi_fit_device_f32(amx_ctx *ctx, const amx_lut *lut, const float *d_y, const double *d_dirs,
                             int64_t n_vox, double lambda1, double lambda2, unsigned flags,
                             double *d_estimates, double *d_rmse, double *d_nrmse, double *d_mod, void *hip_stream);
int amx_freewater_fit_device_f32(amx_ctx *ctx, const amx_lut *lut, const float *d_y, const double *d_dirs, int64_t n_vox,
                                 double lambda1, double lambda2, int is_mouse, unsigned flags, double *d_estimates,
                                 double *d_rmse, double *d_nrmse, double *d_ycorr, void *hip_stream);
int amx_sandi_fit_device_f32(amx_ctx *ctx, const amx_lut *lut, const float *d_y, int64_t n_vox, double lambda1, double lambda2,
                             unsigned flags, double *d_estimates, double *d_rmse, double *d_nrmse, void *hip_stream);
int amx_czb_fit_device_f32(amx_ctx *ctx, const amx_lut *lut, const float *d_y, const double *d_dirs, int64_t n_vox,
                           double lambda1, double lambda2, unsigned flags, double *d_estimates, double *d_rmse,
                           double *d_nrmse, void *hip_stream);
int amx_sync_status(amx_ctx *ctx, void *hip_stream);

/* ---- the solvers' own output: the coefficient vectors `x` that cyspams.interfaces.nnls / lasso hand back to
 * _fit (models.pyx:911, 926, 940, 1238, 1569: `x` fully written, exact zeros off the support).  The reference keeps
 * them internal; here they are observable so that tests can certify the DEVICE solution itself (KKT conditions,
 * supports), not only the maps derived from it.  d_x is a DEVICE buffer the caller owns (zero it first; NULL
 * unregisters).  Every later *_fit / *_fit_device call on this ctx whose flags carry AMX_F_DEBUG_X fills it:
 *   NODDI      f64[n_vox][3][n_atoms]: row 0 = stage-1 NNLS over all atoms (wm..., [dot,] iso); row 1 = LASSO
 *              coefficients of the wm atoms (column-normalised dictionary, models.pyx:917-921) followed by the
 *              stage-1 iso (dot) coefficients (the reference reuses one array); row 2 = the debiased x (:940-942)
 *   FreeWater  f64[n_vox][n_atoms]    the lasso solution (models.pyx:1238)
 *   SANDI      f64[n_vox][n_atoms]    the lasso solution rescaled by KERNELS['norms'] (models.pyx:1570-1571)   */
int amx_set_debug_x(amx_ctx *ctx, double *d_x);

/* ---- diagnosis / tests of the support seeds (csrc/amx_seed.hpp; no counterpart in the reference): copies a workspace
 * buffer of the LAST NODDI fit of this ctx -- which = 0: voxel permutation int32[n] (bucket order), 1: projected signals
 * f64[n][12] (bucket order), 2: support seeds uint64[n] (bucket order; up to 8 atom ids, one per byte, >= 0xf0 = empty;
 * all ones = no seed), 3: projected clipped signals of the LASSO stage f64[n][12], 4: LASSO passive-set seeds uint64[n][4]
 * (bit j = atom j; word 3 all ones = no seed) -- or a table of the dictionary `lut` -- 10: orientation bases U
 * f64[ndirs][nS][12], 11: compressed dictionaries S = U'A f64[ndirs][n_atoms][12], 12 / 13: the same for the LASSO stage's
 * dictionary, f64[ndirs][nS][12] / f64[ndirs][n_wm][12] (the seed solver reads the first 8 components) -- into the HOST buffer dst (synchronises the device).              */
int amx_debug_fetch(amx_ctx *ctx, const amx_lut *lut, int which, void *dst, size_t bytes);

/* ---- the solvers themselves, batched.  What the reference's FFI binds for this path is (models.pyx:18)
 *     from cyspams.interfaces cimport nnls, lasso
 *     nnls (&A[0,0], &y[i,0], m, n, &x[0], rnorm)                        models.pyx:911, 940
 *     lasso(&A[0,0], &y[i,0], m, n, 1, &x[0], lambda1, lambda2)          models.pyx:615, 926, 1238, 1569
 * one call per voxel, A column-major m x n with leading dimension m (a slice of the LUT), y one signal, x fully written.
 * A model injected through AMICO_WIP_MODELS (models.pyx:20-26) that writes its own _fit around these two calls reaches the GPU
 * solvers through the batched forms below: the dictionaries are uploaded once (one per LUT orientation, or a single one), every
 * voxel names its dictionary by index -- the `lut_idx` the reference computes per voxel (models.pyx:904) -- and all voxels are
 * solved by one call.
 *   amx_dict_upload    A f64[n_dicts][n][m]: n_dicts dictionaries, each column-major m x n with leading dimension m
 *                      (n <= 256 atoms, m <= 512 samples; a dictionary that does not fit a compute unit's LDS as fp64 is read from HBM / L2 instead: slower, same results; supports of up to 48 atoms)
 *   amx_nnls_batched   x_v = argmin_{x >= 0} ||A_d x - y_v||_2,  d = dict_idx[v] (NULL: the single dictionary),
 *                      Y f64[n_vox][m] -> X f64[n_vox][n] (exact zeros off the support), rnorm f64[n_vox] = ||A x - y||_2 or NULL
 *   amx_lasso_batched  x_v = argmin_{x >= 0} 1/2 ||y_v - A_d x||^2 + lambda1 sum(x) + lambda2/2 ||x||^2   (SPAMS lasso, mode
 *                      PENALTY, pos = true: what cyspams.lasso computes for p = 1; any lambda1, lambda2 >= 0)
 * Lawson-Hanson / its elastic-net form with a thin QR of the passive columns, one wavefront per voxel, strict Kuhn-Tucker stop
 * (csrc/amx_solver.hpp); a support of more than 48 atoms is beyond it (AMX_E_OVERFLOW -- AMICO's problems end at ~25).  A voxel with a non-finite signal gets NaN; an index outside [0, n_dicts) is reported like a bad
 * direction (AMX_E_DIR_OOB, the first offending voxel in the message) and its x stays zero.  The *_device forms take device
 * pointers and a hipStream_t and are asynchronous (amx_sync_status returns the status).                                      */
typedef struct amx_dict amx_dict;
int  amx_dict_upload(amx_ctx *ctx, const double *A, int m, int n, int n_dicts, amx_dict **out);
void amx_dict_destroy(amx_dict *dict);
int amx_nnls_batched(amx_ctx *ctx, const amx_dict *dict, const int32_t *dict_idx, const double *y, int64_t n_vox,
                     double *x, double *rnorm);
int amx_lasso_batched(amx_ctx *ctx, const amx_dict *dict, const int32_t *dict_idx, const double *y, int64_t n_vox,
                      double lambda1, double lambda2, double *x);
int amx_nnls_batched_device(amx_ctx *ctx, const amx_dict *dict, const int32_t *d_dict_idx, const double *d_y, int64_t n_vox,
                            double *d_x, double *d_rnorm, void *hip_stream);
int amx_lasso_batched_device(amx_ctx *ctx, const amx_dict *dict, const int32_t *d_dict_idx, const double *d_y, int64_t n_vox,
                             double lambda1, double lambda2, double *d_x, void *hip_stream);

/* ---- next rows of the hot-path table (SURVEY.md section 8 f): the steps either side of model.fit ---- */

/* (f1) principal directions, core.py:431-436 + 456-458:
 *     DTI = dipy.reconst.dti.TensorModel(gtab, fit_method='OLS');  DIRs = np.squeeze(DTI.fit(y).directions)
 * i.e. per voxel  p = pinv(design_matrix(gtab)) @ log(max(y, min_signal)),  D = lower-triangular p[0:6]
 * (Dxx Dxy Dyy Dxz Dyz Dzz), direction = eigenvector of the largest eigenvalue of D (dipy/reconst/dti.py:
 * TensorModel.fit, ols_fit_tensor, decompose_tensor; dipy>=1.4.1, requirements.txt:3).  The sign of an
 * eigenvector is not defined (LAPACK's choice in the reference); dir_to_lut_idx folds it away (lut.pyx:335-338).
 * inv_design f64[7][nS] (host, C-order) = numpy.linalg.pinv(design matrix) -- one-off per scheme, host side;
 * min_signal = dipy's MIN_POSITIVE_SIGNAL (1e-4) unless the caller configured another.                        */
typedef struct amx_dti amx_dti;
int  amx_dti_create(amx_ctx *ctx, const double *inv_design, int nS, double min_signal, amx_dti **out);
void amx_dti_destroy(amx_dti *h);
/* y f64[n_vox][nS] -> dirs f64[n_vox][3]; host buffers (blocking) / device buffers (enqueued on hip_stream) */
int amx_dti_directions(amx_ctx *ctx, const amx_dti *h, const double *y, int64_t n_vox, double *out_dirs);
/* (_f32: float32 signals, the dtype amx_prep_gather_device_f32 leaves them in -- same arithmetic, half the bytes)                */
int amx_dti_directions_device_f32(amx_ctx *ctx, const amx_dti *h, const float *d_y, int64_t n_vox, double *d_dirs, void *hip_stream);
int amx_dti_directions_device(amx_ctx *ctx, const amx_dti *h, const double *d_y, int64_t n_vox,
                              double *d_dirs, void *hip_stream);

/* (f2, f3) signal preparation and result scatter, fused around the masked voxel list:
 *   core.py:209-223  mean_b0s = mean(img[..., b0_idx], axis=3); norm_factor = 1 / mean_b0s, 0 where
 *                    mean_b0s <= b0_min_signal * mean(mean_b0s[mean_b0s > 0]);  img[..., i] *= norm_factor
 *   core.py:225-227  doMergeB0: volumes -> [mean of the b0 volumes] + the DWI volumes
 *   core.py:229-252  doDirectionalAverage: volumes -> [mean of the b0s] + the mean of every shell (sorted by b)
 *   core.py:451-452  y = img[mask == 1, :].astype(double); y[y < 0] = 0
 *   core.py:472-498  RESULTS[...] = zeros(float32 volume); RESULTS[...][mask == 1, :] = per-voxel values
 * All arithmetic before the float64 cast is float32 in the reference's operation order (numpy reduces the
 * fancy-indexed volumes sequentially in index order), so `y` is bit-identical to the reference's.
 *
 * A plan holds the geometry: dims = (X, Y, Z); strides = element strides of the float32 image along
 * (x, y, z, volume) -- any layout: C order, or the Fortran order nibabel hands out; rank = int32[X][Y][Z]
 * (host, C order): position of the voxel in the masked list (its row in `y`), -1 outside the mask -- i.e.
 * cumsum(mask == 1) - 1 in C order, which is the order `img[mask == 1, :]` enumerates voxels in.
 * Output volume j of `y` is the float32 mean of the input volumes group_idx[group_ptr[j] .. group_ptr[j+1])
 * (a group of one = plain copy): identity groups, the b0-merge or the shell average.  b0_idx = scheme.b0_idx.
 * overwrite_in_order != 0 reproduces core.py:231-245 literally: the shell averages are written into a VIEW of
 * the first n_out volumes of the image while later groups still read from it, so output j replaces input
 * volume j before group j+1 is averaged (harmless when the b0 volumes come first and the shells are stored in
 * b-value order; otherwise the reference averages already-replaced volumes, and so does this).            */
typedef struct amx_prep amx_prep;
int  amx_prep_create(amx_ctx *ctx, const int64_t dims[3], const int64_t strides[4], int nS,
                     const int32_t *rank, int64_t n_vox, const int32_t *group_ptr,
                     const int32_t *group_idx, int n_out, const int32_t *b0_idx, int n_b0,
                     int overwrite_in_order, amx_prep **out);
void amx_prep_destroy(amx_prep *p);
/* img -> y f64[n_vox][n_out] (+ mean_b0 f32[n_vox] of the masked voxels when normalize != 0 and the pointer is
 * not NULL).  normalize = doNormalizeSignal; b0_threshold = the right-hand side of core.py:217 (0 by default).
 * A plan carries the work counter of its gather kernel: ONE gather of a plan in flight at a time (calls on one stream are).  */
int amx_prep_gather(amx_ctx *ctx, const amx_prep *p, const float *img, int normalize, float b0_threshold,
                    double *out_y, float *out_mean_b0);
/* (_f32: the prepared signals stay float32 -- every value of core.py:209-268 IS a float32, core.py:451-452 only widen them; the
 * amx_*_fit_device_f32 / amx_dti_directions_device_f32 calls read them in place)                                                  */
int amx_prep_gather_device_f32(amx_ctx *ctx, const amx_prep *p, const float *d_img, int normalize, float b0_threshold,
                               float *d_y, float *d_mean_b0, void *hip_stream);
int amx_prep_gather_device(amx_ctx *ctx, const amx_prep *p, const float *d_img, int normalize,
                           float b0_threshold, double *d_y, float *d_mean_b0, void *hip_stream);
/* The gather with the tensor fit taken along (round 5): y AND the principal directions of core.py:431-436, 456-458 in ONE pass over
 * the image -- lane = voxel contracts log(max(y, min_signal)) with the helper's pseudo-inverse while the voxel's values are in the
 * gather's LDS tile; what amx_prep_gather_device[_f32] followed by amx_dti_directions_device[_f32] computes (y bit-identical, the
 * directions to rounding: the sum over the volumes runs in index order here).  h: amx_dti_create for the plan's n_out volumes.     */
int amx_prep_gather_directions_device(amx_ctx *ctx, const amx_prep *p, const amx_dti *h, const float *d_img, int normalize,
                                      float b0_threshold, double *d_y, float *d_mean_b0, double *d_dirs, void *hip_stream);
int amx_prep_gather_directions_device_f32(amx_ctx *ctx, const amx_prep *p, const amx_dti *h, const float *d_img, int normalize,
                                          float b0_threshold, float *d_y, float *d_mean_b0, double *d_dirs, void *hip_stream);
/* self.mean_b0s of EVERY voxel (core.py:213), float32 [X][Y][Z] in C order: input of the threshold above */
int amx_prep_mean_b0(amx_ctx *ctx, const amx_prep *p, const float *img, float *out_mean_b0_volume);
int amx_prep_mean_b0_device(amx_ctx *ctx, const amx_prep *p, const float *d_img, float *d_mean_b0_volume,
                            void *hip_stream);
/* values f64[n_vox][n_cols] -> float32 volume [X][Y][Z][n_cols] (C order), zero outside the mask */
int amx_prep_scatter(amx_ctx *ctx, const amx_prep *p, const double *values, int n_cols, float *out_volume);
int amx_prep_scatter_device(amx_ctx *ctx, const amx_prep *p, const double *d_values, int n_cols,
                            float *d_volume, void *hip_stream);

/* (f4) LUT resampling to the subject's scheme, lut.pyx:274-311 `resample_kernel` (called per atom by
 * NODDI.resample models.pyx:754-792, FreeWater.resample :1113-1144, ...):
 *     KR = np.ones((ndirs, nS), float32);  KR[i, idx_out] = np.dot(Ylm_out, KRlm[i, :])   for i in range(ndirs)
 * batched over all atoms: lm f32[n_rows][n_sh] = the rotated SH coefficients of n_rows = atoms * ndirs
 * (atom, orientation) pairs (an isotropic atom is one row), ylm_out f32[n_out][n_sh] and idx_out int32[n_out]
 * from aux_structures_resample (lut.pyx:196-224); out f32[n_rows][nS] (host).  One float32 GEMM on the matrix
 * cores; sums are ordered differently from the reference's BLAS sgemv (float32 rounding, ~1e-6 relative).     */
int amx_lut_resample(amx_ctx *ctx, const float *lm, int64_t n_rows, int n_sh, const float *ylm_out,
                     const int32_t *idx_out, int n_out, int nS, float *out);

/* (f4 tail) lut.pyx:227-271 `rotate_kernel` fused with the resampling above, for all anisotropic atoms of a model:
 *     KRlm[i, idx_OUT[s]] = AUX['const'] * Klm[s][AUX['idx_m0']] * AUX['Ylm_rot'][i]        (addition theorem, :262-264)
 * is never materialised (ndirs x nSH x shells floats per atom: 52 MB for NODDI) -- the GEMM forms its left operand in
 * registers from  zonal f32[n_atoms][n_shells * n_sh_shell] = const * Klm[s][idx_m0]  (host, a few KB per atom: the SH
 * fit of the z-aligned response function) and  ylm_rot f32[ndirs][n_sh_shell] = AUX['Ylm_rot'];
 * out f32[n_atoms][ndirs][nS] = what resample_kernel returns for rotate_kernel's output.                          */
int amx_lut_rotate_resample(amx_ctx *ctx, const float *zonal, int n_atoms, const float *ylm_rot, int ndirs,
                            int n_sh_shell, int n_shells, const float *ylm_out, const int32_t *idx_out, int n_out,
                            int nS, float *out);

/* ---- measurement hooks (bench.py): HIP-event time of the solver kernels of the LAST
 * *_fit_device call on this ctx, measured on the stream they were launched on.
 * which: 0 = all kernels of the call, 1..3 = solver stage kernels (NODDI: the wavefront-per-voxel kernels of NNLS-1, LASSO,
 * NNLS-3 incl. their re-run kernels -- with seeds on they see only the voxels the Gram-space certificates left over;
 * FreeWater/SANDI: 1 = the single solver kernel), 4 = the last amx_dti_directions_device / amx_prep_gather_device kernel,
 * 5..7 = NODDI kernels ahead of stage 1 / 2 / 3 (A'y on the matrix cores + seed solver + Gram-space certificate,
 * csrc/amx_seed.hpp; an error if seeds are off), 8 = k_nnls_seed<1> alone, 9 = k_lasso_seed alone.
 * Requires amx_set_profiling(1) -- or amx_set_profiling(2 + which): only that pair of events is recorded (every recorded event is a
 * packet of the stream, ~5 us of the call: the full set costs a NODDI fit ~70 us, which is why profiling is off by default). */
int amx_set_profiling(amx_ctx *ctx, int enable);
int amx_last_kernel_ms(amx_ctx *ctx, int which, float *out_ms);
/* solver statistics of the last call: out[0]=voxels re-run with the large active-set
 * variant (stage sum), out[1]=voxels hitting the iteration cap, out[2..3] reserved        */
int amx_last_stats(amx_ctx *ctx, int64_t out[4]);
/* NODDI, the seed -> certificate chain (csrc/amx_seed.hpp; no counterpart in the reference, whose every voxel takes one path,
 * models.pyx:902-981): how the voxels fitted since the previous amx_sync_status were settled.  out[0] = voxels that took the
 * chain (0: the calls were too small / the dictionary has no bases), out[1..3] = voxels the Gram-space certificates of stage 1 /
 * the LASSO stage / stage 3 could NOT settle and handed to the wavefront-per-voxel kernels, out[4] = voxels whose stage-2 signal
 * y2 = max(0, y - x_iso iso) was clipped (models.pyx:924-925), out[5..7] reserved.  Certification rate of stage k = 1 - out[k] / out[0]. */
int amx_last_seed_stats(amx_ctx *ctx, int64_t out[8]);
/* Host-buffer entry points, float64 signals: evaluation.y is the float64 cast of a float32 image (core.py:136, 209-223, 451), so its
 * values cross PCIe as float32 -- host threads narrow slices into pinned slots and CHECK every element; one value that is not a
 * float32 (or a NaN) and that batch and the rest of the call are copied as they are (csrc/amx_stage.hpp).  The kernels read the
 * caller's values bit for bit either way.  Returns the number of batches of the LAST host-buffer call that travelled as float32
 * (0: small call, not float32 data, AMX_HOST_NARROW=0, or a *_f32 call -- those are float32 already).  No counterpart in the
 * reference (its fit reads host memory in place, models.pyx:902).                                                              */
int amx_last_host_narrowed(amx_ctx *ctx);

/* The host threads of the float32 transport above (made at the first large float64 host-buffer call of the ctx): out[0] = threads, out[1] =
 * first CPU and out[2] = number of physical cores of the share of the device's NUMA node they are striped over, out[3] = the device.  A
 * node's cores are divided among the devices that hang on it (one pool per device, whether one process drives them all --
 * amico_amd.models: AMX_DEVICES -- or one process each), so that sibling pools never share a core; AMX_HOST_SIBLINGS="i/n" forces the share. */
int amx_host_pool_info(amx_ctx *ctx, int out[4]);
/* The kernels the LAST *_fit / *_fit_device call on this ctx enqueued, in launch order, as text ("k_noddi_gemm<false,25,9> -> k_nnls_seed<1,8,occ2>
 * -> ..."; the first batch of a host-buffer call): which of the library's paths a dictionary shape / call size / solver parameters took.
 * bench.py labels its roofline kernel from it instead of from a literal.  No counterpart in the reference (one path, models.pyx:902-981). */
int amx_last_path(amx_ctx *ctx, char *buf, int cap);

/* device self-test of the wavefront primitives (DPP reductions, broadcasts): writes 12 rows of
 * 64 doubles (sum, max, min, bcast lane 37, next-lane, popcount(ballot v>0), int bcast, v, and
 * the four batched sums of wave_sum4) */
int amx_selftest(amx_ctx *ctx, double *out768);

#ifdef __cplusplus
}
#endif
#endif /* AMICO_AMD_H */
