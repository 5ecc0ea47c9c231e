/*
 * amico_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the per-voxel AMICO fit path of daducci/AMICO v2.1.0
 * (reference files cited per function in amico_oracle.c).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors for this path
 *  and its solver dependency (spams-cython >= 1.0.0, unpinned,
 * pyproject.toml:5) is absent from /root/reference, so the reference path cannot
 * be compiled or imported here.  The oracle is pinned instead by KKT certificates,
 * by scipy.optimize.nnls / sklearn ElasticNet(positive=True) cross-checks and by
 * golden fixtures whose dictionaries come from the importable amico.synthesis
 * (tests/golden/make_fixtures.py).
 */
#ifndef AMICO_ORACLE_H
#define AMICO_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* lut.pyx:316-356.  Returns LUT index, or -1 if (ii1,ii2) is out of [0,180]^2
 * (the reference raises RuntimeError); ii1/ii2 (optional) receive the degrees. */
int amo_dir_to_lut_idx(const double dir[3], const int16_t *htable, int *ii1, int *ii2);

/* cyspams.interfaces.nnls semantics (models.pyx:911,940): Lawson-Hanson 1974.
 * A is column-major m x n (ld = m), untouched; returns mode (1 ok, 3 iteration cap). */
int amo_nnls(const double *A, const double *y, int m, int n, double *x, double *rnorm);

/* cyspams.interfaces.lasso semantics (models.pyx:926,1238,1569) for p=1:
 * argmin_{x>=0} 1/2||y-Ax||^2 + lambda1*sum(x) + lambda2/2*||x||^2, LARS/homotopy. */
int amo_lasso(const double *A, const double *y, int m, int n, double *x,
              double lambda1, double lambda2);

typedef struct {
    int n_vox, nS, ndirs, n_wm;      /* n_wm = len(IC_ODs)*len(IC_VFs) */
    int is_exvivo;
    int dwi_count;
    const int64_t *dwi_idx;          /* scheme.dwi_idx */
    const float *wm;                 /* KERNELS['wm']  f32 [n_wm][ndirs][nS] */
    const float *iso;                /* KERNELS['iso'] f32 [nS] */
    const double *norms;             /* KERNELS['norms'] f64 [dwi_count][n_wm] */
    const float *icvf, *kappa;       /* f32 [n_wm] */
    const int16_t *htable;           /* int16 [181*181] */
    double lambda1, lambda2;
    int compute_rmse, compute_nrmse, compute_mod;
    int nthreads;                    /* contiguous chunks, models.pyx:204-211 */
} amo_noddi_args;

/* models.pyx:816-991.  y f64[n_vox][nS], dirs f64[n_vox][3];
 * estimates f64[n_vox][3(+1)], rmse/nrmse f64[n_vox] or NULL, mod f64[n_vox][2] or NULL.
 * x_dbg (optional) f64[n_vox][3][n_atoms]: coefficients after each of the 3 stages.
 * returns 0, or -(voxel+1) of the first out-of-bounds direction. */
int64_t amo_noddi_fit(const amo_noddi_args *a, const double *y, const double *dirs,
                      double *estimates, double *rmse, double *nrmse, double *mod,
                      double *x_dbg);

typedef struct {
    int n_vox, nS, ndirs, n_perp, n_iso;
    int is_mouse;
    const float *D;                  /* KERNELS['D']   f32 [n_perp][ndirs][nS] */
    const float *CSF;                /* KERNELS['CSF'] f32 [n_iso][nS] */
    const int16_t *htable;
    double lambda1, lambda2;
    int compute_rmse, compute_nrmse, save_corrected;
    int nthreads;
} amo_fw_args;

/* models.pyx:1168-1286. estimates f64[n_vox][2 or 4]; y_corr f64[n_vox][nS] or NULL */
int64_t amo_freewater_fit(const amo_fw_args *a, const double *y, const double *dirs,
                          double *estimates, double *rmse, double *nrmse, double *y_corr,
                          double *x_dbg);

typedef struct {
    int n_vox, nS, n_rs, n_in, n_iso;
    const double *signal;            /* KERNELS['signal'] f64 col-major [nS][n_atoms] */
    const double *norms;             /* KERNELS['norms'] f64 [n_atoms] */
    const double *Rs, *d_in, *d_isos;
    double lambda1, lambda2;
    int compute_rmse, compute_nrmse;
    int nthreads;
} amo_sandi_args;

/* models.pyx:1509-1627. estimates f64[n_vox][6] */
int64_t amo_sandi_fit(const amo_sandi_args *a, const double *y,
                      double *estimates, double *rmse, double *nrmse, double *x_dbg);

typedef struct {
    int n_vox, nS, ndirs, n_rs, n_perp, n_iso;
    const float *wmr;                /* KERNELS['wmr'] f32 [n_rs][ndirs][nS]   (cylinders)  */
    const float *wmh;                /* KERNELS['wmh'] f32 [n_perp][ndirs][nS] (zeppelins)  */
    const float *iso;                /* KERNELS['iso'] f32 [n_iso][nS]         (balls)      */
    const double *Rs;                /* model.Rs f64 [n_rs] (metres) */
    const int16_t *htable;
    double lambda1, lambda2;
    int compute_rmse, compute_nrmse;
    int nthreads;
} amo_czb_args;

/* CylinderZeppelinBall._fit models.pyx:526-652. estimates f64[n_vox][3] = v, a, d.
 * (The reference reads self.isExvivo, which CylinderZeppelinBall never defines, models.pyx:435/549 -- with it set to
 * False by hand the loop below is what runs; the "ex vivo" branch only adds an all-zero atom, :552-554.) */
int64_t amo_czb_fit(const amo_czb_args *a, const double *y, const double *dirs,
                    double *estimates, double *rmse, double *nrmse, double *x_dbg);

#ifdef __cplusplus
}
#endif
#endif
