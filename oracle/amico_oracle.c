/*
 * amico_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see amico_oracle.h).
 *
 * Restates, in plain C, the per-voxel fit path of daducci/AMICO v2.1.0:
 *   amico/lut.pyx:316-356      dir_to_lut_idx
 *   amico/models.pyx:47-71     _compute_rmse / _compute_nrmse
 *   amico/models.pyx:186-217   BaseModel.fit chunking (contiguous chunks per thread)
 *   amico/models.pyx:816-991   NODDI._fit
 *   amico/models.pyx:1168-1286 FreeWater._fit
 *   amico/models.pyx:1509-1627 SANDI._fit
 * The two solver primitives live in a third-party dependency that is NOT in the
 * reference tree: spams-cython (pyproject.toml:5, ">=1.0.0", no lock file), called at
 * models.pyx:615,911,926,940,1238,1569 as cyspams.interfaces.nnls / .lasso.  Their
 * published algorithms are restated here:
 *   nnls  -- Lawson & Hanson, "Solving Least Squares Problems" (1974), ch. 23,
 *            algorithm NNLS with Householder QR and Givens down-dating;
 *   lasso -- SPAMS lasso(mode=PENALTY, pos=true): LARS / homotopy (Efron et al. 2004,
 *            lasso modification) on the Gram matrix A'A + lambda2*I, restricted to
 *            non-negative coefficients.
 * PARITY UNPINNED by reference tests (there are none); pinned by tests/test_oracle_*.py.
 */
#include "amico_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ lut.pyx:316-356 */
int amo_dir_to_lut_idx(const double dir[3], const int16_t *htable, int *pi1, int *pi2)
{
    /* the reference flips `direction` in place (lut.pyx:335-338); the oracle works on a
     * copy and never mutates the caller's array (axial data: the flip is immaterial). */
    double d0 = dir[0], d1 = dir[1], d2 = dir[2];
    double i1, i2;
    int ii1, ii2;
    if (d1 < 0.0) { d0 = -d0; d1 = -d1; d2 = -d2; }
    i2 = fmod(atan2(d1, d0), 2.0 * M_PI);
    if (i2 < 0.0) i2 = fmod(i2 + 2.0 * M_PI, 2.0 * M_PI);
    if (i2 > M_PI) {
        i2 = fmod(atan2(-d1, -d0), 2.0 * M_PI);
        i1 = atan2(sqrt(d0 * d0 + d1 * d1), -d2);
    } else {
        i1 = atan2(sqrt(d0 * d0 + d1 * d1), d2);
    }
    {
        double r1 = round(i1 / M_PI * 180.0), r2 = round(i2 / M_PI * 180.0);
        /* NaN / huge -> out of bounds (the reference's <int> cast is UB there) */
        if (!(r1 >= -1.0 && r1 <= 181.0) || !(r2 >= -1.0 && r2 <= 181.0)) {
            if (pi1) *pi1 = -1;
            if (pi2) *pi2 = -1;
            return -1;
        }
        ii1 = (int)r1; ii2 = (int)r2;
    }
    if (pi1) *pi1 = ii1;
    if (pi2) *pi2 = ii2;
    if (ii1 < 0 || ii1 > 180 || ii2 < 0 || ii2 > 180) return -1;
    return (int)htable[ii1 * 181 + ii2];
}

/* ------------------------------------------------------------------ Lawson-Hanson NNLS */
/* Householder transformation (L&H algorithm H12).  mode 1: construct and apply to u
 * (pivot lpivot, rows l1..m-1 are zeroed), mode 2: apply previously constructed one.
 * u has stride iue; c vectors: ncv of them, element stride ice, vector stride icv. */
static void h12(int mode, int lpivot, int l1, int m, double *u, int iue, double *up,
                double *c, int ice, int icv, int ncv)
{
    double cl, clinv, sm, b;
    int i, j;
    if (lpivot < 0 || lpivot >= l1 || l1 >= m) return;
    cl = fabs(u[lpivot * iue]);
    if (mode == 1) {
        for (j = l1; j < m; j++) { double t = fabs(u[j * iue]); if (t > cl) cl = t; }
        if (cl <= 0.0) return;
        clinv = 1.0 / cl;
        sm = (u[lpivot * iue] * clinv) * (u[lpivot * iue] * clinv);
        for (j = l1; j < m; j++) sm += (u[j * iue] * clinv) * (u[j * iue] * clinv);
        cl *= sqrt(sm);
        if (u[lpivot * iue] > 0.0) cl = -cl;
        *up = u[lpivot * iue] - cl;
        u[lpivot * iue] = cl;
    } else if (cl <= 0.0) {
        return;
    }
    if (ncv <= 0) return;
    b = (*up) * u[lpivot * iue];
    if (b >= 0.0) return;
    b = 1.0 / b;
    for (j = 0; j < ncv; j++) {
        double *cj = c + (size_t)j * icv;
        sm = cj[lpivot * ice] * (*up);
        for (i = l1; i < m; i++) sm += cj[i * ice] * u[i * iue];
        if (sm != 0.0) {
            sm *= b;
            cj[lpivot * ice] += sm * (*up);
            for (i = l1; i < m; i++) cj[i * ice] += sm * u[i * iue];
        }
    }
}

/* Givens rotation (L&H G1) */
static void g1(double a, double b, double *c, double *s, double *sig)
{
    double xr, yr;
    if (fabs(a) > fabs(b)) {
        xr = b / a; yr = sqrt(1.0 + xr * xr);
        *c = copysign(1.0 / yr, a); *s = (*c) * xr; *sig = fabs(a) * yr;
    } else if (b != 0.0) {
        xr = a / b; yr = sqrt(1.0 + xr * xr);
        *s = copysign(1.0 / yr, b); *c = (*s) * xr; *sig = fabs(b) * yr;
    } else {
        *sig = 0.0; *c = 0.0; *s = 1.0;
    }
}

/* core working in place on a (m x n, col-major) and b (m); w,zz,index are workspaces */
static int nnls_inplace(double *a, int m, int n, double *b, double *x, double *rnorm,
                        double *w, double *zz, int *index)
{
    const double factor = 0.01;
    int i, ii, ip, iter = 0, itmax = 3 * n, iz, iz1 = 0, iz2 = n - 1, izmax = 0, j = 0, jj = 0,
        jz, l, mode = 1, npp1 = 0, nsetp = 0;
    double alpha, asave, cc, sm, ss, t, temp, unorm, up = 0.0, wmax, ztest, dummy = 0.0;

    for (i = 0; i < n; i++) { x[i] = 0.0; index[i] = i; }

    for (;;) {
        /* quit if all coefficients are already in the solution or m columns are */
        if (iz1 > iz2 || nsetp >= m) break;
        /* dual vector w on the set Z */
        for (iz = iz1; iz <= iz2; iz++) {
            j = index[iz];
            sm = 0.0;
            for (l = npp1; l < m; l++) sm += a[(size_t)j * m + l] * b[l];
            w[j] = sm;
        }
        for (;;) {
            wmax = 0.0;
            for (iz = iz1; iz <= iz2; iz++) {
                j = index[iz];
                if (w[j] > wmax) { wmax = w[j]; izmax = iz; }
            }
            if (wmax <= 0.0) goto terminate;   /* Kuhn-Tucker conditions hold */
            iz = izmax; j = index[iz];
            /* sign of w(j) ok: test column j for linear independence of the set P */
            asave = a[(size_t)j * m + npp1];
            h12(1, npp1, npp1 + 1, m, a + (size_t)j * m, 1, &up, &dummy, 1, 1, 0);
            unorm = 0.0;
            for (l = 0; l < nsetp; l++) unorm += a[(size_t)j * m + l] * a[(size_t)j * m + l];
            unorm = sqrt(unorm);
            temp = unorm + fabs(a[(size_t)j * m + npp1]) * factor;
            if (temp - unorm > 0.0) {
                /* column j sufficiently independent: solve for ztest */
                for (l = 0; l < m; l++) zz[l] = b[l];
                h12(2, npp1, npp1 + 1, m, a + (size_t)j * m, 1, &up, zz, 1, 1, 1);
                ztest = zz[npp1] / a[(size_t)j * m + npp1];
                if (ztest > 0.0) break;       /* accept j */
            }
            /* reject j as candidate; restore and try next largest w */
            a[(size_t)j * m + npp1] = asave;
            w[j] = 0.0;
        }
        /* move j from set Z to set P */
        for (l = 0; l < m; l++) b[l] = zz[l];
        index[iz] = index[iz1];
        index[iz1] = j;
        iz1++;
        nsetp = npp1 + 1;
        npp1++;
        if (iz1 <= iz2)
            for (jz = iz1; jz <= iz2; jz++) {
                jj = index[jz];
                h12(2, nsetp - 1, npp1, m, a + (size_t)j * m, 1, &up, a + (size_t)jj * m, 1, m, 1);
            }
        if (nsetp != m)
            for (l = npp1; l < m; l++) a[(size_t)j * m + l] = 0.0;
        w[j] = 0.0;
        /* solve the triangular system, solution in zz */
        for (l = 0; l < nsetp; l++) {
            ip = nsetp - 1 - l;
            if (l != 0)
                for (ii = 0; ii <= ip; ii++) zz[ii] -= a[(size_t)jj * m + ii] * zz[ip + 1];
            jj = index[ip];
            zz[ip] /= a[(size_t)jj * m + ip];
        }
        /* secondary loop */
        for (;;) {
            if (++iter > itmax) { mode = 3; goto terminate; }
            alpha = 2.0;
            for (ip = 0; ip < nsetp; ip++) {
                l = index[ip];
                if (zz[ip] <= 0.0) {
                    t = -x[l] / (zz[ip] - x[l]);
                    if (alpha > t) { alpha = t; jj = ip; }
                }
            }
            if (alpha == 2.0) break;          /* all new coefficients feasible */
            for (ip = 0; ip < nsetp; ip++) {
                l = index[ip];
                x[l] += alpha * (zz[ip] - x[l]);
            }
            /* move coefficient i = index[jj] from P to Z (repeat while infeasible) */
            i = index[jj];
            for (;;) {
                x[i] = 0.0;
                if (jj != nsetp - 1) {
                    jj++;
                    for (j = jj; j < nsetp; j++) {
                        ii = index[j];
                        index[j - 1] = ii;
                        g1(a[(size_t)ii * m + j - 1], a[(size_t)ii * m + j], &cc, &ss,
                           &a[(size_t)ii * m + j - 1]);
                        a[(size_t)ii * m + j] = 0.0;
                        for (l = 0; l < n; l++)
                            if (l != ii) {
                                temp = a[(size_t)l * m + j - 1];
                                a[(size_t)l * m + j - 1] = cc * temp + ss * a[(size_t)l * m + j];
                                a[(size_t)l * m + j] = -ss * temp + cc * a[(size_t)l * m + j];
                            }
                        temp = b[j - 1];
                        b[j - 1] = cc * temp + ss * b[j];
                        b[j] = -ss * temp + cc * b[j];
                    }
                }
                npp1 = nsetp - 1;
                nsetp--;
                iz1--;
                index[iz1] = i;
                /* all coefficients in P should be feasible; if not (round-off) remove too */
                {
                    int again = 0;
                    for (jj = 0; jj < nsetp; jj++) {
                        i = index[jj];
                        if (x[i] <= 0.0) { again = 1; break; }
                    }
                    if (!again) break;
                }
            }
            for (l = 0; l < m; l++) zz[l] = b[l];
            for (l = 0; l < nsetp; l++) {
                ip = nsetp - 1 - l;
                if (l != 0)
                    for (ii = 0; ii <= ip; ii++) zz[ii] -= a[(size_t)jj * m + ii] * zz[ip + 1];
                jj = index[ip];
                zz[ip] /= a[(size_t)jj * m + ip];
            }
        }
        for (ip = 0; ip < nsetp; ip++) x[index[ip]] = zz[ip];
    }
terminate:
    sm = 0.0;
    for (l = npp1; l < m; l++) sm += b[l] * b[l];
    if (rnorm) *rnorm = sqrt(sm);
    return mode;
}

int amo_nnls(const double *A, const double *y, int m, int n, double *x, double *rnorm)
{
    /* inputs are left untouched (callers reuse A after the call: models.pyx:917-939) */
    double *a, *b, *w, *zz;
    int *index, mode;
    if (m <= 0 || n <= 0) return 2;
    a = (double *)malloc(sizeof(double) * ((size_t)m * n + (size_t)m * 2 + n));
    index = (int *)malloc(sizeof(int) * n);
    b = a + (size_t)m * n; zz = b + m; w = zz + m;
    memcpy(a, A, sizeof(double) * (size_t)m * n);
    memcpy(b, y, sizeof(double) * m);
    mode = nnls_inplace(a, m, n, b, x, rnorm, w, zz, index);
    free(index); free(a);
    return mode;
}

/* ------------------------------------------------------------------ LARS / homotopy lasso */
/* dense Cholesky helpers on the (small) active Gram block, row-major k x k, ld = ldk */
static int chol_factor(const double *G, int k, int ldk, double *L)
{
    int i, j, p;
    for (i = 0; i < k; i++)
        for (j = 0; j <= i; j++) {
            double s = G[i * ldk + j];
            for (p = 0; p < j; p++) s -= L[i * ldk + p] * L[j * ldk + p];
            if (i == j) { if (s <= 0.0) return -1; L[i * ldk + i] = sqrt(s); }
            else L[i * ldk + j] = s / L[j * ldk + j];
        }
    return 0;
}
static void chol_solve(const double *L, int k, int ldk, const double *rhs, double *out)
{
    int i, p;
    for (i = 0; i < k; i++) {
        double s = rhs[i];
        for (p = 0; p < i; p++) s -= L[i * ldk + p] * out[p];
        out[i] = s / L[i * ldk + i];
    }
    for (i = k - 1; i >= 0; i--) {
        double s = out[i];
        for (p = i + 1; p < k; p++) s -= L[p * ldk + i] * out[p];
        out[i] = s / L[i * ldk + i];
    }
}

int amo_lasso(const double *A, const double *y, int m, int n, double *x,
              double lambda1, double lambda2)
{
    /* non-negative elastic net along the LARS regularisation path:
     * lam decreases from max_j corr_j to lambda1; active atoms keep corr_j == lam.
     * Gram columns (A'a_j + lambda2 e_j) are formed only for atoms that become active. */
    int kmax = (n < m + n ? n : m + n), k = 0, i, j, p, it, maxit = 8 * n + 16, status = 0;
    int *act = (int *)malloc(sizeof(int) * (size_t)(kmax + 1));
    int *pos = (int *)malloc(sizeof(int) * (size_t)n);          /* atom -> slot or -1 */
    double *c0 = (double *)malloc(sizeof(double) * (size_t)n * 3);
    double *c = c0 + n, *aj = c + n;
    double *Gc = (double *)malloc(sizeof(double) * (size_t)(kmax + 1) * n);  /* slot-major */
    int ldk = kmax + 1;
    double *Gk = (double *)malloc(sizeof(double) * (size_t)ldk * ldk * 2);
    double *L = Gk + (size_t)ldk * ldk;
    double *u = (double *)malloc(sizeof(double) * (size_t)ldk * 3);
    double *xa = u + ldk, *rhs = xa + ldk;
    double lam;
    int dropped = -1;

    for (j = 0; j < n; j++) {
        double s = 0.0;
        for (i = 0; i < m; i++) s += A[(size_t)j * m + i] * y[i];
        c0[j] = c[j] = s; x[j] = 0.0; pos[j] = -1;
    }
    lam = -HUGE_VAL; p = -1;
    for (j = 0; j < n; j++) if (c[j] > lam) { lam = c[j]; p = j; }
    if (p < 0 || lam <= lambda1) goto done;

    for (it = 0; it < maxit; it++) {
        double gam, g_join = HUGE_VAL, g_drop = HUGE_VAL, g_tgt;
        int j_join = -1, k_drop = -1;
        if (p >= 0) {          /* atom p joins the active set */
            double *col = Gc + (size_t)k * n;
            for (j = 0; j < n; j++) {
                double s = 0.0;
                for (i = 0; i < m; i++) s += A[(size_t)j * m + i] * A[(size_t)p * m + i];
                col[j] = s;
            }
            col[p] += lambda2;
            act[k] = p; pos[p] = k; xa[k] = 0.0; k++;
            p = -1;
        }
        for (i = 0; i < k; i++)
            for (j = 0; j < k; j++) Gk[i * ldk + j] = Gc[(size_t)i * n + act[j]];
        if (chol_factor(Gk, k, ldk, L) != 0) { status = -1; break; }
        for (i = 0; i < k; i++) rhs[i] = 1.0;
        chol_solve(L, k, ldk, rhs, u);
        for (j = 0; j < n; j++) {
            double s = 0.0;
            for (i = 0; i < k; i++) s += Gc[(size_t)i * n + j] * u[i];
            aj[j] = s;
        }
        g_tgt = lam - lambda1;
        for (j = 0; j < n; j++) {
            double den, g;
            if (pos[j] >= 0 || j == dropped) continue;
            den = 1.0 - aj[j];
            if (den <= 0.0) continue;
            g = (lam - c[j]) / den;
            if (g < 0.0) g = 0.0;
            if (g < g_join) { g_join = g; j_join = j; }
        }
        for (i = 0; i < k; i++)
            if (u[i] < 0.0) {
                double g = -xa[i] / u[i];
                if (g < g_drop) { g_drop = g; k_drop = i; }
            }
        gam = g_tgt;
        if (g_join < gam) gam = g_join;
        if (g_drop < gam) gam = g_drop;
        for (i = 0; i < k; i++) xa[i] += gam * u[i];
        for (j = 0; j < n; j++) c[j] -= gam * aj[j];
        lam -= gam;
        for (i = 0; i < k; i++) c[act[i]] = lam;
        dropped = -1;
        if (gam >= g_tgt) break;                       /* reached lambda1 */
        if (g_drop <= g_join) {                        /* coefficient hits zero: leave */
            int q = act[k_drop];
            dropped = q; pos[q] = -1;
            for (i = k_drop; i < k - 1; i++) {
                act[i] = act[i + 1]; xa[i] = xa[i + 1]; pos[act[i]] = i;
                memcpy(Gc + (size_t)i * n, Gc + (size_t)(i + 1) * n, sizeof(double) * n);
            }
            k--;
            if (k == 0) {                              /* restart from the largest corr */
                lam = -HUGE_VAL;
                for (j = 0; j < n; j++) if (j != q && c[j] > lam) { lam = c[j]; p = j; }
                if (p < 0 || lam <= lambda1) break;
            }
        } else {
            p = j_join;
            if (k >= kmax) { status = -2; break; }
        }
    }
    if (it >= maxit) status = 3;
    /* polish on the final support: (G_aa) x_a = c0_a - lambda1 (removes path drift) */
    if (k > 0 && status == 0) {
        for (i = 0; i < k; i++)
            for (j = 0; j < k; j++) Gk[i * ldk + j] = Gc[(size_t)i * n + act[j]];
        if (chol_factor(Gk, k, ldk, L) == 0) {
            int ok = 1;
            for (i = 0; i < k; i++) rhs[i] = c0[act[i]] - lambda1;
            chol_solve(L, k, ldk, rhs, u);
            for (i = 0; i < k; i++) if (!(u[i] > 0.0)) ok = 0;
            if (ok) for (i = 0; i < k; i++) xa[i] = u[i];
        }
    }
    for (i = 0; i < k; i++) x[act[i]] = xa[i] > 0.0 ? xa[i] : 0.0;
done:
    free(u); free(Gk); free(Gc); free(c0); free(pos); free(act);
    return status;
}

/* ------------------------------------------------------------------ models.pyx:47-71 */
static double rmse_of(const double *A, int m, int n, const double *y, const double *x, double *yest)
{
    double acc = 0.0; int i, j;
    for (i = 0; i < m; i++) {
        yest[i] = 0.0;
        for (j = 0; j < n; j++) yest[i] += A[(size_t)j * m + i] * x[j];
        acc += (y[i] - yest[i]) * (y[i] - yest[i]) / (double)m;
    }
    return sqrt(acc);
}
static double nrmse_of(const double *A, int m, int n, const double *y, const double *x, double *yest)
{
    double den = 0.0, acc = 0.0; int i, j;
    for (i = 0; i < m; i++) {
        yest[i] = 0.0;
        den += y[i] * y[i];
        for (j = 0; j < n; j++) yest[i] += A[(size_t)j * m + i] * x[j];
    }
    if (den > 1e-16) {
        for (i = 0; i < m; i++) acc += (y[i] - yest[i]) * (y[i] - yest[i]) / den;
        return sqrt(acc);
    }
    return 0.0;
}

/* ------------------------------------------------------------------ threading helper */
typedef struct { void (*fn)(void *, int, int, int); void *ctx; int tid, i0, i1; } amo_job;
static void *job_tramp(void *p) { amo_job *j = (amo_job *)p; j->fn(j->ctx, j->tid, j->i0, j->i1); return NULL; }

/* models.pyx:204-211: c = n // nthreads, chunks (0,c),(c,2c)..., last one extended to n */
static void run_chunked(int n, int nthreads, void (*fn)(void *, int, int, int), void *ctx)
{
    int c, nchunks, k;
    pthread_t *th; amo_job *jobs;
    if (nthreads < 1) nthreads = 1;
    if (n < nthreads) nthreads = n > 0 ? n : 1;      /* reference errors out (c == 0) */
    if (nthreads == 1 || n == 0) { fn(ctx, 0, 0, n); return; }
    c = n / nthreads;
    nchunks = n / c;                                  /* zip(range(0,n,c), range(c,n+1,c)) */
    th = (pthread_t *)malloc(sizeof(pthread_t) * nchunks);
    jobs = (amo_job *)malloc(sizeof(amo_job) * nchunks);
    for (k = 0; k < nchunks; k++) {
        jobs[k].fn = fn; jobs[k].ctx = ctx; jobs[k].tid = k;
        jobs[k].i0 = k * c; jobs[k].i1 = (k == nchunks - 1) ? n : (k + 1) * c;
        pthread_create(&th[k], NULL, job_tramp, &jobs[k]);
    }
    for (k = 0; k < nchunks; k++) pthread_join(th[k], NULL);
    free(jobs); free(th);
}

/* ------------------------------------------------------------------ NODDI models.pyx:816-991 */
typedef struct {
    const amo_noddi_args *a; const double *y, *dirs;
    double *est, *rmse, *nrmse, *mod, *x_dbg;
    int64_t err;               /* -(voxel+1) of first OOB direction */
    pthread_mutex_t mu;
} noddi_ctx;

static void noddi_chunk(void *vctx, int tid, int i0, int i1)
{
    noddi_ctx *cx = (noddi_ctx *)vctx;
    const amo_noddi_args *a = cx->a;
    const int nS = a->nS, n_wm = a->n_wm, dwi = a->dwi_count;
    const int n_atoms = n_wm + 1 + (a->is_exvivo ? 1 : 0);
    const int n_maps = 3 + (a->is_exvivo ? 1 : 0);
    const int single_b0 = (nS == 1 + dwi);                       /* models.pyx:820 */
    double *A = (double *)calloc((size_t)nS * n_atoms, sizeof(double));
    double *A2 = (double *)calloc((size_t)dwi * n_wm, sizeof(double));
    double *A3 = (double *)calloc((size_t)nS * n_atoms, sizeof(double));
    double *x = (double *)calloc((size_t)n_atoms * 2 + nS + dwi, sizeof(double));
    double *x3 = x + n_atoms, *yest = x3 + n_atoms, *y2 = yest + nS;
    int *posidx = (int *)malloc(sizeof(int) * n_atoms);
    int i, j, k, last_lut = -1;
    (void)tid;
    for (i = i0; i < i1; i++) {
        const double *yi = cx->y + (size_t)i * nS;
        double rn, f1 = 0, f2 = 0, k1 = 0, sum_atoms = 0, sum_wm = 0, ndi, odi, fwf;
        int npos = 0;
        int lut = amo_dir_to_lut_idx(cx->dirs + (size_t)i * 3, a->htable, NULL, NULL);
        if (lut < 0 || lut >= a->ndirs) {
            pthread_mutex_lock(&cx->mu);
            if (cx->err == 0 || -(int64_t)(i + 1) > cx->err) cx->err = -(int64_t)(i + 1);
            pthread_mutex_unlock(&cx->mu);
            continue;
        }
        /* prepare dictionary (models.pyx:905-908).  The reference copies the LUT slice for every voxel; the copy is
         * skipped when the previous voxel of this thread had the same orientation (A is never modified by the solvers):
         * identical results, and callers that present the voxels sorted by LUT index pay one copy per orientation
         * ("optimised" CPU baseline of bench.py; unsorted input = the reference's cost) */
        if (lut != last_lut) {
        for (k = 0; k < n_wm; k++) {
            const float *src = a->wm + ((size_t)k * a->ndirs + lut) * nS;
            for (j = 0; j < nS; j++) A[(size_t)k * nS + j] = (double)src[j];
        }
        if (a->is_exvivo) for (j = 0; j < nS; j++) A[(size_t)(n_atoms - 2) * nS + j] = 1.0;
        for (j = 0; j < nS; j++) A[(size_t)(n_atoms - 1) * nS + j] = (double)a->iso[j];
        last_lut = lut;
        }
        /* fit_1 (CSF), models.pyx:911 */
        amo_nnls(A, yi, nS, n_atoms, x, &rn);
        if (cx->x_dbg) memcpy(cx->x_dbg + ((size_t)i * 3 + 0) * n_atoms, x, sizeof(double) * n_atoms);
        /* fit_2 (IC + EC), models.pyx:914-926 */
        for (j = 0; j < dwi; j++) {
            int row = single_b0 ? j + 1 : (int)a->dwi_idx[j];
            for (k = 0; k < n_wm; k++)
                A2[(size_t)k * dwi + j] = A[(size_t)k * nS + row] * a->norms[(size_t)j * n_wm + k];
            y2[j] = yi[row] - x[n_atoms - 1] * (double)a->iso[row];
            if (a->is_exvivo) y2[j] -= x[n_atoms - 2] * 1.0;
            if (y2[j] < 0.0) y2[j] = 0.0;
        }
        amo_lasso(A2, y2, dwi, n_wm, x, a->lambda1, a->lambda2);
        if (cx->x_dbg) memcpy(cx->x_dbg + ((size_t)i * 3 + 1) * n_atoms, x, sizeof(double) * n_atoms);
        /* fit_3 (debias), models.pyx:929-942 */
        if (a->is_exvivo) x[n_atoms - 2] = 1.0;
        x[n_atoms - 1] = 1.0;
        for (j = 0; j < n_atoms; j++) if (x[j] > 0.0) posidx[npos++] = j;
        for (k = 0; k < npos; k++)
            memcpy(A3 + (size_t)k * nS, A + (size_t)posidx[k] * nS, sizeof(double) * nS);
        amo_nnls(A3, yi, nS, npos, x3, &rn);
        for (k = 0; k < npos; k++) x[posidx[k]] = x3[k];
        if (cx->x_dbg) memcpy(cx->x_dbg + ((size_t)i * 3 + 2) * n_atoms, x, sizeof(double) * n_atoms);
        /* estimates, models.pyx:945-967 */
        for (j = 0; j < n_atoms; j++) sum_atoms += x[j];
        sum_atoms += 1e-16;
        for (j = 0; j < n_wm; j++) sum_wm += x[j] / sum_atoms;
        sum_wm += 1e-16;
        for (j = 0; j < n_wm; j++) {
            f1 += (double)a->icvf[j] * x[j] / sum_atoms / sum_wm;
            f2 += (double)((float)(1.0 - (double)a->icvf[j])) * x[j] / sum_atoms / sum_wm;
            k1 += (double)a->kappa[j] * x[j] / sum_atoms / sum_wm;
        }
        ndi = f1 / (f1 + f2 + 1e-16);
        odi = 2.0 / M_PI * atan2(1.0, k1);
        fwf = x[n_atoms - 1] / sum_atoms;
        cx->est[(size_t)i * n_maps + 0] = ndi;
        cx->est[(size_t)i * n_maps + 1] = odi;
        cx->est[(size_t)i * n_maps + 2] = fwf;
        if (a->is_exvivo) cx->est[(size_t)i * n_maps + 3] = x[n_atoms - 2] / sum_atoms;
        if (a->compute_rmse && cx->rmse) cx->rmse[i] = rmse_of(A, nS, n_atoms, yi, x, yest);
        if (a->compute_nrmse && cx->nrmse) cx->nrmse[i] = nrmse_of(A, nS, n_atoms, yi, x, yest);
        if (a->compute_mod && cx->mod) {
            double tf = 1.0 - fwf;
            cx->mod[(size_t)i * 2 + 0] = ndi * tf;
            cx->mod[(size_t)i * 2 + 1] = odi * tf;
        }
    }
    free(posidx); free(x); free(A3); free(A2); free(A);
}

int64_t amo_noddi_fit(const amo_noddi_args *a, const double *y, const double *dirs,
                      double *estimates, double *rmse, double *nrmse, double *mod, double *x_dbg)
{
    noddi_ctx cx;
    cx.a = a; cx.y = y; cx.dirs = dirs; cx.est = estimates; cx.rmse = rmse; cx.nrmse = nrmse;
    cx.mod = mod; cx.x_dbg = x_dbg; cx.err = 0;
    pthread_mutex_init(&cx.mu, NULL);
    run_chunked(a->n_vox, a->nthreads, noddi_chunk, &cx);
    pthread_mutex_destroy(&cx.mu);
    return cx.err;
}

/* ------------------------------------------------------------------ FreeWater models.pyx:1168-1286 */
typedef struct {
    const amo_fw_args *a; const double *y, *dirs;
    double *est, *rmse, *nrmse, *ycorr, *x_dbg; int64_t err; pthread_mutex_t mu;
} fw_ctx;

static void fw_chunk(void *vctx, int tid, int i0, int i1)
{
    fw_ctx *cx = (fw_ctx *)vctx;
    const amo_fw_args *a = cx->a;
    const int nS = a->nS, n_perp = a->n_perp, n_iso = a->n_iso, n_atoms = n_perp + n_iso;
    const int n_maps = a->is_mouse ? 4 : 2;
    double *A = (double *)calloc((size_t)nS * n_atoms, sizeof(double));
    double *x = (double *)calloc((size_t)n_atoms + nS, sizeof(double));
    double *yest = x + n_atoms;
    int i, j, k;
    (void)tid;
    for (k = 0; k < n_iso; k++)                                   /* models.pyx:1235 */
        for (j = 0; j < nS; j++) A[(size_t)(n_perp + k) * nS + j] = (double)a->CSF[(size_t)k * nS + j];
    for (i = i0; i < i1; i++) {
        const double *yi = cx->y + (size_t)i * nS;
        double x_sum = 0.0, x_perp = 0.0, v;
        int lut = amo_dir_to_lut_idx(cx->dirs + (size_t)i * 3, a->htable, NULL, NULL);
        if (lut < 0 || lut >= a->ndirs) {
            pthread_mutex_lock(&cx->mu);
            if (cx->err == 0 || -(int64_t)(i + 1) > cx->err) cx->err = -(int64_t)(i + 1);
            pthread_mutex_unlock(&cx->mu);
            continue;
        }
        for (k = 0; k < n_perp; k++) {
            const float *src = a->D + ((size_t)k * a->ndirs + lut) * nS;
            for (j = 0; j < nS; j++) A[(size_t)k * nS + j] = (double)src[j];
        }
        amo_lasso(A, yi, nS, n_atoms, x, a->lambda1, a->lambda2);   /* models.pyx:1238 */
        if (cx->x_dbg) memcpy(cx->x_dbg + (size_t)i * n_atoms, x, sizeof(double) * n_atoms);
        for (j = 0; j < n_atoms; j++) { x_sum += x[j]; if (j < n_perp) x_perp += x[j]; }
        x_sum += 1e-16;
        v = x_perp / x_sum;
        cx->est[(size_t)i * n_maps + 0] = v;
        cx->est[(size_t)i * n_maps + 1] = 1.0 - v;
        if (a->is_mouse) {
            cx->est[(size_t)i * n_maps + 2] = x[n_perp] / x_sum;
            cx->est[(size_t)i * n_maps + 3] = x[n_perp + 1] / x_sum;
        }
        if (a->compute_rmse && cx->rmse) cx->rmse[i] = rmse_of(A, nS, n_atoms, yi, x, yest);
        if (a->compute_nrmse && cx->nrmse) cx->nrmse[i] = nrmse_of(A, nS, n_atoms, yi, x, yest);
        if (a->save_corrected && cx->ycorr) {                        /* models.pyx:1264-1274 */
            for (j = 0; j < n_atoms - n_iso; j++) x[j] = 0.0;
            for (j = 0; j < nS; j++) {
                double fw = 0.0, yc;
                for (k = 0; k < n_atoms; k++) fw += A[(size_t)k * nS + j] * x[k];
                yc = yi[j] - fw;
                cx->ycorr[(size_t)i * nS + j] = yc < 0.0 ? 0.0 : yc;
            }
        }
    }
    free(x); free(A);
}

int64_t amo_freewater_fit(const amo_fw_args *a, const double *y, const double *dirs,
                          double *estimates, double *rmse, double *nrmse, double *y_corr,
                          double *x_dbg)
{
    fw_ctx cx;
    cx.a = a; cx.y = y; cx.dirs = dirs; cx.est = estimates; cx.rmse = rmse; cx.nrmse = nrmse;
    cx.ycorr = y_corr; cx.x_dbg = x_dbg; cx.err = 0;
    pthread_mutex_init(&cx.mu, NULL);
    run_chunked(a->n_vox, a->nthreads, fw_chunk, &cx);
    pthread_mutex_destroy(&cx.mu);
    return cx.err;
}

/* ------------------------------------------------------------------ SANDI models.pyx:1509-1627 */
typedef struct {
    const amo_sandi_args *a; const double *y; double *est, *rmse, *nrmse, *x_dbg;
} sandi_ctx;

static void sandi_chunk(void *vctx, int tid, int i0, int i1)
{
    sandi_ctx *cx = (sandi_ctx *)vctx;
    const amo_sandi_args *a = cx->a;
    const int nS = a->nS, n_rs = a->n_rs, n_in = a->n_in, n_iso = a->n_iso;
    const int n_atoms = n_rs + n_in + n_iso;
    double *x = (double *)calloc((size_t)n_atoms + nS, sizeof(double));
    double *yest = x + n_atoms;
    int i, j;
    (void)tid;
    for (i = i0; i < i1; i++) {
        const double *yi = cx->y + (size_t)i * nS;
        double x_sum = 0, xsph = 0, xstk = 0, xiso = 0, Rsoma = 0, Din = 0, De = 0;
        amo_lasso(a->signal, yi, nS, n_atoms, x, a->lambda1, a->lambda2);   /* :1569 */
        for (j = 0; j < n_atoms; j++) x[j] *= a->norms[j];                  /* :1570-1571 */
        if (cx->x_dbg) memcpy(cx->x_dbg + (size_t)i * n_atoms, x, sizeof(double) * n_atoms);
        for (j = 0; j < n_atoms; j++) {
            x_sum += x[j];
            if (j < n_rs) xsph += x[j];
            if (j >= n_rs && j < n_rs + n_in) xstk += x[j];
            if (j >= n_rs + n_in) xiso += x[j];
        }
        x_sum += 1e-16;
        cx->est[(size_t)i * 6 + 0] = xsph / x_sum;
        cx->est[(size_t)i * 6 + 1] = xstk / x_sum;
        cx->est[(size_t)i * 6 + 2] = xiso / x_sum;
        for (j = 0; j < n_atoms; j++) {
            if (j < n_rs) Rsoma += a->Rs[j] * x[j];
            if (j >= n_rs && j < n_rs + n_in) Din += a->d_in[j - n_rs] * x[j];
            if (j >= n_rs + n_in) De += a->d_isos[j - (n_rs + n_in)] * x[j];
        }
        xsph += 1e-16; xstk += 1e-16; xiso += 1e-16;
        cx->est[(size_t)i * 6 + 3] = 1e6 * Rsoma / xsph;
        cx->est[(size_t)i * 6 + 4] = 1e3 * Din / xstk;
        cx->est[(size_t)i * 6 + 5] = 1e3 * De / xiso;
        /* quirk kept: errors use the RESCALED x against the NORMALISED A (:1571 then :1615) */
        if (a->compute_rmse && cx->rmse) cx->rmse[i] = rmse_of(a->signal, nS, n_atoms, yi, x, yest);
        if (a->compute_nrmse && cx->nrmse) cx->nrmse[i] = nrmse_of(a->signal, nS, n_atoms, yi, x, yest);
    }
    free(x);
}

int64_t amo_sandi_fit(const amo_sandi_args *a, const double *y,
                      double *estimates, double *rmse, double *nrmse, double *x_dbg)
{
    sandi_ctx cx;
    cx.a = a; cx.y = y; cx.est = estimates; cx.rmse = rmse; cx.nrmse = nrmse; cx.x_dbg = x_dbg;
    run_chunked(a->n_vox, a->nthreads, sandi_chunk, &cx);
    return 0;
}

/* ------------------------------------------------------------------ CylinderZeppelinBall models.pyx:526-652 */
typedef struct {
    const amo_czb_args *a; const double *y, *dirs;
    double *est, *rmse, *nrmse, *x_dbg; int64_t err; pthread_mutex_t mu;
} czb_ctx;

static void czb_chunk(void *vctx, int tid, int i0, int i1)
{
    czb_ctx *cx = (czb_ctx *)vctx;
    const amo_czb_args *a = cx->a;
    const int nS = a->nS, n_rs = a->n_rs, n_perp = a->n_perp, n_iso = a->n_iso;
    const int n_atoms = n_rs + n_perp + n_iso;
    double *A = (double *)calloc((size_t)nS * n_atoms, sizeof(double));
    double *x = (double *)calloc((size_t)n_atoms + nS, sizeof(double));
    double *yest = x + n_atoms;
    int i, j, k;
    (void)tid;
    for (k = 0; k < n_iso; k++)                                   /* models.pyx:610 (the same for every voxel) */
        for (j = 0; j < nS; j++) A[(size_t)(n_rs + n_perp + k) * nS + j] = (double)a->iso[(size_t)k * nS + j];
    for (i = i0; i < i1; i++) {
        const double *yi = cx->y + (size_t)i * nS;
        double f1 = 0.0, f2 = 0.0, v, am = 0.0, d;
        int lut = amo_dir_to_lut_idx(cx->dirs + (size_t)i * 3, a->htable, NULL, NULL);
        if (lut < 0 || lut >= a->ndirs) {
            pthread_mutex_lock(&cx->mu);
            if (cx->err == 0 || -(int64_t)(i + 1) > cx->err) cx->err = -(int64_t)(i + 1);
            pthread_mutex_unlock(&cx->mu);
            continue;
        }
        for (k = 0; k < n_rs; k++) {                               /* models.pyx:608 */
            const float *src = a->wmr + ((size_t)k * a->ndirs + lut) * nS;
            for (j = 0; j < nS; j++) A[(size_t)k * nS + j] = (double)src[j];
        }
        for (k = 0; k < n_perp; k++) {                             /* models.pyx:609 */
            const float *src = a->wmh + ((size_t)k * a->ndirs + lut) * nS;
            for (j = 0; j < nS; j++) A[(size_t)(n_rs + k) * nS + j] = (double)src[j];
        }
        amo_lasso(A, yi, nS, n_atoms, x, a->lambda1, a->lambda2);  /* models.pyx:613 */
        if (cx->x_dbg) memcpy(cx->x_dbg + (size_t)i * n_atoms, x, sizeof(double) * n_atoms);
        /* estimates, models.pyx:616-633 */
        for (j = 0; j < n_rs + n_perp; j++) {
            if (j < n_rs) f1 += x[j];
            if (j >= n_rs && j < n_rs + n_perp) f2 += x[j];
        }
        f2 += 1e-16;
        v = f1 / (f1 + f2 + 1e-16);
        f1 += 1e-16;
        for (j = 0; j < n_rs; j++) am += a->Rs[j] * x[j];
        am = 1e6 * 2.0 * am / f1;
        d = (4.0 * v) / (M_PI * pow(am, 2.0) + 1e-16);
        cx->est[(size_t)i * 3 + 0] = v;
        cx->est[(size_t)i * 3 + 1] = am;
        cx->est[(size_t)i * 3 + 2] = d;
        if (a->compute_rmse && cx->rmse) cx->rmse[i] = rmse_of(A, nS, n_atoms, yi, x, yest);
        if (a->compute_nrmse && cx->nrmse) cx->nrmse[i] = nrmse_of(A, nS, n_atoms, yi, x, yest);
    }
    free(x); free(A);
}

int64_t amo_czb_fit(const amo_czb_args *a, const double *y, const double *dirs,
                    double *estimates, double *rmse, double *nrmse, double *x_dbg)
{
    czb_ctx cx;
    cx.a = a; cx.y = y; cx.dirs = dirs; cx.est = estimates; cx.rmse = rmse; cx.nrmse = nrmse;
    cx.x_dbg = x_dbg; cx.err = 0;
    pthread_mutex_init(&cx.mu, NULL);
    run_chunked(a->n_vox, a->nthreads, czb_chunk, &cx);
    pthread_mutex_destroy(&cx.mu);
    return cx.err;
}
