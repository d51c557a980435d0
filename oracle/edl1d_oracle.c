/* CPU ORACLE, 1D EDL path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, tools/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; gmpnp_amd never does.
 *
 * A second, independent restatement of the reference's 1D hot path (plain C, no tables shared with gmpnp_amd or with
 * oracle/gmpnp_oracle.py): it evaluates the published integrands of reference 1D/MPNP_CO2ER_EDL.py LITERALLY at Gauss points,
 * the way the form compiler's generated code does, instead of through closed-form element integrals:
 *
 *   R_H ... R_CO2            1D:383-410         edl_point()
 *   F_p                      1D:412-427         edl_point()
 *   F_H ... F_cat  (PNP)     1D:429-455         edl_point(), steric = 0
 *   F_H ... F_cat  (MPNP)    1D:457-593         edl_point(), steric = 1
 *   J_CO2 v ds, J_OH v ds + J_H v ds   1D:371-375,553,738   point terms at the x = 0 vertex
 *   J = derivative(F, u)     [3P] inside solve(F == 0, u, bcs) 1D:737-742: analytic derivative of the integrand at each point
 *   quadrature               [3P] UFL degree estimation: 3 for F, 4 for J -> Gauss-Legendre (deg+2)/2 = 2 / 3 points
 *   DirichletBC.apply        1D:350-355; [3P] b[dof] = x[dof] - g, identity rows
 *   Newton                   1D:357-364; [3P] dolfin NewtonSolver: "residual" criterion, checked before the first iteration
 *   linear solve             [3P] default LU (UMFPACK): here a banded LU with partial pivoting (half-bandwidth 13)
 *   time loop                1D:633-796  u_n.assign(u); zero initial guess at step 0 (1D:320)
 *   field projection         1D:802-805  consistent-mass L2 projection of -grad(p) onto P1
 *
 * PARITY STATUS: pinned DIRECTLY on the reference's only stored hot-path outputs, 1D/Stern_CO2ER.py:66-68 — this oracle, run over
 * the reference's 20,000-solve staged schedule, reproduces the recorded field_OHP / eps_rel_OHP digits
 * (tools/oracle_stern_schedule.py -> tests/golden/stern_oracle.json, asserted by tests/test_oracle_pins.py).
 *
 * Arithmetic type: `real` = double by default; -DEDL_REAL_LONG_DOUBLE builds the same code in x87 extended precision (64-bit
 * mantissa) with the state held in that precision across steps — the instrument for separating round-off effects from
 * algorithmic ones.  The C-ABI speaks double either way.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef EDL_REAL_LONG_DOUBLE
typedef long double real;
#define RSQRT sqrtl
#define RFABS fabsl
#else
typedef double real;
#define RSQRT sqrt
#define RFABS fabs
#endif

#define NS 6 /* H, OH, HCO3, CO32, CO2, cat   (mixed-space order, reference 1D:310-317) */
#define NF 7 /* + potential */
#define KL 13
#define KU 13
#define LDAB (2 * KL + KU + 1)

enum { iH = 0, iOH = 1, iHCO3 = 2, iCO32 = 3, iCO2 = 4, iCAT = 5, iP = 6 };

typedef struct {
    int nv;          /* vertices, ascending x */
    int steric;      /* 1 = MPNP (1D:457-593), 0 = PNP (1D:429-455) */
    int nq_f, nq_j;  /* Gauss points of the residual / Jacobian cell integrals */
    int max_it;      /* newton_solver.maximum_iterations */
    double rtol, atol, relax;
    double conc[NS];      /* initial_conc            1D:154-162 */
    double z[NS];         /* z                       1D:158 */
    double scale_R[NS];   /* scale_R (cat unused)    1D:186-190 */
    double scale_vol[NS]; /* scale_vol               1D:196-200 */
    double kw1, kw2, ka1, ka2, kb1, kb2; /* 1D:96-101 */
    double eps_rel, n_water_cat, n_water_H; /* 1D:106-115,139 */
    double q;             /* 1D:193 */
    double del_t, L_D;    /* (u - u_n)/(del_t*L_D)   1D:176,266 */
    double J_CO2, J_OH, J_H; /* 1D:371-375 */
    double voltage;       /* bc2 1D:354 */
} edl_params_t;

static void gauss01(int n, real *xi, real *w)
{ /* Gauss-Legendre on [0,1], weights sum to 1 (FIAT's default interval scheme) */
    if (n == 1) { xi[0] = 0.5L; w[0] = 1.0L; }
    else if (n == 2) {
        real a = RSQRT((real)1.0L / 3.0L);
        xi[0] = 0.5L * (1.0L - a); xi[1] = 0.5L * (1.0L + a); w[0] = w[1] = 0.5L;
    } else if (n == 3) {
        real a = RSQRT((real)3.0L / 5.0L);
        xi[0] = 0.5L * (1.0L - a); xi[1] = 0.5L; xi[2] = 0.5L * (1.0L + a);
        w[0] = w[2] = (real)5.0L / 18.0L; w[1] = (real)8.0L / 18.0L;
    } else if (n == 4) {
        real a = RSQRT((real)3.0L / 7.0L - (real)2.0L / 7.0L * RSQRT((real)6.0L / 5.0L));
        real b = RSQRT((real)3.0L / 7.0L + (real)2.0L / 7.0L * RSQRT((real)6.0L / 5.0L));
        real wa = ((real)18.0L + RSQRT((real)30.0L)) / 72.0L, wb = ((real)18.0L - RSQRT((real)30.0L)) / 72.0L;
        xi[0] = 0.5L * (1.0L - b); xi[1] = 0.5L * (1.0L - a); xi[2] = 0.5L * (1.0L + a); xi[3] = 0.5L * (1.0L + b);
        w[0] = w[3] = wb; w[1] = w[2] = wa;
    } else { /* n == 5 */
        real s = RSQRT((real)10.0L / 7.0L);
        real a = RSQRT((real)5.0L - 2.0L * s) / 3.0L, b = RSQRT((real)5.0L + 2.0L * s) / 3.0L;
        real wa = ((real)322.0L + 13.0L * RSQRT((real)70.0L)) / 1800.0L, wb = ((real)322.0L - 13.0L * RSQRT((real)70.0L)) / 1800.0L;
        xi[0] = 0.5L * (1.0L - b); xi[1] = 0.5L * (1.0L - a); xi[2] = 0.5L; xi[3] = 0.5L * (1.0L + a); xi[4] = 0.5L * (1.0L + b);
        w[0] = w[4] = wb; w[1] = w[3] = wa; w[2] = (real)128.0L / 450.0L;
    }
}

/* Integrand at one point.  Inputs: u[NF] values, du[NF] x-derivatives, un[NS] previous-step values.
 * Outputs: f0[NF] = coefficient of v (test function value), f1[NF] = coefficient of dv/dx, so that the integrand of row i is
 * f0[i]*v + f1[i]*dv.  With want_j: their derivatives  d f0[i]/d u[j] (A0), d f0[i]/d du[j] (B0), d f1[i]/d u[j] (A1),
 * d f1[i]/d du[j] (B1). */
static void edl_point(const edl_params_t *P, const real *u, const real *du, const real *un, real *f0, real *f1, int want_j,
                      real A0[NF][NF], real B0[NF][NF], real A1[NF][NF], real B1[NF][NF])
{
    const real cH = u[iH] * (real)P->conc[iH], cOH = u[iOH] * (real)P->conc[iOH], cHCO3 = u[iHCO3] * (real)P->conc[iHCO3];
    const real cCO32 = u[iCO32] * (real)P->conc[iCO32], cCO2 = u[iCO2] * (real)P->conc[iCO2];
    const real kw1 = P->kw1, kw2 = P->kw2, ka1 = P->ka1, ka2 = P->ka2, kb1 = P->kb1, kb2 = P->kb2;
    real R[NS];
    /* reference 1D:383-410 */
    R[iH] = -(real)P->scale_R[iH] * (kw2 * cH * cOH - kw1);
    R[iOH] = -(real)P->scale_R[iOH] * (kw2 * cH * cOH + ka1 * cOH * cHCO3 + kb1 * cCO2 * cOH - kw1 - ka2 * cCO32 - kb2 * cHCO3);
    R[iHCO3] = -(real)P->scale_R[iHCO3] * (ka1 * cOH * cHCO3 + kb2 * cHCO3 - ka2 * cCO32 - kb1 * cCO2 * cOH);
    R[iCO32] = -(real)P->scale_R[iCO32] * (ka2 * cCO32 - ka1 * (cOH * cHCO3));
    R[iCO2] = -(real)P->scale_R[iCO2] * (kb1 * cCO2 * cOH - kb2 * cHCO3);
    R[iCAT] = 0;
    const real inv_dt = (real)1.0L / ((real)P->del_t * (real)P->L_D);
    real S = 0, G = 0;
    for (int j = 0; j < NS; j++) { S += (real)P->scale_vol[j] * u[j]; G += (real)P->scale_vol[j] * du[j]; }
    const real beta = P->steric ? (real)1.0L / ((real)1.0L - S) : 0;
    for (int i = 0; i < NS; i++) {
        f0[i] = (u[i] - un[i]) * inv_dt - R[i];                       /* time term, - R_i v */
        f1[i] = du[i] + (real)P->z[i] * u[i] * du[iP];                /* grad u.grad v + z u grad p.grad v */
        if (P->steric) f1[i] += u[i] * beta * G;                      /* [u_i/(1-S)] G.grad v */
    }
    /* F_p, reference 1D:412-427 */
    const real w = ((real)P->n_water_cat * u[iCAT] * (real)P->conc[iCAT] + (real)P->n_water_H * u[iH] * (real)P->conc[iH]) * (real)1.0e-3L;
    const real eps = (real)P->eps_rel * (((real)55.0L - w) / (real)55.0L) + (real)6.0L * (w / (real)55.0L);
    real rho = 0;
    for (int j = 0; j < NS; j++) rho += (real)P->z[j] * u[j] * (real)P->conc[j];
    f0[iP] = rho * (real)P->q;
    f1[iP] = -eps * du[iP];
    if (!want_j) return;
    memset(A0, 0, sizeof(real) * NF * NF); memset(B0, 0, sizeof(real) * NF * NF);
    memset(A1, 0, sizeof(real) * NF * NF); memset(B1, 0, sizeof(real) * NF * NF);
    /* dR/du: chain rule through c_X = u_X conc_X */
    real dR[NS][NS];
    memset(dR, 0, sizeof dR);
    const real bH = P->conc[iH], bOH = P->conc[iOH], bHCO3 = P->conc[iHCO3], bCO32 = P->conc[iCO32], bCO2 = P->conc[iCO2];
    dR[iH][iH] = -(real)P->scale_R[iH] * (kw2 * bH * cOH);
    dR[iH][iOH] = -(real)P->scale_R[iH] * (kw2 * cH * bOH);
    dR[iOH][iH] = -(real)P->scale_R[iOH] * (kw2 * bH * cOH);
    dR[iOH][iOH] = -(real)P->scale_R[iOH] * (kw2 * cH * bOH + ka1 * bOH * cHCO3 + kb1 * cCO2 * bOH);
    dR[iOH][iHCO3] = -(real)P->scale_R[iOH] * (ka1 * cOH * bHCO3 - kb2 * bHCO3);
    dR[iOH][iCO32] = -(real)P->scale_R[iOH] * (-ka2 * bCO32);
    dR[iOH][iCO2] = -(real)P->scale_R[iOH] * (kb1 * bCO2 * cOH);
    dR[iHCO3][iOH] = -(real)P->scale_R[iHCO3] * (ka1 * bOH * cHCO3 - kb1 * cCO2 * bOH);
    dR[iHCO3][iHCO3] = -(real)P->scale_R[iHCO3] * (ka1 * cOH * bHCO3 + kb2 * bHCO3);
    dR[iHCO3][iCO32] = -(real)P->scale_R[iHCO3] * (-ka2 * bCO32);
    dR[iHCO3][iCO2] = -(real)P->scale_R[iHCO3] * (-kb1 * bCO2 * cOH);
    dR[iCO32][iCO32] = -(real)P->scale_R[iCO32] * (ka2 * bCO32);
    dR[iCO32][iOH] = -(real)P->scale_R[iCO32] * (-ka1 * (bOH * cHCO3));
    dR[iCO32][iHCO3] = -(real)P->scale_R[iCO32] * (-ka1 * (cOH * bHCO3));
    dR[iCO2][iCO2] = -(real)P->scale_R[iCO2] * (kb1 * bCO2 * cOH);
    dR[iCO2][iOH] = -(real)P->scale_R[iCO2] * (kb1 * cCO2 * bOH);
    dR[iCO2][iHCO3] = -(real)P->scale_R[iCO2] * (-kb2 * bHCO3);
    for (int i = 0; i < NS; i++) {
        A0[i][i] += inv_dt;
        for (int j = 0; j < NS; j++) A0[i][j] -= dR[i][j];
        B1[i][i] += 1;
        A1[i][i] += (real)P->z[i] * du[iP];
        B1[i][iP] += (real)P->z[i] * u[i];
        if (P->steric) {
            A1[i][i] += beta * G;
            for (int j = 0; j < NS; j++) {
                A1[i][j] += u[i] * beta * beta * (real)P->scale_vol[j] * G;
                B1[i][j] += u[i] * beta * (real)P->scale_vol[j];
            }
        }
    }
    for (int j = 0; j < NS; j++) A0[iP][j] = (real)P->z[j] * (real)P->conc[j] * (real)P->q;
    const real deps_dw = ((real)6.0L - (real)P->eps_rel) / (real)55.0L;
    A1[iP][iCAT] = -deps_dw * (real)P->n_water_cat * (real)P->conc[iCAT] * (real)1.0e-3L * du[iP];
    A1[iP][iH] = -deps_dw * (real)P->n_water_H * (real)P->conc[iH] * (real)1.0e-3L * du[iP];
    B1[iP][iP] = -eps;
}

/* Assembled residual b (with DirichletBC.apply(b, x)) and, if ab != NULL, Jacobian in LAPACK band storage
 * ab[j*LDAB + KL + KU + i - j] = A(i,j) with identity rows at the Dirichlet dofs. */
static void edl_assemble(const edl_params_t *P, const double *x, const real *u, const real *un, real *b, real *ab)
{
    const int nv = P->nv, n = nv * NF;
    real xf[8], wf[8], xj[8], wj[8];
    gauss01(P->nq_f, xf, wf);
    gauss01(P->nq_j, xj, wj);
    for (int i = 0; i < n; i++) b[i] = 0;
    if (ab) memset(ab, 0, sizeof(real) * (size_t)n * LDAB);
    real A0[NF][NF], B0[NF][NF], A1[NF][NF], B1[NF][NF], f0[NF], f1[NF];
    for (int e = 0; e + 1 < nv; e++) {
        const real h = (real)x[e + 1] - (real)x[e];
        const real dphi[2] = {-(real)1.0L / h, (real)1.0L / h};
        const real *ue[2] = {u + (size_t)e * NF, u + (size_t)(e + 1) * NF};
        const real *ne[2] = {un + (size_t)e * NF, un + (size_t)(e + 1) * NF};
        real dub[NF];
        for (int f = 0; f < NF; f++) dub[f] = ue[0][f] * dphi[0] + ue[1][f] * dphi[1];
        for (int qd = 0; qd < P->nq_f; qd++) {
            const real phi[2] = {(real)1.0L - xf[qd], xf[qd]};
            real uq[NF], nq[NS];
            for (int f = 0; f < NF; f++) uq[f] = ue[0][f] * phi[0] + ue[1][f] * phi[1];
            for (int f = 0; f < NS; f++) nq[f] = ne[0][f] * phi[0] + ne[1][f] * phi[1];
            edl_point(P, uq, dub, nq, f0, f1, 0, A0, B0, A1, B1);
            const real wt = wf[qd] * h;
            for (int a = 0; a < 2; a++)
                for (int i = 0; i < NF; i++) b[(size_t)(e + a) * NF + i] += wt * (f0[i] * phi[a] + f1[i] * dphi[a]);
        }
        if (!ab) continue;
        for (int qd = 0; qd < P->nq_j; qd++) {
            const real phi[2] = {(real)1.0L - xj[qd], xj[qd]};
            real uq[NF], nq[NS];
            for (int f = 0; f < NF; f++) uq[f] = ue[0][f] * phi[0] + ue[1][f] * phi[1];
            for (int f = 0; f < NS; f++) nq[f] = ne[0][f] * phi[0] + ne[1][f] * phi[1];
            edl_point(P, uq, dub, nq, f0, f1, 1, A0, B0, A1, B1);
            const real wt = wj[qd] * h;
            for (int a = 0; a < 2; a++)
                for (int i = 0; i < NF; i++) {
                    const int row = (e + a) * NF + i;
                    for (int c = 0; c < 2; c++)
                        for (int j = 0; j < NF; j++) {
                            const int col = (e + c) * NF + j;
                            const real v = wt * ((A0[i][j] * phi[c] + B0[i][j] * dphi[c]) * phi[a] + (A1[i][j] * phi[c] + B1[i][j] * dphi[c]) * dphi[a]);
                            ab[(size_t)col * LDAB + KL + KU + row - col] += v;
                        }
                }
        }
    }
    /* exterior-facet (point) integrals: ds covers both end points (Q6); the x = 1 rows are overwritten by bc1 below */
    const int ends[2] = {0, nv - 1};
    for (int k = 0; k < 2; k++) {
        b[(size_t)ends[k] * NF + iCO2] += (real)P->J_CO2;
        b[(size_t)ends[k] * NF + iOH] += (real)P->J_OH;
        b[(size_t)ends[k] * NF + iH] += (real)P->J_H;
    }
    /* bcs = [bc1, bc2] (1D:350-355): all seven fields at x = 1 -> (1,1,1,1,1,1,0); p(0) = voltage */
    int bc_dof[NF + 1]; real bc_val[NF + 1];
    for (int f = 0; f < NF; f++) { bc_dof[f] = (nv - 1) * NF + f; bc_val[f] = f < NS ? 1 : 0; }
    bc_dof[NF] = iP; bc_val[NF] = (real)P->voltage;
    for (int k = 0; k <= NF; k++) {
        const int r = bc_dof[k];
        b[r] = u[r] - bc_val[k];
        if (ab) {
            int c0 = r - KL < 0 ? 0 : r - KL, c1 = r + KU >= n ? n - 1 : r + KU;
            for (int c = c0; c <= c1; c++) ab[(size_t)c * LDAB + KL + KU + r - c] = (c == r) ? 1 : 0;
        }
    }
}

/* Banded LU with partial pivoting (the unblocked algorithm of LAPACK's dgbtf2/dgbtrs), in place; returns 0 or the index
 * (1-based) of a zero pivot. */
static int band_solve(int n, real *ab, real *b, int *ipiv)
{
    const int kv = KU + KL;
    int ju = 0;
    for (int j = 0; j < n; j++) {
        const int km = (KL < n - 1 - j) ? KL : n - 1 - j;
        real *col = ab + (size_t)j * LDAB;
        int jp = 0; real best = RFABS(col[kv]);
        for (int i = 1; i <= km; i++) if (RFABS(col[kv + i]) > best) { best = RFABS(col[kv + i]); jp = i; }
        ipiv[j] = j + jp;
        if (best == 0) return j + 1;
        int t = j + KU + jp; if (t > n - 1) t = n - 1; if (t > ju) ju = t;
        if (jp != 0)
            for (int c = j; c <= ju; c++) {
                real *cc = ab + (size_t)c * LDAB;
                real tmp = cc[kv + jp + j - c]; cc[kv + jp + j - c] = cc[kv + j - c]; cc[kv + j - c] = tmp;
            }
        if (km > 0) {
            const real inv = (real)1.0L / col[kv];
            for (int i = 1; i <= km; i++) col[kv + i] *= inv;
            for (int c = j + 1; c <= ju; c++) {
                real *cc = ab + (size_t)c * LDAB;
                const real ujc = cc[kv + j - c];
                if (ujc != 0) for (int i = 1; i <= km; i++) cc[kv + j - c + i] -= col[kv + i] * ujc;
            }
        }
    }
    /* forward: L y = P b */
    for (int j = 0; j < n; j++) {
        const int km = (KL < n - 1 - j) ? KL : n - 1 - j;
        const int p = ipiv[j];
        if (p != j) { real t = b[p]; b[p] = b[j]; b[j] = t; }
        const real *col = ab + (size_t)j * LDAB;
        const real bj = b[j];
        for (int i = 1; i <= km; i++) b[j + i] -= col[kv + i] * bj;
    }
    /* backward: U x = y */
    for (int j = n - 1; j >= 0; j--) {
        const real *col = ab + (size_t)j * LDAB;
        b[j] /= col[kv];
        const real bj = b[j];
        const int i0 = j - kv < 0 ? 0 : j - kv;
        for (int i = i0; i < j; i++) b[i] -= col[kv + i - j] * bj;
    }
    return 0;
}

static real norm2(const real *v, int n)
{
    real s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return RSQRT(s);
}

typedef struct { int n; real *u, *un, *b, *ab; int *ipiv; } work_t;

/* [3P] dolfin::NewtonSolver::solve, criterion "residual".  returns iterations, or -(iterations) - 1 if not converged,
 * or -1000 on a singular matrix.  res[0..its] receives the l2 residuals. */
static int newton(const edl_params_t *P, const double *x, work_t *W, double *res)
{
    const int n = W->n;
    edl_assemble(P, x, W->u, W->un, W->b, NULL);
    real r = norm2(W->b, n), r0 = r;
    if (res) res[0] = (double)r;
    int it = 0;
    int conv = (r / r0 < (real)P->rtol) || (r < (real)P->atol);
    while (!conv && it < P->max_it) {
        edl_assemble(P, x, W->u, W->un, W->b, W->ab);
        if (band_solve(n, W->ab, W->b, W->ipiv)) return -1000;
        for (int i = 0; i < n; i++) W->u[i] -= (real)P->relax * W->b[i];
        it++;
        edl_assemble(P, x, W->u, W->un, W->b, NULL);
        r = norm2(W->b, n);
        if (res) res[it] = (double)r;
        if (!(r == r)) return -it - 1;
        conv = (r / r0 < (real)P->rtol) || (r < (real)P->atol);
    }
    return conv ? it : -it - 1;
}

static work_t *work_new(int nv)
{
    work_t *W = calloc(1, sizeof *W);
    W->n = nv * NF;
    W->u = calloc(W->n, sizeof(real)); W->un = calloc(W->n, sizeof(real)); W->b = calloc(W->n, sizeof(real));
    W->ab = calloc((size_t)W->n * LDAB, sizeof(real)); W->ipiv = calloc(W->n, sizeof(int));
    return W;
}
static void work_free(work_t *W) { free(W->u); free(W->un); free(W->b); free(W->ab); free(W->ipiv); free(W); }

/* ---- C-ABI (ctypes) ------------------------------------------------------------------------------------------- */

int edl_real_bits(void) { return (int)(sizeof(real) * 8); }
int edl_real_mantissa(void)
{
#ifdef EDL_REAL_LONG_DOUBLE
    return __LDBL_MANT_DIG__;
#else
    return __DBL_MANT_DIG__;
#endif
}

/* residual (length nv*7) and, if jac != NULL, the dense band rows jac[row*27 + 13 + col - row] of the Jacobian at (u, un) */
void edl_residual_jacobian(const edl_params_t *P, const double *x, const double *u, const double *un, double *F, double *jac)
{
    work_t *W = work_new(P->nv);
    for (int i = 0; i < W->n; i++) { W->u[i] = u[i]; W->un[i] = un[i]; }
    edl_assemble(P, x, W->u, W->un, W->b, jac ? W->ab : NULL);
    for (int i = 0; i < W->n; i++) F[i] = (double)W->b[i];
    if (jac) {
        memset(jac, 0, sizeof(double) * (size_t)W->n * (KL + KU + 1));
        for (int r = 0; r < W->n; r++)
            for (int c = (r - KL < 0 ? 0 : r - KL); c <= (r + KU >= W->n ? W->n - 1 : r + KU); c++)
                jac[(size_t)r * (KL + KU + 1) + KL + c - r] = (double)W->ab[(size_t)c * LDAB + KL + KU + r - c];
    }
    work_free(W);
}

/* one Newton solve from (u, un); u is updated in place; returns as newton() */
int edl_newton_solve(const edl_params_t *P, const double *x, double *u, const double *un, double *res)
{
    work_t *W = work_new(P->nv);
    for (int i = 0; i < W->n; i++) { W->u[i] = u[i]; W->un[i] = un[i]; }
    int rc = newton(P, x, W, res);
    for (int i = 0; i < W->n; i++) u[i] = (double)W->u[i];
    work_free(W);
    return rc;
}

/* The time loop of reference 1D:633-796 without the H_OHP controller (H_OHP = None): nsteps solves.  On entry u / un hold the
 * state to start from (step 0 of a run: u = 0, un = (1,..,1,0)); on exit both hold the last solution (u_n.assign(u)).
 * its[k] = Newton iterations of solve k; the function stops at the first solve that does not converge and returns its index
 * (0-based), or nsteps if all converged.  res_last (>= max_it+1 doubles) receives the residual history of the last solve.
 * state_ext (may be NULL, n long doubles-as-two-doubles is not needed): the state is carried in `real` inside one call only. */
int edl_run(const edl_params_t *P, const double *x, int nsteps, double *u, double *un, int *its, double *res_last)
{
    work_t *W = work_new(P->nv);
    for (int i = 0; i < W->n; i++) { W->u[i] = u[i]; W->un[i] = un[i]; }
    int k;
    for (k = 0; k < nsteps; k++) {
        int rc = newton(P, x, W, res_last);
        its[k] = rc;
        if (rc < 0) break;
        memcpy(W->un, W->u, sizeof(real) * W->n);
    }
    for (int i = 0; i < W->n; i++) { u[i] = (double)W->u[i]; un[i] = (double)W->un[i]; }
    work_free(W);
    return k;
}

/* project(-grad(p), W) of reference 1D:802-805: consistent P1 mass matrix (tridiagonal), right-hand side int -p' phi_a */
void edl_project_neg_gradient(int nv, const double *x, const double *p, double *field)
{
    real *dl = calloc(nv, sizeof(real)), *d = calloc(nv, sizeof(real)), *du = calloc(nv, sizeof(real)), *r = calloc(nv, sizeof(real));
    for (int e = 0; e + 1 < nv; e++) {
        const real h = (real)x[e + 1] - (real)x[e];
        const real g = -((real)p[e + 1] - (real)p[e]) / h;
        d[e] += h / 3; d[e + 1] += h / 3; du[e] += h / 6; dl[e + 1] += h / 6;
        r[e] += g * h / 2; r[e + 1] += g * h / 2;
    }
    for (int i = 1; i < nv; i++) { real m = dl[i] / d[i - 1]; d[i] -= m * du[i - 1]; r[i] -= m * r[i - 1]; }
    r[nv - 1] /= d[nv - 1];
    for (int i = nv - 2; i >= 0; i--) r[i] = (r[i] - du[i] * r[i + 1]) / d[i];
    for (int i = 0; i < nv; i++) field[i] = (double)r[i];
    free(dl); free(d); free(du); free(r);
}
