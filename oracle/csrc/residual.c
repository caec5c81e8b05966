/* ORACLE -- test infrastructure only.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
 * never by the product path.
 *
 * Plain C restatement of the per-partition scalar residual sweep of ImmersedBoundary.jl, i.e. the closure of
 * /root/reference/test/advection.jl:67-83 composed from the operators of /root/reference/src/ImmersedBoundary.jl:
 *   at_faces :899-910, green_gauss :918-926, unsigned_green_gauss :934-942, cell_gradient :965-972,
 *   JST_sensor :1077-1097, minmod :1099, MUSCL :1113-1157 (D given, high_order = true),
 *   Accumulator call src/accumulator.jl:78-111 (face accumulators: weights 1/len, sequential reduce).
 * The Euler sweep (R2 of SURVEY.md 8d) is here too: ibo_residual_euler_faithful = JST_sensor(p) + per dim cell_gradient,
 * MUSCL(D, high_order), CFD.inviscid_fluxes (HLL, /root/reference/src/cfd.jl:459-508, Float64 result) and green_gauss,
 * one array pass per broadcast; bit-identical to the numpy composition of oracle/domain.py + oracle/cfd.py.
 * Two forms of the scalar sweep:
 *   ibo_residual_advection_faithful -- one array pass per reference broadcast (same pass structure and the same
 *       Float32 evaluation order as the Julia code; bit-identical to oracle/domain.py), loops split over OpenMP threads;
 *   ibo_residual_advection_fused    -- three passes (cells: gradients + sensor; faces: every flux once; cells: the
 *       Green-Gauss sum), reciprocals instead of repeated divisions: what a hand-fused CPU implementation of the same
 *       closure would do; agrees with the faithful form to rounding.
 * Parity unpinned against the reference itself (no Julia here): see DESIGN.md section 5.
 * Built by oracle/Makefile: gcc -O3 -ffp-contract=off -fopenmp.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t nd, nc;
    const float* spacing;               /* (nc, nd) column-major */
    int32_t nf[3];
    const int32_t *owners[3], *neighbors[3];          /* 0-based */
    const int32_t *loff[3], *lidx[3], *roff[3], *ridx[3]; /* CSR of the left / right face accumulators */
} ibo_part;

int ibo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ibo_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* Accumulator with weights 1/len (ImmersedBoundary.jl:501-506): sum_k v[idx_k] * w, left to right */
static inline float acc_mean(const int32_t* off, const int32_t* idx, int32_t c, const float* v) {
    const int32_t a = off[c], b = off[c + 1];
    if (b == a) return 0.0f;
    const float w = 1.0f / (float)(b - a);
    float s = v[idx[a]] * w;
    for (int32_t k = a + 1; k < b; ++k) s = s + v[idx[k]] * w;
    return s;
}
static inline float sgn(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
static inline float minmod(float a, float b) { return fminf(fabsf(a), fabsf(b)) * (sgn(a) + sgn(b)) / 2.0f; }

/* :899-910 */
void ibo_at_faces(const ibo_part* p, const float* u, int d, float* uf) {
    const float* h = p->spacing + (size_t)d * p->nc;
    const int32_t *o = p->owners[d], *n = p->neighbors[d];
#pragma omp parallel for schedule(static)
    for (int32_t f = 0; f < p->nf[d]; ++f) {
        const float ho = h[o[f]], hn = h[n[f]];
        uf[f] = (u[o[f]] * hn + u[n[f]] * ho) / (hn + ho);
    }
}
/* :918-926 (sign = -1) and :934-942 (sign = +1) */
void ibo_green_gauss(const ibo_part* p, const float* uf, int d, int unsigned_sum, float* out) {
    const float* h = p->spacing + (size_t)d * p->nc;
#pragma omp parallel for schedule(static)
    for (int32_t c = 0; c < p->nc; ++c) {
        const float r = acc_mean(p->roff[d], p->ridx[d], c, uf), l = acc_mean(p->loff[d], p->lidx[d], c, uf);
        out[c] = (unsigned_sum ? r + l : r - l) / h[c];
    }
}
/* :965-972; `uf` is scratch of nf[d] floats */
void ibo_cell_gradient(const ibo_part* p, const float* u, int d, float* uf, float* out) {
    ibo_at_faces(p, u, d, uf);
    ibo_green_gauss(p, uf, d, 0, out);
}
/* :1077-1097 with dim = 0; scratch: fd, fa (max nf), t1, t2 (nc) */
void ibo_jst_sensor(const ibo_part* p, const float* q, float* fd, float* fa, float* t1, float* t2, float* nu) {
#pragma omp parallel for schedule(static)
    for (int32_t c = 0; c < p->nc; ++c) nu[c] = 1e-7f;
    for (int d = 0; d < p->nd; ++d) {
        const int32_t *o = p->owners[d], *n = p->neighbors[d];
#pragma omp parallel for schedule(static)
        for (int32_t f = 0; f < p->nf[d]; ++f) {
            fd[f] = q[n[f]] - q[o[f]];
            fa[f] = fabsf(fd[f]);
        }
        ibo_green_gauss(p, fd, d, 0, t1);
        ibo_green_gauss(p, fa, d, 1, t2);
#pragma omp parallel for schedule(static)
        for (int32_t c = 0; c < p->nc; ++c) nu[c] = fmaxf(nu[c], (1e-7f + fabsf(t1[c])) / (1e-7f + t2[c]));
    }
}
/* :1113-1157 with D and high_order = true */
void ibo_muscl(const ibo_part* p, const float* u, const float* du, int d, const float* D, float* uL, float* uR) {
    const float* h = p->spacing + (size_t)d * p->nc;
    const int32_t *o = p->owners[d], *n = p->neighbors[d];
#pragma omp parallel for schedule(static)
    for (int32_t f = 0; f < p->nf[d]; ++f) {
        const float down = h[o[f]] / 2.0f, dneigh = h[n[f]] / 2.0f;
        const float uo = u[o[f]], un = u[n[f]], duo = du[o[f]], dun = du[n[f]];
        float guf = (un - uo) / (down + dneigh);
        const float gu = (2.0f * duo - guf) * down;
        const float Du = (2.0f * dun - guf) * dneigh;
        guf = minmod(Du, gu);
        float l = uo + guf, r = un - guf;
        const float Df = fmaxf(fmaxf(D[o[f]], D[n[f]]), 1e-7f);
        float uf = (uo * dneigh + un * down) / (down + dneigh);
        uf = uf + (duo * down - dun * dneigh) / 8.0f;
        uL[f] = l * Df + (1.0f - Df) * uf;
        uR[f] = r * Df + (1.0f - Df) * uf;
    }
}

static size_t max_nf(const ibo_part* p) {
    size_t m = 0;
    for (int d = 0; d < p->nd; ++d) if ((size_t)p->nf[d] > m) m = (size_t)p->nf[d];
    return m;
}

/* test/advection.jl:67-83 with ud starting from zero; C is (nc, nd) column-major with leading dimension ldc */
int ibo_residual_advection_faithful(const ibo_part* p, const float* u, const float* C, int64_t ldc, float* ud) {
    const size_t nf = max_nf(p), nc = (size_t)p->nc;
    float* w = (float*)malloc(sizeof(float) * (5 * nf + 4 * nc));
    if (!w) return -1;
    float *f0 = w, *f1 = f0 + nf, *f2 = f1 + nf, *f3 = f2 + nf, *f4 = f3 + nf;
    float *D = f4 + nf, *g = D + nc, *t1 = g + nc, *t2 = t1 + nc;
    memset(ud, 0, sizeof(float) * nc);
    ibo_jst_sensor(p, u, f0, f1, t1, t2, D);
    for (int d = 0; d < p->nd; ++d) {
        ibo_at_faces(p, C + (size_t)d * ldc, d, f0);            /* Cf */
        ibo_cell_gradient(p, u, d, f1, g);
        ibo_muscl(p, u, g, d, D, f2, f3);
#pragma omp parallel for schedule(static)
        for (int32_t f = 0; f < p->nf[d]; ++f)
            f4[f] = (f2[f] + f3[f]) * f0[f] / 2.0f + fabsf(f0[f]) * (f2[f] - f3[f]) / 2.0f;
        ibo_green_gauss(p, f4, d, 0, t1);
#pragma omp parallel for schedule(static)
        for (int32_t c = 0; c < p->nc; ++c) ud[c] -= t1[c];
    }
    free(w);
    return 0;
}

/* ---- fused form: three passes -- cells (gradients + sensor), faces (every flux ONCE), cells (Green-Gauss) ---- */
static inline float face_avg(const float* h, const float* u, int32_t o, int32_t n) {
    return (u[o] * h[n] + u[n] * h[o]) / (h[n] + h[o]);
}
int ibo_residual_advection_fused(const ibo_part* p, const float* u, const float* C, int64_t ldc, float* ud) {
    const size_t nc = (size_t)p->nc;
    const int nd = p->nd;
    size_t nftot = 0, foff[4] = {0, 0, 0, 0};
    for (int d = 0; d < nd; ++d) { foff[d] = nftot; nftot += (size_t)p->nf[d]; }
    float* G = (float*)malloc(sizeof(float) * ((size_t)(nd + 1) * nc + nftot)); /* gradients per dim, sensor, fluxes */
    if (!G) return -1;
    float* F = G + (size_t)(nd + 1) * nc;
#pragma omp parallel for schedule(static)
    for (int32_t c = 0; c < p->nc; ++c) {
        float nu = 1e-7f;
        for (int d = 0; d < nd; ++d) {
            const float* h = p->spacing + (size_t)d * nc;
            const int32_t *o = p->owners[d], *n = p->neighbors[d];
            float sr = 0, sl = 0, dr = 0, dl = 0, ar = 0, al = 0;
            int32_t a = p->roff[d][c], b = p->roff[d][c + 1];
            float w = b > a ? 1.0f / (float)(b - a) : 0.0f;
            for (int32_t k = a; k < b; ++k) {
                const int32_t f = p->ridx[d][k];
                const float df = u[n[f]] - u[o[f]];
                sr += face_avg(h, u, o[f], n[f]) * w; dr += df * w; ar += fabsf(df) * w;
            }
            a = p->loff[d][c]; b = p->loff[d][c + 1];
            w = b > a ? 1.0f / (float)(b - a) : 0.0f;
            for (int32_t k = a; k < b; ++k) {
                const int32_t f = p->lidx[d][k];
                const float df = u[n[f]] - u[o[f]];
                sl += face_avg(h, u, o[f], n[f]) * w; dl += df * w; al += fabsf(df) * w;
            }
            const float rh = 1.0f / h[c];
            G[(size_t)d * nc + c] = (sr - sl) * rh;
            nu = fmaxf(nu, (1e-7f + fabsf((dr - dl) * rh)) / (1e-7f + (ar + al) * rh));
        }
        G[(size_t)nd * nc + c] = nu;
    }
    const float* D = G + (size_t)nd * nc;
    for (int d = 0; d < nd; ++d) {
        const float* h = p->spacing + (size_t)d * nc;
        const float* g = G + (size_t)d * nc;
        const float* Cd = C + (size_t)d * ldc;
        const int32_t *o = p->owners[d], *n = p->neighbors[d];
        float* Fd = F + foff[d];
#pragma omp parallel for schedule(static)
        for (int32_t f = 0; f < p->nf[d]; ++f) {
            const int32_t oo = o[f], nn = n[f];
            const float down = h[oo] * 0.5f, dneigh = h[nn] * 0.5f, inv = 1.0f / (down + dneigh);
            const float uo = u[oo], un = u[nn], go = g[oo], gn = g[nn];
            float guf = (un - uo) * inv;
            const float gu = (2.0f * go - guf) * down, Du = (2.0f * gn - guf) * dneigh;
            guf = minmod(Du, gu);
            const float Df = fmaxf(fmaxf(D[oo], D[nn]), 1e-7f);
            const float uf = (uo * dneigh + un * down) * inv + (go * down - gn * dneigh) * 0.125f;
            const float uL = (uo + guf) * Df + (1.0f - Df) * uf, uR = (un - guf) * Df + (1.0f - Df) * uf;
            const float Cf = (Cd[oo] * dneigh + Cd[nn] * down) * inv;
            Fd[f] = 0.5f * ((uL + uR) * Cf + fabsf(Cf) * (uL - uR));
        }
    }
#pragma omp parallel for schedule(static)
    for (int32_t c = 0; c < p->nc; ++c) {
        float r = 0.0f;
        for (int d = 0; d < nd; ++d) {
            const float* Fd = F + foff[d];
            r -= (acc_mean(p->roff[d], p->ridx[d], c, Fd) - acc_mean(p->loff[d], p->lidx[d], c, Fd)) /
                 p->spacing[(size_t)d * nc + c];
        }
        ud[c] = r;
    }
    free(G);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Euler sweep: P = [p T u v (w)] (n, nv = nd + 2) column-major with leading dimension ldp; R likewise (ldr).
 * ------------------------------------------------------------------------------------------------------------------ */
/* CFD.primitive2state + the flux of one side, cfd.jl:106-123 and :470-500 (Float32) */
static inline void side_flux(int nd, const float* P, int dim, float Rg, float gamma, float* Q, float* F, float* un,
                             float* a) {
    const int nv = nd + 2;
    const float p = P[0];
    const float T = fmaxf(P[1], 10.0f);
    float k = P[2] * P[2];
    for (int j = 1; j < nd; ++j) k = k + P[2 + j] * P[2 + j];
    k = k / 2.0f;
    const float rho = p / (Rg * T);
    Q[0] = rho;
    Q[1] = rho * (Rg / (gamma - 1.0f) * T + k);
    for (int j = 0; j < nd; ++j) Q[2 + j] = rho * P[2 + j];
    for (int v = 0; v < nv; ++v) F[v] = Q[v];
    F[1] = F[1] + p;
    *un = P[2 + dim];
    *a = sqrtf(gamma * Rg * fmaxf(P[1], 10.0f));
    for (int v = 0; v < nv; ++v) F[v] = F[v] * (*un);
    F[2 + dim] = F[2 + dim] + p;
}

static inline double acc_mean_d(const int32_t* off, const int32_t* idx, int32_t c, const double* v) {
    const int32_t a = off[c], b = off[c + 1];
    if (b == a) return 0.0;
    const float w = 1.0f / (float)(b - a);
    double s = v[idx[a]] * (double)w;
    for (int32_t k = a + 1; k < b; ++k) s = s + v[idx[k]] * (double)w;
    return s;
}

int ibo_residual_euler_faithful(const ibo_part* p, const float* P, int64_t ldp, float Rg, float gamma, float* R,
                                int64_t ldr) {
    const int nd = p->nd, nv = nd + 2;
    const size_t nf = max_nf(p), nc = (size_t)p->nc;
    float* w = (float*)malloc(sizeof(float) * ((size_t)(2 + 2 * nv) * nf + 4 * nc));
    double* F = (double*)malloc(sizeof(double) * (size_t)nv * nf);
    if (!w || !F) { free(w); free(F); return -1; }
    float *f0 = w, *f1 = f0 + nf, *PL = f1 + nf, *PR = PL + (size_t)nv * nf;
    float *D = PR + (size_t)nv * nf, *g = D + nc, *t1 = g + nc, *t2 = t1 + nc;
    for (int v = 0; v < nv; ++v) memset(R + (size_t)v * ldr, 0, sizeof(float) * nc);
    ibo_jst_sensor(p, P, f0, f1, t1, t2, D);  /* pressure sensor */
    for (int d = 0; d < nd; ++d) {
        for (int v = 0; v < nv; ++v) {
            ibo_cell_gradient(p, P + (size_t)v * ldp, d, f0, g);
            ibo_muscl(p, P + (size_t)v * ldp, g, d, D, PL + (size_t)v * nf, PR + (size_t)v * nf);
        }
#pragma omp parallel for schedule(static)
        for (int32_t f = 0; f < p->nf[d]; ++f) {  /* CFD.inviscid_fluxes, HLL: the combine is Float64 (cfd.jl:504-507) */
            float pl[5], pr[5], QL[5], FL[5], QR[5], FR[5], uL, aL, uR, aR;
            for (int v = 0; v < nv; ++v) {
                pl[v] = PL[(size_t)v * nf + f];
                pr[v] = PR[(size_t)v * nf + f];
            }
            side_flux(nd, pl, d, Rg, gamma, QL, FL, &uL, &aL);
            side_flux(nd, pr, d, Rg, gamma, QR, FR, &uR, &aR);
            const double SR = fmin((double)(uR - aR), 0.0), SL = fmax((double)(uL + aL), 0.0);
            for (int v = 0; v < nv; ++v)
                F[(size_t)v * nf + f] = (SL * (double)FL[v] - SR * (double)FR[v] + SR * SL * (double)(QR[v] - QL[v])) / (SL - SR);
        }
        const float* h = p->spacing + (size_t)d * nc;
        for (int v = 0; v < nv; ++v) {
            const double* Fv = F + (size_t)v * nf;
            float* Rv = R + (size_t)v * ldr;
#pragma omp parallel for schedule(static)
            for (int32_t c = 0; c < p->nc; ++c) {
                const double gg = (acc_mean_d(p->roff[d], p->ridx[d], c, Fv) - acc_mean_d(p->loff[d], p->lidx[d], c, Fv)) /
                                  (double)h[c];
                Rv[c] = (float)((double)Rv[c] - gg);
            }
        }
    }
    free(w);
    free(F);
    return 0;
}
