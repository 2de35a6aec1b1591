// ba_oracle.c — CPU restatement of the per-pair normal-equation blocks of the bundle adjustment
// (SURVEY.md section 8(f) rank 3).
//
// TEST INFRASTRUCTURE ONLY: called from tests/ as the checker for the HIP path; never linked into or called by the
// product library.
//
// Follows PP/bundleAdjustment/bundleAdjustmentRKf.m:
//   :717-741   the parfor body of accumulateNormalEqnsBlock: Hii = Ji'Ji, Hjj = Jj'Jj, Hij = Ji'Jj, gi = Ji'r, gj = Jj'r
//   :793-899   jacobianPair: per match the j->i rows and (unless opts.OneDirection) the i->j rows, Jacobians at the
//              base cameras, residuals at the incremented cameras, Huber weight on the residual norm
//   :1641-1686 computeSingleResidual: pH = K_o R_o R_s' (K_s \ [u;1]), |z| < 1e-10 -> 1e-10, r = uObs - pH(1:2)/pH(3)
//   :1688-1783 computeJacobianWrtCamera ('obs' and 'src'; columns [dthx dthy dthz df])
//   :1806-1829 huberWeight
//
// PARITY UNPINNED (no MATLAB here, the reference ships no vectors for this path).  MATLAB leaves the evaluation order
// of its matrix products and sums to the BLAS; it is fixed here once and mirrored by csrc/ba.hip:
//   * a matrix chain is evaluated left to right as written in the reference, every 3x3 product / mat-vec with the
//     inner index ascending, multiply and add separately (no fma);
//   * K \ [x;y;1] by back substitution: z = 1, y' = (y - cy z)/f, x' = (x - cx z)/f;
//   * norm(r) = sqrt(r1*r1 + r2*r2);
//   * the sums over the rows of a pair: 64 lane-strided partial sums over the matches (lane l takes matches
//     l, l+64, ...; a match adds its two or four rows in order), then an xor butterfly 32,16,...,1 (the wavefront
//     order of the device);
//   * every block is 4 x 4 / 4 x 1 over the columns [dthx dthy dthz df]; a camera with fewer parameters uses the
//     leading columns, exactly as the reference's JijI(:, 1:numel(colsI)) does (for the one-parameter seed camera
//     that is dthx, not df - kept).
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct {
    double f, cx, cy;
    double R[9];  // column-major
} ba_cam;

#define M3(A, r, c) (A)[(r) + 3 * (c)]

static void mul33(const double* A, const double* B, double* C) {
    double T[9];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
            double s = M3(A, r, 0) * M3(B, 0, c);
            s = s + M3(A, r, 1) * M3(B, 1, c);
            s = s + M3(A, r, 2) * M3(B, 2, c);
            M3(T, r, c) = s;
        }
    memcpy(C, T, sizeof T);
}

static void mulv(const double* A, const double* x, double* y) {
    for (int r = 0; r < 3; ++r) {
        double s = M3(A, r, 0) * x[0];
        s = s + M3(A, r, 1) * x[1];
        s = s + M3(A, r, 2) * x[2];
        y[r] = s;
    }
}

static void transpose33(const double* A, double* T) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M3(T, c, r) = M3(A, r, c);
}

static void kmat(const ba_cam* c, double* K) {
    memset(K, 0, 9 * sizeof(double));
    M3(K, 0, 0) = c->f;
    M3(K, 1, 1) = c->f;
    M3(K, 0, 2) = c->cx;
    M3(K, 1, 2) = c->cy;
    M3(K, 2, 2) = 1.0;
}

static void skew_unit(int m, double* S) {  // skewSymmetric(e_m), :1785-1804
    memset(S, 0, 9 * sizeof(double));
    double v[3] = {0, 0, 0};
    v[m] = 1.0;
    M3(S, 0, 1) = -v[2];
    M3(S, 0, 2) = v[1];
    M3(S, 1, 0) = v[2];
    M3(S, 1, 2) = -v[0];
    M3(S, 2, 0) = -v[1];
    M3(S, 2, 1) = v[0];
}

static void ksolve(const ba_cam* c, double x, double y, double* out) {
    const double z = 1.0;
    out[2] = z;
    out[1] = (y - c->cy * z) / c->f;
    out[0] = (x - c->cx * z) / c->f;
}

typedef struct {
    double M[9];      // K_o R_o R_s'  (base cameras)
    double G[3][9];   // (K_o R_o [e_m]x) R_s'
    double N[3][9];   // (K_o R_o) (-R_s' [e_m]x)
    double D[9];      // (dKdf R_o) R_s'
    double Q[9];      // (K_o R_o R_s') dKinvdf(src)
    double ML[9];     // K_o R_o R_s'  (incremented cameras)
} dir_mats;

static void make_dir(const ba_cam* ob, const ba_cam* sb, const ba_cam* ol, const ba_cam* sl, dir_mats* d) {
    double K[9], A[9], RsT[9], S[9], T[9];
    kmat(ob, K);
    mul33(K, ob->R, A);
    transpose33(sb->R, RsT);
    mul33(A, RsT, d->M);
    for (int m = 0; m < 3; ++m) {
        skew_unit(m, S);
        mul33(A, S, T);
        mul33(T, RsT, d->G[m]);
        double nR[9];
        for (int e = 0; e < 9; ++e) nR[e] = -RsT[e];
        mul33(nR, S, T);
        mul33(A, T, d->N[m]);
    }
    double dK[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0};
    mul33(dK, ob->R, T);
    mul33(T, RsT, d->D);
    const double f = sb->f;
    double dKi[9];
    memset(dKi, 0, sizeof dKi);
    M3(dKi, 0, 0) = -1.0 / (f * f);
    M3(dKi, 1, 1) = -1.0 / (f * f);
    M3(dKi, 0, 2) = sb->cx / (f * f);
    M3(dKi, 1, 2) = sb->cy / (f * f);
    mul33(d->M, dKi, d->Q);
    kmat(ol, K);
    mul33(K, ol->R, A);
    transpose33(sl->R, RsT);
    mul33(A, RsT, d->ML);
}

// one direction of one match: residual (incremented cameras) and the two 2x4 Jacobians (base cameras), scaled by
// sqrt(huber weight); returns w * r'r
static double one_direction(const dir_mats* d, const ba_cam* sb, const ba_cam* sl, double uox, double uoy, double usx,
                            double usy, double sigma, double* r, double Jobs[2][4], double Jsrc[2][4]) {
    double xb[3], pH[3], v[3];
    ksolve(sb, usx, usy, xb);
    mulv(d->M, xb, pH);
    double x = pH[0], y = pH[1], z = pH[2];
    if (fabs(z) < 1e-10) z = 1e-10;
    const double iz = 1.0 / z, zz = z * z;
    const double a = -iz, cx_ = x / zz, cy_ = y / zz;  // Jchain = -Jdehom = [-1/z 0 x/z^2; 0 -1/z y/z^2]
    for (int m = 0; m < 3; ++m) {
        mulv(d->G[m], xb, v);
        Jobs[0][m] = a * v[0] + cx_ * v[2];
        Jobs[1][m] = a * v[1] + cy_ * v[2];
        mulv(d->N[m], xb, v);
        Jsrc[0][m] = a * v[0] + cx_ * v[2];
        Jsrc[1][m] = a * v[1] + cy_ * v[2];
    }
    mulv(d->D, xb, v);
    Jobs[0][3] = a * v[0] + cx_ * v[2];
    Jobs[1][3] = a * v[1] + cy_ * v[2];
    const double uh[3] = {usx, usy, 1.0};
    mulv(d->Q, uh, v);
    Jsrc[0][3] = a * v[0] + cx_ * v[2];
    Jsrc[1][3] = a * v[1] + cy_ * v[2];
    // residual at the incremented cameras
    double xl[3], pL[3];
    ksolve(sl, usx, usy, xl);
    mulv(d->ML, xl, pL);
    double zl = pL[2];
    if (fabs(zl) < 1e-10) zl = 1e-10;
    const double r0 = uox - pL[0] / zl, r1 = uoy - pL[1] / zl;
    const double rr = r0 * r0 + r1 * r1;
    const double nr = sqrt(rr);
    const double w = nr < sigma ? 1.0 : sigma / nr;
    const double sw = sqrt(w);
    r[0] = sw * r0;
    r[1] = sw * r1;
    for (int q = 0; q < 2; ++q)
        for (int e = 0; e < 4; ++e) {
            Jobs[q][e] = sw * Jobs[q][e];
            Jsrc[q][e] = sw * Jsrc[q][e];
        }
    return (sw * sw) * rr;
}

#define NACC 59  // Hii 16, Hjj 16, Hij 16 (column-major 4x4), gi 4, gj 4, E, r2sum, rcnt

static void add_rows(double* acc, const double* r, double Ji[2][4], double Jj[2][4]) {
    for (int q = 0; q < 2; ++q) {
        for (int b = 0; b < 4; ++b)
            for (int a = 0; a < 4; ++a) {
                acc[a + 4 * b] = acc[a + 4 * b] + Ji[q][a] * Ji[q][b];
                acc[16 + a + 4 * b] = acc[16 + a + 4 * b] + Jj[q][a] * Jj[q][b];
                acc[32 + a + 4 * b] = acc[32 + a + 4 * b] + Ji[q][a] * Jj[q][b];
            }
        for (int a = 0; a < 4; ++a) {
            acc[48 + a] = acc[48 + a] + Ji[q][a] * r[q];
            acc[52 + a] = acc[52 + a] + Jj[q][a] * r[q];
        }
    }
}

// cams: n_pairs x 4 cameras (base i, base j, incremented i, incremented j), 12 doubles each (f, cx, cy, R col-major)
// Ui, Uj: total x 2 column-major (x at [k], y at [ldu + k]); pair p owns rows pair_ptr[p] .. pair_ptr[p+1]-1
// out: n_pairs x 59
ORC_API void orc_ba_pair_blocks(const double* Ui, const double* Uj, int64_t ldu, const int64_t* pair_ptr, int n_pairs,
                                const double* cams, double sigma, int both, double* out) {
    for (int p = 0; p < n_pairs; ++p) {
        ba_cam c[4];
        for (int q = 0; q < 4; ++q) {
            const double* s = cams + ((int64_t)p * 4 + q) * 12;
            c[q].f = s[0];
            c[q].cx = s[1];
            c[q].cy = s[2];
            memcpy(c[q].R, s + 3, 9 * sizeof(double));
        }
        dir_mats dji, dij;
        make_dir(&c[0], &c[1], &c[2], &c[3], &dji);  // j -> i: observed in i, source j
        make_dir(&c[1], &c[0], &c[3], &c[2], &dij);  // i -> j
        static double part[64][NACC];
        memset(part, 0, sizeof part);
        const int64_t r0 = pair_ptr[p], m = pair_ptr[p + 1] - r0;
        for (int64_t k = 0; k < m; ++k) {
            double* acc = part[k & 63];
            const double uix = Ui[r0 + k], uiy = Ui[ldu + r0 + k], ujx = Uj[r0 + k], ujy = Uj[ldu + r0 + k];
            double r[2], Jo[2][4], Js[2][4];
            double wr = one_direction(&dji, &c[1], &c[3], uix, uiy, ujx, ujy, sigma, r, Jo, Js);
            add_rows(acc, r, Jo, Js);  // Ji = dJ/d(cam i) = obs, Jj = src
            acc[56] = acc[56] + 0.5 * wr;
            acc[57] = acc[57] + wr;
            acc[58] = acc[58] + 2.0;
            if (both) {
                wr = one_direction(&dij, &c[0], &c[2], ujx, ujy, uix, uiy, sigma, r, Jo, Js);
                add_rows(acc, r, Js, Jo);  // roles swap: Ji = src, Jj = obs
                acc[56] = acc[56] + 0.5 * wr;
                acc[57] = acc[57] + wr;
                acc[58] = acc[58] + 2.0;
            }
        }
        for (int s = 32; s > 0; s >>= 1) {
            static double nxt[64][NACC];
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < NACC; ++e) nxt[l][e] = part[l][e] + part[l ^ s][e];
            memcpy(part, nxt, sizeof part);
        }
        memcpy(out + (int64_t)p * NACC, part[0], NACC * sizeof(double));
    }
}
