/* Scalar-loop CPU restatement of the AIR spatial transformer -- TEST INFRASTRUCTURE ONLY.
 *
 * Independent (loop-by-loop, analytic-gradient) coding of the two resampling ops behind
 * attend_infer_repeat/modules.py:94-109 (SpatialTransformer = snt.AffineGridWarper + snt.resampler,
 * both un-vendored Sonnet v1.1; semantics restated in SURVEY.md Appendix A.4 / A.7):
 *   read  : glimpse[b,i,j] = bilinear(img[b],  x=(W-1)/2*(sx*X_j+tx+1), y=(H-1)/2*(sy*Y_i+ty+1))     cell.py:135
 *   write : out[b,I,J]     = bilinear(glm[b],  x=(w-1)/2*(X_J/sx-tx/sx+1), y likewise)                 cell.py:159
 * with where=[sx,tx,sy,ty], X=linspace(-1,1,n) (fp64 -> real), zero outside, sample valid iff -1<x<Ws && -1<y<Hs.
 * Gradients follow the resampler's registered gradient (d/dx = dy*(I_fc-I_ff)+(1-dy)*(I_cc-I_cf), ...).
 *
 * PARITY UNPINNED: the reference's tests hold no numbers for these ops; this file is checked against the torch
 * coding in oracle/air_oracle.py and grid_sample (tests/test_oracle_st.py).  The product never links it.
 *
 * Built twice by oracle/Makefile: -DREAL=float -> st_loops_f32.so, -DREAL=double -> st_loops_f64.so.
 */
#include <math.h>
#include <stddef.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

static real lin_m11(int k, int n) {
    if (n <= 1) return (real)-1.0;
    if (k == n - 1) return (real)1.0;
    return (real)(-1.0 + (double)k * (2.0 / (double)(n - 1)));
}

static real src_at(const real *src, int Hs, int Ws, long iy, long ix) {
    if (ix < 0 || iy < 0 || ix > Ws - 1 || iy > Hs - 1) return (real)0;
    return src[iy * Ws + ix];
}

/* one bilinear sample; returns value, and d value / dx, dy through gx, gy (0 when invalid) */
static real sample(const real *src, int Hs, int Ws, real x, real y, real *gx, real *gy,
                   long *ofx, long *ofy, real *odx, real *ody, int *valid) {
    *gx = 0; *gy = 0; *valid = 0;
    if (!(x > (real)-1 && y > (real)-1 && x < (real)Ws && y < (real)Hs)) return (real)0;
    real fxr = (real)floor((double)x), fyr = (real)floor((double)y);
    long fx = (long)fxr, fy = (long)fyr;
    real dx = (fxr + 1) - x, dy = (fyr + 1) - y;
    real ff = src_at(src, Hs, Ws, fy, fx), cc = src_at(src, Hs, Ws, fy + 1, fx + 1);
    real cf = src_at(src, Hs, Ws, fy + 1, fx), fc = src_at(src, Hs, Ws, fy, fx + 1);
    *gx = dy * (fc - ff) + (1 - dy) * (cc - cf);
    *gy = dx * (cf - ff) + (1 - dx) * (cc - fc);
    *ofx = fx; *ofy = fy; *odx = dx; *ody = dy; *valid = 1;
    return dx * dy * ff + (1 - dx) * (1 - dy) * cc + dx * (1 - dy) * cf + (1 - dx) * dy * fc;
}

static void scatter(real *dsrc, int Hs, int Ws, long fy, long fx, real dx, real dy, real g) {
    long ys[2] = {fy, fy + 1}, xs[2] = {fx, fx + 1};
    real wy[2] = {dy, 1 - dy}, wx[2] = {dx, 1 - dx};
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            if (xs[b] >= 0 && ys[a] >= 0 && xs[b] <= Ws - 1 && ys[a] <= Hs - 1)
                dsrc[ys[a] * Ws + xs[b]] += wy[a] * wx[b] * g;
}

/* ---- read ------------------------------------------------------------------------------------------- */
void st_read_fwd(const real *img, const real *where, real *out, int B, int H, int W, int h, int w) {
    for (int b = 0; b < B; ++b) {
        const real *wb = where + 4 * b;
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                real x = ((wb[0] * lin_m11(j, w) + wb[1]) + 1) * (real)((W - 1) / 2.0);
                real y = ((wb[2] * lin_m11(i, h) + wb[3]) + 1) * (real)((H - 1) / 2.0);
                real gx, gy, dx, dy; long fx, fy; int v;
                out[((size_t)b * h + i) * w + j] = sample(img + (size_t)b * H * W, H, W, x, y, &gx, &gy, &fx, &fy, &dx, &dy, &v);
            }
    }
}

/* dimg may be NULL */
void st_read_bwd(const real *img, const real *where, const real *dout, real *dwhere, real *dimg,
                 int B, int H, int W, int h, int w) {
    for (int b = 0; b < B; ++b) {
        const real *wb = where + 4 * b;
        real acc[4] = {0, 0, 0, 0};
        real cxs = (real)((W - 1) / 2.0), cys = (real)((H - 1) / 2.0);
        if (dimg) for (int k = 0; k < H * W; ++k) dimg[(size_t)b * H * W + k] = 0;
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                real X = lin_m11(j, w), Y = lin_m11(i, h);
                real x = ((wb[0] * X + wb[1]) + 1) * cxs;
                real y = ((wb[2] * Y + wb[3]) + 1) * cys;
                real gx, gy, dx, dy; long fx, fy; int v;
                sample(img + (size_t)b * H * W, H, W, x, y, &gx, &gy, &fx, &fy, &dx, &dy, &v);
                real g = dout[((size_t)b * h + i) * w + j];
                if (!v) continue;
                acc[0] += g * gx * cxs * X; acc[1] += g * gx * cxs;
                acc[2] += g * gy * cys * Y; acc[3] += g * gy * cys;
                if (dimg) scatter(dimg + (size_t)b * H * W, H, W, fy, fx, dx, dy, g);
            }
        for (int k = 0; k < 4; ++k) dwhere[4 * b + k] = acc[k];
    }
}

/* ---- write (inverse warp); out = inversed glimpse, NOT yet multiplied by presence --------------------- */
void st_write_fwd(const real *glm, const real *where, real *out, int B, int H, int W, int h, int w) {
    for (int b = 0; b < B; ++b) {
        const real *wb = where + 4 * b;
        real ax = (real)1 / wb[0], bx = -wb[1] / wb[0], ay = (real)1 / wb[2], by = -wb[3] / wb[2];
        for (int I = 0; I < H; ++I)
            for (int J = 0; J < W; ++J) {
                real x = ((ax * lin_m11(J, W) + bx) + 1) * (real)((w - 1) / 2.0);
                real y = ((ay * lin_m11(I, H) + by) + 1) * (real)((h - 1) / 2.0);
                real gx, gy, dx, dy; long fx, fy; int v;
                out[((size_t)b * H + I) * W + J] = sample(glm + (size_t)b * h * w, h, w, x, y, &gx, &gy, &fx, &fy, &dx, &dy, &v);
            }
    }
}

void st_write_bwd(const real *glm, const real *where, const real *dout, real *dglm, real *dwhere,
                  int B, int H, int W, int h, int w) {
    for (int b = 0; b < B; ++b) {
        const real *wb = where + 4 * b;
        real sx = wb[0], tx = wb[1], sy = wb[2], ty = wb[3];
        real ax = (real)1 / sx, bx = -tx / sx, ay = (real)1 / sy, by = -ty / sy;
        real cxs = (real)((w - 1) / 2.0), cys = (real)((h - 1) / 2.0);
        real da[4] = {0, 0, 0, 0}; /* d/d(ax), d/d(bx), d/d(ay), d/d(by) */
        for (int k = 0; k < h * w; ++k) dglm[(size_t)b * h * w + k] = 0;
        for (int I = 0; I < H; ++I)
            for (int J = 0; J < W; ++J) {
                real X = lin_m11(J, W), Y = lin_m11(I, H);
                real x = ((ax * X + bx) + 1) * cxs;
                real y = ((ay * Y + by) + 1) * cys;
                real gx, gy, dx, dy; long fx, fy; int v;
                sample(glm + (size_t)b * h * w, h, w, x, y, &gx, &gy, &fx, &fy, &dx, &dy, &v);
                if (!v) continue;
                real g = dout[((size_t)b * H + I) * W + J];
                da[0] += g * gx * cxs * X; da[1] += g * gx * cxs;
                da[2] += g * gy * cys * Y; da[3] += g * gy * cys;
                scatter(dglm + (size_t)b * h * w, h, w, fy, fx, dx, dy, g);
            }
        /* chain through a=1/s, b=-t/s */
        dwhere[4 * b + 0] = da[0] * (-(real)1 / (sx * sx)) + da[1] * (tx / (sx * sx));
        dwhere[4 * b + 1] = da[1] * (-(real)1 / sx);
        dwhere[4 * b + 2] = da[2] * (-(real)1 / (sy * sy)) + da[3] * (ty / (sy * sy));
        dwhere[4 * b + 3] = da[3] * (-(real)1 / sy);
    }
}
