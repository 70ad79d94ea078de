/*
 * oracle/unproject_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (fp32) restatement of the reference's volumetric un-projection, used only as the
 * parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
 * under multiviewhmr_amd/ may link, import or call it.
 *
 * Parity pinning: validated against golden vectors produced by running the reference itself
 * in the build container (tests/golden/make_golden.py -> tests/golden/unproj_*.npz), see
 * tests/test_oracle_golden.py.  The third-party arithmetic on the path (torch
 * F.grid_sample == ATen grid_sampler_2d bilinear / zeros / align_corners=True, F.softmax) is
 * restated from its published algorithm; torch is un-pinned by the reference (no requirements
 * file), the goldens were produced with torch 2.10.0.
 *
 * What it follows (reference file:line, relative to /root/reference):
 *   models/aggregation.py:20-87    unprojection(): loop structure, masks, aggregation modes
 *   utils/multiview.py:55-69       euclidean_to_homogeneous  -> [X, 1]
 *   utils/multiview.py:89-110      project_3d_points_to_image_plane_without_distortion -> [X,1] @ P^T
 *   utils/multiview.py:72-86       homogeneous_to_euclidean  -> divide by last
 *   models/aggregation.py:42-44    invalid = z <= 0 ; z == 0 -> 1
 *   models/aggregation.py:48-51    grid = 2 * (uv / feature_shape[i] - 0.5)   (x by Hf, y by Wf: quirk Q1)
 *   models/aggregation.py:55-58    F.grid_sample(..., align_corners=True)   bilinear, zero padding
 *   models/aggregation.py:61-62    zero the invalid voxels (they still take part in the aggregate: Q2)
 *   models/aggregation.py:71-85    sum | mean | max | softmax-weighted sum over views
 *
 * Layouts (all contiguous, fp32):
 *   features (B,V,C,Hf,Wf)   proj (B,V,3,4)   coords (B,N,3)   out / grad_out (B,C,N)
 *   N = X*Y*Z voxels in row-major (x,y,z) order, i.e. coord_volumes[b].reshape(-1,3).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

enum { AGG_SOFTMAX = 0, AGG_SUM = 1, AGG_MEAN = 2, AGG_MAX = 3 };
#define MAXV 64

/* One voxel seen by one view: the four bilinear taps (offset into a Hf*Wf plane, weight;
 * in[k] = 0 when the tap is outside the map) -- ATen grid_sampler_2d, bilinear. */
typedef struct { int off[4]; float w[4]; unsigned char in[4]; } taps_t;

static void make_taps(const float *P, const float *X, int Hf, int Wf, taps_t *t)
{
    /* [X,1] @ P^T  (multiview.py:105).  ATen's CPU sgemm accumulates k = 0..3 as one FMA chain
     * (checked bit-for-bit against torch 2.10 in the build container); restated the same way. */
    float a = fmaf(P[3], 1.f, fmaf(P[2], X[2], fmaf(P[1], X[1], P[0] * X[0])));
    float b = fmaf(P[7], 1.f, fmaf(P[6], X[2], fmaf(P[5], X[1], P[4] * X[0])));
    float z = fmaf(P[11], 1.f, fmaf(P[10], X[2], fmaf(P[9], X[1], P[8] * X[0])));
    for (int k = 0; k < 4; ++k) { t->off[k] = 0; t->w[k] = 0.f; t->in[k] = 0; }
    if (z <= 0.f) return;                 /* aggregation.py:42 + :62 -> sample is exactly 0 */
    /* z == 0 -> 1 (aggregation.py:44) can only matter when invalid, already handled */
    float u = a / z, v = b / z;           /* multiview.py:84 */
    float gx = 2.f * (u / (float)Hf - 0.5f);   /* aggregation.py:49  (Q1: x by feature_shape[0] = Hf) */
    float gy = 2.f * (v / (float)Wf - 0.5f);   /* aggregation.py:50  (Q1: y by feature_shape[1] = Wf) */
    /* ATen grid_sampler_unnormalize, align_corners=True: ((g + 1) / 2) * (size - 1) */
    float ix = ((gx + 1.f) / 2.f) * (float)(Wf - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(Hf - 1);
    /* anything not strictly inside (-1, size) has all taps out of range; also catches NaN/inf */
    if (!(ix > -1.f && ix < (float)Wf && iy > -1.f && iy < (float)Hf)) return;
    float fx0 = floorf(ix), fy0 = floorf(iy);
    int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    /* ATen: nw = (ix_se - ix)*(iy_se - iy) ; ne = (ix - ix_sw)*(iy_sw - iy) ; sw = (ix_ne - ix)*(iy - iy_ne) ; se */
    float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
    int xs[4] = { x0, x1, x0, x1 }, ys[4] = { y0, y0, y1, y1 };
    float ws[4] = { wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1 };
    for (int k = 0; k < 4; ++k) {
        t->w[k] = ws[k];
        if (xs[k] >= 0 && xs[k] < Wf && ys[k] >= 0 && ys[k] < Hf) { t->off[k] = ys[k] * Wf + xs[k]; t->in[k] = 1; }
    }
}

static inline float sample(const float *plane, const taps_t *t)
{
    /* ATen: nw_val*nw + ne_val*ne + sw_val*sw + se_val*se, an out-of-range tap reads as 0; its
     * vectorised CPU kernel contracts this to mul, fma, fma, fma (checked bit-for-bit) */
    float v0 = t->in[0] ? plane[t->off[0]] : 0.f, v1 = t->in[1] ? plane[t->off[1]] : 0.f;
    float v2 = t->in[2] ? plane[t->off[2]] : 0.f, v3 = t->in[3] ? plane[t->off[3]] : 0.f;
    return fmaf(v3, t->w[3], fmaf(v2, t->w[2], fmaf(v1, t->w[1], v0 * t->w[0])));
}

static inline float aggregate(const float *s, int V, int method)
{
    float r;
    switch (method) {
    case AGG_SUM:  r = 0.f; for (int v = 0; v < V; ++v) r += s[v]; return r;
    case AGG_MEAN: r = 0.f; for (int v = 0; v < V; ++v) r += s[v]; return r / (float)V;
    case AGG_MAX:  r = s[0]; for (int v = 1; v < V; ++v) r = s[v] > r ? s[v] : r; return r;
    default: {
        float m = s[0], den = 0.f, e[MAXV];
        for (int v = 1; v < V; ++v) m = s[v] > m ? s[v] : m;
        for (int v = 0; v < V; ++v) { e[v] = expf(s[v] - m); den += e[v]; }
        r = 0.f; for (int v = 0; v < V; ++v) r += s[v] * (e[v] / den);   /* x * softmax(x), summed over views */
        return r; }
    }
}

#define VOXBLK 128

int mvhmr_oracle_unproject_forward(const float *features, const float *proj, const float *coords, float *out,
                                   int B, int V, int C, int Hf, int Wf, int64_t N, int method)
{
    if (V < 1 || V > MAXV || method < 0 || method > 3) return -1;
    const int64_t plane = (int64_t)Hf * Wf, nblk = (N + VOXBLK - 1) / VOXBLK;
#pragma omp parallel
    {
        taps_t *tp = (taps_t *)malloc(sizeof(taps_t) * (size_t)V * VOXBLK);
#pragma omp for collapse(2) schedule(dynamic, 4)
        for (int b = 0; b < B; ++b)
            for (int64_t blk = 0; blk < nblk; ++blk) {
                int64_t n0 = blk * VOXBLK, n1 = n0 + VOXBLK < N ? n0 + VOXBLK : N;
                for (int v = 0; v < V; ++v)
                    for (int64_t n = n0; n < n1; ++n)
                        make_taps(proj + ((int64_t)b * V + v) * 12, coords + ((int64_t)b * N + n) * 3, Hf, Wf,
                                  &tp[(size_t)v * VOXBLK + (n - n0)]);
                for (int c = 0; c < C; ++c)
                    for (int64_t n = n0; n < n1; ++n) {
                        float s[MAXV];
                        for (int v = 0; v < V; ++v)
                            s[v] = sample(features + (((int64_t)b * V + v) * C + c) * plane, &tp[(size_t)v * VOXBLK + (n - n0)]);
                        out[((int64_t)b * C + c) * N + n] = aggregate(s, V, method);
                    }
            }
        free(tp);
    }
    return 0;
}

/* d(out)/d(features): autograd of the reference graph.
 *   softmax: ds_v = g * p_v * (1 + s_v - out)      sum: g      mean: g / V      max: g at the (first) arg-max
 *   ds_v = 0 where z <= 0 (masked assignment, aggregation.py:62); then grid_sampler_2d_backward scatters
 *   ds_v * w_k into the in-range taps. */
int mvhmr_oracle_unproject_backward(const float *grad_out, const float *features, const float *proj,
                                    const float *coords, float *grad_features,
                                    int B, int V, int C, int Hf, int Wf, int64_t N, int method)
{
    if (V < 1 || V > MAXV || method < 0 || method > 3) return -1;
    const int64_t plane = (int64_t)Hf * Wf;
    memset(grad_features, 0, sizeof(float) * (size_t)B * V * C * plane);
    /* one task owns every view's plane of one (b, c): no two tasks write the same element */
#pragma omp parallel
    {
        taps_t *tp = (taps_t *)malloc(sizeof(taps_t) * (size_t)V);
#pragma omp for collapse(2) schedule(dynamic, 1)
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < C; ++c)
                for (int64_t n = 0; n < N; ++n) {
                    float s[MAXV], ds[MAXV];
                    for (int v = 0; v < V; ++v) {
                        make_taps(proj + ((int64_t)b * V + v) * 12, coords + ((int64_t)b * N + n) * 3, Hf, Wf, &tp[v]);
                        s[v] = sample(features + (((int64_t)b * V + v) * C + c) * plane, &tp[v]);
                    }
                    float g = grad_out[((int64_t)b * C + c) * N + n];
                    if (method == AGG_SUM) for (int v = 0; v < V; ++v) ds[v] = g;
                    else if (method == AGG_MEAN) for (int v = 0; v < V; ++v) ds[v] = g / (float)V;
                    else if (method == AGG_MAX) {
                        int am = 0; for (int v = 1; v < V; ++v) if (s[v] > s[am]) am = v;
                        for (int v = 0; v < V; ++v) ds[v] = v == am ? g : 0.f;
                    } else {
                        float m = s[0], den = 0.f, e[MAXV], o = 0.f;
                        for (int v = 1; v < V; ++v) m = s[v] > m ? s[v] : m;
                        for (int v = 0; v < V; ++v) { e[v] = expf(s[v] - m); den += e[v]; }
                        for (int v = 0; v < V; ++v) o += s[v] * (e[v] / den);
                        for (int v = 0; v < V; ++v) ds[v] = g * (e[v] / den) * (1.f + s[v] - o);
                    }
                    for (int v = 0; v < V; ++v) {
                        float *gp = grad_features + (((int64_t)b * V + v) * C + c) * plane;
                        for (int k = 0; k < 4; ++k) if (tp[v].in[k]) gp[tp[v].off[k]] += ds[v] * tp[v].w[k];
                    }
                }
        free(tp);
    }
    return 0;
}
