/*
 * edge_oracle.c — CPU restatement of the reference's edge extractor (TEST INFRASTRUCTURE ONLY,
 * see rsreg_oracle.h: PARITY UNPINNED -- PCL is not available here; restated from PCL 1.9.1,
 * recalled).
 *
 * Reference: src/edge_extractor.hpp:7-39.  It runs IntegralImageNormalEstimation and
 * pcl::OrganizedEdgeFromRGBNormals with all five edge types switched on, but RETURNS ONLY
 * label_indices[4] (:36-38): the points whose label carries EDGELABEL_RGB_CANNY.  That bit is
 * set by OrganizedEdgeFromRGB::extractEdges alone, from the colours alone -- the normals, the
 * depth discontinuities and the curvature edges never reach the returned cloud.  What is
 * restated is therefore exactly that path:
 *   gray  = float((r + g + b) / 3)                     (integer division)       [organized_edge_detection.hpp]
 *   pcl::Edge::detectEdgeCanny, thresholds 40 / 100 (OrganizedEdgeFromRGB's defaults):
 *     3x3 Gaussian (sigma 1, kernel normalised in float) -> Sobel x, y -> magnitude sqrtf,
 *     direction atan2f (restated correctly rounded, see below) -> discretizeAngles -> suppressNonMaxima(tLow) on the interior ->
 *     hysteresis tracing from every pixel >= tHigh through non-zero pixels (8-neighbourhood)
 *   pcl::2d convolution: correlation, BOUNDARY_OPTION_CLAMP, float sum over kernel rows then columns
 * Output: the indices (ascending) of the points with Canny magnitude 255.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rsreg_oracle.h"

static void convolve3(const float *in, int w, int h, const float k[9], float *out)
{
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            float s = 0.0f;
            for (int kr = 0; kr < 3; kr++)
                for (int kc = 0; kc < 3; kc++) {
                    int r = i + kr - 1, c = j + kc - 1;
                    r = r < 0 ? 0 : (r >= h ? h - 1 : r);
                    c = c < 0 ? 0 : (c >= w ? w - 1 : c);
                    s += k[kr * 3 + kc] * in[r * w + c];
                }
            out[i * w + j] = s;
        }
}

int orc_edge_rgb_canny(const void *pts, size_t stride, uint32_t width, uint32_t height, float t_low, float t_high,
                       int32_t *indices_out, size_t *n_out)
{
    if (!pts || !n_out || stride < 20) return -1;
    const int w = (int)width, h = (int)height;
    const size_t n = (size_t)w * h;
    *n_out = 0;
    if (n == 0) return 0;
    float *gray = (float *)malloc(n * 4), *sm = (float *)malloc(n * 4), *gx = (float *)malloc(n * 4), *gy = (float *)malloc(n * 4);
    float *mag = (float *)malloc(n * 4), *dir = (float *)malloc(n * 4), *mx = (float *)calloc(n, 4);
    int32_t *stack = (int32_t *)malloc(n * 4);
    const char *b = (const char *)pts;
    for (size_t i = 0; i < n; i++) {
        unsigned char c[4];
        memcpy(c, b + i * stride + 16, 4); /* b g r a */
        gray[i] = (float)(((int)c[2] + (int)c[1] + (int)c[0]) / 3);
    }
    /* gaussianKernel(3, sigma 1) */
    float kg[9], sum = 0.0f;
    const double sigma_sqr = 2.0 * 1.0 * 1.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            const int iks = i - 1, jks = j - 1;
            /* PCL: expf (float (-double (iks * iks + jks * jks) / sigma_sqr)) at run time, i.e. the platform's libm in
             * the last ulp (and a compiler folding the call may round differently again); restated as the correctly
             * rounded float of the exponential: exp in double, rounded once */
            kg[i * 3 + j] = (float)exp(-(double)(iks * iks + jks * jks) / sigma_sqr);
            sum += kg[i * 3 + j];
        }
    for (int i = 0; i < 9; i++) kg[i] /= sum;
    static const float kx[9] = {-1, 0, 1, -2, 0, 2, -1, 0, 1}, ky[9] = {-1, -2, -1, 0, 0, 0, 1, 2, 1};
    convolve3(gray, w, h, kg, sm);
    convolve3(sm, w, h, kx, gx);
    convolve3(sm, w, h, ky, gy);
    for (size_t i = 0; i < n; i++) {
        mag[i] = sqrtf(gx[i] * gx[i] + gy[i] * gy[i]);
        /* PCL: direction = atan2f (gy, gx).  atan2f is libm-dependent in its last ulp (glibc 2.35's differs from
         * the correctly rounded value for 16 % of random inputs), so PCL's own class decisions at a boundary
         * depend on the platform; restated here as the correctly rounded float of the angle (atan2 in double,
         * rounded once), which every faithful libm approximates and the GPU can reproduce. */
        const float theta = (float)atan2((double)gy[i], (double)gx[i]);
        const float angle = theta * 57.29578f;                               /* pcl::rad2deg (float) */
        float d = theta;                                                     /* left as it is when no branch matches (NaN) */
        if (((angle <= 22.5f) && (angle >= -22.5f)) || (angle >= 157.5f) || (angle <= -157.5f)) d = 0;
        else if (((angle > 22.5f) && (angle < 67.5f)) || ((angle < -112.5f) && (angle > -157.5f))) d = 45;
        else if (((angle >= 67.5f) && (angle <= 112.5f)) || ((angle <= -67.5f) && (angle >= -112.5f))) d = 90;
        else if (((angle > 112.5f) && (angle < 157.5f)) || ((angle < -22.5f) && (angle > -67.5f))) d = 135;
        dir[i] = d;
    }
    /* suppressNonMaxima: interior pixels only */
    for (int i = 1; i < h - 1; i++)
        for (int j = 1; j < w - 1; j++) {
            const float m = mag[i * w + j];
            if (m < t_low) continue;
            float a, c2;
            switch ((int)dir[i * w + j]) {
                case 0: a = mag[i * w + j - 1]; c2 = mag[i * w + j + 1]; break;
                case 45: a = mag[(i - 1) * w + j - 1]; c2 = mag[(i + 1) * w + j + 1]; break;
                case 90: a = mag[(i - 1) * w + j]; c2 = mag[(i + 1) * w + j]; break;
                case 135: a = mag[(i - 1) * w + j + 1]; c2 = mag[(i + 1) * w + j - 1]; break;
                default: continue;
            }
            if (m >= a && m >= c2) mx[i * w + j] = m;
        }
    /* hysteresis: from every pixel >= tHigh, through non-zero pixels with row > 0 and col > 0 */
    const float MARK = 3.402823466e+38f;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            if (mx[i * w + j] < t_high || mx[i * w + j] == MARK) continue;
            mx[i * w + j] = MARK;
            size_t sp = 0;
            stack[sp++] = i * w + j;
            while (sp) {
                const int p = stack[--sp], r = p / w, c = p % w;
                for (int dr = -1; dr <= 1; dr++)
                    for (int dc = -1; dc <= 1; dc++) {
                        if (!dr && !dc) continue;
                        const int nr = r + dr, nc = c + dc;
                        if (!(nr > 0 && nr < h && nc > 0 && nc < w)) continue;
                        float *q = &mx[nr * w + nc];
                        if (*q == 0.0f || *q == MARK) continue;
                        *q = MARK;
                        stack[sp++] = nr * w + nc;
                    }
            }
        }
    size_t cnt = 0;
    for (size_t i = 0; i < n; i++)
        if (mx[i] == MARK) {
            if (indices_out) indices_out[cnt] = (int32_t)i;
            cnt++;
        }
    *n_out = cnt;
    free(gray); free(sm); free(gx); free(gy); free(mag); free(dir); free(mx); free(stack);
    return 0;
}
