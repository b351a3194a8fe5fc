/*
 * oracle/warp_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the frame-prediction task the reference runs on the flow field,
 * /root/reference/optical_flow_prediction.py:
 *   flow_map = grid + flow (float64) -> float32                       :289-290, :338-339, :581-582
 *   cv2.remap(channel, map_x, map_y, INTER_LINEAR[, BORDER_REPLICATE]) :293-300, :342-349, :584-586
 *   structural_similarity(true[:,:,2], pred[:,:,2], data_range=255.0)  :113-115
 *
 * cv2.remap lives in opencv-python (requirements.txt:1), absent here; restated from the published algorithm of
 * imgproc's remap for 8-bit sources and CV_32FC1 maps with INTER_LINEAR:
 *   sx = cvRound(map_x * 32), sy likewise (round half to even); integer part sx >> 5 saturated to int16,
 *   fraction sx & 31; weights ((32-fx)(32-fy), fx(32-fy), (32-fx)fy, fx fy) * 32 (sum 2^15);
 *   dst = (v0 w0 + v1 w1 + v2 w2 + v3 w3 + 2^14) >> 15;
 *   taps outside the source: BORDER_REPLICATE clamps the tap coordinates, BORDER_CONSTANT (cv2's default, value 0)
 *   substitutes the constant, and writes the constant outright when the 2x2 footprint misses the image entirely.
 * PARITY UNPINNED for remap: the reference holds no prediction outputs, and cv2 cannot be run here.
 *
 * structural_similarity lives in scikit-image (requirements.txt: 0.23.2).  PINNED: tests/golden/ssim_golden.npz was
 * produced by calling skimage.metrics.structural_similarity (0.18.3, the version present in the build container;
 * same defaults: 7x7 uniform window, sample covariance, K1 0.01, K2 0.03) -- tests/golden/gen_ssim_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* border: 0 = BORDER_CONSTANT with value cval, 1 = BORDER_REPLICATE.  src: [sh][sstride] bytes, `cn` interleaved
 * channels; maps [dh][mstride] floats; dst [dh][dstride] bytes, cn interleaved channels. */
int nsof_ref_remap_linear_u8(const uint8_t* src, ptrdiff_t sstride, int sw, int sh, int cn, const float* mapx,
                             const float* mapy, ptrdiff_t mstride, int dw, int dh, int border, int cval, uint8_t* dst,
                             ptrdiff_t dstride)
{
    if (sw < 1 || sh < 1 || cn < 1 || dw < 0 || dh < 0 || (border != 0 && border != 1)) return -1;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            long lx = lrintf(mapx[y * mstride + x] * 32.0f), ly = lrintf(mapy[y * mstride + x] * 32.0f);
            int fx = (int)(lx & 31), fy = (int)(ly & 31);
            int ix = clampi((int)(lx >> 5), -32768, 32767), iy = clampi((int)(ly >> 5), -32768, 32767);
            int w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
            for (int c = 0; c < cn; c++) {
                int v0, v1, v2, v3;
                if (border == 1) {
                    int x0 = clampi(ix, 0, sw - 1), x1 = clampi(ix + 1, 0, sw - 1);
                    int y0 = clampi(iy, 0, sh - 1), y1 = clampi(iy + 1, 0, sh - 1);
                    v0 = src[y0 * sstride + x0 * cn + c]; v1 = src[y0 * sstride + x1 * cn + c];
                    v2 = src[y1 * sstride + x0 * cn + c]; v3 = src[y1 * sstride + x1 * cn + c];
                } else {
                    if (ix >= sw || ix + 1 < 0 || iy >= sh || iy + 1 < 0) {
                        dst[y * dstride + x * cn + c] = (uint8_t)cval;
                        continue;
                    }
                    int xin0 = ix >= 0 && ix < sw, xin1 = ix + 1 >= 0 && ix + 1 < sw;
                    int yin0 = iy >= 0 && iy < sh, yin1 = iy + 1 >= 0 && iy + 1 < sh;
                    v0 = xin0 && yin0 ? src[iy * sstride + ix * cn + c] : cval;
                    v1 = xin1 && yin0 ? src[iy * sstride + (ix + 1) * cn + c] : cval;
                    v2 = xin0 && yin1 ? src[(iy + 1) * sstride + ix * cn + c] : cval;
                    v3 = xin1 && yin1 ? src[(iy + 1) * sstride + (ix + 1) * cn + c] : cval;
                }
                int r = (v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3 + (1 << 14)) >> 15;
                dst[y * dstride + x * cn + c] = (uint8_t)clampi(r, 0, 255);
            }
        }
    return 0;
}

/* The reference's map for the crop [y0,y1) x [x0,x1) of a flow canvas: float32(float64(x) + float64(sign * f)),
 * sign = -1 for Farneback ("flow = -flow", prediction.py:545,574).  flow: float32 canvas [H][fstride] (u,v). */
int nsof_ref_flow_map(const float* flow, ptrdiff_t fstride, int x0, int y0, int x1, int y1, int sign, float* mapx,
                      float* mapy)
{
    int rw = x1 - x0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            double u = flow[y * fstride + 2 * x], v = flow[y * fstride + 2 * x + 1];
            mapx[(size_t)(y - y0) * rw + (x - x0)] = (float)((double)x + sign * u);
            mapy[(size_t)(y - y0) * rw + (x - x0)] = (float)((double)y + sign * v);
        }
    return 0;
}

/* skimage.metrics.structural_similarity(a, b, data_range=R) for 8-bit images (defaults: win_size 7, uniform
 * window, sample covariance): mean over the interior (3-px margin) of S.  pixel step `ps` lets a and b be one
 * channel of an interleaved image.  Window sums are exact integers here; skimage's running-mean filter rounds
 * at the 1e-16 level. */
int nsof_ref_ssim_u8(const uint8_t* a, ptrdiff_t astride, int aps, const uint8_t* b, ptrdiff_t bstride, int bps, int w,
                     int h, double data_range, double* out)
{
    const int win = 7, pad = 3;
    if (w < win || h < win) return -1;
    const double NP = win * win, cov_norm = NP / (NP - 1);
    const double C1 = (0.01 * data_range) * (0.01 * data_range), C2 = (0.03 * data_range) * (0.03 * data_range);
    double total = 0;
    for (int y = pad; y < h - pad; y++) {
        double row = 0;
        for (int x = pad; x < w - pad; x++) {
            long sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
            for (int j = -pad; j <= pad; j++)
                for (int i = -pad; i <= pad; i++) {
                    long p = a[(y + j) * astride + (x + i) * aps], q = b[(y + j) * bstride + (x + i) * bps];
                    sx += p; sy += q; sxx += p * p; syy += q * q; sxy += p * q;
                }
            double ux = sx / NP, uy = sy / NP, uxx = sxx / NP, uyy = syy / NP, uxy = sxy / NP;
            double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
            double A1 = 2 * ux * uy + C1, A2 = 2 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
            row += (A1 * A2) / (B1 * B2);
        }
        total += row;
    }
    *out = total / ((double)(w - 2 * pad) * (h - 2 * pad));
    return 0;
}
