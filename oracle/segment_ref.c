/*
 * oracle/segment_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the motion-segmentation head the reference
 * runs on the flow field, /root/reference/optical_flow_seg.py:
 *   task_results           :253-320   mag of the (negated, float64) flow crop
 *   process_flow_region    :322-357   mask = mag > SEG_TH; 5 x (dilate, erode) with a
 *                                     10x10 MORPH_ELLIPSE element; threshold(1) -> 0/255
 *   baseline path          :503-537   the same chain on the full frame
 * (the HSV / gray / `binary` values computed at :327-343 never reach the result).
 *
 * The arithmetic of getStructuringElement / dilate / erode / cartToPolar lives in
 * opencv-python (requirements.txt:1), absent here; restated from the published
 * definitions (cv::getStructuringElement, cv::dilate, cv::erode: anchor at ksize/2,
 * dst(x,y) = max|min over element(i,j) != 0 of src(x + j - ax, y + i - ay), pixels outside
 * the image do not take part).
 *
 * PINNED by the reference's own recorded result: the "Final Motion Segmentation" panel
 * of demo.ipynb (tests/golden/demo/panel_seg_mask.png).  With the element applied
 * un-reflected in both operations, as above, the mask of this restatement overlaps the
 * authors' cv2 mask with IoU 0.997; reflecting the element in dilate (the textbook closing)
 * gives 0.911 (tests/test_segment.py).  Bit level: unpinned.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* cv::getStructuringElement(MORPH_ELLIPSE, Size(kw, kh)): row i holds ones in [c - dx, c + dx + 1) with
 * dx = round(c * sqrt((r^2 - dy^2) / r^2)), r = kh/2, c = kw/2, dy = i - r.  shape: 0 rect, 1 cross, 2 ellipse. */
int nsof_ref_structuring_element(int shape, int kw, int kh, uint8_t* out)
{
    if (kw < 1 || kh < 1 || shape < 0 || shape > 2) return -1;
    int r = kh / 2, c = kw / 2;
    double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < kh; i++) {
        int j1 = 0, j2 = 0;
        if (shape == 0 || (shape == 1 && i == r)) {
            j2 = kw;
        } else if (shape == 1) {
            j1 = c;
            j2 = c + 1;
        } else {
            int dy = i - r;
            if (abs(dy) <= r) {
                int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
                j1 = c - dx > 0 ? c - dx : 0;
                j2 = c + dx + 1 < kw ? c + dx + 1 : kw;
            }
        }
        for (int j = 0; j < kw; j++) out[i * kw + j] = (uint8_t)(j >= j1 && j < j2);
    }
    return 0;
}

/* op 0 = erode (min), 1 = dilate (max); anchor (ax, ay), -1 = centre.  dst must not alias src. */
int nsof_ref_morph_u8(int op, const uint8_t* src, ptrdiff_t sstride, int w, int h, const uint8_t* elem, int kw, int kh,
                      int ax, int ay, uint8_t* dst, ptrdiff_t dstride)
{
    if (w < 0 || h < 0 || kw < 1 || kh < 1) return -1;
    if (ax < 0) ax = kw / 2;
    if (ay < 0) ay = kh / 2;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = op ? 0 : 255;
            for (int i = 0; i < kh; i++) {
                int yy = y + i - ay;
                if (yy < 0 || yy >= h) continue;
                for (int j = 0; j < kw; j++) {
                    int xx = x + j - ax;
                    if (!elem[i * kw + j] || xx < 0 || xx >= w) continue;
                    int s = src[yy * sstride + xx];
                    v = op ? (s > v ? s : v) : (s < v ? s : v);
                }
            }
            dst[y * dstride + x] = (uint8_t)v;
        }
    return 0;
}

/* process_flow_region on a float32 flow crop [h][w][2] (row stride in floats).  The reference negates the flow
 * first (seg.py:461), which does not change the magnitude; cartToPolar sees float64 there, so the magnitude is
 * sqrt(x*x + y*y) in double.  mask = 255 where mag > thresh; `iters` x (dilate, erode); result 255 where mask > 1. */
int nsof_ref_motion_mask(const float* flow, ptrdiff_t fstride, int w, int h, double thresh, int ksize, int iters,
                         uint8_t* out, ptrdiff_t ostride)
{
    if (w < 0 || h < 0 || ksize < 1 || iters < 0) return -1;
    size_t n = (size_t)w * h;
    uint8_t* a = (uint8_t*)malloc(n ? n : 1);
    uint8_t* b = (uint8_t*)malloc(n ? n : 1);
    uint8_t* el = (uint8_t*)malloc((size_t)ksize * ksize);
    if (!a || !b || !el) { free(a); free(b); free(el); return -2; }
    nsof_ref_structuring_element(2, ksize, ksize, el);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double u = flow[y * fstride + 2 * x], v = flow[y * fstride + 2 * x + 1];
            a[(size_t)y * w + x] = sqrt(u * u + v * v) > thresh ? 255 : 0;
        }
    for (int k = 0; k < iters; k++) {
        nsof_ref_morph_u8(1, a, w, w, h, el, ksize, ksize, -1, -1, b, w);
        nsof_ref_morph_u8(0, b, w, w, h, el, ksize, ksize, -1, -1, a, w);
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[y * ostride + x] = a[(size_t)y * w + x] > 1 ? 255 : 0;
    free(a); free(b); free(el);
    return 0;
}
