/* Oracle (test infrastructure, NOT product): CPU restatement of the NMS path.
 *
 * 1. oracle_nms_kernel  — torchvision.ops.nms CPU semantics.  The reference calls
 *    it at utils/box.py:28 but does not vendor it (requirements.txt:3 lists
 *    `torchvision` unpinned, not installed in this image): PARITY UNPINNED at
 *    this boundary.  Restated from torchvision's published CPU algorithm:
 *      order = stable descending sort of scores
 *      areas = (x2-x1)*(y2-y1)                         (float)
 *      greedy over order: keep i; for every later, not yet suppressed j:
 *        w = max(0, min(x2i,x2j) - max(x1i,x1j)); h likewise
 *        inter = w*h; ovr = inter / (area_i + area_j - inter)      (float)
 *        suppress j iff (double)ovr > (double)thr
 *      returns kept ORIGINAL indices in descending-score order.
 * 2. oracle_nms_per_class — the per-image, per-class driver of utils/box.py:11-31:
 *    rows [n,7] = (x1,y1,x2,y2,conf,cls_score,cls_idx); for class 0..C-1 select
 *    rows with row[6]==class (:21), score = row[5]*row[4] (:27), NMS at 0.45
 *    (:28), gather and concatenate in class order (:29).
 *
 * Plain C, single thread (torchvision's CPU kernel is single-threaded too).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* stable descending merge sort of indices by key */
static void merge_sort_desc(const float *key, int64_t *idx, int64_t *tmp, int64_t n) {
    for (int64_t width = 1; width < n; width *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * width) {
            int64_t mid = lo + width < n ? lo + width : n;
            int64_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            int64_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                /* take from the right run only if strictly greater: stability */
                if (key[idx[b]] > key[idx[a]]) tmp[o++] = idx[b++];
                else tmp[o++] = idx[a++];
            }
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
    }
}

/* boxes [n,4] row-major with row stride `stride` floats; returns #kept, keep[] filled */
int64_t oracle_nms_kernel(const float *boxes, int64_t stride, const float *scores,
                          int64_t n, double thr, int64_t *keep) {
    if (n <= 0) return 0;
    int64_t *order = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    int64_t *tmp = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    float *area = (float *)malloc((size_t)n * sizeof(float));
    unsigned char *dead = (unsigned char *)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; ++i) {
        order[i] = i;
        const float *b = boxes + i * stride;
        area[i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    merge_sort_desc(scores, order, tmp, n);
    int64_t kept = 0;
    for (int64_t oi = 0; oi < n; ++oi) {
        int64_t i = order[oi];
        if (dead[i]) continue;
        keep[kept++] = i;
        const float *bi = boxes + i * stride;
        float ix1 = bi[0], iy1 = bi[1], ix2 = bi[2], iy2 = bi[3], ia = area[i];
        for (int64_t oj = oi + 1; oj < n; ++oj) {
            int64_t j = order[oj];
            if (dead[j]) continue;
            const float *bj = boxes + j * stride;
            float xx1 = ix1 > bj[0] ? ix1 : bj[0];
            float yy1 = iy1 > bj[1] ? iy1 : bj[1];
            float xx2 = ix2 < bj[2] ? ix2 : bj[2];
            float yy2 = iy2 < bj[3] ? iy2 : bj[3];
            float w = xx2 - xx1; if (!(w > 0.0f)) w = 0.0f;
            float h = yy2 - yy1; if (!(h > 0.0f)) h = 0.0f;
            float inter = w * h;
            float ovr = inter / (ia + area[j] - inter);
            if ((double)ovr > thr) dead[j] = 1;
        }
    }
    free(order); free(tmp); free(area); free(dead);
    return kept;
}

/* rows [n,7]; out_idx receives indices into rows in output order; returns #kept */
int64_t oracle_nms_per_class(const float *rows, int64_t n, int32_t num_classes, double thr,
                             int64_t *out_idx) {
    if (n <= 0) return 0;
    int64_t *sel = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    int64_t *keep = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    float *bx = (float *)malloc((size_t)n * 4 * sizeof(float));
    float *sc = (float *)malloc((size_t)n * sizeof(float));
    int64_t total = 0;
    for (int32_t c = 0; c < num_classes; ++c) {
        int64_t m = 0;
        for (int64_t i = 0; i < n; ++i) {
            const float *r = rows + i * 7;
            if (r[6] == (float)c) {
                sel[m] = i;
                memcpy(bx + m * 4, r, 4 * sizeof(float));
                sc[m] = r[5] * r[4];
                ++m;
            }
        }
        int64_t k = oracle_nms_kernel(bx, 4, sc, m, thr, keep);
        for (int64_t q = 0; q < k; ++q) out_idx[total++] = sel[keep[q]];
    }
    free(sel); free(keep); free(bx); free(sc);
    return total;
}
