// VOC07 11-point mAP on the device (include/mnyolo.h: mny_map_eval, mny_eval_pack).
// Replaces utils/eval_mAP.py + utils/iou.py:find_jaccard_overlap of the reference: there an
// O(classes x images x detections) Python loop of 1xn IoU calls; here
//   match  : thread per detection — best ground truth of its image & class (first maximum, fp32 IoU), the object's first
//            claimant by atomicMin over the (stored-order) detection index  = the reference's sequential "already detected" flag
//   flags  : TP / FP per detection + a 40-bit sort key (class | descending score); own bitonic sort of (key, stored index) pairs —
//            the index breaks score ties, which is exactly what a stable sort of the stored order yields
//   ap     : workgroup per class — n_easy, its key segment by binary search, block scan of TP/FP over the sorted run,
//            11 running maxima of precision where recall >= t
// Compiled with -ffp-contract=off: the IoU must round like the CPU tensor ops it replaces.
#include "common.h"

#include <cstring>

namespace mny {
namespace {

constexpr int kApThreads = 1024;
constexpr uint32_t kNoClass = 255;

struct thresholds { float t[11]; };

__device__ __forceinline__ uint32_t class_of(float label, int n_classes) {
    if (!(label >= 1.f && label <= (float)(n_classes - 1))) return 0;
    const uint32_t c = (uint32_t)label;
    return (float)c == label ? c : 0;
}

__device__ __forceinline__ int image_of(const int32_t* off, int n_images, int d) {   // largest i with off[i] <= d
    int lo = 0, hi = n_images;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= d) lo = mid; else hi = mid;
    }
    return lo;
}

// eval_mAP.py:32-63 without the sequential flag: match[d] = object index (candidate TP), -1 = false positive,
// -2 = ignored (difficult object), -3 = label outside 1..n_classes-1
__global__ void map_match_kernel(const float* __restrict__ det_boxes, const float* __restrict__ det_labels, const int32_t* __restrict__ det_off,
                                 const float* __restrict__ true_boxes, const float* __restrict__ true_labels, const float* __restrict__ true_diff,
                                 const int32_t* __restrict__ true_off, int n_images, int D, int n_classes, int* __restrict__ match, int* __restrict__ first) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const float label = det_labels[d];
    if (!class_of(label, n_classes)) { match[d] = -3; return; }
    const int img = image_of(det_off, n_images, d);
    const float4 b = ld4(det_boxes + 4 * (size_t)d);
    const float a1 = (b.z - b.x) * (b.w - b.y);
    float best = 0.f;
    int best_j = -1;
    for (int j = true_off[img], je = true_off[img + 1]; j < je; ++j) {
        if (true_labels[j] != label) continue;
        const float4 g = ld4(true_boxes + 4 * (size_t)j);
        const float iw = fmaxf(fminf(b.z, g.z) - fmaxf(b.x, g.x), 0.f);          // utils/iou.py:8-13
        const float ih = fmaxf(fminf(b.w, g.w) - fmaxf(b.y, g.y), 0.f);
        const float inter = iw * ih;
        const float a2 = (g.z - g.x) * (g.w - g.y);
        const float ov = inter / ((a1 + a2) - inter);                             // utils/iou.py:43-48
        if (best_j < 0 || !(ov <= best)) {                                       // torch.max: first maximum; a NaN wins and ends the scan
            best = ov; best_j = j;
            if (ov != ov) break;
        }
    }
    int m = -1;
    if (best_j >= 0 && best > 0.5f) {
        if (true_diff[best_j] == 0.f) { m = best_j; atomicMin(&first[best_j], d); }
        else m = -2;
    }
    match[d] = m;
}

__global__ void map_flag_kernel(const float* __restrict__ det_labels, const float* __restrict__ det_scores, const int* __restrict__ match,
                                const int* __restrict__ first, int D, int n_classes, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const int m = match[d];
    const uint32_t tp = m >= 0 && first[m] == d;
    const uint32_t fp = m == -1 || (m >= 0 && !tp);
    const uint32_t c = class_of(det_labels[d], n_classes);
    const uint32_t sb = __float_as_uint(det_scores[d]);
    const uint32_t asc = sb ^ ((sb >> 31) ? 0xFFFFFFFFu : 0x80000000u);        // order-preserving map of a float
    keys[d] = ((uint64_t)(c ? c : kNoClass) << 32) | (uint32_t)~asc;            // descending score inside the class
    vals[d] = ((uint32_t)d << 2) | (tp << 1) | fp;
}

// ---- sort of (key, value) pairs, ascending in (key, value) ----------------------------------------------------------------
// value = stored detection index << 2 | flags, so the pair order is total and equals the stable order by key.  Bitonic network
// over n2 = 2^m >= D padded pairs: every run of steps with partner distance < kSortChunk happens inside LDS (one workgroup per
// 4096-pair chunk), the few steps with a longer distance are one global compare-exchange pass each (6 MB for VOC07-test).
constexpr int kSortChunk = 4096;

__device__ __forceinline__ bool pair_gt(uint64_t ka, uint32_t va, uint64_t kb, uint32_t vb) { return ka > kb || (ka == kb && va > vb); }

__global__ void map_sort_pad_kernel(uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int D, int n2) {
    const int i = D + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n2) { keys[i] = ~0ull; vals[i] = ~0u; }
}

// steps j = j_first, j_first/2, ..., 1 of stage k (or, with k_first > 0, ALL stages k = 2 .. k_first) on one chunk
__global__ __launch_bounds__(1024) void map_sort_local_kernel(uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int k_first, int k, int j_first) {
    __shared__ uint64_t sk[kSortChunk];
    __shared__ uint32_t sv[kSortChunk];
    const int base = blockIdx.x * kSortChunk;
    for (int i = threadIdx.x; i < kSortChunk; i += blockDim.x) { sk[i] = keys[base + i]; sv[i] = vals[base + i]; }
    __syncthreads();
    auto steps = [&](int kk, int jf) {
        for (int j = jf; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < kSortChunk; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = ((base + i) & kk) == 0;
                    const uint64_t a = sk[i], b = sk[l];
                    const uint32_t x = sv[i], y = sv[l];
                    if (pair_gt(a, x, b, y) == up) { sk[i] = b; sk[l] = a; sv[i] = y; sv[l] = x; }
                }
            }
            __syncthreads();
        }
    };
    if (k_first > 0) { for (int kk = 2; kk <= k_first; kk <<= 1) steps(kk, kk >> 1); }
    else steps(k, j_first);
    for (int i = threadIdx.x; i < kSortChunk; i += blockDim.x) { keys[base + i] = sk[i]; vals[base + i] = sv[i]; }
}

__global__ void map_sort_global_kernel(uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, int n2, int k, int j) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = i ^ j;
    if (i >= n2 || l <= i) return;
    const bool up = (i & k) == 0;
    const uint64_t a = keys[i], b = keys[l];
    const uint32_t x = vals[i], y = vals[l];
    if (pair_gt(a, x, b, y) == up) { keys[i] = b; keys[l] = a; vals[i] = y; vals[l] = x; }
}

inline int sort_size(int64_t D) { int n2 = kSortChunk; while (n2 < D) n2 <<= 1; return n2; }

int sort_pairs(uint64_t* keys, uint32_t* vals, int D, hipStream_t st) {
    const int n2 = sort_size(D);
    if (n2 > D) map_sort_pad_kernel<<<(int)cdiv(n2 - D, 256), 256, 0, st>>>(keys, vals, D, n2);
    const int chunks = n2 / kSortChunk;
    map_sort_local_kernel<<<chunks, 1024, 0, st>>>(keys, vals, kSortChunk, 0, 0);
    for (int k = 2 * kSortChunk; k <= n2; k <<= 1) {
        for (int j = k >> 1; j >= kSortChunk; j >>= 1) map_sort_global_kernel<<<(int)cdiv(n2, 256), 256, 0, st>>>(keys, vals, n2, k, j);
        map_sort_local_kernel<<<chunks, 1024, 0, st>>>(keys, vals, 0, k, kSortChunk >> 1);
    }
    return check_launch("mny_map sort");
}

__device__ __forceinline__ int lower_bound_key(const uint64_t* keys, int n, uint64_t k) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// eval_mAP.py:101-132 for class c = blockIdx.x + 1
__global__ __launch_bounds__(kApThreads) void map_ap_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, int D,
                                                            const float* __restrict__ true_labels, const float* __restrict__ true_diff, int T, thresholds thr,
                                                            float* __restrict__ ap, float* __restrict__ tp_sum, float* __restrict__ fp_sum, float* __restrict__ prec11) {
    constexpr int NW = kApThreads / kWave;
    __shared__ double s_red[NW];
    __shared__ uint32_t s_scan[NW];
    __shared__ float s_max[NW][11];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x + 1;

    // n_easy_class_objects (:17, :96): sum of (1 - difficulty) over the class's objects, fixed reduction order
    double s = 0.0;
    for (int j = tid; j < T; j += kApThreads)
        if (true_labels[j] == (float)c) s += (double)(1.f - true_diff[j]);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) s_red[wave] = s;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < NW; ++w) tot += s_red[w];
    const float n_easy = (float)tot;

    const int lo = lower_bound_key(keys, D, (uint64_t)c << 32), hi = lower_bound_key(keys, D, (uint64_t)(c + 1) << 32);
    uint32_t ctp0 = 0, cfp0 = 0;                     // counts before this chunk (the packed tp | fp << 16 word is chunk-local)
    float pmax[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) pmax[i] = 0.f;
    for (int base = lo; base < hi; base += kApThreads) {
        const int i = base + tid;
        const uint32_t v = i < hi ? vals[i] : 0u;
        const uint32_t mine = ((v >> 1) & 1u) | ((v & 1u) << 16);               // chunk-local counts fit 16 bits each
        uint32_t inc = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        __syncthreads();                                                       // previous chunk's readers are done
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (int w = 0; w < NW; ++w) {
            const uint32_t t = s_scan[w];
            if (w < wave) before += t;
            total += t;
        }
        inc += before;
        if (i < hi) {
            const float ctp = (float)(ctp0 + (inc & 0xFFFFu)), cfp = (float)(cfp0 + (inc >> 16));
            const float precision = ctp / ((ctp + cfp) + 1e-10f);               // :111-112
            const float recall = ctp / n_easy;                                  // :113 (0/0 -> NaN: never >= t)
#pragma unroll
            for (int k = 0; k < 11; ++k)
                if (recall >= thr.t[k]) pmax[k] = fmaxf(pmax[k], precision);    // :118-123
        }
        ctp0 += total & 0xFFFFu;
        cfp0 += total >> 16;
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        float m = pmax[k];
        for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) s_max[wave][k] = m;
    }
    __syncthreads();
    if (tid == 0) {
        float sum = 0.f;
        for (int k = 0; k < 11; ++k) {
            float m = 0.f;
            for (int w = 0; w < NW; ++w) m = fmaxf(m, s_max[w][k]);
            prec11[(c - 1) * 11 + k] = m;
            sum += m;
        }
        ap[c - 1] = sum / 11.f;                                                 // :125
        tp_sum[c - 1] = (float)ctp0;
        fp_sum[c - 1] = (float)cfp0;
    }
}

__global__ void map_mean_kernel(const float* __restrict__ ap, int n, float* __restrict__ mean_ap) {   // eval_mAP.py:172
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += ap[i];
    mean_ap[0] = s / (float)n;
}

__global__ void eval_pack_kernel(const float* __restrict__ rows, int D, const float* __restrict__ targets, int T, float* __restrict__ det_boxes,
                                 float* __restrict__ det_labels, float* __restrict__ det_scores, float* __restrict__ true_boxes,
                                 float* __restrict__ true_labels, float* __restrict__ true_diff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) {
        const float* r = rows + 7 * (size_t)i;
        st4(det_boxes + 4 * (size_t)i, make_float4(r[0], r[1], r[2], r[3]));     // train.py:381
        det_labels[i] = r[6] + 1.f;                                            // train.py:382
        det_scores[i] = r[4] * r[5];                                           // train.py:383
    }
    if (i < T) {
        const float* t = targets + 5 * (size_t)i;
        const float hw = t[3] / 2.f, hh = t[4] / 2.f;
        st4(true_boxes + 4 * (size_t)i, make_float4(t[1] - hw, t[2] - hh, t[1] + hw, t[2] + hh));   // train.py:371-373
        true_labels[i] = t[0];                                                 // train.py:377
        true_diff[i] = 0.f;                                                    // train.py:378
    }
}

struct map_ws_layout {
    size_t keys, vals, match, first, total;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int map_layout(int64_t D, int64_t T, map_ws_layout& L) {
    size_t o = 0;
    const size_t d = (size_t)(D > 0 ? D : 1), t = (size_t)(T > 0 ? T : 1);
    const size_t n2 = (size_t)sort_size(D);                    // the sort runs in place over the padded power of two
    L.keys = o;  o = align256(o + n2 * 8);
    L.vals = o;  o = align256(o + n2 * 4);
    L.match = o; o = align256(o + d * 4);
    L.first = o; o = align256(o + t * 4);
    L.total = o;
    return 0;
}

}  // namespace
}  // namespace mny

using namespace mny;

extern "C" size_t mny_map_ws_bytes(int64_t D, int64_t T) {
    map_ws_layout L;
    if (D < 0 || T < 0 || D > ((int64_t)1 << 30) || map_layout(D, T, L) != 0) return 0;
    return L.total;
}

extern "C" int mny_map_eval(const float* det_boxes, const float* det_labels, const float* det_scores, const int32_t* det_off,
                            const float* true_boxes, const float* true_labels, const float* true_diff, const int32_t* true_off, int n_images,
                            int64_t D, int64_t T, int n_classes, float* ap, float* tp_sum, float* fp_sum, float* prec11, float* mean_ap, void* ws,
                            void* stream) {
    MNY_REQUIRE(n_classes >= 2 && n_classes <= (int)kNoClass, "mny_map_eval: n_classes %d outside 2..255 (it counts the background entry)", n_classes);
    MNY_REQUIRE(D >= 0 && T >= 0 && n_images >= 0 && D <= ((int64_t)1 << 30) && T <= ((int64_t)1 << 30), "mny_map_eval: bad sizes D=%lld T=%lld images=%d",
                (long long)D, (long long)T, n_images);
    MNY_REQUIRE(ap && tp_sum && fp_sum && prec11 && mean_ap && ws, "mny_map_eval: null output / workspace");
    MNY_REQUIRE(D == 0 || (det_boxes && det_labels && det_scores && det_off && true_off && n_images > 0), "mny_map_eval: null detection input");
    MNY_REQUIRE(T == 0 || (true_boxes && true_labels && true_diff), "mny_map_eval: null ground-truth input");
    hipStream_t st = (hipStream_t)stream;
    map_ws_layout L;
    if (int rc = map_layout(D, T, L)) return rc;
    char* w = (char*)ws;
    uint64_t* keys = (uint64_t*)(w + L.keys);
    uint32_t* vals = (uint32_t*)(w + L.vals);
    int *match = (int*)(w + L.match), *first = (int*)(w + L.first);
    if (D > 0) {
        if (T > 0 && hipMemsetAsync(first, 0x7f, (size_t)T * 4, st) != hipSuccess) { set_error("mny_map_eval: memset failed"); return MNY_EHIP; }
        const int blocks = (int)cdiv(D, 256);
        map_match_kernel<<<blocks, 256, 0, st>>>(det_boxes, det_labels, det_off, true_boxes, true_labels, true_diff, true_off, n_images, (int)D, n_classes,
                                                 match, first);
        map_flag_kernel<<<blocks, 256, 0, st>>>(det_labels, det_scores, match, first, (int)D, n_classes, keys, vals);
        if (int rc = sort_pairs(keys, vals, (int)D, st)) return rc;
    }
    thresholds thr;
    for (int i = 0; i < 11; ++i) thr.t[i] = (float)(0.1 * (double)i);             // torch.arange(0, 1.1, .1): double start + i*step, rounded to fp32
    map_ap_kernel<<<n_classes - 1, kApThreads, 0, st>>>(keys, vals, (int)D, true_labels, true_diff, (int)T, thr, ap, tp_sum, fp_sum, prec11);
    map_mean_kernel<<<1, 1, 0, st>>>(ap, n_classes - 1, mean_ap);
    return check_launch("mny_map_eval");
}

extern "C" int mny_eval_pack(const float* rows, int64_t D, const float* targets, int64_t T, float* det_boxes, float* det_labels, float* det_scores,
                             float* true_boxes, float* true_labels, float* true_diff, void* stream) {
    MNY_REQUIRE(D >= 0 && T >= 0 && D <= ((int64_t)1 << 30) && T <= ((int64_t)1 << 30), "mny_eval_pack: bad sizes");
    MNY_REQUIRE(D == 0 || (rows && det_boxes && det_labels && det_scores), "mny_eval_pack: null detection pointer");
    MNY_REQUIRE(T == 0 || (targets && true_boxes && true_labels && true_diff), "mny_eval_pack: null target pointer");
    const int64_t n = D > T ? D : T;
    if (n == 0) return 0;
    eval_pack_kernel<<<(int)cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(rows, (int)D, targets, (int)T, det_boxes, det_labels, det_scores, true_boxes,
                                                                        true_labels, true_diff);
    return check_launch("mny_eval_pack");
}
