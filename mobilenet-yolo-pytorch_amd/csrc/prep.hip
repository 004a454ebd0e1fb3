// Device-side batch input preparation (include/mnyolo.h: mny_prep_batch).
// Replaces the image half of folder2lmdb.py:223-256 (collate_fn): per image transforms.Resize(size, BILINEAR) on the decoded
// PIL image (= Pillow's ImagingResample: antialiased triangle filter, 22-bit fixed-point taps, horizontal pass then vertical
// pass, each rounded to uint8), ToTensor (/255), Normalize ((x-mean)/std), stack — for a whole batch of differently sized
// RGB uint8 images in three launches, writing the NCHW fp32 batch the stem kernel reads.
//   coef  : per image and axis, the tap window + fixed-point taps of every output index (fp64, the library's operation order)
//   hpass : [h,w,3] -> [h,out_w,3] uint8
//   vpass : [h,out_w,3] -> [3,out_h,out_w] fp32, normalised
// Integer work: bit-exact against Pillow.  Compiled with -ffp-contract=off (the taps must round like the host library's).
#include "common.h"

namespace mny {
namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

struct prep_layout {
    int ks_h, ks_v;                       // tap-row strides (max taps per output index)
    size_t bh, kh, bv, kv, tmp, per_image, status, total;
};

inline int max_taps(int in_max, int out) {
    double fs = (double)in_max / (double)out;
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(fs) * 2 + 1;
}

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

prep_layout make_layout(int N, int max_in_h, int max_in_w, int out_h, int out_w) {
    prep_layout L;
    L.ks_h = max_taps(max_in_w, out_w);
    L.ks_v = max_taps(max_in_h, out_h);
    size_t o = 0;
    L.bh = o; o = al(o + (size_t)out_w * 2 * 4);
    L.kh = o; o = al(o + (size_t)out_w * L.ks_h * 4);
    L.bv = o; o = al(o + (size_t)out_h * 2 * 4);
    L.kv = o; o = al(o + (size_t)out_h * L.ks_v * 4);
    L.tmp = o; o = al(o + (size_t)max_in_h * out_w * 3);
    L.per_image = o;
    L.status = 0;
    L.total = 256 + L.per_image * (size_t)N;
    return L;
}

// Resample.c precompute_coeffs (whole-axis box) + normalize_coeffs_8bpc, bilinear filter (support 1)
__global__ void prep_coef_kernel(const mny_image_desc* __restrict__ desc, int out_h, int out_w, int max_in_h, int max_in_w, prep_layout L, char* __restrict__ ws) {
    const int n = blockIdx.x, axis = blockIdx.y;                                // axis 0: horizontal (w), 1: vertical (h)
    const mny_image_desc d = desc[n];
    if (d.h < 1 || d.w < 1 || d.h > max_in_h || d.w > max_in_w) {
        if (threadIdx.x == 0 && axis == 0) atomicCAS((int*)ws, 0, n + 1);
        return;
    }
    char* base = ws + 256 + L.per_image * (size_t)n;
    const int in_size = axis ? d.h : d.w, out_size = axis ? out_h : out_w, stride = axis ? L.ks_v : L.ks_h;
    int* bounds = (int*)(base + (axis ? L.bv : L.bh));
    int* kk = (int*)(base + (axis ? L.kv : L.kh));
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
    for (int xx = threadIdx.x; xx < out_size; xx += blockDim.x) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            ww += a < 1.0 ? 1.0 - a : 0.0;
        }
        int* k = kk + (size_t)xx * stride;
        for (int x = 0; x < stride; ++x) {
            int v = 0;
            if (x < xmax) {
                double a = (x + xmin - center + 0.5) * ss;
                if (a < 0.0) a = -a;
                double w = a < 1.0 ? 1.0 - a : 0.0;
                if (ww != 0.0) w /= ww;
                v = (int)(0.5 + w * (double)(1 << kPrecisionBits));
            }
            k[x] = v;
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
}

__device__ __forceinline__ int clip8(int v) {
    v >>= kPrecisionBits;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// ImagingResampleHorizontal_8bpc: thread = one pixel of the [h, out_w] intermediate image
__global__ __launch_bounds__(256) void prep_hpass_kernel(const uint8_t* __restrict__ src, const mny_image_desc* __restrict__ desc, int out_w, int max_in_h,
                                                         int max_in_w, prep_layout L, char* __restrict__ ws) {
    const int n = blockIdx.y;
    const mny_image_desc d = desc[n];
    if (d.h < 1 || d.w < 1 || d.h > max_in_h || d.w > max_in_w) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.h * out_w) return;
    const int y = i / out_w, x = i - y * out_w;
    char* base = ws + 256 + L.per_image * (size_t)n;
    const int* bounds = (const int*)(base + L.bh);
    const int* k = (const int*)(base + L.kh) + (size_t)x * L.ks_h;
    const int lo = bounds[2 * x], cnt = bounds[2 * x + 1];
    const uint8_t* p = src + d.offset + ((size_t)y * d.w + lo) * 3;
    int r = 1 << (kPrecisionBits - 1), g = r, b = r;
    for (int t = 0; t < cnt; ++t) {
        const int c = k[t];
        r += p[3 * t] * c; g += p[3 * t + 1] * c; b += p[3 * t + 2] * c;
    }
    uint8_t* o = (uint8_t*)(base + L.tmp) + (size_t)i * 3;
    o[0] = (uint8_t)clip8(r); o[1] = (uint8_t)clip8(g); o[2] = (uint8_t)clip8(b);
}

// ImagingResampleVertical_8bpc + ToTensor + Normalize: thread = one output pixel, writes the three NCHW planes
__global__ __launch_bounds__(256) void prep_vpass_kernel(const mny_image_desc* __restrict__ desc, int out_h, int out_w, int max_in_h, int max_in_w,
                                                         prep_layout L, const char* __restrict__ ws, float3 mean, float3 stdv, float* __restrict__ out) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= out_h * out_w) return;
    const mny_image_desc d = desc[n];
    float* o = out + (size_t)n * 3 * out_h * out_w + i;
    const size_t plane = (size_t)out_h * out_w;
    if (d.h < 1 || d.w < 1 || d.h > max_in_h || d.w > max_in_w) { o[0] = o[plane] = o[2 * plane] = 0.f; return; }
    const int y = i / out_w, x = i - y * out_w;
    const char* base = ws + 256 + L.per_image * (size_t)n;
    const int* bounds = (const int*)(base + L.bv);
    const int* k = (const int*)(base + L.kv) + (size_t)y * L.ks_v;
    const int lo = bounds[2 * y], cnt = bounds[2 * y + 1];
    const uint8_t* p = (const uint8_t*)(base + L.tmp) + ((size_t)lo * out_w + x) * 3;
    int r = 1 << (kPrecisionBits - 1), g = r, b = r;
    for (int t = 0; t < cnt; ++t) {
        const int c = k[t];
        const uint8_t* q = p + (size_t)t * out_w * 3;
        r += q[0] * c; g += q[1] * c; b += q[2] * c;
    }
    o[0] = ((float)clip8(r) / 255.f - mean.x) / stdv.x;                          // ToTensor .div(255); Normalize .sub_(mean).div_(std)
    o[plane] = ((float)clip8(g) / 255.f - mean.y) / stdv.y;
    o[2 * plane] = ((float)clip8(b) / 255.f - mean.z) / stdv.z;
}

}  // namespace
}  // namespace mny

using namespace mny;

extern "C" size_t mny_prep_ws_bytes(int N, int max_in_h, int max_in_w, int out_h, int out_w) {
    if (N < 1 || max_in_h < 1 || max_in_w < 1 || out_h < 1 || out_w < 1) return 0;
    return make_layout(N, max_in_h, max_in_w, out_h, out_w).total;
}

extern "C" int mny_prep_batch(const uint8_t* src, const mny_image_desc* desc, int N, int max_in_h, int max_in_w, int out_h, int out_w, const float* mean3,
                              const float* std3, float* out, void* ws, void* stream) {
    MNY_REQUIRE(src && desc && mean3 && std3 && out && ws, "mny_prep_batch: null pointer");
    MNY_REQUIRE(N >= 1 && max_in_h >= 1 && max_in_w >= 1 && out_h >= 1 && out_w >= 1, "mny_prep_batch: bad sizes N=%d in<=%dx%d out=%dx%d", N, max_in_h,
                max_in_w, out_h, out_w);
    MNY_REQUIRE((int64_t)max_in_h * out_w < ((int64_t)1 << 31) / 3 && N <= 65535, "mny_prep_batch: image too large / batch > 65535");
    MNY_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "mny_prep_batch: std must be non-zero");
    hipStream_t st = (hipStream_t)stream;
    const prep_layout L = make_layout(N, max_in_h, max_in_w, out_h, out_w);
    if (hipMemsetAsync(ws, 0, 256, st) != hipSuccess) { set_error("mny_prep_batch: memset failed"); return MNY_EHIP; }
    prep_coef_kernel<<<dim3(N, 2), 256, 0, st>>>(desc, out_h, out_w, max_in_h, max_in_w, L, (char*)ws);
    prep_hpass_kernel<<<dim3((unsigned)cdiv((int64_t)max_in_h * out_w, 256), N), 256, 0, st>>>(src, desc, out_w, max_in_h, max_in_w, L, (char*)ws);
    prep_vpass_kernel<<<dim3((unsigned)cdiv((int64_t)out_h * out_w, 256), N), 256, 0, st>>>(desc, out_h, out_w, max_in_h, max_in_w, L, (const char*)ws,
                                                                                           make_float3(mean3[0], mean3[1], mean3[2]),
                                                                                           make_float3(std3[0], std3[1], std3[2]), out);
    return check_launch("mny_prep_batch");
}
