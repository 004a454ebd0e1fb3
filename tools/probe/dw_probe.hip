// Stand-alone probe: which part of the 3x3 stride-1 depthwise forward bounds it on MI355X?  (build: hipcc --offload-arch=gfx950 -O3)
// usage: dw_probe N H C  -> one line per variant: ms, algorithmic GB/s (in + out bytes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4nt(const float* p) { f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void st4nt(float* p, float4 v) { f4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(p)); }
// flat references: F 0 copy, 1 copy nt, 2 read only, 3 write only
template <int F>
__global__ __launch_bounds__(256) void flat(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ parts, long n4) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        if (F == 0) st4(y + i * 4, ld4(x + i * 4));
        else if (F == 1) st4nt(y + i * 4, ld4nt(x + i * 4));
        else if (F == 2) { float4 v = ld4(x + i * 4); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        else st4(y + i * 4, make_float4(1.f, 2.f, 3.f, (float)i));
    }
    if (F == 2) st4(parts + ((long)blockIdx.x * 256 + threadIdx.x) * 4, acc);
}

struct G { int N, H, W, C, TH, nHS, cgb, ppb, cg_total; long nstrips; int xcd; };

// V: 6 full + nt stores, 7 full + nt loads and stores;  V: 0 full, 1 centre column load only, 2 no store, 3 copy only (centre load + store), 5 full with loads of row r+1 issued before row r is consumed
template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
        const float* __restrict__ w, float* __restrict__ y, float* __restrict__ parts, G g) {
    const int tid = threadIdx.x, cgl = tid % g.cgb, pix = tid / g.cgb;
    const int cg = blockIdx.y * g.cgb + cgl;
    if (cg >= g.cg_total || pix >= g.ppb) return;
    const int c = cg * 4;
    const int gx = gridDim.x;
    const int lb = (g.xcd && (gx & 7) == 0) ? (int)(blockIdx.x & 7) * (gx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const float4 s4 = ld4(sc + c), h4 = ld4(sh + c);
    float4 wt[9];
    for (int t = 0; t < 9; ++t) wt[t] = make_float4(w[(c) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
    float4 acc = make_float4(0, 0, 0, 0);
    const long pitch = (long)g.W * g.C;
    auto xf = [&](float4 v) {
        float4 r;
        r.x = __builtin_amdgcn_fmed3f(fmaf(v.x, s4.x, h4.x), 0.f, 6.f); r.y = __builtin_amdgcn_fmed3f(fmaf(v.y, s4.y, h4.y), 0.f, 6.f);
        r.z = __builtin_amdgcn_fmed3f(fmaf(v.z, s4.z, h4.z), 0.f, 6.f); r.w = __builtin_amdgcn_fmed3f(fmaf(v.w, s4.w, h4.w), 0.f, 6.f);
        return r;
    };
    auto fma4 = [&](float4& o, float4 a, float4 b) { o.x = fmaf(a.x, b.x, o.x); o.y = fmaf(a.y, b.y, o.y); o.z = fmaf(a.z, b.z, o.z); o.w = fmaf(a.w, b.w, o.w); };
    for (long strip = (long)lb * g.ppb + pix; strip < g.nstrips; strip += (long)gx * g.ppb) {
        const int wo = (int)(strip % g.W), hs = (int)((strip / g.W) % g.nHS), n = (int)(strip / ((long)g.W * g.nHS));
        const int ho0 = hs * g.TH, ho1 = min(ho0 + g.TH, g.H);
        const int wl = max(wo - 1, 0), wr = min(wo + 1, g.W - 1);
        const float* xn = x + (long)n * g.H * pitch + c;
        float* yo = y + ((long)n * g.H + ho0) * pitch + (long)wo * g.C + c;
        float4 r[3][3];
        auto row = [&](int hi, float4 (&o)[3]) {
            const int hc = min(max(hi, 0), g.H - 1);
            const float* p = xn + (long)hc * pitch;
            if (V == 1 || V == 3) { o[1] = ld4(p + (long)wo * g.C); o[0] = o[1]; o[2] = o[1]; }
            else if (V == 7) { o[0] = ld4nt(p + (long)wl * g.C); o[1] = ld4nt(p + (long)wo * g.C); o[2] = ld4nt(p + (long)wr * g.C); }
            else { o[0] = ld4(p + (long)wl * g.C); o[1] = ld4(p + (long)wo * g.C); o[2] = ld4(p + (long)wr * g.C); }
        };
        if (V == 3) {
            for (int ho = ho0; ho < ho1; ++ho) { row(ho, r[0]); st4(yo, r[0][1]); yo += pitch; }
            continue;
        }
        row(ho0 - 1, r[0]); row(ho0, r[1]);
        for (int q = 0; q < 3; ++q) { r[0][q] = xf(r[0][q]); r[1][q] = xf(r[1][q]); }
        for (int ho = ho0; ho < ho1; ++ho) {
            row(ho + 1, r[2]);
            for (int q = 0; q < 3; ++q) r[2][q] = xf(r[2][q]);
            float4 o = make_float4(0, 0, 0, 0);
            for (int a = 0; a < 3; ++a) for (int q = 0; q < 3; ++q) fma4(o, r[a][q], wt[a * 3 + q]);
            if (V == 6 || V == 7) st4nt(yo, o); else if (V != 2) st4(yo, o);
            yo += pitch;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            for (int q = 0; q < 3; ++q) { r[0][q] = r[1][q]; r[1][q] = r[2][q]; }
        }
    }
    if (V != 3) { float* d = parts + ((long)blockIdx.x * gridDim.y + blockIdx.y) * 1024 + tid * 4; st4(d, acc); }
}

// prefetch variant: D input rows in flight per thread (raw loads issued D rows ahead of their use), WPE waves per SIMD
template <int D, int WPE, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void kpf(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
        const float* __restrict__ w, float* __restrict__ y, float* __restrict__ parts, G g) {
    const int tid = threadIdx.x, cgl = tid % g.cgb, pix = tid / g.cgb;
    const int cg = blockIdx.y * g.cgb + cgl;
    if (cg >= g.cg_total || pix >= g.ppb) return;
    const int c = cg * 4;
    const int gx = gridDim.x;
    const int lb = (int)blockIdx.x;
    const float4 s4 = ld4(sc + c), h4 = ld4(sh + c);
    float4 wt[9];
    for (int t = 0; t < 9; ++t) wt[t] = make_float4(w[(c) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
    float4 acc = make_float4(0, 0, 0, 0);
    const long pitch = (long)g.W * g.C;
    auto xf = [&](float4 v) {
        float4 r;
        r.x = __builtin_amdgcn_fmed3f(fmaf(v.x, s4.x, h4.x), 0.f, 6.f); r.y = __builtin_amdgcn_fmed3f(fmaf(v.y, s4.y, h4.y), 0.f, 6.f);
        r.z = __builtin_amdgcn_fmed3f(fmaf(v.z, s4.z, h4.z), 0.f, 6.f); r.w = __builtin_amdgcn_fmed3f(fmaf(v.w, s4.w, h4.w), 0.f, 6.f);
        return r;
    };
    auto fma4 = [&](float4& o, float4 a, float4 b) { o.x = fmaf(a.x, b.x, o.x); o.y = fmaf(a.y, b.y, o.y); o.z = fmaf(a.z, b.z, o.z); o.w = fmaf(a.w, b.w, o.w); };
    for (long strip = (long)lb * g.ppb + pix; strip < g.nstrips; strip += (long)gx * g.ppb) {
        const int wo = (int)(strip % g.W), hs = (int)((strip / g.W) % g.nHS), n = (int)(strip / ((long)g.W * g.nHS));
        const int ho0 = hs * g.TH, ho1 = min(ho0 + g.TH, g.H);
        const int wl = max(wo - 1, 0), wr = min(wo + 1, g.W - 1);
        const float* xn = x + (long)n * g.H * pitch + c;
        float* yo = y + ((long)n * g.H + ho0) * pitch + (long)wo * g.C + c;
        float4 r[2][3], q[D][3];
        auto row = [&](int hi, float4 (&o)[3]) {
            const int hc = min(max(hi, 0), g.H - 1);
            const float* p = xn + (long)hc * pitch;
            o[0] = ld4(p + (long)wl * g.C); o[1] = ld4(p + (long)wo * g.C); o[2] = ld4(p + (long)wr * g.C);
        };
        row(ho0 - 1, r[0]); row(ho0, r[1]);
#pragma unroll
        for (int d = 0; d < D; ++d) row(min(ho0 + 1 + d, ho1), q[d]);
        for (int k = 0; k < 3; ++k) { r[0][k] = xf(r[0][k]); r[1][k] = xf(r[1][k]); }
        for (int ho = ho0; ho < ho1; ho += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (ho + d < ho1) {
                    float4 nw[3];
                    for (int k = 0; k < 3; ++k) nw[k] = xf(q[d][k]);
                    row(min(ho + d + 1 + D, ho1), q[d]);
                    float4 o = make_float4(0, 0, 0, 0);
                    for (int k = 0; k < 3; ++k) { fma4(o, r[0][k], wt[k]); fma4(o, r[1][k], wt[3 + k]); fma4(o, nw[k], wt[6 + k]); }
                    if (NT) st4nt(yo, o); else st4(yo, o);
                    yo += pitch;
                    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
                    for (int k = 0; k < 3; ++k) { r[0][k] = r[1][k]; r[1][k] = nw[k]; }
                }
            }
        }
    }
    float* d = parts + ((long)blockIdx.x * gridDim.y + blockIdx.y) * 1024 + tid * 4; st4(d, acc);
}

template <int D, int WPE, bool NT>
static void runpf(const char* name, const float* x, const float* sc, const float* sh, const float* w, float* y, float* parts, G g, int gx, int chunks, int threads) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((kpf<D, WPE, NT>), dim3(gx, chunks), dim3(threads), 0, 0, x, sc, sh, w, y, parts, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kpf<D, WPE, NT>), dim3(gx, chunks), dim3(threads), 0, 0, x, sc, sh, w, y, parts, g);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double bytes = 2.0 * g.N * g.H * g.W * g.C * 4;
    printf("  %-34s gx %4d %7.3f ms  %7.1f GB/s (in+out)\n", name, gx, ms, bytes / ms / 1e6);
}

template <int V>
static void run(const char* name, const float* x, const float* sc, const float* sh, const float* w, float* y, float* parts, G g, int gx, int chunks, int threads) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<V>, dim3(gx, chunks), dim3(threads), 0, 0, x, sc, sh, w, y, parts, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<V>, dim3(gx, chunks), dim3(threads), 0, 0, x, sc, sh, w, y, parts, g);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double bytes = 2.0 * g.N * g.H * g.W * g.C * 4;
    printf("  %-34s %7.3f ms  %7.1f GB/s (in+out)\n", name, ms, bytes / ms / 1e6);
}

int main(int argc, char** argv) {
    const int N = atoi(argv[1]), H = atoi(argv[2]), C = atoi(argv[3]);
    const int maxcgb = argc > 4 ? atoi(argv[4]) : 64, TH = argc > 5 ? atoi(argv[5]) : 16, cap_blocks = argc > 6 ? atoi(argv[6]) : 1024;
    G g; g.N = N; g.H = H; g.W = H; g.C = C;
    g.cg_total = C / 4;
    const int chunks = (g.cg_total + maxcgb - 1) / maxcgb;
    g.cgb = (g.cg_total + chunks - 1) / chunks;
    g.ppb = 256 / g.cgb; if (g.ppb < 1) g.ppb = 1;
    const int threads = g.cgb * g.ppb;
    const int ns = (H + TH - 1) / TH; g.TH = (H + ns - 1) / ns; g.nHS = (H + g.TH - 1) / g.TH;
    g.nstrips = (long)N * H * g.nHS;
    long want = (g.nstrips + g.ppb - 1) / g.ppb; want = (want + 7) & ~7L;
    int cap = (cap_blocks / chunks) & ~7;
    const int gx = (int)(want < cap ? want : cap);
    const size_t n = (size_t)N * H * H * C;
    float *x, *y, *sc, *sh, *w, *parts;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&sc, C * 4)); CK(hipMalloc(&sh, C * 4)); CK(hipMalloc(&w, C * 9 * 4));
    CK(hipMalloc(&parts, (size_t)4096 * 16 * 1024 * 4));
    std::vector<float> hx(n); for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> hc(C * 9, 0.3f); CK(hipMemcpy(sc, hc.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sh, hc.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hc.data(), C * 9 * 4, hipMemcpyHostToDevice));
    printf("N%d H%d C%d: cgb %d ppb %d threads %d chunks %d TH %d gx %d (%.0f MB in)\n", N, H, C, g.cgb, g.ppb, threads, chunks, g.TH, gx, n * 4 / 1e6);
    for (int xcd = 0; xcd < 2; ++xcd) {
        g.xcd = xcd;
        printf(" xcd-contiguous=%d\n", xcd);
        run<0>("full", x, sc, sh, w, y, parts, g, gx, chunks, threads);
        run<1>("centre-column load only", x, sc, sh, w, y, parts, g, gx, chunks, threads);
        run<2>("no store", x, sc, sh, w, y, parts, g, gx, chunks, threads);
        run<3>("copy (centre load + store)", x, sc, sh, w, y, parts, g, gx, chunks, threads);
    }
    g.xcd = 0;
    run<6>("full, nt stores", x, sc, sh, w, y, parts, g, gx, chunks, threads);
    run<7>("full, nt loads + stores", x, sc, sh, w, y, parts, g, gx, chunks, threads);
    {
        const int g4 = gx, g3 = ((768 / chunks) & ~7) < gx ? ((768 / chunks) & ~7) : gx, g2 = ((512 / chunks) & ~7) < gx ? ((512 / chunks) & ~7) : gx;
        runpf<1, 4, false>("prefetch 1 row, 4 waves/SIMD", x, sc, sh, w, y, parts, g, g4, chunks, threads);
        runpf<2, 4, false>("prefetch 2 rows, 4 waves/SIMD", x, sc, sh, w, y, parts, g, g4, chunks, threads);
        runpf<2, 3, false>("prefetch 2 rows, 3 waves/SIMD", x, sc, sh, w, y, parts, g, g3, chunks, threads);
        runpf<3, 3, false>("prefetch 3 rows, 3 waves/SIMD", x, sc, sh, w, y, parts, g, g3, chunks, threads);
        runpf<3, 3, true>("prefetch 3 rows, 3 w/S, nt store", x, sc, sh, w, y, parts, g, g3, chunks, threads);
        runpf<4, 2, false>("prefetch 4 rows, 2 waves/SIMD", x, sc, sh, w, y, parts, g, g2, chunks, threads);
        runpf<6, 2, true>("prefetch 6 rows, 2 w/S, nt store", x, sc, sh, w, y, parts, g, g2, chunks, threads);
    }
    // references: flat float4 streams over the same bytes
    const char* fn[4] = {"flat copy", "flat copy nt", "flat read only (x1 bytes)", "flat write only (x1 bytes)"};
    for (int f = 0; f < 4; ++f)
        for (int blocks = 1024; blocks <= 4096; blocks *= 4) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            auto launch = [&]() {
                if (f == 0) hipLaunchKernelGGL(flat<0>, dim3(blocks), dim3(256), 0, 0, x, y, parts, (long)(n / 4));
                else if (f == 1) hipLaunchKernelGGL(flat<1>, dim3(blocks), dim3(256), 0, 0, x, y, parts, (long)(n / 4));
                else if (f == 2) hipLaunchKernelGGL(flat<2>, dim3(blocks), dim3(256), 0, 0, x, y, parts, (long)(n / 4));
                else hipLaunchKernelGGL(flat<3>, dim3(blocks), dim3(256), 0, 0, x, y, parts, (long)(n / 4));
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
            printf("  %-28s %4d blocks %7.3f ms  %7.1f GB/s\n", fn[f], blocks, ms, (f < 2 ? 2.0 : 1.0) * n * 4 / ms / 1e6);
        }
    return 0;
}
