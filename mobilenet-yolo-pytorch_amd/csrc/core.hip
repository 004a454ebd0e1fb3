// Error plumbing + version for libmnyolo.
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace mny {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MNY_EHIP;
    }
    return MNY_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: the opt-in is remembered per (kernel, device) so a
// process that builds plans on a second GPU sets it there too, under a mutex (entry points may be called from several threads).
bool allow_lds(const void* kernel, size_t bytes) {
    static std::map<std::pair<const void*, int>, size_t> done;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(kernel, dev);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    done[key] = bytes;
    return true;
}

__global__ __launch_bounds__(256) void reduce_parts_f64_kernel(const float* __restrict__ parts, int nparts, int n, float* __restrict__ out) {
    __shared__ double red[8][32];
    const int ol = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + ol;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < n) {
        int p = slice;
        for (; p + 24 < nparts; p += 32) {                       // four rows in flight per thread
            s0 += (double)parts[(int64_t)p * n + i];        s1 += (double)parts[(int64_t)(p + 8) * n + i];
            s2 += (double)parts[(int64_t)(p + 16) * n + i]; s3 += (double)parts[(int64_t)(p + 24) * n + i];
        }
        for (; p < nparts; p += 8) s0 += (double)parts[(int64_t)p * n + i];
    }
    red[slice][ol] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && i < n) {
        double s = 0.0;
        for (int k = 0; k < 8; ++k) s += red[k][ol];
        out[i] = (float)s;
    }
}

// every deferred partial combine of a backward segment in ONE launch (same arithmetic and order as the per-layer kernels): block b
// serves outputs 32 * (b - job.block0) .. +31 of job block_job[b]
__global__ __launch_bounds__(256) void reduce_batch_kernel(const mny_reduce_job* __restrict__ jobs, const int32_t* __restrict__ block_job) {
    __shared__ double red[8][32];
    const mny_reduce_job jb = jobs[block_job[blockIdx.x]];
    const int ol = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int64_t i = (int64_t)((int)blockIdx.x - jb.block0) * 32 + ol, n = jb.n;
    const float* __restrict__ parts = jb.parts;
    const int nparts = jb.nparts;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < n) {
        int p = slice;
        for (; p + 24 < nparts; p += 32) {
            s0 += (double)parts[(int64_t)p * n + i];        s1 += (double)parts[(int64_t)(p + 8) * n + i];
            s2 += (double)parts[(int64_t)(p + 16) * n + i]; s3 += (double)parts[(int64_t)(p + 24) * n + i];
        }
        for (; p < nparts; p += 8) s0 += (double)parts[(int64_t)p * n + i];
    }
    red[slice][ol] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && i < n) {
        double s = 0.0;
        for (int k = 0; k < 8; ++k) s += red[k][ol];
        jb.out[i] = (float)s;
    }
}

int launch_reduce_parts(const float* parts, int nparts, int n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_parts_f64_kernel, dim3((n + 31) / 32), dim3(256), 0, st, parts, nparts, n, out);
    return check_launch("reduce_parts_f64_kernel");
}
}  // namespace mny

extern "C" int mny_reduce_batch(const mny_reduce_job* jobs, const int32_t* block_job, int nblocks, void* stream) {
    MNY_REQUIRE(jobs && block_job && nblocks > 0, "reduce_batch: bad arguments");
    hipLaunchKernelGGL(mny::reduce_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, jobs, block_job);
    return mny::check_launch("reduce_batch_kernel");
}

extern "C" int mny_version(void) { return 100; }
extern "C" const char* mny_last_error(void) { return mny::g_err; }
extern "C" int mny_max_parts(void) { return mny::kMaxParts; }
