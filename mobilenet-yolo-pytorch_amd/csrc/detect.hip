// On-device detection math: YOLO loss forward+backward (target assignment, ignore mask, CIoU box
// term, label-smoothed class targets, weighted MSE), anchor decode + ordered stream compaction, and
// per-class greedy NMS (LDS bitonic sort + block-parallel suppression).  No host synchronisation
// anywhere: the reference's python triple loop with 4 `.item()` syncs per positive
// (models/yolo_loss.py:107-169) becomes five small launches per head.
//
// This file is compiled with -ffp-contract=off: the NMS / IoU arithmetic must round exactly like the
// CPU kernels it replaces (bit-exact kept indices), so no a*b+c may be fused behind our back.
//
// replaces models/yolo_loss.py:53-60,77-178,180-204,206-236,257-293,425-434, utils/iou.py:4-49,
// utils/box.py:11-31 and torchvision.ops.nms (third-party, see oracle/nms_ref.c for its semantics).
#include <stdlib.h>

#include "common.h"

namespace mny {

constexpr uint32_t F_NOOBJ = 1u;   // conf weight 1, conf target 0
constexpr uint32_t F_POS = 2u;     // conf weight 1, conf target 1, class weights 1

struct LossWs {
    uint32_t* info;      // [cells]
    uint32_t* cmask;     // [cells] class bits of the positives on that cell
    float4* boxgrad;     // [cells] d(sum_p (term_p-1)^2)/d(tx,ty,tw,th)
    float* p1;           // [nb1][2]  (#noobj cells, sum conf)
    float* img;          // [N][8]    per-image: count, obj, recall, iou, cls, sq, new_pos_cells, -
    float* p2;           // [nb2]     sum (out-tgt)^2 w
    float* scal;         // [16]
    int nb1, nb2;
};

__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }

struct Box { float x1, y1, x2, y2; };

// decode one cell; order of operations follows yolo_loss.py:89-92,243-247 (Q6, Q13)
__device__ __forceinline__ Box decode_box(float sx, float sy, float ew, float eh, int gx, int gy, float gf, float aw, float ah) {
    const float cx = (sx + (float)gx) / gf;
    const float cy = (sy + (float)gy) / gf;
    const float w = ew * aw, h = eh * ah;
    Box b;
    b.x1 = cx - w / 2; b.y1 = cy - h / 2;
    b.x2 = w + b.x1; b.y2 = h + b.y1;
    return b;
}

__device__ __forceinline__ float iou_ref(const Box& a, const Box& b) {   // utils/iou.py:32-49 (a = set_1)
    const float iw = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
    const float ih = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
    const float inter = iw * ih;
    const float aa = (a.x2 - a.x1) * (a.y2 - a.y1);
    const float ab = (b.x2 - b.x1) * (b.y2 - b.y1);
    return inter / (aa + ab - inter);
}

// ---- pass 1: one thread per cell: decode, best IoU over the image's GTs, flags ------------------
__global__ __launch_bounds__(256) void yolo_cells_kernel(const float* __restrict__ head, const float* __restrict__ targets,
                                                         const int32_t* __restrict__ t_off, const float* __restrict__ anchors,
                                                         const int32_t* __restrict__ mask, mny_yolo_head hp, LossWs ws) {
    __shared__ float red[2][256];
    const int g = hp.g, A = hp.A, T = 5 + hp.C;
    const int64_t cells = (int64_t)hp.N * A * g * g;
    float cnt = 0.f, csum = 0.f;
    for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < cells; cell += (int64_t)gridDim.x * blockDim.x) {
        // cell index order = (n, a, gy, gx), the reference's [N,A,g,g] order
        const int gx = (int)(cell % g), gy = (int)((cell / g) % g);
        const int a = (int)((cell / ((int64_t)g * g)) % A);
        const int n = (int)(cell / ((int64_t)g * g * A));
        const float* p = head + (((int64_t)n * g + gy) * g + gx) * (A * T) + a * T;
        const float conf = sigmoid_ref(p[4]);
        const int am = mask[a];
        const Box pb = decode_box(sigmoid_ref(p[0]), sigmoid_ref(p[1]), expf(p[2]), expf(p[3]), gx, gy, (float)g,
                                  anchors[am * 2], anchors[am * 2 + 1]);
        const int t0 = t_off[n], t1 = t_off[n + 1];
        uint32_t f = 0;
        if (t1 == t0) {
            f = F_NOOBJ;                                        // yolo_loss.py:108-111
        } else {
            float best = -INFINITY; bool nan = false;
            for (int t = t0; t < t1; ++t) {
                const float* q = targets + (int64_t)t * 5;
                Box gb;                                         // yolo_loss.py:112-113 (same op order as decode)
                gb.x1 = q[1] - q[3] / 2; gb.y1 = q[2] - q[4] / 2; gb.x2 = q[3] + gb.x1; gb.y2 = q[4] + gb.y1;
                const float v = iou_ref(gb, pb);
                if (v != v) nan = true; else best = fmaxf(best, v);
            }
            if (!nan && best < hp.ignore_thresh) f = F_NOOBJ;   // yolo_loss.py:123-125 (NaN max -> not below)
        }
        ws.info[cell] = f;
        ws.cmask[cell] = 0u;
        ws.boxgrad[cell] = f4zero();
        cnt += (f & F_NOOBJ) ? 1.f : 0.f;
        csum += conf;
    }
    red[0][threadIdx.x] = cnt; red[1][threadIdx.x] = csum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { ws.p1[blockIdx.x * 2] = red[0][0]; ws.p1[blockIdx.x * 2 + 1] = red[1][0]; }
}

// forward-mode dual number over the 4 predicted corner coordinates
struct D4 { float v, d[4]; };
__device__ __forceinline__ D4 dc(float c) { return D4{c, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 dvar(float c, int i) { D4 r = dc(c); r.d[i] = 1.f; return r; }
__device__ __forceinline__ D4 operator+(D4 a, D4 b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D4 operator-(D4 a, D4 b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D4 operator*(D4 a, D4 b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
    D4 r; r.v = a.v / b.v;
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return a.v > b.v ? a : b; }
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return a.v < b.v ? a : b; }
__device__ __forceinline__ D4 datan(D4 a) { D4 r; r.v = atanf(a.v); const float s = 1.f / (1.f + a.v * a.v); for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * s; return r; }

// (iou - ciou_term, iou) with gradient wrt the predicted box; yolo_loss.py:257-293 (alpha not detached)
__device__ void ciou_dual(const Box& gb, const Box& pb, D4& term, float& iou_out) {
    const D4 px1 = dvar(pb.x1, 0), py1 = dvar(pb.y1, 1), px2 = dvar(pb.x2, 2), py2 = dvar(pb.y2, 3);
    const D4 gx1 = dc(gb.x1), gy1 = dc(gb.y1), gx2 = dc(gb.x2), gy2 = dc(gb.y2);
    const D4 zero = dc(0.f);
    const D4 c = (dmax(gx2, px2) - dmin(gx1, px1)) * (dmax(gy2, py2) - dmin(gy1, py1));
    const D4 iw = dmax(dmin(gx2, px2) - dmax(gx1, px1), zero);
    const D4 ih = dmax(dmin(gy2, py2) - dmax(gy1, py1), zero);
    const D4 inter = iw * ih;
    const D4 ag = dc((gb.x2 - gb.x1) * (gb.y2 - gb.y1));
    const D4 ap = (px2 - px1) * (py2 - py1);
    const D4 iou = inter / (ag + ap - inter);
    const D4 w1 = gx2 - gx1, h1 = gy2 - gy1, w2 = px2 - px1, h2 = py2 - py1;
    const D4 two = dc(2.f);
    const D4 cx1 = (gx2 + gx1) / two, cy1 = (gy1 + gy2) / two;
    const D4 cx2 = (px2 + px1) / two, cy2 = (py1 + py2) / two;
    const D4 u = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2);
    const D4 dd = u / c;
    const D4 at = datan(w2 / h2) - datan(w1 / h1);
    const D4 v = dc((float)(4.0 / (3.14159265358979323846 * 3.14159265358979323846))) * at * at;
    const D4 alpha = v / (dc(1.f) - iou + v + dc(0.000001f));
    D4 ct = dd + alpha * v;
    if (c.v == 0.f) ct = iou;
    term = iou - ct;
    iou_out = iou.v;
}

// ---- pass 2: one thread per image: positives (order-dependent python loop, yolo_loss.py:127-169) --
__global__ void yolo_assign_kernel(const float* __restrict__ head, const float* __restrict__ targets,
                                   const int32_t* __restrict__ t_off, const float* __restrict__ anchors,
                                   const int32_t* __restrict__ mask, mny_yolo_head hp, LossWs ws) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= hp.N) return;
    const int g = hp.g, A = hp.A, T = 5 + hp.C;
    float count = 0.f, obj = 0.f, recall = 0.f, ious = 0.f, cls_s = 0.f, sq = 0.f, newpos = 0.f;
    for (int t = t_off[n]; t < t_off[n + 1]; ++t) {
        const float* q = targets + (int64_t)t * 5;
        const float gw = q[3], gh = q[4];
        const int gi = (int)(q[1] * (float)g), gj = (int)(q[2] * (float)g);   // :128,:136-137 (truncation)
        // IoU of [0,0,w,h] with all anchors [0,0,aw,ah] (:102,:129-133); first max wins like torch.argmax
        int best_n = 0; float best_v = -INFINITY;
        Box sb{0.f, 0.f, gw, gh};
        for (int k = 0; k < hp.n_anchors_all; ++k) {
            Box ab{0.f, 0.f, anchors[k * 2], anchors[k * 2 + 1]};
            const float v = iou_ref(sb, ab);
            if (v > best_v || (v != v && best_v == best_v)) { best_v = v; best_n = k; }
        }
        if (gi < 0 || gi >= g || gj < 0 || gj >= g) continue;   // reference would raise IndexError (:149)
        Box gb; gb.x1 = q[1] - gw / 2; gb.y1 = q[2] - gh / 2; gb.x2 = gw + gb.x1; gb.y2 = gh + gb.y1;
        const int cls = (int)(q[0] - 1.f);                        // :131,:147
        for (int k = 0; k < A; ++k) {
            const int am = mask[k];
            Box ab{0.f, 0.f, anchors[am * 2], anchors[am * 2 + 1]};
            const bool hit = (am == best_n) || (iou_ref(sb, ab) > hp.iou_thresh);   // :138-145
            if (!hit) continue;
            const int64_t cell = (((int64_t)n * A + k) * g + gj) * g + gi;
            const float* p = head + (((int64_t)n * g + gj) * g + gi) * (A * T) + k * T;
            const float conf = sigmoid_ref(p[4]);
            const float sx = sigmoid_ref(p[0]), sy = sigmoid_ref(p[1]), ew = expf(p[2]), eh = expf(p[3]);
            const Box pb = decode_box(sx, sy, ew, eh, gi, gj, (float)g, anchors[am * 2], anchors[am * 2 + 1]);
            D4 term; float iou;
            ciou_dual(gb, pb, term, iou);
            count += 1.f; obj += conf;
            if (iou > hp.ignore_thresh) recall += 1.f;             // :163-164
            ious += iou;
            if (cls >= 0 && cls < hp.C) cls_s += sigmoid_ref(p[5 + cls]);
            const float e = term.v - 1.f;
            sq += e * e;
            // d(e^2)/d corners -> d/d(tx,ty,tw,th): x1 = cx - w/2, x2 = cx + w/2, cx = (sig+gx)/g with
            // identity sigmoid backward (Q2), w = exp(tw)*aw
            const float g1 = 2.f * e * term.d[0], g2 = 2.f * e * term.d[1], g3 = 2.f * e * term.d[2], g4 = 2.f * e * term.d[3];
            const float wbox = ew * anchors[am * 2], hbox = eh * anchors[am * 2 + 1];
            float4 bg = ws.boxgrad[cell];
            bg.x += (g1 + g3) / (float)g;
            bg.y += (g2 + g4) / (float)g;
            bg.z += (g3 - g1) * 0.5f * wbox;
            bg.w += (g4 - g2) * 0.5f * hbox;
            ws.boxgrad[cell] = bg;
            const uint32_t f = ws.info[cell];
            if (!(f & (F_POS | F_NOOBJ))) newpos += 1.f;           // conf weight becomes 1 for the first time
            if (!(f & F_POS)) newpos += (float)hp.C;               // class weights become 1 (:432-433)
            ws.info[cell] = f | F_POS;
            if (cls >= 0 && cls < hp.C) ws.cmask[cell] |= (1u << cls);
        }
    }
    float* o = ws.img + (int64_t)n * 8;
    o[0] = count; o[1] = obj; o[2] = recall; o[3] = ious; o[4] = cls_s; o[5] = sq; o[6] = newpos; o[7] = 0.f;
}

// ---- pass 3: scalars: sum of weights, positives, ... (single block) ------------------------------
__global__ __launch_bounds__(256) void yolo_scalars_kernel(mny_yolo_head hp, LossWs ws) {
    __shared__ double red[9][256];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < ws.nb1; i += 256) { acc[0] += ws.p1[i * 2]; acc[1] += ws.p1[i * 2 + 1]; }
    for (int i = threadIdx.x; i < hp.N; i += 256)
        for (int k = 0; k < 7; ++k) acc[2 + k] += ws.img[(int64_t)i * 8 + k];
    for (int k = 0; k < 9; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) for (int k = 0; k < 9; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double noobj = red[0][0], confsum = red[1][0], count = red[2][0], obj = red[3][0], recall = red[4][0],
                     ious = red[5][0], cls = red[6][0], sq = red[7][0], newpos = red[8][0];
        float* s = ws.scal;
        s[0] = (float)(noobj + newpos);                         // sum of weights (Q4)
        s[1] = (float)count;                                    // positives incl. duplicates (Q5)
        s[2] = count > 0 ? (float)(sq / count) : 0.f;           // box loss (Q1)
        s[3] = count > 0 ? (float)(recall / count) : 0.f;
        s[4] = count > 0 ? (float)(ious / count) : 0.f;
        s[5] = count > 0 ? (float)(obj / count) : 0.f;
        const double cells = (double)hp.N * hp.A * hp.g * hp.g;
        s[6] = count > 0 ? (float)((confsum - obj) / (cells - count)) : 0.f;
        s[7] = count > 0 ? (float)(cls / count) : 0.f;
        s[8] = (float)(count / hp.N);
    }
}

// ---- pass 4: one thread per head element: gradient + loss partials -------------------------------
__global__ __launch_bounds__(256) void yolo_grad_kernel(const float* __restrict__ head, float* __restrict__ dhead,
                                                        mny_yolo_head hp, LossWs ws) {
    __shared__ float red[256];
    const int g = hp.g, A = hp.A, T = 5 + hp.C;
    const int64_t total = (int64_t)hp.N * g * g * A * T;
    const float sw = ws.scal[0], P = ws.scal[1];
    const float inv_sw = 1.f / sw;
    const float box_scale = P > 0.f ? hp.iou_weighting / P : 0.f;
    float part = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int attr = (int)(e % T);
        const int64_t q = e / T;                 // (n, gy, gx, a)
        const int a = (int)(q % A);
        const int64_t pix = q / A;               // (n, gy, gx)
        const int gx = (int)(pix % g), gy = (int)((pix / g) % g);
        const int64_t n = pix / ((int64_t)g * g);
        const int64_t cell = ((n * A + a) * g + gy) * g + gx;
        const uint32_t f = ws.info[cell];
        float grad = 0.f;
        if (attr < 4) {
            if (f & F_POS) {
                const float4 bg = ws.boxgrad[cell];
                grad = (attr == 0 ? bg.x : attr == 1 ? bg.y : attr == 2 ? bg.z : bg.w) * box_scale;
            }
        } else if (attr == 4) {
            if (f & (F_POS | F_NOOBJ)) {
                const float o = sigmoid_ref(head[e]);
                const float d = o - ((f & F_POS) ? 1.f : 0.f);
                part += d * d;
                grad = 2.f * d * inv_sw;          // Q3
            }
        } else if (f & F_POS) {
            const float o = sigmoid_ref(head[e]);
            const float tgt = ((ws.cmask[cell] >> (attr - 5)) & 1u) ? 0.95f : 0.05f;   // :426-427
            const float d = o - tgt;
            part += d * d;
            grad = 2.f * d * inv_sw;
        }
        dhead[e] = grad;
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) ws.p2[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void yolo_final_kernel(mny_yolo_head hp, LossWs ws, float* __restrict__ out7) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < ws.nb2; i += 256) a += ws.p2[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float* s = ws.scal;
        out7[0] = (float)(red[0] / (double)s[0]) + s[2] * hp.iou_weighting;   // :219,:234
        out7[1] = s[3]; out7[2] = s[4]; out7[3] = s[5]; out7[4] = s[6]; out7[5] = s[7]; out7[6] = s[8];
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t loss_ws_layout(const mny_yolo_head* hp, char* base, LossWs* ws) {
    const size_t cells = (size_t)hp->N * hp->A * hp->g * hp->g;
    const size_t elems = cells * (5 + hp->C);
    int nb1 = (int)cdiv((int64_t)cells, 256); if (nb1 > 1024) nb1 = 1024;
    int nb2 = (int)cdiv((int64_t)elems, 256); if (nb2 > 2048) nb2 = 2048;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
    const size_t o_info = take(cells * 4), o_cm = take(cells * 4), o_bg = take(cells * 16), o_p1 = take((size_t)nb1 * 8),
                 o_img = take((size_t)hp->N * 32), o_p2 = take((size_t)nb2 * 4), o_sc = take(64);
    if (ws) {
        ws->info = (uint32_t*)(base + o_info); ws->cmask = (uint32_t*)(base + o_cm); ws->boxgrad = (float4*)(base + o_bg);
        ws->p1 = (float*)(base + o_p1); ws->img = (float*)(base + o_img); ws->p2 = (float*)(base + o_p2);
        ws->scal = (float*)(base + o_sc); ws->nb1 = nb1; ws->nb2 = nb2;
    }
    return off;
}

// ---- decode + ordered compaction: one block per image ----------------------------------------------
__global__ __launch_bounds__(256) void yolo_decode_kernel(const float* __restrict__ head, const float* __restrict__ anchors,
                                                          const int32_t* __restrict__ mask, mny_yolo_head hp, float val_conf,
                                                          float* __restrict__ rows, int row_stride,
                                                          const int32_t* __restrict__ base_counts, int32_t* __restrict__ counts) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int n = blockIdx.x, g = hp.g, A = hp.A, T = 5 + hp.C;
    const int cells = A * g * g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = base_counts ? base_counts[n] : 0;
    __syncthreads();
    for (int c0 = 0; c0 < cells; c0 += 256) {
        const int cell = c0 + threadIdx.x;       // (a, gy, gx) order: yolo_loss.py:201-203
        bool keep = false;
        float r[7];
        if (cell < cells) {
            const int gx = cell % g, gy = (cell / g) % g, a = cell / (g * g);
            const float* p = head + (((int64_t)n * g + gy) * g + gx) * (A * T) + a * T;
            const int am = mask[a];
            const Box b = decode_box(sigmoid_ref(p[0]), sigmoid_ref(p[1]), expf(p[2]), expf(p[3]), gx, gy, (float)g,
                                     anchors[am * 2], anchors[am * 2 + 1]);
            const float conf = sigmoid_ref(p[4]);
            float best = -INFINITY; int bi = 0;
            for (int c = 0; c < hp.C; ++c) {
                const float s = sigmoid_ref(p[5 + c]);
                if (s > best) { best = s; bi = c; }
            }
            r[0] = b.x1; r[1] = b.y1; r[2] = b.x2; r[3] = b.y2; r[4] = conf; r[5] = best; r[6] = (float)bi;
            keep = conf > val_conf;
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int base = base_s;
        if (keep) {
            float* dst = rows + ((int64_t)n * row_stride + base + woff + before) * 7;
            for (int k = 0; k < 7; ++k) dst[k] = r[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[n] = base_s;
}

// ---- per-class NMS -------------------------------------------------------------------------------
constexpr int NMS_CAP = 8192;    // max boxes of one (image, class) bucket held in LDS

__device__ __forceinline__ uint32_t desc_key(float s) {       // larger score -> smaller key
    uint32_t b = __float_as_uint(s);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // ascending-orderable
    return ~b;
}

// grid (S, C).  Fills the bucket in original row order (stable), sorts by (score desc, position asc),
// greedy suppression with block-parallel IoU passes, writes kept row indices to tmp[bucket_base + r].
// SORT_ONLY (large buckets, see nms_mask_kernel): stops after the sort and writes the sorted boxes / row indices to
// kbox / tmp at bucket_base + j and the bucket size to kept_count; blockDim.x = 1024 there (256 otherwise).
template <bool SORT_ONLY>
__global__ __launch_bounds__(1024) void nms_bucket_kernel(const float* __restrict__ rows, const int32_t* __restrict__ seg_begin,
                                                          const int32_t* __restrict__ seg_count, int num_classes, double thr, int cap, int32_t* __restrict__ tmp,
                                                          float4* kbox, int32_t* __restrict__ bucket_base, int32_t* __restrict__ kept_count,
                                                          int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
    unsigned long long* keys = (unsigned long long*)nms_smem;           // [cap2] (score key << 32 | position)
    __shared__ int32_t s_wsum[2][16];
    __shared__ int32_t s_n, s_lower, s_kept;
    const int BT = blockDim.x, NWV = BT >> 6;
    const int s = blockIdx.x, c = blockIdx.y;
    const int r0 = seg_begin[s], r1 = r0 + seg_count[s];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float cf = (float)c;
    if (threadIdx.x == 0) { s_n = 0; s_lower = 0; }
    __syncthreads();
    // 1. stable fill (original row order) of the bucket's keys and row indices
    int32_t* rowidx = (int32_t*)(keys + cap);                           // [cap]
    unsigned char* dead = (unsigned char*)(rowidx + cap);               // [cap]
    for (int c0 = r0; c0 < r1; c0 += BT) {
        const int i = c0 + threadIdx.x;
        bool mine = false, lower = false;
        float score = 0.f;
        if (i < r1) {
            const float* r = rows + (int64_t)i * 7;
            const float cls = r[6];
            mine = (cls == cf);
            lower = (cls >= 0.f && cls < cf && cls == floorf(cls));
            if (mine) score = r[5] * r[4];                               // utils/box.py:27
        }
        const unsigned long long bm = __ballot(mine), bl = __ballot(lower);
        if (lane == 0) { s_wsum[0][wave] = __popcll(bm); s_wsum[1][wave] = __popcll(bl); }
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_wsum[0][w];
        const int base = s_n;
        if (mine) {
            const int pos = base + woff + __popcll(bm & ((1ull << lane) - 1ull));
            if (pos < cap) {
                keys[pos] = ((unsigned long long)desc_key(score) << 32) | (unsigned)pos;
                rowidx[pos] = i;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int a = 0, b = 0;
            for (int w = 0; w < NWV; ++w) { a += s_wsum[0][w]; b += s_wsum[1][w]; }
            s_n = base + a;
            s_lower += b;
        }
        __syncthreads();
    }
    const int n = s_n;
    const int bbase = r0 + s_lower;
    if (threadIdx.x == 0) bucket_base[s * num_classes + c] = bbase;
    if (n > cap) {                                                       // does not fit the LDS image: leave it to nms_big_bucket_kernel
        if (threadIdx.x == 0) kept_count[s * num_classes + c] = -n;       // (marker: negative size)
        return;
    }
    if (n == 0) { if (threadIdx.x == 0) kept_count[s * num_classes + c] = 0; return; }
    int n2 = 1; while (n2 < n) n2 <<= 1;
    for (int i = n + threadIdx.x; i < n2; i += BT) keys[i] = ~0ull;      // pad: sorts last
    for (int i = threadIdx.x; i < n; i += BT) dead[i] = 0;
    __syncthreads();
    // 2. bitonic sort (ascending key == descending score, ties by original position: stable)
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += BT) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], b = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    if (SORT_ONLY) {                                                    // hand the sorted bucket to nms_mask_kernel / nms_resolve_kernel
        for (int j = threadIdx.x; j < n; j += BT) {
            const int ri = rowidx[(unsigned)(keys[j] & 0xFFFFFFFFull)];
            const float* b = rows + (int64_t)ri * 7;
            kbox[bbase + j] = make_float4(b[0], b[1], b[2], b[3]);
            tmp[bbase + j] = ri;
        }
        if (threadIdx.x == 0) kept_count[s * num_classes + c] = n;
        return;
    }
    // 3. greedy suppression over the sorted order, 64 candidates at a time (identical result to the one-by-one loop):
    //    a candidate survives iff no box KEPT from earlier tiles suppresses it (checked in parallel: candidate = lane,
    //    the 4 waves split the kept list) and no earlier SURVIVOR of its own tile does (64x64 bitmask, resolved serially).
    __shared__ float4 tbox[64];
    __shared__ unsigned short tm16[64][4];
    __shared__ int tsup[64];
    __shared__ unsigned long long s_keepbits;
    if (threadIdx.x == 0) s_kept = 0;
    __syncthreads();
    const int cnd = threadIdx.x & 63, part = threadIdx.x >> 6;
    for (int tile0 = 0; tile0 < n; tile0 += 64) {
        const int j = tile0 + cnd;
        const bool valid = j < n;
        int ri = 0;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            ri = rowidx[(unsigned)(keys[j] & 0xFFFFFFFFull)];
            const float* b = rows + (int64_t)ri * 7;
            bx = make_float4(b[0], b[1], b[2], b[3]);
        }
        const float aj = (bx.z - bx.x) * (bx.w - bx.y);
        if (part == 0) { tbox[cnd] = bx; tsup[cnd] = valid ? 0 : 1; }
        __syncthreads();
        const int K = s_kept;
        bool sup = false;
        for (int k = part; k < K; k += 4) {                       // kept boxes: same address for the whole wave (broadcast)
            const float4 kb = kbox[bbase + k];
            const float ia = (kb.z - kb.x) * (kb.w - kb.y);
            const float xx1 = kb.x > bx.x ? kb.x : bx.x, yy1 = kb.y > bx.y ? kb.y : bx.y;
            const float xx2 = kb.z < bx.z ? kb.z : bx.z, yy2 = kb.w < bx.w ? kb.w : bx.w;
            float w = xx2 - xx1; if (!(w > 0.f)) w = 0.f;
            float h = yy2 - yy1; if (!(h > 0.f)) h = 0.f;
            const float inter = w * h;
            const float ovr = inter / (ia + aj - inter);
            if ((double)ovr > thr) sup = true;
        }
        if (sup && valid) tsup[cnd] = 1;                           // benign race: only ever set to 1
        unsigned bits = 0;                                         // which of candidates part*16..+15 would `cnd` suppress
        for (int q = 0; q < 16; ++q) {
            const int o = part * 16 + q;
            if (o <= cnd || tile0 + o >= n) continue;
            const float4 ob = tbox[o];
            const float ao = (ob.z - ob.x) * (ob.w - ob.y);
            const float xx1 = bx.x > ob.x ? bx.x : ob.x, yy1 = bx.y > ob.y ? bx.y : ob.y;
            const float xx2 = bx.z < ob.z ? bx.z : ob.z, yy2 = bx.w < ob.w ? bx.w : ob.w;
            float w = xx2 - xx1; if (!(w > 0.f)) w = 0.f;
            float h = yy2 - yy1; if (!(h > 0.f)) h = 0.f;
            const float inter = w * h;
            const float ovr = inter / (aj + ao - inter);
            if ((double)ovr > thr) bits |= 1u << q;
        }
        tm16[cnd][part] = (unsigned short)bits;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long removed = 0ull, keep = 0ull;
            const int lim = min(64, n - tile0);
            for (int i = 0; i < lim; ++i) {
                if (tsup[i] || ((removed >> i) & 1ull)) continue;
                keep |= 1ull << i;
                removed |= (unsigned long long)tm16[i][0] | ((unsigned long long)tm16[i][1] << 16) |
                           ((unsigned long long)tm16[i][2] << 32) | ((unsigned long long)tm16[i][3] << 48);
            }
            s_keepbits = keep;
        }
        __syncthreads();
        const unsigned long long keep = s_keepbits;
        if (part == 0 && ((keep >> cnd) & 1ull)) {
            const int r = K + __popcll(keep & ((1ull << cnd) - 1ull));
            tmp[bbase + r] = ri;
            kbox[bbase + r] = bx;
        }
        __syncthreads();                                           // kept boxes visible to the whole workgroup
        if (threadIdx.x == 0) s_kept = K + __popcll(keep);
        __syncthreads();
    }
    if (threadIdx.x == 0) kept_count[s * num_classes + c] = s_kept;
}

// ---- large buckets: suppression bit-matrix + wave-serial resolve ------------------------------------------------------
// One workgroup per (image, class) runs the greedy loop above on ONE CU; for a 100k-row segment with 20 classes that is 20 CUs
// of 256 and a kept list re-scanned for every tile.  Large buckets instead go through
//   nms_bucket_kernel<SORT_ONLY>  sort each bucket (1024 threads), write the sorted boxes
//   nms_mask_kernel               grid (row tile, bucket): bit j of word jt of row i = "i suppresses j" for every later j
//                                 (same IoU arithmetic and double compare as the loop above) — all CUs busy
//   nms_resolve_kernel            one wave per bucket walks the sorted order tile by tile: in-tile survivors by a 64-step
//                                 shuffle resolve, then ORs the kept rows' words into the `removed` bit set (batched loads)
// The kept set is the one the sequential algorithm produces (a box is kept iff no earlier KEPT box has its bit).
__global__ __launch_bounds__(512) void nms_mask_kernel(const float4* __restrict__ sbox, const int32_t* __restrict__ bucket_base,
                                                       const int32_t* __restrict__ bucket_n, double thr, int NT,
                                                       unsigned long long* __restrict__ mask) {
    const int b = blockIdx.y;
    const int n = max(bucket_n[b], 0), bbase = bucket_base[b];            // negative = oversized bucket (nms_big_bucket_kernel's)
    const int ntiles = (n + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // row tile blockIdx.x and its mirror ntiles-1-blockIdx.x: (ntiles - it) + (it + 1) column tiles per workgroup — the
    // triangle folded into equal shares
    for (int half = 0; half < 2; ++half) {
        const int it = half == 0 ? (int)blockIdx.x : ntiles - 1 - (int)blockIdx.x;
        if (it < 0 || it >= ntiles || (half == 1 && it <= (int)blockIdx.x)) continue;
        const int i = it * 64 + lane;
        const bool valid = i < n;
        const float4 bx = valid ? sbox[bbase + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float ai = (bx.z - bx.x) * (bx.w - bx.y);
        for (int jt = it + wave; jt < ntiles; jt += 8) {                  // 8 waves per workgroup share the column tiles
            unsigned long long bits = 0ull;
            const int lim = min(64, n - jt * 64);
#pragma unroll 4
            for (int q = 0; q < lim; ++q) {
                const float4 ob = sbox[bbase + jt * 64 + q];              // wave-uniform address: a scalar load
                const float ao = (ob.z - ob.x) * (ob.w - ob.y);
                const float xx1 = bx.x > ob.x ? bx.x : ob.x, yy1 = bx.y > ob.y ? bx.y : ob.y;
                const float xx2 = bx.z < ob.z ? bx.z : ob.z, yy2 = bx.w < ob.w ? bx.w : ob.w;
                float w = xx2 - xx1; if (!(w > 0.f)) w = 0.f;
                float h = yy2 - yy1; if (!(h > 0.f)) h = 0.f;
                const float inter = w * h;
                const float ovr = inter / (ai + ao - inter);
                if ((double)ovr > thr && (jt > it || q > lane)) bits |= 1ull << q;
            }
            if (valid) mask[(int64_t)(bbase + i) * NT + jt] = bits;
        }
    }
}

// one wave per bucket.  `srow` (= tmp) holds the sorted row indices on entry and the kept row indices on exit (in place:
// the r-th kept box is never behind its sorted position).  Per 64-box tile T:
//   removed = OR over every box kept so far of its word T   (independent 8-B loads, K/64 per lane, one wave-wide OR)
//   survivors of the tile by the 64-step serial resolve on the tile's own words (readlane broadcasts, no memory)
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void nms_resolve_kernel(const unsigned long long* __restrict__ mask, int NT, int32_t* __restrict__ srow,
                                                          const int32_t* __restrict__ bucket_base, int32_t* __restrict__ bucket_n_kept) {
    __shared__ unsigned short kpos[NMS_CAP];                              // sorted position of the r-th kept box
    __shared__ unsigned long long wred[4];
    __shared__ int s_K;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = bucket_n_kept[b], bbase = bucket_base[b];
    if (n < 0) return;                                                    // oversized bucket: keep the marker for nms_big_bucket_kernel
    if (tid == 0) s_K = 0;
    __syncthreads();
    const int ntiles = (n + 63) >> 6;
    for (int T = 0; T < ntiles; ++T) {
        const int K = s_K;
        // removed = OR over every kept box of its word T: all 4 waves, 8 independent loads per thread and batch
        unsigned long long w = 0ull;
        for (int k0 = 0; k0 < K; k0 += 256 * 8) {
            unsigned long long t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * 256 + tid;
                t[u] = k < K ? mask[(int64_t)(bbase + kpos[k]) * NT + T] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) w |= t[u];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) w |= __shfl_xor(w, d);
        if (lane == 0) wred[wave] = w;
        __syncthreads();
        if (wave == 0) {
            const int lim = min(64, n - T * 64);
            const bool valid = lane < lim;
            const unsigned long long own = valid ? mask[(int64_t)(bbase + T * 64 + lane) * NT + T] : 0ull;   // later boxes of this tile
            const int ri = valid ? srow[bbase + T * 64 + lane] : 0;
            unsigned long long removed = (wred[0] | wred[1]) | (wred[2] | wred[3]);
            const unsigned long long in_tile = lim == 64 ? ~0ull : ((1ull << lim) - 1ull);
            unsigned long long keep = 0ull, done = 0ull;
            // serial resolve over the not-yet-removed candidates only (in order): each step keeps one box
            for (;;) {
                const unsigned long long cand = in_tile & ~removed & ~done;
                if (!cand) break;
                const int q = __ffsll((long long)cand) - 1;
                keep |= 1ull << q;
                done |= 1ull << q;
                removed |= readlane64(own, q);
            }
            if ((keep >> lane) & 1ull) {
                const int r = K + __popcll(keep & ((1ull << lane) - 1ull));
                srow[bbase + r] = ri;
                kpos[r] = (unsigned short)(T * 64 + lane);
            }
            if (lane == 0) s_K = K + __popcll(keep);
        }
        __syncthreads();
    }
    if (tid == 0) bucket_n_kept[b] = s_K;
}

// ---- buckets beyond the LDS image (more than NMS_CAP boxes of one class in one image) --------------------------------------------
// utils/box.py:20-29 has no size limit.  Such a bucket is rare (the reference's own eval path produces at most 1 815 candidates
// per image), so it gets the simple treatment: ONE workgroup of 1024 threads with its sort keys and row indices in global
// scratch — a second scan of the segment to collect the bucket (stable), a bitonic sort over global memory (same key = score
// descending, original position ascending, so the order is the one the LDS sort produces), then the same 64-candidates-at-a-time
// greedy loop as nms_bucket_kernel (kept boxes re-read from kbox).  grid (S, C); every bucket that fitted exits at once.
// This is a RARE-PATH FALLBACK, deliberately not a fast path: log^2(n) barrier-separated passes over global memory and a
// kept list re-read per 64-candidate tile make one 50 000-box bucket cost milliseconds on one CU while the rest of the GPU idles; the
// large-segment driver (nms_bucket_kernel<SORT_ONLY> -> nms_mask_kernel -> nms_resolve_kernel, 106 M boxes/s on 100 k x 20) is the
// fast path for big inputs, and this kernel only sees a single (image, class) bucket of more than 8 192 boxes inside a batch of
// small segments.  It is launched only when the caller's max_seg_rows bound admits such a bucket at all.
//   gkeys  [2 * capacity] u64: bucket at 2 * bbase (padded to a power of two < 2n: regions of different buckets stay disjoint)
//   growidx[capacity]      i32: bucket at bbase
__global__ __launch_bounds__(1024) void nms_big_bucket_kernel(const float* __restrict__ rows, const int32_t* __restrict__ seg_begin,
                                                              const int32_t* __restrict__ seg_count, int num_classes, double thr,
                                                              int32_t* __restrict__ tmp, float4* kbox, const int32_t* __restrict__ bucket_base,
                                                              int32_t* __restrict__ kept_count, unsigned long long* __restrict__ gkeys,
                                                              int32_t* __restrict__ growidx) {
    const int s = blockIdx.x, c = blockIdx.y, b = s * num_classes + c;
    const int marker = kept_count[b];
    if (marker >= 0) return;
    const int n = -marker, bbase = bucket_base[b];
    const int BT = blockDim.x, NWV = BT >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = seg_begin[s], r1 = r0 + seg_count[s];
    const float cf = (float)c;
    unsigned long long* keys = gkeys + 2 * (int64_t)bbase;
    int32_t* rowidx = growidx + bbase;
    __shared__ int32_t s_wsum[16];
    __shared__ int32_t s_n, s_kept;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    // 1. stable fill
    for (int c0 = r0; c0 < r1; c0 += BT) {
        const int i = c0 + threadIdx.x;
        bool mine = false;
        float score = 0.f;
        if (i < r1) {
            const float* r = rows + (int64_t)i * 7;
            mine = (r[6] == cf);
            if (mine) score = r[5] * r[4];                               // utils/box.py:27
        }
        const unsigned long long bm = __ballot(mine);
        if (lane == 0) s_wsum[wave] = __popcll(bm);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_wsum[w];
        const int base = s_n;
        if (mine) {
            const int pos = base + woff + __popcll(bm & ((1ull << lane) - 1ull));
            keys[pos] = ((unsigned long long)desc_key(score) << 32) | (unsigned)pos;
            rowidx[pos] = i;
        }
        __syncthreads();
        if (threadIdx.x == 0) { int a = 0; for (int w = 0; w < NWV; ++w) a += s_wsum[w]; s_n = base + a; }
        __syncthreads();
    }
    int n2 = 1; while (n2 < n) n2 <<= 1;
    for (int i = n + threadIdx.x; i < n2; i += BT) keys[i] = ~0ull;
    __syncthreads();
    // 2. bitonic sort in global memory (one workgroup: __syncthreads orders its own global accesses)
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += BT) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], bb = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > bb) == up) { keys[i] = bb; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    // 3. greedy suppression, 64 candidates at a time (see nms_bucket_kernel): 16 waves split the kept list
    __shared__ float4 tbox[64];
    __shared__ unsigned long long tmask[64];
    __shared__ int tsup[64];
    __shared__ unsigned long long s_keepbits;
    if (threadIdx.x == 0) s_kept = 0;
    __syncthreads();
    const int cnd = lane, part = wave;
    for (int tile0 = 0; tile0 < n; tile0 += 64) {
        const int j = tile0 + cnd;
        const bool valid = j < n;
        int ri = 0;
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            ri = rowidx[(unsigned)(keys[j] & 0xFFFFFFFFull)];
            const float* bp = rows + (int64_t)ri * 7;
            bx = make_float4(bp[0], bp[1], bp[2], bp[3]);
        }
        const float aj = (bx.z - bx.x) * (bx.w - bx.y);
        if (part == 0) { tbox[cnd] = bx; tsup[cnd] = valid ? 0 : 1; tmask[cnd] = 0ull; }
        __syncthreads();
        const int K = s_kept;
        bool sup = false;
        for (int k = part; k < K; k += NWV) {
            const float4 kb = kbox[bbase + k];
            const float ia = (kb.z - kb.x) * (kb.w - kb.y);
            const float xx1 = kb.x > bx.x ? kb.x : bx.x, yy1 = kb.y > bx.y ? kb.y : bx.y;
            const float xx2 = kb.z < bx.z ? kb.z : bx.z, yy2 = kb.w < bx.w ? kb.w : bx.w;
            float w = xx2 - xx1; if (!(w > 0.f)) w = 0.f;
            float h = yy2 - yy1; if (!(h > 0.f)) h = 0.f;
            const float inter = w * h;
            const float ovr = inter / (ia + aj - inter);
            if ((double)ovr > thr) sup = true;
        }
        if (sup && valid) tsup[cnd] = 1;                           // benign race: only ever set to 1
        unsigned long long bits = 0ull;                            // which of candidates part*4..+3 would `cnd` suppress
        for (int q = 0; q < 64 / 16; ++q) {
            const int o = part * (64 / 16) + q;
            if (part >= 16 || o <= cnd || tile0 + o >= n) continue;
            const float4 ob = tbox[o];
            const float ao = (ob.z - ob.x) * (ob.w - ob.y);
            const float xx1 = bx.x > ob.x ? bx.x : ob.x, yy1 = bx.y > ob.y ? bx.y : ob.y;
            const float xx2 = bx.z < ob.z ? bx.z : ob.z, yy2 = bx.w < ob.w ? bx.w : ob.w;
            float w = xx2 - xx1; if (!(w > 0.f)) w = 0.f;
            float h = yy2 - yy1; if (!(h > 0.f)) h = 0.f;
            const float inter = w * h;
            const float ovr = inter / (aj + ao - inter);
            if ((double)ovr > thr) bits |= 1ull << o;
        }
        if (bits) atomicOr(&tmask[cnd], bits);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long removed = 0ull, keep = 0ull;
            const int lim = min(64, n - tile0);
            for (int i = 0; i < lim; ++i) {
                if (tsup[i] || ((removed >> i) & 1ull)) continue;
                keep |= 1ull << i;
                removed |= tmask[i];
            }
            s_keepbits = keep;
        }
        __syncthreads();
        const unsigned long long keep = s_keepbits;
        if (part == 0 && ((keep >> cnd) & 1ull)) {
            const int r = K + __popcll(keep & ((1ull << cnd) - 1ull));
            tmp[bbase + r] = ri;
            kbox[bbase + r] = bx;
        }
        __syncthreads();                                           // kept boxes visible to the whole workgroup
        if (threadIdx.x == 0) s_kept = K + __popcll(keep);
        __syncthreads();
    }
    if (threadIdx.x == 0) kept_count[b] = s_kept;
}

// grid (S, slices): concatenate the kept lists of a segment in class order (every block walks the class offsets, copies its
// slice of each class)
__global__ __launch_bounds__(256) void nms_compact_kernel(const int32_t* __restrict__ seg_begin, int num_classes,
                                                          const int32_t* __restrict__ tmp, const int32_t* __restrict__ bucket_base,
                                                          const int32_t* __restrict__ kept_count, int32_t* __restrict__ out_idx,
                                                          int32_t* __restrict__ out_counts, int32_t* __restrict__ status) {
    const int s = blockIdx.x;
    const int begin = seg_begin[s];
    const int t0 = blockIdx.y * 256 + threadIdx.x, tstride = gridDim.y * 256;
    int off = begin;
    for (int c = 0; c < num_classes; ++c) {
        int k = kept_count[s * num_classes + c];
        const int b = bucket_base[s * num_classes + c];
        if (k < 0) {                     // an oversized bucket nobody processed: the caller's max_seg_rows was smaller than a real segment
            if (t0 == 0) atomicMax(status, -k);
            k = 0;
        }
        for (int i = t0; i < k; i += tstride) out_idx[off + i] = tmp[b + i];
        off += k;
    }
    if (t0 == 0) out_counts[s] = off - begin;
}

// single block: exclusive scan of out_counts -> out_prefix[S+1]
__global__ __launch_bounds__(256) void nms_scan_kernel(const int32_t* __restrict__ counts, int S, int32_t* __restrict__ prefix) {
    __shared__ int32_t tot[256];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < S; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = i < S ? counts[i] : 0;
        tot[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int t = threadIdx.x >= d ? tot[threadIdx.x - d] : 0;
            __syncthreads();
            tot[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < S) prefix[i] = carry + tot[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += tot[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) prefix[S] = carry;
}

// grid S: gather the kept rows densely, segment after segment
__global__ __launch_bounds__(256) void nms_gather_kernel(const float* __restrict__ rows, const int32_t* __restrict__ seg_begin,
                                                         const int32_t* __restrict__ out_idx, const int32_t* __restrict__ out_counts,
                                                         const int32_t* __restrict__ prefix, float* __restrict__ out_rows) {
    const int s = blockIdx.x;
    const int k = out_counts[s], b = seg_begin[s], o = prefix[s];
    for (int i = blockIdx.y * 256 + threadIdx.x; i < k * 7; i += gridDim.y * 256) {
        const int r = i / 7, f = i % 7;
        out_rows[(int64_t)(o + r) * 7 + f] = rows[(int64_t)out_idx[b + r] * 7 + f];
    }
}

}  // namespace mny

using namespace mny;

static int nms_cap_for(int total) {
    int cap = 1024;
    while (cap < total && cap < NMS_CAP) cap <<= 1;
    return cap;
}

extern "C" size_t mny_yolo_loss_ws_bytes(const mny_yolo_head* hp, int total_targets) {
    (void)total_targets;
    if (!hp || hp->N <= 0 || hp->g <= 0 || hp->A <= 0 || hp->C <= 0) return 0;
    return loss_ws_layout(hp, nullptr, nullptr);
}

extern "C" int mny_yolo_loss(const float* head, const float* targets, const int32_t* t_off, const float* anchors_all,
                             const int32_t* mask, const mny_yolo_head* hp, float* out7, float* dhead, void* wsp, void* stream) {
    MNY_REQUIRE(head && t_off && anchors_all && mask && hp && out7 && dhead && wsp, "yolo_loss: null pointer");
    MNY_REQUIRE(hp->N > 0 && hp->g > 0 && hp->A > 0 && hp->C > 0 && hp->C <= 32, "yolo_loss: bad head spec (C must be <= 32)");
    LossWs ws;
    loss_ws_layout(hp, (char*)wsp, &ws);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(yolo_cells_kernel, dim3(ws.nb1), dim3(256), 0, st, head, targets, t_off, anchors_all, mask, *hp, ws);
    hipLaunchKernelGGL(yolo_assign_kernel, dim3((hp->N + 63) / 64), dim3(64), 0, st, head, targets, t_off, anchors_all, mask, *hp, ws);
    hipLaunchKernelGGL(yolo_scalars_kernel, dim3(1), dim3(256), 0, st, *hp, ws);
    hipLaunchKernelGGL(yolo_grad_kernel, dim3(ws.nb2), dim3(256), 0, st, head, dhead, *hp, ws);
    hipLaunchKernelGGL(yolo_final_kernel, dim3(1), dim3(256), 0, st, *hp, ws, out7);
    return check_launch("yolo_loss kernels");
}

extern "C" int mny_yolo_decode(const float* head, const float* anchors_all, const int32_t* mask, const mny_yolo_head* hp,
                               float val_conf, float* rows, int row_stride, const int32_t* base_counts, int32_t* counts,
                               void* stream) {
    MNY_REQUIRE(head && anchors_all && mask && hp && rows && counts, "yolo_decode: null pointer");
    MNY_REQUIRE(hp->N > 0 && hp->g > 0 && hp->A > 0 && hp->C > 0, "yolo_decode: bad head spec");
    MNY_REQUIRE(row_stride >= hp->A * hp->g * hp->g, "yolo_decode: row_stride %d smaller than the %d cells of one image", row_stride, hp->A * hp->g * hp->g);
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(hp->N), dim3(256), 0, (hipStream_t)stream, head, anchors_all, mask, *hp, val_conf, rows,
                       row_stride, base_counts, counts);
    return check_launch("yolo_decode_kernel");
}


// large-bucket path (bit-matrix): taken when the segments average more than 2048 rows
static bool nms_large(int S, int capacity) { return capacity > 2048 * S && getenv("MNY_NMS_SMALL") == nullptr; }

extern "C" size_t mny_nms_ws_bytes(int S, int capacity, int num_classes) {
    if (S <= 0 || capacity < 0 || num_classes <= 0) return 0;
    // tmp[capacity] + bucket_base[S*C] + kept_count[S*C] + prefix[S+1] + status + box cache float4[capacity]
    // (+ the suppression bit-matrix [capacity][cap/64] u64 on the large-bucket path)
    size_t bytes = align256((size_t)(capacity > 0 ? capacity : 1) * 4) + 2 * align256((size_t)S * num_classes * 4) + align256((size_t)(S + 1) * 4) + 256 +
                   align256((size_t)(capacity > 0 ? capacity : 1) * 16);
    if (nms_large(S, capacity)) bytes += align256((size_t)capacity * (nms_cap_for(capacity) / 64) * 8);
    if (capacity > NMS_CAP) bytes += align256((size_t)capacity * 16) + align256((size_t)capacity * 4);     // oversized buckets: global sort keys + row indices
    return bytes;
}

extern "C" size_t mny_nms_status_offset(int S, int capacity, int num_classes) {
    return align256((size_t)(capacity > 0 ? capacity : 1) * 4) + 2 * align256((size_t)S * num_classes * 4) + align256((size_t)(S + 1) * 4);
}

extern "C" size_t mny_nms_prefix_offset(int S, int capacity, int num_classes) {
    return align256((size_t)(capacity > 0 ? capacity : 1) * 4) + 2 * align256((size_t)S * num_classes * 4);
}

extern "C" int mny_nms_per_class(const float* rows, const int32_t* seg_begin, const int32_t* seg_count, int S, int capacity,
                                 int max_seg_rows, int num_classes, double thr, int32_t* out_idx, int32_t* out_counts,
                                 float* out_rows, void* wsp, void* stream) {
    MNY_REQUIRE(seg_begin && seg_count && out_idx && out_counts && wsp, "nms: null pointer");
    MNY_REQUIRE(S > 0 && capacity >= 0 && num_classes > 0, "nms: bad sizes");
    char* base = (char*)wsp;
    int32_t* tmp = (int32_t*)base;
    int32_t* bucket_base = (int32_t*)(base + align256((size_t)(capacity > 0 ? capacity : 1) * 4));
    int32_t* kept = (int32_t*)((char*)bucket_base + align256((size_t)S * num_classes * 4));
    int32_t* prefix = (int32_t*)((char*)kept + align256((size_t)S * num_classes * 4));
    int32_t* status = (int32_t*)((char*)prefix + align256((size_t)(S + 1) * 4));   // largest bucket that did NOT fit (0 = ok)
    float4* kbox = (float4*)((char*)status + 256);
    hipStream_t st = (hipStream_t)stream;
    const int cap = nms_cap_for(max_seg_rows > 0 && max_seg_rows < capacity ? max_seg_rows : capacity);
    const size_t lds = (size_t)cap * (8 + 4 + 1);
    if (!allow_lds((const void*)nms_bucket_kernel<false>, NMS_CAP * 13) || !allow_lds((const void*)nms_bucket_kernel<true>, NMS_CAP * 13)) {
        set_error("nms: hipFuncSetAttribute failed"); return MNY_EHIP;
    }
    if (hipMemsetAsync(status, 0, 4, st) != hipSuccess) { set_error("nms: memset failed"); return MNY_EHIP; }
    char* after = (char*)kbox + align256((size_t)(capacity > 0 ? capacity : 1) * 16);
    if (nms_large(S, capacity)) after += align256((size_t)capacity * (nms_cap_for(capacity) / 64) * 8);
    unsigned long long* gkeys = (unsigned long long*)after;               // only laid out when capacity > NMS_CAP (mny_nms_ws_bytes)
    int32_t* growidx = (int32_t*)(after + align256((size_t)capacity * 16));
    const int seg_bound = max_seg_rows > 0 && max_seg_rows < capacity ? max_seg_rows : capacity;       // no bucket is larger than its segment
    if (nms_large(S, capacity)) {
        unsigned long long* mask = (unsigned long long*)((char*)kbox + align256((size_t)(capacity > 0 ? capacity : 1) * 16));
        const int NT = cap / 64;
        hipLaunchKernelGGL(nms_bucket_kernel<true>, dim3(S, num_classes), dim3(1024), lds, st, rows, seg_begin, seg_count, num_classes, thr, cap,
                           tmp, kbox, bucket_base, kept, status);
        hipLaunchKernelGGL(nms_mask_kernel, dim3((NT + 1) / 2, S * num_classes), dim3(512), 0, st, kbox, bucket_base, kept, thr, NT, mask);
        hipLaunchKernelGGL(nms_resolve_kernel, dim3(S * num_classes), dim3(256), 0, st, mask, NT, tmp, bucket_base, kept);
    } else
    hipLaunchKernelGGL(nms_bucket_kernel<false>, dim3(S, num_classes), dim3(256), lds, st, rows, seg_begin, seg_count, num_classes, thr, cap, tmp,
                       kbox, bucket_base, kept, status);
    if (seg_bound > NMS_CAP)                                               // a bucket may exceed the LDS image: those (marked) go through global scratch
        hipLaunchKernelGGL(nms_big_bucket_kernel, dim3(S, num_classes), dim3(1024), 0, st, rows, seg_begin, seg_count, num_classes, thr, tmp, kbox,
                           bucket_base, kept, gkeys, growidx);
    const int slices = nms_large(S, capacity) ? 64 : 1;                    // few huge segments: spread the copies over more blocks
    hipLaunchKernelGGL(nms_compact_kernel, dim3(S, slices), dim3(256), 0, st, seg_begin, num_classes, tmp, bucket_base, kept, out_idx, out_counts, status);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(256), 0, st, out_counts, S, prefix);
    if (out_rows)
        hipLaunchKernelGGL(nms_gather_kernel, dim3(S, slices), dim3(256), 0, st, rows, seg_begin, out_idx, out_counts, prefix, out_rows);
    return check_launch("nms kernels");
}
