// Confidence-weighted merge-NMS of the reference (utils/utils.py:235-273) on gfx950.
//
//   nms_filter_kernel   whole grid: corners in place (:244), conf >= thr filter (:248), score = conf * max cls
//                       (:253), appends a 64-bit sort key per candidate: (~score_bits << 32) | row, so an
//                       ascending sort = score descending, ties -> lower original row first.
//   nms_merge_kernel    one workgroup per image: bitonic sort of the keys (LDS when they fit), gather of the
//                       candidates in sorted order, then the greedy class-aware scan (:260-269): the "alive" set
//                       is a bitmask of 64-bit words, each lane tests one candidate of a word against the current
//                       head, __ballot() gives the suppression mask of the word, the members' conf-weighted corner
//                       sums are wave-reduced for the merge.  Up to 1 024 candidates (the LDS path) the scan runs on
//                       FOUR wavefronts: a cluster only ever holds rows of one class (:262-264), so the candidates
//                       split into four class partitions (class & 3) whose greedy scans are independent; a wave keeps
//                       its partition's alive words in registers and walks all of them per head in one unrolled pass.
//                       The heads of all partitions are emitted in global score order by a prefix count over the head
//                       bitmap.  Larger candidate sets take the one-wavefront scan over the workspace.
//
// IoU uses the reference's +1-pixel rule and operation order (fp32, no contraction) so `> nms_thres`
// decisions are bit-identical; only the merged corners (a sum whose order the reference does not fix)
// differ in the last bits.
#include <stdlib.h>

#include "ay_common.h"

namespace ay {

__device__ __forceinline__ float iou_p1(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2,
                                        float by2) {
    const float ix1 = fmaxf(ax1, bx1), iy1 = fmaxf(ay1, by1);
    const float ix2 = fminf(ax2, bx2), iy2 = fminf(ay2, by2);
    const float inter = fmaxf(ix2 - ix1 + 1.0f, 0.0f) * fmaxf(iy2 - iy1 + 1.0f, 0.0f);
    const float a1 = (ax2 - ax1 + 1.0f) * (ay2 - ay1 + 1.0f);
    const float a2 = (bx2 - bx1 + 1.0f) * (by2 - by1 + 1.0f);
    return inter / (a1 + a2 - inter + 1e-16f);
}

static inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// workspace layout per image (cap = next_pow2(n_rows)):
//   keys  u64[cap] | cand f32[8][cap] (x1,y1,x2,y2,conf,cls_conf,cls_pred,row-as-int) | alive u64[cap/64] (only used
//   when an image has more than 65 536 candidates; smaller alive sets live in LDS)
struct NmsWs {
    unsigned long long* keys;
    float* cand;
    int cap;
};

// Candidates are appended per WORKGROUP: a workgroup collects its keys in LDS (LDS atomics) and reserves their places in the image's key
// array with ONE global atomic when its buffer fills up and at the end.  One atomic per candidate put 31 500 returning atomics per
// batch of 64 tiles on a single 256-byte line (the 64 adjacent counters): ~5 ns each, 160 us for a kernel that moves 200 MB.
constexpr int NMS_FILTER_BUF = 1024;   // keys a workgroup holds between two flushes (an iteration adds at most 256)

__global__ void __launch_bounds__(256) nms_filter_kernel(float* __restrict__ pred, int N, int C, float conf_thres, unsigned long long* keys, int cap,
                                                         int* cand_count) {
    __shared__ unsigned long long kbuf[NMS_FILTER_BUF];
    __shared__ int kcount, kbase;
    const int b = blockIdx.y;
    const int K = 5 + C;
    float* pb = pred + (size_t)b * N * K;
    unsigned long long* kb = keys + (size_t)b * cap;
    if (threadIdx.x == 0) kcount = 0;
    __syncthreads();
    auto flush = [&]() {   // every thread of the workgroup calls it
        __syncthreads();
        const int n = kcount;
        if (threadIdx.x == 0 && n > 0) kbase = atomicAdd(&cand_count[b], n);
        __syncthreads();
        // (a candidate count that does not start at zero -- a zero-fill lost or reordered upstream -- must not turn into a write past
        // this image's keys)
        for (int i = threadIdx.x; i < n; i += 256)
            if ((unsigned)(kbase + i) < (unsigned)cap) kb[kbase + i] = kbuf[i];
        __syncthreads();
        if (threadIdx.x == 0) kcount = 0;
        __syncthreads();
    };
    const bool rows32 = (K == 8) && (reinterpret_cast<uintptr_t>(pred) & 15) == 0;
    const int stride = gridDim.x * 256;
    const int iters = (N + stride - 1) / stride;   // the same trip count for every thread: the flushes are workgroup barriers
    for (int it = 0; it < iters; ++it) {
        const int r = it * stride + blockIdx.x * 256 + threadIdx.x;
        if (r < N) {
            float conf, mc;
            if (rows32) {
                // 3 classes (the paper's model): a row is 32 aligned bytes -- two 16-byte loads and one 16-byte store per row instead of
                // eight scalar loads and four scalar stores (this kernel runs beside the next batch's convolutions: the fewer memory
                // instructions it issues the less it takes from them).  Same operations on the same values.
                float4* p4 = reinterpret_cast<float4*>(pb + (size_t)r * 8);
                const float4 bx = p4[0], cf = p4[1];
                const float hw = bx.z / 2.0f, hh = bx.w / 2.0f;
                p4[0] = make_float4(bx.x - hw, bx.y - hh, bx.x + hw, bx.y + hh);
                conf = cf.x;
                mc = fmaxf(fmaxf(cf.y, cf.z), cf.w);
            } else {
                float* p = pb + (size_t)r * K;
                const float cx = p[0], cy = p[1], hw = p[2] / 2.0f, hh = p[3] / 2.0f;
                p[0] = cx - hw;
                p[1] = cy - hh;
                p[2] = cx + hw;
                p[3] = cy + hh;
                conf = p[4];
                mc = p[5];
                for (int k = 1; k < C; ++k) mc = fmaxf(mc, p[5 + k]);
            }
            if (conf >= conf_thres) {
                const float score = conf * mc;
                const unsigned sb = __builtin_bit_cast(unsigned, score);
                const int pos = atomicAdd(&kcount, 1);   // LDS
                kbuf[pos] = ((unsigned long long)(~sb) << 32) | (unsigned)r;
            }
        }
        if (it + 1 < iters) {
            __syncthreads();
            if (kcount > NMS_FILTER_BUF - 256) flush();   // workgroup-uniform (read behind the barrier)
        }
    }
    flush();
}

// candidate counters start at zero: a kernel rather than a memset, so that a captured detection step consists of this library's
// kernel nodes only
__global__ void nms_zero_counts_kernel(int* cand_count, int batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch) cand_count[i] = 0;
}

template <typename P>
__device__ void bitonic_sort(P d, int n, int tid, int nthreads) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = d[i], y = d[ixj];
                    const bool up = ((i & k) == 0);
                    if ((x > y) == up) {
                        d[i] = y;
                        d[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

constexpr int NMS_FAST = 1024;      // candidates per image whose arrays the fast path keeps in LDS (32 KiB)
constexpr int NMS_WORDS = NMS_FAST / 64;
constexpr int NMS_PARTS = 4;        // class partitions (class & 3) = wavefronts of the workgroup
constexpr int NMS_LDS_KEYS = 1024;  // 8 KiB of keys sorted in LDS (larger candidate sets sort in the workspace): with the 8-KiB alive mask the
                                    // workgroup stays at 16 KiB of LDS, so it can share a CU with a persistent convolution workgroup of the next
                                    // batch (117-144 KiB) instead of keeping 64 CUs away from it

__global__ void __launch_bounds__(256) nms_merge_kernel(const float* __restrict__ pred, int N, int C, float nms_thres,
                                                        unsigned long long* keys, float* cand, int cap,
                                                        const int* __restrict__ cand_count, int max_det,
                                                        float* __restrict__ out_rows, int* __restrict__ keep_idx,
                                                        int* __restrict__ count, unsigned long long* alive_ws, int mid_limit) {
    __shared__ unsigned long long skeys[NMS_LDS_KEYS];
    __shared__ float fc[8][NMS_FAST];  // fast path: x1, y1, x2, y2, conf, class, class conf, original row (as int) in sorted order
    __shared__ unsigned long long head_s[NMS_PARTS][NMS_WORDS];  // fast path: head bitmap per class partition, then ([0]) their union
    __shared__ int wpre[NMS_WORDS];                              // fast path: heads in the words below
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int K = 5 + C;
    int n = cand_count[b];
    if (n > N) n = N;
    if (n == 0) {
        if (tid == 0) count[b] = 0;
        return;
    }
    if (n > NMS_FAST && n <= mid_limit) return;   // nms_merge_mid_kernel's image (launched behind this kernel)
    unsigned long long* kb = keys + (size_t)b * cap;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    // ---- sort ------------------------------------------------------------------------------------
    const bool in_lds = np2 <= NMS_LDS_KEYS;
    if (in_lds) {
        for (int i = tid; i < np2; i += 256) skeys[i] = i < n ? kb[i] : ~0ull;
        __syncthreads();
        bitonic_sort(skeys, np2, tid, 256);
    } else {
        for (int i = n + tid; i < np2; i += 256) kb[i] = ~0ull;
        __syncthreads();
        bitonic_sort(kb, np2, tid, 256);
    }
    // ---- fast path: up to 1024 candidates, gathered into LDS, scanned by ONE wavefront -----------------------------
    // Same scan as below, but the candidate arrays live in LDS instead of the workspace: the greedy loop is a chain of
    // dependent reads (head box, then one candidate per lane per alive word), an L2 round trip each from the workspace
    // (1.3-2.3 ms per batch of 64 tiles) against an LDS access here.  Same arithmetic, same order of the
    // confidence-weighted sums: bit-identical results.
    if (in_lds && n <= NMS_FAST) {
        const float* pb = pred + (size_t)b * N * K;
        for (int i = tid; i < n; i += 256) {
            const int r = (int)(unsigned)(skeys[i] & 0xffffffffu);
            const float* p = pb + (size_t)r * K;
            float mc = p[5];
            int arg = 0;
            for (int k = 1; k < C; ++k) {
                const float v = p[5 + k];
                if (v > mc) {  // first maximum wins, like torch.max
                    mc = v;
                    arg = k;
                }
            }
            fc[0][i] = p[0];
            fc[1][i] = p[1];
            fc[2][i] = p[2];
            fc[3][i] = p[3];
            fc[4][i] = p[4];
            fc[5][i] = (float)arg;
            fc[6][i] = mc;
            reinterpret_cast<int*>(fc[7])[i] = r;
        }
        __syncthreads();
        const int nwords = (n + 63) >> 6;   // <= NMS_WORDS
        const int lane = tid & 63;
        const int part = tid >> 6;          // this wavefront scans the candidates whose class & 3 == part
        // ---- this partition's alive words, in registers (wave-uniform values: every loop over them is fully unrolled) ----------
        unsigned long long al[NMS_WORDS], hm[NMS_WORDS];
#pragma unroll
        for (int w = 0; w < NMS_WORDS; ++w) {
            const int j = w * 64 + lane;
            const bool mine = w < nwords && j < n && (((int)fc[5][min(j, NMS_FAST - 1)]) & 3) == part;
            al[w] = __ballot(mine);
            hm[w] = 0ull;
        }
        while (true) {
            int head = -1;
#pragma unroll
            for (int w = 0; w < NMS_WORDS; ++w)
                if (head < 0 && al[w] != 0ull) head = w * 64 + __builtin_ctzll(al[w]);
            if (head < 0) break;   // wave-uniform
            const float hx1 = fc[0][head], hy1 = fc[1][head], hx2 = fc[2][head], hy2 = fc[3][head], hcls = fc[5][head];
            float sw = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            // (the members of a cluster meet the lanes in the order of the one-wavefront scan -- ascending word, lane = position in the
            // word -- so the partial sums, and with them the merged corners, are the same bits)
#pragma unroll
            for (int w = 0; w < NMS_WORDS; ++w) {
                const unsigned long long a = al[w];
                if (a != 0ull && w * 64 + 63 >= head) {   // wave-uniform
                    const int j = w * 64 + lane;
                    bool member = (j == head);  // the head always leaves the set (also when its IoU is NaN)
                    if ((a >> lane) & 1ull) {
                        const float x1 = fc[0][j], y1 = fc[1][j], x2 = fc[2][j], y2 = fc[3][j];
                        const float iou = iou_p1(hx1, hy1, hx2, hy2, x1, y1, x2, y2);
                        member = member || ((iou > nms_thres) && (fc[5][j] == hcls));
                        if (member) {
                            const float wgt = fc[4][j];
                            sw += wgt;
                            s0 += wgt * x1;
                            s1 += wgt * y1;
                            s2 += wgt * x2;
                            s3 += wgt * y2;
                        }
                    }
                    al[w] = a & ~__ballot(member);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                sw += __shfl_xor(sw, off);
                s0 += __shfl_xor(s0, off);
                s1 += __shfl_xor(s1, off);
                s2 += __shfl_xor(s2, off);
                s3 += __shfl_xor(s3, off);
            }
            // the head's own entry is never read again (it has left every alive set, and other partitions never look at it): it
            // carries the merged corners to the emission pass
            if (lane == 0) {
                fc[0][head] = s0 / sw;
                fc[1][head] = s1 / sw;
                fc[2][head] = s2 / sw;
                fc[3][head] = s3 / sw;
            }
#pragma unroll
            for (int w = 0; w < NMS_WORDS; ++w)
                if ((head >> 6) == w) hm[w] |= 1ull << (head & 63);
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
#pragma unroll
            for (int w = 0; w < NMS_WORDS; ++w) head_s[part][w] = hm[w];
        }
        __syncthreads();
        // ---- emission in global score order (:273 appends the heads as the loop meets them): rank = heads at lower sorted positions
        if (tid < NMS_WORDS) {
            unsigned long long m = 0ull;
            for (int k = 0; k < NMS_PARTS; ++k) m |= head_s[k][tid];
            head_s[0][tid] = m;
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int w = 0; w < NMS_WORDS; ++w) {
                wpre[w] = acc;
                acc += __builtin_popcountll(head_s[0][w]);
            }
            count[b] = acc;  // > max_det means the caller's buffers were too small (rows were dropped)
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const unsigned long long m = head_s[0][i >> 6];
            if (!((m >> (i & 63)) & 1ull)) continue;
            const int rank = wpre[i >> 6] + __builtin_popcountll(m & ((1ull << (i & 63)) - 1ull));
            if (rank >= max_det) continue;
            float* o = out_rows + ((size_t)b * max_det + rank) * 7;
            o[0] = fc[0][i];
            o[1] = fc[1][i];
            o[2] = fc[2][i];
            o[3] = fc[3][i];
            o[4] = fc[4][i];
            o[5] = fc[6][i];
            o[6] = fc[5][i];
            keep_idx[(size_t)b * max_det + rank] = reinterpret_cast<const int*>(fc[7])[i];
        }
        return;
    }
    // ---- gather candidates in sorted order (:255-258) -------------------------------------------
    float* cb = cand + (size_t)b * 8 * cap;
    const float* pb = pred + (size_t)b * N * K;
    for (int i = tid; i < n; i += 256) {
        const unsigned long long key = in_lds ? skeys[i] : kb[i];
        const int r = (int)(unsigned)(key & 0xffffffffu);
        const float* p = pb + (size_t)r * K;
        float mc = p[5];
        int arg = 0;
        for (int k = 1; k < C; ++k) {
            const float v = p[5 + k];
            if (v > mc) {  // first maximum wins, like torch.max
                mc = v;
                arg = k;
            }
        }
        cb[0 * cap + i] = p[0];
        cb[1 * cap + i] = p[1];
        cb[2 * cap + i] = p[2];
        cb[3 * cap + i] = p[3];
        cb[4 * cap + i] = p[4];
        cb[5 * cap + i] = mc;
        cb[6 * cap + i] = (float)arg;
        reinterpret_cast<int*>(cb)[7 * cap + i] = r;
    }
    const int nwords = (n + 63) >> 6;
    // lane 0 publishes, all lanes re-read: never cache (volatile: LDS, or L2-coherent global accesses of one wave)
    volatile unsigned long long* alive = alive_ws + (size_t)b * ((cap >> 6) + 1);
    for (int i = tid; i < nwords; i += 256) {
        const int rem = n - i * 64;
        alive[i] = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    }
    __threadfence();
    __syncthreads();
    if (tid >= 64) return;
    // ---- greedy scan by one wavefront ---------------------------------------------------------------
    const int lane = tid;
    int kept = 0;
    int cw = 0;
    while (true) {
        unsigned long long aw = 0;
        while (cw < nwords && (aw = alive[cw]) == 0ull) ++cw;  // wave-uniform
        if (cw >= nwords) break;
        const int head = cw * 64 + __builtin_ctzll(aw);
        const float hx1 = cb[0 * cap + head], hy1 = cb[1 * cap + head], hx2 = cb[2 * cap + head], hy2 = cb[3 * cap + head];
        const float hcls = cb[6 * cap + head];
        float sw = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int w = cw; w < nwords; ++w) {
            const unsigned long long a = alive[w];
            if (a == 0ull) continue;
            const int j = w * 64 + lane;
            bool member = (j == head);  // the head always leaves the set (also when its IoU is NaN)
            if ((a >> lane) & 1ull) {
                const float x1 = cb[0 * cap + j], y1 = cb[1 * cap + j], x2 = cb[2 * cap + j], y2 = cb[3 * cap + j];
                const float iou = iou_p1(hx1, hy1, hx2, hy2, x1, y1, x2, y2);
                member = member || ((iou > nms_thres) && (cb[6 * cap + j] == hcls));
                if (member) {
                    const float wgt = cb[4 * cap + j];
                    sw += wgt;
                    s0 += wgt * x1;
                    s1 += wgt * y1;
                    s2 += wgt * x2;
                    s3 += wgt * y2;
                }
            }
            const unsigned long long m = __ballot(member);
            if (lane == 0) alive[w] = a & ~m;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sw += __shfl_xor(sw, off);
            s0 += __shfl_xor(s0, off);
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
            s3 += __shfl_xor(s3, off);
        }
        if (lane == 0 && kept < max_det) {
            float* o = out_rows + ((size_t)b * max_det + kept) * 7;
            o[0] = s0 / sw;
            o[1] = s1 / sw;
            o[2] = s2 / sw;
            o[3] = s3 / sw;
            o[4] = cb[4 * cap + head];
            o[5] = cb[5 * cap + head];
            o[6] = hcls;
            keep_idx[(size_t)b * max_det + kept] = reinterpret_cast<const int*>(cb)[7 * cap + head];
        }
        ++kept;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) count[b] = kept;  // > max_det means the caller's buffers were too small (rows were dropped)
}

// ---- 1 025 .. 4 096 candidates per image: the same four-partition scan with everything in LDS (128 KiB: this workgroup has a CU to
// itself, which the larger candidate sets of 2048^2 crops can afford; the workspace scan below it took 8 / 65 ms per image at 2 000 /
// 4 000 candidates, an L2 round trip per alive word and head).  A partition's alive words live one per LANE (lane w = word w, 64 words);
// per head the wave walks the non-empty words only (ballot over the lanes), fetches a word with v_readlane, and the lane that owns the
// word updates it.  Members meet the lanes in the order of the one-wavefront scan (ascending word, lane = position in the word).
constexpr int NMS_MID = 4096;
constexpr int NMS_MID_WORDS = NMS_MID / 64;

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ void __launch_bounds__(256) nms_merge_mid_kernel(const float* __restrict__ pred, int N, int C, float nms_thres,
                                                            const unsigned long long* __restrict__ keys, int cap,
                                                            const int* __restrict__ cand_count, int max_det, float* __restrict__ out_rows,
                                                            int* __restrict__ keep_idx, int* __restrict__ count) {
    __shared__ unsigned long long skeys[NMS_MID];
    __shared__ float fc[6][NMS_MID];   // x1, y1, x2, y2, conf, class in sorted order (class conf and row come back from pred / the keys)
    __shared__ unsigned long long head_s[NMS_PARTS][NMS_MID_WORDS];
    __shared__ int wpre[NMS_MID_WORDS];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int K = 5 + C;
    const int n = min(cand_count[b], N);
    if (n <= NMS_FAST || n > NMS_MID) return;   // nms_merge_kernel's images (LDS path up to 1 024, workspace scan beyond 4 096)
    const unsigned long long* kb = keys + (size_t)b * cap;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) skeys[i] = i < n ? kb[i] : ~0ull;
    for (int i = tid; i < NMS_PARTS * NMS_MID_WORDS; i += 256) (&head_s[0][0])[i] = 0ull;
    __syncthreads();
    bitonic_sort(skeys, np2, tid, 256);
    const float* pb = pred + (size_t)b * N * K;
    for (int i = tid; i < n; i += 256) {
        const int r = (int)(unsigned)(skeys[i] & 0xffffffffu);
        const float* p = pb + (size_t)r * K;
        float mc = p[5];
        int arg = 0;
        for (int k = 1; k < C; ++k) {
            const float v = p[5 + k];
            if (v > mc) {  // first maximum wins, like torch.max
                mc = v;
                arg = k;
            }
        }
        fc[0][i] = p[0];
        fc[1][i] = p[1];
        fc[2][i] = p[2];
        fc[3][i] = p[3];
        fc[4][i] = p[4];
        fc[5][i] = (float)arg;
    }
    __syncthreads();
    const int nwords = (n + 63) >> 6;
    const int lane = tid & 63;
    const int part = tid >> 6;
    // lane w: alive word w of this partition
    unsigned long long al = 0ull;
    for (int w = 0; w < nwords; ++w) {
        const int j = w * 64 + lane;
        const unsigned long long m = __ballot(j < n && (((int)fc[5][min(j, NMS_MID - 1)]) & 3) == part);
        if (lane == w) al = m;
    }
    while (true) {
        unsigned long long nz = __ballot(al != 0ull);   // non-empty words
        if (nz == 0ull) break;
        const int w0 = __builtin_ctzll(nz);
        const int head = w0 * 64 + __builtin_ctzll(readlane64(al, w0));
        const float hx1 = fc[0][head], hy1 = fc[1][head], hx2 = fc[2][head], hy2 = fc[3][head], hcls = fc[5][head];
        float sw = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        while (nz != 0ull) {
            const int w = __builtin_ctzll(nz);
            nz &= nz - 1ull;
            const unsigned long long a = readlane64(al, w);
            const int j = w * 64 + lane;
            bool member = (j == head);  // the head always leaves the set (also when its IoU is NaN)
            if ((a >> lane) & 1ull) {
                const float x1 = fc[0][j], y1 = fc[1][j], x2 = fc[2][j], y2 = fc[3][j];
                const float iou = iou_p1(hx1, hy1, hx2, hy2, x1, y1, x2, y2);
                member = member || ((iou > nms_thres) && (fc[5][j] == hcls));
                if (member) {
                    const float wgt = fc[4][j];
                    sw += wgt;
                    s0 += wgt * x1;
                    s1 += wgt * y1;
                    s2 += wgt * x2;
                    s3 += wgt * y2;
                }
            }
            const unsigned long long m = __ballot(member);
            if (lane == w) al = a & ~m;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sw += __shfl_xor(sw, off);
            s0 += __shfl_xor(s0, off);
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
            s3 += __shfl_xor(s3, off);
        }
        if (lane == 0) {   // the head's own entry is never read again: it carries the merged corners to the emission pass
            fc[0][head] = s0 / sw;
            fc[1][head] = s1 / sw;
            fc[2][head] = s2 / sw;
            fc[3][head] = s3 / sw;
            head_s[part][head >> 6] |= 1ull << (head & 63);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (tid < NMS_MID_WORDS) {
        unsigned long long m = 0ull;
        for (int k = 0; k < NMS_PARTS; ++k) m |= head_s[k][tid];
        head_s[0][tid] = m;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < NMS_MID_WORDS; ++w) {
            wpre[w] = acc;
            acc += __builtin_popcountll(head_s[0][w]);
        }
        count[b] = acc;  // > max_det means the caller's buffers were too small (rows were dropped)
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const unsigned long long m = head_s[0][i >> 6];
        if (!((m >> (i & 63)) & 1ull)) continue;
        const int rank = wpre[i >> 6] + __builtin_popcountll(m & ((1ull << (i & 63)) - 1ull));
        if (rank >= max_det) continue;
        const int r = (int)(unsigned)(skeys[i] & 0xffffffffu);
        const int cl = (int)fc[5][i];
        float* o = out_rows + ((size_t)b * max_det + rank) * 7;
        o[0] = fc[0][i];
        o[1] = fc[1][i];
        o[2] = fc[2][i];
        o[3] = fc[3][i];
        o[4] = fc[4][i];
        o[5] = pb[(size_t)r * K + 5 + cl];
        o[6] = fc[5][i];
        keep_idx[(size_t)b * max_det + rank] = r;
    }
}

}  // namespace ay

using namespace ay;

extern "C" size_t ay_nms_workspace_bytes(int batch, int n_rows) {
    if (batch <= 0 || n_rows <= 0) return 0;
    const size_t cap = (size_t)next_pow2(n_rows);
    return (size_t)batch * cap * (8 + 8 * 4) + (size_t)batch * (cap / 64 + 1) * 8;
}

static int nms_args_ok(int batch, int n_rows, int num_classes, size_t workspace_bytes, const char* who) {
    if (!(batch > 0 && n_rows > 0 && n_rows <= (1 << 24) && num_classes >= 1)) {
        set_error("%s: bad shape (1 <= rows <= 2^24)", who);
        return AY_ERR_ARG;
    }
    if (workspace_bytes < ay_nms_workspace_bytes(batch, n_rows)) {
        set_error("%s: workspace %zu < %zu", who, workspace_bytes, ay_nms_workspace_bytes(batch, n_rows));
        return AY_ERR_WORKSPACE;
    }
    return AY_OK;
}

extern "C" int ay_nms_filter(float* pred, int batch, int n_rows, int num_classes, float conf_thres, int32_t* cand_count,
                             void* workspace, size_t workspace_bytes, ay_stream_t stream) {
    AY_CHECK_ARG(pred && cand_count && workspace, "ay_nms_filter: null");
    if (int rc = nms_args_ok(batch, n_rows, num_classes, workspace_bytes, "ay_nms_filter")) return rc;
    hipStream_t st = S(stream);
    const int cap = next_pow2(n_rows);
    hipLaunchKernelGGL(nms_zero_counts_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, cand_count, batch);
    AY_CHECK_LAUNCH("nms_zero_counts_kernel");
    int gx = (n_rows + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(nms_filter_kernel, dim3(gx, batch), dim3(256), 0, st, pred, n_rows, num_classes, conf_thres,
                       (unsigned long long*)workspace, cap, cand_count);
    AY_CHECK_LAUNCH("nms_filter_kernel");
    return AY_OK;
}

extern "C" int ay_nms_sort_merge(const float* pred, int batch, int n_rows, int num_classes, float nms_thres, int max_det,
                                 float* out_rows, int32_t* keep_idx, int32_t* count, const int32_t* cand_count, void* workspace,
                                 size_t workspace_bytes, ay_stream_t stream) {
    AY_CHECK_ARG(pred && out_rows && keep_idx && count && cand_count && workspace && max_det > 0, "ay_nms_sort_merge: null");
    if (int rc = nms_args_ok(batch, n_rows, num_classes, workspace_bytes, "ay_nms_sort_merge")) return rc;
    const int cap = next_pow2(n_rows);
    unsigned long long* keys = (unsigned long long*)workspace;
    float* cand = (float*)((char*)workspace + (size_t)batch * cap * 8);
    unsigned long long* alive_ws = (unsigned long long*)((char*)workspace + (size_t)batch * cap * (8 + 8 * 4));
    // images with 1 025 .. 4 096 candidates go to the all-LDS kernel behind it (it returns at once for the others); AY_NMS_MID=0: the
    // workspace scan takes them, as before round 4
    static const int mid_on = getenv("AY_NMS_MID") ? atoi(getenv("AY_NMS_MID")) : 1;
    const int mid_limit = (mid_on && n_rows > NMS_FAST) ? NMS_MID : NMS_FAST;
    hipLaunchKernelGGL(nms_merge_kernel, dim3(batch), dim3(256), 0, S(stream), pred, n_rows, num_classes, nms_thres, keys, cand, cap,
                       cand_count, max_det, out_rows, keep_idx, count, alive_ws, mid_limit);
    AY_CHECK_LAUNCH("nms_merge_kernel");
    if (mid_limit > NMS_FAST) {
        hipLaunchKernelGGL(nms_merge_mid_kernel, dim3(batch), dim3(256), 0, S(stream), pred, n_rows, num_classes, nms_thres, keys, cap, cand_count,
                           max_det, out_rows, keep_idx, count);
        AY_CHECK_LAUNCH("nms_merge_mid_kernel");
    }
    return AY_OK;
}

extern "C" int ay_nms_merge(float* pred, int batch, int n_rows, int num_classes, float conf_thres, float nms_thres, int max_det,
                            float* out_rows, int32_t* keep_idx, int32_t* count, int32_t* cand_count, void* workspace,
                            size_t workspace_bytes, ay_stream_t stream) {
    if (int rc = ay_nms_filter(pred, batch, n_rows, num_classes, conf_thres, cand_count, workspace, workspace_bytes, stream)) return rc;
    return ay_nms_sort_merge(pred, batch, n_rows, num_classes, nms_thres, max_det, out_rows, keep_idx, count, cand_count, workspace,
                             workspace_bytes, stream);
}
