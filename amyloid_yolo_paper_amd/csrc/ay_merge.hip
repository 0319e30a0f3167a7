// Union-merge of overlapping same-class detections on the device: the post-processing the paper's inference path
// applies after NMS (reference core.py:366-423 mergeDetections with core.py:326-364 combineIfOverlapping; `--merge_boxes True`).
//
// The reference keeps the rows in a Python set of tuples and, pass after pass, walks all pairs (i < j) of `list(tuple_set)`:
// two rows of the same class (0 or 1) whose truncated integer pixel rectangles [x, x+w) x [y, y+h) share a pixel are replaced by
// one row (left, top, right, bottom of the COVERED PIXELS -- one pixel short of the union rectangle on the far sides --, min of the
// confidences, min of the class confidences); a row merged in a pass is not used again in that pass, a merged row takes part
// from the next pass on, passes repeat until nothing changes.  Which pairs meet first is the set's iteration order, i.e.
// unspecified -- and through the one-pixel shrink it can move the edges of a chain of merges.  This kernel fixes the order: rows in
// input order, merged rows appended in creation order.  For inputs whose clusters are pairs the result is the reference's (as a
// set of rows, pinned by tests/golden/merge_cases.npz); for chains it is the reference's algorithm under that order
// (oracle/boxes_oracle.merge_detections_ordered restates it on the CPU).
//
// One wavefront per image.  The pair loop is sequential by nature (a merge removes its two rows); the search "first j > i that
// can merge with i" and the two membership tests (merged row already present / equal to a row removed earlier -- both by VALUE,
// as Python's set of tuples does) are wave-parallel with ballots.  All rows live in LDS.
#include "ay_common.h"

namespace ay {

constexpr int MERGE_MAX_IN = 1024;             // rows per image
constexpr int MERGE_CAP = 2 * MERGE_MAX_IN;    // every merge appends one row and retires two: at most 2n - 1 rows ever exist

struct MRow {
    float v[7];
};

__device__ __forceinline__ bool rows_equal(const float* a, const float* b) {
    bool eq = true;
#pragma unroll
    for (int k = 0; k < 7; ++k) eq = eq && (a[k] == b[k]);
    return eq;
}

__global__ void __launch_bounds__(64) merge_detections_kernel(const float* __restrict__ rows_in, const int* __restrict__ count_in, int max_in,
                                                              float* __restrict__ rows_out, int* __restrict__ count_out) {
    __shared__ float e[MERGE_CAP][7];
    // state: 0 = never existed / duplicate of an earlier input row, 1 = in the set, 2 = removed by a merge;
    // bit 4 (value 16) on a live row: its VALUE equals a row removed earlier (`entry in removed` is a set of tuples)
    __shared__ unsigned char st[MERGE_CAP];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    int n = count_in[b];
    n = n < 0 ? 0 : (n > max_in ? max_in : n);
    const float* src = rows_in + (size_t)b * max_in * 7;
    for (int i = lane; i < n * 7; i += 64) e[i / 7][i % 7] = src[i];
    for (int i = lane; i < MERGE_CAP; i += 64) st[i] = i < n ? 1 : 0;
    __syncthreads();
    // set(): identical rows collapse to the first one
    for (int i = 1; i < n; ++i) {
        bool dup = false;
        for (int k0 = 0; k0 < i; k0 += 64) {
            const int k = k0 + lane;
            const bool hit = k < i && st[k] == 1 && rows_equal(e[k], e[i]);
            if (__ballot(hit)) {
                dup = true;
                break;
            }
        }
        if (dup && lane == 0) st[i] = 0;
        __syncthreads();
    }
    int total = n;
    // truncated pixel rectangle of row k: x, y, w, h as the reference's int(x1), int(y1), int(x2 - x1), int(y2 - y1)
    auto rect = [&](int k, long long& x, long long& y, long long& w, long long& h) __attribute__((always_inline)) {
        const double x1 = e[k][0], y1 = e[k][1], x2 = e[k][2], y2 = e[k][3];
        x = (long long)x1, y = (long long)y1, w = (long long)(x2 - x1), h = (long long)(y2 - y1);
    };
    while (true) {
        bool changed = false;
        const int limit = total;   // rows created in this pass are not in this pass's list
        for (int i = 0; i < limit; ++i) {
            if (st[i] != 1) continue;   // gone, or live with a value that sits in `removed`
            const float label = e[i][6];
            if (!(label == 0.f || label == 1.f)) continue;
            long long xi, yi, wi, hi;
            rect(i, xi, yi, wi, hi);
            int j0 = i + 1;
            while (j0 < limit) {
                // first j >= j0 of this pass's list that may merge with i
                int found = -1;
                for (int base = j0; base < limit; base += 64) {
                    const int j = base + lane;
                    bool ok = false;
                    if (j < limit && st[j] == 1 && e[j][6] == label) {
                        long long xj, yj, wj, hj;
                        rect(j, xj, yj, wj, hj);
                        ok = wi > 0 && hi > 0 && wj > 0 && hj > 0 && min(xi + wi, xj + wj) > max(xi, xj) && min(yi + hi, yj + hj) > max(yi, yj);
                    }
                    const unsigned long long m = __ballot(ok);
                    if (m) {
                        found = base + __ffsll((long long)m) - 1;
                        break;
                    }
                }
                if (found < 0) break;
                long long xj, yj, wj, hj;
                rect(found, xj, yj, wj, hj);
                const long long left = min(xi, xj), top = min(yi, yj);
                const long long right = max(xi + wi, xj + wj) - 1, bottom = max(yi + hi, yj + hj) - 1;   // last covered pixel
                float m[7] = {(float)left, (float)top, (float)right, (float)bottom, fminf(e[i][4], e[found][4]), fminf(e[i][5], e[found][5]), label};
                // `new_entry not in tuple_set`: by value, among the rows in the set right now
                bool present = false, was_removed = false;
                for (int k0 = 0; k0 < total; k0 += 64) {
                    const int k = k0 + lane;
                    const bool eq = k < total && st[k] != 0 && rows_equal(e[k], m);
                    present = present || __ballot(eq && (st[k] & 3) == 1) != 0;
                    was_removed = was_removed || __ballot(eq && (st[k] & 3) == 2) != 0;
                }
                if (present) {   // the reference leaves both rows alone and goes on to the next j
                    j0 = found + 1;
                    continue;
                }
                if (total >= MERGE_CAP) break;   // cannot happen for n <= MERGE_MAX_IN (2n - 1 rows at most)
                __syncthreads();
                if (lane < 7) e[total][lane] = m[lane];
                if (lane == 0) {
                    st[total] = was_removed ? (1 | 16) : 1;
                    st[i] = 2;
                    st[found] = 2;
                }
                __syncthreads();
                ++total;
                changed = true;
                break;   // row i is in `removed` now: every further pair with it is skipped
            }
        }
        if (!changed) break;
    }
    // surviving rows in index order
    float* dst = rows_out + (size_t)b * max_in * 7;
    int out = 0;
    for (int k0 = 0; k0 < total; k0 += 64) {
        const int k = k0 + lane;
        const bool live = k < total && (st[k] & 3) == 1;
        const unsigned long long m = __ballot(live);
        if (live) {
            const int pos = out + __popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
            for (int c = 0; c < 7; ++c) dst[pos * 7 + c] = e[k][c];
        }
        out += __popcll(m);
    }
    if (lane == 0) count_out[b] = out;
}

}  // namespace ay

extern "C" int ay_merge_detections_max_rows(void) { return ay::MERGE_MAX_IN; }

extern "C" int ay_merge_detections(const float* rows, const int* count, int batch, int max_rows, float* rows_out, int* count_out,
                                   ay_stream_t stream) {
    using namespace ay;
    AY_CHECK_ARG(rows && count && rows_out && count_out && batch > 0, "ay_merge_detections: null / empty argument");
    AY_CHECK_ARG(max_rows > 0 && max_rows <= MERGE_MAX_IN, "ay_merge_detections: %d rows per image (at most %d)", max_rows, MERGE_MAX_IN);
    AY_CHECK_ARG(rows != rows_out, "ay_merge_detections: in-place");
    hipLaunchKernelGGL(merge_detections_kernel, dim3((unsigned)batch), dim3(64), 0, S(stream), rows, count, max_rows, rows_out, count_out);
    AY_CHECK_LAUNCH("merge_detections_kernel");
    return AY_OK;
}
