// Inference plan: the layer interpreter of the reference (Darknet.forward, models.py:237-255) as a flat op list issued
// from native code.  The host lowers the cfg graph once (amyloid_yolo_paper_amd/models.py: Darknet._lower); this file
//   * checks the dataflow (every value written before it is read),
//   * lays the values out in one arena: a value's bytes are free again once the op that reads it last has been issued --
//     the whole plan runs on one stream, so issue order is execution order -- first-fit over [def, last use] lifetimes,
//   * issues the ops through the same extern "C" entry points a per-layer caller uses (bit-identical results).
// Host code only; nothing here touches the device except through those entry points and HIP events.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "ay_common.h"  // pulls in include/amyloid_yolo.h

struct ay_plan {
    std::vector<ay_plan_op> ops;
    std::vector<size_t> bytes, offset;
    std::vector<int> def_op, last_use;
    size_t arena = 0;
    int img_dim = 0, n_total = 0;
    int dtype = AY_DT_BF16;  // storage type of activations and packed filters
    // fused[i] = j > 0: op i is a detection head (linear 1x1 convolution into an fp32 value) whose only reader is the decode op j:
    // issued as ONE ay_head_decode_fwd_* launch, the head value is never written; fused[j] = -1: issued with op i, skip
    std::vector<int> fused;
    // profiling (ay_plan_profile_begin/end): one event list per recorded forward, a pair per selected op
    mutable bool profiling = false;
    mutable int prof_every = 1, prof_seen = 0;   // event pairs on every prof_every-th forward since ay_plan_profile_begin*
    mutable std::vector<std::vector<hipEvent_t>> prof_events;
    std::vector<unsigned char> prof_sel;
};

namespace {

constexpr size_t ALIGN = 256;

bool uses_src2(int kind) { return kind == AY_OP_CONV1X1_CAT || kind == AY_OP_CONCAT_UPSAMPLE; }

// reads of an op, as value ids (AY_PLAN_INPUT / AY_PLAN_NONE filtered by the caller)
void op_reads(const ay_plan_op& o, int (&r)[3]) {
    r[0] = (o.kind == AY_OP_STEM_S2_FUSED || o.kind == AY_OP_STEM) ? AY_PLAN_NONE : o.src;
    r[1] = uses_src2(o.kind) ? o.src2 : AY_PLAN_NONE;
    r[2] = o.kind == AY_OP_CONV ? o.res : AY_PLAN_NONE;
}

int issue(const ay_plan* p, size_t idx, const float* x, uint8_t* ws, float* out, ay_stream_t st) {
    const ay_plan_op& o = p->ops[idx];
    auto at = [&](int v) -> void* { return v >= 0 ? ws + p->offset[v] : nullptr; };
    const ay_conv_desc& d = o.conv;
    if (p->fused[idx] < 0) return AY_OK;   // a decode that went out with its head
    if (p->fused[idx] > 0) {
        const ay_plan_op& y = p->ops[p->fused[idx]];
        return (p->dtype == AY_DT_F16 ? ay_head_decode_fwd_f16 : ay_head_decode_fwd_bf16)(
            &d, at(o.src), o.w, o.scale, o.shift, y.num_anchors, y.num_classes, p->img_dim, y.anchors_wh, out, p->n_total, y.row_offset, st);
    }
    if (p->dtype == AY_DT_F16) {
        switch (o.kind) {
            case AY_OP_STEM_S2_FUSED:
                return ay_stem_s2_fused_fwd_f16(x, o.w, o.scale, o.shift, d.leaky, o.w2, o.scale2, o.shift2, o.leaky2, at(o.dst), d.batch,
                                                p->img_dim, p->img_dim, st);
            case AY_OP_STEM:
                return ay_stem_conv_fwd_f16(x, static_cast<const float*>(o.w), o.scale, o.shift, at(o.dst), d.batch, p->img_dim, p->img_dim,
                                            d.leaky, st);
            case AY_OP_CONV:
                return ay_conv_fwd_f16(&d, at(o.src), o.w, o.scale, o.shift, at(o.res), at(o.dst), st);
            case AY_OP_RESBLOCK:
                return ay_resblock_fwd_f16(at(o.src), o.w, o.scale, o.shift, d.leaky, o.w2, o.scale2, o.shift2, o.leaky2, at(o.dst),
                                           d.batch, d.cin, d.hout, d.wout, st);
            case AY_OP_CONV1X1_CAT:
                return ay_conv1x1_cat_fwd_f16(&d, at(o.src), o.c1, at(o.src2), o.w, o.scale, o.shift, at(o.dst), st);
            default:
                break;  // 16-bit copies and the fp32 decode: the same entry points
        }
    }
    switch (o.kind) {
        case AY_OP_STEM_S2_FUSED:
            return ay_stem_s2_fused_fwd(x, o.w, o.scale, o.shift, d.leaky, o.w2, o.scale2, o.shift2, o.leaky2, at(o.dst), d.batch,
                                        p->img_dim, p->img_dim, st);
        case AY_OP_STEM:
            return ay_stem_conv_fwd(x, static_cast<const float*>(o.w), o.scale, o.shift, at(o.dst), d.batch, p->img_dim, p->img_dim,
                                    d.leaky, st);
        case AY_OP_CONV:
            return ay_conv_fwd_bf16(&d, at(o.src), o.w, o.scale, o.shift, at(o.res), at(o.dst), st);
        case AY_OP_RESBLOCK:
            return ay_resblock_fwd_bf16(at(o.src), o.w, o.scale, o.shift, d.leaky, o.w2, o.scale2, o.shift2, o.leaky2, at(o.dst),
                                        d.batch, d.cin, d.hout, d.wout, st);
        case AY_OP_CONV1X1_CAT:
            return ay_conv1x1_cat_fwd_bf16(&d, at(o.src), o.c1, at(o.src2), o.w, o.scale, o.shift, at(o.dst), st);
        case AY_OP_CONCAT_UPSAMPLE:
            return ay_concat_upsample_bf16(at(o.src), o.c1, o.up1, at(o.src2), o.c2, at(o.dst), d.batch, d.hout, d.wout, st);
        case AY_OP_DECODE:
            return ay_yolo_decode(static_cast<const float*>(at(o.src)), 1, out, d.batch, o.num_anchors, o.num_classes, o.grid,
                                  p->img_dim, o.anchors_wh, p->n_total, o.row_offset, st);
    }
    ay::set_error("ay_plan: unknown op kind %d", o.kind);
    return AY_ERR_ARG;
}

}  // namespace

extern "C" int ay_plan_create(const ay_plan_op* ops, int n_ops, const size_t* value_bytes, int n_values, int img_dim,
                              int n_total_rows, int act_dtype, ay_plan** out_plan) {
    AY_CHECK_ARG(act_dtype == AY_DT_BF16 || act_dtype == AY_DT_F16, "ay_plan_create: activation dtype %d", act_dtype);
    AY_CHECK_ARG(ops && value_bytes && out_plan && n_ops > 0 && n_values > 0, "ay_plan_create: null / empty argument");
    AY_CHECK_ARG(img_dim > 0 && img_dim % 32 == 0 && n_total_rows > 0, "ay_plan_create: image side %d, %d rows", img_dim, n_total_rows);
    ay_plan* p = new ay_plan;
    p->ops.assign(ops, ops + n_ops);
    p->bytes.assign(value_bytes, value_bytes + n_values);
    p->offset.assign(n_values, 0);
    p->def_op.assign(n_values, -1);
    p->last_use.assign(n_values, -1);
    p->img_dim = img_dim;
    p->n_total = n_total_rows;
    p->dtype = act_dtype;
    auto fail = [&](const char* what, int op, int v) {
        ay::set_error("ay_plan_create: op %d: %s (value %d)", op, what, v);
        delete p;
        return AY_ERR_ARG;
    };
    // ---- dataflow ----------------------------------------------------------------------------------------------
    for (int i = 0; i < n_ops; ++i) {
        const ay_plan_op& o = p->ops[i];
        if (o.kind < AY_OP_STEM_S2_FUSED || o.kind > AY_OP_DECODE) return fail("unknown kind", i, o.kind);
        int r[3];
        op_reads(o, r);
        for (int v : r) {
            if (v == AY_PLAN_NONE) continue;
            if (v < 0 || v >= n_values) return fail("read of a value id out of range", i, v);
            if (p->def_op[v] < 0) return fail("value read before it is written", i, v);
            p->last_use[v] = i;
        }
        if (o.kind == AY_OP_DECODE) {
            if (o.num_anchors < 1 || o.num_anchors > 6) return fail("1..6 anchors per head", i, o.num_anchors);
            continue;  // writes the output rows, not a value
        }
        if (o.dst < 0 || o.dst >= n_values) return fail("destination id out of range", i, o.dst);
        if (p->def_op[o.dst] >= 0) return fail("value written twice", i, o.dst);
        for (int v : r)
            if (v == o.dst) return fail("in-place op", i, v);
        p->def_op[o.dst] = i;
        p->last_use[o.dst] = i;
    }
    // ---- detection heads: a linear 1x1 convolution into an fp32 value read by the NEXT op, a decode, and by nothing else ----------
    p->fused.assign(n_ops, 0);
    static const int fuse_heads = getenv("AY_FUSE_HEAD") ? atoi(getenv("AY_FUSE_HEAD")) : 1;
    for (int i = 0; fuse_heads && i + 1 < n_ops; ++i) {
        const ay_plan_op& o = p->ops[i];
        const ay_plan_op& y = p->ops[i + 1];
        if (o.kind == AY_OP_CONV && o.conv.out_f32 && o.conv.ksize == 1 && o.conv.stride == 1 && !o.conv.leaky && o.res == AY_PLAN_NONE &&
            y.kind == AY_OP_DECODE && y.src == o.dst && p->last_use[o.dst] == i + 1 && o.conv.hout == o.conv.wout && y.grid == o.conv.hout &&
            y.num_anchors * (5 + y.num_classes) == o.conv.cout && o.conv.cout_pad == (o.conv.cout + 31) / 32 * 32) {
            p->fused[i] = i + 1;
            p->fused[i + 1] = -1;
        }
    }
    // ---- arena: first fit over lifetimes, in definition order ------------------------------------------------------
    struct Live {
        size_t off, size;
        int until;
    };
    std::vector<Live> live;
    std::vector<int> order;
    for (int v = 0; v < n_values; ++v)
        if (p->def_op[v] >= 0) order.push_back(v);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return p->def_op[a] < p->def_op[b]; });
    for (int v : order) {
        const int born = p->def_op[v];
        // a block is free once its last reader has been ISSUED before this op: stream order makes that safe
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& l) { return l.until < born; }), live.end());
        std::sort(live.begin(), live.end(), [](const Live& a, const Live& b) { return a.off < b.off; });
        const size_t need = (p->bytes[v] + ALIGN - 1) / ALIGN * ALIGN;
        size_t off = 0;
        for (const Live& l : live) {
            if (off + need <= l.off) break;
            off = std::max(off, l.off + l.size);
        }
        p->offset[v] = off;
        live.push_back({off, need, p->last_use[v]});
        p->arena = std::max(p->arena, off + need);
    }
    *out_plan = p;
    return AY_OK;
}

extern "C" void ay_plan_destroy(ay_plan* plan) { delete plan; }

extern "C" size_t ay_plan_workspace_bytes(const ay_plan* plan) { return plan ? plan->arena : 0; }

extern "C" size_t ay_plan_value_offset(const ay_plan* plan, int value) {
    return (plan && value >= 0 && value < (int)plan->offset.size()) ? plan->offset[value] : (size_t)-1;
}

extern "C" int ay_plan_forward(const ay_plan* plan, const float* x_nchw, void* workspace, float* out_rows, ay_stream_t stream) {
    AY_CHECK_ARG(plan && x_nchw && workspace && out_rows, "ay_plan_forward: null argument");
    std::vector<hipEvent_t>* ev = nullptr;
    hipStream_t st = ay::S(stream);
    if (plan->profiling && (plan->prof_seen++ % plan->prof_every) == 0) {  // stream-ordered event records only; elapsed times are read in ay_plan_profile_end
        plan->prof_events.emplace_back(2 * plan->ops.size(), nullptr);
        ev = &plan->prof_events.back();
    }
    for (size_t i = 0; i < plan->ops.size(); ++i) {
        const bool timed = ev && plan->prof_sel[i];
        if (timed) {
            if (hipEventCreate(&(*ev)[2 * i]) != hipSuccess || hipEventCreate(&(*ev)[2 * i + 1]) != hipSuccess) {
                ay::set_error("ay_plan_forward: hipEventCreate failed");
                return AY_ERR_LAUNCH;
            }
            (void)hipEventRecord((*ev)[2 * i], st);
        }
        const int rc = issue(plan, i, x_nchw, static_cast<uint8_t*>(workspace), out_rows, stream);
        if (rc != AY_OK) return rc;  // the entry point has set the message
        if (timed) (void)hipEventRecord((*ev)[2 * i + 1], st);
    }
    return AY_OK;
}

extern "C" int ay_plan_profile_begin_every(ay_plan* plan, const unsigned char* op_selected, int every) {
    AY_CHECK_ARG(plan && !plan->profiling && every >= 1, "ay_plan_profile_begin: null plan / already profiling / every < 1");
    plan->prof_sel.assign(plan->ops.size(), 1);
    if (op_selected) plan->prof_sel.assign(op_selected, op_selected + plan->ops.size());
    plan->prof_every = every;
    plan->prof_seen = 0;
    plan->profiling = true;
    return AY_OK;
}

extern "C" int ay_plan_profile_begin(ay_plan* plan, const unsigned char* op_selected) { return ay_plan_profile_begin_every(plan, op_selected, 1); }

extern "C" int ay_plan_profile_end(ay_plan* plan, float* op_ms_sum, int* n_forwards) {
    AY_CHECK_ARG(plan && plan->profiling && op_ms_sum && n_forwards, "ay_plan_profile_end: not profiling / null argument");
    plan->profiling = false;
    const size_t n = plan->ops.size();
    for (size_t i = 0; i < n; ++i) op_ms_sum[i] = 0.f;
    int rc = AY_OK;
    for (auto& ev : plan->prof_events) {
        for (size_t i = 0; i < n; ++i) {
            if (!ev[2 * i] || !ev[2 * i + 1]) continue;
            float ms = 0.f;
            if (hipEventSynchronize(ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess)
                rc = AY_ERR_LAUNCH;
            op_ms_sum[i] += ms;
        }
        for (auto& e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    *n_forwards = (int)plan->prof_events.size();
    plan->prof_events.clear();
    if (rc != AY_OK) ay::set_error("ay_plan_profile_end: HIP event error");
    return rc;
}

extern "C" int ay_plan_forward_timed(const ay_plan* plan, const float* x_nchw, void* workspace, float* out_rows, float* op_ms,
                                     ay_stream_t stream) {
    AY_CHECK_ARG(plan && x_nchw && workspace && out_rows && op_ms, "ay_plan_forward_timed: null argument");
    hipStream_t st = ay::S(stream);
    const size_t n = plan->ops.size();
    std::vector<hipEvent_t> ev(n + 1, nullptr);
    int rc = AY_OK;
    for (auto& e : ev)
        if (hipEventCreate(&e) != hipSuccess) rc = AY_ERR_LAUNCH;
    if (rc == AY_OK && hipEventRecord(ev[0], st) != hipSuccess) rc = AY_ERR_LAUNCH;
    for (size_t i = 0; i < n && rc == AY_OK; ++i) {
        rc = issue(plan, i, x_nchw, static_cast<uint8_t*>(workspace), out_rows, stream);
        if (rc == AY_OK && hipEventRecord(ev[i + 1], st) != hipSuccess) rc = AY_ERR_LAUNCH;
    }
    if (rc == AY_OK && hipStreamSynchronize(st) != hipSuccess) rc = AY_ERR_LAUNCH;
    for (size_t i = 0; i < n && rc == AY_OK; ++i)
        if (hipEventElapsedTime(&op_ms[i], ev[i], ev[i + 1]) != hipSuccess) rc = AY_ERR_LAUNCH;
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    if (rc == AY_ERR_LAUNCH) ay::set_error("ay_plan_forward_timed: HIP event / stream error");
    return rc;
}

// A stream-ordered fence owned by the library: an event recorded on `stream` and waited for by the same stream (no host wait).
// The events live in a small per-process table keyed by stream and are re-recorded on every call.
extern "C" int ay_stream_fence(ay_stream_t stream) {
    struct Slot {
        hipStream_t st;
        int dev;
        hipEvent_t ev;
    };
    static std::mutex mu;
    static std::vector<Slot> slots;
    hipStream_t st = ay::S(stream);
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const Slot& s : slots)
            if (s.st == st && s.dev == dev) ev = s.ev;
        if (!ev) {
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                ay::set_error("ay_stream_fence: hipEventCreate failed");
                return AY_ERR_LAUNCH;
            }
            slots.push_back({st, dev, ev});
        }
    }
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(st, ev, 0) != hipSuccess) {
        ay::set_error("ay_stream_fence: event record / stream wait failed");
        return AY_ERR_LAUNCH;
    }
    return AY_OK;
}
