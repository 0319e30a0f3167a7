// Weight gradient of a convolution on the MFMA path:
//     dW[co][ci][kh][kw] = sum_{b,oy,ox} dz[b][co][oy][ox] * x[b][ci][oy*s - pad + kh][ox*s - pad + kw]
// (autograd of nn.Conv2d, models.py:33-40).  The contraction runs over PIXELS, while both tensors are stored
// [plane][pixel][16 channels]; v_mfma_f32_16x16x32_bf16 wants, per lane, 8 consecutive k (= pixels) of one row (= channel).
// ds_read_b64_tr_b16 does exactly that transpose on the way out of LDS (4 pixel rows x 16 channels per 16 lanes), so the
// tiles are staged by plain lane-linear LDS-DMA and never re-laid-out (scripts/micro/tr_mfma_test.hip pins the mapping).
//
// Workgroup (8 waves): 8 co planes (128 channels) x 4 ci planes (64 channels) x all taps; wave (cw, iw) owns 4 co planes x
// 1 ci plane x taps = 36 accumulator tiles of 16x16 (3x3).  K step = one 32-pixel segment of one output row: dz tile
// 8 x 32 px, x tile 4 planes x 3 rows x 34 px (65 for stride 2).  Segments are dealt round-robin over a split-K grid
// dimension and the partial filters are added to dW with fp32 atomics (dW is zeroed first).
// LDS slots are XOR-swizzled (slot = px ^ ((px>>3 & 1) << 2)), through the DMA source address, so that the two 4-row blocks
// a 32-lane half reads land on different banks.
#include <stdlib.h>

#include <type_traits>

#include "ay_common.h"

namespace ay {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
    const uint8_t* x;
    const uint8_t* dz;
    float* dw;
    int B, cin, cout, CIP, COP, hin, win, ho, wo;
    int nseg_x, total_segs;
    int ncob, ncib, ks;      // workgroup grid: ncob x ncib filter tiles x ks split-K slices, flattened into blockIdx.x (see the kernel)
    unsigned m_nseg, m_ho;   // floor(2^32 / nseg_x), floor(2^32 / ho) (0xffffffff for a divisor of 1): multiply-high division in the K loop
    float* slab;       // split-K partial filters [gridDim.z][dw_elems] (plain stores, summed in fixed order by wgrad_reduce_kernel);
    size_t dw_elems;   // nullptr: the partial sums are added to dw with fp32 atomics
};

__device__ __forceinline__ int swz(int px) { return px ^ (((px >> 3) & 1) << 2); }
#ifdef AY_PHASE_CLOCK
// instrumented build only: 100 MHz ticks of wave 0 summed over workgroups: [0] stage body (reads + MFMAs + DMA issue), [1] wait for the
// next stage's DMA, [2] barrier, [3] epilogue, [4] K steps, [5] workgroups, [6] prologue
__device__ unsigned long long g_wgrad_ticks[8];
#define WG_TICK(k)                                      \
    if (wave == 0) {                                    \
        const unsigned long long t_ = wall_clock64();   \
        wtk[k] += t_ - wtk_last;                        \
        wtk_last = t_;                                  \
    }
#else
#define WG_TICK(k)
#endif

// SEGS: row segments per K step (and per barrier).  1x1 layers have 4 MFMAs per wave and segment: one segment per step left the
// kernel barrier-bound (173 us per launch on average at B=32 / 1024^2 against ~70 us of HBM time); they take 4 segments per step
// from a ring of 3 stages.
// CO_PL x CI_PL: channel planes (of 16) of dz / x per workgroup.  8 x 4 (128 x 64 filters, 4 co planes per wave) for the body of
// the network; the narrow layers at its ends -- 32 x 3, 64 x 32, 32 x 64 filters over the LARGEST images -- use 4 x 2 (3x3) or 2 x 4
// (1x1) with one plane pair per wave: their launches are bound by memory latency, not by anything a wave does (2048 K steps of
// ~1.4 us with one or two waves multiplying), and the small tile's ring is half the size, so two workgroups share a CU and twice
// the segments are in flight.
template <int KS, int STRIDE, int SEGS = 1, int NBUF = 4, int CO_PL = 8, int CI_PL = 4>
__global__ void __launch_bounds__(512, (CO_PL * CI_PL >= 32 ? 2 : 4)) wgrad_bf16_kernel(WgradArgs a) {
    constexpr int PAD = (KS - 1) / 2;
    constexpr int KK2 = KS * KS;
    constexpr int COW = CO_PL * CI_PL / 8;              // co planes per wave: 8 waves = (CO_PL / COW) x CI_PL
    static_assert(COW >= 1 && (CO_PL / COW) * CI_PL == 8 && CO_PL % COW == 0, "8 waves");
    constexpr int XW = 31 * STRIDE + KS;                // input pixels per row of a segment
    constexpr int DZ_UNITS = CO_PL * 32 * 2;            // 16-byte units
    constexpr int X_UNITS = CI_PL * KS * XW * 2;
    constexpr int DZ_PIECES = DZ_UNITS / 64;
    constexpr int X_PIECES = (X_UNITS + 63) / 64;
    constexpr int NPIECE = DZ_PIECES + X_PIECES;
    constexpr int PW = (NPIECE + 7) / 8;
    constexpr int X_BASE = DZ_PIECES * 1024;
    constexpr int BUF_BYTES = NPIECE * 1024;            // one segment
    constexpr int DUMMY = NBUF * SEGS * BUF_BYTES;
    constexpr int PWS = PW * SEGS;                      // DMA pieces per wave and stage
    constexpr int RING_BYTES = NBUF * SEGS * BUF_BYTES;
    constexpr int STAGING_BYTES = 8 * 16 * 16 * KK2 * 4;   // the epilogue's transposes reuse the stage buffers
    constexpr int LDS_BYTES = (RING_BYTES > STAGING_BYTES ? RING_BYTES : STAGING_BYTES) + 1024;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(NBUF >= 2 && NBUF <= 4, "ring depth");

    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave / CI_PL, iw = wave % CI_PL;
    // Workgroup -> (filter tile, split-K slice).  The ncob x ncib tiles of one slice read the same pixels -- x is shared by the ncob
    // tiles of a ci block, dz by the ncib tiles of a co block -- and the slices sweep the image in step, so the tiles of a slice
    // belong on ONE XCD (one L2) at the same time: workgroups go to XCDs round-robin by their linear id, so id = 8 * idx + xcd
    // takes tile idx % ntiles of slice (idx / ntiles) * 8 + xcd (ks is a multiple of 8 then).  With the plain (x, y, z) grid the four
    // tiles of a slice sat on four XCDs and every byte was fetched from HBM / Infinity Cache once per tile: 4.3 TB/s of traffic at
    // 0.38 of the matrix peak.
    const int ntiles = a.ncob * a.ncib;
    int tile, zslice;
    if ((a.ks & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = idx % ntiles;
        zslice = (idx / ntiles) * 8 + xcd;
    } else {
        tile = blockIdx.x % ntiles;
        zslice = blockIdx.x / ntiles;
    }
    const int cob = tile % a.ncob, cib = tile / a.ncob;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned lds_base = lds_addr_of(lds);

    // segment groups (SEGS consecutive segments) of this workgroup: blockIdx.z, + gridDim.z, ...
    const int seg = zslice;
    const int total_groups = (a.total_segs + SEGS - 1) / SEGS;
    const int nstep = (total_groups - seg + a.ks - 1) / a.ks;
    if (nstep <= 0) return;

    // ---- loader.  Which unit of a tile a lane carries in piece i (plane, row / tap row, pixel slot, channel half) does not depend
    // on the segment: the per-lane part of every address is computed ONCE, a K step adds scalars.  Pieces go through buffer
    // descriptors (dma16_buf): descriptor base = the segment's origin in image b (the top-left corner of its halo: 64-bit scalar
    // arithmetic, may lie before the tensor for the first row -- those lanes are masked), vector offset = the lane's constant
    // (never negative), or 0x80000000 (out of range: reads as zero) for lanes outside the image / beyond the planes.
    // Before, every K step redid the unit decomposition (divisions by the tile width, 64-bit addresses, a branch) per lane and
    // piece: 61 us of a 243-us launch (128->256 3x3 at 128^2, B=16).
    constexpr unsigned OOB = 0x80000000u;
    unsigned lane_off[PW];         // byte offset of the lane's unit from the segment's origin
    int lane_dy[PW], lane_dx[PW];  // x pieces: input row / column of the unit relative to the segment's output origin (may be negative)
    const unsigned x_plane = (unsigned)a.hin * a.win * 32u, dz_plane = (unsigned)a.ho * a.wo * 32u;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int qn = i * 8 + wave;
        lane_off[i] = OOB;
        lane_dy[i] = lane_dx[i] = 0;
        if (qn < DZ_PIECES) {
            const int u = qn * 64 + lane;
            const int half = u & 1, slot = (u >> 1) & 31, pl = u >> 6;
            const int cpl = cob * CO_PL + pl;
            lane_dx[i] = swz(slot);
            if (cpl < a.COP) lane_off[i] = (unsigned)cpl * dz_plane + (unsigned)swz(slot) * 32u + half * 16u;
        } else if (qn < NPIECE) {
            const int u = (qn - DZ_PIECES) * 64 + lane;
            if (u < X_UNITS) {
                const int half = u & 1;
                const int t = u >> 1;
                const int slot = t % XW, sgi = t / XW;  // sgi = ipl * KS + kh
                const int kh = sgi % KS, ipl = sgi / KS;
                const int cpl = cib * CI_PL + ipl;
                lane_dy[i] = kh - PAD;
                lane_dx[i] = swz(slot) - PAD;
                if (cpl < a.CIP) lane_off[i] = (unsigned)cpl * x_plane + (unsigned)((kh * a.win + swz(slot)) * 32) + half * 16u;
            }
        }
    }
    // Source of one segment: wave-uniform.  The coordinates come from two multiply-high divisions (the compiler's expansion of
    // three run-time divisions and two remainders per segment was ~300 scalar instructions per stage and wave, issued in front of
    // the stage's MFMAs: SQ_ACTIVE_INST_SCA 0.20 of the wave cycles, matrix pipe 42 % busy).
    struct SegSrc {
        __amdgpu_buffer_rsrc_t rdz, rx;
        int oy, ox0;
        bool ok;
    };
    auto seg_src = [&](int sg) __attribute__((always_inline)) {
        SegSrc S;
        S.ok = sg < a.total_segs;  // the tail of the last group: every lane out of range, the buffer reads as zeros
        if (!S.ok) sg = 0;
        unsigned row = __umulhi((unsigned)sg, a.m_nseg), xs = (unsigned)sg - row * (unsigned)a.nseg_x;   // row = b * ho + oy
        if (xs >= (unsigned)a.nseg_x) ++row, xs -= (unsigned)a.nseg_x;
        unsigned b = __umulhi(row, a.m_ho), oy = row - b * (unsigned)a.ho;
        if (oy >= (unsigned)a.ho) ++b, oy -= (unsigned)a.ho;
        S.oy = (int)oy;
        S.ox0 = (int)xs * 32;
        const long long org_dz = ((long long)S.oy * a.wo + S.ox0) * 32;
        const long long org_x = ((long long)(S.oy * STRIDE - PAD) * a.win + (S.ox0 * STRIDE - PAD)) * 32;   // negative in the first row
        S.rdz = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.dz) + (long long)b * a.COP * dz_plane + org_dz, 0, 0x7ffffffc, 0x00020000);
        S.rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.x) + (long long)b * a.CIP * x_plane + org_x, 0, 0x7ffffffc, 0x00020000);
        return S;
    };
    auto issue_piece = [&](const SegSrc& S, int i, int buf) __attribute__((always_inline)) {   // piece i (of PW) of this wave
        const int qn = i * 8 + wave;  // wave-uniform piece id
        if (qn < DZ_PIECES) {
            const unsigned vo = (S.ok && S.ox0 + lane_dx[i] < a.wo) ? lane_off[i] : OOB;
            dma16_buf(S.rdz, vo, 0u, lds_base + buf * BUF_BYTES + qn * 1024);
        } else if (qn < NPIECE) {
            const int iy = S.oy * STRIDE + lane_dy[i], ix = S.ox0 * STRIDE + lane_dx[i];
            const unsigned vo = (S.ok && iy >= 0 && iy < a.hin && ix >= 0 && ix < a.win) ? lane_off[i] : OOB;
            dma16_buf(S.rx, vo, 0u, lds_base + buf * BUF_BYTES + qn * 1024);
        } else {
            dma16_buf(S.rx, OOB, 0u, lds_base + DUMMY);   // keeps the per-wave piece count constant (counted vmcnt waits)
        }
    };
    auto issue = [&](int sg, int buf) __attribute__((always_inline)) {
        const SegSrc S = seg_src(sg);
#pragma unroll
        for (int i = 0; i < PW; ++i) issue_piece(S, i, buf);
    };

    // fragment addresses (buffer-relative)
    int aoff[2], boff[KS][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r = 8 * g + 4 * s + q;
        aoff[s] = ((cw * COW) * 32 + swz(r)) * 32 + p * 8;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) boff[kw][s] = X_BASE + ((iw * KS) * XW + swz(r * STRIDE + kw)) * 32 + p * 8;
    }

    f32x4 acc[COW][KK2];
#pragma unroll
    for (int j = 0; j < COW; ++j)
#pragma unroll
        for (int t = 0; t < KK2; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // co planes of this wave that exist (0: the wave has nothing to multiply)
    const int wave_nj = (cib * CI_PL + iw < a.CIP) ? min(max(a.COP - (cob * CO_PL + cw * COW), 0), COW) : 0;

    auto issue_stage = [&](int step, int slot) __attribute__((always_inline)) {
        const int group = seg + step * a.ks;
#pragma unroll
        for (int j = 0; j < SEGS; ++j) issue(group * SEGS + j, slot * SEGS + j);
    };
    auto wait_landed = [&](int ahead) __attribute__((always_inline)) {  // `ahead` stages issued beyond the one that must have landed
        if (NBUF == 4 && ahead >= 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PWS) : "memory");
        else if (ahead >= 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PWS) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // prologue: up to NBUF-1 stages in flight
    int issued = 0;
    for (; issued < NBUF - 1 && issued < nstep; ++issued) issue_stage(issued, issued);
    wait_landed(issued - 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

#ifdef AY_PHASE_CLOCK
    unsigned long long wtk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wtk_last = wall_clock64();
#endif
    int cur = 0, nxt = (NBUF - 1) % NBUF;
    constexpr int NGROUP = SEGS * KK2;   // MFMA groups (one tap of one segment: COW MFMAs) per stage; the stage's PWS pieces are spread over them
    for (int k = 0; k < nstep; ++k) {
        // stage `issued` goes into the slot read in the previous iteration; its DMA pieces are issued one by one BEHIND the MFMA
        // groups of this stage (a piece costs the wave its address selects + the DMA issue; in a burst in front of the MFMAs both
        // waves of a SIMD pay that with the matrix pipe idle)
        const bool do_issue = issued < nstep;
        SegSrc nsrc[SEGS];
        if (do_issue) {
            const int group = seg + issued * a.ks;
#pragma unroll
            for (int j = 0; j < SEGS; ++j) nsrc[j] = seg_src(group * SEGS + j);
        }
        const int slot_ld = nxt;
        if (++nxt == NBUF) nxt = 0;
        // (explicit software pipelining of the fragment reads -- x fragments two tap groups ahead, dz fragments of the next segment in a
        // second buffer -- was measured: +1 % at 18 more VGPRs, which spill once the narrow-layer paths below exist; left to the compiler)
        auto read_frag = [&](const uint8_t* p0, const uint8_t* p1) __attribute__((always_inline)) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
            const short v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bf16x8 f;
            __builtin_memcpy(&f, v, 16);
            return f;
        };
        const uint8_t* Lst = lds + cur * SEGS * BUF_BYTES;
        auto read_b = [&](int u) __attribute__((always_inline)) {   // tap group u of the stage = (segment u / KK2, tap u % KK2)
            const int sj = u / KK2, t = u % KK2, kh = t / KS, kw = t % KS;
            return read_frag(Lst + sj * BUF_BYTES + boff[kw][0] + kh * XW * 32, Lst + sj * BUF_BYTES + boff[kw][1] + kh * XW * 32);
        };
        // NJ: co planes this wave really has (COW, 2, or 0 = none: its ci plane or all its co planes lie beyond the tensor, it only
        // takes part in the DMA and the barriers).  Narrow layers leave most of the 128 x 64 tile empty -- the stem (32 x 3 filters)
        // has ONE wave with two planes -- and MFMAs on zero planes were all of their time (stem 4.1 ms, 32->64 1.1 ms per step).
        auto stage_body = [&](auto njc) __attribute__((always_inline)) {
            constexpr int NJ = decltype(njc)::value;
#pragma unroll
            for (int sj = 0; sj < SEGS; ++sj) {
                bf16x8 af[COW];
#pragma unroll
                for (int j = 0; j < NJ; ++j) af[j] = read_frag(Lst + sj * BUF_BYTES + aoff[0] + j * 1024, Lst + sj * BUF_BYTES + aoff[1] + j * 1024);
#pragma unroll
                for (int t = 0; t < KK2; ++t) {
                    const int u = sj * KK2 + t;
                    if constexpr (NJ > 0) {
                        const bf16x8 bfr = read_b(u);
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bfr, acc[j][t], 0, 0, 0);
                    }
                    if (do_issue) {
#pragma unroll
                        for (int n = 0; n < PWS; ++n)   // constant trip count: lane_off[] must stay in registers
                            if (n >= u * PWS / NGROUP && n < (u + 1) * PWS / NGROUP) issue_piece(nsrc[n / PW], n % PW, slot_ld * SEGS + n / PW);
                    }
                }
            }
        };
        if (wave_nj > 2)
            stage_body(std::integral_constant<int, COW>{});
        else if (wave_nj > 0)
            stage_body(std::integral_constant<int, (COW < 2 ? COW : 2)>{});
        else
            stage_body(std::integral_constant<int, 0>{});
        if (do_issue) ++issued;
        WG_TICK(0)
        if (k + 1 < nstep) {
            wait_landed(issued - (k + 1) - 1);  // stages issued beyond k+1 may stay in flight
            WG_TICK(1)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            WG_TICK(2)
        }
        if (++cur == NBUF) cur = 0;
    }
#ifdef AY_PHASE_CLOCK
    if (wave == 0) wtk[4] += nstep, wtk[5] += 1;
#endif

    // ---- epilogue: the workgroup's partial filters go to its split-K slab (plain contiguous stores) or, without a workspace, to
    // dW by fp32 atomics; either way coalesced.  D of one 16x16 tile: row (co) = 4*(lane>>4) + reg, col (ci) = lane & 15.
    // Adding straight from the accumulators would scatter every wave-instruction over 64 cache lines (measured: 17x
    // slower than contiguous atomics, and it was 90 % of this kernel).  Instead each wave transposes one co plane at a
    // time through its own 9 KiB of LDS into dW order [co][ci 16][tap] (144 contiguous floats per co row of this ci
    // plane) and adds 64 consecutive floats per instruction.
    __builtin_amdgcn_s_barrier();  // every wave is done with the stage buffers
    asm volatile("" ::: "memory");
    float* stage = reinterpret_cast<float*>(lds) + wave * (16 * 16 * KK2);
    const int ci0 = (cib * CI_PL + iw) * 16;
    constexpr int ROW = 16 * KK2;  // floats per co row of one ci plane
#pragma unroll
    for (int j = 0; j < COW; ++j) {
        if (j >= wave_nj) break;   // planes beyond the tensor: nothing to store
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < KK2; ++t) stage[((4 * g + r) * 16 + (lane & 15)) * KK2 + t] = acc[j][t][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int co_base = (cob * CO_PL + cw * COW + j) * 16;
        for (int idx = lane; idx < 16 * ROW; idx += 64) {
            const int col = idx / ROW, rem = idx % ROW;  // rem = ci_local * KK2 + tap
            const int co = co_base + col, ci = ci0 + rem / KK2;
            if (co < a.cout && ci < a.cin) {
                const size_t off = ((size_t)co * a.cin + ci0) * KK2 + rem;
                if (a.slab)
                    a.slab[(size_t)zslice * a.dw_elems + off] = stage[idx];
                else
                    atomicAdd(a.dw + off, stage[idx]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#ifdef AY_PHASE_CLOCK
    WG_TICK(3)
    if (wave == 0 && lane == 0)
        for (int k = 0; k < 6; ++k) atomicAdd(&g_wgrad_ticks[k], wtk[k]);
#endif
}

// dw[i] = (accumulate ? dw[i] : 0) + sum_s slab[s][i] in a FIXED order: the same bits on every run.  Replaces ks x |dW| fp32 atomics
// (~1.3 TB/s chip-wide, 58 us of a 349-us launch at 128->256 3x3, B=32, 128^2) by plain stores + one streaming pass.
// P lanes share one float4 of dW: lane part p sums the slabs p, p + P, p + 2P, ... in ascending order, the P partial sums are combined
// by a butterfly over adjacent lanes (xor 1, 2, ...: a fixed tree).  With one lane per float4 a small filter (32 K floats = 32
// workgroups) walked up to 128 slabs as one chain of dependent adds per thread: 74 launches at 25 us on average, 1.9 ms per step.
template <int P>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, size_t n, int ks, int accumulate) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int part = (int)(t & (P - 1));
    const size_t i = (t / P) * 4;
    const bool live = i + 4 <= n;   // n % 4 == 0 (host-checked)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int s = part; s < ks; s += P) {
            const float4 v = *reinterpret_cast<const float4*>(slab + (size_t)s * n + i);
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
    }
#pragma unroll
    for (int off = 1; off < P; off <<= 1) {   // every lane of the wave takes part (dead lanes carry zeros)
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off);
        acc.w += __shfl_xor(acc.w, off);
    }
    if (live && part == 0) {
        if (accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(dw + i);
            acc.x += o.x, acc.y += o.y, acc.z += o.z, acc.w += o.w;
        }
        *reinterpret_cast<float4*>(dw + i) = acc;
    }
}
// any n (not a multiple of 4): one chain per element
__global__ void __launch_bounds__(256) wgrad_reduce_scalar_kernel(const float* __restrict__ slab, float* __restrict__ dw, size_t n, int ks, int accumulate) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float acc = 0.f;
    for (int s = 0; s < ks; ++s) acc += slab[(size_t)s * n + j];
    dw[j] = accumulate ? dw[j] + acc : acc;
}

constexpr int WGRAD_SEGS_1X1 = 4;
#ifndef AY_WGRAD_WIDE_SEGS
#define AY_WGRAD_WIDE_SEGS 2
#endif
constexpr int WGRAD_SEGS_1X1_WIDE = AY_WGRAD_WIDE_SEGS;                     // the 128 x 128 tile of the 1x1 layers: 16 KiB per segment,
constexpr int WGRAD_NBUF_1X1_WIDE = 8 / AY_WGRAD_WIDE_SEGS;                 // 8 segment buffers (128 KiB) either way
#ifndef AY_WGRAD_SEGS_3X3
#define AY_WGRAD_SEGS_3X3 2
#endif
constexpr int WGRAD_NBUF_3X3 = AY_WGRAD_SEGS_3X3 == 1 ? 4 : (AY_WGRAD_SEGS_3X3 == 2 ? 3 : 2);   // 6 segment buffers (126 KiB) either way
constexpr int WGRAD_SEGS_3X3 = AY_WGRAD_SEGS_3X3;   // 3x3 stride 1: 2 segments per K step from a ring of 3 stages (126 KiB)

// narrow layers (see the kernel template): 3x3 with at most 4 x 2 planes, 1x1 with at most 2 x 4
static bool wgrad_narrow(const ay_conv_desc* d) {
    static const int on = getenv("AY_WGRAD_NARROW") ? atoi(getenv("AY_WGRAD_NARROW")) : 1;
    const int CIP = (d->cin + 15) / 16;
    const int COP = (d->cout_pad > 0 ? d->cout_pad : (d->cout + 15) / 16 * 16) / 16;
    return on && (d->ksize == 3 ? (COP <= 4 && CIP <= 2) : (COP <= 2 && CIP <= 4));
}

// 1x1 layers with at least 128 x 128 filters: a 128 x 128 filter tile (8 x 8 planes, one ci plane and all 8 co planes per wave: 32
// accumulator registers -- a 1x1 has one tap).  dz is then fetched once per 128 input channels instead of once per 64: the 1x1 weight
// gradients are bound by their operand traffic, not by the matrix pipe (256->128 at 128^2, B=32: dz read by 4 ci blocks before).
static bool wgrad_wide1x1(const ay_conv_desc* d) {
    static const int on = getenv("AY_WGRAD_WIDE1") ? atoi(getenv("AY_WGRAD_WIDE1")) : 1;
    const int CIP = (d->cin + 15) / 16;
    const int COP = (d->cout_pad > 0 ? d->cout_pad : (d->cout + 15) / 16 * 16) / 16;
    return on && d->ksize == 1 && COP >= 8 && CIP >= 8;
}

static long long wgrad_split(const ay_conv_desc* d, int* cob, int* cib, long long* total) {
    const int CIP = (d->cin + 15) / 16;
    const int COP = (d->cout_pad > 0 ? d->cout_pad : (d->cout + 15) / 16 * 16) / 16;
    *cob = (COP + 7) / 8;
    *cib = (CIP + 3) / 4;
    if (wgrad_wide1x1(d)) *cib = (CIP + 7) / 8;
    const bool narrow = wgrad_narrow(d);
    if (narrow) *cob = *cib = 1;
    *total = (long long)d->batch * d->hout * ((d->wout + 31) / 32);
    if (d->ksize == 1) {   // K steps = groups of segments
        const int sg = wgrad_wide1x1(d) ? WGRAD_SEGS_1X1_WIDE : WGRAD_SEGS_1X1;
        *total = (*total + sg - 1) / sg;
    }
    if (d->ksize == 3 && d->stride == 1) *total = (*total + WGRAD_SEGS_3X3 - 1) / WGRAD_SEGS_3X3;
    static const int wg_target = getenv("AY_WGRAD_WGS") ? atoi(getenv("AY_WGRAD_WGS")) : 256;
    long long ks = ((narrow ? 2 : 1) * wg_target + (long long)*cob * *cib - 1) / ((long long)*cob * *cib);   // ~1 workgroup per CU overall (narrow: 2) ...
    if (ks > *total / 24) ks = *total / 24;                                            // ... but >= 24 K steps each (pipeline fill, epilogue)
    if (ks > *total) ks = *total;
    if (ks > 65535) ks = 65535;
    if (ks >= 8) ks -= ks % 8;   // whole slices per XCD (see the kernel's workgroup mapping)
    if (ks < 1) ks = 1;
    return ks;
}

}  // namespace ay

extern "C" size_t ay_conv_wgrad_workspace_bytes(const ay_conv_desc* d) {
    if (!d) return 0;
    int cob, cib;
    long long total;
    const long long ks = ay::wgrad_split(d, &cob, &cib, &total);
    return (size_t)ks * d->cout * d->cin * d->ksize * d->ksize * sizeof(float);
}

extern "C" int ay_conv_wgrad_bf16(const ay_conv_desc* d, const void* x_blocked, const void* dz_blocked, float* dw_oihw, ay_stream_t stream) {
    return ay_conv_wgrad_bf16_acc(d, x_blocked, dz_blocked, dw_oihw, 0, stream);
}

extern "C" int ay_conv_wgrad_bf16_acc(const ay_conv_desc* d, const void* x_blocked, const void* dz_blocked, float* dw_oihw, int accumulate,
                                      ay_stream_t stream) {
    return ay_conv_wgrad_bf16_ws(d, x_blocked, dz_blocked, dw_oihw, accumulate, nullptr, 0, stream);
}

extern "C" int ay_conv_wgrad_bf16_ws(const ay_conv_desc* d, const void* x_blocked, const void* dz_blocked, float* dw_oihw, int accumulate,
                                     void* workspace, size_t workspace_bytes, ay_stream_t stream) {
    using namespace ay;
    AY_CHECK_ARG(d && x_blocked && dz_blocked && dw_oihw, "ay_conv_wgrad_bf16: null");
    // cin need not be a multiple of 16: x is read as ceil(cin/16) planes (pad channels must be zero), dW rows ci >= cin are skipped
    AY_CHECK_ARG((d->ksize == 3 && (d->stride == 1 || d->stride == 2)) || (d->ksize == 1 && d->stride == 1), "ay_conv_wgrad_bf16: shape");
    hipStream_t st = S(stream);
    WgradArgs a;
    a.x = (const uint8_t*)x_blocked;
    a.dz = (const uint8_t*)dz_blocked;
    a.dw = dw_oihw;
    a.B = d->batch;
    a.cin = d->cin;
    a.cout = d->cout;
    a.CIP = (d->cin + 15) / 16;
    a.COP = (d->cout_pad > 0 ? d->cout_pad : (d->cout + 15) / 16 * 16) / 16;
    a.hin = d->hin;
    a.win = d->win;
    a.ho = d->hout;
    a.wo = d->wout;
    a.nseg_x = (d->wout + 31) / 32;
    auto magic = [](unsigned dv) { return dv <= 1 ? 0xffffffffu : (unsigned)(0x100000000ULL / dv); };
    a.m_nseg = magic((unsigned)a.nseg_x);
    a.m_ho = magic((unsigned)d->hout);
    const long long total = (long long)d->batch * d->hout * a.nseg_x;
    AY_CHECK_ARG(total > 0 && total < 0x7fffffffLL, "ay_conv_wgrad_bf16: too many segments");
    a.total_segs = (int)total;
    int cob, cib;
    long long total2;
    const long long ks = wgrad_split(d, &cob, &cib, &total2);
    const size_t dw_elems = (size_t)d->cout * d->cin * d->ksize * d->ksize;
    const bool slabs = workspace != nullptr && workspace_bytes >= (size_t)ks * dw_elems * sizeof(float);
    a.slab = slabs ? (float*)workspace : nullptr;
    a.dw_elems = dw_elems;
    // without a workspace the kernel ADDS its split-K partial sums to dw (fp32 atomics): accumulate = 0 starts from zero
    if (!slabs && !accumulate && hipMemsetAsync(dw_oihw, 0, sizeof(float) * dw_elems, st) != hipSuccess) {
        set_error("ay_conv_wgrad_bf16: memset failed");
        return AY_ERR_LAUNCH;
    }
    a.ncob = cob, a.ncib = cib, a.ks = (int)ks;
    AY_CHECK_ARG((long long)cob * cib * ks < 0x7fffffffLL, "ay_conv_wgrad_bf16: grid");
    dim3 grid((unsigned)(cob * cib * ks)), block(512);
    if (wgrad_narrow(d)) {
        if (d->ksize == 3 && d->stride == 1)
            hipLaunchKernelGGL((wgrad_bf16_kernel<3, 1, WGRAD_SEGS_3X3, WGRAD_SEGS_3X3 == 1 ? 4 : 3, 4, 2>), grid, block, 0, st, a);
        else if (d->ksize == 3)
            hipLaunchKernelGGL((wgrad_bf16_kernel<3, 2, 1, 4, 4, 2>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((wgrad_bf16_kernel<1, 1, WGRAD_SEGS_1X1, 3, 2, 4>), grid, block, 0, st, a);
    } else if (wgrad_wide1x1(d))
        hipLaunchKernelGGL((wgrad_bf16_kernel<1, 1, WGRAD_SEGS_1X1_WIDE, WGRAD_NBUF_1X1_WIDE, 8, 8>), grid, block, 0, st, a);
    else if (d->ksize == 3 && d->stride == 1)
        hipLaunchKernelGGL((wgrad_bf16_kernel<3, 1, WGRAD_SEGS_3X3, WGRAD_NBUF_3X3>), grid, block, 0, st, a);
    else if (d->ksize == 3)
        hipLaunchKernelGGL((wgrad_bf16_kernel<3, 2>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((wgrad_bf16_kernel<1, 1, WGRAD_SEGS_1X1, 3>), grid, block, 0, st, a);
    AY_CHECK_LAUNCH("wgrad_bf16_kernel");
#ifdef AY_PHASE_CLOCK
    if (getenv("AY_DBG") && (atoi(getenv("AY_DBG")) & 8)) {
        unsigned long long t[8] = {0}, z[8] = {0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_wgrad_ticks), sizeof(t));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_ticks), z, sizeof(z));
        if (t[4])
            fprintf(stderr, "[ay wgrad] k%d s%d %d->%d @%d ks %lld: per K step (us): body %.2f, dma wait %.2f, barrier %.2f; per workgroup: %.0f steps, epilogue %.1f us\n",
                    d->ksize, d->stride, d->cin, d->cout, d->hout, ks, t[0] * 0.01 / t[4], t[1] * 0.01 / t[4], t[2] * 0.01 / t[4], (double)t[4] / t[5],
                    t[3] * 0.01 / t[5]);
    }
#endif
    if (slabs) {
        if (dw_elems % 4) {
            hipLaunchKernelGGL(wgrad_reduce_scalar_kernel, dim3((unsigned)((dw_elems + 255) / 256)), dim3(256), 0, st, a.slab, dw_oihw, dw_elems, (int)ks,
                               accumulate);
        } else {   // lanes per float4: enough workgroups to fill the chip, no more parts than slabs
            const size_t n4 = dw_elems / 4;
            int P = n4 >= (size_t)256 * 1024 ? 4 : (n4 >= (size_t)64 * 1024 ? 8 : 16);
            while (P > 1 && P > ks) P >>= 1;
            const dim3 rg((unsigned)((n4 * P + 255) / 256));
            if (P == 16)
                hipLaunchKernelGGL(wgrad_reduce_kernel<16>, rg, dim3(256), 0, st, a.slab, dw_oihw, dw_elems, (int)ks, accumulate);
            else if (P == 8)
                hipLaunchKernelGGL(wgrad_reduce_kernel<8>, rg, dim3(256), 0, st, a.slab, dw_oihw, dw_elems, (int)ks, accumulate);
            else if (P == 4)
                hipLaunchKernelGGL(wgrad_reduce_kernel<4>, rg, dim3(256), 0, st, a.slab, dw_oihw, dw_elems, (int)ks, accumulate);
            else if (P == 2)
                hipLaunchKernelGGL(wgrad_reduce_kernel<2>, rg, dim3(256), 0, st, a.slab, dw_oihw, dw_elems, (int)ks, accumulate);
            else
                hipLaunchKernelGGL(wgrad_reduce_kernel<1>, rg, dim3(256), 0, st, a.slab, dw_oihw, dw_elems, (int)ks, accumulate);
        }
        AY_CHECK_LAUNCH("wgrad_reduce_kernel");
    }
    return AY_OK;
}
