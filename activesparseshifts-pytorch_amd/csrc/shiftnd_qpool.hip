// shiftnd_qpool.hip -- quantized shift + average pool in one pass (gfx950 / MI355X): the tail a quantized module that
// emulates a strided depthwise conv attaches to its shift (`_reduction_fn(shift(x))`,
// torchshifts/quantized/modules/shifts.py:19-20 with modules/shifts.py:81-89: avg_pool{N}d(kernel = stride, ceil_mode = True)).
// ATen has no QuantizedCUDA average pool; a composite of float ops on the integer representation takes six passes
// over the tensor (round 2).  Here one thread produces one pooled element: it gathers its window through the
// channel's padding map (arithmetic: canon_shift + fold_index), sums x_int - zero_point, and requantizes exactly like
// ATen's QuantizedCPU kernels (fp32, clamped to the type's range, the scale does not change; ATen rounds with the zero point
// inside -- contiguous 1-D / 2-D -- or outside -- channels-last, 3-D -- the rounding: `requant` picks).  One-byte element types; contiguous tensors; any padding, crop and number of dims.
//
// Reference behaviour restated: kernels/shifts_kernels.h:532-571 (quantized shift: shift = int_repr(w) - w.zero_point,
// fill = x.zero_point); the pool is ATen's.  Roofline: HBM, (1 + 1 / window) bytes per element.
#include <type_traits>

#include "shiftnd_common.hpp"
#include "shiftnd_launch.hpp"

namespace shiftnd {
namespace {

thread_local int g_qpool_tune[2] = {0, 0};  // [0]: 1 = qpool_forward only (no per-channel plane kernel)

struct QPoolParams {
    const void *x;
    void *out;
    const void *w;
    int64_t wzp;
    int32_t xzp, qmin, qmax;
    int wkind, C, nd, pad, zp_outside;
    int S[3], O[3], L[3], K[3], P[3], wcol[3];
    int64_t x_plane, p_plane;
    uint32_t chunks;  // workgroups per (n, c) plane of the pooled output
    FastDiv d_chunks, d_C, d_P2, d_P12;
    FastDiv d_per[3];
};

template <typename EL>
__global__ __launch_bounds__(kThreads) void qpool_forward(const QPoolParams p) {
    const uint32_t plane = fdiv(blockIdx.x, p.d_chunks);
    const uint32_t chunk = blockIdx.x - plane * p.chunks;
    const int c = static_cast<int>(plane - fdiv(plane, p.d_C) * static_cast<uint32_t>(p.C));
    int cs[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0)
            cs[d] = canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd + p.wcol[d]), p.S[d], p.pad, p.d_per[d]);
    const uint32_t t = chunk * kThreads + threadIdx.x;
    if (t >= static_cast<uint32_t>(p.p_plane)) return;
    const int p0 = static_cast<int>(fdiv(t, p.d_P12));
    const uint32_t rem = t - static_cast<uint32_t>(p0) * static_cast<uint32_t>(p.P[1] * p.P[2]);
    const int p1 = static_cast<int>(fdiv(rem, p.d_P2));
    const int p2 = static_cast<int>(rem) - p1 * p.P[2];
    const int n0 = min(p.K[0], p.O[0] - p0 * p.K[0]), n1 = min(p.K[1], p.O[1] - p1 * p.K[1]), n2 = min(p.K[2], p.O[2] - p2 * p.K[2]);
    const EL *xp = static_cast<const EL *>(p.x) + static_cast<int64_t>(plane) * p.x_plane;
    int acc = 0;
    for (int a = 0; a < n0; ++a) {
        const int sa = p.S[0] == 1 ? 0 : fold_index(p0 * p.K[0] + a + p.L[0] - cs[0], p.S[0], p.pad);
        for (int b = 0; b < n1; ++b) {
            const int sb = p.S[1] == 1 ? 0 : fold_index(p1 * p.K[1] + b + p.L[1] - cs[1], p.S[1], p.pad);
            const bool rowok = sa >= 0 && sb >= 0;
            const EL *row = xp + static_cast<int64_t>((rowok ? sa : 0) * p.S[1] + (rowok ? sb : 0)) * p.S[2];
            for (int k = 0; k < n2; ++k) {
                const int sc = p.S[2] == 1 ? 0 : fold_index(p2 * p.K[2] + k + p.L[2] - cs[2], p.S[2], p.pad);
                if (rowok && sc >= 0) acc += static_cast<int>(row[sc]) - p.xzp;  // (fill = zero point: contributes 0)
            }
        }
    }
    // ATen's QuantizedCPU average pool: multiplier = float(in_scale / out_scale / count) = float(1 / count).  Its contiguous
    // 1-D / 2-D kernel requantizes through quantize_val(scale = 1 / multiplier) -> fbgemm::Quantize: nearbyint(zero_point +
    // sum * (1 / scale)), the zero point INSIDE the rounding; its channels-last kernel (every 3-D tensor too) rounds sum *
    // multiplier and adds the zero point afterwards.  Both pinned against torch's CPU kernels by
    // tests/test_quant_convert.py::test_quantized_avg_pool_restatement_matches_aten.
    const float mult = static_cast<float>(1.0 / static_cast<double>(n0 * n1 * n2));
    int q;
    if (p.zp_outside) {
        q = static_cast<int>(nearbyintf(__fmul_rn(static_cast<float>(acc), mult))) + p.xzp;
    } else {
        const float scale = 1.0f / mult;
        const float inv = 1.0f / scale;
        q = static_cast<int>(nearbyintf(__fadd_rn(static_cast<float>(p.xzp), __fmul_rn(static_cast<float>(acc), inv))));  // (no fma)
    }
    q = q < p.qmin ? p.qmin : (q > p.qmax ? p.qmax : q);
    static_cast<EL *>(p.out)[static_cast<int64_t>(plane) * p.p_plane + t] = static_cast<EL>(q);
}


// =====================================================================================================================
// qpool_plane_forward: the same result at stream rate for the planes a quantized network pools (7 x 7 ... 112 x 112).
// One thread per pooled element redoes the padding arithmetic of its window for every (n, c) plane and reads single
// bytes from HBM (0.54 ms for N128 C512 56 x 56: 0.47 TB/s).  But the window of a pooled element is the same set of
// plane offsets for every batch entry of one channel, so a workgroup owns ONE channel and a group of batch entries:
//   * once: every thread works out, for the NI items it owns (item = 4 adjacent pooled elements of a pooled row), the
//     LDS offset of each of the 4 * KV window bytes through the channel's padding map; fill bytes (zeros padding) and the
//     slots a ragged last window does not have point at an LDS byte that holds the zero point, so the sum needs no
//     predicate; plus 1 / count per pooled element;
//   * per round: the planes of `ppw` batch entries stream into LDS as they lie (global_load_lds, 16- or 4-byte pieces;
//     planes that are not whole dwords go through registers); each thread sums its 4 * KV LDS bytes per item,
//     requantizes 4 results and stores one dword.  Several workgroups share a CU (the image of a round is a few KiB), so
//     the loads of one overlap the sums of another.
// Eligibility: one-byte elements, contiguous tensors, plane <= 48 KiB, KV = window bytes in {2, 3, 4, 8, 9},
// NI * KV <= 18.  Everything else stays on qpool_forward.
// =====================================================================================================================
struct QPlaneParams {
    const uint8_t *x;
    uint8_t *out;
    const void *w;
    int64_t wzp;
    int32_t xzp, qmin, qmax;
    int wkind, C, N, nd, pad, zp_outside;
    int S[3], O[3], L[3], K[3], P[3], wcol[3];
    int plane_bytes, pooled_bytes;  // per (n, c)
    int items, groups;              // items per plane = pooled rows * groups; groups = ceil(P2 / 4)
    int ppw, rpw;                   // planes per round, rounds per workgroup
    int vec;                        // staging piece: 16, 4 or 1 bytes
    int pieces;                     // pieces per plane
    int zoff;                       // LDS offset of the zero-point bytes (past the image, which is rounded up to whole wave loads)
    FastDiv d_items, d_groups, d_P1, d_pieces, d_C;
    FastDiv d_per[3];
};

// the window triples served, by their size: 2 = (1, 1, 2), 3 = (1, 1, 3), 4 = (1, 2, 2), 8 = (2, 2, 2), 9 = (1, 3, 3)
template <int KV> struct QWindow {
    static constexpr int K0 = KV == 8 ? 2 : 1;
    static constexpr int K1 = KV == 4 || KV == 8 ? 2 : (KV == 9 ? 3 : 1);
    static constexpr int K2 = KV == 2 || KV == 4 || KV == 8 ? 2 : 3;
};

template <typename EL, int KV, int NI>
__global__ __launch_bounds__(kThreads) void qpool_plane_forward(const QPlaneParams p) {
    constexpr int K0 = QWindow<KV>::K0, K1 = QWindow<KV>::K1, K2 = QWindow<KV>::K2;
    static_assert(K0 * K1 * K2 == KV, "window triple");
    extern __shared__ __attribute__((aligned(16))) unsigned char q_lds[];
    const int t = static_cast<int>(threadIdx.x);
    const uint32_t ng = fdiv(blockIdx.x, p.d_C);
    const int c = static_cast<int>(blockIdx.x - ng * static_cast<uint32_t>(p.C));
    if (t < 16) q_lds[p.zoff + t] = static_cast<unsigned char>(p.xzp);

    // ---- staging of one round: planes n0 .. n0 + ppw - 1 of channel c, as they lie ----------------------------------
    const int total_pieces = p.ppw * p.pieces;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    auto stage = [&](int n0) {
        if (p.vec == 1) {
            for (int q = t; q < total_pieces; q += kThreads) {
                const int slot = static_cast<int>(fdiv(static_cast<uint32_t>(q), p.d_pieces));
                const int pq = q - slot * p.pieces;
                const int n = min(n0 + slot, p.N - 1);
                q_lds[q] = p.x[(static_cast<int64_t>(n) * p.C + c) * p.plane_bytes + pq];
            }
            return;
        }
        for (int base = wave * 64; base < total_pieces; base += kThreads) {
            const int q = min(base + lane, total_pieces - 1);  // (lanes past the end reload the last piece: in bounds)
            const int slot = static_cast<int>(fdiv(static_cast<uint32_t>(q), p.d_pieces));
            const int pq = q - slot * p.pieces;
            const int n = min(n0 + slot, p.N - 1);
            const unsigned char *src = p.x + (static_cast<int64_t>(n) * p.C + c) * p.plane_bytes + pq * p.vec;
            unsigned char *ld = q_lds + base * p.vec;  // wave-uniform: the hardware adds lane * size
            if (p.vec == 16)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)ld, 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)ld, 4, 0, 0);
        }
    };
    const int n_first = static_cast<int>(ng) * p.rpw * p.ppw;
    stage(n_first);

    // ---- once per workgroup: window offsets and 1 / count of the items this thread owns -------------------------------
    int cs[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (p.wcol[d] >= 0)
            cs[d] = canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd + p.wcol[d]), p.S[d], p.pad, p.d_per[d]);
    int off[NI][4][KV];
    float rc[NI][4];
    int slot_of[NI], obyte[NI], ovalid[NI];  // ovalid: how many of the item's 4 pooled elements exist (0 = idle thread)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int item = t + i * kThreads, slot = 0;
        if (NI == 1) {
            slot = static_cast<int>(fdiv(static_cast<uint32_t>(t), p.d_items));
            item = t - slot * p.items;
        }
        const bool live = NI == 1 ? slot < p.ppw : item < p.items;
        if (!live) item = 0, slot = 0;
        const int pr = static_cast<int>(fdiv(static_cast<uint32_t>(item), p.d_groups));
        const int cg = item - pr * p.groups;
        const int p0 = static_cast<int>(fdiv(static_cast<uint32_t>(pr), p.d_P1));
        const int p1 = pr - p0 * p.P[1];
        const int n0 = min(K0, p.O[0] - p0 * K0), n1 = min(K1, p.O[1] - p1 * K1);
        slot_of[i] = slot;
        obyte[i] = pr * p.P[2] + cg * 4;
        ovalid[i] = live ? min(4, p.P[2] - cg * 4) : 0;
        // the map is separable: K0 * K1 row offsets (-1: fill or outside a ragged window), 4 * K2 column offsets
        int rowoff[K0 * K1], coloff[4 * K2];
#pragma unroll
        for (int a = 0; a < K0; ++a) {
            const int sa = p.S[0] == 1 ? 0 : fold_index(p0 * K0 + a + p.L[0] - cs[0], p.S[0], p.pad);
#pragma unroll
            for (int b = 0; b < K1; ++b) {
                const int sb = p.S[1] == 1 ? 0 : fold_index(p1 * K1 + b + p.L[1] - cs[1], p.S[1], p.pad);
                rowoff[a * K1 + b] = (a < n0 && b < n1 && sa >= 0 && sb >= 0) ? slot * p.plane_bytes + (sa * p.S[1] + sb) * p.S[2] : -1;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p2 = cg * 4 + j;
            const int n2 = min(K2, p.O[2] - p2 * K2);  // (<= 0 for a pooled column that does not exist)
            const int cnt = n0 * n1 * max(n2, 1);
            const float mult = static_cast<float>(1.0 / static_cast<double>(cnt));
            rc[i][j] = p.zp_outside ? mult : 1.0f / (1.0f / mult);
#pragma unroll
            for (int k = 0; k < K2; ++k) {
                const int sc = p.S[2] == 1 ? 0 : fold_index(p2 * K2 + k + p.L[2] - cs[2], p.S[2], p.pad);
                coloff[j * K2 + k] = k < n2 ? sc : -1;
            }
#pragma unroll
            for (int ab = 0; ab < K0 * K1; ++ab)
#pragma unroll
                for (int k = 0; k < K2; ++k)
                    off[i][j][ab * K2 + k] = (rowoff[ab] | coloff[j * K2 + k]) < 0 ? p.zoff : rowoff[ab] + coloff[j * K2 + k];
        }
    }
    const int zsum = KV * p.xzp;
    const float zpf = static_cast<float>(p.xzp);
    const bool dword_out = (p.P[2] & 3) == 0 && (p.pooled_bytes & 3) == 0;

    for (int r = 0; r < p.rpw; ++r) {
        const int n0 = n_first + r * p.ppw;
        if (r > 0) {
            __syncthreads();  // everybody has finished reading the previous image
            stage(n0);
        }
        __builtin_amdgcn_s_waitcnt(0);  // our loads into LDS (and LDS writes) have landed ...
        __syncthreads();                // ... everybody's
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = n0 + slot_of[i];
            int qv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int sum = 0;
#pragma unroll
                for (int s = 0; s < KV; ++s) sum += static_cast<int>(reinterpret_cast<const EL *>(q_lds)[off[i][j][s]]);
                // (no clamp: the result is the rounded mean of values of the type, the float error is < 1e-4; the product and the
                // sum are rounded separately, as ATen's are -- never contracted into one fma)
                const float prod = __fmul_rn(static_cast<float>(sum - zsum), rc[i][j]);
                qv[j] = p.zp_outside ? static_cast<int>(nearbyintf(prod)) + p.xzp : static_cast<int>(nearbyintf(__fadd_rn(zpf, prod)));
            }
            if (ovalid[i] > 0 && n < p.N) {
                uint8_t *o = p.out + (static_cast<int64_t>(n) * p.C + c) * p.pooled_bytes + obyte[i];
                if (ovalid[i] == 4 && dword_out) {
                    const uint32_t lo = __builtin_amdgcn_perm(static_cast<uint32_t>(qv[1]), static_cast<uint32_t>(qv[0]), 0x0c0c0400u);
                    const uint32_t hi = __builtin_amdgcn_perm(static_cast<uint32_t>(qv[3]), static_cast<uint32_t>(qv[2]), 0x04000c0cu);
                    *reinterpret_cast<uint32_t *>(o) = lo | hi;
                } else {
                    for (int j = 0; j < ovalid[i]; ++j) o[j] = static_cast<uint8_t>(qv[j]);
                }
            }
        }
    }
}

struct QPlanePlan {
    bool ok = false;
    int KV = 0, NI = 0, ppw = 0, rpw = 0, ngroups = 0, vec = 0, pieces = 0, items = 0, groups = 0, zoff = 0, lds = 0;
};

QPlanePlan qplane_plan(const Geometry &g) {
    QPlanePlan q;
    const int64_t plane = g.S[0] * g.S[1] * g.S[2], pooled = g.P[0] * g.P[1] * g.P[2];
    if (plane < 1 || pooled < 1 || plane > 48 * 1024 || g.N < 1 || g.C < 1 || g.N * g.C >= (1LL << 31)) return q;
    if (g.N * g.C * plane >= (1LL << 40)) return q;
    const int64_t kv = (g.K[0] > 0 ? g.K[0] : 1) * (g.K[1] > 0 ? g.K[1] : 1) * (g.K[2] > 0 ? g.K[2] : 1);
    if (kv != 2 && kv != 3 && kv != 4 && kv != 8 && kv != 9) return q;
    const int64_t k0 = kv == 8 ? 2 : 1, k1 = (kv == 4 || kv == 8) ? 2 : (kv == 9 ? 3 : 1);
    if ((g.K[0] > 0 ? g.K[0] : 1) != k0 || (g.K[1] > 0 ? g.K[1] : 1) != k1) return q;  // (QWindow: the triples compiled)
    q.KV = static_cast<int>(kv);
    q.groups = static_cast<int>((g.P[2] + 3) / 4);
    const int64_t items = g.P[0] * g.P[1] * q.groups;
    int ni = static_cast<int>((items + kThreads - 1) / kThreads);
    ni = ni <= 1 ? 1 : (ni == 2 ? 2 : 4);
    if (items > 4LL * kThreads || ni * q.KV > 18) return q;
    // (round 6, measured and dropped: NI = 4 items per thread spread over five 56 x 56 planes per round -- more bytes in flight per
    //  workgroup -- N128 C512 56x56 uint8 pool 2 0.079 -> 0.112 ms: the offsets' 64 registers cost more waves than the longer rounds return;
    //  and rounds staged global -> registers -> LDS two rounds ahead instead of one LDS-DMA round at a time: 0.0832 vs 0.0836 ms -- the
    //  kernel is bound by its 16 byte reads and ~80 vector instructions per output dword, not by the round trip)
    q.NI = ni;
    q.items = static_cast<int>(items);
    q.ppw = ni == 1 ? static_cast<int>(std::min<int64_t>(kThreads / items, g.N)) : 1;
    q.vec = plane % 16 == 0 ? 16 : (plane % 4 == 0 ? 4 : 1);
    q.pieces = static_cast<int>(plane / q.vec);
    const int total = q.ppw * q.pieces;
    q.zoff = q.vec == 1 ? ((total + 15) / 16) * 16 : ((total + 63) / 64) * 64 * q.vec;
    q.lds = q.zoff + 16;
    if (q.lds > 64 * 1024) return q;
    // batch groups: enough workgroups to fill the chip several times over, as many rounds per workgroup as that allows
    // (the per-workgroup set-up is ~10 rounds' worth of instructions)
    const int64_t rounds = (g.N + q.ppw - 1) / q.ppw;
    const int64_t want_wgs = g_qpool_tune[1] > 0 ? g_qpool_tune[1] : 3072;
    const int64_t want_groups = std::max<int64_t>(1, (want_wgs + g.C - 1) / g.C);
    const int64_t ngroups = std::min<int64_t>(rounds, want_groups);
    q.rpw = static_cast<int>((rounds + ngroups - 1) / ngroups);
    q.ngroups = static_cast<int>((rounds + q.rpw - 1) / q.rpw);
    if (static_cast<int64_t>(q.ngroups) * g.C >= (1LL << 31)) return q;
    q.ok = true;
    return q;
}

template <typename EL, int KV> void launch_qplane(const QPlanePlan &q, const QPlaneParams &p, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(q.ngroups) * static_cast<unsigned>(p.C)), block(kThreads);
    if (q.NI == 1) hipLaunchKernelGGL((qpool_plane_forward<EL, KV, 1>), grid, block, q.lds, st, p);
    else if (q.NI == 2) hipLaunchKernelGGL((qpool_plane_forward<EL, KV, 2>), grid, block, q.lds, st, p);
    else if constexpr (KV <= 4) hipLaunchKernelGGL((qpool_plane_forward<EL, KV, 4>), grid, block, q.lds, st, p);
}

template <typename EL> void launch_qplane_kv(const QPlanePlan &q, const QPlaneParams &p, hipStream_t st) {
    switch (q.KV) {
    case 2: launch_qplane<EL, 2>(q, p, st); break;
    case 3: launch_qplane<EL, 3>(q, p, st); break;
    case 4: launch_qplane<EL, 4>(q, p, st); break;
    case 8: launch_qplane<EL, 8>(q, p, st); break;
    default: launch_qplane<EL, 9>(q, p, st); break;
    }
}

// =====================================================================================================================
// qpool_band_forward (round 4): the same fused pass for planes the plane kernel cannot hold -- 224 x 224 uint8 is 50 KB and
// 12544 pooled elements -- 1-D / 2-D.  A workgroup owns a BAND of pooled rows of one channel and walks the batch: per image it
// stages the band's source rows -- row slot i = the row the map gives for window row band0 * K1 + i, so every padding mode is
// just a different row address -- and each thread sums the windows of up to NI items (4 pooled elements each) through offsets it
// computed once.  Windows as in the plane kernel without the plane-pair forms: KV = 2, 3, 4, 9.
// =====================================================================================================================
struct QBandParams {
    const uint8_t *x;
    uint8_t *out;
    const void *w;
    int64_t wzp;
    int32_t xzp;
    int wkind, C, N, pad, zp_outside, nd;
    int S1, S2, O1, O2, L1, L2, P1, P2, wcol1, wcol2;
    int band, nbands;    // pooled rows per band, bands per plane
    int groups;          // items per pooled row = ceil(P2 / 4)
    int rpw, ngroups;    // images per workgroup, image groups
    int vec, ppr;        // staging piece (16 / 4 / 1 bytes), pieces per source row
    int pitch, zrow;     // LDS bytes per staged row (>= S2 + 1: byte S2 of every row holds the zero point), the all-zero-point row
    FastDiv d_groups, d_ppr, d_C, d_nbands, d_per1, d_per2, d_S2;
};

template <typename EL, int KV, int NI>
__global__ __launch_bounds__(kThreads) void qpool_band_forward(const QBandParams p) {
    constexpr int K1 = QWindow<KV>::K1, K2 = QWindow<KV>::K2;
    static_assert(QWindow<KV>::K0 == 1, "1-D / 2-D windows");
    extern __shared__ __attribute__((aligned(16))) unsigned char q_lds[];
    const int t = static_cast<int>(threadIdx.x);
    uint32_t b = blockIdx.x;
    const int c = static_cast<int>(b - fdiv(b, p.d_C) * static_cast<uint32_t>(p.C));
    b = fdiv(b, p.d_C);
    const int band = static_cast<int>(b - fdiv(b, p.d_nbands) * static_cast<uint32_t>(p.nbands));
    const int ng = static_cast<int>(fdiv(b, p.d_nbands));
    const int pb0 = band * p.band, nb = min(p.band, p.P1 - pb0);   // the band's pooled rows
    // what a window element outside the tensor / the window reads: the zero point -- byte S2 of every staged row (an invalid column)
    // and one whole row of zero points (an invalid row).  Offsets are then row + column, nothing to select: 10 registers per item
    // instead of 16 (4 items: the kernel went from 2 to 3 waves per SIMD)
    for (int q = t; q < p.pitch; q += kThreads) q_lds[p.zrow * p.pitch + q] = static_cast<unsigned char>(p.xzp);
    for (int q = t; q < p.zrow; q += kThreads) q_lds[q * p.pitch + p.S2] = static_cast<unsigned char>(p.xzp);
    int cs1 = 0, cs2 = 0;
    if (p.wcol1 >= 0) cs1 = canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd + p.wcol1), p.S1, p.pad, p.d_per1);
    if (p.wcol2 >= 0) cs2 = canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd + p.wcol2), p.S2, p.pad, p.d_per2);
    auto src_row = [&](int wrow) { return p.S1 == 1 ? 0 : fold_index(wrow + p.L1 - cs1, p.S1, p.pad); };   // window row -> source row (-1: fill)
    const int64_t plane = static_cast<int64_t>(p.S1) * p.S2, pooled = static_cast<int64_t>(p.P1) * p.P2;

    // ---- the pieces this thread stages per image (the same for every image): up to NP ------------------------------------------
    constexpr int NP = 16;   // (host: band * K1 * ppr <= NP * 256)
    const int npieces = nb * K1 * p.ppr;
    int soff[NP];            // byte offset of the piece in the image's plane, or -1
    int doff[NP];            // ... in LDS
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int q = t + k * kThreads;
        const int slot = static_cast<int>(fdiv(static_cast<uint32_t>(q), p.d_ppr)), pq = q - slot * p.ppr;
        const int r = q < npieces ? src_row(pb0 * K1 + slot) : -1;
        soff[k] = r >= 0 ? r * p.S2 + pq * p.vec : -1;
        doff[k] = slot * p.pitch + pq * p.vec;
    }
    auto stage = [&](int n) {
        const unsigned char *xp = p.x + (static_cast<int64_t>(min(n, p.N - 1)) * p.C + c) * plane;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (k * kThreads >= npieces) break;   // (uniform)
            if (soff[k] < 0) continue;
            if (p.vec == 16) {
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<u4 *>(__builtin_assume_aligned(q_lds + doff[k], 16)) = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(xp + soff[k], 16));
            } else if (p.vec == 4) {
                *reinterpret_cast<uint32_t *>(q_lds + doff[k]) = *reinterpret_cast<const uint32_t *>(xp + soff[k]);
            } else {
                q_lds[doff[k]] = xp[soff[k]];
            }
        }
    };
    const int n_first = ng * p.rpw;
    stage(n_first);

    // ---- once per workgroup: the window offsets of the thread's items -----------------------------------------------------------
    int rowoff[NI][K1], coloff[NI][4 * K2];
    float rc[NI][4];
    int obyte[NI], ovalid[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int item = t + i * kThreads;
        const bool live = item < nb * p.groups;
        if (!live) item = 0;
        const int prl = static_cast<int>(fdiv(static_cast<uint32_t>(item), p.d_groups)), cg = item - prl * p.groups;
        const int p1 = pb0 + prl;
        const int n1 = min(K1, p.O1 - p1 * K1);
        obyte[i] = p1 * p.P2 + cg * 4;
        ovalid[i] = live ? min(4, p.P2 - cg * 4) : 0;
#pragma unroll
        for (int bb = 0; bb < K1; ++bb) rowoff[i][bb] = ((bb < n1 && src_row(p1 * K1 + bb) >= 0) ? prl * K1 + bb : p.zrow) * p.pitch;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p2 = cg * 4 + j;
            const int n2 = min(K2, p.O2 - p2 * K2);
            const int cnt = max(n1, 1) * max(n2, 1);
            const float mult = static_cast<float>(1.0 / static_cast<double>(cnt));
            rc[i][j] = p.zp_outside ? mult : 1.0f / (1.0f / mult);
#pragma unroll
            for (int k = 0; k < K2; ++k) {
                const int sc = p.S2 == 1 ? 0 : fold_index(p2 * K2 + k + p.L2 - cs2, p.S2, p.pad);
                coloff[i][j * K2 + k] = (k < n2 && sc >= 0) ? sc : p.S2;
            }
        }
    }
    const int zsum = KV * p.xzp;
    const float zpf = static_cast<float>(p.xzp);
    const bool dword_out = (p.P2 & 3) == 0 && (pooled & 3) == 0;
    for (int r = 0; r < p.rpw; ++r) {
        const int n = n_first + r;
        if (r > 0) {
            __syncthreads();   // everybody has finished reading the previous image's band
            stage(n);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int qv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int sum = 0;
#pragma unroll
                for (int bb = 0; bb < K1; ++bb)
#pragma unroll
                    for (int kk = 0; kk < K2; ++kk) sum += static_cast<int>(reinterpret_cast<const EL *>(q_lds)[rowoff[i][bb] + coloff[i][j * K2 + kk]]);
                const float prod = __fmul_rn(static_cast<float>(sum - zsum), rc[i][j]);   // (ATen's two roundings: qpool_plane_forward)
                qv[j] = p.zp_outside ? static_cast<int>(nearbyintf(prod)) + p.xzp : static_cast<int>(nearbyintf(__fadd_rn(zpf, prod)));
            }
            if (ovalid[i] > 0 && n < p.N) {
                uint8_t *o = p.out + (static_cast<int64_t>(n) * p.C + c) * pooled + obyte[i];
                if (ovalid[i] == 4 && dword_out) {
                    const uint32_t lo = __builtin_amdgcn_perm(static_cast<uint32_t>(qv[1]), static_cast<uint32_t>(qv[0]), 0x0c0c0400u);
                    const uint32_t hi = __builtin_amdgcn_perm(static_cast<uint32_t>(qv[3]), static_cast<uint32_t>(qv[2]), 0x04000c0cu);
                    *reinterpret_cast<uint32_t *>(o) = lo | hi;
                } else {
                    for (int j = 0; j < ovalid[i]; ++j) o[j] = static_cast<uint8_t>(qv[j]);
                }
            }
        }
    }
}

// qpool_band_fast (round 4): the band kernel's inner loop by DWORDS for the common case -- zeros padding, windows two columns wide
// (1 x 2, 2 x 2), output rows of whole windows.  The 8 source bytes of an item's 4 pooled elements are contiguous in the staged
// row whatever the shift, and with 16 zero-point bytes in front of every staged row and 32 behind it the columns the zero padding
// fills need no test either: three aligned ds_read_b32 and two v_alignbyte per row, then one v_dot4 per row and pooled element
// (dot4 of the bytes with 0x00000101 / 0x01010000 = the sum of a byte pair) -- ~45 vector + 6 LDS instructions per item instead of
// ~86 + 16.  A channel whose column shift is beyond +-8 (uniform per workgroup) sums byte by byte.
// PM (round 6): small planes whose ROWS are not whole 16-byte pieces but whose plane is (56 x 56: 196 pieces).  The band form staged
// them as 4-byte row pieces without prefetch, four items per thread of which a 56 x 56 plane fills one: 0.244 ms on N128 C512 56 x 56
// against the plane kernel's byte reads, 0.070.  Here the band is the whole plane, one item per thread (NI = 1): the plane travels as its
// own 16-byte pieces through registers one image ahead and every dword is parked in its padded row (rows of a multiple of 4 bytes: a
// dword never straddles two rows); the row shift moves from the staging to the items' row offsets.
template <typename EL, int K1, int NI, bool PM = false>
__global__ __launch_bounds__(kThreads) void qpool_band_fast(const QBandParams p) {
    constexpr int K2 = 2, KV = K1 * K2, kPadL = 16;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char q_lds[];
    const int t = static_cast<int>(threadIdx.x);
    uint32_t b = blockIdx.x;
    const int c = static_cast<int>(b - fdiv(b, p.d_C) * static_cast<uint32_t>(p.C));
    b = fdiv(b, p.d_C);
    const int band = static_cast<int>(b - fdiv(b, p.d_nbands) * static_cast<uint32_t>(p.nbands));
    const int ng = static_cast<int>(fdiv(b, p.d_nbands));
    const int pb0 = band * p.band, nb = min(p.band, p.P1 - pb0);
    {   // every byte of the tile is the zero point until a row is staged over it: the pads of the rows, the zero-point row
        const uint32_t z = static_cast<uint32_t>(p.xzp & 0xff) * 0x01010101u;
        for (int q = t * 16; q < (p.zrow + 1) * p.pitch; q += kThreads * 16) *reinterpret_cast<u4 *>(__builtin_assume_aligned(q_lds + q, 16)) = u4{z, z, z, z};
    }
    int cs1 = 0, cs2 = 0;
    if (p.wcol1 >= 0) cs1 = canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd + p.wcol1), p.S1, 0, p.d_per1);
    if (p.wcol2 >= 0) cs2 = canon_shift(gather_shift(p.w, p.wkind, p.wzp, static_cast<int64_t>(c) * p.nd + p.wcol2), p.S2, 0, p.d_per2);
    auto src_row = [&](int wrow) { return p.S1 == 1 ? 0 : fold_index(wrow + p.L1 - cs1, p.S1, 0); };
    const int64_t plane = static_cast<int64_t>(p.S1) * p.S2, pooled = static_cast<int64_t>(p.P1) * p.P2;
    constexpr int NP = PM ? 4 : 16;
    const int npieces = PM ? (p.S1 * p.S2) >> 4 : nb * K1 * p.ppr;
    int soff[NP], doff[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int q = t + k * kThreads;
        const int slot = static_cast<int>(fdiv(static_cast<uint32_t>(q), p.d_ppr)), pq = q - slot * p.ppr;
        const int r = q < npieces ? src_row(pb0 * K1 + slot) : -1;
        soff[k] = PM ? (q < npieces ? q * 16 : -1) : (r >= 0 ? r * p.S2 + pq * p.vec : -1);
        doff[k] = slot * p.pitch + kPadL + pq * p.vec;
    }
    auto stage = [&](int n) {
        const unsigned char *xp = p.x + (static_cast<int64_t>(min(n, p.N - 1)) * p.C + c) * plane;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (k * kThreads >= npieces) break;   // (uniform)
            if (soff[k] < 0) continue;
            if (p.vec == 16) *reinterpret_cast<u4 *>(__builtin_assume_aligned(q_lds + doff[k], 16)) = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(xp + soff[k], 16));
            else if (p.vec == 4) *reinterpret_cast<uint32_t *>(q_lds + doff[k]) = *reinterpret_cast<const uint32_t *>(xp + soff[k]);
            else q_lds[doff[k]] = xp[soff[k]];
        }
    };
    // the next image's first pieces travel while this image is summed (16-byte pieces: registers for four per thread -- 1024 pieces,
    // a 224 x 224 band; what is beyond them is staged behind the barrier as before)
    constexpr int NPRE = 4;
    u4 pre[NPRE];
    const bool prefetch = p.vec == 16;
    auto load_pre = [&](int n) {
        const unsigned char *xp = p.x + (static_cast<int64_t>(min(n, p.N - 1)) * p.C + c) * plane;
#pragma unroll
        for (int k = 0; k < NPRE; ++k)   // (unconditional: a thread without a piece re-reads byte 0 of the plane and drops it)
            pre[k] = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(xp + (soff[k] >= 0 ? soff[k] : 0), 16));
    };
    auto store_pre = [&]() {
#pragma unroll
        for (int k = 0; k < NPRE; ++k)
            if (soff[k] >= 0) *reinterpret_cast<u4 *>(__builtin_assume_aligned(q_lds + doff[k], 16)) = pre[k];
    };
    auto scatter_pre = [&]() {   // PM: the four dwords of every prefetched piece into their rows
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            if (k * kThreads >= npieces) break;   // (uniform)
            if (soff[k] < 0) continue;
            const uint32_t w4[4] = {pre[k].x, pre[k].y, pre[k].z, pre[k].w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t o = static_cast<uint32_t>(soff[k] + 4 * d);
                const uint32_t row = fdiv(o, p.d_S2), col = o - row * static_cast<uint32_t>(p.S2);
                *reinterpret_cast<uint32_t *>(q_lds + row * static_cast<uint32_t>(p.pitch) + kPadL + col) = w4[d];
            }
        }
    };
    auto stage_rest = [&](int n) {
        const unsigned char *xp = p.x + (static_cast<int64_t>(min(n, p.N - 1)) * p.C + c) * plane;
#pragma unroll
        for (int k = NPRE; k < NP; ++k) {
            if (k * kThreads >= npieces) break;   // (uniform)
            if (soff[k] < 0) continue;
            *reinterpret_cast<u4 *>(__builtin_assume_aligned(q_lds + doff[k], 16)) = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(xp + soff[k], 16));
        }
    };
    const int n_first = ng * p.rpw;
    __syncthreads();   // the tile is all zero point
    if constexpr (PM) {
        load_pre(n_first);
        scatter_pre();
    } else {
        stage(n_first);
    }

    int rowoff[NI][K1], coff[NI], obyte[NI], ovalid[NI];
    float rc[NI];   // (whole windows along the columns: one count per item)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int item = t + i * kThreads;
        const bool live = item < nb * p.groups;
        if (!live) item = 0;
        const int prl = static_cast<int>(fdiv(static_cast<uint32_t>(item), p.d_groups)), cg = item - prl * p.groups;
        const int p1 = pb0 + prl;
        const int n1 = min(K1, p.O1 - p1 * K1);
        obyte[i] = p1 * p.P2 + cg * 4;
        ovalid[i] = live ? min(4, p.P2 - cg * 4) : 0;
#pragma unroll
        for (int bb = 0; bb < K1; ++bb) {
            const int sr = bb < n1 ? src_row(p1 * K1 + bb) : -1;
            rowoff[i][bb] = (sr >= 0 ? (PM ? sr : prl * K1 + bb) : p.zrow) * p.pitch;   // (PM: the plane lies in LDS row by row)
        }
        coff[i] = cg * (4 * K2) + p.L2 - cs2;   // source column of the item's first window element
        const float mult = static_cast<float>(1.0 / static_cast<double>(max(n1, 1) * K2));
        rc[i] = p.zp_outside ? mult : 1.0f / (1.0f / mult);
    }
    const bool near = cs2 >= -8 && cs2 <= 8;   // (uniform: the pads hold what the zero padding fills)
    const uint32_t sh = static_cast<uint32_t>(p.L2 - cs2) & 3u;   // byte phase of every item's window (uniform)
    const int zsum = KV * p.xzp;
    const float zpf = static_cast<float>(p.xzp);
    const bool dword_out = (p.P2 & 3) == 0 && (pooled & 3) == 0;
    for (int r = 0; r < p.rpw; ++r) {
        const int n = n_first + r;
        if (r > 0) {
            __syncthreads();
            if constexpr (PM) {
                scatter_pre();
            } else if (prefetch) {
                store_pre();
                stage_rest(n);
            } else {
                stage(n);
            }
        }
        __syncthreads();
        if ((PM || prefetch) && r + 1 < p.rpw) load_pre(n + 1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int sums[4] = {0, 0, 0, 0};
            if (near) {
#pragma unroll
                for (int bb = 0; bb < K1; ++bb) {
                    const uint32_t *dp = reinterpret_cast<const uint32_t *>(q_lds + rowoff[i][bb] + ((kPadL + coff[i]) & ~3));
                    const uint32_t d0 = dp[0], d1 = dp[1], d2 = dp[2];
                    const uint32_t a0 = __builtin_amdgcn_alignbyte(d1, d0, sh), a1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
                    if constexpr (std::is_signed<EL>::value) {
                        sums[0] = __builtin_amdgcn_sdot4(static_cast<int>(a0), 0x00000101, sums[0], false);
                        sums[1] = __builtin_amdgcn_sdot4(static_cast<int>(a0), 0x01010000, sums[1], false);
                        sums[2] = __builtin_amdgcn_sdot4(static_cast<int>(a1), 0x00000101, sums[2], false);
                        sums[3] = __builtin_amdgcn_sdot4(static_cast<int>(a1), 0x01010000, sums[3], false);
                    } else {
                        sums[0] = static_cast<int>(__builtin_amdgcn_udot4(a0, 0x00000101u, static_cast<uint32_t>(sums[0]), false));
                        sums[1] = static_cast<int>(__builtin_amdgcn_udot4(a0, 0x01010000u, static_cast<uint32_t>(sums[1]), false));
                        sums[2] = static_cast<int>(__builtin_amdgcn_udot4(a1, 0x00000101u, static_cast<uint32_t>(sums[2]), false));
                        sums[3] = static_cast<int>(__builtin_amdgcn_udot4(a1, 0x01010000u, static_cast<uint32_t>(sums[3]), false));
                    }
                }
            } else {   // a column shift beyond the pads: byte by byte (zeros padding: a column outside the row is the zero point)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int bb = 0; bb < K1; ++bb)
#pragma unroll
                        for (int kk = 0; kk < K2; ++kk) {
                            const int sc = coff[i] + j * K2 + kk;
                            const int at = (sc >= 0 && sc < p.S2) ? kPadL + sc : kPadL + p.S2;
                            sums[j] += static_cast<int>(reinterpret_cast<const EL *>(q_lds)[rowoff[i][bb] + at]);
                        }
            }
            int qv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float prod = __fmul_rn(static_cast<float>(sums[j] - zsum), rc[i]);   // (ATen's two roundings: qpool_plane_forward)
                qv[j] = p.zp_outside ? static_cast<int>(nearbyintf(prod)) + p.xzp : static_cast<int>(nearbyintf(__fadd_rn(zpf, prod)));
            }
            if (ovalid[i] > 0 && n < p.N) {
                uint8_t *o = p.out + (static_cast<int64_t>(n) * p.C + c) * pooled + obyte[i];
                if (ovalid[i] == 4 && dword_out) {
                    const uint32_t lo = __builtin_amdgcn_perm(static_cast<uint32_t>(qv[1]), static_cast<uint32_t>(qv[0]), 0x0c0c0400u);
                    const uint32_t hi = __builtin_amdgcn_perm(static_cast<uint32_t>(qv[3]), static_cast<uint32_t>(qv[2]), 0x04000c0cu);
                    *reinterpret_cast<uint32_t *>(o) = lo | hi;
                } else {
                    for (int j = 0; j < ovalid[i]; ++j) o[j] = static_cast<uint8_t>(qv[j]);
                }
            }
        }
    }
}

struct QBandPlan {
    bool ok = false;
    bool pm = false;     // ... with the whole plane as the band, one item per thread (qpool_band_fast<.., 1, PM>)
    bool fast = false;   // qpool_band_fast: zeros padding, windows two columns wide, output rows of whole windows
    int KV = 0, NI = 0, band = 0, nbands = 0, groups = 0, rpw = 0, ngroups = 0, vec = 0, ppr = 0, pitch = 0, lds = 0;
};

QBandPlan qband_plan(const Geometry &g, const void *x) {
    QBandPlan q;
    if (g.S[0] != 1 || g.K[0] > 1 || g.N < 1 || g.C < 1 || g.P[1] < 1 || g.P[2] < 1) return q;   // 1-D / 2-D
    if (g.S[1] * g.S[2] >= (1LL << 30) || g.N * g.C * g.S[1] * g.S[2] >= (1LL << 40)) return q;
    const int64_t k1 = g.K[1] > 0 ? g.K[1] : 1, k2 = g.K[2] > 0 ? g.K[2] : 1, kv = k1 * k2;
    if (!((k1 == 1 && (k2 == 2 || k2 == 3)) || (k1 == 2 && k2 == 2) || (k1 == 3 && k2 == 3))) return q;
    q.KV = static_cast<int>(kv);
    q.NI = kv == 9 ? 2 : 4;
    q.groups = static_cast<int>((g.P[2] + 3) / 4);
    if (q.groups > q.NI * kThreads) return q;
    q.vec = (g.S[2] % 16 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0) ? 16 : ((g.S[2] % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 4 == 0) ? 4 : 1);
    q.ppr = static_cast<int>(g.S[2] / q.vec);
    q.fast = g.pad == 0 && k2 == 2 && g.O[2] % 2 == 0;
    q.pitch = q.fast ? static_cast<int>(((16 + g.S[2] + 32 + 15) / 16) * 16) : static_cast<int>(((g.S[2] + 1 + 15) / 16) * 16);
    int64_t band = std::min<int64_t>(g.P[1], (q.NI * kThreads) / q.groups);
    // PM: rows that are not whole pieces on a plane that is -- all of it one band of at most 256 items and 1024 pieces
    const int64_t plane_bytes = g.S[1] * g.S[2];
    if (q.fast && g.S[2] % 16 != 0 && g.S[2] % 4 == 0 && plane_bytes % 16 == 0 && plane_bytes <= 16 * 1024 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
        g.P[1] * q.groups <= kThreads && 2 * g.P[1] * q.groups >= kThreads && (g.S[1] + 1) * q.pitch <= 60 * 1024 && g_qpool_tune[0] != 4) {
        // (at least half the threads own an item: a 28 x 28 plane has 56 -- 0.096 ms here against the plane kernel's 0.048, which packs
        //  several planes into a round)
        q.pm = true;
        q.NI = 1;
        q.vec = 16;
        q.ppr = 1;
        q.band = static_cast<int>(g.P[1]);
        q.nbands = 1;
        q.lds = static_cast<int>((g.S[1] + 1) * q.pitch);
        const int64_t want = 4096, ngr = std::min<int64_t>(g.N, std::max<int64_t>(1, (want + g.C - 1) / g.C));
        q.rpw = static_cast<int>((g.N + ngr - 1) / ngr);
        q.ngroups = static_cast<int>((g.N + q.rpw - 1) / q.rpw);
        if (static_cast<int64_t>(q.ngroups) * g.C >= (1LL << 31)) return q;
        q.ok = true;
        return q;
    }
    // the staged rows (+ the row of zero points): at most 16 pieces per thread and 60 KB of LDS
    while (band > 1 && (band * k1 * q.ppr > 16 * kThreads || (band * k1 + 1) * q.pitch > 60 * 1024)) --band;
    if (band * k1 * q.ppr > 16 * kThreads || (band * k1 + 1) * q.pitch > 60 * 1024) return q;
    q.band = static_cast<int>(band);
    q.nbands = static_cast<int>((g.P[1] + band - 1) / band);
    q.lds = static_cast<int>((band * k1 + 1) * q.pitch);
    const int64_t want_wgs = 4096, per_image = g.C * q.nbands;
    const int64_t ngroups = std::min<int64_t>(g.N, std::max<int64_t>(1, (want_wgs + per_image - 1) / per_image));
    q.rpw = static_cast<int>((g.N + ngroups - 1) / ngroups);
    q.ngroups = static_cast<int>((g.N + q.rpw - 1) / q.rpw);
    if (static_cast<int64_t>(q.ngroups) * per_image >= (1LL << 31)) return q;
    q.ok = true;
    return q;
}

template <typename EL> void launch_qband(const QBandPlan &q, const QBandParams &p, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(q.ngroups) * static_cast<unsigned>(p.nbands) * static_cast<unsigned>(p.C)), block(kThreads);
    if (q.fast && q.pm) {
        if (q.KV == 4) hipLaunchKernelGGL((qpool_band_fast<EL, 2, 1, true>), grid, block, q.lds, st, p);
        else hipLaunchKernelGGL((qpool_band_fast<EL, 1, 1, true>), grid, block, q.lds, st, p);
        return;
    }
    if (q.fast) {
        if (q.KV == 4) hipLaunchKernelGGL((qpool_band_fast<EL, 2, 4>), grid, block, q.lds, st, p);
        else hipLaunchKernelGGL((qpool_band_fast<EL, 1, 4>), grid, block, q.lds, st, p);
        return;
    }
    switch (q.KV) {
    case 2: hipLaunchKernelGGL((qpool_band_forward<EL, 2, 4>), grid, block, q.lds, st, p); break;
    case 3: hipLaunchKernelGGL((qpool_band_forward<EL, 3, 4>), grid, block, q.lds, st, p); break;
    case 4: hipLaunchKernelGGL((qpool_band_forward<EL, 4, 4>), grid, block, q.lds, st, p); break;
    default: hipLaunchKernelGGL((qpool_band_forward<EL, 9, 2>), grid, block, q.lds, st, p); break;
    }
}

}  // namespace

void qpool_set_tuning(int knob, int value) {
    if (knob >= 0 && knob < 2) g_qpool_tune[knob] = value;
}

bool qpool_forward_eligible(const Geometry &g, int dtype) {
    if (dtype != SHIFTND_I8 && dtype != SHIFTND_U8) return false;
    const int64_t xe = g.S[0] * g.S[1] * g.S[2], pe = g.P[0] * g.P[1] * g.P[2];
    if (xe < 1 || pe < 1 || xe >= (1LL << 30) || g.N * g.C >= (1LL << 31)) return false;
    const int64_t chunks = (pe + kThreads - 1) / kThreads;
    return g.N * g.C * chunks < (1LL << 31);
}

int qpool_forward(const Geometry &g, int dtype, const void *x, const void *w, int wkind, int64_t wzp, int64_t xzp, int requant, void *out,
                  hipStream_t st) {
    const QPlanePlan qp = qplane_plan(g);
    // (round 6) zeros padding, windows two columns wide, output rows of whole windows: on planes of 8 KiB and more qpool_band_fast's
    // dword reads + v_dot4 sums beat the plane kernel's byte reads (N64 C256 112x112 uint8 pool 2: 0.084 -> 0.070 ms; cut 1/1 0.095 ->
    // 0.089); below that a band is too little work per image and round (56x56: 0.070 vs 0.244 ms).  tools/qpool_route_bench.py;
    // knob 36 = 2: the plane kernel first wherever it serves, 3: the band kernel first
    const QBandPlan qb_first = qband_plan(g, x);
    const bool band_first = qb_first.ok && qb_first.fast &&
                            (g_qpool_tune[0] == 3 || (g_qpool_tune[0] == 0 && (g.S[0] * g.S[1] * g.S[2] >= 8 * 1024 || qb_first.pm)));
    if (qp.ok && g_qpool_tune[0] != 1 && !band_first) {
        QPlaneParams p{};
        p.x = static_cast<const uint8_t *>(x);
        p.out = static_cast<uint8_t *>(out);
        p.w = w;
        p.wzp = wzp;
        p.xzp = static_cast<int32_t>(xzp);
        p.qmin = dtype == SHIFTND_I8 ? -128 : 0;
        p.qmax = dtype == SHIFTND_I8 ? 127 : 255;
        p.wkind = wkind;
        p.zp_outside = requant == SHIFTND_REQUANT_ZP_OUTSIDE ? 1 : 0;
        p.C = static_cast<int>(g.C);
        p.N = static_cast<int>(g.N);
        p.nd = g.nd;
        p.pad = g.pad;
        for (int d = 0; d < 3; ++d) {
            p.S[d] = static_cast<int>(g.S[d]);
            p.O[d] = static_cast<int>(g.O[d]);
            p.L[d] = static_cast<int>(g.L[d]);
            p.K[d] = static_cast<int>(g.K[d] > 0 ? g.K[d] : 1);
            p.P[d] = static_cast<int>(g.P[d]);
            p.wcol[d] = g.wcol[d];
            p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
        }
        p.plane_bytes = static_cast<int>(g.S[0] * g.S[1] * g.S[2]);
        p.pooled_bytes = static_cast<int>(g.P[0] * g.P[1] * g.P[2]);
        p.items = qp.items;
        p.groups = qp.groups;
        p.ppw = qp.ppw;
        p.rpw = qp.rpw;
        p.vec = qp.vec;
        p.pieces = qp.pieces;
        p.zoff = qp.zoff;
        p.d_items = make_fastdiv(static_cast<uint32_t>(qp.items));
        p.d_groups = make_fastdiv(static_cast<uint32_t>(qp.groups));
        p.d_P1 = make_fastdiv(static_cast<uint32_t>(g.P[1]));
        p.d_pieces = make_fastdiv(static_cast<uint32_t>(qp.pieces));
        p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
        note_kernel("qpool_plane_forward");
        if (dtype == SHIFTND_I8) launch_qplane_kv<int8_t>(qp, p, st);
        else launch_qplane_kv<uint8_t>(qp, p, st);
        return SHIFTND_OK;
    }
    const QBandPlan qb = qband_plan(g, x);
    if (qb.ok && g_qpool_tune[0] != 1) {   // planes beyond the plane kernel: bands of pooled rows (round 4)
        QBandParams p{};
        p.x = static_cast<const uint8_t *>(x);
        p.out = static_cast<uint8_t *>(out);
        p.w = w;
        p.wzp = wzp;
        p.xzp = static_cast<int32_t>(xzp);
        p.wkind = wkind;
        p.zp_outside = requant == SHIFTND_REQUANT_ZP_OUTSIDE ? 1 : 0;
        p.C = static_cast<int>(g.C);
        p.N = static_cast<int>(g.N);
        p.nd = g.nd;
        p.pad = g.pad;
        p.S1 = static_cast<int>(g.S[1]);
        p.S2 = static_cast<int>(g.S[2]);
        p.O1 = static_cast<int>(g.O[1]);
        p.O2 = static_cast<int>(g.O[2]);
        p.L1 = static_cast<int>(g.L[1]);
        p.L2 = static_cast<int>(g.L[2]);
        p.P1 = static_cast<int>(g.P[1]);
        p.P2 = static_cast<int>(g.P[2]);
        p.wcol1 = g.wcol[1];
        p.wcol2 = g.wcol[2];
        p.band = qb.band;
        p.nbands = qb.nbands;
        p.groups = qb.groups;
        p.rpw = qb.rpw;
        p.ngroups = qb.ngroups;
        p.vec = qb.vec;
        p.ppr = qb.ppr;
        p.pitch = qb.pitch;
        p.zrow = qb.pm ? static_cast<int>(g.S[1]) : qb.band * static_cast<int>(g.K[1] > 0 ? g.K[1] : 1);
        p.d_S2 = make_fastdiv(static_cast<uint32_t>(g.S[2]));
        p.d_groups = make_fastdiv(static_cast<uint32_t>(qb.groups));
        p.d_ppr = make_fastdiv(static_cast<uint32_t>(qb.ppr));
        p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
        p.d_nbands = make_fastdiv(static_cast<uint32_t>(qb.nbands));
        p.d_per1 = make_fastdiv(static_cast<uint32_t>(map_period(p.S1, g.pad)));
        p.d_per2 = make_fastdiv(static_cast<uint32_t>(map_period(p.S2, g.pad)));
        note_kernel(qb.fast ? "qpool_band_fast" : "qpool_band_forward");
        if (dtype == SHIFTND_I8) launch_qband<int8_t>(qb, p, st);
        else launch_qband<uint8_t>(qb, p, st);
        return SHIFTND_OK;
    }
    QPoolParams p{};
    p.x = x;
    p.out = out;
    p.w = w;
    p.wzp = wzp;
    p.xzp = static_cast<int32_t>(xzp);
    p.qmin = dtype == SHIFTND_I8 ? -128 : 0;
    p.qmax = dtype == SHIFTND_I8 ? 127 : 255;
    p.wkind = wkind;
    p.zp_outside = requant == SHIFTND_REQUANT_ZP_OUTSIDE ? 1 : 0;
    p.C = static_cast<int>(g.C);
    p.nd = g.nd;
    p.pad = g.pad;
    for (int d = 0; d < 3; ++d) {
        p.S[d] = static_cast<int>(g.S[d]);
        p.O[d] = static_cast<int>(g.O[d]);
        p.L[d] = static_cast<int>(g.L[d]);
        p.K[d] = static_cast<int>(g.K[d] > 0 ? g.K[d] : 1);
        p.P[d] = static_cast<int>(g.P[d]);
        p.wcol[d] = g.wcol[d];
        p.d_per[d] = make_fastdiv(static_cast<uint32_t>(map_period(p.S[d], g.pad)));
    }
    p.x_plane = g.S[0] * g.S[1] * g.S[2];
    p.p_plane = g.P[0] * g.P[1] * g.P[2];
    p.chunks = static_cast<uint32_t>((p.p_plane + kThreads - 1) / kThreads);
    p.d_chunks = make_fastdiv(p.chunks);
    p.d_C = make_fastdiv(static_cast<uint32_t>(g.C));
    p.d_P2 = make_fastdiv(static_cast<uint32_t>(g.P[2]));
    p.d_P12 = make_fastdiv(static_cast<uint32_t>(g.P[1] * g.P[2]));
    note_kernel("qpool_forward");
    const dim3 grid(static_cast<unsigned>(g.N * g.C * p.chunks)), block(kThreads);
    if (dtype == SHIFTND_I8) hipLaunchKernelGGL((qpool_forward<int8_t>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((qpool_forward<uint8_t>), grid, block, 0, st, p);
    return SHIFTND_OK;
}

}  // namespace shiftnd
