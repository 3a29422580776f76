// shiftnd_stage.hpp -- reading staged rows back from LDS: the column state of a 16-byte chunk and the E + 1 mapped
// elements of one staged row.  Shared by the LDS-staged kernel families (shiftnd_plane.hip, shiftnd_step.hip).
#pragma once

#include "shiftnd_common.hpp"

namespace shiftnd {
namespace {

// Column state of a chunk for LDS reads.  `affine`: every valid column satisfies cm[e] == base + e (true for all
// interior chunks and for the edge chunks of zeros padding): the E + 1 values are then read as consecutive
// dwords from one base address (compile-time offsets -> ds_read2_b32) and masked, instead of E + 1 independent
// element reads.
template <int E> struct ColState {
    int cm[E + 1];
    int base;     // element index of column 0 when affine
    bool affine;
};
template <int E> __device__ __forceinline__ ColState<E> make_colstate(const int *map, int j0, bool live, bool allow_affine) {
    ColState<E> c;
    c.base = 0;
    bool found = false;  // (static indexing only: a runtime-indexed cm[] would live in scratch)
#pragma unroll
    for (int e = 0; e <= E; ++e) {
        c.cm[e] = live ? map[j0 + e] : -1;
        if (!found && c.cm[e] >= 0) {
            c.base = c.cm[e] - e;
            found = true;
        }
    }
    c.affine = allow_affine;
#pragma unroll
    for (int e = 0; e <= E; ++e) c.affine = c.affine && (c.cm[e] < 0 || c.cm[e] == c.base + e);
    return c;
}

// E + 1 raw elements of one staged row (masked columns -> 0).  `row` points at the row's first byte in LDS; a
// 64-byte pad in front of the tile keeps the few bytes an edge chunk reads before column 0 inside the allocation.
template <typename S, int E>
__device__ __forceinline__ void lds_read_row(const char *row, bool valid, const ColState<E> &c, S (&raw)[E + 1]) {
    S zero;
    __builtin_memset(&zero, 0, sizeof(S));
    if (!valid) {
#pragma unroll
        for (int e = 0; e <= E; ++e) raw[e] = zero;
        return;
    }
    if (c.affine) {
        if constexpr (sizeof(S) == 2) {
            // 18 bytes starting at a 2-byte boundary: five dwords, then a funnel shift by 0 or 16 bits
            const int byte0 = c.base * 2;
            const uint32_t *dwp = reinterpret_cast<const uint32_t *>(row + (byte0 & ~3));
            const uint32_t sh = (byte0 & 2) ? 16u : 0u;
            uint32_t dw[6];
#pragma unroll
            for (int i = 0; i < 5; ++i) dw[i] = dwp[i];
            dw[5] = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const uint32_t t = __builtin_amdgcn_alignbit(dw[i + 1], dw[i], sh);  // v_alignbit_b32 (a 64-bit shift is quarter rate)
                const uint16_t lo = static_cast<uint16_t>(t), hi = static_cast<uint16_t>(t >> 16);
                if (2 * i <= E) __builtin_memcpy(&raw[2 * i], &lo, 2);
                if (2 * i + 1 <= E) __builtin_memcpy(&raw[2 * i + 1], &hi, 2);
            }
        } else {
            const S *p0 = reinterpret_cast<const S *>(row) + c.base;
#pragma unroll
            for (int e = 0; e <= E; ++e) raw[e] = p0[e];
        }
#pragma unroll
        for (int e = 0; e <= E; ++e) raw[e] = c.cm[e] >= 0 ? raw[e] : zero;
    } else {
        // unconditional reads at a clamped column, then one select each: no per-element execution-mask juggling
        const S *p0 = reinterpret_cast<const S *>(row);
        S tmp[E + 1];
#pragma unroll
        for (int e = 0; e <= E; ++e) tmp[e] = p0[c.cm[e] > 0 ? c.cm[e] : 0];
#pragma unroll
        for (int e = 0; e <= E; ++e) raw[e] = c.cm[e] >= 0 ? tmp[e] : zero;
    }
}

// lds_read_row for a chunk whose column state is KNOWN to be affine (zeros padding: every chunk), without control flow: the reads
// are unconditional (an affine state's base always points into the row's slot), "row valid" and "column valid" are one integer
// test per element.  Round 4: the two data-dependent branches of lds_read_row cost two or three exec-mask regions per call -- a
// dozen scalar instructions -- and the one-step kernels that call it three or four times per workgroup are partly bound by the
// CU's scalar unit.  (Going further -- validity as integer masks and-ed into the loaded bits, no lane conditions at
// all -- removed another 60 scalar instructions per wave and was SLOWER: 40 more vector instructions; c2acrop backward 1.76 -> 1.89 ms.)
template <typename S, int E>
__device__ __forceinline__ void lds_read_row_affine(const char *row, bool valid, const ColState<E> &c, S (&raw)[E + 1]) {
    S zero;
    __builtin_memset(&zero, 0, sizeof(S));
    const int dead = valid ? 0 : -1;
    if constexpr (sizeof(S) == 2) {
        const int byte0 = c.base * 2;
        const uint32_t *dwp = reinterpret_cast<const uint32_t *>(row + (byte0 & ~3));
        const uint32_t sh = (byte0 & 2) ? 16u : 0u;
        uint32_t dw[6];
#pragma unroll
        for (int i = 0; i < 5; ++i) dw[i] = dwp[i];
        dw[5] = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint32_t t = __builtin_amdgcn_alignbit(dw[i + 1], dw[i], sh);
            const uint16_t lo = static_cast<uint16_t>(t), hi = static_cast<uint16_t>(t >> 16);
            if (2 * i <= E) __builtin_memcpy(&raw[2 * i], &lo, 2);
            if (2 * i + 1 <= E) __builtin_memcpy(&raw[2 * i + 1], &hi, 2);
        }
    } else {
        const S *p0 = reinterpret_cast<const S *>(row) + c.base;
#pragma unroll
        for (int e = 0; e <= E; ++e) raw[e] = p0[e];
    }
#pragma unroll
    for (int e = 0; e <= E; ++e) raw[e] = (c.cm[e] | dead) >= 0 ? raw[e] : zero;
}

// The six dwords of a staged row starting at dword (byte0 >> 2), read as the two aligned 16-byte spans that hold them.
// Five or six ds_read_b32 at a lane stride of 16 bytes (neighbouring chunks) are 4- to 8-way bank conflicts -- half of the
// LDS time of the 16-bit one-step kernels (SQ_LDS_BANK_CONFLICT); two ds_read_b128 at 16-byte aligned addresses are
// conflict-free.  Which dword of the span the window starts with is `ph >> 2` with ph = byte0 & 15, the same for every
// thread of a workgroup whose chunks share one column shift (row bytes are a multiple of 16): a uniform switch into
// compile-time register naming.  `row` is 16-byte aligned; the tile's 64-byte pads take the spans of the edge chunks.
template <int RR> __device__ __forceinline__ void lds_span_pick(const uint32_t (&d)[8], uint32_t (&o)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = RR + i < 8 ? d[RR + i < 8 ? RR + i : 7] : 0u;
}
__device__ __forceinline__ void lds_window6(const char *row, int byte0, int ph, uint32_t (&o)[6]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const char *p = row + (byte0 & ~15);
    const u4 q0 = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(p, 16));
    const u4 q1 = *reinterpret_cast<const u4 *>(__builtin_assume_aligned(p + 16, 16));
    const uint32_t d[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    switch (ph >> 2) {
    case 0: lds_span_pick<0>(d, o); break;
    case 1: lds_span_pick<1>(d, o); break;
    case 2: lds_span_pick<2>(d, o); break;
    default: lds_span_pick<3>(d, o); break;
    }
}

// lds_read_row with the span reader: `fast` = this chunk's map is affine and its window sits at the workgroup's phase `ph`
// (uniform); other chunks, and invalid rows, take lds_read_row's paths
template <typename S, int E>
__device__ __forceinline__ void lds_read_row_span(const char *row, bool valid, const ColState<E> &c, bool fast, int ph, S (&raw)[E + 1]) {
    if (!valid || !fast) {
        ColState<E> slow = c;
        slow.affine = false;
        lds_read_row<S, E>(row, valid, slow, raw);
        return;
    }
    S zero;
    __builtin_memset(&zero, 0, sizeof(S));
    uint32_t o[6];
    lds_window6(row, c.base * static_cast<int>(sizeof(S)), ph, o);
    if constexpr (sizeof(S) == 2) {
        const uint32_t sh = (ph & 2) ? 16u : 0u;   // uniform
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const uint32_t t = __builtin_amdgcn_alignbit(o[i + 1], o[i], sh);
            const uint16_t lo = static_cast<uint16_t>(t), hi = static_cast<uint16_t>(t >> 16);
            if (2 * i <= E) __builtin_memcpy(&raw[2 * i], &lo, 2);
            if (2 * i + 1 <= E) __builtin_memcpy(&raw[2 * i + 1], &hi, 2);
        }
    } else if constexpr (sizeof(S) == 4) {
#pragma unroll
        for (int e = 0; e <= E; ++e) __builtin_memcpy(&raw[e], &o[e], 4);
    } else {
#pragma unroll
        for (int e = 0; e <= E; ++e) {
            const uint64_t q = static_cast<uint64_t>(o[2 * e]) | (static_cast<uint64_t>(o[2 * e + 1]) << 32);
            __builtin_memcpy(&raw[e], &q, 8);
        }
    }
#pragma unroll
    for (int e = 0; e <= E; ++e) raw[e] = c.cm[e] >= 0 ? raw[e] : zero;
}

}  // namespace
}  // namespace shiftnd
