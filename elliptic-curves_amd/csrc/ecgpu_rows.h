// ecgpu_rows.h — ROW-PARALLEL k256 field arithmetic for one wave that is alone with a serial chain (device only).
//
// k_msm_combine's Horner chain is c (nwin - 1) dependent doublings (~120 for the GLV plan, ~240 for the plain one) carried by ONE
// wave: its time is its instruction count (a lone wave issues a dependent instruction every ~8.5 cycles, an independent one every
// ~5: tools/repro/lone_wave_latency.hip, profiles/r05/lone_wave_latency.txt).  The per-lane field multiplication is 145 instructions
// whoever executes it; round 4 put the four products of a doubling's level on the four lanes of a quad (569 instructions per
// doubling).  Here a field element is spread over a ROW of 16 lanes, limb i in position i, and the four rows of the wave carry the
// four products of a level: a multiplication is 9 multiply-adds per lane instead of 81, its reduction a handful of steps in which
// every limb moves at once, and every cross-lane step of the reduction is a DPP row shift (no LDS round trip):
//
//   columns    c_p = sum_i a_i b_(p - i), p = 0..15: a_i broadcast inside the row (ds_bpermute, nine in flight), b shifted by i
//              positions (row_shr:i), one v_mad_u64_u32 each; column 16 = a_8 b_8 is computed by every lane of the row
//   stage 1    each 64-bit column in pieces of 29 / 29 / 6 bits, added where they weigh 2^(29 p) (row_shr:1, row_shr:2); positions
//              16..18 are kept in a second register at positions 0..2
//   stage 2    positions 9..18 folded down with 2^261 = F1 2^29 + F0: position j takes F0 limb(j + 9) + F1 limb(j + 8)
//              (row_shl:9, row_shl:8; per-position multipliers)
//   stages 3-5 carry pass, fold of positions 9..10, carries of positions 0..2: nine limbs below the magnitude-1 bound LB
//
// Statement for statement the model tools/rows_field_model.py (register widths asserted at every step, adversarial magnitudes,
// results against Python's integers).  On the device: ecgpu_selftest_field op 16 (products against Field::mul) and
// ecgpu_selftest_point op 10 (a chain of doublings against Group::dbl), tests/test_gpu_selftest.py.
#pragma once

#include "ecgpu_point.h"

namespace ecgpu {

#if defined(__HIPCC__)

template <int N>
__device__ __forceinline__ uint32_t row_shr(uint32_t v) {       // lane p <- lane p - N of its row, 0 below the row's first lane
    static_assert(N >= 1 && N <= 15, "row shift");
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, true);
}
template <int N>
__device__ __forceinline__ uint32_t row_shl(uint32_t v) {       // lane p <- lane p + N of its row, 0 above the row's last lane
    static_assert(N >= 1 && N <= 15, "row shift");
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t lane_pull(uint32_t byte_addr, uint32_t v) {   // lane <- lane byte_addr / 4 (LDS crossbar)
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_addr, (int)v);
}

// per-lane constants of the row-parallel code (a handful of VGPRs, set up once per kernel)
struct RowsK256 {
    using KC = consts::K256U;
    static constexpr uint32_t MASK = (1u << 29) - 1;
    uint32_t pos, row;          // position in the row (0..15), row of the wave (0..3)
    uint32_t rowaddr;           // byte address of the row's first lane for lane_pull
    uint32_t f0p, f1p;          // multipliers of the limbs nine / eight positions up (F0 at positions 0..9, F1 at 1..10)
    uint32_t low9;              // all ones at positions 0..8
    uint32_t e0, e1, e2;        // all ones at position 0 / 1 / 2
    uint32_t keep5, carry5;     // stage 5: positions 0..2 keep 29 bits and pass a carry on, the others stay as they are
    uint32_t big;               // limb `pos` of 33 p (K256U::Z[1]): big - x is limb-wise non-negative for a magnitude-1 x

    __device__ __forceinline__ void init() {
        const uint32_t lane = threadIdx.x & 63u;
        pos = lane & 15u;
        row = lane >> 4;
        rowaddr = (lane & 48u) * 4u;
        f0p = pos <= 9 ? KC::F0 : 0u;
        f1p = (pos >= 1 && pos <= 10) ? KC::F1 : 0u;
        low9 = pos <= 8 ? 0xFFFFFFFFu : 0u;
        e0 = pos == 0 ? 0xFFFFFFFFu : 0u;
        e1 = pos == 1 ? 0xFFFFFFFFu : 0u;
        e2 = pos == 2 ? 0xFFFFFFFFu : 0u;
        keep5 = pos <= 2 ? MASK : 0xFFFFFFFFu;
        carry5 = pos <= 2 ? 0xFFFFFFFFu : 0u;
        uint32_t z = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) z = pos == (uint32_t)i ? KC::Z[1][i] : z;
        big = z;
    }

    // a, b: limbs at positions 0..8 (limb magnitudes ma, mb with ma mb <= 7), zeros above.  -> a b, limbs < LB.
    __device__ __forceinline__ uint32_t mul(uint32_t a, uint32_t b) const {
        uint32_t ai[9];
#pragma unroll
        for (int i = 0; i < 9; i++) ai[i] = lane_pull(rowaddr + 4u * i, a);
        const uint32_t b8 = lane_pull(rowaddr + 32u, b);
        // two accumulators: a lone wave waits ~8.5 cycles for a dependent multiply-add and ~5 for an independent one
        uint64_t c = (uint64_t)ai[0] * b, c_odd = (uint64_t)ai[1] * row_shr<1>(b);
        c += (uint64_t)ai[2] * row_shr<2>(b);
        c_odd += (uint64_t)ai[3] * row_shr<3>(b);
        c += (uint64_t)ai[4] * row_shr<4>(b);
        c_odd += (uint64_t)ai[5] * row_shr<5>(b);
        c += (uint64_t)ai[6] * row_shr<6>(b);
        c_odd += (uint64_t)ai[7] * row_shr<7>(b);
        c += (uint64_t)ai[8] * row_shr<8>(b);
        c += c_odd;
        const uint64_t top = (uint64_t)ai[8] * b8;
        // stage 1
        const uint32_t l = (uint32_t)c & MASK, m = (uint32_t)(c >> 29) & MASK, h = (uint32_t)(c >> 58);
        const uint32_t c1 = l + row_shr<1>(m) + row_shr<2>(h);
        const uint32_t tl = (uint32_t)top & MASK, tm = (uint32_t)(top >> 29) & MASK, th = (uint32_t)(top >> 58);
        const uint32_t t = row_shl<15>(m) + row_shl<14>(h) + ((tl & e0) | (tm & e1) | (th & e2));
        // stage 2
        const uint32_t ha = row_shl<9>(c1) + row_shr<7>(t), hb = row_shl<8>(c1) + row_shr<8>(t);
        uint64_t r = (uint64_t)(c1 & low9);
        r += (uint64_t)f0p * ha;
        r += (uint64_t)f1p * hb;
        // stage 3
        const uint32_t r1 = ((uint32_t)r & MASK) + row_shr<1>((uint32_t)(r >> 29));
        // stage 4
        uint64_t r2 = (uint64_t)(r1 & low9);
        r2 += (uint64_t)f0p * row_shl<9>(r1);
        r2 += (uint64_t)f1p * row_shl<8>(r1);
        // stage 5
        return ((uint32_t)r2 & keep5) + row_shr<1>((uint32_t)(r2 >> 29) & carry5);
    }

    // v: per-position 64-bit values at positions 0..8 (a small linear combination of magnitude-1 elements, below 2^38), zeros
    // above -> the same element with limbs < 2 LB
    __device__ __forceinline__ uint32_t norm64(uint64_t v) const {
        const uint32_t v1 = ((uint32_t)v & MASK) + row_shr<1>((uint32_t)(v >> 29));
        return (v1 & low9) + __umul24(f0p, row_shl<9>(v1)) + __umul24(f1p, row_shl<8>(v1));      // (limb 9 is below 2^10)
    }
};

// The complete doubling (Renes-Costello-Batina 2016, algorithm 9, a = 0) on the four rows of a wave, carried from one doubling
// to the next as the OPERANDS of its first level:  A = Y | Y | Z | X,  B = Y | Z | Z | Y  (row 0 | 1 | 2 | 3).
struct RowsDblK256 {
    RowsK256 k;
    uint32_t a0, a2, a2n, ap, b0, b2, bp;           // per-row coefficients of the linear step
    uint32_t src_t0, src_t2;                        // lane_pull addresses: the same position in row 0 / row 2
    uint32_t src_a1, src_a2, src_b1, src_b2, m_a2, m_b2;   // the next operands from the four products of level 2

    __device__ __forceinline__ void init() {
        k.init();
        constexpr uint32_t b3 = 3 * K256Params::B_SMALL;
        const uint32_t r = k.row, p4 = k.pos * 4u;
        a0 = r >= 2 ? 1u : 0u;
        a2 = r == 0 ? b3 : 0u;
        a2n = r >= 2 ? 3 * b3 : 0u;
        ap = r == 1 ? 1u : 0u;
        b0 = r <= 1 ? 8u : r == 2 ? 1u : 0u;
        b2 = r == 2 ? b3 : 0u;
        bp = r == 3 ? 2u : 0u;
        src_t0 = p4;
        src_t2 = 128u + p4;
        // after level 2 the rows hold Q0 | Q1 | Q2 | Q3 with X = Q3, Y = Q0 + Q2, Z = Q1
        const uint32_t ra1 = r <= 1 ? 0u : r == 2 ? 1u : 3u;        // A = Q0 + Q2 | Q0 + Q2 | Q1 | Q3
        const uint32_t rb1 = (r == 0 || r == 3) ? 0u : 1u;          // B = Q0 + Q2 | Q1 | Q1 | Q0 + Q2
        src_a1 = ra1 * 64u + p4;
        src_b1 = rb1 * 64u + p4;
        src_a2 = 128u + p4;
        src_b2 = 128u + p4;
        m_a2 = r <= 1 ? 0xFFFFFFFFu : 0u;
        m_b2 = (r == 0 || r == 3) ? 0xFFFFFFFFu : 0u;
    }

    // one doubling: (A, B) -> the products of level 2 (returned) and the operands of the next doubling
    __device__ __forceinline__ uint32_t step(uint32_t& A, uint32_t& B) const {
        const uint32_t P = k.mul(A, B);                                  // Y^2 | Y Z | Z^2 | X Y
        const uint32_t t0 = lane_pull(src_t0, P), t2 = lane_pull(src_t2, P);
        const uint32_t nt2 = k.big - t2;                                 // 33 p - Z^2 (positions 9..15: 0 - 0)
        uint64_t va = (uint64_t)a0 * t0;
        va += (uint64_t)a2 * t2;
        va += (uint64_t)a2n * nt2;
        va += (uint64_t)ap * P;
        uint64_t vb = (uint64_t)b0 * t0;
        vb += (uint64_t)b2 * t2;
        vb += (uint64_t)bp * P;
        // 3b Z^2 | Y Z | Y^2 - 9b Z^2 | Y^2 - 9b Z^2     times     8 Y^2 | 8 Y^2 | Y^2 + 3b Z^2 | 2 X Y
        const uint32_t Q = k.mul(k.norm64(va), k.norm64(vb));
        A = lane_pull(src_a1, Q) + (lane_pull(src_a2, Q) & m_a2);
        B = lane_pull(src_b1, Q) + (lane_pull(src_b2, Q) & m_b2);
        return Q;
    }

    // the operands of a first doubling from a point every lane holds (limb magnitudes <= 2, 2, 1): through 48 words of LDS
    __device__ __forceinline__ void enter(uint32_t* lds48, const Fe<9>& X, const Fe<9>& Y, const Fe<9>& Z, uint32_t& A, uint32_t& B) const {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            lds48[i] = X.v[i];
            lds48[16 + i] = Y.v[i];
            lds48[32 + i] = Z.v[i];
        }
        // (every lane writes the same 27 words and reads two of them back: plain program order, the compiler places the wait)
        const uint32_t ea = k.row <= 1 ? 16u : k.row == 2 ? 32u : 0u;   // Y | Y | Z | X
        const uint32_t eb = (k.row == 0 || k.row == 3) ? 16u : 32u;     // Y | Z | Z | Y
        A = lds48[ea + k.pos] & k.low9;                                  // (positions 9..15 of a slot are never written)
        B = lds48[eb + k.pos] & k.low9;
    }
    // the point after the last doubling, in every lane: X = Q3, Y = Q0 + Q2 (limb magnitude 2), Z = Q1
    __device__ __forceinline__ void leave(uint32_t Q, Fe<9>& X, Fe<9>& Y, Fe<9>& Z) const {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            X.v[i] = lane_pull(192u + 4u * i, Q);
            Y.v[i] = lane_pull(4u * i, Q) + lane_pull(128u + 4u * i, Q);
            Z.v[i] = lane_pull(64u + 4u * i, Q);
        }
    }
};

#endif  // __HIPCC__

}  // namespace ecgpu
