// Final pairing check kernel:  e(left, s_g2) * e(right, -g2) == 1  (DualMSM::check,
// poly/kzg/msm.rs:185-203).
//
// One WAVE per check.  A pairing is a strictly sequential chain of ~480 Fq12 products (Miller loop over 6x+2,
// then the final exponentiation); on a single lane that is ~20 000 dependent Fq products, and a batch ends with exactly
// one of them.  Here an Fq12 element lives in LDS as six Fq2 coefficients of Fq2[w]/(w^6 - xi), and each product is spread
// over the wave: a lane computes ONE coordinate of one a_i * b_j as a single-pass sum of two Fq products (the factor xi of
// the pairs with i + j >= 6 is applied to the operand by an integer combination on the limbs), the partial products go
// through LDS, and twelve lanes add six of them each.  No modular additions, no Karatsuba, no Montgomery pass in the fold.
//
// The G2 side is constant per context, so its Miller-loop line coefficients are precomputed once on the host
// (g2_prepare); the wave evaluates them at the two G1 points up front, all lines in parallel, into LDS.
// The G1 points are used projectively (line values scaled by Z^3 in Fq*, which the final exponentiation kills),
// so no field inversion is needed for them.
//
// SingleStrategy (one check per proof) launches one such wave per proof.
#include "../../include/h2v.h"
#include "internal.h"
#include "pairing.hip.h"
#include "pairing_api.h"

namespace h2v {

#define N_LINES 102  // 64 doublings + popcount(ATE_LOW) = 36 additions + 2 Frobenius corrections

struct WaveShared {
    // step i of the Miller loop multiplies f by l0_i(P0) * l1_i(P1).  The two sparse lines (coefficients of w^0, w^1, w^3,
    // evaluated at their point) are written here first, then replaced in place by their 6-coefficient product.
    Fq2 line[N_LINES][6];
    Fq2 prod[36];
    Fq2 f[6], r[6], t0[6], t1[6], t2[6], t3[6], t4[6], t5[6], t6[6];
};

// The w^6 = xi = 9 + u reduction is applied to an OPERAND, not to the products: lane (i, j) with i + j >= 6 multiplies a_i by
// xi * b_j = (9 b0 - b1) + (9 b1 + b0) u, an integer combination of two residues on the limbs brought below 2p by
// Fp::from_wide (no Montgomery pass), so that every coefficient of the result is a PLAIN sum of six partial products.
// (Computing xi * b once per product in a phase of its own was tried: the extra barrier costs more than the redundancy.)
// One coordinate of the Fq2 product a * b (or a * xi b), one reduction pass (Fp::dot2_inl, result < 1.05p):
//   real      = a0 * B0 + a1 * (-B1)        imaginary = a0 * B1 + a1 * B0        with (B0, B1) = b or xi b
__device__ __noinline__ Fq fq2_mul_coord(Fq2 a, Fq2 b, uint32_t coord, bool times_xi) {
    Fq s0, s1;
    if (!times_xi) {
        s0 = coord ? b.c1 : b.c0;
        s1 = coord ? b.c0 : b.c1.neg();
    } else {
        // B0 = 9 b0 + (2p - b1);  B1 = 9 b1 + b0;  -B1 = 9 (2p - b1) + (2p - b0).  All positive, < 21p.
        int64_t t0[9], t1[9];
#pragma unroll
        for (int l = 0; l < 9; ++l) {
            const int64_t p2 = 2 * (int64_t)FqParams::P29(l), b0 = b.c0.v[l], b1 = b.c1.v[l];
            const int64_t B0 = 9 * b0 + (p2 - b1), B1 = 9 * b1 + b0, nB1 = 9 * (p2 - b1) + (p2 - b0);
            t0[l] = coord ? B1 : B0;
            t1[l] = coord ? B0 : nB1;
        }
        s0 = Fq::from_wide(t0);
        s1 = Fq::from_wide(t1);
    }
    return Fq::dot2_inl(a.c0, s0, a.c1, s1);
}
// One Fq coordinate of coefficient k of the product: the sum of the six partial products P_ij with i + j = k (mod 6) — the
// xi factor already sits in the ones with i + j >= 6 — added on the limbs (six 29-bit limbs cannot overflow 32 bits) and
// reduced below 2p by Fp::from_wide.
__device__ __noinline__ Fq wfold_coord(const Fq2* __restrict__ prod, uint32_t k, uint32_t c) {
    uint32_t sum[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) sum[l] = 0;
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i) {
        const Fq2& P = prod[i * 6 + (i <= k ? k - i : k + 6 - i)];
        const Fq& v = c ? P.c1 : P.c0;
#pragma unroll
        for (int l = 0; l < 9; ++l) sum[l] += v.v[l];
    }
    int64_t acc[9];
#pragma unroll
    for (int l = 0; l < 9; ++l) acc[l] = (int64_t)sum[l];
    return Fq::from_wide(acc);
}
__device__ __forceinline__ void prod_store(Fq2* prod, uint32_t idx, uint32_t coord, const Fq& v) { if (coord) prod[idx].c1 = v; else prod[idx].c0 = v; }
__device__ __forceinline__ void wfold(WaveShared& s, Fq2* dst, uint32_t lane) {
    __syncthreads();
    if (lane < 12) {
        const Fq r = wfold_coord(s.prod, lane >> 1, lane & 1);
        if (lane & 1) dst[lane >> 1].c1 = r; else dst[lane >> 1].c0 = r;
    }
    __syncthreads();
}
// dst = x * y in Fq2[w]/(w^6 - xi); dst may alias x or y.  36 lanes, two passes each.
__device__ __forceinline__ void wmul(WaveShared& s, Fq2* dst, const Fq2* x, const Fq2* y, uint32_t lane) {
    if (lane < 36) {
        const uint32_t i = lane / 6, j = lane % 6;
        const Fq2 a = x[i], b = y[j];
        s.prod[lane] = {fq2_mul_coord(a, b, 0, i + j >= 6), fq2_mul_coord(a, b, 1, i + j >= 6)};
    }
    wfold(s, dst, lane);
}
// dst = x^2: the 21 products a_i*a_j, i <= j, are each computed once — 42 lanes, ONE pass each — and stored at both (i, j) and (j, i)
__device__ __forceinline__ void wsqr(WaveShared& s, Fq2* dst, const Fq2* x, uint32_t lane) {
    if (lane < 42) {
        const uint32_t pr = lane >> 1, coord = lane & 1;
        // pair number -> (i, j), i <= j, rows of lengths 6, 5, 4, 3, 2, 1
        uint32_t i = 0, r = pr;
        while (r >= 6 - i) { r -= 6 - i; ++i; }
        const uint32_t j = i + r;
        const Fq v = fq2_mul_coord(x[i], x[j], coord, i + j >= 6);
        prod_store(s.prod, i * 6 + j, coord, v);
        prod_store(s.prod, j * 6 + i, coord, v);
    }
    wfold(s, dst, lane);
}
// dst = x * l for a line product l (coefficient of w^5 is zero): 30 Fq2 products, 60 lanes, ONE pass each
__device__ __forceinline__ void wmul_line(WaveShared& s, Fq2* dst, const Fq2* x, const Fq2* l, uint32_t lane) {
    if (lane < 60) {
        const uint32_t pr = lane >> 1, coord = lane & 1, i = pr / 5, j = pr % 5;
        prod_store(s.prod, i * 6 + j, coord, fq2_mul_coord(x[i], l[j], coord, i + j >= 6));
    }
    if (lane < 12) prod_store(s.prod, (lane >> 1) * 6 + 5, lane & 1, Fq::zero());
    wfold(s, dst, lane);
}
__device__ __forceinline__ void wcopy(Fq2* dst, const Fq2* src, uint32_t lane) { if (lane < 6) dst[lane] = src[lane]; __syncthreads(); }
// x -> x^(p^6): w -> -w
__device__ __forceinline__ void wconj(Fq2* dst, const Fq2* src, uint32_t lane) { if (lane < 6) dst[lane] = (lane & 1) ? src[lane].neg() : src[lane]; __syncthreads(); }
// x -> x^p
__device__ __forceinline__ void wfrob(Fq2* dst, const Fq2* src, const PairingConsts* k, uint32_t lane) {
    if (lane < 6) { Fq2 c = src[lane].conj(); dst[lane] = lane == 0 ? c : Fq2::mul(c, k->gamma1[lane]); }
    __syncthreads();
}
// Inverse in the tower view f = c0 + c1 w, c0 = (a0, a2, a4), c1 = (a1, a3, a5) in Fq6 = Fq2[v]/(v^3 - xi), w^2 = v:
// f^-1 = (c0 - c1 w) / N with N = f * (c0 - c1 w) = c0^2 - v c1^2 in Fq6 (only even powers of w).  The two products run on
// the wave; only the Fq6 inversion (one Fq inversion inside) is left to a single lane.  n must hold N on entry and holds
// N^-1 (as an element of the big field) on return.
__device__ __noinline__ void winv6_lane0(Fq2* n) {
    Fq6 t = {n[0], n[2], n[4]};
    t = t.inv();
    n[0] = t.c0; n[2] = t.c1; n[4] = t.c2;
    n[1] = Fq2::zero(); n[3] = Fq2::zero(); n[5] = Fq2::zero();
}
// dst = x^BN_X (x in the cyclotomic subgroup); tmp is scratch; dst must not alias x
__device__ __forceinline__ void wpow_x(WaveShared& s, Fq2* dst, const Fq2* x, uint32_t lane) {
    wcopy(dst, x, lane);
    for (int i = 61; i >= 0; --i) {
        wsqr(s, dst, dst, lane);
        if ((BN_X >> i) & 1) wmul(s, dst, dst, x, lane);
    }
}

__global__ void __launch_bounds__(64) k_pairing_wave(const G1J* __restrict__ pairs, uint32_t n, const LineCoeff* __restrict__ l_sg2,
                                                     const LineCoeff* __restrict__ l_ng2, const PairingConsts* __restrict__ consts,
                                                     uint32_t* __restrict__ ok) {
    __shared__ WaveShared s;
    const uint32_t chk = blockIdx.x, lane = threadIdx.x;
    if (chk >= n) return;
    const G1J P0 = pairs[2 * chk], P1 = pairs[2 * chk + 1];
    const bool skip0 = P0.is_identity(), skip1 = P1.is_identity();
    // line l(P) = a*y + b*x*w + c*w^3 with (x, y) = (X/Z^2, Y/Z^3); scaled by Z^3: a*Y + b*X*Z*w + c*Z^3*w^3
    const Fq xz0 = P0.X * P0.Z, z30 = P0.Z.sqr() * P0.Z, xz1 = P1.X * P1.Z, z31 = P1.Z.sqr() * P1.Z;
    for (uint32_t t = lane; t < 2 * N_LINES; t += 64) {
        const uint32_t pr = t / N_LINES, li = t % N_LINES;
        const LineCoeff c = pr ? l_ng2[li] : l_sg2[li];
        s.line[li][3 * pr + 0] = c.a.scale(pr ? P1.Y : P0.Y);
        s.line[li][3 * pr + 1] = c.b.scale(pr ? xz1 : xz0);
        s.line[li][3 * pr + 2] = c.c.scale(pr ? z31 : z30);
    }
    if (lane < 6) s.f[lane] = lane == 0 ? Fq2::one() : Fq2::zero();
    __syncthreads();
    // all per-step line products up front, one lane per step, in place:
    //   (a0 + b0 w + c0 w^3)(a1 + b1 w + c1 w^3) = (a0a1 + xi c0c1) + (a0b1 + b0a1) w + b0b1 w^2 + (a0c1 + c0a1) w^3 + (b0c1 + c0b1) w^4
    // an identity point contributes the line value 1
    for (uint32_t li = lane; li < N_LINES; li += 64) {
        Fq2 a0 = s.line[li][0], b0 = s.line[li][1], c0 = s.line[li][2], a1 = s.line[li][3], b1 = s.line[li][4], c1 = s.line[li][5];
        if (skip0) { a0 = Fq2::one(); b0 = Fq2::zero(); c0 = Fq2::zero(); }
        if (skip1) { a1 = Fq2::one(); b1 = Fq2::zero(); c1 = Fq2::zero(); }
        Fq2 a0a1 = a0 * a1, b0b1 = b0 * b1, c0c1 = c0 * c1;
        Fq2 ab = (a0 + b0) * (a1 + b1) - a0a1 - b0b1;          // a0b1 + b0a1
        Fq2 ac = (a0 + c0) * (a1 + c1) - a0a1 - c0c1;          // a0c1 + c0a1
        Fq2 bc = (b0 + c0) * (b1 + c1) - b0b1 - c0c1;          // b0c1 + c0b1
        s.line[li][0] = a0a1 + c0c1.mul_xi(); s.line[li][1] = ab; s.line[li][2] = b0b1; s.line[li][3] = ac; s.line[li][4] = bc; s.line[li][5] = Fq2::zero();
    }
    __syncthreads();
    // Miller loop: one squaring and one product per doubling step, one more product per addition step
    uint32_t idx = 0;
    for (int i = 63; i >= 0; --i) {
        wsqr(s, s.f, s.f, lane);
        wmul_line(s, s.f, s.f, s.line[idx++], lane);
        if ((ATE_LOW >> i) & 1) wmul_line(s, s.f, s.f, s.line[idx++], lane);
    }
    wmul_line(s, s.f, s.f, s.line[idx++], lane);
    wmul_line(s, s.f, s.f, s.line[idx++], lane);
    // final exponentiation, easy part: r = f^((p^6 - 1)(p^2 + 1))
    wconj(s.t1, s.f, lane);                    // t1 = conj(f) = f^(p^6)
    wmul(s, s.t2, s.f, s.t1, lane);            // N = f * conj(f), in Fq6
    if (lane == 0) winv6_lane0(s.t2);
    __syncthreads();
    wmul(s, s.t0, s.t1, s.t2, lane);           // f^-1
    wmul(s, s.r, s.t1, s.t0, lane);            // f^(p^6 - 1)
    wfrob(s.t0, s.r, consts, lane);
    wfrob(s.t0, s.t0, consts, lane);
    wmul(s, s.r, s.t0, s.r, lane);             // ^(p^2 + 1)
    // hard part: the x-power chain of Fuentes-Castaneda et al. (y0 .. y16)
    Fq2 *y0 = s.t0, *y1 = s.t1, *y3 = s.t2, *y4 = s.t3, *y6 = s.t4, *u = s.t5, *v = s.t6;
    wpow_x(s, y0, s.r, lane); wconj(y0, y0, lane);               // y0 = r^-x
    wsqr(s, y1, y0, lane);                                       // y1 = y0^2
    wsqr(s, u, y1, lane);                                        // y2 = y1^2
    wmul(s, y3, u, y1, lane);                                    // y3 = y2 * y1
    wpow_x(s, y4, y3, lane); wconj(y4, y4, lane);                // y4 = y3^-x
    wsqr(s, u, y4, lane);                                        // y5 = y4^2
    wpow_x(s, y6, u, lane); wconj(y6, y6, lane);                 // y6 = y5^-x
    wconj(y3, y3, lane);
    wconj(y6, y6, lane);
    wmul(s, u, y6, y4, lane);                                    // y7 = y6 * y4
    wmul(s, u, u, y3, lane);                                     // y8 = y7 * y3          (u = y8)
    wmul(s, v, u, y1, lane);                                     // y9 = y8 * y1          (v = y9)
    wmul(s, y0, u, y4, lane);                                    // y10 = y8 * y4
    wmul(s, y0, y0, s.r, lane);                                  // y11 = y10 * r         (y0 = y11)
    wfrob(y1, v, consts, lane);                                  // y12 = y9^p
    wmul(s, y0, y1, y0, lane);                                   // y13 = y12 * y11       (y0 = y13)
    wfrob(u, u, consts, lane); wfrob(u, u, consts, lane);        // y8^(p^2)
    wmul(s, y0, u, y0, lane);                                    // y14 = y8' * y13       (y0 = y14)
    wconj(y1, s.r, lane);
    wmul(s, y1, y1, v, lane);                                    // conj(r) * y9
    wfrob(y1, y1, consts, lane); wfrob(y1, y1, consts, lane); wfrob(y1, y1, consts, lane);  // y15
    wmul(s, y0, y1, y0, lane);                                   // y16 = y15 * y14
    if (lane == 0) {
        bool one = y0[0] == Fq2::one();
        for (int k2 = 1; k2 < 6; ++k2) one = one && y0[k2].is_zero();
        ok[chk] = one ? 1u : 0u;
    }
}

int pairing_check_enqueue(hipStream_t s, const PairingDevice& pd, const G1J* d_pairs, uint32_t n, uint32_t* d_ok) {
    if (!n) return 0;
    hipLaunchKernelGGL(k_pairing_wave, dim3(n), dim3(64), 0, s, d_pairs, n, pd.l_sg2, pd.l_ng2, pd.consts, d_ok);
    H2V_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace h2v
