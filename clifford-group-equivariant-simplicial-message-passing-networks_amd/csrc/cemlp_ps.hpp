// Parity-split ("PS") variant of the fused CEMLP row program for algebras with an ODD
// number of generators and at most 8 output channels per block (the BASELINE shapes:
// Cl(3,0) 8 channels, Cl(4,1) 8 channels).
//
// Same arithmetic as cemlp_device.hpp (csmpn/models/cegnn_utils.py:34-155,287-338), different
// distribution over the wave: a 16-row tile, lane column n = (s, c) with c = n & 7 the
// channel and s = n >> 3 the blade PARITY the lane owns: s = 0 lanes hold the D/2 even-grade
// blades of (row, channel), s = 1 lanes the D/2 odd-grade ones. Every activation tensor is
// f4 t[D/2] (4 rows of the MFMA accumulator layout), i.e. HALF the registers and HALF the LDS
// tile of the 32-row layout (H = 2) at the same lane utilisation, so twice as many waves are
// resident (2 per SIMD in the backward, 3-4 in the forward) and hide each other's LDS / MFMA /
// memory latencies - the 32-row kernels run one wave per SIMD in the backward and stall 65 %
// of their cycles.
//
// Slot order. Slot j of an s = 0 lane is the j-th even blade ev[j] (blade order); slot j of
// an s = 1 lane is its Hodge partner od[j] = blade(bitmap(ev[j]) ^ (D-1)). Grades pair up as
// (2k, n-2k): "grade class" k has the same slots in both parities, so gates, norms and
// parameters index by class with a per-lane grade number.
//
// Geometric product. With I the pseudoscalar (central for odd n) and the odd part written as
// x_odd = X~ I with X~ in the even subalgebra (coefficients x~[j] = eps[j] x[od[j]],
// ev[j] I = eps[j] od[j]):
//      (Ze + Z~o I)(Re + R~o I) = [Ze Re + I^2 Z~o R~o] + [Ze R~o + Z~o Re] I
// so every lane evaluates TWO products of the even subalgebra ((D/2)^2 terms each) on its own
// and its partner's (lane n ^ 8, one DPP rotate) operands, with one instruction stream for
// both parities; the path weights (cegnn_utils.py:126-140) are per-lane values picked by the
// true grades.
#pragma once
#include "cemlp_kernel.hpp"

namespace csmpn {

template <class ALG>
struct PSTab {
    static constexpr int N = ALG::n, D = ALG::D, DL = D / 2, GC = (N + 1) / 2, G = ALG::G;
    static constexpr int MAXQ = GC * GC * GC;
    int ev[DL], od[DL], eps[DL];   // slot -> even blade, odd blade, sign of ev*I = eps*od
    int slot[D], par[D];           // blade -> slot, parity
    int cls[DL];                   // slot -> grade class
    int cstart[GC + 1];            // class -> first slot
    int I2;                        // I*I
    int pc[DL][DL], psg[DL][DL];   // even subalgebra: ev[a]*ev[b] = psg * ev[pc]
    int nq;                        // path classes (class_a, class_c, class_b) of the even subalgebra
    int qg[MAXQ][3];
    int pid[2][2][MAXQ];           // [lane parity][product 0/1][path class] -> reference path index
};

template <class ALG>
constexpr PSTab<ALG> make_ps_tab() {
    PSTab<ALG> t{};
    constexpr int N = ALG::n, D = ALG::D, DL = D / 2, GC = (N + 1) / 2;
    int j = 0;
    for (int g = 0; g <= N; g += 2) t.cstart[g / 2] = -1;
    for (int d = 0; d < D; ++d) {
        const int g = ALG::t.bo.grade[d];
        t.par[d] = g & 1;
        if ((g & 1) == 0) {
            if (t.cstart[g / 2] < 0) t.cstart[g / 2] = j;
            t.ev[j] = d;
            t.cls[j] = g / 2;
            t.slot[d] = j;
            ++j;
        }
    }
    t.cstart[GC] = DL;
    const int I = ALG::t.bo.index[D - 1];
    for (int s = 0; s < DL; ++s) {
        t.od[s] = ALG::t.bo.index[ALG::t.bo.bitmap[t.ev[s]] ^ (D - 1)];
        t.slot[t.od[s]] = s;
        t.eps[s] = ALG::t.sign[t.ev[s]][I];
    }
    t.I2 = ALG::t.sign[I][I];
    bool present[GC][GC][GC] = {};
    for (int a = 0; a < DL; ++a)
        for (int b = 0; b < DL; ++b) {
            const int c = t.slot[ALG::t.out[t.ev[a]][t.ev[b]]];
            t.pc[a][b] = c;
            t.psg[a][b] = ALG::t.sign[t.ev[a]][t.ev[b]];
            present[t.cls[a]][t.cls[c]][t.cls[b]] = true;
        }
    int q = 0;
    for (int a = 0; a < GC; ++a)
        for (int c = 0; c < GC; ++c)
            for (int b = 0; b < GC; ++b)
                if (present[a][c][b]) {
                    t.qg[q][0] = a; t.qg[q][1] = c; t.qg[q][2] = b;
                    const int ea = 2 * a, ec = 2 * c, eb = 2 * b, oa = N - ea, oc = N - ec, ob = N - eb;
                    t.pid[0][0][q] = ALG::t.path_id[ea][ec][eb];   // Ze  Re   -> even
                    t.pid[0][1][q] = ALG::t.path_id[oa][ec][ob];   // Z~o R~o  -> even
                    t.pid[1][0][q] = ALG::t.path_id[ea][oc][ob];   // Ze  R~o  -> odd
                    t.pid[1][1][q] = ALG::t.path_id[oa][oc][eb];   // Z~o Re   -> odd
                    ++q;
                }
    t.nq = q;
    return t;
}

// compile-time self-check of the tables: every path class maps to four existing reference paths,
// the Hodge pairing is a bijection between even and odd blades, classes tile the slots, and the
// pseudoscalar is central with I^2 = +-1 (what the two-product form of the geometric product needs)
template <class ALG>
constexpr bool ps_tab_ok() {
    constexpr PSTab<ALG> t = make_ps_tab<ALG>();
    constexpr int D = ALG::D, DL = D / 2, GC = (ALG::n + 1) / 2;
    if (ALG::n % 2 == 0) return false;
    bool seen[D] = {};
    for (int s = 0; s < DL; ++s) {
        if (t.par[t.ev[s]] != 0 || t.par[t.od[s]] != 1) return false;
        if (t.slot[t.ev[s]] != s || t.slot[t.od[s]] != s) return false;
        if (seen[t.ev[s]] || seen[t.od[s]]) return false;
        seen[t.ev[s]] = seen[t.od[s]] = true;
        if (t.eps[s] != 1 && t.eps[s] != -1) return false;
        if (ALG::t.bo.grade[t.od[s]] != ALG::n - ALG::t.bo.grade[t.ev[s]]) return false;
    }
    if (t.I2 != 1 && t.I2 != -1) return false;
    const int I = ALG::t.bo.index[D - 1];
    for (int d = 0; d < D; ++d)   // I commutes with every blade
        if (ALG::t.sign[d][I] != ALG::t.sign[I][d]) return false;
    if (t.cstart[0] != 0 || t.cstart[GC] != DL) return false;
    for (int q = 0; q < t.nq; ++q)
        for (int s = 0; s < 2; ++s)
            for (int k = 0; k < 2; ++k)
                if (t.pid[s][k][q] < 0 || t.pid[s][k][q] >= ALG::P) return false;
    return 4 * t.nq == ALG::P;   // the four products of the path classes enumerate all reference paths
}

template <class ALG>
struct PS {
    static_assert(ps_tab_ok<ALG>(), "parity-split tables are inconsistent for this algebra");
    static constexpr PSTab<ALG> t = make_ps_tab<ALG>();
    static constexpr int N = ALG::n, D = ALG::D, DL = D / 2, GC = (N + 1) / 2, G = ALG::G, QP = t.nq;
    static constexpr int R = 16, NW = 8, CS = R * D + 4;
    static constexpr int csize(int k) { return t.cstart[k + 1] - t.cstart[k]; }
};

// geometry of a PS tile
template <class ALG>
struct GeoPS {
    using P = PS<ALG>;
    int lane, n, q, s, cn, r0;
    float tau;   // -1 in odd lanes: sign of the flipped slots when moving to / from the X~ basis
    CSMPN_DEV explicit GeoPS(int lane_) : lane(lane_), n(lane_ & 15), q(lane_ >> 4) {
        s = n >> 3;
        cn = n & 7;
        r0 = 4 * q;
        tau = s ? -1.0f : 1.0f;
    }
    // blade held in slot j by this lane
    template <int J> CSMPN_DEV int blade() const { return s ? P::t.od[J] : P::t.ev[J]; }
    // quadratic-form sign of slot j
    template <int J> CSMPN_DEV float qs() const {
        constexpr int qe = ALG::t.qsign[P::t.ev[J]], qo = ALG::t.qsign[P::t.od[J]];
        if constexpr (qe == qo) return float(qe);
        else return s ? float(qo) : float(qe);
    }
    // true grade of grade class k in this lane
    CSMPN_DEV int grade(int k) const { return s ? P::N - 2 * k : 2 * k; }
    CSMPN_DEV void stamp(int) const {}
};

template <int BANK_MASK>
CSMPN_DEV float dpp_ror8_masked(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 0x128, 0xF, BANK_MASK, false));
}
CSMPN_DEV f4 partner4(f4 v) { return f4{dpp_mov<0x128>(v.x), dpp_mov<0x128>(v.y), dpp_mov<0x128>(v.z), dpp_mov<0x128>(v.w)}; }
// even-lane value in both lanes of a pair (the odd lane takes its partner's) / the converse
CSMPN_DEV f4 even4(f4 v) {
    return f4{dpp_ror8_masked<0xC>(v.x, v.x), dpp_ror8_masked<0xC>(v.y, v.y), dpp_ror8_masked<0xC>(v.z, v.z),
              dpp_ror8_masked<0xC>(v.w, v.w)};
}
CSMPN_DEV f4 odd4(f4 v) {
    return f4{dpp_ror8_masked<0x3>(v.x, v.x), dpp_ror8_masked<0x3>(v.y, v.y), dpp_ror8_masked<0x3>(v.z, v.z),
              dpp_ror8_masked<0x3>(v.w, v.w)};
}
// sum over the 4 row quarters of a lane column (no partner add: the partner lane holds other parameters)
CSMPN_DEV float quarter_sum(float v) {
    const f4 r = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, v, splat(0.f), 0, 0, 0);   // see channel_rows_sum
    return r.x;
}

// ---------------------------------------------------------------------------------
// MFMA pieces (weights in the LDS store [g][O][IP], tiles [channel][blade][16 rows])

// acc[slot(d)][v] += sum_in T[in][d][row] W[out][in][grade(d)]   (TRANS: sum over out, acc over in)
// One MFMA chain per blade d of BOTH parities: the B fragment is zero in the lanes of the
// other parity, so those lanes add 0 to the accumulator they share with their own blade.
template <class ALG, bool TRANS>
CSMPN_DEV void ps_linear(f4 (&acc)[PS<ALG>::DL], const float* tile, int CP, int KK, const WSrc& ws, int nt,
                         const GeoPS<ALG>& ge) {
    using P = PS<ALG>;
    constexpr int G = ALG::G, R = P::R, CS = P::CS, NW = P::NW;
    const int ncol = NW * nt + ge.cn;
    // k-slot (q, v) of k-block kk is contracted channel 16*kk + 4*v + q (see linear_from_tile)
    for (int kk = 0; kk < KK; ++kk) {
        const int c0 = 16 * kk + ge.q;
        const int left = CP - 16 * kk;
        const int nv = left >= 16 ? 4 : left / 4;
        const float* ap = tile + (c0 < CP ? c0 : 0) * CS + ge.n;
        auto body = [&](auto NVc) {
            constexpr int NV = decltype(NVc)::value;
            static_for<0, G>([&](auto g) {
                f4 b;
                const int gi = ws.grades ? int(g) : 0;
                const bool mine = ge.s == (int(g) & 1);
                if constexpr (!TRANS) {
                    const bool okc = mine && ncol < ws.O;
                    const float* wp = ws.w + (gi * ws.O + (ncol < ws.O ? ncol : 0)) * ws.IP;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int i = c0 + 4 * v;
                        b[v] = wp[i < ws.IP ? i : 0] * ((okc && i < ws.IP) ? 1.0f : 0.0f);
                    }
                } else {
                    const bool okc = mine && ncol < ws.IP;
                    const float* wp = ws.w + gi * ws.O * ws.IP + (ncol < ws.IP ? ncol : 0);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int o = c0 + 4 * v;
                        b[v] = wp[(o < ws.O ? o : 0) * ws.IP] * ((okc && o < ws.O) ? 1.0f : 0.0f);
                    }
                }
                constexpr int d0 = ALG::gstart(g), nd = ALG::gsize(g);
                float a[NV][nd];
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int t = 0; t < nd; ++t) a[v][t] = ap[(4 * v < left ? 4 * v : 0) * CS + (d0 + t) * R];
                static_for<0, nd>([&](auto tt) {
                    constexpr int sl = P::t.slot[d0 + decltype(tt)::value];
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[sl] = mfma16(a[v][decltype(tt)::value], b[v], acc[sl]);
                });
            });
        };
        if (nv == 2) body(IC<2>{});
        else body(IC<4>{});
    }
}

// gW[o][cin][grade] += sum_{rows, slots of the grade} Gr[row][slot][o] * T[cin][blade][row]
// A = the lane-layout gradient (registers), B = the input-side LDS tile read at THIS lane's
// blade of the slot. Output element (i, j) is valid where the parity of column i equals the
// parity of column j.
template <class ALG, bool MIRROR>
CSMPN_DEV void ps_weight_grad(const f4 (&gr)[PS<ALG>::DL], const float* tile, int CP, int I, int O, int NTin,
                              const GeoPS<ALG>& ge, float* dstp, bool has_grades) {
    using P = PS<ALG>;
    constexpr int G = ALG::G, R = P::R, CS = P::CS, NW = P::NW, GC = P::GC;
    const int hi = ge.q >> 1;
    const int ob = 4 * (ge.q & 1);
    for (int it = 0; it < NTin; ++it) {
        const int cin = NW * it + ge.cn;
        const float* bp = tile + (cin < CP ? cin : 0) * CS + ge.r0;
        f4 accg[GC];
        static_for<0, GC>([&](auto k) {
            f4 acc = splat(0.f);
            static_for<P::t.cstart[k], P::t.cstart[k + 1]>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                const f4 b = *reinterpret_cast<const f4*>(bp + ge.template blade<j>() * R);
#pragma unroll
                for (int v = 0; v < 4; ++v) acc = mfma16(gr[j][v], b[v], acc);
            });
            accg[k] = acc;
        });
        if (cin < I && hi == ge.s && ob < O) {
            if (has_grades) {
                static_for<0, GC>([&](auto k) {
                    const int g = ge.grade(k);
                    float* p = MIRROR ? dstp + (g * O + ob) * I + cin : dstp + (ob * I + cin) * G + g;
                    const int so = MIRROR ? I : I * G;
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (ob + v < O) atomicAdd(p + v * so, accg[k][v]);
                });
            } else {
                f4 tot = accg[0];
                static_for<1, GC>([&](auto k) { tot += accg[k]; });
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (ob + v < O) atomicAdd(dstp + (ob + v) * I + cin, tot[v]);
            }
        }
    }
}

template <class ALG>
CSMPN_DEV void ps_store_tile(const f4 (&t)[PS<ALG>::DL], float* tile, int CP, const GeoPS<ALG>& ge) {
    using P = PS<ALG>;
    if (ge.cn < CP) {
        float* p = tile + ge.cn * P::CS + ge.r0;
        static_for<0, P::DL>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            *reinterpret_cast<f4*>(p + ge.template blade<j>() * P::R) = t[j];
        });
    }
}

// lane-layout tensor -> dense staging [16 rows][nch*D] in reference order
template <class ALG>
CSMPN_DEV void ps_store_dense(const f4 (&t)[PS<ALG>::DL], float* stage, int nch, int ch, const GeoPS<ALG>& ge) {
    using P = PS<ALG>;
    if (ch < nch) {
        static_for<0, P::DL>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            float* p = stage + (ge.r0 * nch + ch) * P::D + ge.template blade<j>();
#pragma unroll
            for (int v = 0; v < 4; ++v) p[v * nch * P::D] = t[j][v];
        });
    }
}

template <class ALG>
CSMPN_DEV void ps_park(const f4 (&t)[PS<ALG>::DL], float* area, int lane) {
#pragma unroll
    for (int j = 0; j < PS<ALG>::DL; ++j) *reinterpret_cast<f4*>(area + (j * 64 + lane) * 4) = t[j];
}
template <class ALG>
CSMPN_DEV void ps_unpark(f4 (&t)[PS<ALG>::DL], const float* area, int lane) {
#pragma unroll
    for (int j = 0; j < PS<ALG>::DL; ++j) t[j] = *reinterpret_cast<const f4*>(area + (j * 64 + lane) * 4);
}

// ---------------------------------------------------------------------------------
template <class ALG>
struct PSLaneParams {
    float b1, bL, la;
    float sa[PS<ALG>::GC], sb[PS<ALG>::GC], sg[PS<ALG>::GC];
    bool cvalid;
};

template <class ALG>
CSMPN_DEV PSLaneParams<ALG> ps_lane_params(const DevBlock& B, const float* ws, const WOff& wo, const GeoPS<ALG>& ge) {
    constexpr int G = ALG::G, GC = PS<ALG>::GC;
    PSLaneParams<ALG> p;
    const int c = ge.cn;
    p.cvalid = c < B.O;
    const int cc = p.cvalid ? c : 0;
    const float m = p.cvalid ? 1.0f : 0.0f;
    const float m0 = (p.cvalid && ge.s == 0) ? 1.0f : 0.0f;   // biases live on blade 0: even lanes, slot 0
    p.b1 = ws[wo.b1 + cc] * m0;
    p.bL = ws[wo.bL + cc] * m0;
    p.la = ws[wo.la + cc] * m;
#pragma unroll
    for (int k = 0; k < GC; ++k) {
        const int g = ge.grade(k);
        p.sa[k] = ws[wo.sa + cc * G + g] * m;
        p.sb[k] = ws[wo.sb + cc * G + g] * m;
        p.sg[k] = ws[wo.sg + cc * G + g] * m;
    }
    return p;
}

template <class ALG>
struct PSFwdState {
    f4 y[PS<ALG>::DL];
    f4 gate[PS<ALG>::GC];
    f4 R[PS<ALG>::DL];
    f4 invden[PS<ALG>::GC];
    f4 s[PS<ALG>::DL];
    f4 qs, nl, invMn;
};

// The geometric product runs on TWO rows of the lane at a time (f2 = one v_pk_* operand):
// its exchanged operand copies are the widest live set of the whole block, and halving them
// is what keeps the backward within 256 VGPRs (2 waves per SIMD).
typedef float f2 __attribute__((ext_vector_type(2)));
CSMPN_DEV f2 splat2(float v) { return f2{v, v}; }
CSMPN_DEV f2 partner2(f2 v) { return f2{dpp_mov<0x128>(v.x), dpp_mov<0x128>(v.y)}; }
CSMPN_DEV f2 even2(f2 v) { return f2{dpp_ror8_masked<0xC>(v.x, v.x), dpp_ror8_masked<0xC>(v.y, v.y)}; }
CSMPN_DEV f2 odd2(f2 v) { return f2{dpp_ror8_masked<0x3>(v.x, v.x), dpp_ror8_masked<0x3>(v.y, v.y)}; }
template <int HALF> CSMPN_DEV f2 half_of(f4 v) { return HALF == 0 ? f2{v.x, v.y} : f2{v.z, v.w}; }
template <int HALF> CSMPN_DEV void set_half(f4& d, f2 v) {
    if constexpr (HALF == 0) { d.x = v.x; d.y = v.y; } else { d.z = v.x; d.w = v.y; }
}

// to / from the X~ basis: slots with eps = -1 change sign in the odd lanes
template <class ALG>
CSMPN_DEV void ps_tilde(f2 (&t)[PS<ALG>::DL], const GeoPS<ALG>& ge) {
    static_for<0, PS<ALG>::DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        if constexpr (PS<ALG>::t.eps[j] < 0) t[j] *= ge.tau;
    });
}

// per-lane path weights of the two forward products
template <class ALG>
CSMPN_DEV void ps_fwd_weights(const float* wrow, const GeoPS<ALG>& ge, float (&wA)[PS<ALG>::QP], float (&wB)[PS<ALG>::QP]) {
    using P = PS<ALG>;
    static_for<0, P::QP>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        wA[q] = wrow[ge.s ? P::t.pid[1][0][q] : P::t.pid[0][0][q]];
        wB[q] = wrow[ge.s ? P::t.pid[1][1][q] : P::t.pid[0][1][q]] * (ge.s ? 1.0f : float(P::t.I2));
    });
}

// out (reference basis, own parity) += weighted geometric product of z and r (own parity,
// reference basis), rows 2*HALF, 2*HALF+1 of the lane
template <class ALG, int HALF>
CSMPN_DEV void ps_weighted_gp_half(f4 (&out)[PS<ALG>::DL], const f4 (&z)[PS<ALG>::DL], const f4 (&r)[PS<ALG>::DL],
                                   const float (&wA)[PS<ALG>::QP], const float (&wB)[PS<ALG>::QP], const GeoPS<ALG>& ge) {
    using P = PS<ALG>;
    constexpr int DL = P::DL, QP = P::QP;
    f2 zE[DL], zO[DL], rw[DL], ro[DL], gp[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) { zE[j] = half_of<HALF>(z[j]); rw[j] = half_of<HALF>(r[j]); gp[j] = splat2(0.f); }
    ps_tilde<ALG>(zE, ge);
    ps_tilde<ALG>(rw, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        zO[j] = odd2(zE[j]);
        zE[j] = even2(zE[j]);
        ro[j] = partner2(rw[j]);
    }
    static_for<0, QP>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        constexpr int ka = P::t.qg[q][0], kc = P::t.qg[q][1], kb = P::t.qg[q][2];
        constexpr int a0 = P::t.cstart[ka], a1 = P::t.cstart[ka + 1];
        constexpr int c0 = P::t.cstart[kc], nc = P::t.cstart[kc + 1] - c0;
        constexpr int b0 = P::t.cstart[kb], b1 = P::t.cstart[kb + 1];
        f2 tA[nc], tB[nc];
#pragma unroll
        for (int t = 0; t < nc; ++t) { tA[t] = splat2(0.f); tB[t] = splat2(0.f); }
        static_for<a0, a1>([&](auto aa) {
            static_for<b0, b1>([&](auto bb) {
                constexpr int a = decltype(aa)::value, b = decltype(bb)::value;
                constexpr int c = P::t.pc[a][b];
                if constexpr (c >= c0 && c < c0 + nc) {
                    constexpr float sg = float(P::t.psg[a][b]);
                    tA[c - c0] += (sg * zE[a]) * rw[b];
                    tB[c - c0] += (sg * zO[a]) * ro[b];
                }
            });
        });
#pragma unroll
        for (int t = 0; t < nc; ++t) gp[c0 + t] += wA[q] * tA[t] + wB[q] * tB[t];
    });
    ps_tilde<ALG>(gp, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) set_half<HALF>(out[j], half_of<HALF>(out[j]) + gp[j]);
}

template <class ALG>
CSMPN_DEV void ps_weighted_gp(f4 (&out)[PS<ALG>::DL], const f4 (&z)[PS<ALG>::DL], const f4 (&r)[PS<ALG>::DL],
                              const float* wrow, const GeoPS<ALG>& ge) {
    float wA[PS<ALG>::QP], wB[PS<ALG>::QP];
    ps_fwd_weights<ALG>(wrow, ge, wA, wB);
    ps_weighted_gp_half<ALG, 0>(out, z, r, wA, wB, ge);
    CSMPN_PHASE();
    ps_weighted_gp_half<ALG, 1>(out, z, r, wA, wB, ge);
}

// backward of ps_weighted_gp for two rows. ggp: d/d(out) (reference basis). gz accumulates,
// gr is set; gwA / gwB accumulate the per-lane partial sums of the gradients of the lane's two
// forward weight sets (same path indices as ps_fwd_weights; the I^2 factor of wB is applied
// by the caller).
template <class ALG, int HALF>
CSMPN_DEV void ps_weighted_gp_bwd_half(const f4 (&ggp)[PS<ALG>::DL], const f4 (&y)[PS<ALG>::DL],
                                       const f4 (&gate)[PS<ALG>::GC], const f4 (&Rr)[PS<ALG>::DL],
                                       const f4 (&invden)[PS<ALG>::GC], const float* wrow, const GeoPS<ALG>& ge,
                                       f4 (&gz)[PS<ALG>::DL], f4 (&gr)[PS<ALG>::DL], float (&gwA)[PS<ALG>::QP],
                                       float (&gwB)[PS<ALG>::QP]) {
    using P = PS<ALG>;
    constexpr int DL = P::DL, QP = P::QP;
    const float i2 = float(P::t.I2);
    f2 zw[DL], zo[DL], zE[DL], zO[DL], rE[DL], rO[DL], Gw[DL], Go[DL], gzt[DL], grt[DL];
    static_for<0, DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        zw[j] = half_of<HALF>(gate[P::t.cls[j]]) * half_of<HALF>(y[j]);
        rE[j] = half_of<HALF>(Rr[j]) * half_of<HALF>(invden[P::t.cls[j]]);
        Gw[j] = half_of<HALF>(ggp[j]);
        gzt[j] = splat2(0.f);
        grt[j] = splat2(0.f);
    });
    ps_tilde<ALG>(zw, ge);
    ps_tilde<ALG>(rE, ge);
    ps_tilde<ALG>(Gw, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        zo[j] = partner2(zw[j]);
        zE[j] = even2(zw[j]);
        zO[j] = odd2(zw[j]);
        rO[j] = odd2(rE[j]);
        rE[j] = even2(rE[j]);
        Go[j] = partner2(Gw[j]);
    }
    static_for<0, QP>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        constexpr int ka = P::t.qg[q][0], kc = P::t.qg[q][1], kb = P::t.qg[q][2];
        constexpr int a0 = P::t.cstart[ka], na = P::t.cstart[ka + 1] - a0;
        constexpr int c0 = P::t.cstart[kc], nc = P::t.cstart[kc + 1] - c0;
        constexpr int b0 = P::t.cstart[kb], nb = P::t.cstart[kb + 1] - b0;
        // d/dz of the even lanes: w00 Ge (x) Re + w10 Go (x) Ro; of the odd lanes: w11 Go (x) Re + I^2 w01 Ge (x) Ro
        // d/dr of the even lanes: w00 Ge (x) Ze + w11 Go (x) Zo; of the odd lanes: w10 Go (x) Ze + I^2 w01 Ge (x) Zo
        const float w00 = wrow[P::t.pid[0][0][q]], w01 = wrow[P::t.pid[0][1][q]] * i2;
        const float w10 = wrow[P::t.pid[1][0][q]], w11 = wrow[P::t.pid[1][1][q]];
        const float u1 = ge.s ? w11 : w00, u2 = ge.s ? w01 : w10;
        const float v1 = ge.s ? w10 : w00, v2 = ge.s ? w01 : w11;
        f2 S1[na], S2[na], S3[na], V1[nb], V2[nb];
#pragma unroll
        for (int t = 0; t < na; ++t) { S1[t] = splat2(0.f); S2[t] = splat2(0.f); S3[t] = splat2(0.f); }
#pragma unroll
        for (int t = 0; t < nb; ++t) { V1[t] = splat2(0.f); V2[t] = splat2(0.f); }
        static_for<0, na>([&](auto aa) {
            static_for<0, nb>([&](auto bb) {
                constexpr int ai = decltype(aa)::value, bi = decltype(bb)::value;
                constexpr int a = a0 + ai, b = b0 + bi;
                constexpr int c = P::t.pc[a][b];
                if constexpr (c >= c0 && c < c0 + nc) {
                    constexpr float sg = float(P::t.psg[a][b]);
                    const f2 gw_ = sg * Gw[c], go_ = sg * Go[c];
                    S1[ai] += gw_ * rE[b];
                    S3[ai] += gw_ * rO[b];
                    S2[ai] += go_ * rO[b];
                    V1[bi] += gw_ * zE[a];
                    V2[bi] += go_ * zO[a];
                }
            });
        });
        f2 k1 = splat2(0.f), k3 = splat2(0.f);
#pragma unroll
        for (int t = 0; t < na; ++t) {
            gzt[a0 + t] += u1 * S1[t] + u2 * S2[t];
            k1 += zw[a0 + t] * S1[t];
            k3 += zo[a0 + t] * S3[t];
        }
#pragma unroll
        for (int t = 0; t < nb; ++t) grt[b0 + t] += v1 * V1[t] + v2 * V2[t];
        // even lanes: K1 -> w00 (product A), K3 -> w01 (product B); odd lanes: K3 -> w10 (A), K1 -> w11 (B)
        const float h1 = k1.x + k1.y, h3 = k3.x + k3.y;
        gwA[q] += ge.s ? h3 : h1;
        gwB[q] += ge.s ? h1 : h3;
    });
    ps_tilde<ALG>(gzt, ge);
    ps_tilde<ALG>(grt, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) {
        set_half<HALF>(gz[j], half_of<HALF>(gz[j]) + gzt[j]);
        set_half<HALF>(gr[j], grt[j]);
    }
}

template <class ALG>
CSMPN_DEV void ps_weighted_gp_bwd(const f4 (&ggp)[PS<ALG>::DL], const f4 (&y)[PS<ALG>::DL], const f4 (&gate)[PS<ALG>::GC],
                                  const f4 (&Rr)[PS<ALG>::DL], const f4 (&invden)[PS<ALG>::GC], const float* wrow,
                                  const GeoPS<ALG>& ge, f4 (&gz)[PS<ALG>::DL], f4 (&gr)[PS<ALG>::DL],
                                  float (&gwA)[PS<ALG>::QP], float (&gwB)[PS<ALG>::QP]) {
#pragma unroll
    for (int q = 0; q < PS<ALG>::QP; ++q) { gwA[q] = 0.f; gwB[q] = 0.f; }
    ps_weighted_gp_bwd_half<ALG, 0>(ggp, y, gate, Rr, invden, wrow, ge, gz, gr, gwA, gwB);
    CSMPN_PHASE();
    ps_weighted_gp_bwd_half<ALG, 1>(ggp, y, gate, Rr, invden, wrow, ge, gz, gr, gwA, gwB);
}

// ---------------------------------------------------------------------------------
template <class ALG>
CSMPN_DEV void ps_block_forward(const DevBlock& B, const PSLaneParams<ALG>& lp, const float* xin, float* zbuf,
                                const float* wstore, const GeoPS<ALG>& ge, PSFwdState<ALG>& S, f4 (&out)[PS<ALG>::DL]) {
    using P = PS<ALG>;
    constexpr int DL = P::DL, GC = P::GC, G = ALG::G;
    const int c = ge.cn;
    const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, ALG::P, B.w1_sub != 0);
    const WSrc sW1{nullptr, wstore + B.lds_woff + wo.W1, B.O, B.CPi, B.w1_sub};
    const WSrc sWR{nullptr, wstore + B.lds_woff + wo.WR, B.O, B.CPo, 1};
    const WSrc sWL{nullptr, wstore + B.lds_woff + wo.WL, B.O, B.CPo, 1};

    // 1. MVLinear (cegnn_utils.py:326-338)
#pragma unroll
    for (int j = 0; j < DL; ++j) S.y[j] = splat(0.f);
    ps_linear<ALG, false>(S.y, xin, B.CPi, B.KKi, sW1, 0, ge);
    S.y[0] += lp.b1;

    CSMPN_PHASE();
    // 2. MVSiLU (cegnn_utils.py:76-83): class 0 is the scalar in even lanes (invariant = the
    // signed scalar itself) and the pseudoscalar in odd lanes (invariant = q)
    f4 z[DL];
    static_for<0, GC>([&](auto k) {
        constexpr int j0 = P::t.cstart[k], j1 = P::t.cstart[k + 1];
        f4 u = splat(0.f);
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            u += ge.template qs<j>() * S.y[j] * S.y[j];
        });
        if constexpr (k == 0) {
            if (ge.s == 0) u = S.y[0];
        }
        S.gate[k] = sigmoid4(lp.sa[k] * u + lp.sb[k]);
        static_for<j0, j1>([&](auto jj) { z[decltype(jj)::value] = S.gate[k] * S.y[decltype(jj)::value]; });
    });
    tile_sync<VAR_WAVE>();
    ps_store_tile<ALG>(z, zbuf, B.CPo, ge);
    tile_sync<VAR_WAVE>();

    CSMPN_PHASE();
    // 3. linear_right / linear_left (cegnn_utils.py:143-148)
    f4 L[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) { S.R[j] = splat(0.f); L[j] = splat(0.f); }
    ps_linear<ALG, false>(S.R, zbuf, B.CPo, B.KKo, sWR, 0, ge);
    ps_linear<ALG, false>(L, zbuf, B.CPo, B.KKo, sWL, 0, ge);
    L[0] += lp.bL;

    CSMPN_PHASE();
    // 4. NormalizationLayer on the right operand (cegnn_utils.py:42-51)
    f4 r[DL];
    static_for<0, GC>([&](auto k) {
        constexpr int j0 = P::t.cstart[k], j1 = P::t.cstart[k + 1];
        f4 qq = splat(0.f);
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            qq += ge.template qs<j>() * S.R[j] * S.R[j];
        });
        const f4 m = lp.sg[k] * (smooth_abs_sqrt4(qq) - 1.0f) + 1.0f;
        S.invden[k] = rcp4(m + kEps);
        static_for<j0, j1>([&](auto jj) { r[decltype(jj)::value] = S.R[decltype(jj)::value] * S.invden[k]; });
    });

    CSMPN_PHASE();
    // 5. steerable geometric product + first-order term (cegnn_utils.py:126-152). All lanes
    // run it (the partner exchange is lane-uniform); padding channels carry zeros.
    ps_weighted_gp<ALG>(L, z, r, wstore + B.lds_woff + wo.w + (size_t)(lp.cvalid ? c : 0) * ALG::P, ge);
#pragma unroll
    for (int j = 0; j < DL; ++j) S.s[j] = lp.cvalid ? L[j] * kInvSqrt2 : splat(0.f);

    CSMPN_PHASE();
    // 6. MVLayerNorm (cegnn_utils.py:93-96): q over all blades = own half + partner's half
    f4 qs = splat(0.f);
    static_for<0, DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        qs += ge.template qs<j>() * S.s[j] * S.s[j];
    });
    qs += partner4(qs);
    S.qs = qs;
    S.nl = smooth_abs_sqrt4(qs);
    const f4 tot = chan_sum4<2>(lp.cvalid ? S.nl : splat(0.f));
    S.invMn = rcp4(tot * (1.0f / float(B.O)) + kEps);
#pragma unroll
    for (int j = 0; j < DL; ++j) out[j] = lp.la * S.s[j] * S.invMn;
}

template <class ALG>
CSMPN_DEV void ps_block_backward(const DevBlock& B, const PSLaneParams<ALG>& lp, const PSFwdState<ALG>& S,
                                 const f4 (&gout)[PS<ALG>::DL], const float* xin, const float* zbuf, float* gbuf,
                                 float* mirror, const float* wstore, const GeoPS<ALG>& ge, f4 (&gy)[PS<ALG>::DL]) {
    using P = PS<ALG>;
    constexpr int DL = P::DL, GC = P::GC, G = ALG::G, NP = ALG::P, QP = P::QP;
    const int c = ge.cn;
    const bool cv = lp.cvalid;
    const int cc = cv ? c : 0;
    const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, NP, B.w1_sub != 0);
    const WSrc sWRt{nullptr, wstore + B.lds_woff + wo.WR, B.O, B.CPo, 1};
    const WSrc sWLt{nullptr, wstore + B.lds_woff + wo.WL, B.O, B.CPo, 1};
    const MirrorOff mo = mirror_offsets(B.I, B.O, G, NP, B.w1_sub != 0);
    float* mir = mirror + B.lds_goff;
    float *d_b1 = mir + mo.b1, *d_sa = mir + mo.sa, *d_sb = mir + mo.sb, *d_w = mir + mo.w, *d_an = mir + mo.an;
    float *d_bL = mir + mo.bL, *d_la = mir + mo.la, *d_W1 = mir + mo.W1, *d_WR = mir + mo.WR, *d_WL = mir + mo.WL;
    float p_la, p_bL, p_b1, p_an[GC], p_sa[GC], p_sb[GC], p_wA[QP], p_wB[QP];

    // ---- MVLayerNorm backward
    f4 dot = splat(0.f);
#pragma unroll
    for (int j = 0; j < DL; ++j) dot += gout[j] * S.s[j];
    p_la = hsum(dot * S.invMn);          // partial (own half); the partner adds its own
    dot += partner4(dot);
    const f4 gMn = chan_sum4<2>(-(lp.la * dot) * S.invMn * S.invMn);
    const f4 inl = rcp4(S.nl);
    const f4 gqs = (gMn * (1.0f / float(B.O))) * (0.5f * S.qs) * (inl * inl * inl);
    f4 ggp[DL];
    static_for<0, DL>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        const f4 gs = (lp.la * gout[j]) * S.invMn + gqs * (2.0f * ge.template qs<j>()) * S.s[j];
        ggp[j] = cv ? gs * kInvSqrt2 : splat(0.f);
    });
    p_bL = ge.s == 0 ? hsum(ggp[0]) : 0.f;

    CSMPN_PHASE();
    // ---- d/dz from linear_left: gz = GL . WL^T ; gWL += GL (x) Z
    ps_store_tile<ALG>(ggp, gbuf, B.CPo, ge);
    tile_sync<VAR_WAVE>();
    f4 gz[DL];
#pragma unroll
    for (int j = 0; j < DL; ++j) gz[j] = splat(0.f);
    ps_linear<ALG, true>(gz, gbuf, B.CPo, B.KKo, sWLt, 0, ge);
    ps_weight_grad<ALG, true>(ggp, zbuf, B.CPo, B.O, B.O, B.NTo, ge, d_WL, true);

    CSMPN_PHASE();
    // ---- geometric product backward
    f4 gr[DL];
    ps_weighted_gp_bwd<ALG>(ggp, S.y, S.gate, S.R, S.invden, wstore + B.lds_woff + wo.w + (size_t)cc * NP, ge, gz, gr,
                            p_wA, p_wB);

    CSMPN_PHASE();
    // ---- NormalizationLayer backward -> gR
    f4 gR[DL];
    static_for<0, GC>([&](auto k) {
        constexpr int j0 = P::t.cstart[k], j1 = P::t.cstart[k + 1];
        f4 gden = splat(0.f), qR = splat(0.f);
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            gden -= gr[j] * S.R[j];
            qR += ge.template qs<j>() * S.R[j] * S.R[j];
        });
        gden *= S.invden[k] * S.invden[k];
        const f4 nu = smooth_abs_sqrt4(qR);
        p_an[k] = hsum(gden * (nu - 1.0f)) * lp.sg[k] * (1.0f - lp.sg[k]);
        const f4 inu = rcp4(nu);
        const f4 gq = (gden * lp.sg[k]) * (0.5f * qR) * (inu * inu * inu);
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            gR[j] = cv ? gr[j] * S.invden[k] + gq * (2.0f * ge.template qs<j>()) * S.R[j] : splat(0.f);
        });
    });
    tile_sync<VAR_WAVE>();
    ps_store_tile<ALG>(gR, gbuf, B.CPo, ge);
    tile_sync<VAR_WAVE>();
    ps_linear<ALG, true>(gz, gbuf, B.CPo, B.KKo, sWRt, 0, ge);
    ps_weight_grad<ALG, true>(gR, zbuf, B.CPo, B.O, B.O, B.NTo, ge, d_WR, true);

    CSMPN_PHASE();
    // ---- MVSiLU backward -> gy
    static_for<0, GC>([&](auto k) {
        constexpr int j0 = P::t.cstart[k], j1 = P::t.cstart[k + 1];
        f4 ggate = splat(0.f), u = splat(0.f);
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            ggate += gz[j] * S.y[j];
            u += ge.template qs<j>() * S.y[j] * S.y[j];
        });
        const bool scalar_inv = k == 0 && ge.s == 0;
        if (scalar_inv) u = S.y[0];
        const f4 gpre = ggate * S.gate[k] * (1.0f - S.gate[k]);
        p_sa[k] = hsum(gpre * u);
        p_sb[k] = hsum(gpre);
        const f4 gu = gpre * lp.sa[k];
        static_for<j0, j1>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            f4 v = gz[j] * S.gate[k];
            const f4 quad = gu * (2.0f * ge.template qs<j>()) * S.y[j];
            if constexpr (k == 0) v += scalar_inv ? gu : quad;
            else v += quad;
            gy[j] = cv ? v : splat(0.f);
        });
    });
    p_b1 = ge.s == 0 ? hsum(gy[0]) : 0.f;

    CSMPN_PHASE();
    // ---- small-parameter gradients: sum over the 4 row quarters, one lane per (channel, parity) adds
    p_la = quarter_sum(p_la);
    p_bL = quarter_sum(p_bL);
    p_b1 = quarter_sum(p_b1);
#pragma unroll
    for (int k = 0; k < GC; ++k) {
        p_an[k] = quarter_sum(p_an[k]);
        p_sa[k] = quarter_sum(p_sa[k]);
        p_sb[k] = quarter_sum(p_sb[k]);
    }
#pragma unroll
    for (int q = 0; q < QP; ++q) { p_wA[q] = quarter_sum(p_wA[q]); p_wB[q] = quarter_sum(p_wB[q]); }
    if (cv && ge.q == 0) {
        atomicAdd(d_la + c, p_la);
        if (ge.s == 0) {
            atomicAdd(d_bL + c, p_bL);
            if (B.has_b1) atomicAdd(d_b1 + c, p_b1);
        }
#pragma unroll
        for (int k = 0; k < GC; ++k) {
            const int g = ge.grade(k);
            atomicAdd(d_an + c * G + g, p_an[k]);
            atomicAdd(d_sa + c * G + g, p_sa[k]);
            atomicAdd(d_sb + c * G + g, p_sb[k]);
        }
        static_for<0, QP>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            atomicAdd(d_w + (size_t)c * NP + (ge.s ? P::t.pid[1][0][q] : P::t.pid[0][0][q]), p_wA[q]);
            atomicAdd(d_w + (size_t)c * NP + (ge.s ? P::t.pid[1][1][q] : P::t.pid[0][1][q]),
                      p_wB[q] * (ge.s ? 1.0f : float(P::t.I2)));
        });
    }

    // ---- MVLinear weight gradient; gy tile to LDS for the transposed MVLinear
    tile_sync<VAR_WAVE>();
    ps_store_tile<ALG>(gy, gbuf, B.CPo, ge);
    ps_weight_grad<ALG, true>(gy, xin, B.CPi, B.I, B.O, B.NTi, ge, d_W1, B.w1_sub != 0);
    tile_sync<VAR_WAVE>();
}

// ---------------------------------------------------------------------------------
// The row program (single-wave tiles, weights / parameters / gradient mirror in LDS).
// Forward: 512 threads per workgroup (VGPRs bounded to 128: 4 waves per SIMD with two
// workgroups per CU); backward: 512 threads (256 VGPRs: 2 waves per SIMD).
template <class ALG, int MODE, bool BWD>
__global__ void __launch_bounds__(256, (BWD || ALG::n >= 5) ? 1 : 3) cemlp_ps_kernel(const DevCemlp C_arg, const RowIO io_arg) {
    typedef const char __attribute__((address_space(4))) * KArgPtr;
    const KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr size_t kIoOffset = (sizeof(DevCemlp) + alignof(RowIO) - 1) / alignof(RowIO) * alignof(RowIO);
    const DevCemlp& C = *(const DevCemlp*)(const char*)ka;
    const RowIO& io = *(const RowIO*)(const char*)(ka + kIoOffset);
    (void)C_arg; (void)io_arg;
    using P = PS<ALG>;
    constexpr int D = ALG::D, G = ALG::G, DL = P::DL, R = P::R, NW = P::NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int RT = C.RT;
    const int rt = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const GeoPS<ALG> ge(lane);
    float* mirror = smem;
    float* wstore = smem + C.mirror_floats;
    float* base = smem + C.mirror_floats + C.wstore_floats + (size_t)rt * C.tile_floats;
    float* buf_in = base + C.off_in;
    float* buf_p0 = base + C.off_p0;
    float* buf_z = base + C.off_z;
    float* buf_g = base + C.off_g;
    int* tidx = reinterpret_cast<int*>(base + C.off_idx);

    for (int e = threadIdx.x; e < C.mirror_floats + C.wstore_floats; e += blockDim.x) smem[e] = 0.f;
    __syncthreads();
    for (int k = 0; k < C.nblk; ++k) {
        const DevBlock& B = C.b[k];
        const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, ALG::P, B.w1_sub != 0);
        float* ws = wstore + B.lds_woff;
        stage_weight(B.W1, ws + wo.W1, B.O, B.I, B.CPi, G, B.w1_sub != 0, threadIdx.x, blockDim.x);
        stage_weight(B.WR, ws + wo.WR, B.O, B.O, B.CPo, G, true, threadIdx.x, blockDim.x);
        stage_weight(B.WL, ws + wo.WL, B.O, B.O, B.CPo, G, true, threadIdx.x, blockDim.x);
        for (int e = threadIdx.x; e < B.O; e += blockDim.x) {
            ws[wo.b1 + e] = B.has_b1 ? B.b1[e] : 0.f;
            ws[wo.bL + e] = B.bL[e];
            ws[wo.la + e] = B.la[e];
        }
        for (int e = threadIdx.x; e < B.O * G; e += blockDim.x) {
            ws[wo.sa + e] = B.sa[e];
            ws[wo.sb + e] = B.sb[e];
            ws[wo.sg + e] = sigmoidf(B.an[e]);
        }
        for (int e = threadIdx.x; e < B.O * ALG::P; e += blockDim.x) ws[wo.w + e] = B.w[e];
    }
    __syncthreads();

    auto lane_params = [&](const DevBlock& B) -> PSLaneParams<ALG> {
        const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, ALG::P, B.w1_sub != 0);
        return ps_lane_params<ALG>(B, wstore + B.lds_woff, wo, ge);
    };
    const DevBlock& B0 = C.b[0];
    const DevBlock& BL = C.b[C.nblk - 1];
    const long ntiles = (io.rows + R - 1) / R;
    const long tiles_per_iter = (long)gridDim.x * RT;
    const long niter = (ntiles + tiles_per_iter - 1) / tiles_per_iter;
    auto save_off = [&](int kb) -> size_t {
        size_t o = 0;
        for (int j = 0; j + 1 < kb; ++j) o += (size_t)C.b[j].O;
        return o * (size_t)io.rows * D;
    };
    const bool use_saved = BWD && io.saved != nullptr && C.nblk > 1;
    TileIdx nidx = load_tile_indices<R>(io, ((long)blockIdx.x * RT + rt) * R, lane);
    const int c = ge.cn;

    for (long iter = 0; iter < niter; ++iter) {
        const long tile = iter * tiles_per_iter + (long)blockIdx.x * RT + rt;
        const long row0 = tile * R;
        store_tile_indices<R>(nidx, tidx, lane);
        tile_sync<VAR_WAVE>();
        nidx = load_tile_indices<R>(io, row0 + tiles_per_iter * R, lane);
        if (use_saved) {
            stage_plain<ALG, 1>(io.saved + save_off(C.nblk - 1), BL.I, io.rows, buf_in, BL.CPi, row0, lane, 64);
        } else {
            stage_input<ALG, 1, kModeSegs<MODE>>(io, buf_in, tidx, B0.CPi, row0, lane, 64);
        }
        tile_sync<VAR_WAVE>();

        if constexpr (!BWD) {
            const float* in = buf_in;
            f4 out[DL];
            for (int k = 0; k < C.nblk; ++k) {
                const DevBlock& B = C.b[k];
                const PSLaneParams<ALG> lp = lane_params(B);
                PSFwdState<ALG> S;
                ps_block_forward<ALG>(B, lp, in, buf_z, wstore, ge, S, out);
                if (k + 1 < C.nblk) {
                    tile_sync<VAR_WAVE>();
                    ps_store_tile<ALG>(out, buf_p0, B.CPo, ge);
                    if (io.save && c < B.O) {
                        float* sp = io.save + save_off(k + 1);
                        static_for<0, DL>([&](auto jj) {
                            constexpr int j = decltype(jj)::value;
                            const int bl = ge.template blade<j>();
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const long grow = row0 + ge.r0 + v;
                                if (grow < io.rows) sp[(grow * B.O + c) * D + bl] = out[j][v];
                            }
                        });
                    }
                    tile_sync<VAR_WAVE>();
                    in = buf_p0;
                }
            }
            const int O = BL.O;
            if constexpr (MODE == MODE_EDGE) {
                tile_sync<VAR_WAVE>();
                ps_store_dense<ALG>(out, buf_g, O, c, ge);
                tile_sync<VAR_WAVE>();
                scatter_tile<ALG, 1>(buf_g, O * D, tidx, nullptr, io.agg, lane);
                tile_sync<VAR_WAVE>();
            } else {
                if (c < O) {
                    float res[DL][4];
                    static_for<0, DL>([&](auto jj) {
                        constexpr int j = decltype(jj)::value;
                        const int bl = ge.template blade<j>();
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const long grow = row0 + ge.r0 + v;
                            res[j][v] = (MODE == MODE_NODE && io.resid && grow < io.rows) ? io.resid[(grow * O + c) * D + bl] : 0.f;
                        }
                    });
                    static_for<0, DL>([&](auto jj) {
                        constexpr int j = decltype(jj)::value;
                        const int bl = ge.template blade<j>();
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const long grow = row0 + ge.r0 + v;
                            if (grow < io.rows) io.y[(grow * O + c) * D + bl] = out[j][v] + res[j][v];
                        }
                    });
                }
            }
        } else {
            const int OL = BL.O;
            f4 gout[DL];
            static_for<0, DL>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                const int bl = ge.template blade<j>();
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const long grow = row0 + ge.r0 + v;
                    const bool ok = grow < io.rows && c < OL;
                    long srow = grow;
                    if (MODE == MODE_EDGE && ok) srow = tidx[ge.r0 + v];
                    gout[j][v] = ok ? io.gy[(srow * OL + c) * D + bl] : 0.f;
                }
            });
            ps_park<ALG>(gout, buf_g, lane);
            for (int k = C.nblk - 1; k >= 0; --k) {
                const DevBlock& B = C.b[k];
                const float* in = buf_in;
                if (use_saved && k + 1 < C.nblk) {
                    if (k == 0) stage_input<ALG, 1, kModeSegs<MODE>>(io, buf_in, tidx, B0.CPi, row0, lane, 64);
                    else stage_plain<ALG, 1>(io.saved + save_off(k), B.I, io.rows, buf_in, B.CPi, row0, lane, 64);
                    tile_sync<VAR_WAVE>();
                }
                for (int j = 0; !use_saved && j < k; ++j) {
                    const DevBlock& Bj = C.b[j];
                    const PSLaneParams<ALG> lpj = lane_params(Bj);
                    PSFwdState<ALG> Sj;
                    f4 oj[DL];
                    ps_block_forward<ALG>(Bj, lpj, in, buf_z, wstore, ge, Sj, oj);
                    tile_sync<VAR_WAVE>();
                    ps_store_tile<ALG>(oj, buf_p0, Bj.CPo, ge);
                    tile_sync<VAR_WAVE>();
                    in = buf_p0;
                }
                const PSLaneParams<ALG> lp = lane_params(B);
                f4 gy[DL];
                {
                    PSFwdState<ALG> S;
                    f4 unused[DL];
                    ps_block_forward<ALG>(B, lp, in, buf_z, wstore, ge, S, unused);
                    tile_sync<VAR_WAVE>();
                    ps_unpark<ALG>(gout, buf_g, lane);
                    tile_sync<VAR_WAVE>();
                    ps_block_backward<ALG>(B, lp, S, gout, in, buf_z, buf_g, mirror, wstore, ge, gy);
                }
                const WOff wo = wstore_offsets(B.O, B.CPi, B.CPo, G, ALG::P, B.w1_sub != 0);
                const WSrc sW1t{nullptr, wstore + B.lds_woff + wo.W1, B.O, B.CPi, B.w1_sub};
                if (k > 0) {
#pragma unroll
                    for (int j = 0; j < DL; ++j) gout[j] = splat(0.f);
                    ps_linear<ALG, true>(gout, buf_g, B.CPo, B.KKo, sW1t, 0, ge);
                    tile_sync<VAR_WAVE>();
                    ps_park<ALG>(gout, buf_g, lane);
                    tile_sync<VAR_WAVE>();
                } else {
                    float* stage = buf_in;
                    const int Cs0 = io.seg[0].ch;
                    for (int it = 0; it < B.NTi; ++it) {
                        bool wanted = false;
                        for (int t = 0; t < io.nseg; ++t) {
                            const bool overlaps = io.seg[t].off < NW * (it + 1) && io.seg[t].off + io.seg[t].ch > NW * it;
                            wanted |= overlaps && ((MODE == MODE_EDGE && t == 0) || io.gx[t] != nullptr);
                        }
                        if (!wanted) continue;
                        f4 gx[DL];
#pragma unroll
                        for (int j = 0; j < DL; ++j) gx[j] = splat(0.f);
                        ps_linear<ALG, true>(gx, buf_g, B.CPo, B.KKo, sW1t, it, ge);
                        const int i = NW * it + ge.cn;
                        int s = -1;
                        for (int t = 0; t < io.nseg; ++t)
                            if (i >= io.seg[t].off && i < io.seg[t].off + io.seg[t].ch) s = t;
                        if (MODE == MODE_EDGE && s == 0) {
                            ps_store_dense<ALG>(gx, stage, Cs0, i, ge);
                        } else if (s >= 0 && io.gx[s]) {
                            const Seg& sg = io.seg[s];
                            const int ci = i - sg.off;
                            static_for<0, DL>([&](auto jj) {
                                constexpr int j = decltype(jj)::value;
                                const int bl = ge.template blade<j>();
#pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    const long grow = row0 + ge.r0 + v;
                                    if (grow < io.rows) {
                                        long trow = grow;
                                        if (MODE == MODE_EDGE) trow = tidx[2 * R + ge.r0 + v];
                                        float val = gx[j][v];
                                        if (sg.deg) { const int dg = sg.deg[grow]; val *= 1.0f / float(dg > 1 ? dg : 1); }
                                        if (MODE == MODE_NODE && s == 0 && io.resid_bwd) val += io.gy[(grow * OL + ci) * D + bl];
                                        io.gx[s][(trow * sg.ch + ci) * D + bl] = val;
                                    }
                                }
                            });
                        }
                    }
                    if constexpr (MODE == MODE_EDGE) {
                        tile_sync<VAR_WAVE>();
                        if (io.gx[0]) scatter_tile<ALG, 1>(stage, Cs0 * D, tidx, tidx + R, io.gx[0], lane);
                    }
                    tile_sync<VAR_WAVE>();
                }
            }
        }
    }

    if constexpr (BWD) {
        __syncthreads();
        for (int k = 0; k < C.nblk; ++k) flush_mirror<ALG>(C.b[k], mirror, threadIdx.x, blockDim.x);
    }
}

}  // namespace csmpn
