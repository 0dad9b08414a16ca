// Compile-time Clifford algebra tables for the HIP kernels (gfx950).
//
// Restates, as constexpr integer arithmetic, what the reference builds at
// import time in Python:
//   blade order (grade, then lexicographic)   csmpn/algebra/metric.py:18-29
//   reordering sign * metric factor           csmpn/algebra/metric.py:50-89
//   Cayley table cayley[left, out, right]     csmpn/algebra/metric.py:92-120
//   quadratic-form sign beta_d*cayley[d,0,d]  csmpn/algebra/cliffordalgebra.py:69-71,119-146
//   grade paths (g_left, g_out, g_right)      csmpn/algebra/cliffordalgebra.py:238-252
//
// The Cayley tensor has exactly one non-zero per (left, right) pair, so the
// kernels never touch a dense [D,D,D] tensor: (out index, sign) are immediates
// in fully unrolled code.
//
// Template parameters: N = number of generators (D = 2^N blades), NEG = bit
// mask of generators with metric -1 (all others +1). Metrics with entries
// outside {+1,-1} are rejected by the host before launch.
#pragma once

namespace csmpn {

constexpr int popcount_u(unsigned x) {
    int c = 0;
    while (x) { c += int(x & 1u); x >>= 1; }
    return c;
}

template <int N>
struct BladeOrder {
    int bitmap[1 << N];   // index -> bitmap
    int index[1 << N];    // bitmap -> index
    int grade[1 << N];    // index -> grade
    int gstart[N + 2];    // grade -> first blade index (gstart[N+1] = D)
};

template <int N>
constexpr BladeOrder<N> make_blade_order() {
    BladeOrder<N> b{};
    int pos = 0;
    for (int g = 0; g <= N; ++g) {
        b.gstart[g] = pos;
        int comb[N + 1] = {};
        for (int t = 0; t < g; ++t) comb[t] = t;
        while (true) {
            int bm = 0;
            for (int t = 0; t < g; ++t) bm |= 1 << comb[t];
            b.bitmap[pos] = bm;
            b.grade[pos] = g;
            b.index[bm] = pos;
            ++pos;
            int t = g - 1;
            while (t >= 0 && comb[t] == N - g + t) --t;
            if (t < 0) break;
            ++comb[t];
            for (int u = t + 1; u < g; ++u) comb[u] = comb[u - 1] + 1;
        }
    }
    b.gstart[N + 1] = pos;
    return b;
}

// sign of e_a e_b (bitmaps) for a diagonal metric with -1 on the NEG bits
constexpr int product_sign(unsigned a, unsigned b, unsigned neg) {
    int s = 0;
    unsigned t = a >> 1;
    while (t) { s += popcount_u(t & b); t >>= 1; }
    s += popcount_u(a & b & neg);
    return (s & 1) ? -1 : 1;
}

template <int N, unsigned NEG>
struct AlgTables {
    static constexpr int D = 1 << N, G = N + 1;
    BladeOrder<N> bo;
    signed char sign[D][D];      // sign of blade_i * blade_k
    unsigned char out[D][D];     // index of blade_i * blade_k
    signed char qsign[D];        // beta_d * cayley[d,0,d]
    signed char path_id[G][G][G];  // (g_left, g_out, g_right) -> rank among non-zero paths, -1 if none
    signed char path_g[G * G * G][3];
    int n_paths;
};

template <int N, unsigned NEG>
constexpr AlgTables<N, NEG> make_alg_tables() {
    AlgTables<N, NEG> t{};
    t.bo = make_blade_order<N>();
    constexpr int D = 1 << N, G = N + 1;
    bool present[G][G][G] = {};
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) {
            unsigned a = unsigned(t.bo.bitmap[i]), b = unsigned(t.bo.bitmap[k]);
            int j = t.bo.index[a ^ b];
            t.out[i][k] = (unsigned char)j;
            t.sign[i][k] = (signed char)product_sign(a, b, NEG);
            present[t.bo.grade[i]][t.bo.grade[j]][t.bo.grade[k]] = true;
        }
    for (int d = 0; d < D; ++d) {
        int g = t.bo.grade[d];
        int beta = ((g * (g - 1) / 2) & 1) ? -1 : 1;
        t.qsign[d] = (signed char)(beta * t.sign[d][d]);
    }
    int p = 0;
    for (int a = 0; a < G; ++a)
        for (int b = 0; b < G; ++b)
            for (int c = 0; c < G; ++c) {
                if (present[a][b][c]) {
                    t.path_g[p][0] = (signed char)a;
                    t.path_g[p][1] = (signed char)b;
                    t.path_g[p][2] = (signed char)c;
                    t.path_id[a][b][c] = (signed char)p++;
                } else {
                    t.path_id[a][b][c] = -1;
                }
            }
    t.n_paths = p;
    return t;
}

template <int N, unsigned NEG>
struct Alg {
    static constexpr int n = N, D = 1 << N, G = N + 1;
    static constexpr unsigned neg = NEG;
    static constexpr AlgTables<N, NEG> t = make_alg_tables<N, NEG>();
    static constexpr int P = t.n_paths;
    static constexpr int gstart(int g) { return t.bo.gstart[g]; }
    static constexpr int gsize(int g) { return t.bo.gstart[g + 1] - t.bo.gstart[g]; }
    static constexpr int grade(int d) { return t.bo.grade[d]; }
};

// compile-time loop with integral-constant index
template <int V> struct IC { static constexpr int value = V; constexpr operator int() const { return V; } };
template <int B, int E, class F>
__host__ __device__ __forceinline__ constexpr void static_for(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

}  // namespace csmpn
