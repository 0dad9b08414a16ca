// C++ CPU twin of the hot path (include/csmpn_cpu.h). TEST INFRASTRUCTURE: a checker and a CPU
// baseline, never loaded by the product package.
//
// Restates, per row and in the sparse sign-table formulation:
//   tables        csmpn/algebra/metric.py:18-120, cliffordalgebra.py:119-146,238-252
//   CEMLP block   csmpn/models/cegnn_utils.py:34-155,287-338 (MVLinear, MVSiLU, NormalizationLayer,
//                 SteerableGeometricProductLayer, MVLayerNorm)
//   EGCL          csmpn/models/cegnn_utils.py:254-284 + PyG propagate (gather, scatter sum | mean)
// and the analytic backward of each (the reference has none: autograd derives it).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef CSMPN_CPU_REAL64
// float64 build of this same source (oracle/_build/libcsmpn_cpu64.so, round 4): every `float` from here on - the API's
// pointers and the descriptor structs included - is a double, so the library takes and returns float64 arrays and computes
// in float64. It is the TRUTH of the full-size GPU parity tests (tests/test_full_size_twin.py); the float32 build next to
// it is their yardstick. The f-suffixed literals that remain are exactly representable or are the reference's own
// float32 constants (1e-16f, 1e-6f).
#define float double
#endif

#include "../../include/csmpn_cpu.h"

namespace {

thread_local char g_err[256] = "";
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

struct Tables {
    int n = 0, D = 0, G = 0, P = 0;
    std::vector<int> grade, gstart, out, path;     // out[i*D+k], path[(gi*G+gj)*G+gk] (-1: none)
    std::vector<float> sign, qsign;                // sign[i*D+k] incl. the metric; qsign[d] = beta_d cayley[d,0,d]
};

int popcount(unsigned x) { int c = 0; while (x) { c += x & 1u; x >>= 1; } return c; }

bool build_tables(const float* metric, int n, Tables& t) {
    if (n < 1 || n > 6) return false;
    t.n = n; t.D = 1 << n; t.G = n + 1;
    const int D = t.D, G = t.G;
    std::vector<int> bm(D), idx(D);
    t.grade.assign(D, 0); t.gstart.assign(G + 1, 0);
    int pos = 0;
    for (int g = 0; g <= n; ++g) {            // blade order: by grade, then lexicographic (metric.py:18-29)
        t.gstart[g] = pos;
        std::vector<int> comb(g);
        for (int i = 0; i < g; ++i) comb[i] = i;
        while (true) {
            int b = 0;
            for (int i = 0; i < g; ++i) b |= 1 << comb[i];
            bm[pos] = b; idx[b] = pos; t.grade[pos] = g; ++pos;
            int i = g - 1;
            while (i >= 0 && comb[i] == n - g + i) --i;
            if (i < 0) break;
            ++comb[i];
            for (int u = i + 1; u < g; ++u) comb[u] = comb[u - 1] + 1;
        }
    }
    t.gstart[G] = pos;
    t.out.assign(D * D, 0); t.sign.assign(D * D, 0.f); t.qsign.assign(D, 0.f);
    std::vector<char> present(G * G * G, 0);
    for (int i = 0; i < D; ++i)
        for (int k = 0; k < D; ++k) {
            const unsigned a = bm[i], b = bm[k];
            int s = 0;
            for (unsigned u = a >> 1; u; u >>= 1) s += popcount(u & b);     // metric.py:50-79
            float c = (s & 1) ? -1.f : 1.f;
            for (int bit = 0; bit < n; ++bit) if ((a & b) >> bit & 1u) c *= metric[bit];
            const int j = idx[a ^ b];
            t.out[i * D + k] = j; t.sign[i * D + k] = c;
            if (c != 0.f) present[(t.grade[i] * G + t.grade[j]) * G + t.grade[k]] = 1;
        }
    for (int d = 0; d < D; ++d) {
        const int g = t.grade[d];
        const float beta = ((g * (g - 1) / 2) & 1) ? -1.f : 1.f;           // cliffordalgebra.py:69-71
        t.qsign[d] = beta * t.sign[d * D + d];
    }
    t.path.assign(G * G * G, -1);
    int p = 0;
    for (int e = 0; e < G * G * G; ++e) if (present[e]) t.path[e] = p++;   // row-major order of the True entries
    t.P = p;
    return true;
}

inline float sigmoid(float x) { return 1.0f / (1.0f + std::exp(-x)); }
inline float smooth_abs_sqrt(float q) { return std::sqrt(std::sqrt(q * q + 1e-16f)); }
constexpr float kEps = 1e-6f;
const float kInvSqrt2 = 0.70710678118654752440;

// forward state of one block on one row
struct State {
    std::vector<float> x, y, gate, z, R, invden, s, qs, nl;
    float invMn = 0.f;
    void size(int I, int O, int D, int G) {
        x.resize((size_t)I * D); y.resize((size_t)O * D); gate.resize((size_t)O * G); z.resize((size_t)O * D);
        R.resize((size_t)O * D); invden.resize((size_t)O * G); s.resize((size_t)O * D); qs.resize(O); nl.resize(O);
    }
};

struct Block {
    int I, O, sub;
    const float *W1, *b1, *sa, *sb, *w, *an, *WR, *WL, *bL, *la;
};

struct Grads {   // flat per-thread accumulators of one block, reference layouts
    std::vector<float> W1, b1, sa, sb, w, an, WR, WL, bL, la;
    void size(const Block& b, int G, int P) {
        W1.assign((size_t)b.O * b.I * (b.sub ? G : 1), 0.f); b1.assign(b.O, 0.f); sa.assign((size_t)b.O * G, 0.f);
        sb.assign((size_t)b.O * G, 0.f); w.assign((size_t)b.O * P, 0.f); an.assign((size_t)b.O * G, 0.f);
        WR.assign((size_t)b.O * b.O * G, 0.f); WL.assign((size_t)b.O * b.O * G, 0.f); bL.assign(b.O, 0.f); la.assign(b.O, 0.f);
    }
};

// out[O*D] from S.x (already filled)
void block_forward(const Tables& t, const Block& B, State& S, float* out) {
    const int D = t.D, G = t.G, I = B.I, O = B.O;
    for (int o = 0; o < O; ++o)
        for (int d = 0; d < D; ++d) {
            const int g = t.grade[d];
            float acc = (d == 0 && B.b1) ? B.b1[o] : 0.f;
            for (int i = 0; i < I; ++i) acc += (B.sub ? B.W1[(o * I + i) * G + g] : B.W1[o * I + i]) * S.x[i * D + d];
            S.y[o * D + d] = acc;
        }
    for (int o = 0; o < O; ++o)
        for (int g = 0; g < G; ++g) {
            float u = 0.f;
            if (g == 0) u = S.y[o * D];
            else for (int d = t.gstart[g]; d < t.gstart[g + 1]; ++d) u += t.qsign[d] * S.y[o * D + d] * S.y[o * D + d];
            const float gt = sigmoid(B.sa[o * G + g] * u + B.sb[o * G + g]);
            S.gate[o * G + g] = gt;
            for (int d = t.gstart[g]; d < t.gstart[g + 1]; ++d) S.z[o * D + d] = gt * S.y[o * D + d];
        }
    std::vector<float> L((size_t)O * D), r((size_t)O * D);
    for (int o = 0; o < O; ++o)
        for (int d = 0; d < D; ++d) {
            const int g = t.grade[d];
            float ar = 0.f, al = (d == 0) ? B.bL[o] : 0.f;
            for (int i = 0; i < O; ++i) {
                ar += B.WR[(o * O + i) * G + g] * S.z[i * D + d];
                al += B.WL[(o * O + i) * G + g] * S.z[i * D + d];
            }
            S.R[o * D + d] = ar; L[o * D + d] = al;
        }
    for (int o = 0; o < O; ++o)
        for (int g = 0; g < G; ++g) {
            float qq = 0.f;
            for (int d = t.gstart[g]; d < t.gstart[g + 1]; ++d) qq += t.qsign[d] * S.R[o * D + d] * S.R[o * D + d];
            const float m = sigmoid(B.an[o * G + g]) * (smooth_abs_sqrt(qq) - 1.0f) + 1.0f;
            S.invden[o * G + g] = 1.0f / (m + kEps);
            for (int d = t.gstart[g]; d < t.gstart[g + 1]; ++d) r[o * D + d] = S.R[o * D + d] * S.invden[o * G + g];
        }
    float nsum = 0.f;
    for (int o = 0; o < O; ++o) {
        float* Lo = &L[o * D];
        for (int i = 0; i < D; ++i) {
            const float zi = S.z[o * D + i];
            for (int k = 0; k < D; ++k) {
                const float sg = t.sign[i * D + k];
                if (sg == 0.f) continue;
                const int j = t.out[i * D + k];
                const int p = t.path[(t.grade[i] * G + t.grade[j]) * G + t.grade[k]];
                Lo[j] += sg * B.w[o * t.P + p] * zi * r[o * D + k];
            }
        }
        float qs = 0.f;
        for (int d = 0; d < D; ++d) { S.s[o * D + d] = Lo[d] * kInvSqrt2; qs += t.qsign[d] * S.s[o * D + d] * S.s[o * D + d]; }
        S.qs[o] = qs; S.nl[o] = smooth_abs_sqrt(qs); nsum += S.nl[o];
    }
    S.invMn = 1.0f / (nsum / float(O) + kEps);
    for (int o = 0; o < O; ++o)
        for (int d = 0; d < D; ++d) out[o * D + d] = B.la[o] * S.s[o * D + d] * S.invMn;
}

// gx[I*D] (may be null) from gout[O*D]; parameter gradients added to Gr
void block_backward(const Tables& t, const Block& B, const State& S, const float* gout, Grads& Gr, float* gx) {
    const int D = t.D, G = t.G, I = B.I, O = B.O, P = t.P;
    std::vector<float> ggp((size_t)O * D), gz((size_t)O * D, 0.f), gr((size_t)O * D, 0.f), gR((size_t)O * D), gy((size_t)O * D);
    float gMn = 0.f;
    std::vector<float> dot(O);
    for (int o = 0; o < O; ++o) {
        float dt = 0.f;
        for (int d = 0; d < D; ++d) dt += gout[o * D + d] * S.s[o * D + d];
        dot[o] = dt;
        Gr.la[o] += dt * S.invMn;
        gMn -= B.la[o] * dt;
    }
    gMn *= S.invMn * S.invMn / float(O);
    for (int o = 0; o < O; ++o) {
        const float inl = 1.0f / S.nl[o];
        const float gqs = gMn * 0.5f * S.qs[o] * inl * inl * inl;
        for (int d = 0; d < D; ++d)
            ggp[o * D + d] = (B.la[o] * gout[o * D + d] * S.invMn + gqs * 2.0f * t.qsign[d] * S.s[o * D + d]) * kInvSqrt2;
        Gr.bL[o] += ggp[o * D];
    }
    // linear_left
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < O; ++i)
            for (int d = 0; d < D; ++d) {
                const int g = t.grade[d];
                gz[i * D + d] += B.WL[(o * O + i) * G + g] * ggp[o * D + d];
                Gr.WL[(o * O + i) * G + g] += ggp[o * D + d] * S.z[i * D + d];
            }
    // geometric product
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < D; ++i)
            for (int k = 0; k < D; ++k) {
                const float sg = t.sign[i * D + k];
                if (sg == 0.f) continue;
                const int j = t.out[i * D + k];
                const int p = t.path[(t.grade[i] * G + t.grade[j]) * G + t.grade[k]];
                const float w = B.w[o * P + p], g = sg * ggp[o * D + j];
                const float zi = S.z[o * D + i], rk = S.R[o * D + k] * S.invden[o * G + t.grade[k]];
                gz[o * D + i] += w * g * rk;
                gr[o * D + k] += w * g * zi;
                Gr.w[o * P + p] += g * zi * rk;
            }
    // NormalizationLayer
    for (int o = 0; o < O; ++o)
        for (int g = 0; g < G; ++g) {
            float gden = 0.f, qR = 0.f;
            for (int d = t.gstart[g]; d < t.gstart[g + 1]; ++d) {
                gden -= gr[o * D + d] * S.R[o * D + d];
                qR += t.qsign[d] * S.R[o * D + d] * S.R[o * D + d];
            }
            const float inv = S.invden[o * G + g];
            gden *= inv * inv;
            const float nu = smooth_abs_sqrt(qR), sg = sigmoid(B.an[o * G + g]);
            Gr.an[o * G + g] += gden * (nu - 1.0f) * sg * (1.0f - sg);
            const float gq = gden * sg * 0.5f * qR / (nu * nu * nu);
            for (int d = t.gstart[g]; d < t.gstart[g + 1]; ++d)
                gR[o * D + d] = gr[o * D + d] * inv + gq * 2.0f * t.qsign[d] * S.R[o * D + d];
        }
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < O; ++i)
            for (int d = 0; d < D; ++d) {
                const int g = t.grade[d];
                gz[i * D + d] += B.WR[(o * O + i) * G + g] * gR[o * D + d];
                Gr.WR[(o * O + i) * G + g] += gR[o * D + d] * S.z[i * D + d];
            }
    // MVSiLU
    for (int o = 0; o < O; ++o)
        for (int g = 0; g < G; ++g) {
            float ggate = 0.f, u = 0.f;
            for (int d = t.gstart[g]; d < t.gstart[g + 1]; ++d) ggate += gz[o * D + d] * S.y[o * D + d];
            if (g == 0) u = S.y[o * D];
            else for (int d = t.gstart[g]; d < t.gstart[g + 1]; ++d) u += t.qsign[d] * S.y[o * D + d] * S.y[o * D + d];
            const float gt = S.gate[o * G + g];
            const float gpre = ggate * gt * (1.0f - gt);
            Gr.sa[o * G + g] += gpre * u; Gr.sb[o * G + g] += gpre;
            const float gu = gpre * B.sa[o * G + g];
            for (int d = t.gstart[g]; d < t.gstart[g + 1]; ++d)
                gy[o * D + d] = gz[o * D + d] * gt + (g == 0 ? gu : gu * 2.0f * t.qsign[d] * S.y[o * D + d]);
        }
    // MVLinear
    if (gx) std::fill(gx, gx + (size_t)I * D, 0.f);
    for (int o = 0; o < O; ++o) {
        if (B.b1) Gr.b1[o] += gy[o * D];
        for (int i = 0; i < I; ++i)
            for (int d = 0; d < D; ++d) {
                const int g = t.grade[d];
                const size_t wi = B.sub ? (size_t)(o * I + i) * G + g : (size_t)(o * I + i);
                Gr.W1[wi] += gy[o * D + d] * S.x[i * D + d];
                if (gx) gx[i * D + d] += B.W1[wi] * gy[o * D + d];
            }
    }
}

struct Model {
    std::vector<Block> blocks;
    std::vector<const csmpn_block_grads*> gptr;
    int in() const { return blocks.front().I; }
    int outc() const { return blocks.back().O; }
};

bool make_model(const csmpn_block_params* bp, int nb, Model& m) {
    if (!bp || nb < 1 || nb > CSMPN_MAX_BLOCKS) return false;
    for (int k = 0; k < nb; ++k) {
        const csmpn_block_params& b = bp[k];
        if (!b.lin_w || !b.silu_a || !b.silu_b || !b.gp_w || !b.norm_a || !b.right_w || !b.left_w || !b.left_b || !b.ln_a) return false;
        if (k > 0 && b.in_features != bp[k - 1].out_features) return false;
        m.blocks.push_back(Block{b.in_features, b.out_features, b.lin_subspaces ? 1 : 0, b.lin_w, b.lin_b, b.silu_a, b.silu_b,
                                 b.gp_w, b.norm_a, b.right_w, b.left_w, b.left_b, b.ln_a});
    }
    return true;
}

// per-thread workspace of a CEMLP
struct Work {
    std::vector<State> S;
    std::vector<Grads> G;
    std::vector<float> a, b;     // ping-pong activation / gradient rows
    void init(const Tables& t, const Model& m, bool grads) {
        S.resize(m.blocks.size());
        int mx = 0;
        for (size_t k = 0; k < m.blocks.size(); ++k) {
            S[k].size(m.blocks[k].I, m.blocks[k].O, t.D, t.G);
            mx = std::max(mx, std::max(m.blocks[k].I, m.blocks[k].O));
        }
        a.resize((size_t)mx * t.D); b.resize((size_t)mx * t.D);
        if (grads) { G.resize(m.blocks.size()); for (size_t k = 0; k < m.blocks.size(); ++k) G[k].size(m.blocks[k], t.G, t.P); }
    }
};

// x already in W.S[0].x; result in out[O_last * D]
void cemlp_forward(const Tables& t, const Model& m, Work& W, float* out) {
    for (size_t k = 0; k < m.blocks.size(); ++k) {
        float* dst = (k + 1 < m.blocks.size()) ? W.S[k + 1].x.data() : out;
        block_forward(t, m.blocks[k], W.S[k], dst);
    }
}
// gx[I_0 * D] may be null
void cemlp_backward(const Tables& t, const Model& m, Work& W, const float* gout, float* gx) {
    const float* g = gout;
    for (int k = (int)m.blocks.size() - 1; k >= 0; --k) {
        float* dst = k > 0 ? ((g == W.a.data()) ? W.b.data() : W.a.data()) : gx;
        block_backward(t, m.blocks[k], W.S[k], g, W.G[k], dst);
        g = dst;
    }
}

void add_grads(const std::vector<Work>& works, const Model& m, const csmpn_block_grads* gr) {
    for (size_t k = 0; k < m.blocks.size(); ++k) {
        const csmpn_block_grads& g = gr[k];
        auto acc = [&](float* dst, std::vector<float> Grads::*f) {
            if (!dst) return;
            const size_t n = (works[0].G[k].*f).size();
            for (size_t w = 0; w < works.size(); ++w)          // fixed thread order: deterministic
                for (size_t e = 0; e < n; ++e) dst[e] += (works[w].G[k].*f)[e];
        };
        acc(g.lin_w, &Grads::W1); acc(g.lin_b, &Grads::b1); acc(g.silu_a, &Grads::sa); acc(g.silu_b, &Grads::sb);
        acc(g.gp_w, &Grads::w); acc(g.norm_a, &Grads::an); acc(g.right_w, &Grads::WR); acc(g.left_w, &Grads::WL);
        acc(g.left_b, &Grads::bL); acc(g.ln_a, &Grads::la);
    }
}

int n_threads(int req) {
#ifdef _OPENMP
    return req > 0 ? req : omp_get_max_threads();
#else
    (void)req;
    return 1;
#endif
}

}  // namespace

extern "C" {

const char* csmpn_cpu_last_error(void) { return g_err; }

int csmpn_cemlp_cpu(const float* metric, int n, const csmpn_block_params* blocks, const csmpn_block_grads* grads,
                    int n_blocks, const float* x, int64_t rows, const float* gy, float* y, float* gx, int32_t threads) {
    Tables t;
    if (!build_tables(metric, n, t)) return fail(CSMPN_ERR_UNSUPPORTED, "n=%d not in 1..6", n);
    Model m;
    if (!make_model(blocks, n_blocks, m)) return fail(CSMPN_ERR_INVALID, "bad block descriptors");
    if (gy && !grads) return fail(CSMPN_ERR_INVALID, "grads is null");
    const int nt = n_threads(threads), D = t.D, I = m.in(), O = m.outc();
    std::vector<Work> works(nt);
    for (auto& w : works) w.init(t, m, gy != nullptr);
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
#ifdef _OPENMP
        Work& W = works[omp_get_thread_num()];
#else
        Work& W = works[0];
#endif
        std::copy(x + r * I * D, x + (r + 1) * I * D, W.S[0].x.begin());
        std::vector<float> out((size_t)O * D);
        cemlp_forward(t, m, W, out.data());
        if (y) std::copy(out.begin(), out.end(), y + r * O * D);
        if (gy) cemlp_backward(t, m, W, gy + r * O * D, gx ? gx + r * I * D : nullptr);
    }
    if (gy) add_grads(works, m, grads);
    return CSMPN_OK;
}

int csmpn_egcl_layer_cpu(const float* metric, int n,
                         const csmpn_block_params* edge_blocks, const csmpn_block_grads* edge_grads, int n_edge_blocks,
                         const csmpn_block_params* node_blocks, const csmpn_block_grads* node_grads, int n_node_blocks,
                         const float* h, int32_t C, const int64_t* edge_index, int64_t E, int64_t N,
                         const float* edge_attr, int32_t A, const float* node_attr, int32_t T, int32_t mean_aggr,
                         int32_t residual, const float* gout, float* out, float* gh, float* g_edge_attr,
                         float* g_node_attr, int32_t threads) {
    Tables t;
    if (!build_tables(metric, n, t)) return fail(CSMPN_ERR_UNSUPPORTED, "n=%d not in 1..6", n);
    Model em, nm;
    if (!make_model(edge_blocks, n_edge_blocks, em) || !make_model(node_blocks, n_node_blocks, nm))
        return fail(CSMPN_ERR_INVALID, "bad block descriptors");
    const int D = t.D, O = em.outc(), ON = nm.outc();
    if (em.in() != C + A) return fail(CSMPN_ERR_INVALID, "edge model in_features %d != %d + %d", em.in(), C, A);
    if (nm.in() != C + O + T) return fail(CSMPN_ERR_INVALID, "node model in_features %d != %d + %d + %d", nm.in(), C, O, T);
    if (residual && ON != C) return fail(CSMPN_ERR_INVALID, "residual needs out_features == channels");
    if ((A > 0 && !edge_attr) || (T > 0 && !node_attr)) return fail(CSMPN_ERR_INVALID, "attribute tensor is null");
    if (gout && (!edge_grads || !node_grads || !gh)) return fail(CSMPN_ERR_INVALID, "gradient outputs missing");
    for (int64_t e = 0; e < 2 * E; ++e)
        if (edge_index[e] < 0 || edge_index[e] >= N) return fail(CSMPN_ERR_INVALID, "edge_index has entries outside [0, %lld)", (long long)N);
    const int nt = n_threads(threads);
    const bool bwd = gout != nullptr;
    const int64_t* src = edge_index;
    const int64_t* dst = edge_index + E;
    // incoming edges of every node, ascending edge id
    std::vector<int64_t> row_ptr(N + 1, 0), by_dst(E);
    for (int64_t e = 0; e < E; ++e) ++row_ptr[dst[e] + 1];
    for (int64_t v = 0; v < N; ++v) row_ptr[v + 1] += row_ptr[v];
    {
        std::vector<int64_t> cur(row_ptr.begin(), row_ptr.end() - 1);
        for (int64_t e = 0; e < E; ++e) by_dst[cur[dst[e]]++] = e;
    }
    std::vector<Work> ew(nt), nw(nt);
    for (auto& w : ew) w.init(t, em, bwd);
    for (auto& w : nw) w.init(t, nm, bwd);
    std::vector<float> msg((size_t)E * O * D), agg((size_t)N * O * D, 0.f);
    auto edge_input = [&](int64_t e, float* x) {
        const float* hi = h + dst[e] * C * D;
        const float* hj = h + src[e] * C * D;
        for (int f = 0; f < C * D; ++f) x[f] = hi[f] - hj[f];
        if (A > 0) std::copy(edge_attr + e * A * D, edge_attr + (e + 1) * A * D, x + (size_t)C * D);
    };
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t e = 0; e < E; ++e) {
#ifdef _OPENMP
        Work& W = ew[omp_get_thread_num()];
#else
        Work& W = ew[0];
#endif
        edge_input(e, W.S[0].x.data());
        cemlp_forward(t, em, W, &msg[(size_t)e * O * D]);
    }
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t v = 0; v < N; ++v) {
        float* a = &agg[(size_t)v * O * D];
        for (int64_t q = row_ptr[v]; q < row_ptr[v + 1]; ++q) {
            const float* mrow = &msg[(size_t)by_dst[q] * O * D];
            for (int f = 0; f < O * D; ++f) a[f] += mrow[f];
        }
    }
    auto node_scale = [&](int64_t v) {
        const int64_t dg = row_ptr[v + 1] - row_ptr[v];
        return mean_aggr ? 1.0f / float(dg > 1 ? dg : 1) : 1.0f;
    };
    auto node_input = [&](int64_t v, float* x) {
        std::copy(h + v * C * D, h + (v + 1) * C * D, x);
        const float sc = node_scale(v);
        for (int f = 0; f < O * D; ++f) x[(size_t)C * D + f] = agg[(size_t)v * O * D + f] * sc;
        if (T > 0) std::copy(node_attr + v * T * D, node_attr + (v + 1) * T * D, x + (size_t)(C + O) * D);
    };
    std::vector<float> g_agg(bwd ? (size_t)N * O * D : 0);
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t v = 0; v < N; ++v) {
#ifdef _OPENMP
        Work& W = nw[omp_get_thread_num()];
#else
        Work& W = nw[0];
#endif
        node_input(v, W.S[0].x.data());
        std::vector<float> o((size_t)ON * D), gx;
        cemlp_forward(t, nm, W, o.data());
        if (out) for (int f = 0; f < ON * D; ++f) out[v * ON * D + f] = o[f] + (residual ? h[v * C * D + f] : 0.f);
        if (bwd) {
            gx.resize((size_t)(C + O + T) * D);
            cemlp_backward(t, nm, W, gout + v * ON * D, gx.data());
            const float sc = node_scale(v);
            for (int f = 0; f < C * D; ++f) gh[v * C * D + f] = gx[f] + (residual ? gout[v * ON * D + f] : 0.f);
            for (int f = 0; f < O * D; ++f) g_agg[(size_t)v * O * D + f] = gx[(size_t)C * D + f] * sc;
            if (g_node_attr && T > 0) std::copy(gx.begin() + (size_t)(C + O) * D, gx.end(), g_node_attr + v * T * D);
        }
    }
    if (!bwd) return CSMPN_OK;
    std::vector<float> gxe((size_t)E * C * D);
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t e = 0; e < E; ++e) {
#ifdef _OPENMP
        Work& W = ew[omp_get_thread_num()];
#else
        Work& W = ew[0];
#endif
        edge_input(e, W.S[0].x.data());
        std::vector<float> o((size_t)O * D), gx((size_t)(C + A) * D);
        cemlp_forward(t, em, W, o.data());
        cemlp_backward(t, em, W, &g_agg[(size_t)dst[e] * O * D], gx.data());
        std::copy(gx.begin(), gx.begin() + (size_t)C * D, &gxe[(size_t)e * C * D]);
        if (g_edge_attr && A > 0) std::copy(gx.begin() + (size_t)C * D, gx.end(), g_edge_attr + e * A * D);
    }
    // d/dh: +g to the target, -g to the source, ascending edge id per node (deterministic)
    std::vector<int64_t> rp2(N + 1, 0), by_src(E);
    for (int64_t e = 0; e < E; ++e) ++rp2[src[e] + 1];
    for (int64_t v = 0; v < N; ++v) rp2[v + 1] += rp2[v];
    {
        std::vector<int64_t> cur(rp2.begin(), rp2.end() - 1);
        for (int64_t e = 0; e < E; ++e) by_src[cur[src[e]]++] = e;
    }
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int64_t v = 0; v < N; ++v) {
        float* g = gh + v * C * D;
        for (int64_t q = row_ptr[v]; q < row_ptr[v + 1]; ++q) {
            const float* r = &gxe[(size_t)by_dst[q] * C * D];
            for (int f = 0; f < C * D; ++f) g[f] += r[f];
        }
        for (int64_t q = rp2[v]; q < rp2[v + 1]; ++q) {
            const float* r = &gxe[(size_t)by_src[q] * C * D];
            for (int f = 0; f < C * D; ++f) g[f] -= r[f];
        }
    }
    add_grads(ew, em, edge_grads);
    add_grads(nw, nm, node_grads);
    return CSMPN_OK;
}

}  // extern "C"
