// Host-side problem construction of RunOpt_NLP (include/eepacc_nlp.h: eepacc_nlp_problem_from_settings): the lookup
// tables ABO/RunOpt_NLP.m:63-184 builds before it formulates the problem -- stop profile, traffic-light profiles and
// phases, the velocity-incentive profile through minPWA / SaturateSlopePWA / FixCrossingPWA / SimplifyPWA
// (ABO/Functions/PWA_function_manipulation/*.m).  Plain C++, no GPU: what a MEX gateway for RunOpt_NLP needs in
// front of eepacc_nlp_create.  Vectors are 0-based here; the reference's loop bounds are kept in the comments.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/eepacc_nlp.h"

namespace eepacc { int set_error(int code, const std::string& msg); }

namespace {

using Vec = std::vector<double>;
struct Pwa { Vec d, v; };

// InterpPWA.m:14-27
double interp_pwa(double x, const Pwa& f) {
    const size_t n = f.d.size();
    if (x < f.d[0]) return f.v[0];
    if (x > f.d[n - 1]) return f.v[n - 1];
    for (size_t i = 0; i + 1 < n; ++i)
        if (x >= f.d[i] && x <= f.d[i + 1]) {
            const double frac = (x - f.d[i]) / (f.d[i + 1] - f.d[i]);
            return f.v[i] + frac * (f.v[i + 1] - f.v[i]);
        }
    return f.v[n - 1];
}

// SimplifyPWA.m:14-49 (x/0 = +-inf and 0/0 = nan as in MATLAB; nan ~= nan keeps the point)
Pwa simplify_pwa(const Pwa& f) {
    const size_t n = f.d.size();
    Pwa a;
    a.d.push_back(f.d[0]); a.v.push_back(f.v[0]);
    for (size_t i = 1; i + 1 < n; ++i) {                              // i = 2 : length-1
        const double prev = (f.v[i] - f.v[i - 1]) / (f.d[i] - f.d[i - 1]);
        const double curr = (f.v[i + 1] - f.v[i]) / (f.d[i + 1] - f.d[i]);
        if (prev != curr) { a.d.push_back(f.d[i]); a.v.push_back(f.v[i]); }
    }
    a.d.push_back(f.d[n - 1]); a.v.push_back(f.v[n - 1]);
    Pwa o;
    size_t j = 0;                                                     // j = 1
    for (size_t i = 0; i + 1 < a.d.size(); ++i) {
        if (a.d[i] == a.d[i + 1]) {
            if (a.v[i] != a.v[i + 1]) { o.d.push_back(a.d[i] - .1); o.v.push_back(a.v[i]); ++j; }
        } else {
            o.d.push_back(a.d[j]); o.v.push_back(a.v[j]); ++j;        // sic: doms(j), :43-44
        }
    }
    o.d.push_back(a.d.back()); o.v.push_back(a.v.back());
    return o;
}

// minPWA.m:14-124
Pwa min_pwa(Pwa A, Pwa B) {
    if (A.d.front() != B.d.front()) {                                 // fix start
        if (A.d.front() > B.d.front()) { A.d.insert(A.d.begin(), B.d.front()); A.v.insert(A.v.begin(), A.v.front()); }
        else { B.d.insert(B.d.begin(), A.d.front()); B.v.insert(B.v.begin(), B.v.front()); }
    }
    if (A.d.back() != B.d.back()) {                                   // fix end
        if (A.d.back() > B.d.back()) { B.d.push_back(A.d.back()); B.v.push_back(B.v.back()); }
        else { A.d.push_back(B.d.back()); A.v.push_back(A.v.back()); }
    }
    A = simplify_pwa(A); B = simplify_pwa(B);
    for (Pwa* f : {&A, &B}) {                                         // dummy points
        const double e = f->d.back(), ev = f->v.back();
        f->d.push_back(e + 1); f->d.push_back(e + 2); f->v.push_back(ev); f->v.push_back(ev);
    }
    Pwa Cc;
    bool doneA = false, doneB = false;
    size_t iA = 0, iB = 0;
    for (;;) {
        const double Ad1 = A.d[iA], Av1 = A.v[iA], Ad2 = A.d[iA + 1], Av2 = A.v[iA + 1];
        const double Bd1 = B.d[iB], Bv1 = B.v[iB], Bd2 = B.d[iB + 1], Bv2 = B.v[iB + 1];
        const double As = (Av2 - Av1) / (Ad2 - Ad1), Bs = (Bv2 - Bv1) / (Bd2 - Bd1);
        if ((Av1 > Bv1 && Av2 < Bv2) || (Av1 < Bv1 && Av2 > Bv2)) {
            const double s1 = (Bv1 - Av1 + (Ad1 - Bd1) * Bs) / (As - Bs);
            const double Id = Ad1 + s1;
            if (Id >= Ad1 && Id <= Ad2 && Id >= Bd1 && Id <= Bd2) { Cc.d.push_back(Id); Cc.v.push_back(Av1 + As * s1); }
        }
        if (Ad2 < Bd2) {
            if (Av1 <= interp_pwa(Ad1, B)) { Cc.d.push_back(Ad1); Cc.v.push_back(Av1); }
            ++iA;
            if (iA + 2 == A.d.size()) doneA = true;                   // i_A == length(Adom) - 1 (1-based)
        } else {
            if (Bv1 <= interp_pwa(Bd1, A)) { Cc.d.push_back(Bd1); Cc.v.push_back(Bv1); }
            ++iB;
            if (iB + 2 == B.d.size()) doneB = true;
        }
        if (doneA && doneB) break;
        if (iA + 1 >= A.d.size() || iB + 1 >= B.d.size()) break;      // malformed input: the reference would index out of range
    }
    std::vector<size_t> order(Cc.d.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return Cc.d[x] < Cc.d[y]; });
    Pwa o;
    for (size_t q : order) { o.d.push_back(Cc.d[q]); o.v.push_back(Cc.v[q]); }
    return o;
}

// FixCrossingPWA.m:14-48
void fix_crossing_pwa(Pwa& f) {
    const Vec d0 = f.d;
    const size_t n = f.d.size();
    std::vector<size_t> cross;
    for (size_t i = 0; i + 1 < n; ++i) if (d0[i + 1] - d0[i] <= 0) cross.push_back(i);      // find(diff(doms) <= 0), 0-based
    for (size_t cc : cross) {
        if (cc < 1 || cc + 2 >= n) continue;                          // the reference indexes curCross-1 and curCross+2
        const double Ad1 = f.d[cc - 1], Av1 = f.v[cc - 1], Ad2 = f.d[cc], Av2 = f.v[cc];
        const double Bd1 = f.d[cc + 1], Bv1 = f.v[cc + 1], Bd2 = f.d[cc + 2], Bv2 = f.v[cc + 2];
        const double As = (Av2 - Av1) / (Ad2 - Ad1), Bs = (Bv2 - Bv1) / (Bd2 - Bd1);
        const double s1 = (Bv1 - Av1 + (Ad1 - Bd1) * Bs) / (As - Bs);
        const double Iv = Av1 + As * s1;
        f.d[cc] = d0[cc + 1]; f.v[cc] = Iv;
        f.d[cc + 1] = d0[cc]; f.v[cc + 1] = Iv;
    }
}

// SaturateSlopePWA.m:13-33
void saturate_slope_pwa(Pwa& f, double c_des) {
    auto pass = [&]() {
        for (size_t i = 1; i < f.d.size(); ++i) {
            const double c = (f.v[i] - f.v[i - 1]) / (f.d[i] - f.d[i - 1]);
            if (c > 0 && c > c_des) f.d[i] = f.d[i - 1] + (f.v[i] - f.v[i - 1]) / c_des;
            else if (c < 0 && c < -c_des) f.d[i - 1] = f.d[i] + (f.v[i] - f.v[i - 1]) / c_des;
        }
    };
    pass();
    fix_crossing_pwa(f);
    pass();
}

double matlab_mod(double a, double m) { return m == 0.0 ? a : a - std::floor(a / m) * m; }

}  // namespace

struct eepacc_nlp_tables {
    Vec s_vlim, v_vlim, s_curv, curvature, s_slope, slope, s_stop, v_stop, s_vinc, v_vinc, tl_s, tl_state;
};

extern "C" void eepacc_nlp_tables_free(eepacc_nlp_tables* t) { delete t; }

extern "C" int eepacc_nlp_problem_from_settings(eepacc_nlp_tables** owner, eepacc_nlp_problem* p, const eepacc_settings* S,
                                                const double W_NLP[7], const double b[21], double Ts, double t_sim) {
    if (!owner || !p || !S || !W_NLP || !b) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_problem_from_settings: null argument");
    *owner = nullptr;
    if (!(Ts > 0) || !(t_sim >= Ts)) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_problem_from_settings: Ts > 0 and t_sim >= Ts required");
    if (S->n_speedLim < 2 || S->n_curv < 2 || S->n_slope < 1 || !S->s_speedLim || !S->v_speedLim || !S->s_curv || !S->curvature ||
        !S->s_slope || !S->slope || S->n_stop < 0 || S->n_TL < 0 || (S->n_stop > 0 && !S->stopLoc) || (S->n_TL > 0 && !S->TLLoc))
        return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_problem_from_settings: route tables missing or too short");
    if (S->n_TL > EEPACC_NLP_MAX_TL) return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_problem_from_settings: at most 8 traffic lights");
    eepacc_nlp_tables* T = new (std::nothrow) eepacc_nlp_tables();
    if (!T) return eepacc::set_error(EEPACC_ENOMEM, "eepacc_nlp_problem_from_settings: out of memory");
    std::memset(p, 0, sizeof(*p));
    const int N = (int)std::llround(t_sim / Ts);                                        // :190
    p->N = N; p->Ts = Ts;
    std::memcpy(p->W, W_NLP, 7 * sizeof(double));
    std::memcpy(p->b, b, 21 * sizeof(double));
    p->s_goal = S->s_goal; p->h_min = S->h_min; p->tau_min = S->tau_min; p->alpha_TTL = S->alpha_TTL;
    T->s_slope.assign(S->s_slope, S->s_slope + S->n_slope); T->slope.assign(S->slope, S->slope + S->n_slope);
    if (S->n_slope == 1) {            // casadi.interpolant needs two knots; a one-knot table is that constant everywhere
        T->s_slope.push_back(T->s_slope[0] + 1.0); T->slope.push_back(T->slope[0]);
    }
    double ssum = 0.0;
    for (int i = 0; i < S->n_slope; ++i) ssum += S->slope[i];
    p->flat = ssum < 1e-1 ? 1 : 0;                                                      // :363
    T->s_vlim.assign(S->s_speedLim, S->s_speedLim + S->n_speedLim); T->v_vlim.assign(S->v_speedLim, S->v_speedLim + S->n_speedLim);
    T->s_curv.assign(S->s_curv, S->s_curv + S->n_curv); T->curvature.assign(S->curvature, S->curvature + S->n_curv);
    // stops :88-117
    const double incr = S->stopRefDist * S->stopRefVelSlope;                             // :92
    Vec locs(S->stopLoc, S->stopLoc + S->n_stop);
    std::sort(locs.begin(), locs.end());
    for (double loc : locs) {                                                            // :95-98
        T->s_stop.insert(T->s_stop.end(), {loc - S->stopRefDist, loc, loc + S->stopRefDist});
        T->v_stop.insert(T->v_stop.end(), {incr, S->stopVel, incr});
    }
    for (size_t i = 0; i + 1 < T->v_stop.size(); ++i) {                                  // :101-109
        if (T->s_stop[i + 1] <= T->s_stop[i]) {
            const double gap = T->s_stop[i] - T->s_stop[i + 1];
            const double corr = .5 * gap + T->s_stop[i + 1];
            const double val = incr / (1 + S->stopRefDist / gap);
            T->v_stop[i] = val; T->v_stop[i + 1] = val;
            T->s_stop[i] = corr - 1; T->s_stop[i + 1] = corr + 1;
        }
    }
    if (locs.empty()) { T->s_stop = {0.0, 1.0}; T->v_stop = {1e5, 1e5}; }              // :112-115
    // traffic lights :120-157
    p->n_tl = S->n_TL;
    p->tl_v[0] = incr; p->tl_v[1] = S->TLstopVel; p->tl_v[2] = incr;                     // :136
    for (int i = 0; i < S->n_TL; ++i) {
        const double* row = S->TLLoc + 4 * i;                                            // loc, phase, red, green
        T->tl_s.insert(T->tl_s.end(), {row[0] - S->stopRefDist, row[0], row[0] + S->stopRefDist});
        for (int j = 0; j < N; ++j)
            T->tl_state.push_back(matlab_mod(j * Ts - row[1], row[2] + row[3]) < row[2] ? .2 : 1e3);   // :146-150
    }
    // velocity incentive :160-184
    Pwa lim{T->s_vlim, T->v_vlim}, curve{T->s_curv, Vec()};
    for (double c : T->curvature) curve.v.push_back(S->alpha_TTL * std::pow(std::fabs(c), -1.0 / 3.0));
    Pwa inc = min_pwa(lim, curve);                                                       // :162
    if (inc.d.size() < 2) { delete T; return eepacc::set_error(EEPACC_EINVAL, "eepacc_nlp_problem_from_settings: degenerate speed-limit / curvature tables"); }
    saturate_slope_pwa(inc, 0.5);                                                        // :165
    {   // :168-172  pointsToKeep = ~diff(s)==0 parses as (~diff(s)) == 0: keep where diff ~= 0 (a mask one shorter than the vector)
        Pwa k;
        for (size_t i = 0; i + 1 < inc.d.size(); ++i)
            if (inc.d[i + 1] - inc.d[i] != 0) { k.d.push_back(inc.d[i]); k.v.push_back(inc.v[i]); }
        k.d.push_back(inc.d.back()); k.v.push_back(inc.v.back());
        inc = k;
    }
    inc = simplify_pwa(inc);                                                             // :175
    T->s_vinc = inc.d; T->v_vinc = inc.v;
    auto tab = [](const Vec& x, const Vec& y, int32_t& n, const double*& px, const double*& py) { n = (int32_t)x.size(); px = x.data(); py = y.data(); };
    tab(T->s_vlim, T->v_vlim, p->n_vlim, p->s_vlim, p->v_vlim);
    tab(T->s_curv, T->curvature, p->n_curv, p->s_curv, p->curvature);
    tab(T->s_slope, T->slope, p->n_slope, p->s_slope, p->slope);
    tab(T->s_stop, T->v_stop, p->n_stop, p->s_stop, p->v_stop);
    tab(T->s_vinc, T->v_vinc, p->n_vinc, p->s_vinc, p->v_vinc);
    p->tl_s = S->n_TL ? T->tl_s.data() : nullptr;
    p->tl_state = S->n_TL ? T->tl_state.data() : nullptr;
    *owner = T;
    return EEPACC_OK;
}
