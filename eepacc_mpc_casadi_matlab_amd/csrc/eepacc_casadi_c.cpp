// eepacc_casadi_c.cpp -- B4 adaptor: CasADi's C evaluation API (the subset ABO/casadi_fun.c uses) in front of the
// per-step operators eepacc_ab_step / eepacc_fb_step.  See include/eepacc_casadi_c.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <sstream>
#include <fstream>
#include <string>
#include <vector>
#include "../../include/eepacc.h"
#include "../../include/eepacc_casadi_c.h"

namespace {

struct Config {                       // one pushed settings file
    std::map<std::string, std::vector<double>> kv;
    eepacc_settings S;
    eepacc_vehicle V;
    std::vector<int32_t> Mb;
    bool ok = false;
};

struct Fun {
    std::string name;
    bool fb;
    int cfg;                          // index into g_cfgs
    std::vector<std::string> in_names, out_names;
    std::vector<std::vector<casadi_int>> sp_in, sp_out;
    eepacc_handle* h = nullptr;
    int refs = 0;
    double* d_io = nullptr;           // device scratch: inputs | out | s_pred | v_pred | status
};

std::vector<Config*> g_cfgs;
std::vector<Fun*> g_funs;
std::vector<int> g_batches;           // functions added per push (for pop)
std::mutex g_mu;

std::vector<casadi_int> dense_col(int n) {      // CCS of a dense n x 1 column: [nrow, ncol, colind(2), row(n)]
    std::vector<casadi_int> sp = {n, 1, 0, n};
    for (int i = 0; i < n; ++i) sp.push_back(i);
    return sp;
}

const std::vector<double>* get(const Config& c, const char* k) {
    auto it = c.kv.find(k);
    return it == c.kv.end() ? nullptr : &it->second;
}
bool scalar(const Config& c, const char* k, double& out) {
    const auto* v = get(c, k);
    if (!v || v->empty()) return false;
    out = (*v)[0];
    return true;
}

bool parse(const char* path, Config& c) {
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        std::vector<double> vals;
        std::string tok;
        while (ss >> tok) {
            if (tok == "inf" || tok == "Inf") vals.push_back(1e308 * 10);
            else if (tok == "-inf" || tok == "-Inf") vals.push_back(-1e308 * 10);
            else vals.push_back(strtod(tok.c_str(), nullptr));
        }
        c.kv[key] = vals;
    }
    memset(&c.S, 0, sizeof c.S);
    memset(&c.V, 0, sizeof c.V);
    eepacc_settings& S = c.S;
    double x;
#define REQ(key, dst) do { if (!scalar(c, key, x)) return false; dst = x; } while (0)
#define REQI(key, dst) do { if (!scalar(c, key, x)) return false; dst = (int32_t)x; } while (0)
#define ARR(key, ptr, cnt) do { const auto* v_ = get(c, key); if (v_ && !v_->empty()) { ptr = v_->data(); cnt = (int32_t)v_->size(); } else { ptr = nullptr; cnt = 0; } } while (0)
    REQI("N_hor", S.N_hor);
    int32_t n = 0;
    ARR("Tvec", S.Tvec, n);
    if (n != S.N_hor) return false;
    if (const auto* mb = get(c, "Mb")) { for (double m : *mb) c.Mb.push_back(m != 0.0); }
    c.Mb.resize(S.N_hor, 0);
    S.Mb = c.Mb.data();
    const auto* wab = get(c, "W_AB"); const auto* wfb = get(c, "W_FB");
    if (!wab || wab->size() != 7 || !wfb || wfb->size() != 7) return false;
    for (int i = 0; i < 7; ++i) { S.W_AB[i] = (*wab)[i]; S.W_FB[i] = (*wfb)[i]; }
    REQI("ab_fuel_term", S.ab_fuel_term); REQI("ab_route_rows", S.ab_route_rows);
    REQ("tau_min", S.tau_min); REQ("h_min", S.h_min); REQ("s_goal", S.s_goal);
    REQI("paramEstSetting", S.paramEstSetting); REQI("TVestSetting", S.TVestSetting);
    REQ("tConstACC_ego", S.tConstACC_ego); REQ("tConstACC_tar", S.tConstACC_tar);
    REQI("N_integratePlant", S.N_integratePlant); REQI("solverToUse", S.solverToUse); REQI("FBuseTaylor", S.FBuseTaylor);
    const auto* bq = get(c, "b_quadr"); const auto* b5 = get(c, "b_fifthOrder");
    if (!bq || bq->size() != 6 || !b5 || b5->size() != 21) return false;
    for (int i = 0; i < 6; ++i) S.b_quadr[i] = (*bq)[i];
    for (int i = 0; i < 21; ++i) S.b_fifthOrder[i] = (*b5)[i];
    // paired tables must have equal lengths (the MEX reader checks the same, mex/eepacc_mex_common.h)
    ARR("s_speedLim", S.s_speedLim, S.n_speedLim); ARR("v_speedLim", S.v_speedLim, n);
    if (n != S.n_speedLim || n < 1 || !S.s_speedLim) return false;
    ARR("s_curv", S.s_curv, S.n_curv); ARR("curvature", S.curvature, n);
    if (n != S.n_curv || n < 1 || !S.s_curv) return false;
    ARR("s_slope", S.s_slope, S.n_slope); ARR("slope", S.slope, n);
    if (n != S.n_slope || n < 1 || !S.s_slope) return false;
    ARR("stopLoc", S.stopLoc, S.n_stop);
    ARR("TLLoc", S.TLLoc, n);
    if (n % 4 != 0) return false;
    S.n_TL = n / 4;
    REQ("stopRefDist", S.stopRefDist); REQ("stopRefVelSlope", S.stopRefVelSlope); REQ("stopVel", S.stopVel);
    REQ("TLstopVel", S.TLstopVel); REQ("TLStopRegionSize", S.TLStopRegionSize); REQ("alpha_TTL", S.alpha_TTL);
    eepacc_vehicle& V = c.V;
#define VEH(field) do { if (scalar(c, "vehicle." #field, x)) V.field = x; } while (0)
    VEH(m); VEH(A_f); VEH(c_d); VEH(L); VEH(h_g); VEH(WD_s_F); VEH(L_f); VEH(L_r); VEH(F0); VEH(F1); VEH(F2);
    VEH(p00); VEH(p10); VEH(p01); VEH(P_m_max); VEH(T_m_max); VEH(omega_m_r); VEH(omega_m_max); VEH(c_r); VEH(R_w);
    VEH(beta_gb); VEH(beta_fd); VEH(phi); VEH(v_max); VEH(eta_TF); VEH(lambda); VEH(mu); VEH(rho_a); VEH(g); VEH(zeta_a);
    V.tau_fd = 1.0; V.eta_drive = 1.0;
    for (int i = 0; i < 7; ++i) V.upSpd[i] = 1e9;
    for (int i = 0; i < 8; ++i) V.tau_gb[i] = 1.0;
    VEH(k00); VEH(k10); VEH(k01); VEH(tau_fd); VEH(eta_drive);
    { const auto* up = get(c, "vehicle.upSpd"); const auto* gb = get(c, "vehicle.tau_gb");
      if (up && gb) {
          if (up->size() != 7 || gb->size() != 8) return false;
          for (int i = 0; i < 7; ++i) V.upSpd[i] = (*up)[i];
          for (int i = 0; i < 8; ++i) V.tau_gb[i] = (*gb)[i];
      } }
#undef VEH
#undef REQ
#undef REQI
#undef ARR
    if (!(V.m > 0.0) || !(V.lambda > 0.0) || !(V.phi > 0.0)) return false;
    c.ok = true;
    return true;
}

Fun* fun(int id) { return (id >= 0 && id < (int)g_funs.size()) ? g_funs[id] : nullptr; }

void release_handle(Fun* f) {
    if (f->h) { eepacc_destroy(f->h); f->h = nullptr; }
    if (f->d_io) { (void)hipFree(f->d_io); f->d_io = nullptr; }
}

}  // namespace

extern "C" int casadi_c_push_file(const char* filename) {
    std::lock_guard<std::mutex> lk(g_mu);
    Config* c = new Config();
    if (!filename || !parse(filename, *c)) { delete c; return 1; }
    g_cfgs.push_back(c);
    const int N = c->S.N_hor;
    for (int fb = 0; fb < 2; ++fb) {
        Fun* f = new Fun();
        f->name = fb ? "eepacc_fb_step" : "eepacc_ab_step";
        f->fb = fb != 0; f->cfg = (int)g_cfgs.size() - 1;
        f->in_names = fb ? std::vector<std::string>{"s", "v", "v_prev", "a_prev", "Fm_prev", "Fb_prev", "t0", "s_tv", "v_tv", "a_tv_prev"}
                         : std::vector<std::string>{"s", "v", "a_prev", "t0", "s_tv", "v_tv", "a_tv_prev"};
        f->out_names = {"out", "s_pred", "v_pred", "status"};
        for (size_t i = 0; i < f->in_names.size(); ++i) f->sp_in.push_back(dense_col(1));
        f->sp_out = {dense_col(EEPACC_OUT_N), dense_col(N + 1), dense_col(N + 1), dense_col(1)};
        g_funs.push_back(f);
    }
    g_batches.push_back(2);
    return 0;
}

extern "C" void casadi_c_pop(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_batches.empty()) return;
    for (int i = 0; i < g_batches.back(); ++i) { release_handle(g_funs.back()); delete g_funs.back(); g_funs.pop_back(); }
    g_batches.pop_back();
    delete g_cfgs.back(); g_cfgs.pop_back();
}
extern "C" void casadi_c_clear(void) { while (casadi_c_n_loaded() > 0) casadi_c_pop(); }
extern "C" int casadi_c_n_loaded(void) { return (int)g_funs.size(); }
extern "C" int casadi_c_id(const char* funname) {
    if (!funname) return -1;
    for (int i = (int)g_funs.size() - 1; i >= 0; --i) if (g_funs[i]->name == funname) return i;
    return -1;
}
extern "C" const char* casadi_c_name_id(int id) { Fun* f = fun(id); return f ? f->name.c_str() : ""; }
extern "C" int casadi_c_int_width(void) { return (int)sizeof(casadi_int); }
extern "C" int casadi_c_real_width(void) { return (int)sizeof(double); }

extern "C" void casadi_c_incref_id(int id) {
    std::lock_guard<std::mutex> lk(g_mu);
    Fun* f = fun(id);
    if (!f) return;
    if (f->refs++ == 0 && !f->h) {
        Config* c = g_cfgs[f->cfg];
        if (eepacc_create(&f->h, &c->S, &c->V, 0, 1) != EEPACC_OK) { f->h = nullptr; return; }
        const size_t n = 16 + EEPACC_OUT_N + 2 * (size_t)(c->S.N_hor + 1) + 2;
        if (hipMalloc(&f->d_io, n * sizeof(double)) != hipSuccess) { f->d_io = nullptr; release_handle(f); }
    }
}
extern "C" void casadi_c_decref_id(int id) {
    std::lock_guard<std::mutex> lk(g_mu);
    Fun* f = fun(id);
    if (f && f->refs > 0 && --f->refs == 0) release_handle(f);
}
extern "C" int casadi_c_checkout_id(int id) { return fun(id) ? 0 : -1; }
extern "C" void casadi_c_release_id(int id, int mem) { (void)id; (void)mem; }
extern "C" casadi_int casadi_c_n_in_id(int id) { Fun* f = fun(id); return f ? (casadi_int)f->in_names.size() : 0; }
extern "C" casadi_int casadi_c_n_out_id(int id) { Fun* f = fun(id); return f ? (casadi_int)f->out_names.size() : 0; }
extern "C" const char* casadi_c_name_in_id(int id, casadi_int i) {
    Fun* f = fun(id); return (f && i >= 0 && i < (casadi_int)f->in_names.size()) ? f->in_names[i].c_str() : nullptr;
}
extern "C" const char* casadi_c_name_out_id(int id, casadi_int i) {
    Fun* f = fun(id); return (f && i >= 0 && i < (casadi_int)f->out_names.size()) ? f->out_names[i].c_str() : nullptr;
}
extern "C" const casadi_int* casadi_c_sparsity_in_id(int id, casadi_int i) {
    Fun* f = fun(id); return (f && i >= 0 && i < (casadi_int)f->sp_in.size()) ? f->sp_in[i].data() : nullptr;
}
extern "C" const casadi_int* casadi_c_sparsity_out_id(int id, casadi_int i) {
    Fun* f = fun(id); return (f && i >= 0 && i < (casadi_int)f->sp_out.size()) ? f->sp_out[i].data() : nullptr;
}
extern "C" int casadi_c_work_id(int id, casadi_int* sz_arg, casadi_int* sz_res, casadi_int* sz_iw, casadi_int* sz_w) {
    Fun* f = fun(id);
    if (!f) return 1;
    if (sz_arg) *sz_arg = (casadi_int)f->in_names.size();
    if (sz_res) *sz_res = (casadi_int)f->out_names.size();
    if (sz_iw) *sz_iw = 0;
    if (sz_w) *sz_w = 0;
    return 0;
}

extern "C" int casadi_c_eval_id(int id, const double** arg, double** res, casadi_int* iw, double* w, int mem) {
    (void)iw; (void)w; (void)mem;
    Fun* f = fun(id);
    if (!f || !f->h || !f->d_io || !arg || !res) return 1;
    const int N = g_cfgs[f->cfg]->S.N_hor, n_in = (int)f->in_names.size();
    double host[16];
    for (int i = 0; i < n_in; ++i) host[i] = arg[i] ? arg[i][0] : 0.0;      // a NULL argument is zero (CasADi convention)
    double* d = f->d_io;
    if (hipMemcpy(d, host, sizeof(double) * n_in, hipMemcpyHostToDevice) != hipSuccess) return 1;
    double *d_out = d + 16, *d_sp = d_out + EEPACC_OUT_N, *d_vp = d_sp + (N + 1);
    int32_t* d_st = reinterpret_cast<int32_t*>(d_vp + (N + 1));
    int rc;
    if (f->fb) rc = eepacc_fb_step(f->h, 1, d + 0, d + 1, d + 2, d + 3, d + 4, d + 5, d + 6, d + 7, d + 8, d + 9, d_out, d_sp, d_vp, d_st, nullptr);
    else rc = eepacc_ab_step(f->h, 1, d + 0, d + 1, d + 2, d + 3, d + 4, d + 5, d + 6, d_out, d_sp, d_vp, d_st, nullptr);
    if (rc != EEPACC_OK || eepacc_synchronize(f->h, nullptr) != EEPACC_OK) return 1;
    int32_t st = 0;
    if (res[0] && hipMemcpy(res[0], d_out, sizeof(double) * EEPACC_OUT_N, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (res[1] && hipMemcpy(res[1], d_sp, sizeof(double) * (N + 1), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (res[2] && hipMemcpy(res[2], d_vp, sizeof(double) * (N + 1), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (hipMemcpy(&st, d_st, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (res[3]) res[3][0] = (double)st;
    return 0;
}
