/*
 * eepacc_casadi_c.h -- B4 adaptor: the subset of CasADi's C API ("codegen-like API for
 * evaluating CasADi Functions", CAS/include/casadi/casadi_c.h:98-129) that the reference's Simulink S-function
 * ABO/casadi_fun.c consumes (casadi_fun.c:61-85 load, :83-119 dimensions, :170 eval, :178-187 memory), exported
 * by libeepacc.so under the same names.  Linking that S-function against libeepacc instead of libcasadi turns the
 * block into the EEPACC per-step operator (B2) on the GPU:
 *
 *   casadi_c_push_file(path)   path = a settings file written by eepacc_mpc_casadi_matlab_amd.casadi_c.write_config
 *                              (flat "key values..." text of the OPTsettings / vehicle fields of include/eepacc.h);
 *                              registers the functions "eepacc_ab_step" and "eepacc_fb_step"
 *   casadi_c_id(name)          -> 0 / 1, or -1
 *   inputs  (dense 1x1 each)   ab: s, v, a_prev, t0, s_tv, v_tv, a_tv_prev
 *                              fb: s, v, v_prev, a_prev, Fm_prev, Fb_prev, t0, s_tv, v_tv, a_tv_prev
 *   outputs (dense columns)    out [EEPACC_OUT_N], s_pred [N_hor+1], v_pred [N_hor+1], status [1]
 *   casadi_c_eval_id           returns 0 on success, non-zero on failure (casadi_fun.c:170-172); a QP that does not
 *                              converge is not a failure: status = 1 like the reference's exitMessage
 *   incref/decref thread-safe reference counting of the GPU handle, checkout/release not thread-safe (as documented
 *   at casadi_fun.c:177-187); casadi_int is long long (casadi_c.h:42).
 */
#ifndef EEPACC_CASADI_C_H
#define EEPACC_CASADI_C_H
#ifdef __cplusplus
extern "C" {
#endif
#ifndef casadi_int
#define casadi_int long long int
#endif
int casadi_c_push_file(const char* filename);
void casadi_c_pop(void);
void casadi_c_clear(void);
int casadi_c_n_loaded(void);
int casadi_c_id(const char* funname);
const char* casadi_c_name_id(int id);
int casadi_c_int_width(void);
int casadi_c_real_width(void);
void casadi_c_incref_id(int id);
void casadi_c_decref_id(int id);
int casadi_c_checkout_id(int id);
void casadi_c_release_id(int id, int mem);
casadi_int casadi_c_n_in_id(int id);
casadi_int casadi_c_n_out_id(int id);
const char* casadi_c_name_in_id(int id, casadi_int i);
const char* casadi_c_name_out_id(int id, casadi_int i);
const casadi_int* casadi_c_sparsity_in_id(int id, casadi_int i);
const casadi_int* casadi_c_sparsity_out_id(int id, casadi_int i);
int casadi_c_work_id(int id, casadi_int* sz_arg, casadi_int* sz_res, casadi_int* sz_iw, casadi_int* sz_w);
int casadi_c_eval_id(int id, const double** arg, double** res, casadi_int* iw, double* w, int mem);
#ifdef __cplusplus
}
#endif
#endif
