/* kmdiff_hip_test.h -- test hooks of libkmdiff_hip.so.  NOT part of the interface a host binds (kmdiff_hip.h):
 * they expose, on the host, the correctly rounded log / exp (kmdiff_amd/csrc/kmd_ddmath.h) and Cephes igamc(1/2, x)
 * over them, which decide the rows whose p-value lies within 1e-8 of the threshold (KMD_CNT_NEAR_THRESHOLD) -- so that
 * tests/test_rounded_math.py and tests/test_gpu_threshold.py can hold them to mpmath and to the reference's alglib. */
#ifndef KMDIFF_HIP_TEST_H
#define KMDIFF_HIP_TEST_H
#include "kmdiff_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
double kmd_test_log_rounded(double x);
double kmd_test_exp_rounded(double x);
double kmd_test_igamc_half_rounded(double x);
double kmd_test_row_pvalue_rounded(const kmd_model* m, uint64_t sum_control, uint64_t sum_case);
/* the reference's log-factorial running sum (log_factorial_table.cpp:13-22) of each d_k[i] < 2^20 as the device repeats
 * it: term by term (d_plain, may be NULL) and through the table of logarithms + the binade-wise exact integer sums of
 * kmd_pvalues_refine (d_fast).  Device pointers; asynchronous on `stream`. */
int kmd_test_running_sums(const uint64_t* d_k, size_t n, double* d_plain, double* d_fast, void* stream);
/* Stage 2's device linear algebra on inputs of the caller's (host pointers; synchronous) -- so that the vectors the
 * reference's own tests hold (tests/linear_test.cpp:29-31, 80-151: sigmoid(1), predict, the 4 x 4 LU and inverse) go
 * through the code K3 runs.  a: F x F row-major, b: F.  lane_out / group_out (2 F F + F + 1 doubles each; the lane
 * kernel's lu_solve, the group kernel's group_lu_solve): [LU in place: L below the diagonal, U on and above | the
 * inverse, row-major | w = a^-1 b | status: 0 ok, 1 det == 0, 2 det NaN (group: 1 for either)]. */
int kmd_test_popstrat_linear(int F, const double* a, const double* b, double* lane_out, double* group_out);
/* glm_irls (src/linear_model.cpp:297-410) on a design of the caller's -- X: n x f row-major, y: n (host pointers;
 * synchronous) -- through the loops K3 runs: the lane kernel's (irls_fit) and the group kernel's (k_popstrat_group, whose
 * k-mer column is count / total: it gets counts = the design's last column and totals of 1.0).  Each returns the weights
 * as glm_irls returns them (f doubles) and the iteration count it reports (:385).  2 <= f <= 13. */
int kmd_test_popstrat_irls(const double* X, const double* y, int n, int f, int max_iter, double* w_lane, int* iters_lane,
                           double* w_group, int* iters_group);
/* out[i] = sigmoid(x[i]) as K3 evaluates linear_model.cpp:191-195 */
int kmd_test_popstrat_sigmoid(const double* x, size_t n, double* out);
/* *eta_out = sum_j x[j] w[j] in index order, *p_out = its sigmoid: linear_predictor / predict (linear_model.cpp:197-211) */
int kmd_test_popstrat_predict(const double* w, const double* x, int n, double* eta_out, double* p_out);
#ifdef __cplusplus
}
#endif
#endif
