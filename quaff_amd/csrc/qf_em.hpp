// qf_em.hpp — host-side training arithmetic around the E-step kernels: expected-count containers, the M-step
// (maximum-likelihood negative-binomial fit), priors, the null-model fit and the counts JSON format.
// Off the hot path; pure C++17.  GSL is not available here, so digamma/trigamma, the bracketing
// root finder and the Newton polish are this repo's own.  Reference lines cited per function.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "qf_model.hpp"

namespace qf {

double digamma(double x);   // gsl_sf_psi
double trigamma(double x);  // gsl_sf_psi_1

// fitNegativeBinomial, src/negbinom.cpp:112-129.  kFreq[k] = weight of quality value k.  Returns 0 on success.
int fit_negbinom(const std::vector<double>& kFreq, double& pSuccess, double& nSuccess);
double negbinom_pdf(unsigned k, double p, double n);  // gsl_ran_negative_binomial_pdf

// QuaffParamCounts, src/qmodel.h:205-233, flattened exactly like qf_count_result.counts:
//   insert[4][94] | match[4][Km][94] | beginInsertNo[Kg] beginInsertYes[Kg] beginDeleteNo[Kg] beginDeleteYes[Kg]
//   | extendInsertNo extendInsertYes extendDeleteNo extendDeleteYes
struct ParamCounts {
  unsigned match_len = 1, gap_len = 0;
  std::vector<double> v;
  ParamCounts() { resize(1, 0); }
  ParamCounts(unsigned ml, unsigned gl) { resize(ml, gl); }
  void resize(unsigned ml, unsigned gl);
  uint32_t Km() const { return 1u << (2 * match_len); }
  uint32_t Kg() const { return 1u << (2 * gap_len); }
  size_t ne() const { return (size_t)(4 + 4 * Km()) * kNQual; }
  double* ins(int tok) { return &v[(size_t)tok * kNQual]; }
  const double* ins(int tok) const { return &v[(size_t)tok * kNQual]; }
  double* mat(int tok, uint32_t k) { return &v[((size_t)4 + (size_t)tok * Km() + k) * kNQual]; }
  const double* mat(int tok, uint32_t k) const { return &v[((size_t)4 + (size_t)tok * Km() + k) * kNQual]; }
  double& beginInsertNo(uint32_t g) { return v[ne() + g]; }
  double& beginInsertYes(uint32_t g) { return v[ne() + Kg() + g]; }
  double& beginDeleteNo(uint32_t g) { return v[ne() + 2 * Kg() + g]; }
  double& beginDeleteYes(uint32_t g) { return v[ne() + 3 * Kg() + g]; }
  double beginInsertNo(uint32_t g) const { return v[ne() + g]; }
  double beginInsertYes(uint32_t g) const { return v[ne() + Kg() + g]; }
  double beginDeleteNo(uint32_t g) const { return v[ne() + 2 * Kg() + g]; }
  double beginDeleteYes(uint32_t g) const { return v[ne() + 3 * Kg() + g]; }
  double& extendInsertNo() { return v[ne() + 4 * Kg()]; }
  double& extendInsertYes() { return v[ne() + 4 * Kg() + 1]; }
  double& extendDeleteNo() { return v[ne() + 4 * Kg() + 2]; }
  double& extendDeleteYes() { return v[ne() + 4 * Kg() + 3]; }
  double extendInsertNo() const { return v[ne() + 4 * Kg()]; }
  double extendInsertYes() const { return v[ne() + 4 * Kg() + 1]; }
  double extendDeleteNo() const { return v[ne() + 4 * Kg() + 2]; }
  double extendDeleteYes() const { return v[ne() + 4 * Kg() + 3]; }

  void init_counts(double noBegin, double yesExtend, double matchIdent, double other, const NullParams* null);  // :431-456
  void add_weighted(const ParamCounts& c, double w);  // src/qmodel.cpp:1656-1673
  Params fit() const;                                 // :1733-1768 (M-step)
  double log_prior(const Params& qp) const;           // :1681-1710
  double expected_log_like(const Params& qp) const;   // :1712-1731
  std::string write_json() const;                     // :458-470, :341-362
  bool read_json(const Json& j, std::string& err);    // :491-536
};

// The EM loop's stopping rule (QuaffTrainer::fitUnlimited, src/qmodel.cpp:2204-2206), tested after the E-step of iteration
// `iter` (0-based) and before its M-step: stop when iter > 0 and logLike + logPrior has not risen by the fraction min_inc of
// |previous|.
inline bool em_converged(int iter, double loglike_with_prior, double prev_loglike_with_prior, double min_inc) {
  return iter > 0 && loglike_with_prior < prev_loglike_with_prior + std::fabs(prev_loglike_with_prior) * min_inc;
}

// QuaffNullParams(seqs, pseudocount), src/qmodel.cpp:1811-1843
NullParams fit_null(const std::vector<std::string>& seqs, const std::vector<std::string>& quals, double pseudocount = 1);

}  // namespace qf
