// qf_em.cpp — see qf_em.hpp.
#include "qf_em.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <sstream>

namespace qf {

// ---------------------------------------------------------------------------- special functions
double digamma(double x) {  // recurrence up to x >= 6, then the asymptotic series
  double r = 0;
  while (x < 6) { r -= 1 / x; x += 1; }
  const double f = 1 / (x * x);
  return r + log(x) - 0.5 / x - f * (1. / 12 - f * (1. / 120 - f * (1. / 252 - f * (1. / 240 - f * (1. / 132)))));
}
double trigamma(double x) {
  double r = 0;
  while (x < 6) { r += 1 / (x * x); x += 1; }
  const double f = 1 / (x * x);
  return r + 1 / x + f / 2 + (f / x) * (1. / 6 - f * (1. / 30 - f * (1. / 42 - f * (1. / 30))));
}
double negbinom_pdf(unsigned k, double p, double n) {
  return exp(lgamma(k + n) - lgamma(n) - lgamma(k + 1.0) + n * log(p) + k * log1p(-p));
}

// ---------------------------------------------------------------------------- negative-binomial ML fit
namespace {
double nb_loglike(const std::vector<double>& kFreq, double p, double n) {  // src/negbinom.cpp:34-39
  double lp = 0;
  for (int k = 0; k < (int)kFreq.size(); ++k) lp += kFreq[k] * log_negbinom(k, p, n);
  return lp;
}
double opt_p(double n, const std::vector<double>& kFreq) {  // :78-86
  double fs = 0, ks = 0;
  for (size_t k = 0; k < kFreq.size(); ++k) { fs += kFreq[k]; ks += kFreq[k] * k; }
  return 1. / (1 + ks / (fs * n));
}
double d1(double n, const std::vector<double>& kFreq) {  // logNegativeBinomialSingleDeriv1, :45-58
  double fs = 0, ks = 0, kd = 0;
  for (size_t k = 0; k < kFreq.size(); ++k)
    if (kFreq[k] > 0) { fs += kFreq[k]; ks += kFreq[k] * k; kd += kFreq[k] * digamma(n + k); }
  return -fs * log(1. + ks / (fs * n)) - fs * digamma(n) + kd;
}
double d2(double n, const std::vector<double>& kFreq) {  // :60-71
  double fs = 0, kt = 0;
  for (size_t k = 0; k < kFreq.size(); ++k)
    if (kFreq[k] > 0) { fs += kFreq[k]; kt += kFreq[k] * trigamma(n + k); }
  return -fs * trigamma(n) + kt;
}
int sgn(double x) { return (x > 0) - (x < 0); }
}  // namespace

int fit_negbinom(const std::vector<double>& kFreq, double& pSuccess, double& nSuccess) {
  double count = 0, ksum = 0, ksq = 0;
  for (size_t k = 0; k < kFreq.size(); ++k) { count += kFreq[k]; ksum += kFreq[k] * k; ksq += kFreq[k] * k * k; }
  if (count <= 0) { pSuccess = nSuccess = std::nan(""); return 1; }
  const double mean = ksum / count, variance = ksq / count - mean * mean;
  double lo = 1., hi = std::max(1., kFreq.size() - 1.);
  if (variance > 0 && variance > mean) {  // momentFitNegativeBinomial, :142-162
    const double p0 = mean / variance, n0 = mean * p0 / (1 - p0);
    lo = std::max(1., n0 / 2);
    hi = std::min(kFreq.size() - 1., n0 * 2);
  }
  // bracketFitNegativeBinomial, :168-260: root of dLL/dn in [lo, hi] to the 1e-3 interval test; same-sign -> better endpoint
  double n;
  const double flo = d1(lo, kFreq), fhi = d1(hi, kFreq);
  if (sgn(flo) == sgn(fhi)) {
    n = nb_loglike(kFreq, opt_p(lo, kFreq), lo) > nb_loglike(kFreq, opt_p(hi, kFreq), hi) ? lo : hi;
  } else {
    double a = lo, b = hi, fa = flo;
    for (int it = 0; it < 100; ++it) {
      const double mid = 0.5 * (a + b), fm = d1(mid, kFreq);
      if (sgn(fm) == sgn(fa)) { a = mid; fa = fm; } else b = mid;
      if (fabs(b - a) < 1e-3 + 1e-3 * std::min(fabs(a), fabs(b))) break;
    }
    n = 0.5 * (a + b);
  }
  // gradientFitNegativeBinomial, :262-322: Newton on dLL/dn, relative step test 1e-4 (gsl_root_test_delta with epsabs 0),
  // runaway guard.  GSL's Newton step reports a zero derivative (GSL_EZERODIV) or a non-finite function value at the new
  // point (GSL_EBADFUNC) as an error, and the reference then keeps the last root it had accepted (it reads the solver's root
  // only after a successful iterate, :288-290): same here, with the status handed back (the reference's callers ignore it).
  int status = 0;
  double f = d1(n, kFreq), df = d2(n, kFreq);
  for (int it = 0; it < 100; ++it) {
    if (df == 0.0 || !std::isfinite(f) || !std::isfinite(df)) { status = 2; break; }
    const double nn = n - f / df;
    const double fn = d1(nn, kFreq), dfn = d2(nn, kFreq);
    if (!std::isfinite(nn) || !std::isfinite(fn) || !std::isfinite(dfn)) { status = 2; break; }
    const bool done = fabs(nn - n) < 1e-4 * fabs(nn) || nn == n;
    n = nn; f = fn; df = dfn;
    if (done) break;
    if (n > (double)kFreq.size()) { status = 3; break; }   // GSL_ERUNAWAY
  }
  nSuccess = n;
  pSuccess = opt_p(n, kFreq);
  return status;
}

// ---------------------------------------------------------------------------- counts
void ParamCounts::resize(unsigned ml, unsigned gl) {
  match_len = ml;
  gap_len = gl;
  v.assign(ne() + 4 * Kg() + 4, 0.0);
}

void ParamCounts::init_counts(double noBegin, double yesExtend, double matchIdent, double other, const NullParams* null) {
  for (int j = 0; j < 4; ++j)
    for (int k = 0; k < kNQual; ++k)
      ins(j)[k] = null ? other * null->null[j].p * 4 * negbinom_pdf(k, null->null[j].q, null->null[j].r) : other / kNQual;
  for (uint32_t i = 0; i < 4; ++i)
    for (uint32_t j = 0; j < Km(); ++j) {
      const uint32_t js = j & 3;
      for (int k = 0; k < kNQual; ++k) {
        // NB the identity test compares the base index with the FULL context k-mer index (src/qmodel.cpp:439-446)
        if (null)
          mat(i, j)[k] = (i == j ? matchIdent : (other * null->null[js].p * 4 / (1 - null->null[i].p))) *
                         negbinom_pdf(k, null->null[js].q, null->null[js].r);
        else
          mat(i, j)[k] = (i == j ? matchIdent : other) / kNQual;
      }
    }
  for (uint32_t g = 0; g < Kg(); ++g) {
    beginInsertNo(g) = noBegin; beginInsertYes(g) = other;
    beginDeleteNo(g) = noBegin; beginDeleteYes(g) = other;
  }
  extendInsertNo() = other; extendInsertYes() = yesExtend;
  extendDeleteNo() = other; extendDeleteYes() = yesExtend;
}

void ParamCounts::add_weighted(const ParamCounts& c, double w) {
  for (size_t a = 0; a < v.size(); ++a) v[a] += w * c.v[a];
}

Params ParamCounts::fit() const {
  Params qp;
  qp.match_len = match_len;
  qp.gap_len = gap_len;
  qp.resize();
  for (uint32_t g = 0; g < Kg(); ++g) {
    qp.beginDelete[g] = 1. / (1. + beginDeleteNo(g) / beginDeleteYes(g));
    qp.beginInsert[g] = 1. / (1. + beginInsertNo(g) / beginInsertYes(g));
  }
  qp.extendDelete = 1. / (1. + extendDeleteNo() / extendDeleteYes());
  qp.extendInsert = 1. / (1. + extendInsertNo() / extendInsertYes());
  double insFreq[4], insNorm = 0;
  for (int i = 0; i < 4; ++i) { insFreq[i] = std::accumulate(ins(i), ins(i) + kNQual, 0.); }
  for (int i = 0; i < 4; ++i) insNorm += insFreq[i];
  for (int i = 0; i < 4; ++i) {
    qp.insert[i].p = insFreq[i] / insNorm;
    fit_negbinom(std::vector<double>(ins(i), ins(i) + kNQual), qp.insert[i].q, qp.insert[i].r);
  }
  for (uint32_t i = 0; i < 4; ++i)
    for (uint32_t jp = 0; jp < Km(); jp += 4) {
      double f[4], norm = 0;
      for (uint32_t js = 0; js < 4; ++js) f[js] = std::accumulate(mat(i, jp + js), mat(i, jp + js) + kNQual, 0.);
      for (uint32_t js = 0; js < 4; ++js) norm += f[js];
      for (uint32_t js = 0; js < 4; ++js) {
        SymQualDist& d = qp.match[(size_t)i * Km() + jp + js];
        d.p = f[js] / norm;
        fit_negbinom(std::vector<double>(mat(i, jp + js), mat(i, jp + js) + kNQual), d.q, d.r);
      }
    }
  return qp;
}

namespace {
double log_beta_pdf(double x, double yes, double no) {  // logBetaPdf, src/qmodel.cpp:35-37 (gsl_ran_beta_pdf)
  const double a = yes + 1, b = no + 1;
  if (x < 0 || x > 1) return -INFINITY;
  const double gab = lgamma(a + b), ga = lgamma(a), gb = lgamma(b);
  double p;
  if (x == 0.0 || x == 1.0) p = exp(gab - ga - gb) * pow(x, a - 1) * pow(1 - x, b - 1);
  else p = exp(gab - ga - gb + log(x) * (a - 1) + log1p(-x) * (b - 1));
  return log(p);
}
double log_dirichlet4(const double* alpha, const double* theta) {  // log(gsl_ran_dirichlet_pdf(4, alpha, theta))
  double lp = 0, sum = 0;
  for (int i = 0; i < 4; ++i) { lp += (alpha[i] - 1) * log(theta[i]); sum += alpha[i]; }
  lp += lgamma(sum);
  for (int i = 0; i < 4; ++i) lp -= lgamma(alpha[i]);
  return log(exp(lp));
}
double log_qual_prob(const SymQualDist& d, const double* kFreq) {  // SymQualDist::logQualProb(kFreq), :83-85
  double lp = 0;
  for (int k = 0; k < kNQual; ++k) lp += kFreq[k] * log_negbinom(k, d.q, d.r);
  return lp;
}
}  // namespace

double ParamCounts::log_prior(const Params& qp) const {
  double lp = 0;
  for (uint32_t g = 0; g < Kg(); ++g) {
    lp += log_beta_pdf(qp.beginInsert[g], beginInsertYes(g), beginInsertNo(g));
    lp += log_beta_pdf(qp.beginDelete[g], beginDeleteYes(g), beginDeleteNo(g));
  }
  lp += log_beta_pdf(qp.extendInsert, extendInsertYes(), extendInsertNo());
  lp += log_beta_pdf(qp.extendDelete, extendDeleteYes(), extendDeleteNo());
  double alpha[4], theta[4];
  for (int i = 0; i < 4; ++i) {
    lp += log_qual_prob(qp.insert[i], ins(i));
    theta[i] = qp.insert[i].p;
    alpha[i] = std::accumulate(ins(i), ins(i) + kNQual, 1.);
  }
  lp += log_dirichlet4(alpha, theta);
  for (uint32_t i = 0; i < 4; ++i)
    for (uint32_t jp = 0; jp < Km(); jp += 4) {
      for (uint32_t js = 0; js < 4; ++js) {
        const SymQualDist& d = qp.match[(size_t)i * Km() + jp + js];
        lp += log_qual_prob(d, mat(i, jp + js));
        theta[js] = d.p;
        alpha[js] = std::accumulate(mat(i, jp + js), mat(i, jp + js) + kNQual, 1.);
      }
      lp += log_dirichlet4(alpha, theta);
    }
  return lp;
}

double ParamCounts::expected_log_like(const Params& qp) const {
  double ll = 0;
  for (uint32_t g = 0; g < Kg(); ++g) {
    ll += log(qp.beginInsert[g]) * beginInsertYes(g) + log(1 - qp.beginInsert[g]) * beginInsertNo(g);
    ll += log(qp.beginDelete[g]) * beginDeleteYes(g) + log(1 - qp.beginDelete[g]) * beginDeleteNo(g);
  }
  ll += log(qp.extendInsert) * extendInsertYes() + log(1 - qp.extendInsert) * extendInsertNo();
  ll += log(qp.extendDelete) * extendDeleteYes() + log(1 - qp.extendDelete) * extendDeleteNo();
  for (int i = 0; i < 4; ++i) {
    ll += log_qual_prob(qp.insert[i], ins(i));
    ll += log(qp.insert[i].p) * std::accumulate(ins(i), ins(i) + kNQual, 0.);
  }
  for (uint32_t i = 0; i < 4; ++i)
    for (uint32_t j = 0; j < Km(); ++j) {
      const SymQualDist& d = qp.match[(size_t)i * Km() + j];
      ll += log_qual_prob(d, mat(i, j));
      ll += log(d.p) * std::accumulate(mat(i, j), mat(i, j) + kNQual, 0.);
    }
  return ll;
}

static std::string join94(const double* v) {  // to_string_join: default-precision stream, src/util.h:96-106
  std::string s;
  for (int k = 0; k < kNQual; ++k) { if (k) s += ", "; s += fmt6(v[k]); }
  return s;
}

std::string ParamCounts::write_json() const {
  std::ostringstream o;
  o << "{\n";
  if (match_len != 1) o << "  \"matchOrder\": " << match_len << ",\n";
  if (gap_len != 0) o << "  \"gapOrder\": " << gap_len << ",\n";
  o << "  \"insert\": {\n";
  for (int i = 0; i < 4; ++i) o << "    \"" << "ACGT"[i] << "\": [ " << join94(ins(i)) << " ]" << (i == 3 ? " }," : ",") << "\n";
  o << "  \"match\": {\n";
  for (uint32_t jp = 0; jp < Km(); jp += 4) {
    o << "   \"" << kmer_to_string(jp, match_len).substr(0, match_len - 1) << "\": {\n";
    for (int i = 0; i < 4; ++i) {
      o << "    \"" << "ACGT"[i] << "\": {\n";
      for (uint32_t js = 0; js < 4; ++js)
        o << "      \"" << "ACGT"[js] << "\": [ " << join94(mat(i, jp + js)) << " ]" << (js == 3 ? " }" : ",\n");
      o << (i == 3 ? " }" : ",\n");
    }
    o << (jp == Km() - 4 ? " }" : ",") << "\n";
  }
  o << ",\n";
  auto kmers = [&](const char* name, size_t base) {
    o << "  \"" << name << "\": {";
    for (uint32_t g = 0; g < Kg(); ++g) o << (g == 0 ? "" : ",") << " \"" << kmer_to_string(g, gap_len) << "\": " << fmt6(v[base + g]);
    o << " }";
  };
  kmers("beginInsertNo", ne()); o << ",\n";
  kmers("beginInsertYes", ne() + Kg()); o << ",\n";
  kmers("beginDeleteNo", ne() + 2 * Kg()); o << ",\n";
  kmers("beginDeleteYes", ne() + 3 * Kg()); o << ",\n";
  o << "  \"extendInsertNo\": " << fmt6(extendInsertNo()) << ",\n";
  o << "  \"extendInsertYes\": " << fmt6(extendInsertYes()) << ",\n";
  o << "  \"extendDeleteNo\": " << fmt6(extendDeleteNo()) << ",\n";
  o << "  \"extendDeleteYes\": " << fmt6(extendDeleteYes()) << " }";
  return o.str();
}

bool ParamCounts::read_json(const Json& jm, std::string& err) {
  if (jm.type != Json::Object) { err = "JSON value is not an object"; return false; }
  const unsigned ml = jm.has("matchOrder", Json::Number) ? (unsigned)(int)jm.find("matchOrder")->num : 1;
  const unsigned gl = jm.has("gapOrder", Json::Number) ? (unsigned)(int)jm.find("gapOrder")->num : 0;
  if (ml < 1 || ml > 4 || gl > 4) { err = "unsupported matchOrder/gapOrder"; return false; }
  resize(ml, gl);
  const char* knames[4] = {"beginInsertNo", "beginInsertYes", "beginDeleteNo", "beginDeleteYes"};
  for (int a = 0; a < 4; ++a) {
    if (!jm.has(knames[a], Json::Object)) { err = std::string("Missing parameter: \"") + knames[a] + "\""; return false; }
    const Json& o = *jm.find(knames[a]);
    for (uint32_t g = 0; g < Kg(); ++g) {
      const std::string ks = kmer_to_string(g, gap_len);
      if (!o.has(ks, Json::Number)) { err = std::string("Missing parameter: \"") + knames[a] + "\".\"" + ks + "\""; return false; }
      v[ne() + a * Kg() + g] = o.find(ks)->num;
    }
  }
  const char* snames[4] = {"extendInsertNo", "extendInsertYes", "extendDeleteNo", "extendDeleteYes"};
  for (int a = 0; a < 4; ++a) {
    if (!jm.has(snames[a], Json::Number)) { err = std::string("Missing parameter: \"") + snames[a] + "\""; return false; }
    v[ne() + 4 * Kg() + a] = jm.find(snames[a])->num;
  }
  auto read_arr = [&](const Json& arr, double* out) {
    if (arr.type != Json::Array || arr.arr.size() != (size_t)kNQual) return false;
    for (int k = 0; k < kNQual; ++k) out[k] = arr.arr[k].num;
    return true;
  };
  if (!jm.has("insert", Json::Object)) { err = "Missing parameter: \"insert\""; return false; }
  for (int i = 0; i < 4; ++i) {
    const std::string k(1, "ACGT"[i]);
    const Json* e = jm.find("insert")->find(k);
    if (!e || !read_arr(*e, ins(i))) { err = "Couldn't read \"insert\".\"" + k + "\""; return false; }
  }
  if (!jm.has("match", Json::Object)) { err = "Missing parameter: \"match\""; return false; }
  for (uint32_t jp = 0; jp < Km(); jp += 4) {
    const std::string pref = kmer_to_string(jp, match_len).substr(0, match_len - 1);
    const Json* jj = jm.find("match")->find(pref);
    if (!jj) { err = "Missing parameter: \"match\".\"" + pref + "\""; return false; }
    for (int i = 0; i < 4; ++i) {
      const Json* ji = jj->find(std::string(1, "ACGT"[i]));
      if (!ji) { err = "Missing parameter in \"match\""; return false; }
      for (uint32_t js = 0; js < 4; ++js) {
        const Json* e = ji->find(std::string(1, "ACGT"[js]));
        if (!e || !read_arr(*e, mat(i, jp + js))) { err = "Couldn't read \"match\" counts"; return false; }
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------- null model
NullParams fit_null(const std::vector<std::string>& seqs, const std::vector<std::string>& quals, double pseudocount) {
  std::vector<std::vector<double>> cnt(4, std::vector<double>(kNQual, pseudocount / kNQual));
  double yes = pseudocount, no = pseudocount, sym[4] = {pseudocount, pseudocount, pseudocount, pseudocount};
  for (size_t n = 0; n < seqs.size(); ++n) {
    const std::string& s = seqs[n];
    ++no;
    yes += s.size();
    const bool hasQual = n < quals.size() && quals[n].size() == s.size();
    for (size_t i = 0; i < s.size(); ++i) {
      const int c = s[i] & ~0x20;
      const int t = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
      ++sym[t];
      if (hasQual) ++cnt[t][std::max(0, std::min(kNQual - 1, (int)(signed char)quals[n][i] - '!'))];
    }
  }
  NullParams np;
  np.nullEmit = 1 / (1 + no / yes);
  const double norm = sym[0] + sym[1] + sym[2] + sym[3];
  for (int t = 0; t < 4; ++t) {
    np.null[t].p = sym[t] / norm;
    fit_negbinom(cnt[t], np.null[t].q, np.null[t].r);
  }
  return np;
}

}  // namespace qf
