// qf_model.cpp — host-side quaff model (see qf_model.hpp).  No HIP in this file.
#include "qf_model.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <sstream>

namespace qf {

// ------------------------------------------------------------------------------ JSON
const Json* Json::find(const std::string& key) const {
  for (const auto& kv : obj)
    if (kv.first == key) return &kv.second;
  return nullptr;
}

// Decimal -> double exactly as the reference's JSON library does it (src/gason.cpp:73-117):
// digit-by-digit accumulation, fraction digits scaled by a running 0.1 product, exponent by
// repeated squaring.  Not correctly rounded; every model number inherits its rounding.
double gason_number(const char* s, const char** end) {
  const char first = *s;
  if (first == '-') ++s;
  double result = 0;
  while (*s >= '0' && *s <= '9') result = (result * 10) + (*s++ - '0');
  if (*s == '.') {
    ++s;
    double fraction = 1;
    while (*s >= '0' && *s <= '9') {
      fraction *= 0.1;
      result += (*s++ - '0') * fraction;
    }
  }
  if (*s == 'e' || *s == 'E') {
    ++s;
    double base = 10;
    if (*s == '+')
      ++s;
    else if (*s == '-') {
      ++s;
      base = 0.1;
    }
    unsigned int exponent = 0;
    while (*s >= '0' && *s <= '9') exponent = (exponent * 10) + (unsigned)(*s++ - '0');
    double power = 1;
    for (; exponent; exponent >>= 1, base *= base)
      if (exponent & 1) power *= base;
    result *= power;
  }
  if (end) *end = s;
  return first == '-' ? -result : result;
}

namespace {
struct Parser {
  const char* p;
  const char* e;
  std::string err;
  void ws() {
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
  }
  bool fail(const char* m) {
    if (err.empty()) err = m;
    return false;
  }
  bool str(std::string& out) {
    if (p >= e || *p != '"') return fail("expected string");
    ++p;
    out.clear();
    while (p < e && *p != '"') {
      if (*p == '\\' && p + 1 < e) {
        ++p;
        switch (*p) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {
            unsigned v = 0;
            for (int a = 0; a < 4 && p + 1 < e; ++a) {
              ++p;
              char c = *p;
              v = v * 16 + (c <= '9' ? c - '0' : (c & ~' ') - 'A' + 10);
            }
            if (v < 0x80)
              out += (char)v;
            else if (v < 0x800) {
              out += (char)(0xC0 | (v >> 6));
              out += (char)(0x80 | (v & 0x3F));
            } else {
              out += (char)(0xE0 | (v >> 12));
              out += (char)(0x80 | ((v >> 6) & 0x3F));
              out += (char)(0x80 | (v & 0x3F));
            }
            break;
          }
          default: out += *p;
        }
        ++p;
      } else
        out += *p++;
    }
    if (p >= e) return fail("unterminated string");
    ++p;
    return true;
  }
  bool value(Json& j, int depth) {
    if (depth > 64) return fail("nesting too deep");
    ws();
    if (p >= e) return fail("unexpected end");
    const char c = *p;
    if (c == '{') {
      ++p;
      j.type = Json::Object;
      ws();
      if (p < e && *p == '}') { ++p; return true; }
      while (true) {
        ws();
        std::string k;
        if (!str(k)) return false;
        ws();
        if (p >= e || *p != ':') return fail("expected ':'");
        ++p;
        Json v;
        if (!value(v, depth + 1)) return false;
        j.obj.emplace_back(std::move(k), std::move(v));
        ws();
        if (p < e && *p == ',') { ++p; continue; }
        if (p < e && *p == '}') { ++p; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      ++p;
      j.type = Json::Array;
      ws();
      if (p < e && *p == ']') { ++p; return true; }
      while (true) {
        Json v;
        if (!value(v, depth + 1)) return false;
        j.arr.push_back(std::move(v));
        ws();
        if (p < e && *p == ',') { ++p; continue; }
        if (p < e && *p == ']') { ++p; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') {
      j.type = Json::String;
      return str(j.str);
    }
    if (c == '-' || (c >= '0' && c <= '9')) {
      j.type = Json::Number;
      const char* end = nullptr;
      j.num = gason_number(p, &end);
      p = end;
      return true;
    }
    if (e - p >= 4 && !strncmp(p, "true", 4)) { j.type = Json::Bool; j.b = true; p += 4; return true; }
    if (e - p >= 5 && !strncmp(p, "false", 5)) { j.type = Json::Bool; j.b = false; p += 5; return true; }
    if (e - p >= 4 && !strncmp(p, "null", 4)) { j.type = Json::Null; p += 4; return true; }
    return fail("unexpected character");
  }
};
}  // namespace

bool parse_json(const std::string& text, Json& out, std::string& err) {
  // the buffer is NUL-terminated (std::string), which gason_number relies on
  Parser ps{text.c_str(), text.c_str() + text.size(), {}};
  out = Json();
  if (!ps.value(out, 0)) {
    err = "JSON: " + ps.err;
    return false;
  }
  return true;
}

std::string fmt6(double x) {
  char buf[64];
  snprintf(buf, sizeof buf, "%g", x);
  return buf;
}

// ------------------------------------------------------------------------ parameters
std::string kmer_to_string(uint64_t kmer, unsigned k) {
  std::string s(k, 'A');
  for (unsigned j = 0; j < k; ++j, kmer /= 4) s[k - 1 - j] = "ACGT"[kmer % 4];
  return s;
}

void Params::resize() {
  match.assign((size_t)4 * Km(), SymQualDist());
  beginInsert.assign(Kg(), .5);
  beginDelete.assign(Kg(), .5);
}

static bool read_sqd(const Json& j, SymQualDist& d) {  // SymQualDist::readJson, src/qmodel.cpp:68-77
  if (j.type != Json::Object || !j.has("p", Json::Number) || !j.has("q", Json::Number) || !j.has("r", Json::Number))
    return false;
  d.p = j.find("p")->num;
  d.q = j.find("q")->num;
  d.r = j.find("r")->num;
  return true;
}

bool Params::read_json(const Json& jm, std::string& err) {
  if (jm.type != Json::Object) { err = "JSON value is not an object"; return false; }
  // readJsonKmerLen, src/qmodel.cpp:122-128
  match_len = jm.has("matchOrder", Json::Number) ? (unsigned)(int)jm.find("matchOrder")->num : 1;
  gap_len = jm.has("gapOrder", Json::Number) ? (unsigned)(int)jm.find("gapOrder")->num : 0;
  if (match_len < 1 || match_len > 4 || gap_len > 4) { err = "unsupported matchOrder/gapOrder (need 1..4 / 0..4)"; return false; }
  resize();
  // refBase is written by the reference but never read back (src/qmodel.cpp:236-271): stays .25
  for (const char* key : {"beginInsert", "beginDelete"}) {
    if (!jm.has(key, Json::Object)) { err = std::string("Missing parameter: \"") + key + "\""; return false; }
    const Json& o = *jm.find(key);
    for (uint32_t g = 0; g < Kg(); ++g) {
      const std::string ks = kmer_to_string(g, gap_len);
      if (!o.has(ks, Json::Number)) { err = std::string("Missing parameter: \"") + key + "\".\"" + ks + "\""; return false; }
      (key[5] == 'I' ? beginInsert : beginDelete)[g] = o.find(ks)->num;
    }
  }
  if (!jm.has("extendInsert", Json::Number)) { err = "Missing parameter: \"extendInsert\""; return false; }
  if (!jm.has("extendDelete", Json::Number)) { err = "Missing parameter: \"extendDelete\""; return false; }
  extendInsert = jm.find("extendInsert")->num;
  extendDelete = jm.find("extendDelete")->num;
  if (!jm.has("insert", Json::Object)) { err = "Missing parameter: \"insert\""; return false; }
  const Json& ji = *jm.find("insert");
  for (int i = 0; i < 4; ++i) {
    const std::string k(1, "ACGT"[i]);
    if (!ji.has(k, Json::Object) || !read_sqd(*ji.find(k), insert[i])) { err = "Missing parameter: \"insert\".\"" + k + "\""; return false; }
  }
  if (!jm.has("match", Json::Object)) { err = "Missing parameter: \"match\""; return false; }
  const Json& jmat = *jm.find("match");
  for (uint32_t jp = 0; jp < Km(); jp += 4) {
    const std::string pref = kmer_to_string(jp, match_len).substr(0, match_len - 1);
    if (!jmat.has(pref, Json::Object)) { err = "Missing parameter: \"match\".\"" + pref + "\""; return false; }
    const Json& jj = *jmat.find(pref);
    for (int i = 0; i < 4; ++i) {
      const std::string ik(1, "ACGT"[i]);
      if (!jj.has(ik, Json::Object)) { err = "Missing parameter: \"match\".\"" + pref + "\".\"" + ik + "\""; return false; }
      const Json& jji = *jj.find(ik);
      for (int js = 0; js < 4; ++js) {
        const std::string sk(1, "ACGT"[js]);
        if (!jji.has(sk, Json::Object) || !read_sqd(*jji.find(sk), match[(size_t)i * Km() + jp + js])) {
          err = "Missing parameter: \"match\".\"" + pref + "\".\"" + ik + "\".\"" + sk + "\"";
          return false;
        }
      }
    }
  }
  return true;
}

static std::string sqd_json(const SymQualDist& d) {  // SymQualDist::writeJson, src/qmodel.cpp:58-66
  const double m = d.r * (1 - d.q) / d.q;                   // negativeBinomialMean, src/negbinom.cpp:104-106
  const double sd = std::sqrt(d.r * (1 - d.q) / (d.q * d.q));  // sqrt(negativeBinomialVariance) :108-110
  return "{ \"p\": " + fmt6(d.p) + ", \"q\": " + fmt6(d.q) + ", \"r\": " + fmt6(d.r) + ", \"m\": " + fmt6(m) +
         ", \"sd\": " + fmt6(sd) + " }";
}

std::string Params::write_json() const {  // QuaffParams::writeJson, src/qmodel.cpp:187-218
  std::ostringstream o;
  o << "{\n";
  if (match_len != 1) o << "  \"matchOrder\": " << match_len << ",\n";
  if (gap_len != 0) o << "  \"gapOrder\": " << gap_len << ",\n";
  o << "  \"refBase\": {";
  for (int i = 0; i < 4; ++i) o << " \"" << "ACGT"[i] << "\": " << fmt6(refBase[i]) << (i == 3 ? " },\n" : ",");
  auto kmers = [&](const char* name, const std::vector<double>& v) {
    o << "  \"" << name << "\": {";
    for (uint32_t g = 0; g < Kg(); ++g) o << (g == 0 ? "" : ",") << " \"" << kmer_to_string(g, gap_len) << "\": " << fmt6(v[g]);
    o << " }";
  };
  kmers("beginInsert", beginInsert);
  o << ",\n";
  kmers("beginDelete", beginDelete);
  o << ",\n";
  o << "  \"extendInsert\": " << fmt6(extendInsert) << ",\n";
  o << "  \"extendDelete\": " << fmt6(extendDelete) << ",\n";
  o << "  \"insert\": {\n";
  for (int i = 0; i < 4; ++i) o << "    \"" << "ACGT"[i] << "\": " << sqd_json(insert[i]) << (i == 3 ? " }," : ",") << "\n";
  o << "  \"match\": {\n";
  for (uint32_t jp = 0; jp < Km(); jp += 4) {
    o << "   \"" << kmer_to_string(jp, match_len).substr(0, match_len - 1) << "\": {\n";
    for (int i = 0; i < 4; ++i) {
      o << "    \"" << "ACGT"[i] << "\": {\n";
      for (int js = 0; js < 4; ++js)
        o << "      \"" << "ACGT"[js] << "\": " << sqd_json(match[(size_t)i * Km() + jp + js]) << (js == 3 ? " }" : ",\n");
      o << (i == 3 ? " }" : ",\n");
    }
    o << (jp == Km() - 4 ? " }" : ",\n");
  }
  o << " }";
  return o.str();
}

const char* const kDefaultParamsJson =
#include "default_params.inc"
    ;

// log(NB pdf), src/negbinom.cpp:30-32.  GSL (absent here, un-pinned upstream) defines the pdf as
// exp(lngamma(k+n) - lngamma(n) - lngamma(k+1) + n log p + k log1p(-p)); evaluated with libm lgamma,
// keeping the exp/log round trip of the reference.
double log_negbinom(int k, double p, double n) {
  const double f = lgamma(k + n), a = lgamma(n), b = lgamma(k + 1.0);
  const double P = exp(f - a - b + n * log(p) + k * log1p(-p));
  return log(P);
}

static void sym_qual_scores(const SymQualDist& d, double* out) {  // SymQualScores ctor, src/qmodel.cpp:87-93
  const double lsp = log(d.p);
  for (int k = 0; k < kNQual; ++k) out[k] = lsp + log_negbinom(k, d.q, d.r);
  out[kNQual] = lsp;
}

void Scores::build(const Params& p) {  // QuaffScores ctor, src/qmodel.cpp:296-325
  match_len = p.match_len;
  gap_len = p.gap_len;
  Km = p.Km();
  Kg = p.Kg();
  ins.assign((size_t)4 * kNQ1, 0);
  mat.assign((size_t)4 * Km * kNQ1, 0);
  trans.assign((size_t)4 * Kg + 4, 0);
  for (int i = 0; i < 4; ++i) {
    sym_qual_scores(p.insert[i], &ins[(size_t)i * kNQ1]);
    for (uint32_t j = 0; j < Km; ++j) sym_qual_scores(p.match[(size_t)i * Km + j], &mat[((size_t)i * Km + j) * kNQ1]);
  }
  for (uint32_t j = 0; j < Kg; ++j) {
    trans[j] = log(1 - p.beginInsert[j]) + log(1 - p.beginDelete[j]);  // m2m
    trans[Kg + j] = log(p.beginInsert[j]);                             // m2i
    trans[2 * Kg + j] = log(1 - p.beginInsert[j]) + log(p.beginDelete[j]);  // m2d
    trans[3 * Kg + j] = log(p.beginInsert[j]);                         // m2e (sic: src/qmodel.cpp:317)
  }
  trans[4 * Kg + 0] = log(p.extendDelete);      // d2d
  trans[4 * Kg + 1] = log(1 - p.extendDelete);  // d2m
  trans[4 * Kg + 2] = log(p.extendInsert);      // i2i
  trans[4 * Kg + 3] = log(1 - p.extendInsert);  // i2m
}

// ------------------------------------------------------------------------ null model
bool NullParams::read_json(const Json& jm, std::string& err) {  // src/qmodel.cpp:1856-1866
  if (jm.type != Json::Object) { err = "JSON value is not an object"; return false; }
  if (!jm.has("nullEmit", Json::Number)) { err = "Missing parameter: \"nullEmit\""; return false; }
  nullEmit = jm.find("nullEmit")->num;
  if (!jm.has("null", Json::Object)) { err = "Missing parameter: \"null\""; return false; }
  const Json& jr = *jm.find("null");
  for (int i = 0; i < 4; ++i) {
    const std::string k(1, "ACGT"[i]);
    if (!jr.has(k, Json::Object) || !read_sqd(*jr.find(k), null[i])) { err = "Couldn't read null model"; return false; }
  }
  return true;
}

std::string NullParams::write_json() const {  // src/qmodel.cpp:1892-1901
  std::string o = "{\n  \"nullEmit\": " + fmt6(nullEmit) + ",\n  \"null\": {";
  for (int i = 0; i < 4; ++i) o += std::string(" \"") + "ACGT"[i] + "\": " + sqd_json(null[i]) + (i == 3 ? " }" : ",");
  o += " }";
  return o;
}

void NullParams::tables(double& logEmit, double& log1mEmit, double* logSym, double* logQual) const {
  // operands of QuaffNullParams::logLikelihood, src/qmodel.cpp:1875-1890
  logEmit = log(nullEmit);
  log1mEmit = log(1. - nullEmit);
  for (int i = 0; i < 4; ++i) {
    logSym[i] = log(null[i].p);
    for (int k = 0; k < kNQual; ++k) logQual[i * kNQual + k] = log_negbinom(k, null[i].q, null[i].r);
  }
}

// ------------------------------------------------------------------------ log-sum-exp table
const std::vector<double>& lse_table() {  // LogSumExpLookupTable ctor, src/logsumexp.cpp:20-28
  static std::vector<double> table;
  static std::once_flag once;
  std::call_once(once, [] {
    table.resize(kLseEntries + 1);
    for (int n = 0; n < kLseEntries; ++n) {
      const double x = n * .0001;
      table[n] = log(1. + exp(-x));  // log_sum_exp_unary_slow, :105-107
    }
    table[kLseEntries] = 0;  // never interpolated against (x >= 10 returns 0), keeps n+1 in bounds
  });
  return table;
}

static inline double lse_unary(double x) {  // log_sum_exp_unary, src/logsumexp.cpp:84-103
  if (x >= 10 || std::isnan(x) || std::isinf(x)) return 0;
  if (x < 0) return -x;
  const std::vector<double>& t = lse_table();
  const int n = (int)(x / .0001);
  const double dx = x - (n * .0001);
  const double f0 = t[n], f1 = t[n + 1];
  const double df = f1 - f0;
  return f0 + df * (dx / .0001);
}
double log_sum_exp(double a, double b) {  // src/logsumexp.cpp:34-50
  double mx, diff;
  if (a == b) { mx = a; diff = 0; }
  else if (a < b) { mx = b; diff = b - a; }
  else { mx = a; diff = a - b; }
  return mx + lse_unary(diff);
}

// QuaffOverlapScores ctor, src/qoverlap.cpp:9-75: a 3-state approximation of the intersection of two quaff
// transducers.  Built once per (parameters, strand flag) instead of once per read pair.
void OverlapScores::build(const Params& p, const Scores& s, bool yComp) {
  Km = s.Km;
  Kg = s.Kg;
  yComplemented = yComp;
  const size_t KQ = (size_t)Km * kNQ1;
  mmi.assign(KQ * KQ, -INFINITY);
  gap.assign((size_t)3 * Kg * Kg + 6, 0);
  std::vector<double> gapOpen(Kg);
  double pGapIsInsertSum = 0, gapAdjSum = 0;
  for (uint32_t j = 0; j < Kg; ++j) {  // :24-32
    const double readInsertProb = p.beginInsert[j];
    const double readDeleteProb = (1 - p.beginInsert[j]) * p.beginDelete[j];
    gapOpen[j] = readInsertProb + readDeleteProb;
    const double pGapIsInsert = readInsertProb / gapOpen[j];
    const double gapAdjacentProb =
        pGapIsInsert * readInsertProb + (1 - pGapIsInsert) * gapOpen[j] / (1 - p.extendDelete * (1 - gapOpen[j]));
    pGapIsInsertSum += pGapIsInsert;
    gapAdjSum += gapAdjacentProb;
  }
  double* m2m = gap.data();
  double* m2i = m2m + (size_t)Kg * Kg;
  double* m2d = m2i + (size_t)Kg * Kg;
  double* sc = m2d + (size_t)Kg * Kg;
  for (uint32_t i = 0; i < Kg; ++i)
    for (uint32_t j = 0; j < Kg; ++j) {  // :34-39
      m2m[i * Kg + j] = log(1 - gapOpen[i]) + log(1 - gapOpen[j]);
      m2i[i * Kg + j] = log(gapOpen[i]);
      m2d[i * Kg + j] = log(1 - gapOpen[i]) + log(gapOpen[j]);
    }
  const double pGapIsInsert = pGapIsInsertSum / Kg;  // accumulate(...)/size(), :41
  const double meanGapLength = pGapIsInsert / p.extendInsert + (1 - pGapIsInsert) / p.extendDelete;
  const double gapExtendProb = 1 / meanGapLength;
  const double gapAdjacentProb = gapAdjSum / Kg;
  sc[1] = sc[5] = log(gapExtendProb);                                  // i2i = d2d, :46
  sc[2] = sc[4] = log(1 - gapExtendProb) + log(gapAdjacentProb);      // i2d = d2i
  sc[0] = sc[3] = log(1 - gapExtendProb) + log(1 - gapAdjacentProb);  // i2m = d2m
  auto at = [&](uint32_t ki, int qi, uint32_t kj, int qj) -> double& {
    return mmi[((size_t)ki * kNQ1 + qi) * KQ + (size_t)kj * kNQ1 + qj];
  };
  for (uint32_t i = 0; i < Km; ++i)
    for (uint32_t j = 0; j < Km; ++j) {
      const double* xi = &s.ins[(size_t)(i & 3) * kNQ1];  // insert scores of the k-mer's last (emitted) base, :54-58
      const double* yi = &s.ins[(size_t)(j & 3) * kNQ1];
      for (int ik = 0; ik < kNQual; ++ik)
        for (int jk = 0; jk < kNQual; ++jk) {
          double mij = -INFINITY;
          for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t yr = yComp ? 3 - r : r;
            mij = log_sum_exp(mij, log(p.refBase[r]) + s.mat[((size_t)r * Km + i) * kNQ1 + ik] + s.mat[((size_t)yr * Km + j) * kNQ1 + jk]);
          }
          at(i, ik, j, jk) = mij - xi[ik] - yi[jk];
          at(i, ik, j, kNQual) = log_sum_exp(at(i, ik, j, kNQual), mij - xi[ik] - yi[kNQual]);
          at(i, kNQual, j, jk) = log_sum_exp(at(i, kNQual, j, jk), mij - xi[kNQual] - yi[jk]);
          at(i, kNQual, j, kNQual) = log_sum_exp(at(i, kNQual, j, kNQual), mij - xi[kNQual] - yi[kNQual]);
        }
    }
}

// ------------------------------------------------------------------------ synthetic data
static inline uint64_t splitmix64(uint64_t& x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
Rng::Rng(uint64_t seed) {
  for (auto& v : s) v = splitmix64(seed);
}
uint64_t Rng::next() {
  auto rotl = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
  const uint64_t result = rotl(s[1] * 5, 7) * 9;
  const uint64_t t = s[1] << 17;
  s[2] ^= s[0];
  s[3] ^= s[1];
  s[1] ^= s[2];
  s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl(s[3], 45);
  return result;
}

void synth_ref(uint64_t seed, uint64_t len, char* seq) {
  Rng rng(seed);
  for (uint64_t i = 0; i < len; ++i) seq[i] = "ACGT"[rng.below(4)];
}

// start uniform in [0, refLen-readLen]; odd-numbered reads reverse-complemented; per source
// base: delete w.p. .03, else optionally insert one uniform base before it w.p. .03, substitute w.p. .05
// (uniform over ACGT incl. same); qualities uniform Phred 5..25.
void synth_reads(uint64_t seed, const char* ref, uint64_t ref_len, uint32_t n_reads, uint32_t read_len, char* seq,
                 char* qual, uint64_t* offsets) {
  Rng rng(seed);
  uint64_t off = 0;
  std::vector<char> src(read_len);
  for (uint32_t n = 0; n < n_reads; ++n) {
    offsets[n] = off;
    const uint64_t span = ref_len >= read_len ? ref_len - read_len + 1 : 1;
    const uint64_t start = (uint64_t)(rng.uniform() * (double)span);
    const uint32_t len = (uint32_t)std::min<uint64_t>(read_len, ref_len - start);
    for (uint32_t i = 0; i < len; ++i) src[i] = ref[start + i];
    if (n & 1) {
      for (uint32_t i = 0; i < len / 2; ++i) std::swap(src[i], src[len - 1 - i]);
      for (uint32_t i = 0; i < len; ++i) {
        const char c = src[i];
        src[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
      }
    }
    for (uint32_t i = 0; i < len; ++i) {
      if (rng.uniform() < .03) continue;
      if (rng.uniform() < .03) {
        seq[off] = "ACGT"[rng.below(4)];
        qual[off++] = (char)('!' + 5 + rng.below(21));
      }
      char c = src[i];
      if (rng.uniform() < .05) c = "ACGT"[rng.below(4)];
      seq[off] = c;
      qual[off++] = (char)('!' + 5 + rng.below(21));
    }
    if (off == offsets[n]) {  // never emit an empty read
      seq[off] = src[0];
      qual[off++] = (char)('!' + 15);
    }
  }
  offsets[n_reads] = off;
}

}  // namespace qf
