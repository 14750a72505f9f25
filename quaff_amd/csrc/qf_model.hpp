// qf_model.hpp — host-side model of the quaff pair HMM: parameter JSON, log-space score
// tables, null model, log-sum-exp table, synthetic workload generator.  Pure C++17, no HIP.
// Reference behaviour cited per function (paths under /root/reference).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace qf {

constexpr int kNQual = 94;   // FastSeq::qualScoreRange, src/fastseq.cpp:69
constexpr int kNQ1 = 95;     // + slot 94: quality-marginalised logSymProb
constexpr int kLseEntries = 100001;  // src/logsumexp.cpp:7-9

// ---- JSON (value tree; numbers parsed like gason's string2double, src/gason.cpp:73-117) ----
struct Json {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  double num = 0;
  bool b = false;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;  // insertion order kept
  const Json* find(const std::string& key) const;
  bool has(const std::string& key, Type t) const { const Json* j = find(key); return j && j->type == t; }
};
double gason_number(const char* s, const char** end);
bool parse_json(const std::string& text, Json& out, std::string& err);

// ---- parameters ----------------------------------------------------------------------
struct SymQualDist { double p = 0.25, q = 0.5, r = kNQual / 2; };  // src/qmodel.cpp:52-56

std::string kmer_to_string(uint64_t kmer, unsigned k);  // src/fastseq.cpp:44-49

struct Params {  // QuaffParams, src/qmodel.h:147-163
  unsigned match_len = 1, gap_len = 0;  // DefaultMatchKmerContext / DefaultIndelKmerContext
  uint32_t Km() const { return 1u << (2 * match_len); }
  uint32_t Kg() const { return 1u << (2 * gap_len); }
  double refBase[4] = {.25, .25, .25, .25};
  std::vector<double> beginInsert, beginDelete;
  double extendInsert = .5, extendDelete = .5;
  SymQualDist insert[4];
  std::vector<SymQualDist> match;  // [4][Km]: [ref token][read context k-mer]
  void resize();
  bool read_json(const Json& j, std::string& err);  // src/qmodel.cpp:230-271
  std::string write_json() const;                   // src/qmodel.cpp:187-218
};
extern const char* const kDefaultParamsJson;  // data of src/defaultparams.cpp:12-46

struct Scores {  // QuaffScores, src/qmodel.cpp:296-325, flattened
  unsigned match_len = 1, gap_len = 0;
  uint32_t Km = 4, Kg = 1;
  std::vector<double> ins;    // [4][95]
  std::vector<double> mat;    // [4][Km][95]
  std::vector<double> trans;  // m2m[Kg] m2i[Kg] m2d[Kg] m2e[Kg] d2d d2m i2i i2m
  void build(const Params& p);
};
double log_negbinom(int k, double p, double n);  // src/negbinom.cpp:30-32

struct NullParams {  // QuaffNullParams, src/qmodel.cpp:1806-1907
  double nullEmit = .5;
  SymQualDist null[4];
  bool read_json(const Json& j, std::string& err);
  std::string write_json() const;
  // tables consumed by the device prep kernel: logEmit, log1mEmit, logSym[4], logQual[4][94]
  void tables(double& logEmit, double& log1mEmit, double* logSym, double* logQual) const;
};

const std::vector<double>& lse_table();  // src/logsumexp.cpp:20-28
double log_sum_exp(double a, double b);   // src/logsumexp.cpp:34-50,84-103 (table-interpolated)

struct OverlapScores {  // QuaffOverlapScores, src/qoverlap.cpp:9-75, one strand flag
  uint32_t Km = 4, Kg = 1;
  bool yComplemented = false;
  // mmi[(ki*95+qi) * (Km*95) + (kj*95+qj)]: match-minus-insert pair emission; q index 94 = that read has no quality
  std::vector<double> mmi;
  std::vector<double> gap;   // m2m[Kg][Kg] m2i[Kg][Kg] m2d[Kg][Kg] | i2m i2i i2d d2m d2i d2d
  void build(const Params& p, const Scores& s, bool yComp);
};

std::string fmt6(double x);  // default ostream << double

// ---- synthetic generator -----------------------------------------------------
struct Rng {  // splitmix64-seeded xoshiro256**
  uint64_t s[4];
  explicit Rng(uint64_t seed);
  uint64_t next();
  double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
  uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};
void synth_ref(uint64_t seed, uint64_t len, char* seq);
void synth_reads(uint64_t seed, const char* ref, uint64_t ref_len, uint32_t n_reads, uint32_t read_len,
                 char* seq, char* qual, uint64_t* offsets);

}  // namespace qf
