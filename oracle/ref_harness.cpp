// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" shim over the parts of the REAL reference that build here
// without GSL: src/diagenv.cpp, src/fastseq.cpp, src/logsumexp.cpp, src/gason.cpp
// (+ util.cpp, logger.cpp they link against).  oracle/Makefile compiles those
// sources where they lie under /root/reference into oracle/_ref/libquaffref.so;
// nothing from the reference is copied into this repository.  qmodel.cpp,
// negbinom.cpp (and hence qoverlap.cpp's link closure) need GSL headers and are
// NOT built — see DESIGN.md.
//
// Used by tests/test_oracle_vs_ref.py to pin oracle/quaff_oracle.c's seeding,
// k-mer, log-sum-exp and JSON-number restatements bit-for-bit.  The tests skip
// when /root/reference (and so the .so) is absent, e.g. on the GPU box.
#include <cstring>
#include <vector>
#include "fastseq.h"
#include "diagenv.h"
#include "logsumexp.h"
#include "gason.h"

static FastSeq mk(const char* seq) { FastSeq fs; fs.name = "s"; fs.seq = seq; return fs; }

extern "C" {

// DiagonalEnvelope::initSparse / initFull (src/diagenv.cpp:11-106) with a KmerIndex of y
// (src/fastseq.cpp:240-256).  Returns #diagonals; *storage = totalStorageSize.
int ref_envelope(const char* xseq, const char* yseq, int sparse, int k, int band, int threshold,
                 unsigned long long cellSize, unsigned long long maxSize, int* diags, unsigned long long* storage)
{
  FastSeq x = mk(xseq), y = mk(yseq);
  DiagonalEnvelope env(x, y);
  if (sparse) {
    KmerIndex idx(y, dnaAlphabet, (SeqIdx)k);
    env.initSparse(idx, (unsigned)band, threshold, (size_t)cellSize, (size_t)maxSize);
  } else
    env.initFull();
  for (size_t n = 0; n < env.diagonals.size(); ++n) diags[n] = env.diagonals[n];
  *storage = env.totalStorageSize;
  return (int)env.diagonals.size();
}

// number of cells visited by the fill loops: sum_j |begin(j)..end(j)| (src/diagenv.h:75-141)
unsigned long long ref_envelope_cells(const char* xseq, const char* yseq, const int* diags, int nd)
{
  FastSeq x = mk(xseq), y = mk(yseq);
  DiagonalEnvelope env(x, y);
  env.diagonals = vguard<int>(diags, diags + nd);
  env.initStorage();
  unsigned long long cells = 0;
  for (SeqIdx j = 1; j <= env.yLen; ++j)
    for (DiagonalEnvelope::iterator pi = env.begin(j); !pi.finished(); ++pi) ++cells;
  return cells;
}

// FastSeq::kmers (src/fastseq.cpp:85-99)
void ref_kmers(const char* seq, int k, unsigned long long* out)
{
  FastSeq s = mk(seq);
  const vguard<Kmer> km = s.kmers(dnaAlphabet, (unsigned)k);
  for (size_t n = 0; n < km.size(); ++n) out[n] = km[n];
}

// FastSeq::qualScores (src/fastseq.cpp:101-109)
void ref_quals(const char* seq, const char* qual, unsigned int* out)
{
  FastSeq s = mk(seq); s.qual = qual;
  const vguard<QualScore> q = s.qualScores();
  for (size_t n = 0; n < q.size(); ++n) out[n] = q[n];
}

// revcomp (src/fastseq.cpp:209-216)
void ref_revcomp(const char* seq, char* out)
{
  const string r = revcomp(string(seq));
  memcpy(out, r.c_str(), r.size() + 1);
}

// log_sum_exp (src/logsumexp.cpp:34-50) and the 3-argument form (:52-54)
double ref_lse(double a, double b) { return log_sum_exp(a, b); }
double ref_lse3(double a, double b, double c) { return log_sum_exp(a, b, c); }
double ref_lse_unary(double x) { return log_sum_exp_unary(x); }

// gason number parsing (src/gason.cpp:73-117) through the public parser
double ref_json_number(const char* text)
{
  std::vector<char> buf(text, text + strlen(text) + 1);
  char* endptr = 0;
  JsonValue value;
  JsonAllocator alloc;
  if (jsonParse(buf.data(), &endptr, &value, alloc) != JSON_OK) return -12345.678;
  return value.toNumber();
}

}
