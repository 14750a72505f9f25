// quaff_cli.cpp — `quaff {align,count,train,overlap}` command-line shell over libquaffhip.
// Keeps the reference's commands, the DP-relevant flags (t/quaff.cpp:122-236, src/qmodel.cpp:747-833,1916-1993,
// 2485-2529), its input formats (FASTA/FASTQ, optionally gzipped; params / null / counts JSON) and its output
// formats (Stockholm, gapped FASTA, SAM, refseq; params / counts JSON), so that it is a drop-in for the hot path.
// All DP runs on the GPU through the C ABI (include/quaff_hip.h); this file is host plumbing only.  Not provided:
// remote/ssh/EC2/qsub execution, logging levels (-v* are accepted and ignored); -threads only divides -kmatchmax's memory.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <iterator>
#include <thread>
#include <vector>
#include <chrono>
#include <future>
#include <map>
#include <memory>
#include <mutex>

#include "../../include/quaff_hip.h"
extern "C" double qf_debug_alloc_ms(void);   // qf_internal.h: time this process spent growing device buffers
#include "qf_em.hpp"
#include "qf_model.hpp"

using namespace std;
using namespace qf;

// (EarlySession below: the device contexts come up on a thread of their own while the input is parsed; the process must not run
// its exit handlers under that thread's feet)
static std::shared_future<void>* g_background = nullptr;
static thread_local bool t_in_background = false;
[[noreturn]] static void Fail(const string& msg) {  // Fail(), src/util.cpp:89-98
  cerr << msg << endl;
  if (g_background && g_background->valid() && !t_in_background) g_background->wait();
  exit(EXIT_FAILURE);
}
#define Require(cond, msg) do { if (!(cond)) Fail(msg); } while (0)

// ------------------------------------------------------------------------------------------ sequences
struct Coords {  // SeqIntervalCoords, src/fastseq.h:29-39
  string name;
  unsigned start = 0, end = 0;
  bool rev = false;
  bool isNull() const { return name.empty(); }
  Coords compose(const Coords& src) const {  // src/fastseq.cpp:51-65
    if (src.isNull()) return *this;
    Coords c;
    c.name = src.name;
    c.rev = rev != src.rev;
    if (src.rev) { c.start = src.end - end + 1; c.end = src.end - start + 1; }
    else { c.start = start + src.start - 1; c.end = end + src.start - 1; }
    return c;
  }
};
// A sequence or quality string: its own storage, or a view of the input file as it lies mapped in memory (`keep` holds the
// mapping).  A FASTQ record whose sequence and quality sit on one line each -- every record of a sequencer's file -- is never
// copied: building 200 000 strings for the 100 k reads of config 2 was 0.2 s of page faults, five device calls' worth.
class Text {
  const char* p_ = nullptr;
  size_t n_ = 0;
  bool owned_ = false;
  string own_;
  std::shared_ptr<const void> keep_;
  void fix() { if (owned_) { p_ = own_.data(); n_ = own_.size(); } }
 public:
  Text() = default;
  Text(const Text& o) : p_(o.p_), n_(o.n_), owned_(o.owned_), own_(o.own_), keep_(o.keep_) { fix(); }
  Text(Text&& o) noexcept : p_(o.p_), n_(o.n_), owned_(o.owned_), own_(std::move(o.own_)), keep_(std::move(o.keep_)) { fix(); o.clear(); }
  Text& operator=(const Text& o) { if (this != &o) { p_ = o.p_; n_ = o.n_; owned_ = o.owned_; own_ = o.own_; keep_ = o.keep_; fix(); } return *this; }
  Text& operator=(Text&& o) noexcept {
    if (this != &o) { p_ = o.p_; n_ = o.n_; owned_ = o.owned_; own_ = std::move(o.own_); keep_ = std::move(o.keep_); fix(); o.clear(); }
    return *this;
  }
  Text& operator=(string t) { own_ = std::move(t); owned_ = true; keep_.reset(); fix(); return *this; }
  void view(const char* p, size_t n, std::shared_ptr<const void> keep) { own_.clear(); owned_ = false; p_ = p; n_ = n; keep_ = std::move(keep); }
  void clear() { own_.clear(); owned_ = false; p_ = nullptr; n_ = 0; keep_.reset(); }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  const char* data() const { return p_; }
  const char* begin() const { return p_; }
  const char* end() const { return p_ + n_; }
  char operator[](size_t i) const { return p_[i]; }
  operator std::string_view() const { return std::string_view(p_, n_); }
  string str() const { return string(p_, n_); }
  string substr(size_t pos, size_t len) const { return pos >= n_ ? string() : string(p_ + pos, std::min(len, n_ - pos)); }
};
static ostream& operator<<(ostream& o, const Text& t) { return o.write(t.data(), (std::streamsize)t.size()); }
struct FastSeq {
  string name, comment;
  Text seq, qual;
  Coords source;
  bool hasQual() const { return qual.size() == seq.size(); }
};
static char complementChar(char c) {  // dnaComplementChar, src/fastseq.cpp:22-25
  switch (toupper(c)) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; }
  return c;
}
static FastSeq revcomp(const FastSeq& s) {  // FastSeq::revcomp, src/fastseq.cpp:218-230
  FastSeq r;
  r.name = "revcomp(" + s.name + ")";
  r.comment = s.comment;
  string rs(s.seq.size(), ' ');
  for (size_t i = 0; i < s.seq.size(); ++i) rs[s.seq.size() - 1 - i] = complementChar(s.seq[i]);
  r.seq = std::move(rs);
  r.qual = string(std::make_reverse_iterator(s.qual.end()), std::make_reverse_iterator(s.qual.begin()));
  Coords c;
  c.name = s.name; c.start = 1; c.end = (unsigned)s.seq.size(); c.rev = true;
  r.source = c.compose(s.source);
  return r;
}
static void writeFasta(ostream& out, const FastSeq& s) {
  out << '>' << s.name;
  if (s.comment.size()) out << ' ' << s.comment;
  out << endl << s.seq << endl;
}

// FASTA / FASTQ reader with kseq's conventions (kseq/kseq.h, src/fastseq.cpp:133-171): name up to the first blank,
// rest of the header line is the comment, multi-line sequence, quality kept only when as long as the sequence.
// Three passes over the file in memory (config 2's device call takes 40 ms; a 200 MB FASTQ of 100 k reads parsed by one
// thread with a string per field took 430): (1) the lines, by memchr; (2) the records -- which line is a header, which lines
// are its sequence and its quality: the one sequential decision, and it reads only first characters and lengths; (3) the
// strings, built by up to eight threads, a block of records each.
static vector<FastSeq> readFastSeqs(const string& filename) {
  // the bytes: the file mapped as it lies in the page cache, or inflated by zlib when it is gzipped (or cannot be mapped)
  struct Mapping {
    void* p = MAP_FAILED;
    size_t n = 0;
    string inflated;
    ~Mapping() { if (p != MAP_FAILED) munmap(p, n); }
  };
  const auto keep = std::make_shared<Mapping>();   // lives as long as a record still views it
  Mapping& map = *keep;
  string& inflated = map.inflated;
  const char* d = nullptr;
  size_t n = 0;
  {
    const int fd = open(filename.c_str(), O_RDONLY);
    Require(fd >= 0, "Couldn't open " + filename);
    unsigned char magic[2] = {0, 0};
    const bool gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    struct stat st;
    if (!gz && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      map.p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
      if (map.p != MAP_FAILED) { map.n = n = (size_t)st.st_size; d = (const char*)map.p; }
    }
    close(fd);
    if (!d) {
      gzFile fp = gzopen(filename.c_str(), "r");
      Require(fp != Z_NULL, "Couldn't open " + filename);
      (void)gzbuffer(fp, 1 << 20);
      vector<char> buf(1 << 20);
      int m;
      while ((m = gzread(fp, buf.data(), (unsigned)buf.size())) > 0) inflated.append(buf.data(), m);
      gzclose(fp);
      d = inflated.data();
      n = inflated.size();
    }
  }
  // (1) lines: [begin, end) without the newline and a trailing carriage return
  vector<std::pair<size_t, size_t>> lines;
  lines.reserve(n / 64 + 16);
  for (size_t p = 0; p < n;) {
    const char* nl = (const char*)memchr(d + p, '\n', n - p);
    const size_t e = nl ? (size_t)(nl - d) : n;
    size_t le = e;
    if (le > p && d[le - 1] == '\r') --le;
    lines.push_back({p, le});
    p = e + 1;
  }
  // a line's length without white space; every isspace() character is <= ' ', sequence and quality characters are not: eight
  // bytes at a time, "has a byte below 0x21" = (w - 0x21 x ones) & ~w & (0x80 x ones)
  auto blanks = [&](size_t k) -> bool {
    size_t q = lines[k].first;
    const size_t e = lines[k].second;
    for (; q + 8 <= e; q += 8) {
      uint64_t w;
      memcpy(&w, d + q, 8);
      if ((w - 0x2121212121212121ull) & ~w & 0x8080808080808080ull) return true;
    }
    for (; q < e; ++q) if ((unsigned char)d[q] <= (unsigned char)' ') return true;
    return false;
  };
  auto seqLength = [&](size_t k) -> size_t {
    if (!blanks(k)) return lines[k].second - lines[k].first;
    size_t len = 0;
    for (size_t q = lines[k].first; q < lines[k].second; ++q) len += !isspace((unsigned char)d[q]);
    return len;
  };
  auto first = [&](size_t k) -> char { return lines[k].second > lines[k].first ? d[lines[k].first] : '\0'; };
  // (2) records
  struct Rec { size_t name, seq0, seq1, qual0, qual1, seqLen; bool qualOk; };
  vector<Rec> recs;
  const size_t nl = lines.size();
  for (size_t k = 0; k < nl;) {
    if (first(k) != '>' && first(k) != '@') { ++k; continue; }
    Rec r{k, k + 1, k + 1, 0, 0, 0, false};
    ++k;
    while (k < nl && !(first(k) == '>' || first(k) == '@' || first(k) == '+')) { r.seqLen += seqLength(k); ++k; }
    r.seq1 = r.qual0 = r.qual1 = k;
    if (k < nl && first(k) == '+') {
      ++k;
      r.qual0 = k;
      size_t qlen = 0;
      while (k < nl && qlen < r.seqLen) { qlen += lines[k].second - lines[k].first; ++k; }
      r.qual1 = k;
      r.qualOk = qlen == r.seqLen;
    }
    recs.push_back(r);
  }
  // (3) strings
  vector<FastSeq> seqs(recs.size());
  auto build = [&](size_t lo, size_t hi) {
    for (size_t x = lo; x < hi; ++x) {
      const Rec& r = recs[x];
      FastSeq& s = seqs[x];
      const size_t lb = lines[r.name].first, le = lines[r.name].second;
      size_t sp = lb + 1;
      while (sp < le && d[sp] != ' ' && d[sp] != '\t') ++sp;
      s.name.assign(d + lb + 1, sp - lb - 1);
      if (sp < le) s.comment.assign(d + sp + 1, le - sp - 1);
      // one line of sequence without blanks (one line of quality): a view of the file; anything else is put together
      if (r.seq1 == r.seq0 + 1 && lines[r.seq0].second - lines[r.seq0].first == r.seqLen) s.seq.view(d + lines[r.seq0].first, r.seqLen, keep);
      else {
        string t;
        t.reserve(r.seqLen);
        for (size_t k = r.seq0; k < r.seq1; ++k) {
          if (!blanks(k)) t.append(d + lines[k].first, lines[k].second - lines[k].first);
          else for (size_t q = lines[k].first; q < lines[k].second; ++q) if (!isspace((unsigned char)d[q])) t += d[q];
        }
        s.seq = std::move(t);
      }
      if (r.qualOk && r.qual1 == r.qual0 + 1) s.qual.view(d + lines[r.qual0].first, r.seqLen, keep);
      else if (r.qualOk) {
        string t;
        t.reserve(r.seqLen);
        for (size_t k = r.qual0; k < r.qual1; ++k) t.append(d + lines[k].first, lines[k].second - lines[k].first);
        s.qual = std::move(t);
      }
    }
  };
  const size_t T = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)std::thread::hardware_concurrency(), recs.size() / 512 + 1}));
  vector<std::thread> th;
  for (size_t t = 1; t < T; ++t) th.emplace_back(build, recs.size() * t / T, recs.size() * (t + 1) / T);
  build(0, recs.size() / T);
  for (auto& t : th) t.join();
  if (seqs.empty()) cerr << "Warning: Couldn't read any sequences from " << filename << endl;
  return seqs;
}

// ------------------------------------------------------------------------------------------ alignments
// A pairwise alignment in the form the device returns it: edit runs over an interval of each source sequence.  Every
// output format is produced from the runs (the reference materialises gapped rows first and derives CIGAR strings and
// ungapped sequences back from them, src/qmodel.cpp:553-676; the bytes written are the same).
//   run = (length << 2) | op;  op 0: a column with a base of x and a base of y,  1: y only (x gapped),  2: x only (y gapped)
struct Hit {
  const FastSeq* src[2] = {nullptr, nullptr};   // x (reference / read_x), y (read / read_y)
  unsigned lo[2] = {0, 0};                      // 1-based position of the first source base each side covers
  vector<uint32_t> runs;
  string label[2], note[2];                     // row names and "#=GS CC" comments
  Coords origin[2];                             // the covered intervals in the coordinates of the original (unreversed) input
  double score = -INFINITY;

  static bool consumes(int side, uint32_t op) { return op == 0 || op == (side == 0 ? 2u : 1u); }
  size_t columns() const {
    size_t n = 0;
    for (uint32_t r : runs) n += r >> 2;
    return n;
  }
  size_t span(int side) const {   // source bases covered
    size_t n = 0;
    for (uint32_t r : runs) if (consumes(side, r & 3u)) n += r >> 2;
    return n;
  }
  // the side's bases (or quality characters) laid out over the alignment columns, `gap` where the side has none
  string laidOut(int side, bool quality, char gap) const {
    const Text& text = quality ? src[side]->qual : src[side]->seq;
    string out;
    out.reserve(columns());
    size_t at = lo[side] - 1;
    for (uint32_t r : runs) {
      const uint32_t len = r >> 2;
      if (consumes(side, r & 3u)) { out.append(text.data() + at, len); at += len; }
      else out.append(len, gap);
    }
    return out;
  }
  string covered(int side, bool quality) const { return (quality ? src[side]->qual : src[side]->seq).substr(lo[side] - 1, span(side)); }
  // Alignment::cigarString (letter before count); backwards = the alignment seen from the other strand
  string cigar(bool backwards) const {
    vector<uint32_t> rr(runs);
    if (backwards) reverse(rr.begin(), rr.end());
    string out(qf_cigar_string(rr.data(), (uint32_t)rr.size(), nullptr, 0), '\0');
    if (!out.empty()) { out.push_back('\0'); qf_cigar_string(rr.data(), (uint32_t)rr.size(), &out[0], out.size()); out.pop_back(); }
    return out;
  }
};

// adjacent runs of the same op become one (the device never emits such neighbours; the overlap re-pairing below can)
static void appendRun(vector<uint32_t>& runs, uint32_t op, size_t len) {
  if (!len) return;
  if (!runs.empty() && (runs.back() & 3u) == op) runs.back() += (uint32_t)len << 2;
  else runs.push_back(((uint32_t)len << 2) | op);
}

// Stockholm block (writeStockholm, src/qmodel.cpp:553-606): quality lines outside, the identity line between the rows
static void writeStockholm(ostream& out, const Hit& h) {
  const string xs = h.laidOut(0, false, '-'), ys = h.laidOut(1, false, '-');
  string ident(xs.size(), '-');
  for (size_t c = 0; c < xs.size(); ++c)
    if (xs[c] != '-' && ys[c] != '-') ident[c] = toupper(xs[c]) == toupper(ys[c]) ? (char)toupper(xs[c]) : ':';
  vector<pair<string, string>> lines;
  if (h.src[0]->hasQual()) lines.push_back({"#=GR " + h.label[0] + " QS", h.laidOut(0, true, '~')});
  lines.push_back({h.label[0], xs});
  lines.push_back({"#=GC id", ident});
  lines.push_back({h.label[1], ys});
  if (h.src[1]->hasQual()) lines.push_back({"#=GR " + h.label[1] + " QS", h.laidOut(1, true, '~')});
  size_t tag = 0;
  for (const auto& l : lines) tag = max(tag, l.first.size());
  const size_t perBlock = max(tag, 79 - tag);
  // the record is composed in memory and written in one piece (a line at a time with std::endl the writer flushed 24 M times
  // for 57 k overlaps of 2 kb reads: 17 of the command's 20 s)
  string buf;
  buf.reserve(lines.size() * (xs.size() + (xs.size() / perBlock + 1) * (tag + 2)) + 256);
  buf += "# STOCKHOLM 1.0\n#=GF Score ";
  buf += fmt6(h.score);
  buf += '\n';
  for (int k = 0; k < 2; ++k) if (h.note[k].size()) { buf += "#=GS CC "; buf += h.label[k]; buf += ' '; buf += h.note[k]; buf += '\n'; }
  for (size_t c = 0; c < xs.size(); c += perBlock) {
    if (c) buf += '\n';
    for (const auto& l : lines) {
      buf += l.first;
      buf.append(tag - l.first.size() + 1, ' ');
      buf.append(l.second, c, perBlock);
      buf += '\n';
    }
  }
  buf += "//\n";
  out.write(buf.data(), (std::streamsize)buf.size());
}

// SAM record (writeSam, src/qmodel.cpp:608-616).  An alignment against a reverse-complemented reference is reported on
// the reference's forward strand: the reference flips the whole alignment first (Alignment::revcomp, :655-660), so the
// read's strand bit toggles, the CIGAR reads backwards, and - because FastSeq::revcomp (src/fastseq.cpp:218-230) composes
// the coordinates from the length of the GAPPED row - the position printed is end - columns + 1, not the interval's start.
static void writeSam(ostream& out, const Hit& h) {
  const bool flip = h.origin[0].rev;
  const bool readRev = h.origin[1].rev != flip;
  const unsigned pos = flip ? h.origin[0].end - (unsigned)h.columns() + 1 : h.origin[0].start;
  out << h.origin[1].name << '\t' << (readRev ? 16 : 0) << '\t' << h.origin[0].name << '\t' << pos << "\t0\t"
      << h.cigar(flip) << "\t*\t0\t0\t*\t*\tAS:i:" << ((int)round(h.score)) << '\n';
}

struct Printer {  // QuaffAlignmentPrinter, src/qmodel.cpp:2480-2600
  enum Format { Stockholm, Fasta, Sam, Refseq } format = Stockholm;
  double threshold = 0;
  string alignFilename;
  ofstream alignFile;
  ostream& stream(ostream& out) { return alignFilename.size() ? (ostream&)alignFile : out; }
  bool parse(deque<string>& av) {
    if (av.empty()) return false;
    const string arg = av[0];
    if (arg == "-format") {
      Require(av.size() > 1, arg + " must have an argument");
      const string f = av[1];
      if (f == "fasta") format = Fasta; else if (f == "stockholm") format = Stockholm; else if (f == "sam") format = Sam;
      else if (f == "refseq") format = Refseq; else Fail("Unknown format: " + f);
      av.pop_front(); av.pop_front();
      return true;
    }
    if (arg == "-threshold") { Require(av.size() > 1, arg + " must have an argument"); threshold = atof(av[1].c_str()); av.pop_front(); av.pop_front(); return true; }
    if (arg == "-nothreshold") { threshold = -INFINITY; av.pop_front(); return true; }
    if (arg == "-savealign") { Require(av.size() > 1, arg + " must have an argument"); alignFilename = av[1]; av.pop_front(); av.pop_front(); return true; }
    return false;
  }
  void header(ostream& out, const vector<FastSeq>& refs, bool groupByQuery) {  // writeAlignmentHeader :2559-2564
    if (alignFilename.size()) alignFile.open(alignFilename);
    if (format == Sam) {
      ostream& o = stream(out);
      o << "@HD\tVN:1.0\t" << (groupByQuery ? "GO:query" : "SO:unknown") << endl;
      for (const auto& s : refs) if (s.source.isNull()) o << "@SQ\tSN:" << s.name << "\tLN:" << s.seq.size() << endl;
    }
  }
  // A batch of alignments: the records are composed by a few threads side by side (1.9 GB of Stockholm text for the overlaps of
  // 5 000 reads is the command's longest phase) and written in order.
  void writeAll(ostream& out, const vector<vector<Hit>>& blocks) {
    vector<const Hit*> all;
    for (const auto& b : blocks) for (const Hit& h : b) all.push_back(&h);
    const size_t T = std::min<size_t>(8, std::max<size_t>(1, all.size() / 64));
    vector<string> text(T);
    vector<std::thread> th;
    auto work = [&](size_t k) {
      std::ostringstream os;
      const size_t lo = all.size() * k / T, hi = all.size() * (k + 1) / T;
      for (size_t a = lo; a < hi; ++a) emit(os, *all[a]);
      text[k] = os.str();
    };
    for (size_t k = 1; k < T; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& t : th) t.join();
    ostream& o = stream(out);
    for (const string& t : text) o.write(t.data(), (std::streamsize)t.size());
  }
  void write(ostream& out, const Hit& h) { emit(stream(out), h); }
  void emit(ostream& o, const Hit& h) {  // writeAlignment :2566-2600
    if (!(h.score >= threshold)) return;
    auto fasta = [&](const string& name, const string& comment, const string& text) {
      o << '>' << name;
      if (comment.size()) o << ' ' << comment;
      o << endl << text << endl;
    };
    switch (format) {
      case Fasta:
        for (int k = 0; k < 2; ++k) fasta(h.label[k], h.note[k], h.laidOut(k, false, '-'));
        o << endl;
        break;
      case Stockholm: writeStockholm(o, h); break;
      case Sam: writeSam(o, h); break;
      case Refseq: fasta(h.label[0], "matches(" + h.label[1] + ") " + h.note[0], h.covered(0, false)); break;   // Alignment::getUngapped(0)
    }
  }
};

// ------------------------------------------------------------------------------------------ options
struct Opts {
  deque<string> av;
  deque<string> implicit;
  qf_dp_config cfg{1, 1, 6, 14, 64, 0, 0};
  bool autoMem = false;
  uint64_t memTotal = 0;
  unsigned threads = 1;
  bool wantComm = false;   // train / count: the device contexts share an RCCL communicator
  int gpus = 1;   // -gpus <n|all>: devices the read batches are spread over (this build's counterpart of -threads)
  vector<string> refFiles, readFiles;
  bool fwdstrand = false, noquals = false;
  string paramsFile, nullFile, saveNull;
  bool parseConfig(bool refseq) {  // parseRefSeqConfigArgs / parseGeneralConfigArgs, src/qmodel.cpp:747-833
    if (av.empty()) return false;
    const string arg = av[0];
    auto val = [&]() { Require(av.size() > 1, arg + " must have an argument"); const string v = av[1]; av.pop_front(); av.pop_front(); return v; };
    if (refseq && arg == "-global") { cfg.local = 0; av.pop_front(); return true; }
    if (arg == "-kmatchband") { cfg.band_size = atoi(val().c_str()); return true; }
    if (arg == "-kmatch") {
      cfg.kmer_len = atoi(val().c_str());
      Require(cfg.kmer_len >= 5 && cfg.kmer_len <= 32, arg + " out of range (" + to_string(cfg.kmer_len) + "). Try 5 to 32");
      return true;
    }
    if (arg == "-kmatchn") { cfg.kmer_threshold = atoi(val().c_str()); return true; }
    if (arg == "-kmatchmb") {
      cfg.max_size = (uint64_t)atoi(val().c_str()) << 20;
      if (cfg.max_size == 0) cfg.max_size = hostMemory();   // "-kmatchmb 0": all of system RAM, src/qmodel.cpp:790-793
      cfg.kmer_threshold = -1;
      autoMem = false;
      return true;
    }
    if (arg == "-kmatchmax") {   // system RAM shared by the worker threads, src/qmodel.cpp:801-807,1058-1060
      memTotal = hostMemory();
      cfg.kmer_threshold = -1;
      autoMem = true;
      av.pop_front();
      return true;
    }
    if (arg == "-kmatchoff") { cfg.sparse = 0; av.pop_front(); return true; }
    if (arg == "-gpus") { const string v = val(); gpus = v == "all" ? -1 : atoi(v.c_str()); Require(gpus == -1 || gpus >= 1, "-gpus needs a positive count or 'all'"); return true; }
    // the GPU batches instead of threading; the count only divides -kmatchmax's memory, as in the reference
    if (arg == "-threads") { threads = (unsigned)atoi(val().c_str()); return true; }
    if (arg == "-maxthreads") { threads = std::max(1u, std::thread::hardware_concurrency()); av.pop_front(); return true; }
    return false;
  }
  static uint64_t hostMemory() {   // getMemorySize(), src/memsize.cpp (the sysconf branch)
    const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
    Require(pages > 0 && psz > 0, "Can't figure out available system memory; you will need to specify a size");
    return (uint64_t)pages * (uint64_t)psz;
  }
  void finishConfig() {   // QuaffDPConfig::effectiveMaxSize, src/qmodel.cpp:1058-1060
    if (!autoMem) return;
    Require(threads > 0, "Please allocate at least one thread");
    cfg.max_size = memTotal / threads;
  }
  bool parseLog() {  // Logger::parseLogArgs, src/logger.cpp:48-83: accepted and ignored
    if (av.empty()) return false;
    const string& arg = av[0];
    if (arg == "-log") { Require(av.size() > 1, "-log must have an argument"); av.pop_front(); av.pop_front(); return true; }
    if (arg == "-verbose" || arg == "-nocolor" || (arg.size() >= 2 && arg[0] == '-' && arg[1] == 'v' &&
        arg.find_first_not_of("v0123456789", 1) == string::npos)) { av.pop_front(); return true; }
    return false;
  }
  bool parseFiles(bool wantRefs) {
    if (av.empty()) return false;
    const string arg = av[0];
    auto val = [&]() { Require(av.size() > 1, arg + " needs an argument"); const string v = av[1]; av.pop_front(); av.pop_front(); return v; };
    if (arg == "-params") { paramsFile = val(); return true; }
    if (arg == "-null") { nullFile = val(); return true; }
    if (arg == "-savenull") { saveNull = val(); return true; }
    if (wantRefs && arg == "-ref") { refFiles.push_back(val()); return true; }
    if (arg == "-read") { readFiles.push_back(val()); return true; }
    if (arg == "-fwdstrand") { fwdstrand = true; av.pop_front(); return true; }
    return false;
  }
  bool parseUnknown() {  // OptParser::parseUnknown, src/optparser.cpp:30-53
    if (av.empty()) return false;
    const string arg = av[0];
    if (arg[0] == '-' || implicit.empty()) Fail("Unknown option: " + arg + "\nError parsing command-line options");
    av.push_front(implicit.front());
    if (implicit.size() > 1) implicit.pop_front();
    return true;
  }
};

static string slurp(const string& fn) {
  ifstream in(fn);
  Require(!in.fail(), "Couldn't open " + fn);
  stringstream ss;
  ss << in.rdbuf();
  return ss.str();
}

struct SeqSet {  // SeqList::loadSequences, t/quaff.cpp:610-636
  vector<FastSeq> seqs;
  size_t nOriginals = 0;
  void load(const vector<string>& files, const string& type, const string& tag, bool wantQual, bool wantRevcomps, bool requireUnique) {
    Require(!files.empty(), "Please specify at least one " + type + " file using " + tag);
    for (const auto& f : files)
      for (auto& fs : readFastSeqs(f)) {
        if (wantQual) Require(fs.hasQual(), "Sequence " + fs.name + " in file " + f + " does not have quality scores");
        else fs.qual.clear();
        if (fs.seq.size()) seqs.push_back(std::move(fs));
      }
    nOriginals = seqs.size();
    if (wantRevcomps) for (size_t n = 0; n < nOriginals; ++n) seqs.push_back(revcomp(seqs[n]));
    Require(!seqs.empty(), "Please specify a valid " + type + " file using " + tag);
    if (requireUnique) {
      set<string> names, dups;
      for (const auto& s : seqs) { if (names.count(s.name)) dups.insert(s.name); names.insert(s.name); }
      if (!dups.empty()) {
        cerr << "Duplicate names:";
        for (const auto& d : dups) cerr << ' ' << d;
        cerr << endl;
        Fail("All " + type + " sequence names are required to be unique");
      }
    }
  }
};

#define QF(ctx, call) do { if ((call) != QF_OK) Fail(string("libquaffhip: ") + qf_last_error(ctx)); } while (0)
// inside a per-device worker thread: reported by the main thread once every worker has finished
#define QFT(ctx, call) do { if ((call) != QF_OK) throw std::runtime_error(string("libquaffhip: ") + qf_last_error(ctx)); } while (0)

// fn(k) for k < n, one host thread per device context (the reference runs one task loop per -threads thread,
// src/qmodel.cpp:2870-2882; here a "thread" owns a GPU and takes whole batches)
template <class F>
static void onDevices(size_t n, F&& fn) {
  vector<string> err(n);
  vector<std::thread> th;
  auto guarded = [&](size_t k) { try { fn(k); } catch (const std::exception& e) { err[k] = e.what(); } };
  for (size_t k = 1; k < n; ++k) th.emplace_back(guarded, k);
  guarded(0);
  for (auto& t : th) t.join();
  for (const auto& e : err) if (e.size()) Fail(e);
}

// QUAFF_HIP_TIMING=1: one JSON line on stderr when a command ends -- wall seconds of its phases (tools/cli_bench.py reads it).
struct PhaseClock {
  std::map<string, double> sec;
  std::mutex mu;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void add(const string& k, double dt) { std::lock_guard<std::mutex> lk(mu); sec[k] += dt; }
  template <class F> void time(const string& k, F&& f) { const double a = now(); f(); add(k, now() - a); }
  void report(const char* cmd) {
    if (!getenv("QUAFF_HIP_TIMING")) return;
    std::ostringstream o;
    o << "{\"quaff_hip_timing\": \"" << cmd << "\", \"wall_s\": " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (const auto& kv : sec) o << ", \"" << kv.first << "_s\": " << kv.second;
    o << ", \"device_alloc_s\": " << qf_debug_alloc_ms() * 1e-3;
    o << "}";
    std::cerr << o.str() << std::endl;
  }
};

static void packSeqs(const vector<FastSeq>& v, size_t lo, size_t hi, string& seq, string& qual, vector<uint64_t>& off, bool& allQual) {
  seq.clear(); qual.clear(); off.assign(1, 0);
  allQual = true;
  for (size_t n = lo; n < hi; ++n) allQual = allQual && v[n].hasQual() && !v[n].qual.empty();
  for (size_t n = lo; n < hi; ++n) {
    seq.append(v[n].seq.data(), v[n].seq.size());
    if (allQual) qual.append(v[n].qual.data(), v[n].qual.size());
    off.push_back(seq.size());
  }
}

struct Session {
  qf_ctx* ctx = nullptr;       // ctxs[0]
  vector<qf_ctx*> ctxs;        // one per device; models and references are replicated, read batches are spread
  Params params;
  NullParams null;
  explicit Session(const Opts& o) {
    vector<int> dev;
    if (const char* e = getenv("QUAFF_HIP_DEVICES")) {   // explicit ids, e.g. "0,2,3" (an id may repeat: two contexts on one GPU)
      std::istringstream in(e);
      for (string f; std::getline(in, f, ',');) if (f.size()) dev.push_back(atoi(f.c_str()));
    } else {
      const int have = qf_device_count();
      const int n = o.gpus < 0 ? have : o.gpus;
      Require(n >= 1 && n <= std::max(have, 1), "-gpus " + to_string(n) + ": only " + to_string(have) + " HIP device(s) visible");
      for (int d = 0; d < n; ++d) dev.push_back(d);
    }
    Require(!dev.empty(), "no device selected");
    for (int d : dev) {
      qf_ctx* c = nullptr;
      if (qf_ctx_create(d, &c) != QF_OK) Fail(string("libquaffhip: ") + qf_last_error(nullptr));
      ctxs.push_back(c);
    }
    ctx = ctxs[0];
    // E-step sums of `train` / `count` go over RCCL when every context has a GPU of its own (one rank per GPU); several
    // contexts on one GPU (QUAFF_HIP_DEVICES=0,0: a test arrangement) are summed by the host instead
    bool distinct = ctxs.size() > 1;
    for (size_t a = 0; a < dev.size(); ++a) for (size_t b = 0; b < a; ++b) distinct = distinct && dev[a] != dev[b];
    if (distinct && o.wantComm) {
      if (qf_comm_init_all(ctxs.data(), (int)ctxs.size()) != QF_OK) Fail(string("libquaffhip: ") + qf_last_error(ctxs[0]));
      rccl = true;
    }
  }
  bool rccl = false;
  ~Session() { for (qf_ctx* c : ctxs) qf_ctx_destroy(c); }
  size_t devices() const { return ctxs.size(); }
  // the printer's -threshold, applied on the device before the traceback (the printer still checks it)
  void setThreshold(double t) { for (qf_ctx* c : ctxs) QF(c, qf_set_score_threshold(c, t)); }
  void loadParams(const Opts& o) {  // requireParamsOrUseDefaults, t/quaff.cpp:362-368
    const string text = o.paramsFile.size() ? slurp(o.paramsFile) : string(kDefaultParamsJson);
    Json j;
    string err;
    if (!parse_json(text, j, err) || !params.read_json(j, err)) Fail("Couldn't read parameters: " + err);
    for (qf_ctx* c : ctxs) QF(c, qf_set_params_json(c, text.c_str()));
  }
  void setParams(const Params& p) {
    params = p;
    vector<double> ipqr(12), mpqr((size_t)4 * p.Km() * 3);
    for (int i = 0; i < 4; ++i) { ipqr[i * 3] = p.insert[i].p; ipqr[i * 3 + 1] = p.insert[i].q; ipqr[i * 3 + 2] = p.insert[i].r; }
    for (size_t m = 0; m < p.match.size(); ++m) { mpqr[m * 3] = p.match[m].p; mpqr[m * 3 + 1] = p.match[m].q; mpqr[m * 3 + 2] = p.match[m].r; }
    for (qf_ctx* c : ctxs)
      QF(c, qf_set_params_raw(c, p.match_len, p.gap_len, p.refBase, p.beginInsert.data(), p.beginDelete.data(), p.extendInsert,
                              p.extendDelete, ipqr.data(), mpqr.data()));
  }
  void loadNull(const Opts& o, const vector<FastSeq>& reads) {  // requireNullModelOrFit, t/quaff.cpp:419-429
    if (o.nullFile.size()) {
      const string text = slurp(o.nullFile);
      Json j;
      string err;
      if (!parse_json(text, j, err) || !null.read_json(j, err)) Fail("Couldn't read null model parameters: " + err);
      for (qf_ctx* c : ctxs) QF(c, qf_set_null_json(c, text.c_str()));
    } else {
      vector<string> s, q;
      for (const auto& r : reads) { s.push_back(r.seq.str()); q.push_back(r.qual.str()); }
      null = fit_null(s, q);
      double pqr[12];
      for (int i = 0; i < 4; ++i) { pqr[i * 3] = null.null[i].p; pqr[i * 3 + 1] = null.null[i].q; pqr[i * 3 + 2] = null.null[i].r; }
      for (qf_ctx* c : ctxs) QF(c, qf_set_null_raw(c, null.nullEmit, pqr));
    }
    if (o.saveNull.size()) { ofstream out(o.saveNull); out << null.write_json(); }
  }
  void setRefs(const vector<FastSeq>& x) {
    string s;
    vector<uint64_t> off(1, 0);
    for (const auto& fs : x) { s.append(fs.seq.data(), fs.seq.size()); off.push_back(s.size()); }
    for (qf_ctx* c : ctxs) QF(c, qf_set_refs(c, s.data(), off.data(), (uint32_t)x.size()));
  }
};

// The device contexts (HIP start-up, streams, the library's tables: ~0.15 s) created while the main thread reads the input.
struct EarlySession {
  std::shared_future<void> done;
  std::unique_ptr<Session> s;
  explicit EarlySession(const Opts& o) {
    done = std::async(std::launch::async, [this, &o] { t_in_background = true; s.reset(new Session(o)); }).share();
    g_background = &done;
  }
  Session& get() { done.get(); g_background = nullptr; return *s; }
  ~EarlySession() { if (done.valid()) done.wait(); g_background = nullptr; }
};


// a read-to-reference alignment as `quaff align` labels it (QuaffViterbiMatrix::alignment, src/qmodel.cpp:1623-1645)
static Hit makeAlignment(const FastSeq& x, const FastSeq& y, const qf_alignment& al, const uint32_t* runs, bool local) {
  Hit h;
  h.src[0] = &x; h.src[1] = &y;
  h.lo[0] = al.x_start; h.lo[1] = 1;
  h.runs.assign(runs, runs + al.n_runs);
  h.label[0] = "Ref";
  h.note[0] = local ? "substr(" + x.name + "," + to_string(al.x_start) + ".." + to_string(al.x_end) + ")" : x.name;
  h.label[1] = "Read";
  h.note[1] = y.name;
  Coords cx, cy;
  cx.name = x.name; cx.start = al.x_start; cx.end = al.x_end;
  cy.name = y.name; cy.start = 1; cy.end = (unsigned)y.seq.size();
  h.origin[0] = cx.compose(x.source);
  h.origin[1] = cy.compose(y.source);
  h.score = al.score;
  return h;
}

static int cmdAlign(Opts& o) {
  Printer pr;
  bool printAll = false;
  o.implicit = {"-ref", "-read"};
  o.cfg.kmer_threshold = 20;  // DEFAULT_REFSEQ_KMER_THRESHOLD, t/quaff.cpp:128
  while (o.parseLog() || [&] { if (!o.av.empty() && o.av[0] == "-printall") { printAll = true; o.av.pop_front(); return true; } return false; }() ||
         pr.parse(o.av) || o.parseConfig(true) || o.parseFiles(true) ||
         [&] { if (!o.av.empty() && o.av[0] == "-noquals") { o.noquals = true; o.av.pop_front(); return true; } return false; }() ||
         o.parseUnknown()) {}
  o.finishConfig();
  PhaseClock clk;
  EarlySession early(o);
  SeqSet reads, refs;
  clk.time("parse", [&] {
    reads.load(o.readFiles, "read", "-read", !o.noquals, false, true);
    refs.load(o.refFiles, "reference", "-ref", false, !o.fwdstrand, true);
  });
  Session& s = early.get();
  s.loadParams(o);
  clk.time("null_model", [&] { s.loadNull(o, reads.seqs); });
  s.setRefs(refs.seqs);
  s.setThreshold(pr.threshold);
  pr.header(cout, refs.seqs, false);
  // contiguous blocks of reads, one per device at a time; printed in read order whatever the device count.  A round's
  // alignments are written by a thread of their own while the next round is on the devices (the reference's printer is one
  // mutex-guarded stream too, src/qmodel.cpp:2570-2600, fed by its worker threads).
  const size_t G = s.devices(), n = reads.seqs.size();
  const size_t perCall = max<size_t>(1, ((size_t)1 << 28) / max<size_t>(1, refs.seqs.size()));   // the library takes 2^28 pairs per call
  const size_t batch = max<size_t>(1, min<size_t>({(size_t)(n > 4 * 65536 * G ? 65536 : 16384), perCall, (n + G - 1) / G}));
  std::future<void> writer;
  for (size_t lo0 = 0; lo0 < n; lo0 += batch * G) {
    const size_t nrun = min(G, (n - lo0 + batch - 1) / batch);
    auto got = std::make_shared<vector<vector<Hit>>>(nrun);
    onDevices(nrun, [&](size_t k) {
      const size_t lo = lo0 + k * batch, hi = min(n, lo + batch);
      qf_ctx* c = s.ctxs[k];
      string seq, qual;
      vector<uint64_t> off;
      bool allQual;
      double t = PhaseClock::now();
      packSeqs(reads.seqs, lo, hi, seq, qual, off, allQual);
      if (!k) clk.add("pack", PhaseClock::now() - t);
      qf_align_result res;
      t = PhaseClock::now();
      QFT(c, qf_align_batch(c, &o.cfg, seq.data(), allQual ? qual.data() : nullptr, off.data(), (uint32_t)(hi - lo),
                            printAll ? QF_ALIGN_ALL : QF_ALIGN_BEST, &res));
      if (!k) { clk.add("device_call", PhaseClock::now() - t); clk.add("device_ms_reported", res.ms_total * 1e-3); }
      t = PhaseClock::now();
      for (uint32_t a = 0; a < res.n_alignments; ++a) {
        const qf_alignment& al = res.alignments[a];
        (*got)[k].push_back(makeAlignment(refs.seqs[al.ref], reads.seqs[lo + al.read], al, res.cigar_runs + al.run_offset, o.cfg.local));
      }
      if (!k) clk.add("collect", PhaseClock::now() - t);
    });
    if (writer.valid()) writer.get();
    writer = std::async(std::launch::async, [&pr, &clk, got] {
      const double t = PhaseClock::now();
      pr.writeAll(cout, *got);
      clk.add("write", PhaseClock::now() - t);
    });
  }
  if (writer.valid()) writer.get();
  cout.flush();
  clk.report("align");
  return EXIT_SUCCESS;
}

// One E-step over all reads (QuaffTrainer::getCounts, src/qmodel.cpp:2005-2032), batched.
static ParamCounts eStep(Session& s, Opts& o, const SeqSet& reads, uint32_t n_refs, bool useNull, vector<vector<uint32_t>>& sortOrder, double& logLike) {
  ParamCounts total(s.params.match_len, s.params.gap_len);
  logLike = 0;
  // Read blocks spread over the devices, one round of blocks at a time.  The library returns every block's counts and
  // log-likelihood as 128-bit fixed-point words (qf_count_result.counts_exact: the device adds count terms as integers), which
  // add exactly: the host's sum over blocks -- and RCCL's over devices (qf_allreduce_counts_exact) -- is the same whatever the
  // block size, the number of devices and the order of arrival; one conversion to double at the end.  (The reference adds
  // per-read counts in read order, src/qmodel.cpp:2416-2422: one fixed rounding sequence; this is another.)
  const size_t G = s.devices(), n = reads.seqs.size();
  const size_t perCall = max<size_t>(1, ((size_t)1 << 28) / max<size_t>(1, (size_t)n_refs));   // the library takes 2^28 pairs per call
  const size_t batch = max<size_t>(1, min<size_t>({(size_t)16384, perCall, (n + G - 1) / G}));
  const bool haveOrder = !sortOrder.empty();
  if (!haveOrder) sortOrder.assign(n, vector<uint32_t>());
  const size_t nv = total.v.size() + 1;             // the counts, then the log-likelihood
  vector<uint64_t> totalFx(2 * nv, 0);
  for (size_t lo0 = 0; lo0 < n; lo0 += batch * G) {
    const size_t nrun = min(G, (n - lo0 + batch - 1) / batch);
    const size_t nact = s.rccl ? G : nrun;   // an all-reduce needs every rank, also those without a block in the last round
    vector<vector<uint64_t>> fx(nact);
    onDevices(nact, [&](size_t k) {
      qf_ctx* c = s.ctxs[k];
      string failure;
      auto countBlock = [&]() {
        if (k >= nrun) return;
        const size_t lo = lo0 + k * batch, hi = min(n, lo + batch);
        string seq, qual;
        vector<uint64_t> off;
        bool allQual;
        packSeqs(reads.seqs, lo, hi, seq, qual, off, allQual);
        QFT(c, qf_upload_reads(c, seq.data(), allQual ? qual.data() : nullptr, off.data(), (uint32_t)(hi - lo)));
        vector<uint32_t> sin, snin;
        if (haveOrder) {
          for (size_t r = lo; r < hi; ++r) {
            vector<uint32_t> row = sortOrder[r];
            snin.push_back((uint32_t)row.size());
            row.resize(n_refs, 0);
            sin.insert(sin.end(), row.begin(), row.end());
          }
        }
        qf_count_result res;
        QFT(c, qf_count_resident(c, &o.cfg, useNull ? 0 : QF_COUNT_FORCE, sin.empty() ? nullptr : sin.data(),
                                 snin.empty() ? nullptr : snin.data(), &res));
        if ((size_t)res.counts_size + 1 != nv) throw std::runtime_error("unexpected count vector size");
        fx[k].assign(res.counts_exact, res.counts_exact + 2 * (size_t)res.counts_size);
        fx[k].push_back(res.loglike_exact[0]);
        fx[k].push_back(res.loglike_exact[1]);
        for (size_t r = lo; r < hi; ++r) {   // each block owns its reads' rows
          vector<uint32_t>& so = sortOrder[r];
          so.clear();
          for (uint32_t q = 0; q < res.sort_count[r - lo]; ++q) so.push_back(res.sort_order[(r - lo) * res.n_refs + q]);
        }
      };
      if (!s.rccl) { countBlock(); return; }
      // QuaffCountingScheduler::finalCounts / finalLogLike (src/qmodel.cpp:2416-2422) across the devices: one RCCL
      // all-reduce per round.  It is a collective: a device whose block failed (or that has no block in the last round)
      // still takes part, with zeros, and reports afterwards - the others must not be left waiting for it (and if it cannot
      // even do that, the library's deadline releases them with an error).
      try { countBlock(); } catch (const std::exception& e) { failure = e.what(); }
      if (fx[k].size() != 2 * nv || failure.size()) fx[k].assign(2 * nv, 0);
      QFT(c, qf_allreduce_counts_exact(c, fx[k].data(), (uint32_t)nv));
      if (failure.size()) throw std::runtime_error(failure);
    });
    for (size_t k = 0; k < (s.rccl ? (size_t)1 : nrun); ++k) qf_exact_add(totalFx.data(), fx[k].data(), (uint32_t)nv);
  }
  vector<double> vals(nv);
  qf_exact_to_double(totalFx.data(), (uint32_t)nv, vals.data());
  std::copy(vals.begin(), vals.end() - 1, total.v.begin());
  logLike = vals.back();
  return total;
}

static bool parseTrainArgs(Opts& o, bool training, int& maxIter, double& minInc, long& maxReadBases, bool& allowNull,
                           string& rawCounts, string& countsWithPrior, string& saveParams) {  // src/qmodel.cpp:1916-1993
  if (o.av.empty()) return false;
  const string arg = o.av[0];
  auto val = [&]() { Require(o.av.size() > 1, arg + " must have an argument"); const string v = o.av[1]; o.av.pop_front(); o.av.pop_front(); return v; };
  if (training && arg == "-maxiter") { maxIter = atoi(val().c_str()); return true; }
  if (training && arg == "-mininc") { minInc = atof(val().c_str()); return true; }
  if (training && arg == "-maxreadmb") { maxReadBases = atol(val().c_str()) << 20; return true; }
  if (arg == "-force") { allowNull = false; o.av.pop_front(); return true; }
  if (arg == "-savecounts") { rawCounts = val(); return true; }
  if (training && arg == "-savecountswithprior") { countsWithPrior = val(); return true; }
  if (training && arg == "-saveparams") { saveParams = val(); return true; }
  return false;
}

static void fitRefSeqs(Params& qp, const vector<FastSeq>& refs) {  // src/qmodel.cpp:284-294; the reference reads an
  long total = 0;                                                   // uninitialised totalLen there
  long cnt[4] = {0, 0, 0, 0};
  for (const auto& fs : refs) { total += fs.seq.size(); for (char c : fs.seq) { const int u = toupper(c); ++cnt[u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : 3]; } }
  for (int i = 0; i < 4; ++i) qp.refBase[i] = cnt[i] / (double)total;
}

static int cmdTrainOrCount(Opts& o, bool training) {
  int maxIter = 100;  // QuaffMaxEMIterations
  double minInc = .01;
  long maxReadBases = 0;
  bool allowNull = true;
  string rawCounts, countsWithPriorFile, saveParams, priorFile, savePrior;
  int order = -1, suborder = -1, gaporder = -1;
  o.implicit = {"-ref", "-read"};
  o.cfg.kmer_threshold = 20;
  auto parsePrior = [&]() {  // QuaffPriorIn::parsePriorArgs, t/quaff.cpp:431-480
    if (!training || o.av.empty()) return false;
    const string arg = o.av[0];
    auto val = [&]() { Require(o.av.size() > 1, arg + " needs an argument"); const string v = o.av[1]; o.av.pop_front(); o.av.pop_front(); return v; };
    if (arg == "-prior") { priorFile = val(); return true; }
    if (arg == "-order") { order = atoi(val().c_str()); return true; }
    if (arg == "-suborder") { suborder = atoi(val().c_str()); return true; }
    if (arg == "-gaporder") { gaporder = atoi(val().c_str()); return true; }
    if (arg == "-saveprior") { savePrior = val(); return true; }
    return false;
  };
  while (o.parseLog() || parseTrainArgs(o, training, maxIter, minInc, maxReadBases, allowNull, rawCounts, countsWithPriorFile, saveParams) ||
         o.parseConfig(true) || o.parseFiles(true) || parsePrior() || o.parseUnknown()) {}
  o.finishConfig();
  o.wantComm = true;
  EarlySession early(o);
  SeqSet reads, refs;
  reads.load(o.readFiles, "read", "-read", true, false, false);
  refs.load(o.refFiles, "reference", "-ref", false, !o.fwdstrand, false);
  Session& s = early.get();
  s.loadNull(o, reads.seqs);
  if (!training) {
    s.loadParams(o);
    s.setRefs(refs.seqs);
    vector<vector<uint32_t>> so;
    double ll;
    ParamCounts counts = eStep(s, o, reads, (uint32_t)refs.seqs.size(), true, so, ll);
    if (rawCounts.size()) { ofstream out(rawCounts); out << counts.write_json() << endl; }
    else cout << counts.write_json();
    return EXIT_SUCCESS;
  }
  // prior: file, or auto from the null model (requirePriorOrUseNullModel, t/quaff.cpp:490-515)
  unsigned ml = 1, gl = 0;
  bool lenSpecified = false;
  if (order >= 0) { gl = order; ml = 1 + order; lenSpecified = true; }
  if (suborder >= 0) { ml = 1 + suborder; lenSpecified = true; }
  if (gaporder >= 0) { gl = gaporder; lenSpecified = true; }
  Params seed;
  const bool haveParams = o.paramsFile.size() > 0;
  if (haveParams) { s.loadParams(o); seed = s.params; }
  ParamCounts prior(ml, gl);
  if (priorFile.size()) {
    Json j;
    string err;
    if (!parse_json(slurp(priorFile), j, err) || !prior.read_json(j, err)) Fail("Couldn't read counts: " + err);
    if (haveParams) Require(prior.match_len == seed.match_len && prior.gap_len == seed.gap_len, "Order of dependence in prior file does not match order in parameter file");
  } else {
    if (haveParams) {
      if (lenSpecified) Require(ml == seed.match_len && gl == seed.gap_len, "Order of dependence specified on command line does not match order in parameter file");
      else prior.resize(seed.match_len, seed.gap_len);
    }
    prior.init_counts(9, 9, 5, 1, &s.null);
  }
  if (savePrior.size()) { ofstream out(savePrior); out << prior.write_json(); }
  if (!haveParams) seed = prior.fit();  // requireParamsOrUsePrior, t/quaff.cpp:370-376
  // QuaffTrainer::fit / fitUnlimited, src/qmodel.cpp:2169-2231
  SeqSet used = reads;
  if (maxReadBases > 0) {
    used.seqs.clear();
    long bases = 0;
    for (const auto& y : reads.seqs) { used.seqs.push_back(y); bases += y.seq.size(); if (bases >= maxReadBases) break; }
  }
  Params qp = seed;
  s.setRefs(refs.seqs);
  vector<vector<uint32_t>> sortOrder;
  double prevLL = -INFINITY;
  for (int iter = 0; iter < maxIter; ++iter) {
    s.setParams(qp);
    double logLike = 0;
    ParamCounts counts = eStep(s, o, used, (uint32_t)refs.seqs.size(), allowNull, sortOrder, logLike);
    if (rawCounts.size()) { ofstream out(rawCounts); out << counts.write_json() << endl; }
    const double logPrior = prior.log_prior(qp);
    const double llp = logLike + logPrior;
    cerr << "EM iteration " << (iter + 1) << ": log-likelihood (" << fmt6(logLike) << ") + log-prior (" << fmt6(logPrior) << ") = " << fmt6(llp) << endl;
    if (em_converged(iter, llp, prevLL, minInc)) break;
    prevLL = llp;
    ParamCounts withPrior = counts;
    withPrior.add_weighted(prior, 1.);
    if (countsWithPriorFile.size()) { ofstream out(countsWithPriorFile); out << withPrior.write_json() << endl; }
    qp = withPrior.fit();
    fitRefSeqs(qp, refs.seqs);
    if (saveParams.size()) { ofstream out(saveParams); out << qp.write_json() << endl; }
  }
  if (saveParams.empty()) cout << qp.write_json();
  return EXIT_SUCCESS;
}

// An overlap alignment as `quaff overlap` prints it.  The reference re-pairs every stretch of gap states between two
// match stretches when it writes the rows (src/qoverlap.cpp:231-267): of `nd` x-only and `ni` y-only columns, min(nd, ni)
// become ordinary two-base columns (x's bases in order against y's), and only the excess stays gapped, after them.
static Hit makeOverlapAlignment(const FastSeq& x, const FastSeq& y, const qf_overlap_alignment& al, const uint32_t* runs) {
  Hit h;
  h.src[0] = &x; h.src[1] = &y;
  h.lo[0] = al.x_start; h.lo[1] = al.y_start;
  for (uint32_t r = 0; r < al.n_runs;) {
    if ((runs[r] & 3u) == 0) { appendRun(h.runs, 0, runs[r] >> 2); ++r; continue; }
    size_t only[3] = {0, 0, 0};   // [1] y-only (insert states), [2] x-only (delete states)
    for (; r < al.n_runs && (runs[r] & 3u) != 0; ++r) only[runs[r] & 3u] += runs[r] >> 2;
    const size_t paired = min(only[1], only[2]);
    appendRun(h.runs, 0, paired);
    appendRun(h.runs, 2, only[2] - paired);
    appendRun(h.runs, 1, only[1] - paired);
  }
  h.label[0] = "read_x";
  h.note[0] = "substr(" + x.name + "," + to_string(al.x_start) + ".." + to_string(al.x_end) + ")";
  h.label[1] = "read_y";
  h.note[1] = "substr(" + y.name + "," + to_string(al.y_start) + ".." + to_string(al.y_end) + ")";
  Coords cx, cy;
  cx.name = x.name; cx.start = al.x_start; cx.end = al.x_end;
  cy.name = y.name; cy.start = al.y_start; cy.end = al.y_end;
  h.origin[0] = cx.compose(x.source);
  h.origin[1] = cy.compose(y.source);
  h.score = al.score;
  return h;
}

static int cmdOverlap(Opts& o) {
  Printer pr;
  o.implicit = {"-read"};
  while (o.parseLog() || pr.parse(o.av) || o.parseConfig(false) || o.parseFiles(false) ||
         [&] { if (!o.av.empty() && o.av[0] == "-noquals") { o.noquals = true; o.av.pop_front(); return true; } return false; }() ||
         o.parseUnknown()) {}
  o.finishConfig();
  PhaseClock clk;
  EarlySession early(o);
  SeqSet reads;
  clk.time("parse", [&] { reads.load(o.readFiles, "read", "-read", !o.noquals, !o.fwdstrand, true); });
  Session& s = early.get();
  s.loadParams(o);
  clk.time("null_model", [&] { s.loadNull(o, reads.seqs); });
  s.setThreshold(pr.threshold);
  pr.header(cout, reads.seqs, true);
  string seq, qual;
  vector<uint64_t> off;
  bool allQual;
  clk.time("pack", [&] { packSeqs(reads.seqs, 0, reads.seqs.size(), seq, qual, off, allQual); });
  clk.time("upload", [&] {
    for (qf_ctx* c : s.ctxs) QF(c, qf_upload_reads(c, seq.data(), allQual ? qual.data() : nullptr, off.data(), (uint32_t)reads.seqs.size()));
  });
  // QuaffOverlapScheduler's enumeration (src/qoverlap.cpp:475-480,528-547: nx = 0 ... nOriginals - 2, ny = nx + 1 ... over the
  // originals and then their reverse complements) happens on the device: qf_overlap_rows takes a block of rows, generates its
  // (nx, ny, yComplemented) triples, applies the printer's threshold and returns the alignments that pass, in the
  // scheduler's order.  Every device holds all reads; consecutive row blocks go to the devices in turn and are printed in
  // block order, by a thread of their own while the next blocks are on the devices.
  const size_t N = reads.nOriginals, total = reads.seqs.size(), G = s.devices();
  size_t chunk = (size_t)1 << 26;   // pairs per device call (the library cuts a call into blocks that fit its tables)
  if (const char* e = getenv("QUAFF_HIP_OVERLAP_CHUNK")) chunk = max<size_t>(1, (size_t)atol(e));   // tests: many small blocks
  struct Block { uint32_t x0, x1; };
  vector<Block> blocks;
  for (size_t nx = 0, acc = 0, b0 = 0; nx + 1 < N; ++nx) {
    acc += total - 1 - nx;
    if (acc >= chunk || nx + 2 == N) { blocks.push_back({(uint32_t)b0, (uint32_t)(nx + 1)}); b0 = nx + 1; acc = 0; }
  }
  std::future<void> writer;
  for (size_t r0 = 0; r0 < blocks.size(); r0 += G) {
    const size_t nrun = min(G, blocks.size() - r0);
    auto got = std::make_shared<vector<vector<Hit>>>(nrun);
    onDevices(nrun, [&](size_t k) {
      qf_ctx* c = s.ctxs[k];
      qf_overlap_rows_result res;
      double t = PhaseClock::now();
      QFT(c, qf_overlap_rows(c, &o.cfg, (uint32_t)N, blocks[r0 + k].x0, blocks[r0 + k].x1, &res));
      if (!k) {
        clk.add("device_call", PhaseClock::now() - t);
        clk.add("device_ms_reported", res.ms_total * 1e-3);
        clk.add("device_seed", res.ms_seed * 1e-3);
        clk.add("device_fill", res.ms_fill * 1e-3);
        clk.add("device_traceback", res.ms_traceback * 1e-3);
      }
      t = PhaseClock::now();
      // (the hits become alignments on several threads when only one device is at work: 160 k of them were 0.35 s on one)
      vector<Hit>& mine = (*got)[k];
      mine.resize(res.n_hits);
      auto convert = [&](uint32_t lo, uint32_t hi) {
        for (uint32_t a = lo; a < hi; ++a) {
          const qf_overlap_hit& h = res.hits[a];
          qf_overlap_alignment al;
          al.pair = 0;
          al.viterbi = h.viterbi; al.score = h.score;
          al.x_start = h.x_start; al.x_end = h.x_end; al.y_start = h.y_start; al.y_end = h.y_end;
          al.n_columns = h.n_columns; al.n_runs = h.n_runs; al.run_offset = h.run_offset;
          mine[a] = makeOverlapAlignment(reads.seqs[h.x], reads.seqs[h.y], al, res.state_runs + h.run_offset);
        }
      };
      const uint32_t T = (uint32_t)std::max<size_t>(1, std::min<size_t>({(size_t)8 / nrun, (size_t)std::thread::hardware_concurrency(), (size_t)res.n_hits / 2048 + 1}));
      vector<std::thread> helpers;
      for (uint32_t q = 1; q < T; ++q) helpers.emplace_back(convert, (uint32_t)((uint64_t)res.n_hits * q / T), (uint32_t)((uint64_t)res.n_hits * (q + 1) / T));
      convert(0, (uint32_t)((uint64_t)res.n_hits / T));
      for (auto& th : helpers) th.join();
      if (!k) clk.add("collect", PhaseClock::now() - t);
    });
    if (writer.valid()) writer.get();
    writer = std::async(std::launch::async, [&pr, &clk, got] {
      const double t = PhaseClock::now();
      pr.writeAll(cout, *got);
      clk.add("write", PhaseClock::now() - t);
    });
  }
  if (writer.valid()) writer.get();
  cout.flush();
  clk.report("overlap");
  return EXIT_SUCCESS;
}

// `quaff selftest <what> ...`: the host-side unit-test programs of the reference (t/testfasta.cpp, t/testfastq.cpp,
// t/testquaffjsonio.cpp, t/testquaffnulljsonio.cpp, t/testquaffcountsjsonio.cpp, t/testnegbinom.cpp; Makefile:103-133) over
// this build's readers, writers and fitter.  No device is touched.
static int cmdSelfTest(deque<string>& av) {
  Require(!av.empty(), "selftest needs a name: fasta fastq params null counts fit initcounts logprior converge negbinom");
  const string what = av[0];
  av.pop_front();
  auto parsed = [&](const string& file) { Json j; string err; if (!parse_json(slurp(file), j, err)) Fail("Couldn't parse " + file + ": " + err); return j; };
  if (what == "fasta" || what == "fastq") {
    Require(av.size() == 1, "selftest " + what + " <seqs>");
    const auto t0 = std::chrono::steady_clock::now();
    const vector<FastSeq> all = readFastSeqs(av[0]);
    if (getenv("QUAFF_HIP_TIMING"))
      cerr << "{\"quaff_hip_timing\": \"selftest\", \"parse_s\": " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << "}" << endl;
    for (const FastSeq& fs : all) {
      if (what == "fasta") writeFasta(cout, fs);
      else {   // FastSeq::writeFastq, src/fastseq.cpp:119-127
        cout << '@' << fs.name;
        if (fs.comment.size()) cout << ' ' << fs.comment;
        cout << endl << fs.seq << endl;
        if (fs.hasQual()) cout << '+' << endl << fs.qual << endl;
      }
    }
    return EXIT_SUCCESS;
  }
  if (what == "params" || what == "null" || what == "counts") {
    Require(av.size() == 1, "selftest " + what + " <file.json>");
    const Json j = parsed(av[0]);
    string err;
    if (what == "params") { Params p; if (!p.read_json(j, err)) Fail(err); cout << p.write_json(); }
    else if (what == "null") { NullParams p; if (!p.read_json(j, err)) Fail(err); cout << p.write_json(); }
    else { ParamCounts c(1, 0); if (!c.read_json(j, err)) Fail(err); cout << c.write_json(); }
    return EXIT_SUCCESS;
  }
  if (what == "fit") {   // the M-step alone: QuaffParamCounts::fit, src/qmodel.cpp:1731-1768
    Require(av.size() == 1, "selftest fit <counts.json>");
    ParamCounts c(1, 0);
    string err;
    if (!c.read_json(parsed(av[0]), err)) Fail(err);
    cout << c.fit().write_json();
    return EXIT_SUCCESS;
  }
  if (what == "initcounts") {   // QuaffParamCounts::initCounts (src/qmodel.cpp:431-456); the auto-prior is 9 9 5 1 + the null model
    Require(av.size() == 6 || av.size() == 7, "selftest initcounts matchLen gapLen noBegin yesExtend matchIdent other [null.json]");
    ParamCounts c((unsigned)atoi(av[0].c_str()), (unsigned)atoi(av[1].c_str()));
    NullParams null;
    string err;
    if (av.size() == 7 && !null.read_json(parsed(av[6]), err)) Fail(err);
    c.init_counts(atof(av[2].c_str()), atof(av[3].c_str()), atof(av[4].c_str()), atof(av[5].c_str()), av.size() == 7 ? &null : nullptr);
    cout << c.write_json();
    return EXIT_SUCCESS;
  }
  if (what == "logprior") {   // QuaffParamCounts::logPrior (src/qmodel.cpp:1681-1710), full precision
    Require(av.size() == 2, "selftest logprior <pseudocounts.json> <params.json>");
    ParamCounts c(1, 0);
    Params p;
    string err;
    if (!c.read_json(parsed(av[0]), err)) Fail(err);
    if (!p.read_json(parsed(av[1]), err)) Fail(err);
    cout << setprecision(17) << c.log_prior(p) << ' ' << c.expected_log_like(p) << endl;
    return EXIT_SUCCESS;
  }
  if (what == "converge") {   // the EM stopping rule (src/qmodel.cpp:2204-2206) over a series of logLike + logPrior values
    Require(av.size() >= 2, "selftest converge minInc v1 v2 ...");
    const double minInc = atof(av[0].c_str());
    double prev = -INFINITY;
    int iter = 0;
    for (; iter + 1 < (int)av.size(); ++iter) {
      const double v = atof(av[iter + 1].c_str());
      if (em_converged(iter, v, prev, minInc)) break;
      prev = v;
    }
    cout << iter << endl;      // E-steps run before the loop stopped (all of them if it never did)
    return EXIT_SUCCESS;
  }
  if (what == "negbinom") {   // t/testnegbinom.cpp with the exact expected frequencies instead of GSL's sampler
    Require(av.size() == 4, "selftest negbinom pSuccess nFail nSamples relativeError");
    const double p = atof(av[0].c_str()), r = atof(av[1].c_str()), N = atof(av[2].c_str()), eps = atof(av[3].c_str());
    vector<double> kFreq;
    for (int k = 0; k < 2000; ++k) {
      const double f = N * exp(lgamma(k + r) - lgamma(r) - lgamma(k + 1.) + r * log(p) + k * log1p(-p));
      if (k > r / p && f < 1e-9) break;
      kFreq.push_back(f);
    }
    double pFit = 0, rFit = 0;
    const int status = fit_negbinom(kFreq, pFit, rFit);
    const bool ok = status == 0 && fabs(pFit - p) < eps * fabs(p) && fabs(rFit - r) < eps * fabs(r);   // gsl_root_test_delta(x1, x0, 0, eps)
    cout << (ok ? "ok" : "not ok") << ": (" << pFit << ',' << rFit << ") " << (ok ? "~=" : "!=") << " (" << p << ',' << r << ')' << endl;
    return EXIT_SUCCESS;
  }
  if (what == "fitvec") {   // fit_negbinom on a literal count vector (degenerate inputs): prints status p r
    vector<double> kFreq;
    for (const auto& v : av) kFreq.push_back(atof(v.c_str()));
    double pFit = 0, rFit = 0;
    const int status = fit_negbinom(kFreq, pFit, rFit);
    cout << status << ' ' << setprecision(17) << pFit << ' ' << rFit << endl;
    return EXIT_SUCCESS;
  }
  Fail("Unknown selftest: " + what);
  return EXIT_FAILURE;
}

int main(int argc, char** argv) {
  Opts o;
  for (int n = 1; n < argc; ++n) o.av.push_back(argv[n]);
  if (o.av.empty()) { cerr << "Usage: quaff {help,train,align,overlap} [options]" << endl; return EXIT_FAILURE; }
  const string command = o.av[0];
  o.av.pop_front();
  if (command == "align") return cmdAlign(o);
  if (command == "count") return cmdTrainOrCount(o, false);
  if (command == "train") return cmdTrainOrCount(o, true);
  if (command == "overlap") return cmdOverlap(o);
  if (command == "selftest") return cmdSelfTest(o.av);
  if (command == "help" || command == "-h" || command == "--help" || command == "-help") {
    cout << "Usage: quaff {help,train,align,overlap} [options]\n\n"
            " quaff train refs.fasta reads.fastq >params.json   (-maxiter -mininc -maxreadmb -force -order/-suborder/-gaporder\n"
            "                                                    -prior -saveprior -saveparams -savecounts -savecountswithprior)\n"
            " quaff align refs.fasta reads.fastq                (-printall)\n"
            " quaff overlap reads.fastq\n"
            " quaff count refs.fasta reads.fastq                (one E-step; -savecounts)\n\n"
            "Alignment options: -threshold <n> -nothreshold -noquals -savealign <file> -format {fasta,stockholm,sam,refseq}\n"
            "General: -params <file> -ref <file> -read <file> -fwdstrand -global -null <file> -savenull <file>\n"
            "         -kmatch <k> -kmatchn <n> -kmatchband <n> -kmatchmb <M> -kmatchmax -kmatchoff\n"
            "         -gpus <n|all>   spread the read batches over n GPUs (default 1; results do not depend on n)\n"
            "All dynamic programming runs on the GPU(s) through libquaffhip (include/quaff_hip.h).\n";
    return EXIT_SUCCESS;
  }
  if (command == "version" || command == "-V" || command == "--version") { cout << "quaff (hip) 0.1" << endl; return EXIT_SUCCESS; }
  cerr << "Usage: quaff {help,train,align,overlap} [options]\nUnrecognized command: " << command << endl;
  return EXIT_FAILURE;
}
