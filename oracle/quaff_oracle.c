/*
 * oracle/quaff_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the quaff hot path (k-mer
 * diagonal seeding + banded pair-HMM Viterbi / Forward / Backward).  It is the
 * checker that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * compare the HIP path against.  Nothing under quaff_amd/ may include, link or
 * call it.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - seeding / tokens / k-mers / log-sum-exp: checked bit-for-bit against the
 *     reference's own sources compiled into oracle/_ref (diagenv.cpp,
 *     fastseq.cpp, logsumexp.cpp, gason.cpp — none of which need GSL);
 *   - DP fills, traceback, counts, null model: the reference's qmodel.cpp and
 *     negbinom.cpp include GSL headers, GSL is not installed, so they are
 *     unbuildable here; these parts are pinned by the reference's own goldens
 *     data/c8f30-self-{align,counts}.json at the 6 significant figures those
 *     files carry (tests/test_oracle_goldens.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  Written from the behaviour, not copied: flat arrays, no
 * std::map, dense per-diagonal storage.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#define QO_NQUAL 94          /* FastSeq::qualScoreRange, src/fastseq.cpp:69 */
#define QO_NQ1   95          /* slot 94 = quality-marginalised logSymProb */
#define NEG_INF  (-INFINITY)

/* ------------------------------------------------------------------------ */
/* sequences: src/fastseq.cpp                                                */
/* ------------------------------------------------------------------------ */

/* tokenize(), src/fastseq.cpp:11-16; FastSeq::tokens :71-83.
 * Returns 0, or -(pos+1) for the first non-ACGT symbol (reference terminates). */
int qo_tokenize(const char *seq, int len, uint8_t *tok)
{
    for (int i = 0; i < len; ++i) {
        int c = toupper((unsigned char)seq[i]);
        int t = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
        if (t < 0) return -(i + 1);
        tok[i] = (uint8_t)t;
    }
    return 0;
}

/* FastSeq::qualScores / qualScoreForChar, src/fastseq.cpp:101-109, fastseq.h:56-58 */
void qo_quals(const char *qual, int len, uint8_t *q)
{
    for (int i = 0; i < len; ++i) {
        int v = (int)(signed char)qual[i] - '!';
        if (v < 0) v = 0;
        if (v > QO_NQUAL - 1) v = QO_NQUAL - 1;
        q[i] = (uint8_t)v;
    }
}

/* FastSeq::kmers, src/fastseq.cpp:85-99 (+ makeKmer :27-35): context k-mer
 * ENDING at each position, left-padded with k-1 copies of the most frequent
 * token (first maximum on ties); k == 0 -> all zero. */
void qo_kmers(const uint8_t *tok, int len, int k, uint32_t *out)
{
    if (k == 0) { for (int i = 0; i < len; ++i) out[i] = 0; return; }
    int count[4] = {0, 0, 0, 0};
    for (int i = 0; i < len; ++i) ++count[tok[i]];
    int best = 0;
    for (int t = 1; t < 4; ++t) if (count[t] > count[best]) best = t;
    for (int pos = 0; pos < len; ++pos) {
        uint32_t km = 0;
        for (int a = 0; a < k; ++a) {           /* padded index pos+a <-> seq index pos+a-(k-1) */
            int s = pos + a - (k - 1);
            uint32_t t = s < 0 ? (uint32_t)best : tok[s];
            km = km * 4 + t;
        }
        out[pos] = km;
    }
}

/* revcomp(), src/fastseq.cpp:209-216 on tokens: complement = 3 - tok (:18-20) */
void qo_revcomp_tok(const uint8_t *tok, int len, uint8_t *out)
{
    for (int i = 0; i < len; ++i) out[len - 1 - i] = (uint8_t)(3 - tok[i]);
}

/* ------------------------------------------------------------------------ */
/* gason's decimal parser: src/gason.cpp:73-117 (NOT correctly rounded)      */
/* ------------------------------------------------------------------------ */
double qo_gason_number(const char *s)
{
    char ch = *s;
    if (ch == '-') ++s;
    double result = 0;
    while (*s >= '0' && *s <= '9') result = (result * 10) + (*s++ - '0');
    if (*s == '.') {
        ++s;
        double fraction = 1;
        while (*s >= '0' && *s <= '9') { fraction *= 0.1; result += (*s++ - '0') * fraction; }
    }
    if (*s == 'e' || *s == 'E') {
        ++s;
        double base = 10;
        if (*s == '+') ++s;
        else if (*s == '-') { ++s; base = 0.1; }
        unsigned int exponent = 0;
        while (*s >= '0' && *s <= '9') exponent = (exponent * 10) + (unsigned)(*s++ - '0');
        double power = 1;
        for (; exponent; exponent >>= 1, base *= base) if (exponent & 1) power *= base;
        result *= power;
    }
    return ch == '-' ? -result : result;
}

/* ------------------------------------------------------------------------ */
/* log-sum-exp: src/logsumexp.cpp                                            */
/* ------------------------------------------------------------------------ */
#define LSE_MAX   10
#define LSE_PREC  .0001
#define LSE_N     (((int)(LSE_MAX / LSE_PREC)) + 1)      /* 100001, :9 */
static double *lse_table = 0;

/* LogSumExpLookupTable ctor, src/logsumexp.cpp:20-28; entry = log(1+exp(-x)) :105-107 */
const double *qo_lse_table(void)
{
    if (!lse_table) {
        double *t = (double *)malloc(sizeof(double) * (LSE_N + 1));
        for (int n = 0; n < LSE_N; ++n) { double x = n * LSE_PREC; t[n] = log(1. + exp(-x)); }
        t[LSE_N] = 0;
        lse_table = t;
    }
    return lse_table;
}
int qo_lse_table_size(void) { return LSE_N; }

/* log_sum_exp_unary, src/logsumexp.cpp:84-103 */
static inline double lse_unary(double x)
{
    if (x >= LSE_MAX || isnan(x) || isinf(x)) return 0;
    if (x < 0) return -x;
    int n = (int)(x / LSE_PREC);
    double dx = x - (n * LSE_PREC);
    double f0 = lse_table[n], f1 = lse_table[n + 1];
    double df = f1 - f0;
    return f0 + df * (dx / LSE_PREC);
}

/* log_sum_exp(a,b), src/logsumexp.cpp:34-50 */
double qo_lse(double a, double b)
{
    double mx, diff;
    if (!lse_table) qo_lse_table();
    if (a == b) { mx = a; diff = 0; }
    else if (a < b) { mx = b; diff = b - a; }
    else { mx = a; diff = a - b; }
    return mx + lse_unary(diff);
}
/* 3-argument form, src/logsumexp.cpp:52-54 */
static inline double lse3(double a, double b, double c) { return qo_lse(qo_lse(a, b), c); }

/* ------------------------------------------------------------------------ */
/* score tables: src/qmodel.cpp:87-93 (SymQualScores), :296-325 (QuaffScores) */
/* ------------------------------------------------------------------------ */

/* logNegativeBinomial, src/negbinom.cpp:30-32 = log(gsl_ran_negative_binomial_pdf).
 * GSL (un-vendored, un-pinned; doc/manual.tex:78) publishes the pdf as
 * exp(lngamma(k+n) - lngamma(n) - lngamma(k+1) + n log p + k log1p(-p)); restated
 * here with libm lgamma. */
double qo_log_negbinom(int k, double p, double n)
{
    double f = lgamma(k + n), a = lgamma(n), b = lgamma(k + 1.0);
    double P = exp(f - a - b + n * log(p) + k * log1p(-p));
    return log(P);
}

/* one SymQualScores: out[0..93] = logSymProb + logQualProb(q); out[94] = logSymProb */
static void sym_qual_scores(double p, double q, double r, double *out)
{
    double lsp = log(p);
    for (int k = 0; k < QO_NQUAL; ++k) out[k] = lsp + qo_log_negbinom(k, q, r);
    out[QO_NQUAL] = lsp;
}

/* QuaffScores::QuaffScores, src/qmodel.cpp:296-325.
 *   insert_pqr[4][3], match_pqr[4][Km][3] hold (symProb, qualTrialSuccessProb, qualNumSuccessfulTrials)
 *   ins_out[4][95]; mat_out[4][Km][95] indexed [ref token][read context k-mer]
 *   trans_out: m2m[Kg] m2i[Kg] m2d[Kg] m2e[Kg] d2d d2m i2i i2m   (4*Kg+4 doubles)
 * Note m2e = log(beginInsert) (sic, :317). */
void qo_build_scores(int Km, int Kg, const double *insert_pqr, const double *match_pqr,
                     const double *beginInsert, const double *beginDelete,
                     double extendInsert, double extendDelete,
                     double *ins_out, double *mat_out, double *trans_out)
{
    for (int i = 0; i < 4; ++i) {
        sym_qual_scores(insert_pqr[i * 3], insert_pqr[i * 3 + 1], insert_pqr[i * 3 + 2], ins_out + i * QO_NQ1);
        for (int j = 0; j < Km; ++j) {
            const double *m = match_pqr + ((size_t)i * Km + j) * 3;
            sym_qual_scores(m[0], m[1], m[2], mat_out + ((size_t)i * Km + j) * QO_NQ1);
        }
    }
    for (int j = 0; j < Kg; ++j) {
        trans_out[j]          = log(1 - beginInsert[j]) + log(1 - beginDelete[j]);
        trans_out[Kg + j]     = log(beginInsert[j]);
        trans_out[2 * Kg + j] = log(1 - beginInsert[j]) + log(beginDelete[j]);
        trans_out[3 * Kg + j] = log(beginInsert[j]);
    }
    trans_out[4 * Kg + 0] = log(extendDelete);
    trans_out[4 * Kg + 1] = log(1 - extendDelete);
    trans_out[4 * Kg + 2] = log(extendInsert);
    trans_out[4 * Kg + 3] = log(1 - extendInsert);
}

/* QuaffNullParams::logLikelihood, src/qmodel.cpp:1875-1890.
 * null_pqr[4][3]; qual may be NULL (no quality scores). */
double qo_null_loglike(double nullEmit, const double *null_pqr, const uint8_t *tok, const uint8_t *qual, int len)
{
    double ll = len * log(nullEmit) + log(1. - nullEmit);
    for (int i = 0; i < len; ++i) {
        const double *n = null_pqr + tok[i] * 3;
        ll += log(n[0]);
        if (qual) ll += qo_log_negbinom(qual[i], n[1], n[2]);
    }
    return ll;
}

/* ------------------------------------------------------------------------ */
/* diagonal envelope: src/diagenv.cpp                                        */
/* ------------------------------------------------------------------------ */

typedef struct { uint64_t km; int pos; } qo_kp;
static int cmp_kp(const void *a, const void *b)
{
    const qo_kp *x = (const qo_kp *)a, *y = (const qo_kp *)b;
    if (x->km != y->km) return x->km < y->km ? -1 : 1;
    return (x->pos > y->pos) - (x->pos < y->pos);
}

/* k-mer match histogram, src/diagenv.cpp:33-40 with KmerIndex src/fastseq.cpp:240-256.
 * hist has xLen+yLen-1 bins, bin d+yLen-1 for diagonal d = i-j (0-based k-mer starts).
 * Sequences shorter than k have no k-mers (the reference's unsigned loop bound
 * underflows there: SURVEY quirk 5; behaviour defined here as "no matches"). */
void qo_diag_histogram(const uint8_t *xtok, int xLen, const uint8_t *ytok, int yLen, int k, uint32_t *hist)
{
    memset(hist, 0, sizeof(uint32_t) * (size_t)(xLen + yLen - 1));
    if (xLen < k || yLen < k) return;
    int nx = xLen - k + 1, ny = yLen - k + 1;
    uint64_t *xk = (uint64_t *)malloc(sizeof(uint64_t) * nx), *yk = (uint64_t *)malloc(sizeof(uint64_t) * ny);
    for (int i = 0; i < nx; ++i) { uint64_t v = 0; for (int a = 0; a < k; ++a) v = v * 4 + xtok[i + a]; xk[i] = v; }
    for (int j = 0; j < ny; ++j) { uint64_t v = 0; for (int a = 0; a < k; ++a) v = v * 4 + ytok[j + a]; yk[j] = v; }
    /* sort read k-mers with positions; binary-search each x k-mer */
    qo_kp *idx = (qo_kp *)malloc(sizeof(qo_kp) * ny);
    for (int j = 0; j < ny; ++j) { idx[j].km = yk[j]; idx[j].pos = j; }
    qsort(idx, ny, sizeof(qo_kp), cmp_kp);
    for (int i = 0; i < nx; ++i) {
        int lo = 0, hi = ny;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (idx[mid].km < xk[i]) lo = mid + 1; else hi = mid; }
        for (int p = lo; p < ny && idx[p].km == xk[i]; ++p) ++hist[i - idx[p].pos + yLen - 1];
    }
    free(idx); free(xk); free(yk);
}

/* DiagonalEnvelope::initSparse / initFull, src/diagenv.cpp:11-106.
 *   threshold >= 0: fixed threshold; < 0: memory mode (maxSize bytes, cellSize bytes/cell).
 *   sparse == 0 -> initFull.
 * Writes the sorted diagonal list to diags (capacity xLen+yLen-1), returns its length. */
int qo_envelope(const uint8_t *xtok, int xLen, const uint8_t *ytok, int yLen,
                int sparse, int k, int bandSize, int threshold, uint64_t cellSize, uint64_t maxSize, int *diags)
{
    const int minD = 1 - yLen, maxD = xLen - 1;
    int full = !sparse;
    if (!full && threshold >= 0) {                       /* :23-29 */
        unsigned minLen = 2u * (unsigned)(k + threshold);
        if ((unsigned)xLen < minLen || (unsigned)yLen < minLen) full = 1;
    }
    if (full) {                                          /* :11-18 */
        int n = 0;
        for (int d = minD; d <= maxD; ++d) diags[n++] = d;
        return n;
    }
    const int nd = xLen + yLen - 1;
    uint32_t *hist = (uint32_t *)malloc(sizeof(uint32_t) * nd);
    qo_diag_histogram(xtok, xLen, ytok, yLen, k, hist);
    /* in[d-minD]: envelope membership; st[d-minD+1]: storage membership (range minD-1..maxD+1) */
    uint8_t *in = (uint8_t *)calloc(nd, 1), *st = (uint8_t *)calloc(nd + 2, 1);
    in[0 - minD] = 1; st[0 - minD + 1] = 1;              /* :52-54: diagonal 0 always present */
    const int half = bandSize / 2;
    const uint64_t diagSize = (uint64_t)(xLen < yLen ? xLen : yLen) * cellSize;
    if (threshold >= 0) {                                /* :63-65,:72-73: plain filter (count >= 1 implied by map membership) */
        for (int b = 0; b < nd; ++b) {
            if (hist[b] == 0 || hist[b] < (uint32_t)threshold) continue;
            int seed = b + minD;
            int lo = seed - half < minD ? minD : seed - half, hi = seed + half > maxD ? maxD : seed + half;
            for (int d = lo; d <= hi; ++d) in[d - minD] = 1;
        }
    } else {                                             /* :68-96: levels in descending count order */
        uint32_t maxc = 0;
        for (int b = 0; b < nd; ++b) if (hist[b] > maxc) maxc = hist[b];
        uint8_t *in2 = (uint8_t *)malloc(nd), *st2 = (uint8_t *)malloc(nd + 2);
        for (uint32_t c = maxc; c >= 1; --c) {
            int any = 0;
            memcpy(in2, in, nd); memcpy(st2, st, nd + 2);
            for (int b = 0; b < nd; ++b) {
                if (hist[b] != c) continue;
                any = 1;
                int seed = b + minD;
                int lo = seed - half < minD ? minD : seed - half, hi = seed + half > maxD ? maxD : seed + half;
                for (int d = lo; d <= hi; ++d) in2[d - minD] = 1;
                for (int d = lo - 1; d <= hi + 1; ++d) st2[d - minD + 1] = 1;
            }
            if (!any) continue;
            uint64_t nst = 0;
            for (int b = 0; b < nd + 2; ++b) nst += st2[b];
            if (nst * diagSize >= maxSize) break;        /* :88-89 */
            memcpy(in, in2, nd); memcpy(st, st2, nd + 2);
        }
        free(in2); free(st2);
    }
    int n = 0;
    for (int b = 0; b < nd; ++b) if (in[b]) diags[n++] = b + minD;
    free(in); free(st); free(hist);
    return n;
}

/* number of DP cells the fill loop visits: sum_j #{d : -j < d <= xLen-j}, diagenv.h:75-85 */
uint64_t qo_envelope_cells(const int *diags, int nd, int xLen, int yLen)
{
    uint64_t cells = 0;
    for (int a = 0; a < nd; ++a) {
        int d = diags[a];
        int jlo = 1 - d > 1 ? 1 - d : 1, jhi = xLen - d < yLen ? xLen - d : yLen;
        if (jhi >= jlo) cells += (uint64_t)(jhi - jlo + 1);
    }
    return cells;
}

/* ------------------------------------------------------------------------ */
/* DP: src/qmodel.cpp:1241-1654                                              */
/* ------------------------------------------------------------------------ */

typedef struct {
    int xLen, yLen, nd;
    const int *diags;
    int *slot;            /* slot[d + yLen + 1] = storage row of diagonal d, or -1; range d in [-yLen-1, xLen+1] */
    int nslot;
    double *mat, *ins, *del;   /* [nslot][yLen+1], -inf initialised (QuaffDPMatrixContainer, :1243-1253) */
} qo_matrix;

static void mx_init(qo_matrix *m, const int *diags, int nd, int xLen, int yLen)
{
    m->xLen = xLen; m->yLen = yLen; m->nd = nd; m->diags = diags;
    int range = xLen + yLen + 3;
    m->slot = (int *)malloc(sizeof(int) * range);
    for (int a = 0; a < range; ++a) m->slot[a] = -1;
    for (int a = 0; a < nd; ++a) m->slot[diags[a] + yLen + 1] = a;
    m->nslot = nd;
    size_t n = (size_t)nd * (yLen + 1);
    m->mat = (double *)malloc(sizeof(double) * n);
    m->ins = (double *)malloc(sizeof(double) * n);
    m->del = (double *)malloc(sizeof(double) * n);
    for (size_t a = 0; a < n; ++a) m->mat[a] = m->ins[a] = m->del[a] = NEG_INF;
}
static void mx_free(qo_matrix *m) { free(m->slot); free(m->mat); free(m->ins); free(m->del); }

/* Outside the envelope, and on row 0 / column 0, every state is -inf: the pad
 * diagonals and the i==0 / j==0 storage cells of the reference are never
 * written (src/qmodel.h:367-387, diagenv.cpp:108-133). */
static inline long mx_idx(const qo_matrix *m, int i, int j)
{
    if (i < 1 || j < 1 || i > m->xLen || j > m->yLen) return -1;
    int s = m->slot[i - j + m->yLen + 1];
    return s < 0 ? -1 : (long)s * (m->yLen + 1) + j;
}
#define MAT(m, i, j) ({ long _x = mx_idx(m, i, j); _x < 0 ? NEG_INF : (m)->mat[_x]; })
#define INS(m, i, j) ({ long _x = mx_idx(m, i, j); _x < 0 ? NEG_INF : (m)->ins[_x]; })
#define DEL(m, i, j) ({ long _x = mx_idx(m, i, j); _x < 0 ? NEG_INF : (m)->del[_x]; })

/* Per-pair inputs shared by the fills (QuaffDPMatrix ctor, src/qmodel.cpp:1308-1324). */
typedef struct {
    int xLen, yLen, Km, Kg, local;
    const uint8_t *xtok, *ytok;
    const uint8_t *yqual;          /* NULL => no quality scores */
    const uint32_t *ymk, *ygk;     /* match / indel context k-mers, index j-1 = context ending at read pos j */
    const double *ins, *mat;       /* [4][95], [4][Km][95] */
    const double *trans;           /* m2m[Kg] m2i[Kg] m2d[Kg] m2e[Kg] d2d d2m i2i i2m */
} qo_pair;

static inline double p_m2m(const qo_pair *p, int j) { return p->trans[j == 0 ? 0 : p->ygk[j - 1]]; }          /* yIndelKmer padded with 0, :1322-1323 */
static inline double p_m2i(const qo_pair *p, int j) { return p->trans[p->Kg + (j == 0 ? 0 : p->ygk[j - 1])]; }
static inline double p_m2d(const qo_pair *p, int j) { return p->trans[2 * p->Kg + (j == 0 ? 0 : p->ygk[j - 1])]; }
static inline double p_m2e(const qo_pair *p, int j) { return p->trans[3 * p->Kg + (j == 0 ? 0 : p->ygk[j - 1])]; }
static inline double p_d2d(const qo_pair *p) { return p->trans[4 * p->Kg]; }
static inline double p_d2m(const qo_pair *p) { return p->trans[4 * p->Kg + 1]; }
static inline double p_i2i(const qo_pair *p) { return p->trans[4 * p->Kg + 2]; }
static inline double p_i2m(const qo_pair *p) { return p->trans[4 * p->Kg + 3]; }
/* matchEmitScore / insertEmitScore, src/qmodel.h:406-413 */
static inline double p_memit(const qo_pair *p, int i, int j)
{
    int q = p->yqual ? p->yqual[j - 1] : QO_NQUAL;
    return p->mat[((size_t)p->xtok[i - 1] * p->Km + p->ymk[j - 1]) * QO_NQ1 + q];
}
static inline double p_iemit(const qo_pair *p, int j)
{
    int q = p->yqual ? p->yqual[j - 1] : QO_NQUAL;
    return p->ins[(size_t)p->ytok[j - 1] * QO_NQ1 + q];
}
static inline int qmax_i(int a, int b) { return a > b ? a : b; }
static inline int qmin_i(int a, int b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }   /* std::max(a,b): returns a unless a<b */

/* QuaffViterbiMatrix ctor, src/qmodel.cpp:1512-1560 */
static double viterbi_fill(const qo_pair *p, qo_matrix *m)
{
    double end = NEG_INF;
    const int xLen = p->xLen, yLen = p->yLen;
    for (int j = 1; j <= yLen; ++j)
        for (int a = 0; a < m->nd; ++a) {
            int d = m->diags[a], i = d + j;
            if (i < 1 || i > xLen) continue;                     /* diagenv.h:75-85 */
            long c = (long)a * (yLen + 1) + j;
            double mt = dmax(dmax(MAT(m, i - 1, j - 1) + p_m2m(p, j - 1), DEL(m, i - 1, j - 1) + p_d2m(p)), INS(m, i - 1, j - 1) + p_i2m(p));
            if (j == 1 && (i == 1 || p->local)) mt = dmax(mt, 0.);
            mt += p_memit(p, i, j);
            m->mat[c] = mt;
            m->ins[c] = p_iemit(p, j) + dmax(INS(m, i, j - 1) + p_i2i(p), MAT(m, i, j - 1) + p_m2i(p, j - 1));
            m->del[c] = dmax(DEL(m, i - 1, j) + p_d2d(p), MAT(m, i - 1, j) + p_m2d(p, j));
            if (j == yLen && (i == xLen || p->local)) end = dmax(end, mt + p_m2e(p, j));
        }
    return end;
}

/* QuaffViterbiMatrix::alignment, src/qmodel.cpp:1562-1646: end-cell choice and
 * traceback.  ops_out receives 'M','I','D' in alignment order; returns the
 * number of columns, or -1 if the result is -inf.  updateMax (:1294-1299) is a
 * strict '>' tested in the order Match, Insert, Delete, Start. */
static int viterbi_traceback(const qo_pair *p, const qo_matrix *m, int *xStart, int *xEnd, char *ops_out, int cap)
{
    const int xLen = p->xLen, yLen = p->yLen;
    int xe = xLen;
    if (p->local) {
        double best = NEG_INF;
        for (int ie = xLen; ie > 0; --ie) {
            double sc = MAT(m, ie, yLen) + p_m2e(p, yLen);
            if (ie == xLen || sc > best) { best = sc; xe = ie; }
        }
    }
    int i = xe, j = yLen, n = 0;
    enum { Start, Match, Insert, Delete } state = Match;
    char *rev = (char *)malloc((size_t)xLen + yLen + 2);
    while (state != Start) {
        double src = NEG_INF, e, c;
        switch (state) {
        case Match:
            e = p_memit(p, i, j); --i; --j; rev[n++] = 'M';
            c = MAT(m, i, j) + p_m2m(p, j) + e; if (c > src) { src = c; state = Match; }
            c = INS(m, i, j) + p_i2m(p) + e;    if (c > src) { src = c; state = Insert; }
            c = DEL(m, i, j) + p_d2m(p) + e;    if (c > src) { src = c; state = Delete; }
            if (j == 0 && (i == 0 || p->local)) { if (e > src) { src = e; state = Start; } }
            break;
        case Insert:
            e = p_iemit(p, j); --j; rev[n++] = 'I';
            c = MAT(m, i, j) + p_m2i(p, j) + e; if (c > src) { src = c; state = Match; }
            c = INS(m, i, j) + p_i2i(p) + e;    if (c > src) { src = c; state = Insert; }
            break;
        case Delete:
            --i; rev[n++] = 'D';
            c = MAT(m, i, j) + p_m2d(p, j);     if (c > src) { src = c; state = Match; }
            c = DEL(m, i, j) + p_d2d(p);        if (c > src) { src = c; state = Delete; }
            break;
        default: break;
        }
        if (n > xLen + yLen || i < 0 || j < 0) { free(rev); return -2; }
    }
    *xStart = i + 1; *xEnd = xe;
    if (n > cap) { free(rev); return -3; }
    for (int a = 0; a < n; ++a) ops_out[a] = rev[n - 1 - a];
    free(rev);
    return n;
}

/* One (ref,read) Viterbi: fill + (if finite) traceback.  Returns result (raw
 * Viterbi log-likelihood, before null-model adjustment). */
double qo_viterbi(int xLen, int yLen, int Km, int Kg, int local,
                  const uint8_t *xtok, const uint8_t *ytok, const uint8_t *yqual,
                  const uint32_t *ymk, const uint32_t *ygk,
                  const double *ins, const double *mat, const double *trans,
                  const int *diags, int nd,
                  int want_tb, int *xStart, int *xEnd, char *ops, int ops_cap, int *n_ops,
                  double *mat_dump /* optional [nd][yLen+1][3] */)
{
    qo_pair p = { xLen, yLen, Km, Kg, local, xtok, ytok, yqual, ymk, ygk, ins, mat, trans };
    qo_matrix m;
    mx_init(&m, diags, nd, xLen, yLen);
    double result = viterbi_fill(&p, &m);
    if (n_ops) *n_ops = -1;
    if (want_tb && result > NEG_INF) *n_ops = viterbi_traceback(&p, &m, xStart, xEnd, ops, ops_cap);
    if (mat_dump) {
        size_t n = (size_t)nd * (yLen + 1);
        for (size_t a = 0; a < n; ++a) { mat_dump[a * 3] = m.mat[a]; mat_dump[a * 3 + 1] = m.ins[a]; mat_dump[a * 3 + 2] = m.del[a]; }
    }
    mx_free(&m);
    return result;
}

/* QuaffForwardMatrix ctor, src/qmodel.cpp:1343-1391 */
static double forward_fill(const qo_pair *p, qo_matrix *m)
{
    double end = NEG_INF;
    const int xLen = p->xLen, yLen = p->yLen;
    for (int j = 1; j <= yLen; ++j)
        for (int a = 0; a < m->nd; ++a) {
            int d = m->diags[a], i = d + j;
            if (i < 1 || i > xLen) continue;
            long c = (long)a * (yLen + 1) + j;
            double mt = lse3(MAT(m, i - 1, j - 1) + p_m2m(p, j - 1), DEL(m, i - 1, j - 1) + p_d2m(p), INS(m, i - 1, j - 1) + p_i2m(p));
            if (j == 1 && (i == 1 || p->local)) mt = qo_lse(mt, 0.);
            mt += p_memit(p, i, j);
            m->mat[c] = mt;
            m->ins[c] = p_iemit(p, j) + qo_lse(INS(m, i, j - 1) + p_i2i(p), MAT(m, i, j - 1) + p_m2i(p, j - 1));
            m->del[c] = qo_lse(DEL(m, i - 1, j) + p_d2d(p), MAT(m, i - 1, j) + p_m2d(p, j));
            if (j == yLen && (i == xLen || p->local)) end = qo_lse(end, mt + p_m2e(p, yLen));
        }
    return end;
}

/* Flattened QuaffCounts layout used by qo_forward_backward (and by the product's
 * C-ABI): ins[4][94] | mat[4][Km][94] | m2m[Kg] m2i[Kg] m2d[Kg] m2e[Kg] | d2d d2m i2i i2m */
static inline size_t cnt_size(int Km, int Kg) { return (size_t)(4 + 4 * Km) * QO_NQUAL + 4 * Kg + 4; }
int qo_counts_size(int Km, int Kg) { return (int)cnt_size(Km, Kg); }

/* QuaffBackwardMatrix ctor + transCount, src/qmodel.cpp:1393-1510.  Push-style:
 * each destination cell adds into its source cells; the Backward matrix needs
 * storage for row 0 / column 0 and the pad diagonals because the reference
 * writes (harmlessly) into them; here such writes go to a scratch cell. */
typedef struct { double *bm, *bi, *bd; double scratch; } qo_back;
static inline double *bk(qo_back *b, const qo_matrix *m, double *arr, int i, int j)
{
    long x = mx_idx(m, i, j);
    if (x < 0) { b->scratch = NEG_INF; return &b->scratch; }
    return arr + x;
}
static inline double trans_count(double *backSrc, double fwdSrc, double trans, double backDest, double fwdResult)
{
    double tbd = trans + backDest;
    double count = exp(fwdSrc + tbd - fwdResult);
    *backSrc = qo_lse(*backSrc, tbd);
    return count;
}

double qo_forward_backward(int xLen, int yLen, int Km, int Kg, int local,
                           const uint8_t *xtok, const uint8_t *ytok, const uint8_t *yqual,
                           const uint32_t *ymk, const uint32_t *ygk,
                           const double *ins, const double *mat, const double *trans,
                           const int *diags, int nd,
                           int want_back, double *counts /* cnt_size, zeroed here */, double *back_result)
{
    qo_pair p = { xLen, yLen, Km, Kg, local, xtok, ytok, yqual, ymk, ygk, ins, mat, trans };
    qo_matrix f;
    mx_init(&f, diags, nd, xLen, yLen);
    const double fres = forward_fill(&p, &f);
    if (want_back && yqual) {
        size_t n = (size_t)nd * (yLen + 1);
        qo_back b;
        b.bm = (double *)malloc(sizeof(double) * n); b.bi = (double *)malloc(sizeof(double) * n); b.bd = (double *)malloc(sizeof(double) * n);
        for (size_t a = 0; a < n; ++a) b.bm[a] = b.bi[a] = b.bd[a] = NEG_INF;
        memset(counts, 0, sizeof(double) * cnt_size(Km, Kg));
        double *cins = counts, *cmat = counts + 4 * QO_NQUAL, *ctr = counts + (size_t)(4 + 4 * Km) * QO_NQUAL;
        double *c_m2m = ctr, *c_m2i = ctr + Kg, *c_m2d = ctr + 2 * Kg, *c_m2e = ctr + 3 * Kg;
        double *c_d2d = ctr + 4 * Kg, *c_d2m = c_d2d + 1, *c_i2i = c_d2d + 2, *c_i2m = c_d2d + 3;
        double start = NEG_INF;
        const double end = 0;
#define GK(j) ((j) == 0 ? 0 : ygk[(j) - 1])
        for (int j = yLen; j > 0; --j)
            for (int a = nd - 1; a >= 0; --a) {
                int d = diags[a], i = d + j;
                if (i < 1 || i > xLen) continue;
                long c = (long)a * (yLen + 1) + j;
                if (j == yLen && (i == xLen || local))
                    c_m2e[GK(yLen)] += trans_count(&b.bm[c], f.mat[c], p_m2e(&p, yLen), end, fres);
                const double matEmit = p_memit(&p, i, j), matDest = b.bm[c];
                double *matCount = &cmat[((size_t)xtok[i - 1] * Km + ymk[j - 1]) * QO_NQUAL + yqual[j - 1]];
                double v;
                v = trans_count(bk(&b, &f, b.bm, i - 1, j - 1), MAT(&f, i - 1, j - 1), p_m2m(&p, j - 1) + matEmit, matDest, fres);
                c_m2m[GK(j - 1)] += v; *matCount += v;
                v = trans_count(bk(&b, &f, b.bd, i - 1, j - 1), DEL(&f, i - 1, j - 1), p_d2m(&p) + matEmit, matDest, fres);
                *c_d2m += v; *matCount += v;
                v = trans_count(bk(&b, &f, b.bi, i - 1, j - 1), INS(&f, i - 1, j - 1), p_i2m(&p) + matEmit, matDest, fres);
                *c_i2m += v; *matCount += v;
                if (j == 1 && (i == 1 || local)) {
                    v = trans_count(&start, 0., matEmit, matDest, fres);     /* fwd.start == 0, :1346 */
                    *matCount += v;
                }
                const double insEmit = p_iemit(&p, j), insDest = b.bi[c];
                double *insCount = &cins[(size_t)ytok[j - 1] * QO_NQUAL + yqual[j - 1]];
                v = trans_count(bk(&b, &f, b.bm, i, j - 1), MAT(&f, i, j - 1), p_m2i(&p, j - 1) + insEmit, insDest, fres);
                c_m2i[GK(j - 1)] += v; *insCount += v;
                v = trans_count(bk(&b, &f, b.bi, i, j - 1), INS(&f, i, j - 1), p_i2i(&p) + insEmit, insDest, fres);
                *c_i2i += v; *insCount += v;
                const double delDest = b.bd[c];
                v = trans_count(bk(&b, &f, b.bm, i - 1, j), MAT(&f, i - 1, j), p_m2d(&p, j), delDest, fres);
                c_m2d[GK(j)] += v;
                v = trans_count(bk(&b, &f, b.bd, i - 1, j), DEL(&f, i - 1, j), p_d2d(&p), delDest, fres);
                *c_d2d += v;
            }
#undef GK
        if (back_result) *back_result = start;
        free(b.bm); free(b.bi); free(b.bd);
    }
    mx_free(&f);
    return fres;
}

/* ------------------------------------------------------------------------ */
/* read-vs-read overlap: src/qoverlap.cpp                                    */
/* ------------------------------------------------------------------------ */

/* QuaffOverlapScores ctor, src/qoverlap.cpp:9-75.
 *   mmi_out[Km][Km][95][95]: [qi][qj] = logSymQualPairProb; [qi][94] = logSymPairXQualProb; [94][qj] = ...YQualProb;
 *                            [94][94] = logSymPairProb  (all "match minus insert")
 *   gap_out: m2m[Kg][Kg] m2i[Kg][Kg] m2d[Kg][Kg] | i2m i2i i2d d2m d2i d2d   (3*Kg*Kg + 6 doubles)
 * ins/mat/trans-free inputs are the QuaffScores tables of qo_build_scores. */
void qo_overlap_scores(int Km, int Kg, const double *refBase, const double *beginInsert, const double *beginDelete,
                       double extendInsert, double extendDelete, const double *ins, const double *mat,
                       int yComplemented, double *mmi_out, double *gap_out)
{
    if (!lse_table) qo_lse_table();
    double *gapOpen = (double *)malloc(sizeof(double) * Kg);
    double pGapIsInsertSum = 0, gapAdjSum = 0;
    for (int j = 0; j < Kg; ++j) {                      /* :24-32 */
        const double readInsertProb = beginInsert[j];
        const double readDeleteProb = (1 - beginInsert[j]) * beginDelete[j];
        gapOpen[j] = readInsertProb + readDeleteProb;
        const double pGapIsInsert = readInsertProb / gapOpen[j];
        const double gapAdjacentProb = pGapIsInsert * readInsertProb + (1 - pGapIsInsert) * gapOpen[j] / (1 - extendDelete * (1 - gapOpen[j]));
        pGapIsInsertSum += pGapIsInsert;                /* accumulate(..., 0.) in index order, :41,:44 */
        gapAdjSum += gapAdjacentProb;
    }
    double *m2m = gap_out, *m2i = gap_out + Kg * Kg, *m2d = gap_out + 2 * Kg * Kg, *sc = gap_out + 3 * Kg * Kg;
    for (int i = 0; i < Kg; ++i)
        for (int j = 0; j < Kg; ++j) {                  /* :34-39 */
            m2m[i * Kg + j] = log(1 - gapOpen[i]) + log(1 - gapOpen[j]);
            m2i[i * Kg + j] = log(gapOpen[i]);
            m2d[i * Kg + j] = log(1 - gapOpen[i]) + log(gapOpen[j]);
        }
    const double pGapIsInsert = pGapIsInsertSum / Kg;
    const double meanGapLength = pGapIsInsert / extendInsert + (1 - pGapIsInsert) / extendDelete;
    const double gapExtendProb = 1 / meanGapLength;
    const double gapAdjacentProb = gapAdjSum / Kg;
    sc[1] = sc[5] = log(gapExtendProb);                                   /* i2i = d2d */
    sc[2] = sc[4] = log(1 - gapExtendProb) + log(gapAdjacentProb);       /* i2d = d2i */
    sc[0] = sc[3] = log(1 - gapExtendProb) + log(1 - gapAdjacentProb);   /* i2m = d2m */
    const size_t Q2 = (size_t)QO_NQ1 * QO_NQ1;
    for (int i = 0; i < Km; ++i)
        for (int j = 0; j < Km; ++j) {
            const int iS = i & 3, jS = j & 3;          /* k-mer suffix = the emitted base (:54-58) */
            double *t = mmi_out + ((size_t)i * Km + j) * Q2;
            for (int a = 0; a < QO_NQ1; ++a) t[a * QO_NQ1 + QO_NQUAL] = NEG_INF;
            for (int a = 0; a < QO_NQ1; ++a) t[QO_NQUAL * QO_NQ1 + a] = NEG_INF;
            for (int ik = 0; ik < QO_NQUAL; ++ik)
                for (int jk = 0; jk < QO_NQUAL; ++jk) {
                    double mij = NEG_INF;
                    for (int r = 0; r < 4; ++r) {
                        const int yr = yComplemented ? 3 - r : r;
                        mij = qo_lse(mij, log(refBase[r]) + mat[((size_t)r * Km + i) * QO_NQ1 + ik] + mat[((size_t)yr * Km + j) * QO_NQ1 + jk]);
                    }
                    const double *xi = ins + (size_t)iS * QO_NQ1, *yi = ins + (size_t)jS * QO_NQ1;
                    t[ik * QO_NQ1 + jk] = mij - xi[ik] - yi[jk];
                    t[ik * QO_NQ1 + QO_NQUAL] = qo_lse(t[ik * QO_NQ1 + QO_NQUAL], mij - xi[ik] - yi[QO_NQUAL]);
                    t[QO_NQUAL * QO_NQ1 + jk] = qo_lse(t[QO_NQUAL * QO_NQ1 + jk], mij - xi[QO_NQUAL] - yi[jk]);
                    t[QO_NQUAL * QO_NQ1 + QO_NQUAL] = qo_lse(t[QO_NQUAL * QO_NQ1 + QO_NQUAL], mij - xi[QO_NQUAL] - yi[QO_NQUAL]);
                }
        }
    free(gapOpen);
}

typedef struct {
    int xLen, yLen, Km, Kg;
    const uint32_t *xmk, *xgk, *ymk, *ygk;   /* context k-mers; y-side already in the (possibly complemented) orientation */
    const uint8_t *xq, *yq;                 /* NULL => no quality */
    const double *mmi, *gap;
} qo_opair;
/* accessor swaps of src/qoverlap.h:46-50, replicated literally: i2mScore()=i2i, i2iScore()=i2m, i2dScore()=i2d,
 * d2mScore()=d2i, d2iScore()=d2m, d2dScore()=d2d.  gap scalars: [0]=i2m [1]=i2i [2]=i2d [3]=d2m [4]=d2i [5]=d2d */
#define O_SC(p) ((p)->gap + 3 * (p)->Kg * (p)->Kg)
static inline double o_i2mScore(const qo_opair *p) { return O_SC(p)[1]; }
static inline double o_i2iScore(const qo_opair *p) { return O_SC(p)[0]; }
static inline double o_i2dScore(const qo_opair *p) { return O_SC(p)[2]; }
static inline double o_d2mScore(const qo_opair *p) { return O_SC(p)[4]; }
static inline double o_d2iScore(const qo_opair *p) { return O_SC(p)[3]; }
static inline double o_d2dScore(const qo_opair *p) { return O_SC(p)[5]; }
static inline int o_gx(const qo_opair *p, int i) { return i == 0 ? 0 : (int)p->xgk[i - 1]; }
static inline int o_gy(const qo_opair *p, int j) { return j == 0 ? 0 : (int)p->ygk[j - 1]; }
static inline double o_m2m(const qo_opair *p, int i, int j) { return p->gap[o_gx(p, i) * p->Kg + o_gy(p, j)]; }
static inline double o_m2i(const qo_opair *p, int i, int j) { return p->gap[p->Kg * p->Kg + o_gx(p, i) * p->Kg + o_gy(p, j)]; }
static inline double o_m2d(const qo_opair *p, int i, int j) { return p->gap[2 * p->Kg * p->Kg + o_gx(p, i) * p->Kg + o_gy(p, j)]; }
static inline double o_emit(const qo_opair *p, int i, int j)   /* matchEmitScore, src/qoverlap.h:52-61 */
{
    const int qi = p->xq ? p->xq[i - 1] : QO_NQUAL, qj = p->yq ? p->yq[j - 1] : QO_NQUAL;
    return p->mmi[((size_t)p->xmk[i - 1] * p->Km + p->ymk[j - 1]) * QO_NQ1 * QO_NQ1 + qi * QO_NQ1 + qj];
}

/* QuaffOverlapViterbiMatrix ctor (fill) + alignment() (end cell, traceback), src/qoverlap.cpp:77-290.
 * Returns `end` (max over the last row / last column of mat; the caller adds the two insert scores);
 * ops_out: raw traceback states 'M','I','D' in alignment order (before indel squashing). */
double qo_overlap_viterbi(int xLen, int yLen, int Km, int Kg,
                          const uint32_t *xmk, const uint32_t *xgk, const uint8_t *xq,
                          const uint32_t *ymk, const uint32_t *ygk, const uint8_t *yq,
                          const double *mmi, const double *gap, const int *diags, int nd,
                          int want_tb, int *xStart, int *xEnd, int *yStart, int *yEnd, char *ops, int ops_cap, int *n_ops)
{
    qo_opair p = { xLen, yLen, Km, Kg, xmk, xgk, ymk, ygk, xq, yq, mmi, gap };
    qo_matrix m;
    mx_init(&m, diags, nd, xLen, yLen);
    if (!lse_table) qo_lse_table();
    double end = NEG_INF;
    for (int j = 1; j <= yLen; ++j)
        for (int a = 0; a < nd; ++a) {
            const int d = diags[a], i = d + j;
            if (i < 1 || i > xLen) continue;
            const long c = (long)a * (yLen + 1) + j;
            double mt = dmax(dmax(MAT(&m, i - 1, j - 1) + o_m2m(&p, i - 1, j - 1), DEL(&m, i - 1, j - 1) + o_d2mScore(&p)),
                             INS(&m, i - 1, j - 1) + o_i2mScore(&p));
            if (j == 1 || i == 1) mt = dmax(mt, 0.);
            mt += o_emit(&p, i, j);
            m.mat[c] = mt;
            m.ins[c] = dmax(qo_lse(INS(&m, i, j - 1) + o_i2iScore(&p), DEL(&m, i, j - 1) + o_d2iScore(&p)), MAT(&m, i, j - 1) + o_m2i(&p, i, j - 1));
            m.del[c] = dmax(qo_lse(DEL(&m, i - 1, j) + o_d2dScore(&p), INS(&m, i - 1, j) + o_d2iScore(&p)), MAT(&m, i - 1, j) + o_m2d(&p, i - 1, j));
            if (j == yLen || i == xLen) end = dmax(end, mt);
        }
    if (n_ops) *n_ops = -1;
    if (want_tb && end > NEG_INF) {
        int xe = xLen, ye = yLen;
        double best = MAT(&m, xLen, yLen), sc;
        for (int ie = xLen; ie > 0; --ie) { sc = MAT(&m, ie, yLen); if (sc > best) { best = sc; xe = ie; ye = yLen; } }
        for (int je = yLen; je > 0; --je) { sc = MAT(&m, xLen, je); if (sc > best) { best = sc; xe = xLen; ye = je; } }
        int i = xe, j = ye, n = 0;
        enum { Start, Match, Insert, Delete } state = Match;
        char *rev = (char *)malloc((size_t)xLen + yLen + 2);
        int bad = 0;
        while (state != Start) {
            double src = NEG_INF, e, cnd;
            switch (state) {
            case Match:
                e = o_emit(&p, i, j); --i; --j; rev[n++] = 'M';
                cnd = MAT(&m, i, j) + o_m2m(&p, i, j) + e; if (cnd > src) { src = cnd; state = Match; }
                cnd = INS(&m, i, j) + o_i2mScore(&p) + e;  if (cnd > src) { src = cnd; state = Insert; }
                cnd = DEL(&m, i, j) + o_d2mScore(&p) + e;  if (cnd > src) { src = cnd; state = Delete; }
                if (j == 0 || i == 0) { if (e > src) { src = e; state = Start; } }
                break;
            case Insert:
                --j; rev[n++] = 'I';
                cnd = MAT(&m, i, j) + o_m2i(&p, i, j);     if (cnd > src) { src = cnd; state = Match; }
                cnd = INS(&m, i, j) + o_i2iScore(&p);      if (cnd > src) { src = cnd; state = Insert; }
                cnd = DEL(&m, i, j) + o_d2iScore(&p);      if (cnd > src) { src = cnd; state = Delete; }
                break;
            case Delete:
                --i; rev[n++] = 'D';
                cnd = MAT(&m, i, j) + o_m2d(&p, i, j);     if (cnd > src) { src = cnd; state = Match; }
                cnd = INS(&m, i, j) + o_i2dScore(&p);      if (cnd > src) { src = cnd; state = Insert; }
                cnd = DEL(&m, i, j) + o_d2dScore(&p);      if (cnd > src) { src = cnd; state = Delete; }
                break;
            default: break;
            }
            if (n > xLen + yLen || i < 0 || j < 0) { bad = 1; break; }
        }
        if (bad) *n_ops = -2;
        else if (n > ops_cap) *n_ops = -3;
        else {
            *xStart = i + 1; *xEnd = xe; *yStart = j + 1; *yEnd = ye;
            for (int a = 0; a < n; ++a) ops[a] = rev[n - 1 - a];
            *n_ops = n;
        }
        free(rev);
    }
    mx_free(&m);
    return end;
}

/* ------------------------------------------------------------------------ */
/* Path re-scoring: the log-probability of ONE given alignment path, accumulated in the same association as the Viterbi
 * recurrence (src/qmodel.cpp:1532-1552): O(path) instead of O(cells), so it checks alignments of full-size runs.  For the
 * Viterbi path it reproduces QuaffViterbiMatrix::result bit-for-bit; for any other path it is <= result. */
double qo_rescore_path(int xLen, int yLen, int Km, int Kg, int local,
                       const uint8_t *xtok, const uint8_t *ytok, const uint8_t *yqual,
                       const uint32_t *ymk, const uint32_t *ygk,
                       const double *ins, const double *mat, const double *trans,
                       int xStart, const char *ops, int n_ops)
{
    qo_pair p = { xLen, yLen, Km, Kg, local, xtok, ytok, yqual, ymk, ygk, ins, mat, trans };
    int i = xStart - 1, j = 0;            /* cell before the first column */
    double sc = 0;                        /* start */
    char prev = 'S';
    if (!local && xStart != 1) return NEG_INF;
    for (int a = 0; a < n_ops; ++a) {
        const char op = ops[a];
        if (op == 'M') {
            ++i; ++j;
            if (i > xLen || j > yLen) return NEG_INF;
            double t;
            if (prev == 'S') { if (j != 1) return NEG_INF; t = 0; }
            else if (prev == 'M') t = sc + p_m2m(&p, j - 1);
            else if (prev == 'I') t = sc + p_i2m(&p);
            else t = sc + p_d2m(&p);
            sc = t + p_memit(&p, i, j);
        } else if (op == 'I') {
            ++j;
            if (j > yLen || prev == 'S' || prev == 'D') return NEG_INF;
            sc = p_iemit(&p, j) + (prev == 'I' ? sc + p_i2i(&p) : sc + p_m2i(&p, j - 1));
        } else {
            ++i;
            if (i > xLen || prev == 'S' || prev == 'I') return NEG_INF;
            sc = prev == 'D' ? sc + p_d2d(&p) : sc + p_m2d(&p, j);
        }
        prev = op;
    }
    if (j != yLen || prev != 'M') return NEG_INF;
    if (!local && i != xLen) return NEG_INF;
    return sc + p_m2e(&p, yLen);
}
