/*
 * kmx_oracle.c -- CPU restatement of the kmtricks counting/merge hot path.
 * TEST INFRASTRUCTURE ONLY (see kmx_oracle.h).  Plain C99, single thread.
 * Parity: PINNED against the reference's golden vectors (tests/golden/).
 * File:line citations are relative to /root/reference.
 */
#include "kmx_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ===================================================================== */
/* XXH64 -- Cyan4973/xxHash (absent submodule thirdparty/xxHash, pin unknown;
 * the algorithm is specification-stable).  Call sites:
 * include/kmtricks/gatb/sorting_count.hpp:356, include/kmtricks/repartition.hpp:52 */

#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xxh_round(uint64_t acc, uint64_t in) { return rotl64(acc + in * P2, 31) * P1; }
static inline uint64_t xxh_merge(uint64_t acc, uint64_t v) { return (acc ^ xxh_round(0, v)) * P1 + P4; }

uint64_t orc_xxh64(const void* data, size_t len, uint64_t seed)
{
  const uint8_t* p = (const uint8_t*)data;
  const uint8_t* end = p + len;
  uint64_t h;
  if (len >= 32) {
    uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    const uint8_t* lim = end - 32;
    do {
      v1 = xxh_round(v1, rd64(p)); v2 = xxh_round(v2, rd64(p + 8));
      v3 = xxh_round(v3, rd64(p + 16)); v4 = xxh_round(v4, rd64(p + 24));
      p += 32;
    } while (p <= lim);
    h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
    h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
  } else {
    h = seed + P5;
  }
  h += (uint64_t)len;
  while (p + 8 <= end) { h ^= xxh_round(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
  if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
  while (p < end) { h ^= (uint64_t)(*p) * P5; h = rotl64(h, 11) * P1; p++; }
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

/* ===================================================================== */
/* nucleotides and k-mers */

/* gatb tools/misc/api/Data.hpp:179-196: only ACGTacgt are valid */
int orc_nt_valid(unsigned char c)
{
  switch (c) { case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': return 1; }
  return 0;
}
static inline unsigned nt_code(unsigned char c) { return (c >> 1) & 3u; } /* A0 C1 T2 G3 */

/* multi-word helpers: value = sum nt_j 4^(k-1-j), little-endian words (kmer.hpp:938) */
static inline unsigned kw_digit(const uint64_t* w, int i) { return (unsigned)(w[i >> 5] >> ((i & 31) * 2)) & 3u; }
static inline void kw_set_digit(uint64_t* w, int i, unsigned d) { w[i >> 5] |= (uint64_t)d << ((i & 31) * 2); }

/* words of a k-mer as files, hashes and comparisons see it: ceil(k / 32) (kmer.hpp:215 m_n_data, io/kmer_file.hpp:84 kmer_slots,
 * gatb/sorting_count.hpp:351 the hashed bytes) -- whatever Kmer<MAX_K> the reference instantiates for k (the first KMER_LIST entry
 * 32 64 96 128 above k, loop_executor.hpp:47-63), which only sets how many k-mers a super-k-mer holds: (MAX_K * 2 - 8) / 2 */
static int kw_of_k(int k) { return (k + 31) / 32; }

static inline int kw_less(const uint64_t* a, const uint64_t* b, int kw)
{ /* most significant word first, kmer.hpp:262-268 */
  for (int i = kw - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] < b[i]; }
  return 0;
}
static inline int kw_eq(const uint64_t* a, const uint64_t* b, int kw)
{
  for (int i = 0; i < kw; i++) if (a[i] != b[i]) return 0;
  return 1;
}

static inline uint64_t revcomp1(uint64_t x, int k)
{ /* reverse the 2-bit digits, complement = digit ^ 2 (A<->T, C<->G) */
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = __builtin_bswap64(x);
  x ^= 0xAAAAAAAAAAAAAAAAULL;
  return x >> (64 - 2 * k);
}

void orc_revcomp(const uint64_t* in, uint64_t* out, int k, int kw)
{ /* rc.digit[i] = in.digit[k-1-i] ^ 2 ; gatb Model.hpp:857-884 (table[1]) */
  if (kw == 1) { out[0] = revcomp1(in[0], k); return; }
  uint64_t tmp[4] = {0, 0, 0, 0};
  for (int i = 0; i < k; i++) kw_set_digit(tmp, i, kw_digit(in, k - 1 - i) ^ 2u);
  for (int i = 0; i < kw; i++) out[i] = tmp[i];
}

void orc_kmer_to_string(const uint64_t* words, int k, char* out)
{ /* kmer.hpp:797-810, alphabet index A0 C1 T2 G3 */
  static const char bToN[4] = {'A', 'C', 'T', 'G'};
  for (int j = 0; j < k; j++) out[j] = bToN[kw_digit(words, k - 1 - j)];
  out[k] = 0;
}

void orc_kmer_from_string(const char* s, int k, uint64_t* words, int kw)
{
  for (int i = 0; i < kw; i++) words[i] = 0;
  for (int j = 0; j < k; j++) kw_set_digit(words, k - 1 - j, nt_code((unsigned char)s[j]));
}

/* ===================================================================== */
/* minimizers and repartition */

static int mmer_allowed(uint32_t mmer, uint32_t len)
{ /* gatb Model.hpp:1220-1251 : ban m-mers with AA inside, except as prefix */
  uint64_t mmask_m1 = ((uint64_t)1 << ((len - 2) * 2)) - 1;
  uint64_t mask_ma1 = 0x5555555555555555ULL & mmask_m1;
  uint64_t a1 = mmer;
  a1 = ~(a1 | (a1 >> 2));
  a1 = ((a1 >> 1) & a1) & mask_ma1;
  return a1 == 0;
}

void orc_minimizer_lut(int m, uint32_t* lut)
{ /* gatb Model.hpp:1040-1064 */
  uint64_t n = (uint64_t)1 << (2 * m);
  uint32_t mask = (uint32_t)(n - 1);
  for (uint64_t x = 0; x < n; x++) {
    uint32_t rc = (uint32_t)revcomp1(x, m);
    uint32_t v = rc < (uint32_t)x ? rc : (uint32_t)x;
    if (!mmer_allowed(v, (uint32_t)m)) v = mask;
    lut[x] = v;
  }
}

static inline uint32_t kw_low_bits(const uint64_t* w, int shift_digits, uint32_t mask, int kw)
{ /* (value >> 2*shift_digits) & mask, mask < 2^32 */
  int word = shift_digits >> 5, off = (shift_digits & 31) * 2;
  uint64_t v = w[word] >> off;
  if (off && word + 1 < kw) v |= w[word + 1] << (64 - off);
  return (uint32_t)v & mask;
}

uint32_t orc_minimizer_of(const uint64_t* fwd, int k, int kw, int m, const uint32_t* lut)
{ /* gatb Model.hpp:1254-1287: scan from the rightmost m-mer, strict '<' */
  uint32_t mask = (uint32_t)(((uint64_t)1 << (2 * m)) - 1);
  uint32_t best = mask;
  for (int s = 0; s <= k - m; s++) {
    uint32_t cand = lut[kw_low_bits(fwd, s, mask, kw)];
    if (cand < best) best = cand;
  }
  return best;
}

void orc_repart_static(int m, uint32_t nb_parts, uint16_t* table)
{ /* include/kmtricks/repartition.hpp:45-56 */
  uint64_t n = (uint64_t)1 << (2 * m);
  for (uint64_t x = 0; x < n; x++) {
    uint32_t mm = (uint32_t)x;
    table[x] = (uint16_t)(orc_xxh64(&mm, sizeof(mm), 0) % nb_parts);
  }
}

/* ===================================================================== */
/* super-k-mer partitioner */

static int buf_reserve(orc_buf* b, size_t extra)
{
  if (b->len + extra <= b->cap) return 0;
  size_t nc = b->cap ? b->cap * 2 : 4096;
  while (nc < b->len + extra) nc *= 2;
  uint8_t* nd = (uint8_t*)realloc(b->data, nc);
  if (!nd) return -1;
  b->data = nd; b->cap = nc;
  return 0;
}
void orc_buf_free(orc_buf* b) { free(b->data); b->data = NULL; b->len = b->cap = 0; }

#define SK_MAX 255
typedef struct {
  uint64_t fwd[SK_MAX + 1][4]; /* forward value of each k-mer (up to Kmer<128>: four words) */
  uint8_t  which[SK_MAX + 1];  /* forward < revcomp */
  uint64_t canon[SK_MAX + 1][4];
  int n;
  uint64_t minimizer;
  int valid;                   /* minimizer != DEFAULT_MINIMIZER */
} superk_t;

static inline unsigned radix_of(const uint64_t* v, int k, int kw)
{ /* top 4 nucleotides, fill_partitions.hpp:52-56 (getHeavyWeight) */
  return kw_low_bits(v, k - 4, 255u, kw);
}

/* per-minimizer records of the call in progress (orc_superk_partition_stats): [0] nb_superks, [1] nb_kmers,
 * [2] nb_kxmers, each 4^m entries; PartiInfo::incSuperKmer_per_minimBin / incKxmer_per_minimBin */
static __thread uint64_t* g_mstats[3] = {0, 0, 0};

static int superk_flush(superk_t* sk, int k, int kw, const uint16_t* repart, uint32_t nb_parts,
                        orc_buf* out, uint64_t* pinfo)
{ /* fill_partitions.hpp:59-105 + gatb Model.hpp:1388-1433 (SuperKmer::save) */
  if (!sk->valid || sk->n == 0) return 0;
  if (g_mstats[0]) { g_mstats[0][sk->minimizer] += 1; g_mstats[1][sk->minimizer] += (uint64_t)sk->n; }
  if (g_mstats[2]) { /* SampleRepart::processSuperkmer, gatb RepartitionAlgorithm.cpp:182-215 */
    int prev = sk->which[0]; int kx = 0;
    for (int ii = 1; ii < sk->n; ii++) {
      if (sk->which[ii] != prev || kx >= 4) { g_mstats[2][sk->minimizer] += 1; kx = 0; } else kx++;
      prev = sk->which[ii];
    }
    g_mstats[2][sk->minimizer] += 1;
  }
  if (!out) return 0;                                       /* statistics only */
  uint32_t p = repart[sk->minimizer];
  if (p >= nb_parts) return -2;
  orc_buf* b = &out[p];
  int n = sk->n;
  size_t nbytes = (size_t)(n + k + 3) / 4;
  if (buf_reserve(b, nbytes + 2)) return -1;
  uint8_t* d = b->data + b->len;
  size_t idx = 0;
  d[idx++] = (uint8_t)n;
  uint64_t base[4] = {0, 0, 0, 0};
  for (int w = 0; w < kw; w++) base[w] = sk->fwd[0][w];
  int rem = k;
  while (rem >= 4) {
    d[idx++] = (uint8_t)(base[0] & 255u);
    for (int w = 0; w < 3; w++) base[w] = (base[w] >> 8) | (base[w + 1] << 56);
    base[3] >>= 8;
    rem -= 4;
  }
  uint8_t nb = (uint8_t)(base[0] & 255u);
  int uid = rem, skid = 1;
  for (;;) {
    while (uid < 4 && skid < n) {
      nb |= (uint8_t)((sk->fwd[skid][0] & 3u) << (uid * 2));
      uid++; skid++;
    }
    if (uid > 0) d[idx++] = nb;
    if (skid >= n) break;
    nb = 0; uid = 0;
  }
  b->len += idx;
  b->nb_kmers += (uint64_t)n;
  b->nb_superk += 1;

  if (pinfo) { /* PartiInfo<5> kx-mer / radix counters, fill_partitions.hpp:67-102 */
    uint64_t* pi = pinfo + (size_t)p * (2 + 5 * 256);
    int prev_which = sk->which[0];
    int kx = 0;
    unsigned radix_fwd = radix_of(sk->canon[0], k, kw), radix;
    for (int ii = 1; ii < n; ii++) {
      if (sk->which[ii] != prev_which || kx >= 4) {
        radix = prev_which ? radix_fwd : radix_of(sk->canon[ii - 1], k, kw);
        pi[0] += (uint64_t)kx + 1; pi[1] += 1; pi[2 + kx * 256 + radix] += 1;
        radix_fwd = radix_of(sk->canon[ii], k, kw);
        kx = 0;
      } else kx++;
      prev_which = sk->which[ii];
    }
    radix = prev_which ? radix_fwd : radix_of(sk->canon[n - 1], k, kw);
    pi[0] += (uint64_t)kx + 1; pi[1] += 1; pi[2 + kx * 256 + radix] += 1;
  }
  return 0;
}

int orc_superk_partition(const char* seq, size_t len, int k, int m,
                         const uint32_t* lut, const uint16_t* repart,
                         uint32_t nb_parts, orc_buf* out, uint64_t* pinfo)
{
  if (k < m || k > 127 || m > 15) return -3;
  if ((int64_t)len - k + 1 <= 0) return 0;                 /* Sequence2SuperKmer.hpp:143-144 */
  const int kw = kw_of_k(k);
  /* Type::getSize(): the reference dispatches k to the first KMER_LIST entry with k < entry
   * (include/kmtricks/loop_executor.hpp:47-52), so k = 32 runs as Kmer<64> (128 bits) */
  const int span_bits = k < 32 ? 64 : k < 64 ? 128 : k < 96 ? 192 : 256;   /* (Kmer<96>: 92 k-mers a super-k-mer, Kmer<128>: 124) */
  int maxs = (span_bits - 8) / 2; if (maxs > 255) maxs = 255; /* Sequence2SuperKmer.hpp:146 */
  const int nbm = k - m + 1;                                /* _nbMinimizers */
  const uint32_t maskm = (uint32_t)(((uint64_t)1 << (2 * m)) - 1);
  const uint64_t DEFAULT_MIN = 1000000000ULL;               /* Model.hpp:1349 */

  uint64_t kmask[4] = {0, 0, 0, 0};
  for (int w = 0; w < kw; w++) {
    const int bits = 2 * k - 64 * w;
    kmask[w] = bits >= 64 ? ~0ULL : bits > 0 ? (((uint64_t)1 << bits) - 1) : 0;
  }

  superk_t* sk = (superk_t*)calloc(1, sizeof(superk_t));
  if (!sk) return -1;
  sk->minimizer = DEFAULT_MIN; sk->valid = 0;

  uint64_t fwd[4] = {0, 0, 0, 0}, rev[4] = {0, 0, 0, 0};
  int bad = -1;
  /* first k-mer: polynom, Model.hpp:636-657, 857-866 */
  for (int i = 0; i < k; i++) {
    unsigned c = nt_code((unsigned char)seq[i]);
    for (int w = 3; w > 0; w--) fwd[w] = (fwd[w] << 2) | (fwd[w - 1] >> 62);
    fwd[0] = (fwd[0] << 2) + c;
    if (!orc_nt_valid((unsigned char)seq[i])) bad = i;
  }
  for (int w = 0; w < 4; w++) fwd[w] &= kmask[w];
  orc_revcomp(fwd, rev, k, kw);
  /* minimizer state, Model.hpp:1254-1287 */
  uint32_t minim = orc_minimizer_of(fwd, k, kw, m, lut);
  int pos = -1;
  { /* position of the rightmost-scanned winner (kept for the window rule below) */
    uint32_t best = maskm;
    for (int idx = nbm - 1, s = 0; idx >= 0; idx--, s++) {
      uint32_t cand = lut[kw_low_bits(fwd, s, maskm, kw)];
      if (cand < best) { best = cand; pos = idx; }
    }
  }
  int rc = 0;
  size_t idx = (size_t)k;
  int is_valid = bad < 0;
  for (;;) {
    /* ---- KmerFunctor, Sequence2SuperKmer.hpp:90-132 ---- */
    if (!is_valid) {
      if ((rc = superk_flush(sk, k, kw, repart, nb_parts, out, pinfo))) break;
      sk->n = 0; sk->minimizer = DEFAULT_MIN; sk->valid = 0;
    } else {
      uint64_t h = minim;
      if (!sk->valid) { sk->minimizer = h; sk->valid = 1; }
      if (h != sk->minimizer || sk->n >= maxs) {
        if ((rc = superk_flush(sk, k, kw, repart, nb_parts, out, pinfo))) break;
        sk->n = 0;
      }
      sk->minimizer = h; sk->valid = 1;
      int w = kw_less(fwd, rev, kw);
      sk->which[sk->n] = (uint8_t)w;
      for (int q = 0; q < 4; q++) { sk->fwd[sk->n][q] = fwd[q]; sk->canon[sk->n][q] = w ? fwd[q] : rev[q]; }
      sk->n++;
    }
    if (idx >= len) break;
    /* ---- next k-mer, Model.hpp:740-757, 868-884, 1106-1139 ---- */
    unsigned char ch = (unsigned char)seq[idx++];
    unsigned c = nt_code(ch);
    if (!orc_nt_valid(ch)) bad = k - 1; else bad--;
    is_valid = bad < 0;
    for (int w = 3; w > 0; w--) fwd[w] = ((fwd[w] << 2) | (fwd[w - 1] >> 62)) & kmask[w];
    fwd[0] = ((fwd[0] << 2) + c) & kmask[0];
    { /* rev = (rev >> 2) + comp(c) << 2(k-1) */
      uint64_t cc = (uint64_t)(c ^ 2u);
      for (int w = 0; w < 3; w++) rev[w] = (rev[w] >> 2) | (rev[w + 1] << 62);
      rev[3] >>= 2;
      int sh = 2 * (k - 1);
      rev[sh >> 6] |= cc << (sh & 63);
      for (int w = 0; w < 4; w++) rev[w] &= kmask[w];
    }
    uint32_t mmer = lut[(uint32_t)fwd[0] & maskm];
    pos--;
    if (mmer < minim) { minim = mmer; pos = nbm - 1; }
    else if (pos < 0) {
      uint32_t best = maskm; pos = -1;
      for (int ii = nbm - 1, s = 0; ii >= 0; ii--, s++) {
        uint32_t cand = lut[kw_low_bits(fwd, s, maskm, kw)];
        if (cand < best) { best = cand; pos = ii; }
      }
      minim = best;
    }
  }
  if (!rc) rc = superk_flush(sk, k, kw, repart, nb_parts, out, pinfo); /* Sequence2SuperKmer.hpp:155 */
  free(sk);
  return rc;
}

int orc_superk_partition_stats(const char* seq, size_t len, int k, int m,
                               const uint32_t* lut, const uint16_t* repart,
                               uint32_t nb_parts, orc_buf* out, uint64_t* pinfo,
                               uint64_t* minim_superks, uint64_t* minim_kmers, uint64_t* minim_kxmers)
{
  g_mstats[0] = minim_superks; g_mstats[1] = minim_kmers; g_mstats[2] = minim_kxmers;
  int rc = orc_superk_partition(seq, len, k, m, lut, repart, nb_parts, out, pinfo);
  g_mstats[0] = g_mstats[1] = g_mstats[2] = 0;
  return rc;
}


/* ===================================================================== */
/* count */

uint64_t orc_superk_decode(const uint8_t* recs, size_t len, int k, int kw, uint64_t* out)
{ /* sorting_count.hpp:141-312 (decode), canonical = min(fwd, revcomp); keys of up to four words (Kmer<128>) */
  uint64_t kmask[4] = {0, 0, 0, 0};
  for (int w = 0; w < kw; w++) {
    const int bits = 2 * k - 64 * w;
    kmask[w] = bits >= 64 ? ~0ULL : bits > 0 ? (((uint64_t)1 << bits) - 1) : 0;
  }
  const uint8_t* p = recs; const uint8_t* end = recs + len;
  uint64_t n = 0;
  while (p < end) {
    unsigned nbk = *p++;
    uint64_t seed[4] = {0, 0, 0, 0};
    int rem = k, nbr = 0;
    uint8_t nb = 0;
    while (rem >= 4) {
      nb = *p++;
      seed[nbr >> 3] |= (uint64_t)nb << (8 * (nbr & 7));
      rem -= 4; nbr++;
    }
    int uid = 4;
    if (rem > 0) {
      nb = *p++;
      seed[nbr >> 3] |= (uint64_t)nb << (8 * (nbr & 7));
      uid = rem;
    }
    uint64_t fwd[4], rev[4];
    for (int w = 0; w < 4; w++) fwd[w] = w < kw ? seed[w] & kmask[w] : 0;
    orc_revcomp(fwd, rev, k, kw);
    for (int w = kw; w < 4; w++) rev[w] = 0;
    for (unsigned ii = 0; ii < nbk; ii++) {
      if (out) {
        const uint64_t* c = kw_less(fwd, rev, kw) ? fwd : rev;
        for (int w = 0; w < kw; w++) out[n * kw + w] = c[w];
      }
      n++;
      if (ii + 1 >= nbk) break;
      if (uid >= 4) { nb = *p++; uid = 0; }
      unsigned nt = (nb >> (2 * uid)) & 3u; uid++;
      for (int w = kw - 1; w > 0; w--) fwd[w] = ((fwd[w] << 2) | (fwd[w - 1] >> 62)) & kmask[w];
      fwd[0] = ((fwd[0] << 2) | nt) & kmask[0];
      uint64_t cc = (uint64_t)(nt ^ 2u);
      for (int w = 0; w + 1 < kw; w++) rev[w] = (rev[w] >> 2) | (rev[w + 1] << 62);
      rev[kw - 1] >>= 2;
      int sh = 2 * (k - 1);
      rev[sh >> 6] |= cc << (sh & 63);
      for (int w = 0; w < kw; w++) rev[w] &= kmask[w];
    }
  }
  return n;
}

static int cmp_u64(const void* a, const void* b)
{ uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : x > y; }
static int cmp_u128(const void* a, const void* b)
{
  const uint64_t* x = (const uint64_t*)a; const uint64_t* y = (const uint64_t*)b;
  if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
  return x[0] < y[0] ? -1 : x[0] > y[0];
}

static int cmp_u192(const void* a, const void* b)
{
  const uint64_t* x = (const uint64_t*)a; const uint64_t* y = (const uint64_t*)b;
  for (int w = 2; w > 0; w--) if (x[w] != y[w]) return x[w] < y[w] ? -1 : 1;
  return x[0] < y[0] ? -1 : x[0] > y[0];
}
static int cmp_u256(const void* a, const void* b)
{
  const uint64_t* x = (const uint64_t*)a; const uint64_t* y = (const uint64_t*)b;
  for (int w = 3; w > 0; w--) if (x[w] != y[w]) return x[w] < y[w] ? -1 : 1;
  return x[0] < y[0] ? -1 : x[0] > y[0];
}

static int rle_filter(uint64_t* arr, uint64_t n, int kw, uint32_t hard_min,
                      uint64_t** keys, uint32_t** counts, uint64_t* n_out)
{ /* sorting_count.hpp:971-990 run-length; count_processor.hpp:61-70, 135-146 filter+saturate */
  uint64_t* ok = (uint64_t*)malloc((n ? n : 1) * (size_t)kw * 8);
  uint32_t* oc = (uint32_t*)malloc((n ? n : 1) * 4);
  if (!ok || !oc) { free(ok); free(oc); return -1; }
  uint64_t d = 0, i = 0;
  while (i < n) {
    uint64_t j = i + 1;
    while (j < n && kw_eq(arr + i * kw, arr + j * kw, kw)) j++;
    uint64_t c = j - i;
    if (c >= hard_min) {
      for (int w = 0; w < kw; w++) ok[d * kw + w] = arr[i * kw + w];
      oc[d] = c >= 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)c;
      d++;
    }
    i = j;
  }
  *keys = ok; *counts = oc; *n_out = d;
  return 0;
}

int orc_count_kmer(const uint8_t* recs, size_t len, int k, uint32_t hard_min,
                   uint64_t** keys, uint32_t** counts, uint64_t* n_out)
{
  int kw = kw_of_k(k);
  uint64_t n = orc_superk_decode(recs, len, k, kw, NULL);
  uint64_t* arr = (uint64_t*)malloc((n ? n : 1) * (size_t)kw * 8);
  if (!arr) return -1;
  orc_superk_decode(recs, len, k, kw, arr);
  qsort(arr, n, (size_t)kw * 8, kw == 1 ? cmp_u64 : kw == 2 ? cmp_u128 : kw == 3 ? cmp_u192 : cmp_u256);
  int rc = rle_filter(arr, n, kw, hard_min, keys, counts, n_out);
  free(arr);
  return rc;
}

int orc_count_hash(const uint8_t* recs, size_t len, int k, uint64_t win, uint64_t part,
                   uint32_t hard_min, uint64_t** hashes, uint32_t** counts, uint64_t* n_out)
{
  int kw = kw_of_k(k);
  uint64_t n = orc_superk_decode(recs, len, k, kw, NULL);
  uint64_t* arr = (uint64_t*)malloc((n ? n : 1) * (size_t)kw * 8);
  uint64_t* h = (uint64_t*)malloc((n ? n : 1) * 8);
  if (!arr || !h) { free(arr); free(h); return -1; }
  orc_superk_decode(recs, len, k, kw, arr);
  for (uint64_t i = 0; i < n; i++) /* KmXXHash, sorting_count.hpp:350-357 */
    h[i] = orc_xxh64(arr + i * kw, (size_t)kw * 8, 0) % win + win * part;
  free(arr);
  qsort(h, n, 8, cmp_u64);
  int rc = rle_filter(h, n, 1, hard_min, hashes, counts, n_out);
  free(h);
  return rc;
}

/* ===================================================================== */
/* merge -- include/kmtricks/merge.hpp:150-260 (KmerMerger) / 410-517 (HashMerger).
 * Restated as the reference does it: for every distinct key a linear scan
 * over all N streams. */

int orc_merge(const orc_list* lists, uint32_t N, int kw,
              const uint32_t* soft_min, uint32_t rec_min, uint32_t share_min,
              orc_row_cb cb, void* user, uint64_t* stats)
{
  uint64_t* cur = (uint64_t*)calloc(N ? N : 1, 8);
  uint32_t* counts = (uint32_t*)calloc(N ? N : 1, 4);
  uint32_t* need = (uint32_t*)malloc((N ? N : 1) * 4);
  if (!cur || !counts || !need) { free(cur); free(counts); free(need); return -1; }
  if (stats) memset(stats, 0, (size_t)6 * N * 8);
  uint64_t* ns = stats, *rd = stats ? stats + N : 0, *uwo = stats ? stats + 2 * (size_t)N : 0,
          *uw = stats ? stats + 3 * (size_t)N : 0, *two = stats ? stats + 4 * (size_t)N : 0,
          *tw = stats ? stats + 5 * (size_t)N : 0;
  uint64_t current[4] = {0, 0, 0, 0}, next[4] = {0, 0, 0, 0};      /* (Kmer<128>: four words) */
  int next_set = 0;
  /* init_state: smallest head, merge.hpp:150-168 */
  for (uint32_t i = 0; i < N; i++) {
    if (lists[i].n == 0) continue;
    const uint64_t* v = lists[i].keys;
    if (!next_set || kw_less(v, next, kw)) { for (int w = 0; w < kw; w++) next[w] = v[w]; next_set = 1; }
  }
  while (next_set) {
    for (int w = 0; w < kw; w++) current[w] = next[w];
    next_set = 0;
    uint32_t recurrence = 0, solid_in = 0, n_need = 0;
    for (uint32_t i = 0; i < N; i++) {
      const orc_list* L = &lists[i];
      if (cur[i] < L->n && kw_eq(L->keys + cur[i] * kw, current, kw)) {
        counts[i] = L->counts[cur[i]];
        if (counts[i] >= soft_min[i]) {
          recurrence++; solid_in++;
          if (stats) { two[i] += counts[i]; tw[i] += counts[i]; uw[i]++; uwo[i]++; }
        } else {
          if (stats) ns[i]++;
          if (share_min) need[n_need++] = i; else counts[i] = 0;
        }
        cur[i]++;
      } else counts[i] = 0;
      if (cur[i] < L->n) {
        const uint64_t* v = L->keys + cur[i] * kw;
        if (!next_set || kw_less(v, next, kw)) { for (int w = 0; w < kw; w++) next[w] = v[w]; next_set = 1; }
      }
    }
    for (uint32_t j = 0; j < n_need; j++) { /* rescue, merge.hpp:234-247 */
      uint32_t f = need[j];
      if (!(solid_in >= share_min)) counts[f] = 0;
      else if (stats) { rd[f]++; uw[f]++; tw[f] += counts[f]; }
    }
    cb(user, current, counts, recurrence >= rec_min);
  }
  free(cur); free(counts); free(need);
  return 0;
}

/* ---- row encoders ------------------------------------------------------ */

uint32_t orc_to_n_b(uint32_t c, uint32_t max_width)
{ /* packc.hpp:26-35 */
  if (c) {
    uint64_t r = 32 - (uint64_t)__builtin_clz(c);
    uint64_t cap = ((uint64_t)1 << max_width) - 1;
    return (uint32_t)(r > cap ? cap : r);
  }
  return 0;
}
uint64_t orc_byte_count_pack(uint64_t n, uint64_t bits) { return (n * bits + 7) >> 3; } /* packc.hpp:18-21 */

static void pack_insert_msb(uint8_t* buf, uint64_t offset, unsigned size, uint32_t v)
{ /* bitpacker::insert (thirdparty/bitpacker/include/bitpacker/bitpacker.hpp:190-233):
   * bit `offset` counts from the MSB of byte 0 */
  for (unsigned b = 0; b < size; b++) {
    uint64_t bit = offset + b;
    unsigned val = (v >> (size - 1 - b)) & 1u;
    uint8_t m = (uint8_t)(0x80u >> (bit & 7));
    if (val) buf[bit >> 3] |= m; else buf[bit >> 3] &= (uint8_t)~m;
  }
}

typedef struct {
  int mode; uint32_t N; int kw; int bitw;
  uint64_t lower, upper, current; /* BF window cursor, merge.hpp:579-599 */
  uint8_t* body; uint64_t len, cap; uint64_t rows; uint64_t rowbytes;
  int err;
} mat_ctx;

static int mat_reserve(mat_ctx* c, uint64_t extra)
{
  if (c->len + extra <= c->cap) return 0;
  uint64_t nc = c->cap ? c->cap * 2 : 65536;
  while (nc < c->len + extra) nc *= 2;
  uint8_t* nb = (uint8_t*)realloc(c->body, nc);
  if (!nb) { c->err = -1; return -1; }
  c->body = nb; c->cap = nc;
  return 0;
}

static void mat_row(void* user, const uint64_t* key, const uint32_t* counts, int keep)
{
  mat_ctx* c = (mat_ctx*)user;
  if (c->err) return;
  if (c->mode == ORC_MODE_COUNT || c->mode == ORC_MODE_PA) {
    if (!keep) return;
    uint64_t rb = (uint64_t)c->kw * 8 + c->rowbytes;
    if (mat_reserve(c, rb)) return;
    uint8_t* d = c->body + c->len;
    memcpy(d, key, (size_t)c->kw * 8); d += c->kw * 8;
    if (c->mode == ORC_MODE_COUNT) memcpy(d, counts, (size_t)c->N * 4);
    else { /* set_bit_vector, utils.hpp:104-116 */
      memset(d, 0, c->rowbytes);
      for (uint32_t i = 0; i < c->N; i++) if (counts[i]) d[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    c->len += rb; c->rows++;
    return;
  }
  /* BF / BFC: merge.hpp:581-594, 610-623 */
  uint64_t h = key[0];
  while (h > c->current) {
    if (mat_reserve(c, c->rowbytes)) return;
    memset(c->body + c->len, 0, c->rowbytes); c->len += c->rowbytes; c->current++; c->rows++;
  }
  if (keep) {
    if (mat_reserve(c, c->rowbytes)) return;
    uint8_t* d = c->body + c->len;
    memset(d, 0, c->rowbytes);
    if (c->mode == ORC_MODE_BFC)
      for (uint32_t i = 0; i < c->N; i++) pack_insert_msb(d, (uint64_t)i * c->bitw, (unsigned)c->bitw, orc_to_n_b(counts[i], (uint32_t)c->bitw));
    else
      for (uint32_t i = 0; i < c->N; i++) if (counts[i]) d[i >> 3] |= (uint8_t)(1u << (i & 7));
    c->len += c->rowbytes; c->rows++;
    c->current = h + 1;
  }
}

int orc_merge_matrix(const orc_list* lists, uint32_t N, int kw,
                     const uint32_t* soft_min, uint32_t rec_min, uint32_t share_min,
                     int mode, uint64_t lower, uint64_t upper, int bitw,
                     uint8_t** body, uint64_t* body_len, uint64_t* rows_out, uint64_t* stats)
{
  mat_ctx c; memset(&c, 0, sizeof(c));
  c.mode = (mode == ORC_MODE_BFT) ? ORC_MODE_BF : mode;
  c.N = N; c.kw = kw; c.bitw = bitw; c.lower = lower; c.upper = upper; c.current = lower;
  if (c.mode == ORC_MODE_COUNT) c.rowbytes = (uint64_t)N * 4;
  else if (c.mode == ORC_MODE_BFC) c.rowbytes = orc_byte_count_pack(N, (uint64_t)bitw);
  else c.rowbytes = ((uint64_t)N + 7) / 8;
  int rc = orc_merge(lists, N, kw, soft_min, rec_min, share_min, mat_row, &c, stats);
  if (rc || c.err) { free(c.body); return rc ? rc : c.err; }
  if (c.mode == ORC_MODE_BF || c.mode == ORC_MODE_BFC) {
    while (c.current <= upper) { /* merge.hpp:595-599 */
      if (mat_reserve(&c, c.rowbytes)) { free(c.body); return -1; }
      memset(c.body + c.len, 0, c.rowbytes); c.len += c.rowbytes; c.current++; c.rows++;
    }
  }
  if (mode == ORC_MODE_BFT) { /* merge.hpp:631-644 */
    uint64_t W = upper - lower + 1;
    uint64_t nr = (W + 7) / 8 * 8, ncb = (((uint64_t)N + 7) / 8 * 8) / 8; /* BitMatrix(n, m bytes) */
    uint8_t* in = (uint8_t*)calloc((nr * ncb) != 0 ? nr * ncb : 1, 1);
    uint8_t* out = (uint8_t*)calloc((nr * ncb) != 0 ? nr * ncb : 1, 1);
    if (!in || !out) { free(in); free(out); free(c.body); return -1; }
    memcpy(in, c.body, (size_t)(W * c.rowbytes)); /* VectorMatrixReader::load: W rows of ncb bytes */
    orc_transpose_bits(in, out, nr, ncb * 8);
    free(in); free(c.body);
    c.body = out; c.len = nr * ncb; c.rows = ncb * 8;
  }
  if (!c.body) c.body = (uint8_t*)malloc(1);
  *body = c.body; *body_len = c.len; *rows_out = c.rows;
  return 0;
}

/* ===================================================================== */
/* bit transpose, bitmatrix.hpp:238-289 (semantics: out[c][r] = in[r][c]) */

void orc_transpose_bits(const uint8_t* in, uint8_t* out, uint64_t nrows, uint64_t ncols)
{
  uint64_t in_stride = ncols / 8, out_stride = nrows / 8;
  memset(out, 0, (size_t)(ncols * out_stride));
  for (uint64_t r = 0; r < nrows; r++)
    for (uint64_t cb = 0; cb < in_stride; cb++) {
      uint8_t v = in[r * in_stride + cb];
      while (v) {
        int b = __builtin_ctz(v); v &= (uint8_t)(v - 1);
        uint64_t c = cb * 8 + (uint64_t)b;
        out[c * out_stride + (r >> 3)] |= (uint8_t)(1u << (r & 7));
      }
    }
}

/* ---- the abundance histogram of a sample: KHist::inc (histogram.hpp:48-68), called for every distinct k-mer / hash before
 *      the hard-min filter (count_processor.hpp:61, 135).  uniq_bins / total_bins hold upper - lower + 1 entries;
 *      oob = {lower unique, upper unique, lower total, upper total}; sums = {unique, total}. ---- */
void orc_khist(const uint32_t* counts, uint64_t n, uint64_t lower, uint64_t upper, uint64_t* uniq_bins, uint64_t* total_bins,
               uint64_t* oob, uint64_t* sums)
{
  for (uint64_t i = 0; i < n; i++) {
    const uint64_t c = counts[i];
    sums[0]++; sums[1] += c;
    if (c < lower) { oob[0]++; oob[2] += c; }
    else if (c > upper) { oob[1]++; oob[3] += c; }
    else { uniq_bins[c - lower]++; total_bins[c - lower] += c; }
  }
}

void orc_free(void* p) { free(p); }
