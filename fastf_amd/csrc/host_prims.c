/*
 * host_prims.c — host-side primitives of the bam2db path (plain C).
 *
 *   fastf_mt_*            explicit-state MT19937, same stream as the reference's
 *                         global generator (mt19937ar.c:60-73, 105-140)
 *   fastf_draw_threshold  integer form of the depth test (bam2db_ds.c:385-390)
 *   fastf_sample_cells    cell sub-sampling (bam2db_ds.c:240-244, utils.c:29-75)
 *   fastf_pack_umi        2-bit UMI codec into a u32 (bam2db_ds.c:5-51, :419)
 *   fastf_keydict_*       exact string → 64-bit key packing; stands in for the
 *                         strcmp-verified chained table (hashtable.c:70-115)
 */
#include "fastf_amd.h"
#include "host_io.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* MT19937                                                             */
/* ------------------------------------------------------------------ */
enum { MT_LEN = 624, MT_GAP = 397 };

void fastf_mt_seed(fastf_mt_t *mt, uint32_t seed)
{
    uint32_t x = seed;
    mt->s[0] = x;
    for (uint32_t i = 1; i < MT_LEN; i++) {
        x = 1812433253u * (x ^ (x >> 30)) + i;
        mt->s[i] = x;
    }
    mt->idx = MT_LEN;                 /* state exhausted: twist before the first draw */
}

static inline uint32_t mt_mix(uint32_t hi, uint32_t lo)
{
    uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

static void mt_twist(fastf_mt_t *mt)
{
    uint32_t *s = mt->s;
    int k = 0;
    for (; k < MT_LEN - MT_GAP; k++) s[k] = s[k + MT_GAP] ^ mt_mix(s[k], s[k + 1]);
    for (; k < MT_LEN - 1; k++)      s[k] = s[k + MT_GAP - MT_LEN] ^ mt_mix(s[k], s[k + 1]);
    s[MT_LEN - 1] = s[MT_GAP - 1] ^ mt_mix(s[MT_LEN - 1], s[0]);
    mt->idx = 0;
}

static inline uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

uint32_t fastf_mt_next(fastf_mt_t *mt)
{
    if (mt->idx >= MT_LEN) mt_twist(mt);
    return mt_temper(mt->s[mt->idx++]);
}

void fastf_mt_fill(fastf_mt_t *mt, uint32_t *out, size_t n)
{
    size_t done = 0;
    while (done < n) {
        if (mt->idx >= MT_LEN) mt_twist(mt);
        size_t take = (size_t)(MT_LEN - mt->idx);
        if (take > n - done) take = n - done;
        const uint32_t *src = mt->s + mt->idx;
        for (size_t i = 0; i < take; i++) out[done + i] = mt_temper(src[i]);
        mt->idx += (int)take;
        done += take;
    }
}

void fastf_mt_skip(fastf_mt_t *mt, uint64_t n)
{
    while (n) {
        if (mt->idx >= MT_LEN) mt_twist(mt);
        uint64_t take = (uint64_t)(MT_LEN - mt->idx);
        if (take > n) take = n;
        mt->idx += (int)take;
        n -= take;
    }
}

/* ------------------------------------------------------------------ */
/* depth test as an integer threshold                                  */
/* ------------------------------------------------------------------ */
/* The reference drops a record when draw*(1.0/4294967295.0) >= (double)rate
 * (mt19937ar.c:151, bam2db_ds.c:387).  The left side is monotone in draw, so the
 * kept set is {draw < T}; T is found by bisection on the very same expression. */
static int depth_drop(uint64_t draw, float rate)
{
    double r = (uint32_t)draw * (1.0 / 4294967295.0);
    return r >= rate;
}

uint64_t fastf_draw_threshold(float rate_depth)
{
    if (!depth_drop(0xFFFFFFFFull, rate_depth)) return 1ull << 32;   /* nothing dropped (rate > 1, NaN) */
    if (depth_drop(0, rate_depth)) return 0;                        /* everything dropped (rate <= 0)  */
    uint64_t lo = 0, hi = 0xFFFFFFFFull;   /* kept(lo), dropped(hi) */
    while (hi - lo > 1) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (depth_drop(mid, rate_depth)) hi = mid; else lo = mid;
    }
    return hi;
}

/* ------------------------------------------------------------------ */
/* cell sub-sampling                                                   */
/* ------------------------------------------------------------------ */
static int cmp_line(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

int fastf_sample_cells(size_t n_cells, float rate_cell, unsigned int seed,
                       uint64_t *lines_out, size_t *n_sampled, uint64_t *draws_used)
{
    if (!(rate_cell >= 0.0f)) return 1;
    /* size_t * float is float arithmetic in C (bam2db_ds.c:241) */
    float prod = (float)n_cells * rate_cell;
    if (!(prod < 18446744073709551616.0f)) return 1;
    size_t want = (size_t)prod;
    if (want > n_cells) return 1;                 /* utils.c:38-43: the reference exits */
    *n_sampled = want;
    *draws_used = 0;
    if (want == n_cells) {                        /* utils.c:48-51: identity, no draw   */
        for (size_t i = 0; i < n_cells; i++) lines_out[i] = i;
        return 0;
    }
    uint64_t *pool = (uint64_t *)malloc((n_cells ? n_cells : 1) * sizeof *pool);
    if (!pool) return 1;
    for (size_t i = 0; i < n_cells; i++) pool[i] = i;
    fastf_mt_t mt;
    fastf_mt_seed(&mt, seed);                     /* utils.c:32 re-seeds */
    size_t live = n_cells;
    for (size_t i = 0; i < want; i++, live--) {   /* swap-with-last draw without replacement */
        size_t pick = fastf_mt_next(&mt) % live;
        lines_out[i] = pool[pick];
        pool[pick] = pool[live - 1];
    }
    free(pool);
    *draws_used = want;
    qsort(lines_out, want, sizeof *lines_out, cmp_line);
    return 0;
}

/* ------------------------------------------------------------------ */
/* UMI codec                                                           */
/* ------------------------------------------------------------------ */
/* base → 2-bit code, 4 = not a base (encode_base, bam2db_ds.c:5-20: upper-case ACGT only) */
static const uint8_t k_base_code[256] = {
#define X4 4, 4, 4, 4
#define X16 X4, X4, X4, X4
    X16, X16, X16, X16,                                  /* 0x00-0x3f */
    4, 0, 4, 1, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4,      /* @ A B C D E F G H I J K L M N O */
    4, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,      /* P Q R S T ...                   */
    X16, X16, X16, X16, X16, X16, X16, X16, X16, X16
#undef X16
#undef X4
};

uint32_t fastf_pack_umi(const char *ub, size_t len, uint32_t *umi_out)
{
    uint32_t meta = FASTF_META_HAS_UB;
    *umi_out = 0;
    if (len > 16) return meta | FASTF_META_UMI_TOOLONG;
    uint32_t packed = 0, bad = 0;
    for (size_t i = 0; i < len; i++) {
        const uint32_t code = k_base_code[(unsigned char)ub[i]];
        bad |= code;
        packed = (packed << 2) | (code & 3);
    }
    if (len) packed <<= 32 - 2 * len;
    const int ok = !(bad & 4);
    if (ok) { meta |= FASTF_META_UMI_NONNULL; *umi_out = packed; }
    meta |= (uint32_t)((len + 3) / 4) << FASTF_META_LEN_SHIFT;
    return meta;
}

uint32_t fastf_pack_umi_long(const char *ub, size_t len, uint32_t *umi_out, uint32_t *ext_out)
{
    uint32_t meta = FASTF_META_HAS_UB;
    *umi_out = 0; *ext_out = 0;
    if (len > 32) return meta | FASTF_META_UMI_TOOLONG;      /* (8 blob bytes: two 32-bit words of bases) */
    uint64_t packed = 0; uint32_t bad = 0;
    for (size_t i = 0; i < len; i++) {
        const uint32_t code = k_base_code[(unsigned char)ub[i]];
        bad |= code;
        packed = (packed << 2) | (code & 3);
    }
    if (len) packed <<= 64 - 2 * len;
    if (!(bad & 4)) { meta |= FASTF_META_UMI_NONNULL; *umi_out = (uint32_t)(packed >> 32); *ext_out = (uint32_t)packed; }
    meta |= (uint32_t)((len + 3) / 4) << FASTF_META_LEN_SHIFT;
    return meta;
}

/* ------------------------------------------------------------------ */
/* exact string → key                                                  */
/* ------------------------------------------------------------------ */
/*
 * key[63:62] = form
 *   1  DNA form    [ACGT]{1,24} ( '-' canonical-decimal 0..254 )?
 *                  [61:57] length  [56:49] suffix+1 (0 = none)  [47:0] bases, first base on top
 *   2  ID form     <prefix><1..13 digits>; prefix = everything before the trailing digit run
 *                  [61:48] prefix id  [47:44] digit count  [43:0] value
 *   3  bit 61 = 0: escape   [60:0] ordinal of the registered string (anything else that was registered)
 *      bit 61 = 1: DNA+N    [ACGTN]{1,16} with at least one N ( '-' canonical-decimal 0..254 )?   (raw CR/UR tags)
 *                  [60:56] length  [55:48] suffix+1  [47:0] 3 bits per base, first base on top (N = 4)
 *   0  (whole key 0) cannot equal any registered string
 * Classification depends on the string alone (plus the append-only prefix table), so a
 * registered string and an equal tag string always take the same route.
 */
typedef struct { char *s; uint32_t len; uint64_t val; } kd_ent;
typedef struct { kd_ent *e; size_t cap, n; } kd_map;

struct fastf_keydict {
    kd_map prefixes;      /* prefix string → id        */
    kd_map escapes;       /* whole string  → ordinal   */
    const char **prefix_by_id; size_t prefix_cap;     /* reverse tables for fastf_keydict_decode */
    const char **escape_by_ord; size_t escape_cap;
    pthread_mutex_t mu;   /* fastf_keydict_intern: writers and map readers */
    uint64_t generation;  /* distinguishes dictionaries that reuse an address (thread-local caches) */
};
static uint64_t g_kd_generation;

/* one-entry per-thread cache: real inputs use a handful of prefixes.  Prefix strings are
 * individually malloc'ed and never freed before the dict, so the pointer stays valid
 * across rehashes; the dict pointer guards against reuse across dictionaries. */
static __thread struct { const fastf_keydict_t *d; uint64_t gen; const char *p; uint32_t len; uint64_t id; } tl_prefix;

#define KD_MAX_PREFIX (1u << 14)

static uint64_t fnv1a(const char *s, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)s[i]; h *= 1099511628211ull; }
    return h;
}

static const kd_ent *map_find(const kd_map *m, const char *s, size_t len)
{
    if (!m->cap) return NULL;
    size_t mask = m->cap - 1;
    for (size_t i = fnv1a(s, len) & mask;; i = (i + 1) & mask) {
        const kd_ent *e = &m->e[i];
        if (!e->s) return NULL;
        if (e->len == len && memcmp(e->s, s, len) == 0) return e;
    }
}

static const char *map_put(kd_map *m, const char *s, size_t len, uint64_t val)
{
    if ((m->n + 1) * 2 > m->cap) {
        size_t ncap = m->cap ? m->cap * 2 : 64;
        kd_ent *ne = (kd_ent *)calloc(ncap, sizeof *ne);
        for (size_t i = 0; i < m->cap; i++)
            if (m->e[i].s) {
                size_t j = fnv1a(m->e[i].s, m->e[i].len) & (ncap - 1);
                while (ne[j].s) j = (j + 1) & (ncap - 1);
                ne[j] = m->e[i];
            }
        free(m->e);
        m->e = ne; m->cap = ncap;
    }
    size_t j = fnv1a(s, len) & (m->cap - 1);
    while (m->e[j].s) j = (j + 1) & (m->cap - 1);
    m->e[j].s = (char *)malloc(len + 1);
    memcpy(m->e[j].s, s, len); m->e[j].s[len] = '\0';
    m->e[j].len = (uint32_t)len; m->e[j].val = val;
    m->n++;
    return m->e[j].s;
}

static void rev_put(const char ***tab, size_t *cap, uint64_t id, const char *s)
{
    if (id >= *cap) {
        size_t ncap = *cap ? *cap * 2 : 64;
        while (id >= ncap) ncap *= 2;
        *tab = (const char **)realloc((void *)*tab, ncap * sizeof **tab);
        *cap = ncap;
    }
    (*tab)[id] = s;
}

static void map_free(kd_map *m)
{
    for (size_t i = 0; i < m->cap; i++) free(m->e[i].s);
    free(m->e);
}

fastf_keydict_t *fastf_keydict_create(void)
{
    fastf_keydict_t *d = (fastf_keydict_t *)calloc(1, sizeof(fastf_keydict_t));
    if (!d) return NULL;
    pthread_mutex_init(&d->mu, NULL);
    d->generation = __atomic_add_fetch(&g_kd_generation, 1, __ATOMIC_RELAXED);
    return d;
}

void fastf_keydict_destroy(fastf_keydict_t *d)
{
    if (!d) return;
    if (tl_prefix.d == d) tl_prefix.d = NULL;
    map_free(&d->prefixes); map_free(&d->escapes);
    free((void *)d->prefix_by_id); free((void *)d->escape_by_ord);
    pthread_mutex_destroy(&d->mu);
    free(d);
}

/* DNA form; returns 0 when the string is not of that form */
static uint64_t pack_dna(const char *s, size_t len)
{
    uint64_t bases = 0;
    size_t n = 0;
    while (n < len) {
        const uint64_t c = k_base_code[(unsigned char)s[n]];
        if (c & 4) break;
        if (n == 24) return 0;                    /* 25th base: too long */
        bases = (bases << 2) | c;
        n++;
    }
    if (n == 0) return 0;
    bases <<= 48 - 2 * n;
    uint64_t suffix = 0;
    if (n < len) {
        if (s[n] != '-') return 0;
        size_t d = n + 1, nd = len - d;
        if (nd == 0 || nd > 3) return 0;
        if (s[d] == '0' && nd > 1) return 0;      /* canonical decimal only */
        uint32_t v = 0;
        for (size_t i = d; i < len; i++) {
            if (s[i] < '0' || s[i] > '9') return 0;
            v = v * 10 + (uint32_t)(s[i] - '0');
        }
        if (v > 254) return 0;
        suffix = v + 1;
    }
    return (1ull << 62) | ((uint64_t)n << 57) | (suffix << 49) | bases;
}

/* DNA+N form (raw barcodes / UMIs with uncalled bases); 0 when the string is not of that form.  Pure ACGT strings
 * never get here as keys of this form: pack_dna() is tried first and this form requires at least one N. */
static uint64_t pack_dna_n(const char *s, size_t len)
{
    uint64_t bases = 0;
    size_t n = 0;
    int has_n = 0;
    while (n < len) {
        uint64_t c = k_base_code[(unsigned char)s[n]];
        if (c & 4) {
            if (s[n] != 'N') break;
            c = 4; has_n = 1;
        }
        if (n == 16) return 0;
        bases = (bases << 3) | c;
        n++;
    }
    if (n == 0 || !has_n) return 0;
    bases <<= 48 - 3 * n;
    uint64_t suffix = 0;
    if (n < len) {
        if (s[n] != '-') return 0;
        size_t d = n + 1, nd = len - d;
        if (nd == 0 || nd > 3) return 0;
        if (s[d] == '0' && nd > 1) return 0;
        uint32_t v = 0;
        for (size_t i = d; i < len; i++) {
            if (s[i] < '0' || s[i] > '9') return 0;
            v = v * 10 + (uint32_t)(s[i] - '0');
        }
        if (v > 254) return 0;
        suffix = v + 1;
    }
    return (3ull << 62) | (1ull << 61) | ((uint64_t)n << 56) | (suffix << 48) | bases;
}

/* splits <prefix><digits>; returns digit count (0 = not of the ID form) */
static size_t split_id(const char *s, size_t len, uint64_t *value)
{
    size_t nd = 0;
    while (nd < len && s[len - 1 - nd] >= '0' && s[len - 1 - nd] <= '9') nd++;
    if (nd == 0 || nd > 13) return 0;
    uint64_t v = 0;
    for (size_t i = len - nd; i < len; i++) v = v * 10 + (uint64_t)(s[i] - '0');
    *value = v;
    return nd;
}

static uint64_t make_id(uint64_t prefix_id, size_t nd, uint64_t value)
{
    return (2ull << 62) | (prefix_id << 48) | ((uint64_t)nd << 44) | value;
}

uint64_t fastf_keydict_add(fastf_keydict_t *d, const char *s, size_t len)
{
    uint64_t k = pack_dna(s, len);
    if (k) return k;
    if ((k = pack_dna_n(s, len))) return k;
    uint64_t value;
    size_t nd = split_id(s, len, &value);
    if (nd) {
        size_t plen = len - nd;
        const kd_ent *p = map_find(&d->prefixes, s, plen);
        if (!p && d->prefixes.n < KD_MAX_PREFIX) {
            const uint64_t id = d->prefixes.n;
            rev_put(&d->prefix_by_id, &d->prefix_cap, id, map_put(&d->prefixes, s, plen, id));
            p = map_find(&d->prefixes, s, plen);
        }
        if (p) return make_id(p->val, nd, value);
    }
    const kd_ent *e = map_find(&d->escapes, s, len);
    if (e) return (3ull << 62) | e->val;
    uint64_t ord = d->escapes.n;
    rev_put(&d->escape_by_ord, &d->escape_cap, ord, map_put(&d->escapes, s, len, ord));
    return (3ull << 62) | ord;
}

/* the ID-form prefixes for the device-side packer (gpu_records.hpp pack_key); 1 = this dictionary needs the host packer */
int fastf_keydict_export(const fastf_keydict_t *d, fastf_keydict_view_t *v)
{
    memset(v, 0, sizeof *v);
    if (!d || d->escapes.n || d->prefixes.n > 8) return 1;
    for (size_t i = 0; i < d->prefixes.cap; i++) {
        const kd_ent *e = &d->prefixes.e[i];
        if (!e->s) continue;
        if (e->len > 32) return 1;
        const uint32_t q = v->n_prefix++;
        v->prefix_len[q] = e->len; v->prefix_id[q] = e->val;
        memcpy(v->prefix[q], e->s, e->len);
    }
    return 0;
}

uint64_t fastf_keydict_pack(const fastf_keydict_t *d, const char *s, size_t len)
{
    uint64_t k = pack_dna(s, len);
    if (k) return k;
    if ((k = pack_dna_n(s, len))) return k;
    uint64_t value;
    size_t nd = split_id(s, len, &value);
    if (nd) {
        size_t plen = len - nd;
        if (tl_prefix.d == d && tl_prefix.gen == d->generation && tl_prefix.len == plen && memcmp(tl_prefix.p, s, plen) == 0)
            return make_id(tl_prefix.id, nd, value);
        const kd_ent *p = map_find(&d->prefixes, s, plen);
        if (p) {
            tl_prefix.d = d; tl_prefix.gen = d->generation; tl_prefix.p = p->s; tl_prefix.len = p->len; tl_prefix.id = p->val;
            return make_id(p->val, nd, value);
        }
    }
    const kd_ent *e = map_find(&d->escapes, s, len);
    return e ? ((3ull << 62) | e->val) : 0;
}

/* Thread-safe "pack, registering the string if it is new" — the tag-histogram paths (crb / extract) have no list of
 * strings up front.  DNA forms need no dictionary; prefixes and escapes are looked up in small per-thread caches
 * first (entries point at strings the dictionary never frees or moves before its destruction) and under the
 * dictionary mutex otherwise.  Must not run concurrently with fastf_keydict_add / _pack on the same dictionary. */
#define TL_ESC_SLOTS 2048
static __thread struct { uint64_t gen; const char *p; uint32_t len; uint64_t key; } tl_esc[TL_ESC_SLOTS];

uint64_t fastf_keydict_intern(fastf_keydict_t *d, const char *s, size_t len)
{
    uint64_t k = pack_dna(s, len);
    if (k) return k;
    if ((k = pack_dna_n(s, len))) return k;
    uint64_t value;
    size_t nd = split_id(s, len, &value);
    if (nd) {
        size_t plen = len - nd;
        if (tl_prefix.d == d && tl_prefix.gen == d->generation && tl_prefix.len == plen && memcmp(tl_prefix.p, s, plen) == 0)
            return make_id(tl_prefix.id, nd, value);
        pthread_mutex_lock(&d->mu);
        const kd_ent *p = map_find(&d->prefixes, s, plen);
        if (!p && d->prefixes.n < KD_MAX_PREFIX) {
            const uint64_t id = d->prefixes.n;
            rev_put(&d->prefix_by_id, &d->prefix_cap, id, map_put(&d->prefixes, s, plen, id));
            p = map_find(&d->prefixes, s, plen);
        }
        if (p) {
            tl_prefix.d = d; tl_prefix.gen = d->generation; tl_prefix.p = p->s; tl_prefix.len = p->len; tl_prefix.id = p->val;
            k = make_id(p->val, nd, value);
        }
        pthread_mutex_unlock(&d->mu);
        if (k) return k;
    }
    const uint64_t h = fnv1a(s, len);
    const size_t slot = (size_t)(h ^ (h >> 32)) & (TL_ESC_SLOTS - 1);
    if (tl_esc[slot].gen == d->generation && tl_esc[slot].len == len && memcmp(tl_esc[slot].p, s, len) == 0)
        return tl_esc[slot].key;
    pthread_mutex_lock(&d->mu);
    const kd_ent *e = map_find(&d->escapes, s, len);
    if (!e) {
        const uint64_t ord = d->escapes.n;
        rev_put(&d->escape_by_ord, &d->escape_cap, ord, map_put(&d->escapes, s, len, ord));
        e = map_find(&d->escapes, s, len);
    }
    k = (3ull << 62) | e->val;
    tl_esc[slot].gen = d->generation; tl_esc[slot].p = e->s; tl_esc[slot].len = e->len; tl_esc[slot].key = k;
    pthread_mutex_unlock(&d->mu);
    return k;
}

/* key → string (inverse of add / pack / intern on this dictionary).  Returns the length, or -1 for a key this
 * dictionary did not produce or a buffer that is too small (cap includes the terminating NUL). */
long fastf_keydict_decode(const fastf_keydict_t *d, uint64_t key, char *buf, size_t cap)
{
    static const char B[5] = {'A', 'C', 'G', 'T', 'N'};
    const unsigned form = (unsigned)(key >> 62);
    size_t n = 0;
    if (form == 1 || (form == 3 && (key >> 61 & 1))) {
        const int wide = form == 3;
        const size_t nb = wide ? (size_t)(key >> 56 & 31) : (size_t)(key >> 57 & 31);
        const unsigned suffix = wide ? (unsigned)(key >> 48 & 255) : (unsigned)(key >> 49 & 255);
        if (nb == 0 || nb > (wide ? 16u : 24u) || nb + 5 > cap) return -1;
        for (size_t i = 0; i < nb; i++) {
            const unsigned c = wide ? (unsigned)(key >> (45 - 3 * i) & 7) : (unsigned)(key >> (46 - 2 * i) & 3);
            if (c > 4) return -1;
            buf[n++] = B[c];
        }
        if (suffix) n += (size_t)snprintf(buf + n, cap - n, "-%u", suffix - 1);
        buf[n] = '\0';
        return (long)n;
    }
    if (form == 2) {
        const uint64_t pid = key >> 48 & 0x3fff, nd = key >> 44 & 15, value = key & 0xfffffffffffull;
        if (pid >= d->prefixes.n || nd == 0 || nd > 13) return -1;
        const char *p = d->prefix_by_id[pid];
        const size_t pl = strlen(p);
        if (pl + nd + 1 > cap) return -1;
        memcpy(buf, p, pl);
        snprintf(buf + pl, cap - pl, "%0*llu", (int)nd, (unsigned long long)value);
        return (long)(pl + nd);
    }
    if (form == 3) {
        const uint64_t ord = key & ((1ull << 61) - 1);
        if (ord >= d->escapes.n) return -1;
        const char *p = d->escape_by_ord[ord];
        const size_t pl = strlen(p);
        if (pl + 1 > cap) return -1;
        memcpy(buf, p, pl + 1);
        return (long)pl;
    }
    return -1;
}

void fastf_keydict_pack_many(const fastf_keydict_t *d, const char *strs, size_t stride,
                             size_t n, const uint8_t *present, uint64_t *out)
{
    for (size_t i = 0; i < n; i++) {
        const char *s = strs + i * stride;
        if (present && !present[i]) { out[i] = 0; continue; }
        out[i] = fastf_keydict_pack(d, s, strnlen(s, stride));
    }
}
