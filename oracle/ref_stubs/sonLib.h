/*
 * sonLib.h -- STAND-IN, test infrastructure only.  The one reference source of the hot path's "next" rows that compiles
 * from its own file, /root/reference/preprocessor/lastzRepeatMasking/cactus_covered_intervals.c, includes sonLib.h for a
 * string-keyed hash (stHash_construct3 / insert / search / destruct, lines 308, 351, 427, 563) and nothing else; the sonLib
 * submodule is empty in the reference checkout.  This header supplies exactly those five entry points (own code: separate
 * chaining, FNV-1a) so that oracle/Makefile can build the UNMODIFIED reference file into oracle/_ref/ as a checker for
 * cactus_amd.preprocessor.lastz_repeat_mask.covered_intervals.  Never part of the product.
 */
#ifndef MIBLAST_SONLIB_STANDIN_H
#define MIBLAST_SONLIB_STANDIN_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct st_standin_entry { void *key, *value; struct st_standin_entry *next; } st_standin_entry;
typedef struct stHash {
    uint64_t (*hashKey)(const void *);
    int (*equalKey)(const void *, const void *);
    void (*destructKey)(void *);
    void (*destructValue)(void *);
    st_standin_entry *bucket[4096];
} stHash;

static inline uint64_t stHash_stringKey(const void *k) {
    uint64_t h = 1469598103934665603ull;
    for (const unsigned char *p = (const unsigned char *)k; *p; p++) { h ^= *p; h *= 1099511628211ull; }
    return h;
}
static inline int stHash_stringEqualKey(const void *a, const void *b) { return strcmp((const char *)a, (const char *)b) == 0; }

static inline stHash *stHash_construct3(uint64_t (*hashKey)(const void *), int (*equalKey)(const void *, const void *),
                                        void (*destructKey)(void *), void (*destructValue)(void *)) {
    stHash *h = (stHash *)calloc(1, sizeof(stHash));
    h->hashKey = hashKey; h->equalKey = equalKey; h->destructKey = destructKey; h->destructValue = destructValue;
    return h;
}
static inline void *stHash_search(stHash *h, void *key) {
    for (st_standin_entry *e = h->bucket[h->hashKey(key) & 4095u]; e; e = e->next)
        if (h->equalKey(e->key, key)) return e->value;
    return NULL;
}
static inline void stHash_insert(stHash *h, void *key, void *value) {
    st_standin_entry *e = (st_standin_entry *)malloc(sizeof *e);
    const uint64_t b = h->hashKey(key) & 4095u;
    e->key = key; e->value = value; e->next = h->bucket[b]; h->bucket[b] = e;
}
static inline void stHash_destruct(stHash *h) {
    for (int b = 0; b < 4096; b++)
        for (st_standin_entry *e = h->bucket[b], *n; e; e = n) {
            n = e->next;
            if (h->destructKey) h->destructKey(e->key);
            if (h->destructValue) h->destructValue(e->value);
            free(e);
        }
    free(h);
}
#endif
