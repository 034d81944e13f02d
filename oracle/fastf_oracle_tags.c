/*
 * fastf_oracle_tags.c — CPU oracle for the fastF `crb` and `extract` paths.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as fastf_oracle.c): loaded by tests/ only, as the checker.
 *
 * Plain-C restatement of
 *   extract.c:3-31    insert_CB_node      (unbalanced BST over CB, one CR tree per node)
 *   extract.c:47-62   print_CB_node       ("CB;" + CR tree on one row + "\n", PRE-order: node, left, right)
 *   extract.c:64-133  read_bam            (record loop of `crb`)
 *   extract.c:135-216 extract_bam         (record loop of `extract`, tag_summary.csv)
 *   filter.c:76-89    new_node            (count starts at 1)
 *   filter.c:105-124  insert_tree         (strcmp BST, equal key → count++)
 *   filter.c:139-148  print_tree          ("%s,%ld\n", pre-order)
 *   filter.c:160-169  print_tree_same_row ("%s,%ld;", pre-order)
 * on in-memory records (fixed-stride NUL-terminated strings), writing the decompressed bytes of the
 * output file into a malloc'ed buffer.  htslib (absent) only parses; the BAM side is the repo's reader.
 *
 * Pinned by oracle/_ref: filter.c compiles from the reference tree on its own (gcc + zlib), so the
 * reference's own insert_tree / print_tree / print_tree_same_row are run on the same strings
 * (tests/test_oracle_pins.py::test_tag_tree_*).
 */
#include "fastf_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct tnode { char *data; long count; struct tnode *left, *right; } tnode;              /* filter.h:28-34 */
typedef struct cbnode { tnode *cr; char *cb; struct cbnode *left, *right; } cbnode;              /* extract.h:8-13 */

typedef struct { char *p; size_t len, cap; } sbuf;
static void sb_put(sbuf *b, const char *s, size_t n)
{
    if (b->len + n + 1 > b->cap) {
        while (b->len + n + 1 > b->cap) b->cap = b->cap ? b->cap * 2 : 4096;
        b->p = (char *)realloc(b->p, b->cap);
    }
    memcpy(b->p + b->len, s, n);
    b->len += n;
    b->p[b->len] = '\0';
}

static tnode *t_new(const char *data)                                       /* filter.c:76-89 */
{
    tnode *o = (tnode *)malloc(sizeof *o);
    o->data = strdup(data); o->count = 1; o->left = o->right = NULL;
    return o;
}

/* filter.c:105-124, iterative (the reference recurses; the shape of the tree is the same) */
static tnode *t_insert(tnode *root, const char *data)
{
    if (!root) return t_new(data);
    tnode *n = root;
    for (;;) {
        int cmp = strcmp(data, n->data);
        if (cmp < 0) { if (!n->left) { n->left = t_new(data); break; } n = n->left; }
        else if (cmp > 0) { if (!n->right) { n->right = t_new(data); break; } n = n->right; }
        else { n->count++; break; }
    }
    return root;
}

/* pre-order walk with an explicit stack: node, left subtree, right subtree (filter.c:139-148 / 160-169) */
static void t_print(const tnode *root, sbuf *out, char sep_after)           /* sep '\n' → print_tree, ';' → same_row */
{
    size_t cap = 64, top = 0;
    const tnode **st = (const tnode **)malloc(cap * sizeof *st);
    if (root) st[top++] = root;
    char num[32];
    while (top) {
        const tnode *n = st[--top];
        sb_put(out, n->data, strlen(n->data));
        int k = snprintf(num, sizeof num, ",%ld%c", n->count, sep_after);
        sb_put(out, num, (size_t)k);
        if (top + 2 > cap) { cap *= 2; st = (const tnode **)realloc(st, cap * sizeof *st); }
        if (n->right) st[top++] = n->right;
        if (n->left) st[top++] = n->left;
    }
    free(st);
}

static void t_free(tnode *root)
{
    size_t cap = 64, top = 0;
    tnode **st = (tnode **)malloc(cap * sizeof *st);
    if (root) st[top++] = root;
    while (top) {
        tnode *n = st[--top];
        if (top + 2 > cap) { cap *= 2; st = (tnode **)realloc(st, cap * sizeof *st); }
        if (n->left) st[top++] = n->left;
        if (n->right) st[top++] = n->right;
        free(n->data); free(n);
    }
    free(st);
}

/* extract_bam (extract.c:135-216).  type 0: tag value is a string; type 1: `sprintf("%d", bam_aux2i())`.
 * present[i] != 0 ⇔ bam_aux_get found the tag.  total_printed reproduces the double increment of
 * total_count (extract.c:163,165): the reference prints 2 x the number of records. */
int oracle_extract(size_t n, const uint8_t *present, const char *vals, size_t stride, const int64_t *ivals, int type,
                   char **csv_out, size_t *csv_len, uint64_t *total_printed, uint64_t *valid)
{
    tnode *root = NULL;
    uint64_t total = 0, nvalid = 0;
    char tmp[32];
    for (size_t i = 0; i < n; i++) {
        total++; total++;                                                   /* :163, :165 */
        if (!present[i]) continue;                                          /* :183 */
        nvalid++;                                                           /* :185 */
        if (type == 0) root = t_insert(root, vals + i * stride);           /* :188-190 */
        else { snprintf(tmp, sizeof tmp, "%d", (int)ivals[i]); root = t_insert(root, tmp); }   /* :191-195 */
    }
    sbuf out = {0};
    sb_put(&out, "", 0);
    t_print(root, &out, '\n');                                              /* :200-202 */
    t_free(root);
    *csv_out = out.p; *csv_len = out.len;
    if (total_printed) *total_printed = total;
    if (valid) *valid = nvalid;
    return 0;
}

/* read_bam + print_CB_node (extract.c:64-133, 47-62): records with a CB are inserted as (CB, CR); the
 * reference dereferences CR unchecked (:102 strcpy of bam_aux2Z(NULL)), so has_cr must hold wherever has_cb
 * does — callers keep to that (a record that breaks it is skipped and counted in *undefined). */
int oracle_crb(size_t n, const uint8_t *has_cb, const uint8_t *has_cr, const char *cb, size_t cb_stride,
               const char *cr, size_t cr_stride, char **txt_out, size_t *txt_len, uint64_t *read_count, uint64_t *undefined)
{
    cbnode *root = NULL;
    uint64_t undef = 0;
    for (size_t i = 0; i < n; i++) {
        if (!has_cb[i]) continue;                                           /* :94 */
        if (!has_cr[i]) { undef++; continue; }
        const char *CB = cb + i * cb_stride, *CR = cr + i * cr_stride;
        /* insert_CB_node (extract.c:3-31), iterative */
        cbnode **slot = &root;
        for (;;) {
            if (!*slot) {
                cbnode *c = (cbnode *)malloc(sizeof *c);
                c->cb = strdup(CB); c->cr = t_new(CR); c->left = c->right = NULL;
                *slot = c;
                break;
            }
            int cmp = strcmp(CB, (*slot)->cb);
            if (cmp < 0) slot = &(*slot)->left;
            else if (cmp > 0) slot = &(*slot)->right;
            else { (*slot)->cr = t_insert((*slot)->cr, CR); break; }
        }
    }
    sbuf out = {0};
    sb_put(&out, "", 0);
    /* print_CB_node: pre-order */
    size_t cap = 64, top = 0;
    cbnode **st = (cbnode **)malloc(cap * sizeof *st);
    if (root) st[top++] = root;
    while (top) {
        cbnode *c = st[--top];
        sb_put(&out, c->cb, strlen(c->cb)); sb_put(&out, ";", 1);           /* :55 */
        t_print(c->cr, &out, ';');                                          /* :56 */
        sb_put(&out, "\n", 1);                                              /* :57 */
        if (top + 2 > cap) { cap *= 2; st = (cbnode **)realloc(st, cap * sizeof *st); }
        if (c->right) st[top++] = c->right;
        if (c->left) st[top++] = c->left;
        t_free(c->cr); free(c->cb); free(c);
    }
    free(st);
    *txt_out = out.p; *txt_len = out.len;
    if (read_count) *read_count = n;                                        /* :111, :131 */
    if (undefined) *undefined = undef;
    return 0;
}

void oracle_free(void *p) { free(p); }
