/* TEST INFRASTRUCTURE (oracle) -- CPU restatement of the DBoW2 vocabulary path ORB-SLAM2 uses
 * (SURVEY.md §8(f) rank 3).  PARITY UNPINNED: the reference's DBoW2 needs OpenCV and cannot be built
 * here, and the vocabulary file is absent; this file restates the algorithm from
 *   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h  transform :1140-1187 (vector) and :1215-1260 (one
 *   feature), loadFromTextFile :1351-1431, loadFromBinaryFile :1456-1496, saveToBinaryFile :1500-1521,
 *   Thirdparty/DBoW2/DBoW2/FORB.cpp distance :81-100,
 *   Thirdparty/DBoW2/DBoW2/BowVector.cpp addWeight :30-42, addIfNotExist :46-54, normalize :58-82,
 *   Thirdparty/DBoW2/DBoW2/FeatureVector.cpp addFeature :31-45,
 *   Thirdparty/DBoW2/DBoW2/ScoringObject.cpp L1Scoring::score :23-72.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "orb_oracle.h"

typedef struct {
    uint32_t parent;
    uint8_t desc[32];
    double weight;
    uint32_t word_id;
    uint32_t *children;
    int nchild, cchild;
} vnode_t;

struct orc_vocab {
    int k, L, scoring, weighting;
    vnode_t *nodes;
    int n_nodes, cap_nodes; /* includes the root (node 0) */
    int n_words;
};

static void node_init(vnode_t *n)
{
    memset(n, 0, sizeof(*n)); /* Node(): id(0), weight(0), parent(0), word_id(0) :329 */
}
static void add_child(vnode_t *p, uint32_t id)
{
    if (p->nchild == p->cchild) {
        p->cchild = p->cchild ? 2 * p->cchild : 12;
        p->children = (uint32_t *)realloc(p->children, sizeof(uint32_t) * p->cchild);
    }
    p->children[p->nchild++] = id;
}
static void vocab_clear(orc_vocab_t *v)
{
    for (int i = 0; i < v->n_nodes; ++i) free(v->nodes[i].children);
    free(v->nodes);
    v->nodes = NULL;
    v->n_nodes = v->cap_nodes = v->n_words = 0;
}
static void vocab_resize(orc_vocab_t *v, int n)
{
    if (n > v->cap_nodes) {
        int c = v->cap_nodes ? v->cap_nodes : 64;
        while (c < n) c *= 2;
        v->nodes = (vnode_t *)realloc(v->nodes, sizeof(vnode_t) * c);
        v->cap_nodes = c;
    }
    for (int i = v->n_nodes; i < n; ++i) node_init(&v->nodes[i]);
    v->n_nodes = n;
}

orc_vocab_t *orc_vocab_create(void) { return (orc_vocab_t *)calloc(1, sizeof(orc_vocab_t)); }
void orc_vocab_destroy(orc_vocab_t *v)
{
    if (!v) return;
    vocab_clear(v);
    free(v);
}
int orc_vocab_k(const orc_vocab_t *v) { return v->k; }
int orc_vocab_L(const orc_vocab_t *v) { return v->L; }
int orc_vocab_scoring(const orc_vocab_t *v) { return v->scoring; }
int orc_vocab_weighting(const orc_vocab_t *v) { return v->weighting; }
int orc_vocab_nodes(const orc_vocab_t *v) { return v->n_nodes; }
int orc_vocab_size(const orc_vocab_t *v) { return v->n_words; } /* size() = m_words.size() */

/* one record of either loader: nodes arrive in id order, children in order of appearance */
static void append_node(orc_vocab_t *v, int nid, uint32_t parent, const uint8_t *desc, double weight, int is_leaf)
{
    vnode_t *n = &v->nodes[nid];
    n->parent = parent;
    add_child(&v->nodes[parent], (uint32_t)nid);
    memcpy(n->desc, desc, 32);
    n->weight = weight;
    if (is_leaf) n->word_id = (uint32_t)v->n_words++;
}

/* programmatic construction (tests/bench: no loader quirks).  Node i+1 = record i. */
int orc_vocab_set_nodes(orc_vocab_t *v, int k, int L, int scoring, int weighting, int n, const int32_t *parent,
                        const uint8_t *desc, const double *weight, const uint8_t *is_leaf)
{
    vocab_clear(v);
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    vocab_resize(v, n + 1);
    for (int i = 0; i < n; ++i) {
        if (parent[i] < 0 || parent[i] > i) return -1; /* parents precede children in both file formats */
        append_node(v, i + 1, (uint32_t)parent[i], desc + (size_t)i * 32, weight[i], is_leaf[i]);
    }
    return 0;
}

/* loadFromBinaryFile :1456-1496, including its `while(!f.eof())` tail: after the last record the read
 * fails, buf still holds that record, and it is appended once more as node nb_nodes. */
int orc_vocab_load_binary(orc_vocab_t *v, const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint32_t nb_nodes, size_node;
    int32_t hdr[4];
    if (fread(&nb_nodes, 4, 1, f) != 1 || fread(&size_node, 4, 1, f) != 1 || fread(hdr, 4, 4, f) != 4 || size_node < 41 ||
        size_node > 4096) {
        fclose(f);
        return -2;
    }
    vocab_clear(v);
    v->k = hdr[0]; v->L = hdr[1]; v->scoring = hdr[2]; v->weighting = hdr[3];
    vocab_resize(v, (int)nb_nodes + 1);
    uint8_t *buf = (uint8_t *)calloc(1, size_node);
    int nid = 1, at_eof = 0;
    while (!at_eof) {
        if (fread(buf, 1, size_node, f) != size_node) at_eof = 1; /* eof is only seen by a failing read; buf is reused */
        if (nid > (int)nb_nodes) break;                           /* (the reference would write out of bounds) */
        int32_t parent;
        float w;
        memcpy(&parent, buf, 4);
        memcpy(&w, buf + 36, 4);
        if (parent < 0 || parent >= nid) {
            free(buf);
            fclose(f);
            return -3;
        }
        append_node(v, nid, (uint32_t)parent, buf + 4, (double)w, buf[40] != 0);
        nid++;
    }
    free(buf);
    fclose(f);
    return 0;
}

/* saveToBinaryFile :1500-1521 */
int orc_vocab_save_binary(const orc_vocab_t *v, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    uint32_t nb_nodes = (uint32_t)v->n_nodes, size_node = 41;
    int32_t hdr[4] = {v->k, v->L, v->scoring, v->weighting};
    fwrite(&nb_nodes, 4, 1, f);
    fwrite(&size_node, 4, 1, f);
    fwrite(hdr, 4, 4, f);
    for (uint32_t i = 1; i < nb_nodes; ++i) {
        const vnode_t *n = &v->nodes[i];
        float w = (float)n->weight;
        uint8_t leaf = n->nchild == 0;
        fwrite(&n->parent, 4, 1, f);
        fwrite(n->desc, 1, 32, f);
        fwrite(&w, 4, 1, f);
        fwrite(&leaf, 1, 1, f);
    }
    fclose(f);
    return 0;
}

/* loadFromTextFile :1351-1431.  `while(!f.eof()) getline` also turns the empty string after the final
 * newline into a node: every `>>` fails and leaves 0 (C++11), i.e. parent 0, not a leaf by flag, all-zero
 * descriptor (FORB::fromString :120-135 on a fresh Mat -- taken as zero here), weight 0, no word. */
int orc_vocab_load_text(orc_vocab_t *v, const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    size_t cap = 1 << 12;
    char *line = (char *)malloc(cap);
    int n1, n2, k, L;
    if (!fgets(line, (int)cap, f) || sscanf(line, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 ||
        n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        free(line);
        fclose(f);
        return -2;
    }
    vocab_clear(v);
    v->k = k; v->L = L; v->scoring = n1; v->weighting = n2;
    vocab_resize(v, 1);
    int last_had_newline = 1; /* the header line ended with '\n' */
    for (;;) {
        const int got = fgets(line, (int)cap, f) != NULL;
        if (!got && !last_had_newline) break; /* previous getline hit eof while reading a non-empty last line */
        const int nid = v->n_nodes;
        vocab_resize(v, nid + 1);
        int pid = 0, leaf = 0;
        uint8_t d[32] = {0};
        double w = 0;
        if (got) {
            const size_t len = strlen(line);
            last_had_newline = len > 0 && line[len - 1] == '\n';
            char *p = line;
            pid = (int)strtol(p, &p, 10);
            leaf = (int)strtol(p, &p, 10);
            for (int i = 0; i < 32; ++i) d[i] = (uint8_t)strtol(p, &p, 10);
            w = strtod(p, &p);
        }
        if (pid < 0 || pid >= nid) {
            free(line);
            fclose(f);
            return -3;
        }
        append_node(v, nid, (uint32_t)pid, d, w, leaf > 0);
        if (!got) break; /* that was the empty trailing "line" */
    }
    free(line);
    fclose(f);
    return 0;
}

/* FORB::distance :81-100 */
static int forb_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        unsigned int x = pa ^ pb;
        x = x - ((x >> 1) & 0x55555555);
        x = (x & 0x33333333) + ((x >> 2) & 0x33333333);
        dist += (((x + (x >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* transform(feature, word_id, weight, nid, levelsup) :1215-1260.  *nid keeps its incoming value when the
 * descent ends above nid_level (the reference leaves it uninitialised; callers here pass 0). */
static void transform_one(const orc_vocab_t *v, const uint8_t *feature, uint32_t *word_id, double *weight, uint32_t *nid,
                          int levelsup)
{
    const int nid_level = v->L - levelsup;
    if (nid_level <= 0) *nid = 0;
    uint32_t final_id = 0;
    int current_level = 0;
    do {
        ++current_level;
        const vnode_t *n = &v->nodes[final_id];
        final_id = n->children[0];
        double best_d = forb_distance(feature, v->nodes[final_id].desc);
        for (int j = 1; j < n->nchild; ++j) {
            const uint32_t id = n->children[j];
            const double d = forb_distance(feature, v->nodes[id].desc);
            if (d < best_d) {
                best_d = d;
                final_id = id;
            }
        }
        if (current_level == nid_level) *nid = final_id;
    } while (v->nodes[final_id].nchild != 0);
    *word_id = v->nodes[final_id].word_id;
    *weight = v->nodes[final_id].weight;
}

/* transform(features, BowVector&, FeatureVector&, levelsup) :1140-1187.  The two std::maps come back as
 * key-ascending arrays: bow_word/bow_value[n_bow]; fv_node[n_fv], fv_off[n_fv+1], fv_idx[...].
 * Optional per-feature taps: word_of[n], node_of[n].  Returns 0, or -1 for an empty vocabulary (:1147). */
int orc_vocab_transform(const orc_vocab_t *v, const uint8_t *desc, int n, int levelsup, uint32_t *bow_word, double *bow_value,
                        int *n_bow, int32_t *fv_node, int32_t *fv_off, int32_t *fv_idx, int *n_fv, uint32_t *word_of,
                        uint32_t *node_of)
{
    *n_bow = 0;
    *n_fv = 0;
    fv_off[0] = 0;
    if (v->n_words == 0 || v->n_nodes <= 1 || v->nodes[0].nchild == 0) return -1;
    const int must = v->scoring != 5; /* DOT_PRODUCT is the only scoring that does not normalise */
    const int l2 = v->scoring == 1;
    const int tf = v->weighting == 0 || v->weighting == 1;
    int nb = 0;
    /* FeatureVector as (node, feature) list kept sorted by node, features in arrival order */
    int32_t *pn = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1)), *pi = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
    int np = 0;
    for (int i = 0; i < n; ++i) {
        uint32_t id, nid = 0;
        double w;
        transform_one(v, desc + (size_t)i * 32, &id, &w, &nid, levelsup);
        if (word_of) word_of[i] = id;
        if (node_of) node_of[i] = nid;
        if (w > 0) {
            int lo = 0, hi = nb; /* lower_bound */
            while (lo < hi) {
                const int mid = (lo + hi) / 2;
                if (bow_word[mid] < id) lo = mid + 1; else hi = mid;
            }
            if (lo < nb && bow_word[lo] == id) {
                if (tf) bow_value[lo] += w; /* addWeight; addIfNotExist leaves it */
            } else {
                memmove(bow_word + lo + 1, bow_word + lo, sizeof(uint32_t) * (nb - lo));
                memmove(bow_value + lo + 1, bow_value + lo, sizeof(double) * (nb - lo));
                bow_word[lo] = id;
                bow_value[lo] = w;
                nb++;
            }
            /* addFeature: stable insert after the last entry of node nid */
            int p = np;
            while (p > 0 && pn[p - 1] > (int32_t)nid) --p;
            memmove(pn + p + 1, pn + p, sizeof(int32_t) * (np - p));
            memmove(pi + p + 1, pi + p, sizeof(int32_t) * (np - p));
            pn[p] = (int32_t)nid;
            pi[p] = i;
            np++;
        }
    }
    if (tf && nb > 0 && !must) {
        const double nd = nb;
        for (int j = 0; j < nb; ++j) bow_value[j] /= nd;
    }
    if (must) { /* BowVector::normalize :58-82 */
        double norm = 0.0;
        if (!l2)
            for (int j = 0; j < nb; ++j) norm += fabs(bow_value[j]);
        else {
            for (int j = 0; j < nb; ++j) norm += bow_value[j] * bow_value[j];
            norm = sqrt(norm);
        }
        if (norm > 0.0)
            for (int j = 0; j < nb; ++j) bow_value[j] /= norm;
    }
    *n_bow = nb;
    int nf = 0;
    for (int p = 0; p < np; ++p) {
        if (p == 0 || pn[p] != pn[p - 1]) {
            fv_node[nf] = pn[p];
            fv_off[nf] = p;
            nf++;
        }
        fv_idx[p] = pi[p];
    }
    fv_off[nf] = np;
    *n_fv = nf;
    free(pn);
    free(pi);
    return 0;
}

/* L1Scoring::score :23-72 (the ORB vocabulary's scoring) */
double orc_vocab_score_l1(const uint32_t *w1, const double *v1, int n1, const uint32_t *w2, const double *v2, int n2)
{
    int i = 0, j = 0;
    double score = 0;
    while (i < n1 && j < n2) {
        if (w1[i] == w2[j]) {
            const double vi = v1[i], wi = v2[j];
            score += fabs(vi - wi) - fabs(vi) - fabs(wi);
            ++i;
            ++j;
        } else if (w1[i] < w2[j]) {
            while (i < n1 && w1[i] < w2[j]) ++i; /* lower_bound */
        } else {
            while (j < n2 && w2[j] < w1[i]) ++j;
        }
    }
    score = -score / 2.0;
    return score;
}
