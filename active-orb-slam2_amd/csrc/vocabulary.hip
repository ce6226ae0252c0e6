// ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) on gfx950: the tree descent
// of transform() (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1215-1260), the BowVector /
// FeatureVector assembly of :1140-1187 and the two file loaders (:1351-1431, :1456-1496).
//
// Device layout: nodes are renumbered breadth-first into "slots" so that the children of a node
// are contiguous; a slot record carries the descriptor AND the node's own child range, so every
// level of the descent costs exactly one dependent load (the k child records, 48 B each,
// contiguous).  16 lanes serve one feature (one child per lane, more than 16 children in rounds),
// 4 features per wave; the per-level argmin is a 16-lane DPP row reduction over dist<<8|j, which
// keeps the reference's "first child wins ties" rule (strict '<', :1240).
//
// Assembly (one workgroup per image): the two std::maps become key-ascending arrays by a bitonic
// sort of (key<<32 | feature) in LDS; BowVector values repeat the reference's double additions
// (w + w + ... in arrival order; all addends of a word are equal) and its ascending-word norm
// (a sequential sum: the order of a double sum is part of the result).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "aos2_common.h"

using namespace aos2;

namespace {

struct VocRec {             // 48 bytes
    uint32_t d[8];          // node descriptor
    int32_t first_child;    // slot of the first child
    int32_t n_child;        // 0 = leaf (Node::isLeaf(), :341)
    uint32_t node_id;       // NodeId in the reference's numbering
    uint32_t word_id;       // WordId (leaves)
};
static_assert(sizeof(VocRec) == 48, "record layout");

struct HostNode {
    uint32_t parent = 0;
    uint8_t desc[32] = {};
    double weight = 0;
    uint32_t word_id = 0;
    std::vector<uint32_t> children;
};

template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, kCtrl, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t row_min_u32(uint32_t k)  // min over the 16 lanes of a DPP row, in every lane
{
    k = min(k, dpp_u32<0xB1>(k));
    k = min(k, dpp_u32<0x4E>(k));
    k = min(k, dpp_u32<0x141>(k));
    k = min(k, dpp_u32<0x140>(k));
    return k;
}

// ---- descent: word, node at level L - levelsup and weight of every feature
__global__ __launch_bounds__(256) void voc_descend_kernel(const VocRec *__restrict__ rec, const double *__restrict__ slot_weight,
                                                          int root_first, int root_count, int nid_level,
                                                          const uint8_t *__restrict__ desc, const int32_t *__restrict__ n_feat,
                                                          int cap, uint32_t *__restrict__ word_of, uint32_t *__restrict__ node_of,
                                                          double *__restrict__ weight_of)
{
    const int b = blockIdx.y;
    const int n = n_feat[b];
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int i = (int)(blockIdx.x * 16 + (threadIdx.x >> 4));
    bool active = i < n;
    const size_t base = (size_t)b * cap;
    uint4 f0 = make_uint4(0, 0, 0, 0), f1 = f0;
    if (active) {
        const uint4 *fp = reinterpret_cast<const uint4 *>(desc + (base + i) * 32);
        f0 = fp[0];
        f1 = fp[1];
    }
    int first = root_first, cnt = root_count, level = 0;
    uint32_t nid = 0;  // root when nid_level <= 0; also the value kept when a leaf sits above nid_level
    uint32_t word = 0;
    int leaf_slot = 0;
    while (__any(active)) {
        ++level;
        uint32_t key = 0xFFFFFFFFu;
        int b_first = 0, b_cnt = 0;
        uint32_t b_node = 0, b_word = 0;
        if (active) {
            for (int j = sub; j < cnt; j += 16) {
                const uint4 *rp = reinterpret_cast<const uint4 *>(rec + first + j);
                const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
                const int d = __popc(f0.x ^ r0.x) + __popc(f0.y ^ r0.y) + __popc(f0.z ^ r0.z) + __popc(f0.w ^ r0.w) +
                              __popc(f1.x ^ r1.x) + __popc(f1.y ^ r1.y) + __popc(f1.z ^ r1.z) + __popc(f1.w ^ r1.w);
                const uint32_t kk = ((uint32_t)d << 16) | (uint32_t)j;   // children per node < 65536
                if (kk < key) {
                    key = kk;
                    b_first = (int)r2.x;
                    b_cnt = (int)r2.y;
                    b_node = r2.z;
                    b_word = r2.w;
                }
            }
        }
        const uint32_t best = row_min_u32(key);
        // the lane owning the winner broadcasts its record tail to the 16-lane group
        const int owner = (lane & ~15) | (int)((best & 0xFFFFu) & 15u);
        const int w_first = __shfl(b_first, owner), w_cnt = __shfl(b_cnt, owner);
        const uint32_t w_node = (uint32_t)__shfl((int)b_node, owner), w_word = (uint32_t)__shfl((int)b_word, owner);
        if (active) {
            if (level == nid_level) nid = w_node;
            if (w_cnt == 0) {
                word = w_word;
                leaf_slot = first + (int)(best & 0xFFFFu);
                active = false;
            } else {
                first = w_first;
                cnt = w_cnt;
            }
        }
    }
    if (i < n && sub == 0) {
        word_of[base + i] = word;
        node_of[base + i] = nid;
        weight_of[base + i] = slot_weight[leaf_slot];
    }
}

// ---- bitonic sort of `np2` (power of two) u64 keys in LDS by a 256-thread workgroup
__device__ void bitonic_sort_u64(unsigned long long *a, int np2)
{
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < np2; t += 256) {
                const int ixj = t ^ j;
                if (ixj > t) {
                    const unsigned long long x = a[t], y = a[ixj];
                    const bool up = (t & k) == 0;
                    if ((x > y) == up) {
                        a[t] = y;
                        a[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

struct AssembleOut {
    uint32_t *bow_word;   // [batch][cap]
    double *bow_value;    // [batch][cap]
    int32_t *n_bow;       // [batch]
    int32_t *fv_node;     // [batch][cap]
    int32_t *fv_off;      // [batch][cap + 1]
    int32_t *fv_idx;      // [batch][cap]
    int32_t *n_fv;        // [batch]
};

// block-wide exclusive scan of one int per thread (256 threads)
__device__ int block_excl_scan(int v, int *sh, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += sh[w];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return off + x - v;
}

__global__ __launch_bounds__(256) void voc_assemble_kernel(const uint32_t *__restrict__ word_of, const uint32_t *__restrict__ node_of,
                                                           const double *__restrict__ weight_of, const int32_t *__restrict__ n_feat,
                                                           int cap, int np2, int tf, int must, int l2, AssembleOut O)
{
    extern __shared__ unsigned long long keys[];
    __shared__ int sh[4];
    __shared__ double s_norm;
    const int b = blockIdx.x;
    const int n = n_feat[b];
    const size_t base = (size_t)b * cap;
    const unsigned long long NONE = ~0ull;

    // ---------------- FeatureVector: (node, feature) ascending; stopped words (w <= 0) dropped (:1176-1180)
    for (int t = threadIdx.x; t < np2; t += 256) {
        unsigned long long k = NONE;
        if (t < n && weight_of[base + t] > 0) k = ((unsigned long long)node_of[base + t] << 32) | (unsigned)t;
        keys[t] = k;
    }
    __syncthreads();
    bitonic_sort_u64(keys, np2);
    int carry = 0;  // groups emitted so far (uniform)
    int kept = 0;
    for (int t0 = 0; t0 < np2; t0 += 256) {
        const int t = t0 + threadIdx.x;
        const unsigned long long k = keys[t];
        const bool valid = k != NONE;
        const bool head = valid && (t == 0 || (uint32_t)(keys[t - 1] >> 32) != (uint32_t)(k >> 32));
        int tot;
        const int pos = carry + block_excl_scan(head ? 1 : 0, sh, tot);
        if (valid) O.fv_idx[base + t] = (int32_t)(uint32_t)k;
        if (head) {
            O.fv_node[base + pos] = (int32_t)(uint32_t)(k >> 32);
            O.fv_off[(size_t)b * (cap + 1) + pos] = t;
        }
        carry += tot;
        int tv;
        block_excl_scan(valid ? 1 : 0, sh, tv);
        kept += tv;
    }
    if (threadIdx.x == 0) {
        O.fv_off[(size_t)b * (cap + 1) + carry] = kept;
        O.n_fv[b] = carry;
    }
    __syncthreads();

    // ---------------- BowVector: (word, feature) ascending -> one entry per word
    for (int t = threadIdx.x; t < np2; t += 256) {
        unsigned long long k = NONE;
        if (t < n && weight_of[base + t] > 0) k = ((unsigned long long)word_of[base + t] << 32) | (unsigned)t;
        keys[t] = k;
    }
    __syncthreads();
    bitonic_sort_u64(keys, np2);
    carry = 0;
    for (int t0 = 0; t0 < np2; t0 += 256) {
        const int t = t0 + threadIdx.x;
        const unsigned long long k = keys[t];
        const bool valid = k != NONE;
        const uint32_t w = (uint32_t)(k >> 32);
        const bool head = valid && (t == 0 || (uint32_t)(keys[t - 1] >> 32) != w);
        int tot;
        const int pos = carry + block_excl_scan(head ? 1 : 0, sh, tot);
        if (head) {
            // addWeight (:30-42): the word's weight added once per occurrence, in arrival order
            const double wt = weight_of[base + (uint32_t)k];
            double v = wt;
            if (tf)
                for (int u = t + 1; u < np2 && (uint32_t)(keys[u] >> 32) == w && keys[u] != NONE; ++u) v += wt;
            O.bow_word[base + pos] = w;
            O.bow_value[base + pos] = v;
        }
        carry += tot;
    }
    const int nb = carry;
    __syncthreads();
    if (threadIdx.x == 0) {
        O.n_bow[b] = nb;
        double norm = 0.0;
        if (must) {  // BowVector::normalize (:58-82): ascending-word sequential sum
            if (!l2) {
                for (int j = 0; j < nb; ++j) norm += fabs(O.bow_value[base + j]);
            } else {
                for (int j = 0; j < nb; ++j) {
                    const double v = O.bow_value[base + j];
                    norm = __dadd_rn(norm, __dmul_rn(v, v));
                }
                norm = sqrt(norm);
            }
        } else if (tf) {
            norm = (double)nb;  // :1183-1187 (TF weighting without normalisation: divide by the size)
        }
        s_norm = norm;
    }
    __syncthreads();
    const double norm = s_norm;
    if (norm > 0.0)
        for (int j = threadIdx.x; j < nb; j += 256) O.bow_value[base + j] = O.bow_value[base + j] / norm;
}

}  // namespace

struct aos2_vocabulary {
    int device = 0;
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<HostNode> nodes;   // m_nodes (node 0 = root)
    uint32_t n_words = 0;          // m_words.size()
    bool dev_ready = false, tree_on_device = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {};
    DevBuf<VocRec> d_rec;
    DevBuf<double> d_weight;
    int root_first = 0, root_count = 0;
    DevBuf<uint32_t> d_word, d_node;
    DevBuf<double> d_wt;
    DevBuf<uint8_t> d_io;
    PinnedBuf<uint8_t> h_io;
    float last_ms = 0;
};

static void voc_reset(aos2_vocabulary *v)
{
    v->nodes.clear();
    v->n_words = 0;
    v->tree_on_device = false;
}

static void voc_append(aos2_vocabulary *v, uint32_t nid, uint32_t parent, const uint8_t *desc, double weight, bool leaf)
{
    HostNode &n = v->nodes[nid];
    n.parent = parent;
    v->nodes[parent].children.push_back(nid);
    memcpy(n.desc, desc, 32);
    n.weight = weight;
    if (leaf) n.word_id = v->n_words++;
}

static int voc_init_device(aos2_vocabulary *v)
{
    int st = bind_device(v->device);
    if (st) return st;
    if (v->dev_ready) return AOS2_OK;
    if (int st_ = stream_create(&v->stream, stream_priority_env("AOS2_PRIO_VOCABULARY"))) return st_;
    for (auto &e : v->ev) AOS2_HIP_CHECK(hipEventCreate(&e));
    v->dev_ready = true;
    return AOS2_OK;
}

// breadth-first slot numbering + upload
static int voc_upload(aos2_vocabulary *v)
{
    int st = voc_init_device(v);
    if (st) return st;
    if (v->tree_on_device) return AOS2_OK;
    const size_t nn = v->nodes.size();
    std::vector<VocRec> rec;
    std::vector<double> wt;
    rec.reserve(nn);
    wt.reserve(nn);
    std::vector<uint32_t> order;  // node id of slot s
    order.reserve(nn);
    for (uint32_t c : v->nodes[0].children) order.push_back(c);
    v->root_first = 0;
    v->root_count = (int)v->nodes[0].children.size();
    for (size_t s = 0; s < order.size(); ++s) {
        const HostNode &n = v->nodes[order[s]];
        VocRec r;
        memcpy(r.d, n.desc, 32);
        r.first_child = (int32_t)order.size();
        r.n_child = (int32_t)n.children.size();
        r.node_id = order[s];
        r.word_id = n.word_id;
        for (uint32_t c : n.children) order.push_back(c);
        rec.push_back(r);
        wt.push_back(n.weight);
    }
    if ((st = v->d_rec.alloc(rec.size() + 1))) return st;
    if ((st = v->d_weight.alloc(wt.size() + 1))) return st;
    if (!rec.empty()) {
        AOS2_HIP_CHECK(hipMemcpy(v->d_rec.p, rec.data(), rec.size() * sizeof(VocRec), hipMemcpyHostToDevice));
        AOS2_HIP_CHECK(hipMemcpy(v->d_weight.p, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    v->tree_on_device = true;
    return AOS2_OK;
}

extern "C" {

int aos2_vocabulary_create(int device, aos2_vocabulary_t **out)
{
    if (!out) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    aos2_vocabulary *v = new aos2_vocabulary();
    v->device = device;
    *out = v;
    return AOS2_OK;
}

void aos2_vocabulary_destroy(aos2_vocabulary_t *v)
{
    if (!v) return;
    if (v->dev_ready) {
        (void)hipSetDevice(v->device);
        (void)hipStreamSynchronize(v->stream);
        v->d_rec.release(); v->d_weight.release(); v->d_word.release(); v->d_node.release(); v->d_wt.release();
        v->d_io.release(); v->h_io.release();
        for (auto &e : v->ev) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(v->stream);
    }
    delete v;
}

int aos2_vocabulary_set_nodes(aos2_vocabulary_t *v, int k, int L, int scoring, int weighting, int n_nodes,
                              const int32_t *parent, const uint8_t *desc, const double *weight, const uint8_t *is_leaf)
{
    if (!v || n_nodes < 0 || (n_nodes > 0 && (!parent || !desc || !weight || !is_leaf)) || scoring < 0 || scoring > 5 ||
        weighting < 0 || weighting > 3) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    voc_reset(v);
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    v->nodes.resize((size_t)n_nodes + 1);
    for (int i = 0; i < n_nodes; ++i) {
        if (parent[i] < 0 || parent[i] > i) {
            voc_reset(v);
            set_error("node %d: parent %d does not precede it", i + 1, parent[i]);
            return AOS2_ERR_ARG;
        }
        voc_append(v, (uint32_t)i + 1, (uint32_t)parent[i], desc + (size_t)i * 32, weight[i], is_leaf[i] != 0);
    }
    return AOS2_OK;
}

// loadFromBinaryFile (:1456-1496).  The reference loops `while(!f.eof())`: the read after the last record
// fails, the buffer still holds that record, and it becomes node nb_nodes once more (a duplicate last
// child; it never wins a tie, but size() counts its word).  Reproduced.
int aos2_vocabulary_load_binary(aos2_vocabulary_t *v, const char *filename)
{
    if (!v || !filename) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    std::ifstream f(filename, std::ios::in | std::ios::binary);
    if (!f.is_open()) {
        set_error("cannot open %s", filename);
        return AOS2_ERR_ARG;
    }
    uint32_t nb_nodes = 0, size_node = 0;
    int32_t hdr[4] = {};
    f.read(reinterpret_cast<char *>(&nb_nodes), 4);
    f.read(reinterpret_cast<char *>(&size_node), 4);
    f.read(reinterpret_cast<char *>(hdr), 16);
    if (!f.good() || size_node < 41 || size_node > 4096 || hdr[2] < 0 || hdr[2] > 5 || hdr[3] < 0 || hdr[3] > 3) {
        set_error("%s is not a binary vocabulary", filename);
        return AOS2_ERR_ARG;
    }
    voc_reset(v);
    v->k = hdr[0]; v->L = hdr[1]; v->scoring = hdr[2]; v->weighting = hdr[3];
    v->nodes.resize((size_t)nb_nodes + 1);
    std::vector<char> buf(size_node, 0);
    uint32_t nid = 1;
    while (!f.eof()) {
        f.read(buf.data(), size_node);
        if (nid > nb_nodes) break;  // the reference would write past m_nodes here
        int32_t parent;
        float w;
        memcpy(&parent, buf.data(), 4);
        memcpy(&w, buf.data() + 36, 4);
        if (parent < 0 || (uint32_t)parent >= nid) {
            voc_reset(v);
            set_error("%s: node %u has parent %d", filename, nid, parent);
            return AOS2_ERR_ARG;
        }
        voc_append(v, nid, (uint32_t)parent, reinterpret_cast<const uint8_t *>(buf.data()) + 4, (double)w, buf[40] != 0);
        ++nid;
    }
    return AOS2_OK;
}

// saveToBinaryFile (:1500-1521)
int aos2_vocabulary_save_binary(const aos2_vocabulary_t *v, const char *filename)
{
    if (!v || !filename) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    std::ofstream f(filename, std::ios::out | std::ios::binary);
    if (!f.is_open()) {
        set_error("cannot open %s", filename);
        return AOS2_ERR_ARG;
    }
    const uint32_t nb_nodes = (uint32_t)v->nodes.size(), size_node = 41;
    const int32_t hdr[4] = {v->k, v->L, v->scoring, v->weighting};
    f.write(reinterpret_cast<const char *>(&nb_nodes), 4);
    f.write(reinterpret_cast<const char *>(&size_node), 4);
    f.write(reinterpret_cast<const char *>(hdr), 16);
    for (uint32_t i = 1; i < nb_nodes; ++i) {
        const HostNode &n = v->nodes[i];
        const float w = (float)n.weight;
        const char leaf = n.children.empty() ? 1 : 0;
        f.write(reinterpret_cast<const char *>(&n.parent), 4);
        f.write(reinterpret_cast<const char *>(n.desc), 32);
        f.write(reinterpret_cast<const char *>(&w), 4);
        f.write(&leaf, 1);
    }
    return f.good() ? AOS2_OK : AOS2_ERR_ARG;
}

// loadFromTextFile (:1351-1431).  `while(!f.eof()) getline` turns the empty string after a final newline
// into one more node (every `>>` fails -> parent 0, not a leaf by flag, zero descriptor, weight 0, no
// word): a childless child of the root that stops the features that land on it.  Reproduced.
int aos2_vocabulary_load_text(aos2_vocabulary_t *v, const char *filename)
{
    if (!v || !filename) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    std::ifstream f(filename);
    if (!f.is_open()) {
        set_error("cannot open %s", filename);
        return AOS2_ERR_ARG;
    }
    std::string s;
    std::getline(f, s);
    std::stringstream ss(s);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    ss >> k >> L >> n1 >> n2;
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        set_error("Vocabulary loading failure: This is not a correct text file!");
        return AOS2_ERR_ARG;
    }
    voc_reset(v);
    v->k = k; v->L = L; v->scoring = n1; v->weighting = n2;
    v->nodes.resize(1);
    while (!f.eof()) {
        std::string snode;
        std::getline(f, snode);
        std::stringstream sn(snode);
        const uint32_t nid = (uint32_t)v->nodes.size();
        v->nodes.resize((size_t)nid + 1);
        int pid = 0, leaf = 0;
        sn >> pid;
        sn >> leaf;
        uint8_t d[32] = {};
        for (int i = 0; i < 32; ++i) {
            int x = 0;
            sn >> x;
            if (!sn.fail()) d[i] = (uint8_t)x;
        }
        double w = 0;
        sn >> w;
        if (sn.fail() && snode.empty()) { pid = 0; leaf = 0; w = 0; }
        if (pid < 0 || (uint32_t)pid >= nid) {
            voc_reset(v);
            set_error("%s: node %u has parent %d", filename, nid, pid);
            return AOS2_ERR_ARG;
        }
        voc_append(v, nid, (uint32_t)pid, d, w, leaf > 0);
    }
    return AOS2_OK;
}

int aos2_vocabulary_k(const aos2_vocabulary_t *v) { return v ? v->k : 0; }
int aos2_vocabulary_levels(const aos2_vocabulary_t *v) { return v ? v->L : 0; }
int aos2_vocabulary_scoring(const aos2_vocabulary_t *v) { return v ? v->scoring : 0; }
int aos2_vocabulary_weighting(const aos2_vocabulary_t *v) { return v ? v->weighting : 0; }
int aos2_vocabulary_nodes(const aos2_vocabulary_t *v) { return v ? (int)v->nodes.size() : 0; }
unsigned aos2_vocabulary_size(const aos2_vocabulary_t *v) { return v ? v->n_words : 0; }
int aos2_vocabulary_empty(const aos2_vocabulary_t *v) { return !v || v->n_words == 0; }
float aos2_vocabulary_last_device_ms(const aos2_vocabulary_t *v) { return v ? v->last_ms : 0.0f; }
void *aos2_vocabulary_stream(aos2_vocabulary_t *v)
{
    if (!v || voc_init_device(v)) return nullptr;
    return v->stream;
}

static int voc_run(aos2_vocabulary *v, int batch, const uint8_t *d_desc, const int32_t *d_n, int cap, int levelsup,
                   const AssembleOut &O, uint32_t *d_word_of, uint32_t *d_node_of)
{
    int st;
    if (cap > 8192) {
        set_error("transform: more than 8192 features per image");
        return AOS2_ERR_CAPACITY;
    }
    if ((st = voc_upload(v))) return st;
    const size_t tot = (size_t)batch * cap;
    if (!d_word_of) {
        if ((st = v->d_word.alloc(tot))) return st;
        d_word_of = v->d_word.p;
    }
    if (!d_node_of) {
        if ((st = v->d_node.alloc(tot))) return st;
        d_node_of = v->d_node.p;
    }
    if ((st = v->d_wt.alloc(tot))) return st;
    int np2 = 256;
    while (np2 < cap) np2 <<= 1;
    const int must = v->scoring != 5;           // every scoring but DOT_PRODUCT normalises (ScoringObject.h:74-89)
    const int l2 = v->scoring == 1;             // L2_NORM; all others use the L1 norm
    const int tf = v->weighting == 0 || v->weighting == 1;  // TF_IDF, TF -> addWeight; IDF, BINARY -> addIfNotExist
    AOS2_HIP_CHECK(hipEventRecord(v->ev[0], v->stream));
    hipLaunchKernelGGL(voc_descend_kernel, dim3((cap + 15) / 16, batch), dim3(256), 0, v->stream, v->d_rec.p, v->d_weight.p,
                       v->root_first, v->root_count, v->L - levelsup, d_desc, d_n, cap, d_word_of, d_node_of, v->d_wt.p);
    AOS2_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(voc_assemble_kernel, dim3(batch), dim3(256), sizeof(unsigned long long) * (size_t)np2, v->stream,
                       d_word_of, d_node_of, v->d_wt.p, d_n, cap, np2, tf, must, l2, O);
    AOS2_HIP_CHECK(hipGetLastError());
    AOS2_HIP_CHECK(hipEventRecord(v->ev[1], v->stream));
    AOS2_HIP_CHECK(hipStreamSynchronize(v->stream));
    AOS2_HIP_CHECK(hipEventElapsedTime(&v->last_ms, v->ev[0], v->ev[1]));
    return AOS2_OK;
}

int aos2_vocabulary_transform_device(aos2_vocabulary_t *v, int batch, const uint8_t *d_desc, const int32_t *d_n, int cap,
                                     int levelsup, uint32_t *d_bow_word, double *d_bow_value, int32_t *d_n_bow,
                                     int32_t *d_fv_node, int32_t *d_fv_off, int32_t *d_fv_idx, int32_t *d_n_fv,
                                     uint32_t *d_word_of, uint32_t *d_node_of)
{
    if (!v || batch <= 0 || cap <= 0 || !d_desc || !d_n || !d_bow_word || !d_bow_value || !d_n_bow || !d_fv_node ||
        !d_fv_off || !d_fv_idx || !d_n_fv) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (aos2_vocabulary_empty(v) || v->nodes[0].children.empty()) {
        set_error("transform: empty vocabulary");
        return AOS2_ERR_ARG;
    }
    const AssembleOut O{d_bow_word, d_bow_value, d_n_bow, d_fv_node, d_fv_off, d_fv_idx, d_n_fv};
    return voc_run(v, batch, d_desc, d_n, cap, levelsup, O, d_word_of, d_node_of);
}

int aos2_vocabulary_transform(aos2_vocabulary_t *v, const uint8_t *desc, int n, int levelsup, uint32_t *bow_word,
                              double *bow_value, int *n_bow, int32_t *fv_node, int32_t *fv_off, int32_t *fv_idx, int *n_fv,
                              uint32_t *word_of, uint32_t *node_of)
{
    if (!v || n < 0 || !n_bow || !n_fv || !fv_off || (n > 0 && (!desc || !bow_word || !bow_value || !fv_node || !fv_idx))) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    *n_bow = 0;
    *n_fv = 0;
    fv_off[0] = 0;
    if (aos2_vocabulary_empty(v) || v->nodes[0].children.empty()) return AOS2_OK;  // v.clear(); fv.clear(); return (:1147-1150)
    if (n == 0) return AOS2_OK;
    int st;
    if ((st = voc_init_device(v))) return st;
    const int cap = (n + 3) & ~3;
    // device block: desc | n | bow_word | fv_node | fv_off | fv_idx | word_of | node_of | counts | bow_value
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_desc = take((size_t)cap * 32), o_n = take(4), o_bw = take(4 * (size_t)cap), o_fn = take(4 * (size_t)cap),
                 o_fo = take(4 * ((size_t)cap + 1)), o_fi = take(4 * (size_t)cap), o_wo = take(4 * (size_t)cap),
                 o_no = take(4 * (size_t)cap), o_cnt = take(8), o_bv = take(8 * (size_t)cap);
    if ((st = v->d_io.alloc(off))) return st;
    if ((st = v->h_io.alloc(off))) return st;
    uint8_t *hp = v->h_io.p, *dp = v->d_io.p;
    memcpy(hp + o_desc, desc, (size_t)n * 32);
    const int32_t n32 = n;
    memcpy(hp + o_n, &n32, 4);
    AOS2_HIP_CHECK(hipMemcpyAsync(dp, hp, o_bw, hipMemcpyHostToDevice, v->stream));
    const AssembleOut O{(uint32_t *)(dp + o_bw), (double *)(dp + o_bv), (int32_t *)(dp + o_cnt), (int32_t *)(dp + o_fn),
                        (int32_t *)(dp + o_fo), (int32_t *)(dp + o_fi), (int32_t *)(dp + o_cnt) + 1};
    st = voc_run(v, 1, dp + o_desc, (const int32_t *)(dp + o_n), cap, levelsup, O, (uint32_t *)(dp + o_wo), (uint32_t *)(dp + o_no));
    if (st) return st;
    AOS2_HIP_CHECK(hipMemcpyAsync(hp + o_bw, dp + o_bw, off - o_bw, hipMemcpyDeviceToHost, v->stream));
    AOS2_HIP_CHECK(hipStreamSynchronize(v->stream));
    int32_t cnt[2];
    memcpy(cnt, hp + o_cnt, 8);
    *n_bow = cnt[0];
    *n_fv = cnt[1];
    memcpy(bow_word, hp + o_bw, 4 * (size_t)cnt[0]);
    memcpy(bow_value, hp + o_bv, 8 * (size_t)cnt[0]);
    memcpy(fv_node, hp + o_fn, 4 * (size_t)cnt[1]);
    memcpy(fv_off, hp + o_fo, 4 * ((size_t)cnt[1] + 1));
    memcpy(fv_idx, hp + o_fi, 4 * (size_t)fv_off[cnt[1]]);
    if (word_of) memcpy(word_of, hp + o_wo, 4 * (size_t)n);
    if (node_of) memcpy(node_of, hp + o_no, 4 * (size_t)n);
    return AOS2_OK;
}

// score(v1, v2) for L1_NORM (L1Scoring::score, ScoringObject.cpp:23-72): host scalar helper over two
// key-ascending BowVectors (what ORB-SLAM2's KeyFrameDatabase / LoopClosing call per keyframe pair).
int aos2_vocabulary_score(const aos2_vocabulary_t *v, const uint32_t *w1, const double *v1, int n1, const uint32_t *w2,
                          const double *v2, int n2, double *score)
{
    if (!v || !score || n1 < 0 || n2 < 0 || (n1 > 0 && (!w1 || !v1)) || (n2 > 0 && (!w2 || !v2))) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (v->scoring != 0) {
        set_error("score: only L1_NORM (the ORB vocabulary's scoring) is implemented");
        return AOS2_ERR_ARG;
    }
    int i = 0, j = 0;
    double s = 0;
    while (i < n1 && j < n2) {
        if (w1[i] == w2[j]) {
            const double vi = v1[i], wi = v2[j];
            s += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi);
            ++i;
            ++j;
        } else if (w1[i] < w2[j]) {
            i = (int)(std::lower_bound(w1 + i, w1 + n1, w2[j]) - w1);
        } else {
            j = (int)(std::lower_bound(w2 + j, w2 + n2, w1[i]) - w2);
        }
    }
    *score = -s / 2.0;
    return AOS2_OK;
}

}  // extern "C"
