// Quad-tree keypoint distribution (ORBextractor::DistributeOctTree + ExtractorNode::DivideNode,
// reference src/ORBextractor.cc:481-763) as an allocation-free routine over caller-provided
// scratch, written so the same source runs on the host (thread per (image, level)) and inside a
// HIP kernel (one lane per (image, level)).
//
// Formulation differences from the reference (results are identical):
//  * a node is an axis-aligned rectangle [x0,x1) x [y0,y1) (the reference stores 4 corners that
//    always form one);
//  * a node's keys are a contiguous segment of one permutation array; DivideNode is a stable
//    4-way partition of that segment (the reference copies KeyPoints into 4 new vectors);
//  * std::list<ExtractorNode> is an index-linked list over a node arena;
//  * the (size, pointer) sort of :684 uses the arena index as the pointer stand-in (creation
//    order; later-created = larger), DESIGN.md parity convention 1.
// Candidate coordinates are integer pixel positions relative to (minBorderX, minBorderY).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
// always inlined on the device: the LDS instantiation relies on address-space inference (ds_*
// instead of flat_* accesses), which stops at call boundaries
#define AOS2_OCT_HD __host__ __device__ __forceinline__
#else
#define AOS2_OCT_HD inline
#endif

#if defined(AOS2_OCT_PROF) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ long long g_oct_prof[16];
// phase counters accumulate in registers and reach memory once, when the job ends (atomics per tick perturbed the
// very latencies they were measuring)
#define OCT_T0()                          \
    long long t_prof = wall_clock64();    \
    long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define OCT_TICK(i)                                \
    do {                                           \
        const long long t_now = wall_clock64();    \
        prof_acc[i] += t_now - t_prof;             \
        t_prof = t_now;                            \
    } while (0)
#define OCT_COUNT(i, v) prof_acc[i] += (long long)(v)
#define OCT_FLUSH()                                                                                              \
    do {                                                                                                         \
        if (prof_on && (threadIdx.x & 63) == 0)                                                                  \
            for (int i_ = 0; i_ < 16; ++i_)                                                                      \
                if (prof_acc[i_]) atomicAdd((unsigned long long *)&g_oct_prof[i_], (unsigned long long)prof_acc[i_]); \
    } while (0)
#else
#define OCT_T0()
#define OCT_TICK(i)
#define OCT_COUNT(i, v)
#define OCT_FLUSH()
#endif

namespace aos2 {

// bNoMore of the reference (:1012-1019, :573) is exactly "holds one key", so it is not stored.
struct OctNode {
    int16_t x0, y0, x1, y1;
    int32_t beg, cnt;
    int32_t prev, next;
};
// compact node for LDS-resident jobs (n and the arena both < 32768)
struct OctNode16 {
    int16_t x0, y0, x1, y1;
    int16_t beg, cnt;
    int16_t prev, next;
};

// candidate access: split arrays (host, debug tap, device fallback) or the packed words the FAST
// kernel emits (x | y << 12 | score << 24)
struct OctCandsSplit {
    const int16_t *xs, *ys;
    const uint8_t *sc;
    AOS2_OCT_HD int x(int k) const { return xs[k]; }
    AOS2_OCT_HD int y(int k) const { return ys[k]; }
    AOS2_OCT_HD int score(int k) const { return sc[k]; }
};
struct OctCandsPacked {
    const uint32_t *c;
    AOS2_OCT_HD int x(int k) const { return (int)(c[k] & 0xfffu); }
    AOS2_OCT_HD int y(int k) const { return (int)((c[k] >> 12) & 0xfffu); }
    AOS2_OCT_HD int score(int k) const { return (int)(c[k] >> 24); }
};
struct OctWide {
    using Node = OctNode;
    using Idx = int32_t;
    using Cands = OctCandsSplit;
};
struct OctCompact {
    using Node = OctNode16;
    using Idx = int16_t;
    using Cands = OctCandsPacked;
};

// scratch requirements for n candidates and target N:
//   nodes: arena of at most oct_max_nodes(n, N) OctNode
//   perm, tmp: n int32 each; pairs: 2 * oct_max_nodes int32
AOS2_OCT_HD int oct_max_nodes(int n, int N)
{
    // every split allocates <= 4 nodes and the list never exceeds min(n, N+3) live leaves; the
    // number of splits is bounded by the number of nodes ever created.  A node is split at most
    // once, and each split either increases the live count or keeps it (degenerate), the latter
    // ends the pass.  Conservative bound used for sizing:
    // Productive splits <= live leaves; degenerate splits (all keys fall into one child) form
    // chains no longer than the box-halving depth (<= 12 for 4096-px levels).
    int live = (n < N + 3 ? n : N + 3) + 8;
    return 48 * live + 64;
}

template <class Tr>
struct OctScratchT {
    typename Tr::Node *nodes;
    typename Tr::Idx *perm;
    typename Tr::Idx *tmp;
    int32_t *pairs_a;  // (size,node) pairs of the current pass
    int32_t *pairs_b;  // previous pass (sorted)
    int max_nodes;
    int max_pairs;     // capacity of each pairs array, in pairs
};
using OctScratch = OctScratchT<OctWide>;

// Execution policy of the O(n) inner loops (stable partitions, per-node maxima):
//   SerialCoop -- one thread does everything (host threads; also what the CPU tests run)
//   WaveCoop   -- device: all 64 lanes of ONE wave run the list/tree control flow redundantly on
//                 wave-uniform data (same loads, same stores), and share the key loops through
//                 ballots; the results are identical because a stable partition is unique.
struct SerialCoop {
    static constexpr bool kWave = false;
    static constexpr bool kGroup = false;
};
#if defined(__HIPCC__)
struct WaveCoop {
    static constexpr bool kWave = true;
    static constexpr bool kGroup = false;
};
//   GroupCoop  -- WaveCoop whose big stable partitions (>= kGroupMin keys) are shared with the other waves of the job's
//                 workgroup: the job's wave posts the segment in an LDS mailbox, every wave partitions the 64-key chunks
//                 c = wave, wave + kGroupWaves, ..., the chunk counts are prefix-summed so that the partition stays the
//                 stable one (4 workgroup barriers per such divide).  Everything else is the one-wave code.
struct GroupCoop {
    static constexpr bool kWave = true;
    static constexpr bool kGroup = true;
};
#endif

namespace octdetail {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline int coop_lane() { return (int)(threadIdx.x & 63); }
// wave-local fence: the memory operations of one wave complete in order, so waiting for them (and keeping the
// compiler from reordering) is all a one-wave job needs; no workgroup barrier, so several independent jobs can
// share a workgroup
__device__ inline void coop_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
__device__ inline unsigned long long coop_ballot(bool p) { return __ballot(p); }
__device__ inline int coop_popc(unsigned long long m) { return __popcll(m); }
__device__ inline int coop_shfl(int v, int src) { return __shfl(v, src); }   // lane-varying source: ds_bpermute_b32
// Wave reductions / scans on DPP (4 + 2 row-level steps, no LDS round trips; __shfl_xor / __shfl_up compile to
// ds_bpermute_b32, i.e. 6 dependent LDS accesses per call -- these helpers sit on the jobs' serial path).
__device__ inline int coop_row_reduce_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror
    return v;   // every lane: the sum of its row of 16
}
__device__ inline int coop_sum(int v)
{
    v = coop_row_reduce_add(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ inline int coop_max(int v)   // v >= 0
{
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ inline int coop_excl_scan(int v)  // exclusive prefix sum over the lanes
{
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1 (lanes without a source add 0)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8  -> inclusive scan inside each row
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return x - v;
}
__device__ inline int coop_ffs(unsigned long long m) { return __ffsll((long long)m) - 1; }
#else
inline int coop_lane() { return 0; }
inline int coop_ffs(unsigned long long m) { return __builtin_ffsll((long long)m) - 1; }
inline int coop_shfl(int v, int) { return v; }
inline int coop_sum(int v) { return v; }
inline int coop_max(int v) { return v; }
inline int coop_excl_scan(int) { return 0; }
inline void coop_sync() {}
inline unsigned long long coop_ballot(bool p) { return p ? 1ull : 0ull; }
inline int coop_popc(unsigned long long m) { return __builtin_popcountll(m); }
#endif

#if defined(__HIPCC__)
constexpr int kGroupWaves = 4, kGroupMin = 256, kGroupMaxChunks = 64;
struct GroupMail {
    int on;                 // the job runs with helper waves
    int cmd;                // 1 = partition the posted segment, 0 = the job is over
    int beg, cnt, mx, my;   // segment of perm, quadrant boundaries
    const void *cands;      // Tr::Cands::c
    void *perm, *tmp;       // Tr::Idx arrays
    int cc[kGroupMaxChunks][4];   // quadrant counts of every 64-key chunk
};
AOS2_OCT_HD GroupMail &group_mail()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ GroupMail m;
#else
    static GroupMail m;   // (host pass of the compiler: never runs)
#endif
    return m;
}
// every wave of the workgroup, after the barrier that published the order; cnt[] = the quadrant totals (all waves)
template <class Tr>
AOS2_OCT_HD void group_partition_body(int wave, int cnt[4])
{
#if defined(__HIP_DEVICE_COMPILE__)
    GroupMail &m = group_mail();
    const typename Tr::Cands C{static_cast<decltype(Tr::Cands::c)>(m.cands)};
    typename Tr::Idx *perm = static_cast<typename Tr::Idx *>(m.perm), *tmp = static_cast<typename Tr::Idx *>(m.tmp);
    const int beg = m.beg, n = m.cnt, mx = m.mx, my = m.my;
    const int lane = coop_lane(), nch = (n + 63) >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // chunk counts
    for (int c = wave; c < nch; c += kGroupWaves) {
        const int i = 64 * c + lane;
        const bool valid = i < n;
        const int k = valid ? perm[beg + i] : 0;
        const int q = valid ? ((C.x(k) < mx ? 0 : 1) + (C.y(k) < my ? 0 : 2)) : -1;
        const int c0 = coop_popc(coop_ballot(q == 0)), c1 = coop_popc(coop_ballot(q == 1));
        const int c2 = coop_popc(coop_ballot(q == 2)), c3 = coop_popc(coop_ballot(q == 3));
        if (lane == 0) {
            m.cc[c][0] = c0; m.cc[c][1] = c1; m.cc[c][2] = c2; m.cc[c][3] = c3;
        }
    }
    __syncthreads();
    // exclusive prefix over the chunks (lane = chunk), redundantly in every wave
    int v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (lane < nch) {
        v0 = m.cc[lane][0]; v1 = m.cc[lane][1]; v2 = m.cc[lane][2]; v3 = m.cc[lane][3];
    }
    const int e0 = coop_excl_scan(v0), e1 = coop_excl_scan(v1), e2 = coop_excl_scan(v2), e3 = coop_excl_scan(v3);
    cnt[0] = __builtin_amdgcn_readlane(e0 + v0, 63); cnt[1] = __builtin_amdgcn_readlane(e1 + v1, 63);
    cnt[2] = __builtin_amdgcn_readlane(e2 + v2, 63); cnt[3] = __builtin_amdgcn_readlane(e3 + v3, 63);
    const int o1 = cnt[0], o2 = cnt[0] + cnt[1], o3 = cnt[0] + cnt[1] + cnt[2];
    for (int c = wave; c < nch; c += kGroupWaves) {
        const int i = 64 * c + lane;
        const bool valid = i < n;
        const int k = valid ? perm[beg + i] : 0;
        const int q = valid ? ((C.x(k) < mx ? 0 : 1) + (C.y(k) < my ? 0 : 2)) : -1;
        const unsigned long long b0 = coop_ballot(q == 0), b1 = coop_ballot(q == 1);
        const unsigned long long b2 = coop_ballot(q == 2), b3 = coop_ballot(q == 3);
        const int p0 = __builtin_amdgcn_readlane(e0, c), p1 = o1 + __builtin_amdgcn_readlane(e1, c);
        const int p2 = o2 + __builtin_amdgcn_readlane(e2, c), p3 = o3 + __builtin_amdgcn_readlane(e3, c);
        if (valid) {
            const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
            const int base = q == 0 ? p0 : q == 1 ? p1 : q == 2 ? p2 : p3;
            tmp[base + coop_popc(bq & lt)] = (typename Tr::Idx)k;
        }
    }
    __syncthreads();
    for (int i = 64 * wave + lane; i < n; i += 64 * kGroupWaves) perm[beg + i] = tmp[i];
    __syncthreads();
#else
    (void)wave; (void)cnt;
#endif
}
// the other waves of a job's workgroup: serve partition orders until the job is over
template <class Tr>
AOS2_OCT_HD void group_helper_loop(int wave)
{
#if defined(__HIP_DEVICE_COMPILE__)
    GroupMail &m = group_mail();
    for (;;) {
        __syncthreads();
        if (m.cmd == 0) return;
        int cnt[4];
        group_partition_body<Tr>(wave, cnt);
    }
#else
    (void)wave;
#endif
}
#endif

template <class Node>
struct List {
    Node *nodes;
    int n_alloc, head, tail, size, cap;
};

template <class Node>
AOS2_OCT_HD int new_node(List<Node> &L)
{
    if (L.n_alloc >= L.cap) return -1;
    Node &n = L.nodes[L.n_alloc];
    n.prev = n.next = -1;
    n.cnt = 0;
    n.beg = 0;
    return L.n_alloc++;
}
template <class Node>
AOS2_OCT_HD void push_back(List<Node> &L, int id)
{
    Node &n = L.nodes[id];
    n.prev = (decltype(n.prev))L.tail;
    n.next = -1;
    if (L.tail >= 0) L.nodes[L.tail].next = (decltype(n.next))id; else L.head = id;
    L.tail = id;
    L.size++;
}
template <class Node>
AOS2_OCT_HD void push_front(List<Node> &L, int id)
{
    Node &n = L.nodes[id];
    n.next = (decltype(n.next))L.head;
    n.prev = -1;
    if (L.head >= 0) L.nodes[L.head].prev = (decltype(n.prev))id; else L.tail = id;
    L.head = id;
    L.size++;
}
template <class Node>
AOS2_OCT_HD int erase(List<Node> &L, int id)
{
    Node &n = L.nodes[id];
    int nx = n.next;
    if (n.prev >= 0) L.nodes[n.prev].next = n.next; else L.head = n.next;
    if (n.next >= 0) L.nodes[n.next].prev = n.prev; else L.tail = n.prev;
    L.size--;
    return nx;
}

// DivideNode: stable 4-way partition of the parent's segment; children c[0..3] = n1..n4
// (UL, UR, BL, BR quadrants).  Returns false if the arena is exhausted.
template <class Coop, class Tr>
AOS2_OCT_HD bool divide(List<typename Tr::Node> &L, int id, int c[4], int ccnt[4], int &next_of_id,
                        const typename Tr::Cands &C, typename Tr::Idx *perm, typename Tr::Idx *tmp)
{
    using Node = typename Tr::Node;
    using Idx = typename Tr::Idx;
    const Node p = L.nodes[id];
    next_of_id = p.next;
    const int hx = (p.x1 - p.x0 + 1) / 2;  // ceil(float(x1-x0)/2)
    const int hy = (p.y1 - p.y0 + 1) / 2;
    const int mx = p.x0 + hx, my = p.y0 + hy;
    // four arena slots in creation order n1..n4 (the creation index is the tie-break of the size sort, :681-685); the
    // records are built in registers and stored ONCE each, fully linked, after the partition (below)
    if (L.n_alloc + 4 > L.cap) return false;
    for (int i = 0; i < 4; ++i) c[i] = L.n_alloc + i;
    L.n_alloc += 4;
    int cnt[4] = {0, 0, 0, 0};
    int off[4];
    if (!Coop::kWave) {
        for (int i = 0; i < p.cnt; ++i) {
            const int k = perm[p.beg + i];
            const int q = (C.x(k) < mx ? 0 : 1) + (C.y(k) < my ? 0 : 2);
            cnt[q]++;
        }
        off[0] = 0; off[1] = cnt[0]; off[2] = cnt[0] + cnt[1]; off[3] = cnt[0] + cnt[1] + cnt[2];
        int fill[4] = {off[0], off[1], off[2], off[3]};
        for (int i = 0; i < p.cnt; ++i) {
            const int k = perm[p.beg + i];
            const int q = (C.x(k) < mx ? 0 : 1) + (C.y(k) < my ? 0 : 2);
            tmp[fill[q]++] = (Idx)k;
        }
        for (int i = 0; i < p.cnt; ++i) perm[p.beg + i] = tmp[i];
    } else {
        const int lane = coop_lane();
        const unsigned long long lt = (1ull << lane) - 1ull;
#if defined(__HIPCC__)
        bool grouped = false;
        if constexpr (Coop::kGroup) {
            GroupMail &gm = group_mail();
            if (gm.on && p.cnt >= kGroupMin && p.cnt <= 64 * kGroupMaxChunks) {
                if (lane == 0) {
                    gm.cmd = 1; gm.beg = p.beg; gm.cnt = p.cnt; gm.mx = mx; gm.my = my;
                }
#if defined(__HIP_DEVICE_COMPILE__)
                __syncthreads();   // the order (and everything this wave wrote before) is visible to the helpers
#endif
                group_partition_body<Tr>(0, cnt);
                off[0] = 0; off[1] = cnt[0]; off[2] = cnt[0] + cnt[1]; off[3] = cnt[0] + cnt[1] + cnt[2];
                grouped = true;
            }
        }
        if (grouped) {
        } else
#endif
        if (p.cnt <= 64) {
            // whole segment in registers: read, barrier, scatter in place
            const bool valid = lane < p.cnt;
            const int k = valid ? perm[p.beg + lane] : 0;
            const int q = valid ? ((C.x(k) < mx ? 0 : 1) + (C.y(k) < my ? 0 : 2)) : -1;
            const unsigned long long b0 = coop_ballot(q == 0), b1 = coop_ballot(q == 1);
            const unsigned long long b2 = coop_ballot(q == 2), b3 = coop_ballot(q == 3);
            cnt[0] = coop_popc(b0); cnt[1] = coop_popc(b1); cnt[2] = coop_popc(b2); cnt[3] = coop_popc(b3);
            off[0] = 0; off[1] = cnt[0]; off[2] = cnt[0] + cnt[1]; off[3] = cnt[0] + cnt[1] + cnt[2];
            coop_sync();
            if (valid) {
                const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
                perm[p.beg + off[q] + coop_popc(bq & lt)] = (Idx)k;
            }
            coop_sync();
        } else {
            for (int base = 0; base < p.cnt; base += 64) {
                const int i = base + lane;
                const bool valid = i < p.cnt;
                const int k = valid ? perm[p.beg + i] : 0;
                const int q = valid ? ((C.x(k) < mx ? 0 : 1) + (C.y(k) < my ? 0 : 2)) : -1;
                cnt[0] += coop_popc(coop_ballot(q == 0));
                cnt[1] += coop_popc(coop_ballot(q == 1));
                cnt[2] += coop_popc(coop_ballot(q == 2));
                cnt[3] += coop_popc(coop_ballot(q == 3));
            }
            off[0] = 0; off[1] = cnt[0]; off[2] = cnt[0] + cnt[1]; off[3] = cnt[0] + cnt[1] + cnt[2];
            int run[4] = {off[0], off[1], off[2], off[3]};
            for (int base = 0; base < p.cnt; base += 64) {
                const int i = base + lane;
                const bool valid = i < p.cnt;
                const int k = valid ? perm[p.beg + i] : 0;
                const int q = valid ? ((C.x(k) < mx ? 0 : 1) + (C.y(k) < my ? 0 : 2)) : -1;
                const unsigned long long b0 = coop_ballot(q == 0), b1 = coop_ballot(q == 1);
                const unsigned long long b2 = coop_ballot(q == 2), b3 = coop_ballot(q == 3);
                if (valid) {
                    const unsigned long long bq = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;
                    tmp[run[q] + coop_popc(bq & lt)] = (Idx)k;
                }
                run[0] += coop_popc(b0); run[1] += coop_popc(b1); run[2] += coop_popc(b2); run[3] += coop_popc(b3);
            }
            coop_sync();
            for (int i = lane; i < p.cnt; i += 64) perm[p.beg + i] = tmp[i];
            coop_sync();
        }
    }
    // unlink the parent (its prev / next are in registers), then push the non-empty children to the front in the
    // order n1..n4 (:1000-1040 push_front): the links are resolved in registers, so a child costs one record store
    if (p.prev >= 0) L.nodes[p.prev].next = p.next; else L.head = p.next;
    if (p.next >= 0) L.nodes[p.next].prev = p.prev; else L.tail = p.prev;
    L.size--;
    int nx[4] = {-1, -1, -1, -1}, pv[4] = {-1, -1, -1, -1};
    int head = L.head, head_q = -1;   // head_q >= 0: the current head is child head_q (still in registers)
    for (int q = 0; q < 4; ++q) {
        ccnt[q] = cnt[q];
        if (cnt[q] <= 0) continue;
        nx[q] = head;
        if (head_q >= 0) pv[head_q] = c[q];
        else if (head >= 0) L.nodes[head].prev = (decltype(p.prev))c[q];
        else L.tail = c[q];
        head = c[q];
        head_q = q;
        L.size++;
    }
    L.head = head;
    const int bx0[4] = {p.x0, mx, p.x0, mx}, by0[4] = {p.y0, p.y0, my, my};
    const int bx1[4] = {mx, p.x1, mx, p.x1}, by1[4] = {my, my, p.y1, p.y1};
    for (int q = 0; q < 4; ++q) {
        Node n;
        n.x0 = (int16_t)bx0[q]; n.y0 = (int16_t)by0[q]; n.x1 = (int16_t)bx1[q]; n.y1 = (int16_t)by1[q];
        n.beg = (decltype(n.beg))(p.beg + off[q]);
        n.cnt = (decltype(n.cnt))cnt[q];
        n.prev = (decltype(n.prev))pv[q];
        n.next = (decltype(n.next))nx[q];
        L.nodes[c[q]] = n;
    }
    return true;
}

// ascending (size, node) -- heap sort, in place, no recursion (device friendly)
AOS2_OCT_HD bool pair_less(const int32_t *a, int i, int j)
{
    if (a[2 * i] != a[2 * j]) return a[2 * i] < a[2 * j];
    return a[2 * i + 1] < a[2 * j + 1];
}
AOS2_OCT_HD void pair_swap(int32_t *a, int i, int j)
{
    int32_t t0 = a[2 * i], t1 = a[2 * i + 1];
    a[2 * i] = a[2 * j]; a[2 * i + 1] = a[2 * j + 1];
    a[2 * j] = t0; a[2 * j + 1] = t1;
}
AOS2_OCT_HD void sift_down(int32_t *a, int start, int end)
{
    int root = start;
    while (2 * root + 1 <= end) {
        int child = 2 * root + 1, sw = root;
        if (pair_less(a, sw, child)) sw = child;
        if (child + 1 <= end && pair_less(a, sw, child + 1)) sw = child + 1;
        if (sw == root) return;
        pair_swap(a, root, sw);
        root = sw;
    }
}
AOS2_OCT_HD void pair_sort(int32_t *a, int n)
{
    for (int start = (n - 2) / 2; start >= 0; --start) sift_down(a, start, n - 1);
    for (int end = n - 1; end > 0; --end) {
        pair_swap(a, 0, end);
        sift_down(a, 0, end - 1);
    }
}

// same order as pair_sort, out of place, one element per lane and round (keys are unique: the
// node index is part of the key)
template <class Coop>
AOS2_OCT_HD void pair_sort_to(const int32_t *src, int n, int32_t *dst)
{
    if (!Coop::kWave) {
        for (int i = 0; i < 2 * n; ++i) dst[i] = src[i];
        pair_sort(dst, n);
    } else {
        for (int i = coop_lane(); i < n; i += 64) {
            const int32_t a0 = src[2 * i], a1 = src[2 * i + 1];
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const int32_t b0 = src[2 * j], b1 = src[2 * j + 1];
                rank += (b0 < a0 || (b0 == a0 && b1 < a1)) ? 1 : 0;
            }
            dst[2 * rank] = a0;
            dst[2 * rank + 1] = a1;
        }
        coop_sync();
    }
}

}  // namespace octdetail

// xs, ys, score: n candidates in the reference's emission order.  [minX,maxX) x [minY,maxY) is
// the level's search box (16 .. w-16).  Writes the kept candidate indices in list order to
// out_idx (capacity cap) and returns their number; <0 if scratch is exhausted / cap too small.
template <class Coop = SerialCoop, class Tr = OctWide>
AOS2_OCT_HD int distribute_octree(const typename Tr::Cands &C, int n, int minX, int maxX, int minY, int maxY, int N,
                                  const OctScratchT<Tr> &S, int32_t *out_idx, int cap)
{
    using namespace octdetail;
    using Node = typename Tr::Node;
    using Idx = typename Tr::Idx;
    if (n <= 0) return 0;
#if defined(AOS2_OCT_PROF) && defined(__HIP_DEVICE_COMPILE__)
    const bool prof_on = N > 200;  // level 0 only
#endif
    OCT_T0();
    List<Node> L;
    L.nodes = S.nodes;
    L.n_alloc = 0;
    L.head = L.tail = -1;
    L.size = 0;
    L.cap = S.max_nodes;
    Idx *perm = S.perm, *tmp = S.tmp;

    // roots :541-585
    const float ratio = (float)(maxX - minX) / (float)(maxY - minY);
    // round(): half away from zero; ratio > 0 and (ratio - floor) is exact in float
    const float fl = (float)(int)ratio;
    const int nIni = (int)fl + ((ratio - fl) >= 0.5f ? 1 : 0);
    if (nIni < 1) return -1;  // reference divides by zero here (tall images): unsupported
    const float hX = (float)(maxX - minX) / (float)nIni;
    // count per root, then stable bucket the candidates
    for (int i = 0; i < nIni; ++i) {
        int id = new_node(L);
        if (id < 0) return -2;
        Node &r = L.nodes[id];
        r.x0 = (int16_t)(int)(hX * (float)i);
        r.x1 = (int16_t)(int)(hX * (float)(i + 1));
        r.y0 = 0;
        r.y1 = (int16_t)(maxY - minY);
        push_back(L, id);
    }
    if (!Coop::kWave) {
        for (int i = 0; i < n; ++i) {
            int r = (int)((float)C.x(i) / hX);
            if (r >= nIni) r = nIni - 1;  // cannot happen for x < maxX-minX; guards the arena
            L.nodes[r].cnt++;
        }
        int acc = 0;
        for (int i = 0; i < nIni; ++i) {
            L.nodes[i].beg = acc;
            acc += L.nodes[i].cnt;
            L.nodes[i].cnt = 0;
        }
        for (int i = 0; i < n; ++i) {
            int r = (int)((float)C.x(i) / hX);
            if (r >= nIni) r = nIni - 1;
            perm[L.nodes[r].beg + L.nodes[r].cnt++] = (Idx)i;
        }
    } else {
        // stable bucketing by root, one ballot per root and 64-key chunk
        const int lane = coop_lane();
        const unsigned long long lt = (1ull << lane) - 1ull;
        int acc = 0;
        for (int r = 0; r < nIni; ++r) {
            int c = 0;
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                int rr = -1;
                if (i < n) {
                    rr = (int)((float)C.x(i) / hX);
                    if (rr >= nIni) rr = nIni - 1;
                }
                const unsigned long long bal = coop_ballot(rr == r);
                if (rr == r) perm[acc + c + coop_popc(bal & lt)] = (Idx)i;
                c += coop_popc(bal);
            }
            L.nodes[r].beg = (decltype(L.nodes[r].beg))acc;
            L.nodes[r].cnt = (decltype(L.nodes[r].cnt))c;
            acc += c;
        }
        coop_sync();
    }
    for (int lit = L.head; lit >= 0;) {
        Node &nd = L.nodes[lit];
        if (nd.cnt == 0)
            lit = erase(L, lit);
        else
            lit = nd.next;
    }

    OCT_TICK(0);  // roots + bucketing
    bool finish = false;
    int32_t *cur = S.pairs_a, *prv = S.pairs_b;
    int ncur = 0;
    while (!finish) {
        const int prevSize = L.size;
        int nToExpand = 0;
        ncur = 0;
        for (int lit = L.head; lit >= 0;) {
            if (L.nodes[lit].cnt == 1) {  // bNoMore
                lit = L.nodes[lit].next;
                continue;
            }
            int c[4], ccnt[4], nxt;
            if (!divide<Coop, Tr>(L, lit, c, ccnt, nxt, C, perm, tmp)) return -2;
            if (ncur + 4 > S.max_pairs) return -2;
            for (int q = 0; q < 4; ++q) {
                const int cn = ccnt[q];
                if (cn > 1) {   // (divide() has pushed the non-empty children to the front)
                    nToExpand++;
                    cur[2 * ncur] = cn;
                    cur[2 * ncur + 1] = c[q];
                    ncur++;
                }
            }
            lit = nxt;
            OCT_COUNT(8, 1);
        }
        OCT_TICK(1);  // main passes
        OCT_COUNT(9, 1);
        if (L.size >= N || L.size == prevSize) {
            finish = true;
        } else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                const int prevSize2 = L.size;
                // sort the recorded pairs into prv; cur is then free for this round's records
                const int nprev = ncur;
                ncur = 0;
                pair_sort_to<Coop>(cur, nprev, prv);
                OCT_TICK(2);  // sort
                OCT_COUNT(10, nprev);
                int j_start = nprev - 1;
                if (Coop::kWave) {
                    // Device: the divides of this round are independent except for the stop test "size >= N after a
                    // divide".  Up to 64 nodes of the sorted order are taken together, ONE NODE PER LANE: every lane
                    // counts the quadrants of its node's <= 64 keys; a prefix sum of the leaf increments (non-empty
                    // children - 1) in processing order gives the exact node at which the sequential loop would
                    // break; the lanes up to it then do the stable partition, the four child records, the list
                    // links and the pair records, all placed by prefix sums in the order the sequential loop would
                    // have produced.  A divided parent is not unlinked but marked dead (cnt = -1); the final walk
                    // skips it.
                    const int lane = coop_lane();
                    while (j_start >= 12 && L.size < N) {   // a batch costs about as much as a dozen one-at-a-time divides
                        const int K0 = j_start + 1 < 64 ? j_start + 1 : 64;
                        const bool cand = lane < K0;
                        const int id = cand ? prv[2 * (j_start - lane) + 1] : 0;
                        const Node p = L.nodes[id];
                        const int pcnt = cand ? (int)p.cnt : 0;
                        if (coop_ballot(pcnt > 64) != 0ull) {
                            // The largest node (lane 0: the order is by size) holds more keys than a lane partitions by itself:
                            // that ONE divide goes the cooperative way, then the batch looks again.  (Leaving the batch for
                            // good at the first crowded node made a level-0 job divide all its ~50 final nodes one at a time:
                            // 54 us.)
                            const int id0 = prv[2 * j_start + 1];
                            int c[4], ccnt[4], nxt;
                            if (!divide<Coop, Tr>(L, id0, c, ccnt, nxt, C, perm, tmp)) return -2;
                            if (ncur + 4 > S.max_pairs) return -2;
                            for (int q = 0; q < 4; ++q) {
                                const int cn = ccnt[q];
                                if (cn > 1) {
                                    cur[2 * ncur] = cn;
                                    cur[2 * ncur + 1] = c[q];
                                    ncur++;
                                }
                            }
                            --j_start;
                            OCT_COUNT(11, 1);
                            continue;
                        }
                        const int hx = (p.x1 - p.x0 + 1) / 2, hy = (p.y1 - p.y0 + 1) / 2;
                        const int mx = p.x0 + hx, my = p.y0 + hy;
                        int cq[4] = {0, 0, 0, 0};
                        unsigned long long qbits_lo = 0, qbits_hi = 0;   // quadrant of key i in bits 2i, 2i+1 (i < 64)
                        for (int i = 0; i < pcnt; i += 4) {               // 4 keys per step: independent LDS reads
                            int kk[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) kk[u] = i + u < pcnt ? (int)perm[p.beg + i + u] : -1;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (kk[u] < 0) continue;
                                const int q = (C.x(kk[u]) < mx ? 0 : 1) + (C.y(kk[u]) < my ? 0 : 2);
                                cq[0] += q == 0; cq[1] += q == 1; cq[2] += q == 2; cq[3] += q == 3;
                                const int ii = i + u;
                                if (ii < 32) qbits_lo |= (unsigned long long)q << (2 * ii); else qbits_hi |= (unsigned long long)q << (2 * (ii - 32));
                            }
                        }
                        const int grow = cand ? (cq[0] > 0) + (cq[1] > 0) + (cq[2] > 0) + (cq[3] > 0) - 1 : 0;
                        const int incl = coop_excl_scan(grow) + grow;
                        // first lane after whose divide the list holds >= N leaves: the sequential loop breaks there
                        const unsigned long long hit = coop_ballot(cand && L.size + incl >= N);
                        const int K = hit ? (int)coop_ffs(hit) + 1 : K0;
                        const bool act = lane < K;
                        if (L.n_alloc + 4 * K > L.cap || ncur + 4 * K > S.max_pairs) return -2;
                        const int pn = act ? pcnt : 0;
                        int w0 = p.beg, w1 = w0 + cq[0], w2 = w1 + cq[1], w3 = w2 + cq[2];
                        const int b0 = w0, b1 = w1, b2 = w2, b3 = w3;
                        for (int i = 0; i < pn; i += 4) {   // the quadrants are remembered: only the keys are re-read
                            int kk[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) kk[u] = i + u < pn ? (int)perm[p.beg + i + u] : -1;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (kk[u] < 0) continue;
                                const int ii = i + u;
                                const int q = (int)((ii < 32 ? qbits_lo >> (2 * ii) : qbits_hi >> (2 * (ii - 32))) & 3ull);
                                const int w = q == 0 ? w0 : q == 1 ? w1 : q == 2 ? w2 : w3;
                                tmp[w] = (Idx)kk[u];
                                w0 += q == 0; w1 += q == 1; w2 += q == 2; w3 += q == 3;
                            }
                        }
                        coop_sync();
                        for (int i = 0; i < pn; i += 4) {
                            int kk[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) kk[u] = i + u < pn ? (int)tmp[p.beg + i + u] : 0;
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (i + u < pn) perm[p.beg + i + u] = (Idx)kk[u];
                        }
                        // children: arena slots in creation order (4 per divide, n1..n4)
                        const int base = L.n_alloc + 4 * lane;
                        const int cb[4] = {b0, b1, b2, b3};
                        // list: push_front of the non-empty children in the order n1..n4.  first = first pushed
                        // (deepest of the group), last = last pushed (front of the group)
                        int first = -1, last = -1, npush = 0;
                        for (int q = 0; q < 4; ++q)
                            if (act && cq[q] > 0) {
                                if (first < 0) first = base + q;
                                last = base + q;
                                npush++;
                            }
                        const int below = coop_shfl(last, lane > 0 ? lane - 1 : 0);    // front of the previous lane's group
                        const int above = coop_shfl(first, lane + 1 < 64 ? lane + 1 : 63);  // bottom of the next lane's group
                        const int old_head = L.head;
                        if (act) {
                            int prev_pushed = -1;  // the child pushed just before (it sits right behind in the list)
                            for (int q = 0; q < 4; ++q) {
                                Node &n = L.nodes[base + q];
                                n.x0 = (q & 1) ? (int16_t)mx : p.x0;
                                n.x1 = (q & 1) ? p.x1 : (int16_t)mx;
                                n.y0 = (q & 2) ? (int16_t)my : p.y0;
                                n.y1 = (q & 2) ? p.y1 : (int16_t)my;
                                n.beg = (decltype(n.beg))cb[q];
                                n.cnt = (decltype(n.cnt))cq[q];
                                n.prev = n.next = -1;
                                if (cq[q] > 0) {
                                    // next (towards the tail): the previously pushed child, or for the first pushed one the
                                    // front of the earlier lane's group / the old head
                                    const int nx = prev_pushed >= 0 ? prev_pushed : (lane > 0 ? below : old_head);
                                    n.next = (decltype(n.next))nx;
                                    if (prev_pushed >= 0) L.nodes[prev_pushed].prev = (decltype(n.prev))(base + q);
                                    prev_pushed = base + q;
                                }
                            }
                            // front of my group: its prev is the bottom of the next lane's group (or -1 at the new head)
                            L.nodes[last].prev = (decltype(p.prev))(lane + 1 < K ? above : -1);
                            if (lane == 0 && old_head >= 0) L.nodes[old_head].prev = (decltype(p.prev))first;
                            L.nodes[id].cnt = -1;  // dead parent
                        }
                        if (old_head < 0) L.tail = coop_shfl(first, 0);  // (cannot happen: the list is never empty here)
                        // pair records of children with more than one key, in divide order then n1..n4
                        int r = 0;
                        for (int q = 0; q < 4; ++q) r += act && cq[q] > 1;
                        const int eoff = ncur + coop_excl_scan(r);
                        if (act) {
                            int e = eoff;
                            for (int q = 0; q < 4; ++q)
                                if (cq[q] > 1) {
                                    cur[2 * e] = cq[q];
                                    cur[2 * e + 1] = base + q;
                                    e++;
                                }
                        }
                        ncur += coop_sum(r);
                        L.head = coop_shfl(last, K - 1);
                        L.size += coop_sum(act ? npush - 1 : 0);
                        L.n_alloc += 4 * K;
                        j_start -= K;
                        OCT_COUNT(11, K);
                        coop_sync();
                    }
                }
                for (int j = j_start; j >= 0; --j) {
                    if (L.size >= N) break;
                    const int id = prv[2 * j + 1];
                    int c[4], ccnt[4], nxt;
                    if (!divide<Coop, Tr>(L, id, c, ccnt, nxt, C, perm, tmp)) return -2;
                    if (ncur + 4 > S.max_pairs) return -2;
                    for (int q = 0; q < 4; ++q) {
                        const int cn = ccnt[q];
                        if (cn > 1) {
                            cur[2 * ncur] = cn;
                            cur[2 * ncur + 1] = c[q];
                            ncur++;
                        }
                    }
                    OCT_COUNT(11, 1);
                    if (L.size >= N) break;
                }
                OCT_TICK(3);  // final-phase divides
                if (L.size >= N || L.size == prevSize2) finish = true;
            }
        }
    }
    // best response per node, first wins ties :741-762
    int nout = 0;
    if (!Coop::kWave) {
        for (int lit = L.head; lit >= 0; lit = L.nodes[lit].next) {
            const Node &nd = L.nodes[lit];
            int best = perm[nd.beg];
            int maxResponse = C.score(best);
            for (int k = 1; k < nd.cnt; ++k) {
                const int idx = perm[nd.beg + k];
                if (C.score(idx) > maxResponse) {
                    best = idx;
                    maxResponse = C.score(idx);
                }
            }
            if (nout >= cap) return -3;
            out_idx[nout++] = best;
        }
    } else {
        // list order -> array, then one lane per node.  The list is ranked in parallel (Wyllie's pointer jumping:
        // every arena record carries (live nodes from here to the tail, jump pointer) and doubles its jump ~log2(n)
        // times) instead of chasing ~300 `next` pointers one LDS round trip at a time (27 -> ~3 us for level 0).
        // Records that are not in the list any more (unlinked parents, empty children) are recognised by their
        // predecessor not pointing back at them; tombstones (cnt < 0, still linked) count 0.
        int32_t *order = S.pairs_a;  // the pair buffers (a | b, contiguous: 4 * max_pairs words) are free at this point
        constexpr int kRankSlots = 16;   // arena records per lane
        const int n_arena = L.n_alloc;
        bool ranked = false;
        if (n_arena <= 64 * kRankSlots && n_arena < 65535 && n_arena + 8 <= 4 * S.max_pairs) {
            uint32_t *word = reinterpret_cast<uint32_t *>(S.pairs_a);   // [n_arena]: rank << 16 | jump (0xffff = none)
            const int lane = coop_lane();
            const int nslots = (n_arena + 63) >> 6;   // uniform: empty slots are skipped by scalar branches
            uint32_t w[kRankSlots];
            bool live[kRankSlots];
#pragma unroll
            for (int u = 0; u < kRankSlots; ++u) {
                const int id = lane + 64 * u;
                w[u] = 0xffffu;
                live[u] = false;
                if (u < nslots && id < n_arena) {
                    const Node nd = L.nodes[id];
                    const bool linked = nd.prev >= 0 ? (int)L.nodes[nd.prev].next == id : id == L.head;
                    if (linked && nd.cnt != 0) {   // in the list (live node or tombstone)
                        live[u] = nd.cnt > 0;
                        w[u] = ((uint32_t)live[u] << 16) | (uint32_t)(nd.next >= 0 ? nd.next : 0xffff);
                    }
                    word[id] = w[u];
                }
            }
            coop_sync();
            for (int span = 1; span < n_arena; span <<= 1) {   // uniform trip count
#pragma unroll
                for (int u = 0; u < kRankSlots; ++u) {
                    if (u >= nslots) continue;
                    const uint32_t j = w[u] & 0xffffu;
                    if (lane + 64 * u < n_arena && j != 0xffffu) {
                        const uint32_t t = word[j];
                        w[u] = ((w[u] & 0xffff0000u) + (t & 0xffff0000u)) | (t & 0xffffu);
                    }
                }
                coop_sync();
#pragma unroll
                for (int u = 0; u < kRankSlots; ++u)
                    if (u < nslots && lane + 64 * u < n_arena) word[lane + 64 * u] = w[u];
                coop_sync();
            }
            nout = L.head >= 0 ? (int)(word[L.head] >> 16) : 0;
            if (nout > cap) return -3;
            if (n_arena + nout <= 4 * S.max_pairs) {   // room for the order array behind the rank words
                order = S.pairs_a + n_arena;
#pragma unroll
                for (int u = 0; u < kRankSlots; ++u)
                    if (u < nslots && live[u]) order[nout - (int)(w[u] >> 16)] = lane + 64 * u;
                ranked = true;
            } else
                nout = 0;
        }
        if (!ranked) {
            for (int lit = L.head; lit >= 0;) {
                if (nout >= cap || nout >= 2 * S.max_pairs) return -3;
                // branch-free on the node's state: only the `next` load sits on the pointer-chasing chain; a dead parent
                // (cnt < 0, divided by a lane-parallel batch) is overwritten by the next live node
                const int nx = L.nodes[lit].next;
                const int live1 = L.nodes[lit].cnt >= 0;
                order[nout] = lit;
                nout += live1;
                lit = nx;
            }
        }
        OCT_TICK(5);  // list walk
        coop_sync();
        for (int j = coop_lane(); j < nout; j += 64) {
            const Node nd = L.nodes[order[j]];
            int best = perm[nd.beg];
            int maxResponse = C.score(best);
            for (int k = 1; k < nd.cnt; ++k) {
                const int idx = perm[nd.beg + k];
                if (C.score(idx) > maxResponse) {
                    best = idx;
                    maxResponse = C.score(idx);
                }
            }
            out_idx[j] = best;
        }
        coop_sync();
    }
    OCT_TICK(4);  // best response
    OCT_COUNT(12, 1);
    OCT_COUNT(13, L.n_alloc);
    OCT_FLUSH();
    return nout;
}

// split-array form used by the host paths
template <class Coop = SerialCoop>
AOS2_OCT_HD int distribute_octree(const int16_t *xs, const int16_t *ys, const uint8_t *score, int n, int minX, int maxX,
                                  int minY, int maxY, int N, const OctScratch &S, int32_t *out_idx, int cap)
{
    const OctCandsSplit C{xs, ys, score};
    return distribute_octree<Coop, OctWide>(C, n, minX, maxX, minY, maxY, N, S, out_idx, cap);
}

}  // namespace aos2
