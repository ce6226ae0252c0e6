// ORBmatcher hot path on gfx950: 256-bit Hamming (xor + popcount) brute force, SearchByBoW and the
// two SearchByProjection variants (reference src/ORBmatcher.cc:45-129, 159-288, 1328-1470,
// DescriptorDistance :1647-1663).  Integer VALU work; descriptors stay L2/LDS resident.
//
// The reference loops are sequential over the query set (a later query sees the features an
// earlier one took, :87-89, :209, :1403-1405).  Each problem therefore runs on ONE wave: the
// 64 lanes share the candidate set of the current query (Hamming distances + a two-smallest
// reduction with the reference's first-wins tie-break), while the greedy bookkeeping is
// wave-uniform.  Independent problems ((KF,F) pairs) are spread over the grid.
#include <type_traits>
#include <vector>

#include "aos2_common.h"
#include "wave_ops.h"

namespace aos2 {

constexpr int TH_HIGH = AOS2_TH_HIGH;
constexpr int TH_LOW = AOS2_TH_LOW;
constexpr int HISTO = AOS2_HISTO_LENGTH;
constexpr int GRID_COLS = AOS2_GRID_COLS;
constexpr int GRID_ROWS = AOS2_GRID_ROWS;
constexpr uint32_t KEY_NONE = 0xffffffffu;

struct Desc {
    uint32_t w[8];
};

__device__ __forceinline__ Desc load_desc(const uint8_t *p)
{
    Desc d;
    const uint4 a = *reinterpret_cast<const uint4 *>(p);
    const uint4 b = *reinterpret_cast<const uint4 *>(p + 16);
    d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w;
    d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w;
    return d;
}

__device__ __forceinline__ int hamming(const Desc &a, const Desc &b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d += __popc(a.w[i] ^ b.w[i]);
    return d;
}

// merge two (smallest, second smallest) pairs
__device__ __forceinline__ void merge2(uint32_t &k1, uint32_t &k2, uint32_t o1, uint32_t o2)
{
    const uint32_t lo = min(k1, o1), hi = max(k1, o1);
    k2 = min(hi, min(k2, o2));
    k1 = lo;
}

// Two smallest keys of the wave, result uniform.  The serial matcher loops sit on this latency, so
// no LDS crossbar (ds_bpermute) is used: four DPP butterfly steps reduce each 16-lane row at VALU
// speed (every step merges two disjoint lane sets), then the four row results are read into SGPRs.
template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, kCtrl, 0xf, 0xf, false);
}

__device__ __forceinline__ void wave_min2(uint32_t &k1, uint32_t &k2)
{
    merge2(k1, k2, dpp_mov<0xB1>(k1), dpp_mov<0xB1>(k2));    // quad_perm [1,0,3,2]
    merge2(k1, k2, dpp_mov<0x4E>(k1), dpp_mov<0x4E>(k2));    // quad_perm [2,3,0,1]
    merge2(k1, k2, dpp_mov<0x141>(k1), dpp_mov<0x141>(k2));  // row_half_mirror
    merge2(k1, k2, dpp_mov<0x140>(k1), dpp_mov<0x140>(k2));  // row_mirror
    uint32_t a1 = __builtin_amdgcn_readlane(k1, 0), a2 = __builtin_amdgcn_readlane(k2, 0);
    merge2(a1, a2, __builtin_amdgcn_readlane(k1, 16), __builtin_amdgcn_readlane(k2, 16));
    merge2(a1, a2, __builtin_amdgcn_readlane(k1, 32), __builtin_amdgcn_readlane(k2, 32));
    merge2(a1, a2, __builtin_amdgcn_readlane(k1, 48), __builtin_amdgcn_readlane(k2, 48));
    k1 = a1;
    k2 = a2;
}

// Single-wave kernels: lane 0's LDS write (taken/state flags) must be visible to the other lanes'
// later LDS reads.  LDS operations of one wave execute in order, so only the compiler has to be
// kept from reordering; a full __syncthreads() would also drain the prefetched global loads
// (vmcnt(0)) and put their latency back on the serial chain.
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// value of `v` in the (unique) lane selected by a ballot mask, uniform result
__device__ __forceinline__ uint32_t read_owner(uint32_t v, unsigned long long mask)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __ffsll((long long)mask) - 1);
}

// ---------------------------------------------------------------------------------------------
// Brute force: best and second best train descriptor per query.  Block = 256 threads = 256
// queries; the train set is streamed through LDS in 256-descriptor tiles (8 KB), every lane
// reads the same train word (LDS broadcast).  grid.y splits the train set; partial results are
// merged by a second tiny kernel.  key = dist << 20 | train index  (first index wins ties).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hamming_best2_kernel(const uint8_t *__restrict__ q, int nq,
                                                            const uint8_t *__restrict__ t, int nt,
                                                            int t_per_split, uint32_t *__restrict__ part1,
                                                            uint32_t *__restrict__ part2)
{
    __shared__ __attribute__((aligned(16))) uint32_t tile[256 * 8];
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int t0 = blockIdx.y * t_per_split, t1 = min(nt, t0 + t_per_split);
    Desc dq{};
    if (qi < nq) dq = load_desc(q + (size_t)qi * 32);
    uint32_t k1 = KEY_NONE, k2 = KEY_NONE;
    for (int base = t0; base < t1; base += 256) {
        const int cnt = min(256, t1 - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 2; i += 256)
            reinterpret_cast<uint4 *>(tile)[i] = reinterpret_cast<const uint4 *>(t + (size_t)base * 32)[i];
        __syncthreads();
        for (int j = 0; j < cnt; ++j) {
            const uint4 a = reinterpret_cast<const uint4 *>(tile)[2 * j];
            const uint4 b = reinterpret_cast<const uint4 *>(tile)[2 * j + 1];
            int d = __popc(dq.w[0] ^ a.x) + __popc(dq.w[1] ^ a.y) + __popc(dq.w[2] ^ a.z) + __popc(dq.w[3] ^ a.w) +
                    __popc(dq.w[4] ^ b.x) + __popc(dq.w[5] ^ b.y) + __popc(dq.w[6] ^ b.z) + __popc(dq.w[7] ^ b.w);
            const uint32_t key = ((uint32_t)d << 20) | (uint32_t)(base + j);
            if (key < k1) {
                k2 = k1;
                k1 = key;
            } else if (key < k2)
                k2 = key;
        }
    }
    if (qi < nq) {
        part1[(size_t)blockIdx.y * nq + qi] = k1;
        part2[(size_t)blockIdx.y * nq + qi] = k2;
    }
}

__global__ void hamming_merge_kernel(const uint32_t *__restrict__ part1, const uint32_t *__restrict__ part2,
                                     int nq, int splits, int32_t *__restrict__ best_idx,
                                     int32_t *__restrict__ best_dist, int32_t *__restrict__ second_dist)
{
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    uint32_t k1 = KEY_NONE, k2 = KEY_NONE;
    for (int s = 0; s < splits; ++s) merge2(k1, k2, part1[(size_t)s * nq + qi], part2[(size_t)s * nq + qi]);
    best_idx[qi] = k1 == KEY_NONE ? -1 : (int32_t)(k1 & 0xfffff);
    best_dist[qi] = k1 == KEY_NONE ? 256 : (int32_t)(k1 >> 20);
    second_dist[qi] = k2 == KEY_NONE ? 256 : (int32_t)(k2 >> 20);
}

// ---------------------------------------------------------------------------------------------
// rotation histogram helpers (ComputeThreeMaxima :1601-1642, bin quirk factor = 1/30 kept, :172)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int rot_bin(float rot)
{
    const float factor = 1.0f / HISTO;
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == HISTO) bin = 0;
    return bin;
}

__device__ __forceinline__ void three_maxima(const int *histo, int &ind1, int &ind2, int &ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < HISTO; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
        ind2 = -1;
        ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
        ind3 = -1;
    }
}

// ---------------------------------------------------------------------------------------------
// Two-stage structure of the three search methods (SURVEY.md App. E option (a)):
//   stage A (parallel, one wave per query): enumerate the query's candidates in the reference's
//           visiting order and compute the Hamming distances -> entries {key, payload},
//           key = dist << 20 | visiting position (KEY_NONE for candidates the static gates reject);
//   stage B (sequential, one wave per problem): the reference's greedy loop over the queries;
//           per query one coalesced read of its entries (prefetched one query ahead), the
//           "already taken" mask applied from LDS, a two-smallest wave reduction, the accept
//           test, and the state update.  (best, second) of the strict '<' loops = the two
//           smallest keys, so ties resolve exactly like the reference.
// ---------------------------------------------------------------------------------------------
struct Entry {
    uint32_t key;      // dist << 20 | position, KEY_NONE = rejected before the distance test
    uint32_t payload;  // feature index | octave << 24
};


// ---------------------------------------------------------------------------------------------
// Stage B without the serial chain (the greedy searches).  The reference's loops are greedy assignments: query q
// (a map point / keyframe feature, visited in a fixed order) sees feature f as taken iff an EARLIER query that
// "blocks" chose f, or f was taken before the call.  Let B[f] = the smallest such q (-1: before the call, INT_MAX:
// never).  Then every query can decide on its own ("f is blocked for q iff B[f] < q") and B follows from the
// decisions: a fixed point that is unique and equal to the sequential result (induction over q: the decision of q
// only reads B restricted to queries < q).  Iterating "all decisions from the previous B, in parallel -> next B"
// makes at least one more leading query final per pass and in practice converges in a handful of passes, because
// conflicts between search windows are rare (2-3 passes on the bench problems: 1500-3000 queries, 1000-2000 features).  One workgroup per problem, B double-buffered in LDS, the decisions
// (choice[q] = feature or -1) in global scratch; the caller turns the final decisions into the method's outputs.
// ---------------------------------------------------------------------------------------------
constexpr int kFixQpt = 3;   // queries per thread whose candidates stay in registers over the passes (VGPR budget of a 1024-thread workgroup)

struct Best2 {
    uint32_t k1, k2, p1, p2;   // smallest / second smallest key among the free candidates and their payloads
};

// (key, payload) pairs ordered by key (keys are unique: they carry the visiting position): branch-free insert of a
// candidate into the running (smallest, second smallest) pair
__device__ __forceinline__ void best2_insert(unsigned long long &b1, unsigned long long &b2, unsigned long long kk)
{
    const unsigned long long lo = kk < b1 ? kk : b1, hi = kk < b1 ? b1 : kk;
    b1 = lo;
    b2 = hi < b2 ? hi : b2;
}

__device__ __forceinline__ Best2 best2_unpack(unsigned long long b1, unsigned long long b2)
{
    return Best2{(uint32_t)(b1 >> 32), (uint32_t)(b2 >> 32), (uint32_t)b1, (uint32_t)b2};
}

// 8 entries per memory round trip (a window / bucket rarely holds more); all B reads are issued together
__device__ __forceinline__ void scan_best2_free(const Entry *__restrict__ ent, int cnt, const int32_t *B, int q,
                                                uint32_t idx_mask, unsigned long long &b1, unsigned long long &b2)
{
    for (int j0 = 0; j0 < cnt; j0 += 8) {
        Entry eb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) eb[u] = j0 + u < cnt ? ent[j0 + u] : Entry{KEY_NONE, 0};
        int bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bv[u] = eb[u].key != KEY_NONE ? B[eb[u].payload & idx_mask] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool free_ = eb[u].key != KEY_NONE && bv[u] >= q;
            best2_insert(b1, b2, free_ ? ((unsigned long long)eb[u].key << 32) | eb[u].payload : ~0ull);
        }
    }
}

// initB(f) -> -1 (taken before the call) or INT_MAX; ent_of(q, cnt) -> the query's entries; accept(q, Best2) -> chosen
// feature or -1; blocks(q) -> whether a choice of q hides the feature from later queries.  All NT threads call it.
// CACHED (n_q <= kFixQpt * NT): up to kFixQpt queries per thread keep their (up to 8) valid entries, their blocking flag and their current
// decision in registers, so a pass touches only LDS (B) -- no global round trips on the pass loop.
template <int NT, bool CACHED, class InitB, class EntOf, class Accept, class Blocks>
__device__ __forceinline__ void resolve_fixpoint_impl(int n_f, int n_q, int32_t *lds, int32_t *choice, uint32_t idx_mask,
                                                      InitB initB, EntOf ent_of, Accept accept, Blocks blocks)
{
    __shared__ int changed;
    // valid candidates beyond the 8 a query keeps in registers (wide windows: SearchByProjection(Current, Last) with th = 15
    // has them for ~10 % of the queries): an LDS arena, so that no query walks its entries in global memory on the pass loop
    // (one such query made every pass ~20 us long).  A query that finds the arena full falls back to that walk.
    constexpr int kOvfCap = 2048;
    __shared__ Entry ovf_arena[kOvfCap];
    __shared__ int ovf_used;
    const int tid = threadIdx.x;
    int32_t *Bc = lds, *Bn = lds + n_f;
    for (int i = tid; i < n_f; i += NT) Bc[i] = initB(i);
    if (tid == 0) ovf_used = 0;
    constexpr int QPT = CACHED ? kFixQpt : 1;
    Entry ce[QPT][8];
    const Entry *cent[QPT];
    int ccnt[QPT], cchoice[QPT], covo[QPT], covn[QPT];
    bool cblk[QPT], covf[QPT];
    if (CACHED) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int q = tid + NT * j;
            ccnt[j] = 0;
            cent[j] = nullptr;
            cchoice[j] = -2;
            cblk[j] = false;
            covf[j] = false;
            covo[j] = -1;
            covn[j] = 0;
            if (q < n_q) {
                cent[j] = ent_of(q, ccnt[j]);
                cblk[j] = blocks(q);
            }
            // keep the VALID entries (a window's population is mostly features rejected by the level / radius tests
            // before the distance, key == KEY_NONE): a shift register, so the array is only indexed statically.
#pragma unroll
            for (int u = 0; u < 8; ++u) ce[j][u] = Entry{KEY_NONE, 0};
            int nvalid = 0;
            for (int j0 = 0; j0 < ccnt[j]; j0 += 8) {
                Entry eb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) eb[u] = j0 + u < ccnt[j] ? cent[j][j0 + u] : Entry{KEY_NONE, 0};
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (eb[u].key != KEY_NONE) {
#pragma unroll
                        for (int w = 7; w > 0; --w) ce[j][w] = ce[j][w - 1];
                        ce[j][0] = eb[u];
                        ++nvalid;
                    }
            }
            covf[j] = nvalid > 8;
            if (nvalid > 8) {   // the registers hold the LAST 8 valid entries; the first nvalid - 8 go to the arena
                const int extra = nvalid - 8;
                const int off = atomicAdd(&ovf_used, extra);
                if (off + extra <= kOvfCap) {
                    covo[j] = off;
                    covn[j] = extra;
                    int w = 0;
                    for (int j0 = 0; j0 < ccnt[j] && w < extra; j0 += 8) {
                        Entry eb[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) eb[u] = j0 + u < ccnt[j] ? cent[j][j0 + u] : Entry{KEY_NONE, 0};
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (eb[u].key != KEY_NONE && w < extra) ovf_arena[off + w++] = eb[u];
                    }
                }
            }
        }
    } else {
        for (int q = tid; q < n_q; q += NT) choice[q] = -2;
    }
    __syncthreads();
#ifdef AOS2_FIX_DEBUG
    __shared__ int dbg_ovf, dbg_ent;
    if (tid == 0) dbg_ovf = dbg_ent = 0;
    __syncthreads();
    if (CACHED) for (int j = 0; j < QPT; ++j) { if (covf[j]) atomicAdd(&dbg_ovf, 1); atomicAdd(&dbg_ent, ccnt[j]); }
    int dbg_pass = 0;
    const long long dbg_t0 = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
#ifdef AOS2_FIX_DEBUG
        ++dbg_pass;
#endif
        if (tid == 0) changed = 0;
        for (int i = tid; i < n_f; i += NT) Bn[i] = initB(i);
        __syncthreads();
        bool ch = false;
        if (CACHED) {
#pragma unroll
            for (int j = 0; j < QPT; ++j) {
                const int q = tid + NT * j;
                int bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) bv[u] = ce[j][u].key != KEY_NONE ? Bc[ce[j][u].payload & idx_mask] : -1;
                unsigned long long b1 = ~0ull, b2 = ~0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool free_ = ce[j][u].key != KEY_NONE && bv[u] >= q;
                    best2_insert(b1, b2, free_ ? ((unsigned long long)ce[j][u].key << 32) | ce[j][u].payload : ~0ull);
                }
                if (covf[j]) {   // more than 8 valid candidates
                    if (covo[j] >= 0) {
                        for (int t0 = 0; t0 < covn[j]; t0 += 4) {   // 4 entries (and their B lookups) in flight
                            Entry e[4];
                            int bv[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) e[u] = t0 + u < covn[j] ? ovf_arena[covo[j] + t0 + u] : Entry{KEY_NONE, 0};
#pragma unroll
                            for (int u = 0; u < 4; ++u) bv[u] = e[u].key != KEY_NONE ? Bc[e[u].payload & idx_mask] : -1;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const bool free_ = e[u].key != KEY_NONE && bv[u] >= q;
                                best2_insert(b1, b2, free_ ? ((unsigned long long)e[u].key << 32) | e[u].payload : ~0ull);
                            }
                        }
                    } else {
                        b1 = b2 = ~0ull;
                        scan_best2_free(cent[j], ccnt[j], Bc, q, idx_mask, b1, b2);
                    }
                }
                const int c = ccnt[j] > 0 ? accept(q, best2_unpack(b1, b2)) : -1;   // (q >= n_q: ccnt = 0)
                ch |= c != cchoice[j];
                cchoice[j] = c;
                if (c >= 0 && cblk[j]) atomicMin(&Bn[c], q);
            }
        } else {
            for (int q = tid; q < n_q; q += NT) {
                int cnt = 0;
                const Entry *ent = ent_of(q, cnt);
                unsigned long long b1 = ~0ull, b2 = ~0ull;
                scan_best2_free(ent, cnt, Bc, q, idx_mask, b1, b2);
                const int c = cnt > 0 ? accept(q, best2_unpack(b1, b2)) : -1;
                if (c != choice[q]) {
                    ch = true;
                    choice[q] = c;
                }
                if (c >= 0 && blocks(q)) atomicMin(&Bn[c], q);
            }
        }
        if (ch) changed = 1;
        __syncthreads();
        if (!changed) break;   // the decisions reproduced themselves: B is the fixed point
        int32_t *t = Bc; Bc = Bn; Bn = t;
        __syncthreads();
    }
#ifdef AOS2_FIX_DEBUG
    if (tid == 0 && blockIdx.x == 0) printf("FIX n_f %d n_q %d cached %d passes %d overflow queries %d entries %d cycles %lld\n", n_f, n_q, (int)CACHED, dbg_pass, dbg_ovf, dbg_ent, __builtin_amdgcn_s_memtime() - dbg_t0);
#endif
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
            const int q = tid + NT * j;
            if (q < n_q) choice[q] = cchoice[j];
        }
        __syncthreads();   // the callers read choice[] with a different thread -> query mapping
    }
}

template <int NT, class InitB, class EntOf, class Accept, class Blocks>
__device__ __forceinline__ void resolve_fixpoint(int n_f, int n_q, int32_t *lds, int32_t *choice, uint32_t idx_mask,
                                                 InitB initB, EntOf ent_of, Accept accept, Blocks blocks)
{
    if (n_q <= kFixQpt * NT)   // uniform
        resolve_fixpoint_impl<NT, true>(n_f, n_q, lds, choice, idx_mask, initB, ent_of, accept, blocks);
    else
        resolve_fixpoint_impl<NT, false>(n_f, n_q, lds, choice, idx_mask, initB, ent_of, accept, blocks);
}

// SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)  :159-288
struct BowQuery {
    int32_t kf_idx;   // realIdxKF
    int32_t f_beg;    // start of the frame's bucket in node_idx_f
    int32_t f_cnt;    // bucket size
    int32_t ent_off;  // offset of this query's entries
};

struct BowPairDev {
    int n_kf, n_f, n_queries;
    const uint8_t *desc_kf, *desc_f;
    const float *angle_kf, *angle_f;
    const int32_t *node_idx_f;
    const BowQuery *queries;  // in the reference's visiting order (common nodes ascending, KF features in node order)
    Entry *entries;
    int32_t *match_f;   // n_f
    uint32_t *bin_f;    // n_f scratch: bit b set = pushed into rotHist[b]
    int32_t *nmatches;  // 1
    // SearchByBoW(KF1, KF2) :522-655 (kf_kf != 0): "kf" = KF1, "f" = KF2
    int kf_kf;
    const uint8_t *f_has_mp;  // vpMapPoints2[idx2] && !isBad(); candidates without one are skipped (:572-577)
    int32_t *match_1;         // n_kf: vpMatches12 as KF2 feature indices
    int32_t *bin_1;           // n_kf scratch: rotHist bin + 1 of a matched KF1 feature
    int32_t *choice;          // n_queries scratch of the parallel stage B
};

// stage A: grid = (max queries, pairs); one wave per query
__global__ __launch_bounds__(64) void bow_distances_kernel(const BowPairDev *__restrict__ pairs)
{
    const BowPairDev P = pairs[blockIdx.y];
    if ((int)blockIdx.x >= P.n_queries) return;
    const BowQuery q = P.queries[blockIdx.x];
    const Desc dKF = load_desc(P.desc_kf + (size_t)q.kf_idx * 32);
    for (int j = threadIdx.x; j < q.f_cnt; j += 64) {
        const int realIdxF = P.node_idx_f[q.f_beg + j];
        const int dist = hamming(dKF, load_desc(P.desc_f + (size_t)realIdxF * 32));
        Entry e;
        e.key = ((uint32_t)dist << 20) | (uint32_t)j;
        e.payload = (uint32_t)realIdxF;
        P.entries[q.ent_off + j] = e;
    }
}

// stage B: one wave per pair
__global__ __launch_bounds__(64) void bow_resolve_kernel(const BowPairDev *__restrict__ pairs, float nnratio,
                                                         int check_ori)
{
    extern __shared__ uint8_t taken[];  // vpMapPointMatches[j] != NULL
    __shared__ int histo[HISTO];
    const BowPairDev P = pairs[blockIdx.x];
    const int lane = threadIdx.x;
    if (!P.kf_kf) {
        for (int i = lane; i < P.n_f; i += 64) {
            P.match_f[i] = -1;
            P.bin_f[i] = 0;
            taken[i] = 0;
        }
    } else {
        for (int i = lane; i < P.n_f; i += 64) taken[i] = P.f_has_mp[i] ? 0 : 1;  // vbMatched2 || !pMP2 || isBad
        for (int i = lane; i < P.n_kf; i += 64) {
            P.match_1[i] = -1;
            P.bin_1[i] = 0;
        }
    }
    if (lane < HISTO) histo[lane] = 0;
    __syncthreads();
    int nmatches = 0;
    // queries are read 64 at a time (one per lane) and broadcast with shuffles; the entries of
    // the next two queries are prefetched, so no global-memory latency sits on the serial chain
    BowQuery qreg{0, 0, 0, 0};
    auto query_at = [&](int qi) {
        BowQuery r;
        const int src = qi & 63;
        r.kf_idx = __shfl(qreg.kf_idx, src);
        r.f_beg = __shfl(qreg.f_beg, src);
        r.f_cnt = __shfl(qreg.f_cnt, src);
        r.ent_off = __shfl(qreg.ent_off, src);
        return r;
    };
    auto fetch = [&](const BowQuery &qq) {
        Entry r{KEY_NONE, 0};
        if (lane < qq.f_cnt) r = P.entries[qq.ent_off + lane];
        return r;
    };
    Entry e1{KEY_NONE, 0}, e2{KEY_NONE, 0};
    BowQuery q1{0, 0, 0, 0}, q2{0, 0, 0, 0};
    if (P.n_queries > 0) {
        if (lane < P.n_queries) qreg = P.queries[lane];
        q1 = query_at(0);
        e1 = fetch(q1);
        if (P.n_queries > 1) {
            q2 = query_at(1);
            e2 = fetch(q2);
        }
    }
    for (int qi = 0; qi < P.n_queries; ++qi) {
        const BowQuery q = q1;
        Entry e = e1;
        q1 = q2;
        e1 = e2;
        if (qi + 2 < P.n_queries) {
            if (((qi + 2) & 63) == 0) {  // next chunk of queries (uniform branch)
                qreg = BowQuery{0, 0, 0, 0};
                if (qi + 2 + lane < P.n_queries) qreg = P.queries[qi + 2 + lane];
            }
            q2 = query_at(qi + 2);
            e2 = fetch(q2);
        }
        uint32_t k1 = KEY_NONE, k2 = KEY_NONE, p1 = 0;
        for (int j0 = 0; j0 < q.f_cnt; j0 += 64) {
            if (j0 > 0) e = (j0 + lane < q.f_cnt) ? P.entries[q.ent_off + j0 + lane] : Entry{KEY_NONE, 0};
            if (j0 + lane < q.f_cnt && !taken[e.payload]) {  // vpMapPointMatches[realIdxF] (:209)
                if (e.key < k1) {
                    k2 = k1;
                    k1 = e.key;
                    p1 = e.payload;
                } else if (e.key < k2)
                    k2 = e.key;
            }
        }
        const uint32_t my1 = k1;
        wave_min2(k1, k2);
        const int bestDist1 = k1 == KEY_NONE ? 256 : (int)(k1 >> 20);
        const int bestDist2 = k2 == KEY_NONE ? 256 : (int)(k2 >> 20);
        // (KF, F): bestDist1 <= TH_LOW (:237); (KF, KF): bestDist1 < TH_LOW (:599)
        const bool low = P.kf_kf ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW;
        if (low && (float)bestDist1 < __fmul_rn(nnratio, (float)bestDist2)) {
            const unsigned long long own = __ballot(my1 == k1);
            const int bestIdxF = (int)read_owner(p1, own);
            if (lane == 0) {
                taken[bestIdxF] = 1;
                int bin = 0;
                if (check_ori) {
                    bin = rot_bin(__fsub_rn(P.angle_kf[q.kf_idx], P.angle_f[bestIdxF]));
                    histo[bin]++;
                }
                if (!P.kf_kf) {
                    P.match_f[bestIdxF] = q.kf_idx;
                    if (check_ori) P.bin_f[bestIdxF] |= 1u << bin;
                } else {
                    P.match_1[q.kf_idx] = bestIdxF;
                    P.bin_1[q.kf_idx] = bin + 1;
                }
            }
            nmatches++;
            lds_fence();
        }
    }
    __syncthreads();
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(histo, i1, i2, i3);
        uint32_t culled = 0;
        for (int i = 0; i < HISTO; ++i)
            if (i != i1 && i != i2 && i != i3) {
                culled |= 1u << i;
                nmatches -= histo[i];  // one decrement per pushed entry (:281, :650)
            }
        if (!P.kf_kf) {
            for (int i = lane; i < P.n_f; i += 64)
                if (P.bin_f[i] & culled) P.match_f[i] = -1;
        } else {
            for (int i = lane; i < P.n_kf; i += 64) {
                const int bn = P.bin_1[i];
                if (bn > 0 && ((culled >> (bn - 1)) & 1u)) P.match_1[i] = -1;
            }
        }
    }
    if (lane == 0) *P.nmatches = nmatches;
}

// parallel stage B of both SearchByBoW forms (resolve_fixpoint): every match hides its frame feature (:209, :577)
__global__ __launch_bounds__(512) void bow_resolve_fix_kernel(const BowPairDev *__restrict__ pairs, float nnratio, int check_ori)
{
    constexpr int NT = 512;
    extern __shared__ int32_t fix_lds[];
    __shared__ int histo[HISTO];
    __shared__ int total;
    const BowPairDev P = pairs[blockIdx.x];
    const int tid = threadIdx.x;
    if (!P.kf_kf) {
        for (int i = tid; i < P.n_f; i += NT) {
            P.match_f[i] = -1;
            P.bin_f[i] = 0;
        }
    } else {
        for (int i = tid; i < P.n_kf; i += NT) {
            P.match_1[i] = -1;
            P.bin_1[i] = 0;
        }
    }
    if (tid < HISTO) histo[tid] = 0;
    if (tid == 0) total = 0;
    resolve_fixpoint<NT>(
        P.n_f, P.n_queries, fix_lds, P.choice, 0xffffffffu,
        [&](int i) { return (P.kf_kf && !P.f_has_mp[i]) ? -1 : INT_MAX; },   // vbMatched2 || !pMP2 || isBad (:572-577)
        [&](int q, int &cnt) {
            const BowQuery bq = P.queries[q];
            cnt = bq.f_cnt;
            return (const Entry *)(P.entries + bq.ent_off);
        },
        [&](int, const Best2 &b) {
            const int bestDist1 = b.k1 == KEY_NONE ? 256 : (int)(b.k1 >> 20);
            const int bestDist2 = b.k2 == KEY_NONE ? 256 : (int)(b.k2 >> 20);
            // (KF, F): bestDist1 <= TH_LOW (:237); (KF, KF): bestDist1 < TH_LOW (:599)
            const bool low = P.kf_kf ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW;
            return (low && (float)bestDist1 < __fmul_rn(nnratio, (float)bestDist2)) ? (int)b.p1 : -1;
        },
        [&](int) { return true; });
    int cnt = 0;
    for (int q = tid; q < P.n_queries; q += NT) {
        const int c = P.choice[q];
        if (c < 0) continue;
        const int kf = P.queries[q].kf_idx;
        int bin = 0;
        if (check_ori) {
            bin = rot_bin(__fsub_rn(P.angle_kf[kf], P.angle_f[c]));
            atomicAdd(&histo[bin], 1);
        }
        if (!P.kf_kf) {   // a frame feature is chosen by at most one query
            P.match_f[c] = kf;
            if (check_ori) P.bin_f[c] = 1u << bin;
        } else {
            P.match_1[kf] = c;
            P.bin_1[kf] = bin + 1;
        }
        ++cnt;
    }
    if (cnt) atomicAdd(&total, cnt);
    __syncthreads();
    int nmatches = total;
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(histo, i1, i2, i3);
        uint32_t culled = 0;
        for (int i = 0; i < HISTO; ++i)
            if (i != i1 && i != i2 && i != i3) {
                culled |= 1u << i;
                nmatches -= histo[i];  // one decrement per pushed entry (:281, :650)
            }
        if (!P.kf_kf) {
            for (int i = tid; i < P.n_f; i += NT)
                if (P.bin_f[i] & culled) P.match_f[i] = -1;
        } else {
            for (int i = tid; i < P.n_kf; i += NT) {
                const int bn = P.bin_1[i];
                if (bn > 0 && ((culled >> (bn - 1)) & 1u)) P.match_1[i] = -1;
            }
        }
    }
    if (tid == 0) *P.nmatches = nmatches;
}

// ---------------------------------------------------------------------------------------------
// SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)  :657-823.
// vbMatched2 is declared but never set by the reference, so the KF1 features are independent: one
// wave per query, lanes over the KF2 bucket.  The sequential rule "dist <= bestDist and the geometry
// holds -> take it" (:728, :741-745) selects the smallest distance and, among equals, the LAST
// candidate visited: key = dist << 20 | (0xFFFFF - position).
// ---------------------------------------------------------------------------------------------
struct TriQuery {
    int32_t idx1, f_beg, f_cnt;
};

struct TriPairDev {
    int n1, n2, n_queries, only_stereo;
    const uint8_t *desc1, *desc2, *has_mp2;
    const float *x1, *y1, *angle1, *u_right1, *x2, *y2, *angle2, *u_right2;
    const int32_t *octave2;
    const float *scale_factors2, *level_sigma2_2;
    float F12[9], ex, ey;
    const int32_t *node_idx2;
    const TriQuery *queries;
    int32_t *match12;   // n1
    int32_t *nmatches;
};

__global__ __launch_bounds__(64) void triang_match_kernel(const TriPairDev *__restrict__ pairs)
{
    const TriPairDev &P = pairs[blockIdx.y];
    if ((int)blockIdx.x >= P.n_queries) return;
    const TriQuery q = P.queries[blockIdx.x];
    const int lane = threadIdx.x;
    const Desc d1 = load_desc(P.desc1 + (size_t)q.idx1 * 32);
    const float kx = P.x1[q.idx1], ky = P.y1[q.idx1];
    const bool bStereo1 = P.u_right1[q.idx1] >= 0;
    // epipolar line l = x1' F12 (:142-144)
    const float a = __fadd_rn(__fadd_rn(__fmul_rn(kx, P.F12[0]), __fmul_rn(ky, P.F12[3])), P.F12[6]);
    const float b = __fadd_rn(__fadd_rn(__fmul_rn(kx, P.F12[1]), __fmul_rn(ky, P.F12[4])), P.F12[7]);
    const float c = __fadd_rn(__fadd_rn(__fmul_rn(kx, P.F12[2]), __fmul_rn(ky, P.F12[5])), P.F12[8]);
    const float den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
    uint32_t key = KEY_NONE, pay = 0;
    for (int j = lane; j < q.f_cnt; j += 64) {
        const int idx2 = P.node_idx2[q.f_beg + j];
        if (P.has_mp2[idx2]) continue;                       // :723
        const bool bStereo2 = P.u_right2[idx2] >= 0;
        if (P.only_stereo && !bStereo2) continue;
        const int dist = hamming(d1, load_desc(P.desc2 + (size_t)idx2 * 32));
        if (dist > TH_LOW) continue;                         // :734 (bestDist never exceeds TH_LOW)
        const float x2 = P.x2[idx2], y2 = P.y2[idx2];
        const int oct = P.octave2[idx2];
        if (!bStereo1 && !bStereo2) {                        // too close to the epipole (:739-745)
            const float dx = __fsub_rn(P.ex, x2), dy = __fsub_rn(P.ey, y2);
            if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.0f, P.scale_factors2[oct])) continue;
        }
        // CheckDistEpipolarLine :139-157
        const float num = __fadd_rn(__fadd_rn(__fmul_rn(a, x2), __fmul_rn(b, y2)), c);
        if (den == 0.0f) continue;
        const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
        if (!((double)dsqr < __dmul_rn(3.84, (double)P.level_sigma2_2[oct]))) continue;
        const uint32_t k = ((uint32_t)dist << 20) | (0xFFFFFu - (uint32_t)j);
        if (k < key) {
            key = k;
            pay = (uint32_t)idx2;
        }
    }
    uint32_t k1 = key, k2 = KEY_NONE;
    wave_min2(k1, k2);
    if (k1 != KEY_NONE) {
        const unsigned long long own = __ballot(key == k1);
        const int best = (int)read_owner(pay, own);
        if (lane == 0) P.match12[q.idx1] = best;
    }
}

// rotation consistency (:776-808) + count, one wave per pair
__global__ __launch_bounds__(64) void triang_finish_kernel(const TriPairDev *__restrict__ pairs, int check_ori)
{
    __shared__ int histo[HISTO];
    const TriPairDev &P = pairs[blockIdx.x];
    const int lane = threadIdx.x;
    if (lane < HISTO) histo[lane] = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = lane; i < P.n1; i += 64) {
        const int m2 = P.match12[i];
        if (m2 >= 0) {
            cnt++;
            if (check_ori) atomicAdd(&histo[rot_bin(__fsub_rn(P.angle1[i], P.angle2[m2]))], 1);
        }
    }
    __syncthreads();
    cnt = wave_sum_i32(cnt);
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(histo, i1, i2, i3);
        for (int i = lane; i < P.n1; i += 64) {
            const int m2 = P.match12[i];
            if (m2 >= 0) {
                const int bin = rot_bin(__fsub_rn(P.angle1[i], P.angle2[m2]));
                if (bin != i1 && bin != i2 && bin != i3) P.match12[i] = -1;
            }
        }
        for (int i = 0; i < HISTO; ++i)
            if (i != i1 && i != i2 && i != i3) cnt -= histo[i];
    }
    if (lane == 0) *P.nmatches = cnt;
}

// ---------------------------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptors  src/MapPoint.cc:275-340, batched over map points: one wave
// per point, lane = row of the N x N distance matrix.  The row median vDists[0.5*(N-1)] is found by
// bisection on the value (distances are 0..256) instead of sorting; the first row with the smallest
// median wins (strict '<', :326-330).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void distinctive_kernel(int n_points, const int32_t *__restrict__ off,
                                                         const uint8_t *__restrict__ desc, int32_t *__restrict__ best)
{
    const int p = blockIdx.x;
    if (p >= n_points) return;
    const int lane = threadIdx.x;
    const int beg = off[p], N = off[p + 1] - off[p];
    if (N <= 0) {
        if (lane == 0) best[p] = -1;
        return;
    }
    const uint8_t *d = desc + (size_t)beg * 32;
    const int kth = (N - 1) / 2;  // (int)(0.5 * (N - 1))
    uint32_t key = KEY_NONE;
    for (int i = lane; i < N; i += 64) {
        const Desc di = load_desc(d + (size_t)i * 32);
        int lo = 0, hi = 256;
        while (lo < hi) {  // smallest v with #{j : dist(i, j) <= v} >= kth + 1
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int j = 0; j < N; ++j) c += (j == i ? 0 : hamming(di, load_desc(d + (size_t)j * 32))) <= mid;
            if (c >= kth + 1) hi = mid; else lo = mid + 1;
        }
        const uint32_t k = ((uint32_t)lo << 20) | (uint32_t)i;
        key = min(key, k);
    }
    uint32_t k1 = key, k2 = KEY_NONE;
    wave_min2(k1, k2);
    if (lane == 0) best[p] = (int32_t)(k1 & 0xFFFFFu);
}

// ---------------------------------------------------------------------------------------------
// Frame::GetFeaturesInArea (src/Frame.cc:356-409) candidate walk shared by both projection
// searches.  Lanes take grid cells of the window (ix-major, iy-minor like the reference loops);
// a wave prefix sum over the cell populations gives every candidate its position in the
// reference's visiting order, which is the tie-break of the strict '<' updates.
// ---------------------------------------------------------------------------------------------
struct FrameDev {
    int n_f, n_levels;
    const uint8_t *desc_f;
    const float *kp_x, *kp_y, *kp_angle, *u_right, *scale_factors;
    const int32_t *kp_octave;
    float min_x, min_y, max_x, max_y, grid_w_inv, grid_h_inv;
    const int32_t *grid_off, *grid_idx;
    const uint8_t *f_mp_state;
};

// The sticky "did not fit" word of a search (read by the host after the stream drained; may live in page-locked HOST memory:
// frames_impl.inc).  A plain system-scope store: any non-zero value raises the capacity error, the value (one of the window
// populations that did not fit) only feeds its message -- an integer max is not a PCIe AtomicOp, and nothing here needs one.
__device__ __forceinline__ void overflow_note(int32_t *word, int need)
{
    __hip_atomic_store(word, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct Window {
    int x0, x1, y0, y1;  // cell range, valid iff ok
    bool ok;
};

__device__ __forceinline__ Window window_cells(const FrameDev &F, float x, float y, float r)
{
    Window w;
    w.ok = false;
    w.x0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, F.min_x), r), F.grid_w_inv)));
    if (w.x0 >= GRID_COLS) return w;
    w.x1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, F.min_x), r), F.grid_w_inv)));
    if (w.x1 < 0) return w;
    w.y0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, F.min_y), r), F.grid_h_inv)));
    if (w.y0 >= GRID_ROWS) return w;
    w.y1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, F.min_y), r), F.grid_h_inv)));
    if (w.y1 < 0) return w;
    w.ok = true;
    return w;
}

// population of the window (all features of its cells, before any gate); wave-uniform result
__device__ __forceinline__ int window_population(const FrameDev &F, const Window &w, int lane)
{
    const int ny = w.y1 - w.y0 + 1, ncell = (w.x1 - w.x0 + 1) * ny;
    int total = 0;
    for (int c = lane; c < ncell; c += 64) {
        const int cell = (w.x0 + c / ny) * GRID_ROWS + w.y0 + c % ny;
        total += F.grid_off[cell + 1] - F.grid_off[cell];
    }
    return wave_sum_i32(total);
}

// the same number by ONE thread: the cells of a grid column are neighbours in the CSR, so a column of the window is one
// difference of offsets (the counting pass of the two-pass form runs a thread per query instead of a wave per query)
__device__ __forceinline__ int window_population_solo(const FrameDev &F, const Window &w)
{
    int total = 0;
    for (int ix = w.x0; ix <= w.x1; ++ix) total += F.grid_off[ix * GRID_ROWS + w.y1 + 1] - F.grid_off[ix * GRID_ROWS + w.y0];
    return total;
}

// stage A body: writes one entry per feature of the window at out[position]
__device__ __forceinline__ void window_entries(const FrameDev &F, const Window &w, const Desc &dq, float x, float y,
                                               float r, int minLevel, int maxLevel, float xr_proj, float xr_tol,
                                               int lane, Entry *out)
{
    // One candidate per lane.  The cells of a grid column are neighbours in the CSR, so the window is (x1 - x0 + 1) runs of
    // grid_idx, and position k of the window -- cells column by column, the features of a cell in their CSR order: the order
    // GetFeaturesInArea returns them in -- is element k of the concatenated runs.  (A lane per CELL walking the cell's
    // features one after the other kept 20-40 % of the lanes busy on a chain of dependent gathers as long as the fullest
    // cell.)
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const int ncol = w.x1 - w.x0 + 1;   // <= GRID_COLS = 64: one lane per column
    int cb = 0, cnt = 0;
    if (lane < ncol) {
        const int c0 = (w.x0 + lane) * GRID_ROWS;
        cb = F.grid_off[c0 + w.y0];
        cnt = F.grid_off[c0 + w.y1 + 1] - cb;
    }
    const int incl = wave_incl_scan_i32(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const int delta = cb - (incl - cnt);   // run start - first position of the column
    for (int k0 = 0; k0 < total; k0 += 64) {
        const int k = k0 + lane;
        int src = 0;
        for (int c = 0; c < ncol; ++c) {   // (uniform) the last column that starts at or before k
            const int first = __builtin_amdgcn_readlane(incl - cnt, c), d = __builtin_amdgcn_readlane(delta, c);
            if (k >= first) src = k + d;
        }
        if (k < total) {
            const int idx = F.grid_idx[src];
            // everything the gates and the distance need of the candidate is requested at once: gate by gate (octave, then position,
            // then uRight, then the descriptor) is four dependent round trips on a wave whose whole life is a handful of them
            const int oct = F.kp_octave[idx];
            const float kx = F.kp_x[idx], ky = F.kp_y[idx], ur = F.u_right[idx];
            const Desc dc = load_desc(F.desc_f + (size_t)idx * 32);
            Entry e;
            e.key = KEY_NONE;
            e.payload = (uint32_t)idx | ((uint32_t)oct << 24);
            bool pass = true;
            if (bCheckLevels) {
                if (oct < minLevel) pass = false;
                if (maxLevel >= 0 && oct > maxLevel) pass = false;
            }
            {
                const float distx = __fsub_rn(kx, x), disty = __fsub_rn(ky, y);
                if (!(fabsf(distx) < r && fabsf(disty) < r)) pass = false;
            }
            if (ur > 0) {
                const float er = fabsf(__fsub_rn(xr_proj, ur));
                if (er > xr_tol) pass = false;
            }
            if (pass) {
                const int dist = hamming(dq, dc);
                e.key = ((uint32_t)dist << 20) | (uint32_t)k;
            }
            out[k] = e;
        }
    }
}

struct QuerySlot {
    int32_t cnt;      // window population (0 = query skipped / empty window)
    int32_t ent_off;  // offset of its entries in the pool
};
// what the counting pass of a two-pass window search found out about a query with a non-empty window: the fill pass (a wave per
// query, hundreds of thousands of them per batch, each a chain of dependent loads) reads this record and its slot instead of
// repeating map point -> position -> projection -> octave -> scale factor -> window
struct QueryRec {
    float u, v, radius, ur;
    int16_t minL, maxL;
    int32_t row;      // descriptor row of the query
};
static_assert(sizeof(QueryRec) == 24, "QueryRec layout");

// best / second of one query in stage B; e = entries of the first 64 candidates (prefetched)
struct Pick {
    uint32_t k1, k2;
    int idx1, oct1, oct2;
};

__device__ __forceinline__ Pick pick_best2(const Entry *__restrict__ ent, int cnt, Entry e, const uint8_t *state,
                                           int lane, bool skip_any = false)
{
    uint32_t k1 = KEY_NONE, k2 = KEY_NONE, p1 = 0, p2 = 0;
    for (int j0 = 0; j0 < cnt; j0 += 64) {
        if (j0 > 0) e = (j0 + lane < cnt) ? ent[j0 + lane] : Entry{KEY_NONE, 0};
        const uint8_t stv = (j0 + lane < cnt && e.key != KEY_NONE) ? state[e.payload & 0xffffffu] : (uint8_t)2;
        if (j0 + lane < cnt && e.key != KEY_NONE && (skip_any ? stv == 0 : stv != 2)) {  // Observations()>0 skip / any map point
            if (e.key < k1) {
                k2 = k1; p2 = p1;
                k1 = e.key; p1 = e.payload;
            } else if (e.key < k2) {
                k2 = e.key; p2 = e.payload;
            }
        }
    }
    Pick r;
    r.k1 = k1;
    r.k2 = k2;
    wave_min2(r.k1, r.k2);
    r.idx1 = -1;
    r.oct1 = r.oct2 = -1;
    // keys are unique (unique position): find the owners' payloads
    const unsigned long long own1 = __ballot(k1 == r.k1 && r.k1 != KEY_NONE);
    if (own1) {
        const uint32_t pl = read_owner(p1, own1);
        r.idx1 = (int)(pl & 0xffffffu);
        r.oct1 = (int)(pl >> 24);
    }
    if (r.k2 != KEY_NONE) {
        const unsigned long long o2a = __ballot(k1 == r.k2), o2b = __ballot(k2 == r.k2);
        if (o2a)
            r.oct2 = (int)(read_owner(p1, o2a) >> 24);
        else if (o2b)
            r.oct2 = (int)(read_owner(p2, o2b) >> 24);
    }
    return r;
}

// stage-B query stream: the slot of query i + 2 is fetched with a wave-uniform (scalar) load and its first 64
// entries with one vector load, two queries ahead of their use, so neither latency sits on the serial chain
struct SlotStream {
    const QuerySlot *slots;
    const Entry *pool;
    int n, lane;
    QuerySlot s1, s2;
    Entry e1, e2;
    __device__ __forceinline__ QuerySlot at(int i) const
    {
        const int iu = __builtin_amdgcn_readfirstlane(i);  // tell the compiler the index is uniform
        return slots[iu];
    }
    __device__ __forceinline__ Entry fetch(const QuerySlot &q) const
    {
        Entry r{KEY_NONE, 0};
        if (lane < q.cnt) r = pool[q.ent_off + lane];
        return r;
    }
    __device__ __forceinline__ void init(const QuerySlot *sl, const Entry *pl, int n_, int lane_)
    {
        slots = sl; pool = pl; n = n_; lane = lane_;
        s1 = s2 = QuerySlot{0, 0};
        e1 = e2 = Entry{KEY_NONE, 0};
        if (n > 0) {
            s1 = at(0);
            e1 = fetch(s1);
            if (n > 1) {
                s2 = at(1);
                e2 = fetch(s2);
            }
        }
    }
    // returns query i (slot + first 64 entries) and advances the prefetch window
    __device__ __forceinline__ void next(int i, QuerySlot &s, Entry &e)
    {
        s = s1;
        e = e1;
        s1 = s2;
        e1 = e2;
        if (i + 2 < n) {
            s2 = at(i + 2);
            e2 = fetch(s2);
        }
    }
};

struct ProjMpDev {
    int n_mp;
    const uint8_t *track_in_view, *desc, *has_obs;
    const int32_t *pred_level;
    const float *view_cos, *proj_x, *proj_y, *proj_xr;
    const int32_t *desc_idx;   // optional: row of `desc` that holds query i's descriptor (a map-point table), else row i
};

__device__ __forceinline__ float mp_radius(const FrameDev &F, const ProjMpDev &P, int i, float th)
{
    float r = (double)P.view_cos[i] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos :131-137
    if (th != 1.0f) r = __fmul_rn(r, th);
    return __fmul_rn(r, F.scale_factors[P.pred_level[i]]);
}

// SearchByProjection(Frame&, const vector<MapPoint*>&, th)  :45-129, stage A: one wave per map point
// pool_base / pool_cap: this problem's region of the entry pool, cut into one fixed slice per map point (a counter
// shared by all waves serialised ~100 k same-address atomics in one L2 channel: 0.94 ms for 64 frames); pool_used is
// only an overflow flag
// phase 0: the query's entries go to its fixed slice of the pool (below).  Two-pass form (device-resident frames):
// phase 1 only counts the window populations (slots[i].cnt), a scan assigns slots[i].ent_off, phase 2 fills.
// kSolo (phase 1 only): the caller is one thread, not a wave.
template <bool kSolo = false>
__device__ __forceinline__ void proj_mp_entries_body(const FrameDev &F, const ProjMpDev &P, float th, QuerySlot *slots,
                                                     Entry *pool, int32_t *pool_used, int pool_cap, int i, int pool_base = 0,
                                                     int phase = 0, QueryRec *rec = nullptr)
{
    const int lane = kSolo ? 0 : (int)(threadIdx.x & 63);   // a wave per query (workgroups may hold several waves)
    if (phase == 2 && rec) {   // fill pass with the counting pass's record (QueryRec): slot + record, then straight to the window
        const QuerySlot sl = slots[i];
        if (sl.cnt <= 0) return;
        const QueryRec r = rec[i];
        const Window w = window_cells(F, r.u, r.v, r.radius);
        window_entries(F, w, load_desc(P.desc + (size_t)r.row * 32), r.u, r.v, r.radius, r.minL, r.maxL, r.ur, r.radius, lane,
                       pool + sl.ent_off);
        return;
    }
    QuerySlot s{0, 0};
    if (P.track_in_view[i]) {
        const float rs = mp_radius(F, P, i, th);
        const Window w = window_cells(F, P.proj_x[i], P.proj_y[i], rs);
        if (w.ok) {
            const int pop = kSolo ? window_population_solo(F, w) : window_population(F, w, lane);
            if (pop > 0 && (phase == 1 || kSolo)) {
                s.cnt = pop;
                if (rec && lane == 0) {
                    const int lvl = P.pred_level[i];
                    QueryRec r;
                    r.u = P.proj_x[i]; r.v = P.proj_y[i]; r.radius = rs; r.ur = P.proj_xr[i];
                    r.minL = (int16_t)(lvl - 1); r.maxL = (int16_t)lvl; r.row = P.desc_idx ? P.desc_idx[i] : i;
                    rec[i] = r;
                }
            } else if (pop > 0) {
                // query i owns a fixed slice of the pool (no shared counter: same-address atomics serialise in L2)
                const int stride = phase == 2 ? pop : pool_cap / max(P.n_mp, 1);
                int off = phase == 2 ? slots[i].ent_off - pool_base : i * stride;
                if (pop > stride && lane == 0) overflow_note(pool_used, pop);   // overflow flag for the host (batched form)
                if (pop <= stride && (phase != 2 || slots[i].cnt == pop)) {
                    const int lvl = P.pred_level[i];
                    off += pool_base;
                    const size_t drow = P.desc_idx ? (size_t)P.desc_idx[i] : (size_t)i;
                    window_entries(F, w, load_desc(P.desc + drow * 32), P.proj_x[i], P.proj_y[i], rs, lvl - 1, lvl,
                                   P.proj_xr[i], rs, lane, pool + off);
                    s.cnt = pop;
                    s.ent_off = off;
                } else
                    s.cnt = -1;  // pool exhausted (cannot happen: the pool holds n_mp * n_f entries)
            }
        }
    }
    if (lane == 0 && phase != 2) slots[i] = s;
}

__global__ __launch_bounds__(64) void proj_mp_entries_kernel(FrameDev F, ProjMpDev P, float th, QuerySlot *slots,
                                                             Entry *pool, int32_t *pool_used, int pool_cap, int phase = 0)
{
    proj_mp_entries_body(F, P, th, slots, pool, pool_used, pool_cap, blockIdx.x, 0, phase);
}

// (frames_impl.inc) exclusive scan of the window populations -> entry offsets; raises *overflow to the total if it exceeds `share`
__global__ void frames_scan_slots_kernel(QuerySlot *__restrict__ slots, const int32_t *__restrict__ nq_of, int nq_cap, int share,
                                         int32_t *__restrict__ overflow);

// stage B
__device__ __forceinline__ void proj_mp_resolve_body(const FrameDev &F, const ProjMpDev &P, float nnratio,
                                                     const QuerySlot *__restrict__ slots, const Entry *__restrict__ pool,
                                                     int32_t *match_f, int32_t *nmatches_out)
{
    extern __shared__ uint8_t state[];
    const int lane = threadIdx.x;
    for (int i = lane; i < F.n_f; i += 64) {
        state[i] = F.f_mp_state[i];
        match_f[i] = -1;
    }
    __syncthreads();
    int nmatches = 0;
    SlotStream Q;
    Q.init(slots, pool, P.n_mp, lane);
    for (int iMP = 0; iMP < P.n_mp; iMP++) {
        QuerySlot s;
        Entry e;
        Q.next(iMP, s, e);
        if (s.cnt <= 0) continue;  // not in view, empty window (vIndices.empty()) -- nothing to do
        const Pick b = pick_best2(pool + s.ent_off, s.cnt, e, state, lane);
        const int bestDist = b.k1 == KEY_NONE ? 256 : (int)(b.k1 >> 20);
        const int bestDist2 = b.k2 == KEY_NONE ? 256 : (int)(b.k2 >> 20);
        if (bestDist <= TH_HIGH) {
            if (b.oct1 == b.oct2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2)) continue;
            if (lane == 0) {
                match_f[b.idx1] = iMP;
                state[b.idx1] = P.has_obs[iMP] ? 2 : 1;
            }
            nmatches++;
            lds_fence();
        }
    }
    if (lane == 0) *nmatches_out = nmatches;
}

__global__ __launch_bounds__(64) void proj_mp_resolve_kernel(FrameDev F, ProjMpDev P, float nnratio,
                                                             const QuerySlot *__restrict__ slots,
                                                             const Entry *__restrict__ pool, int32_t *match_f,
                                                             int32_t *nmatches_out)
{
    proj_mp_resolve_body(F, P, nnratio, slots, pool, match_f, nmatches_out);
}

// parallel stage B of SearchByProjection(F, vpMapPoints, th) (resolve_fixpoint above): a map point WITH observations
// hides its feature from later map points (:62-64); match_f[f] = the LAST map point that chose f (:94 overwrites)
template <int NT>
__device__ __forceinline__ void proj_mp_resolve_fix_body(const FrameDev &F, const ProjMpDev &P, float nnratio,
                                                         const QuerySlot *__restrict__ slots, const Entry *__restrict__ pool,
                                                         int32_t *match_f, int32_t *nmatches_out, int32_t *choice)
{
    extern __shared__ int32_t fix_lds[];
    __shared__ int total;
    const int tid = threadIdx.x;
    for (int i = tid; i < F.n_f; i += NT) match_f[i] = -1;
    if (tid == 0) total = 0;
    resolve_fixpoint<NT>(
        F.n_f, P.n_mp, fix_lds, choice, 0xffffffu, [&](int i) { return F.f_mp_state[i] == 2 ? -1 : INT_MAX; },
        [&](int q, int &cnt) {
            const QuerySlot s = slots[q];
            cnt = s.cnt;
            return pool + s.ent_off;
        },
        [&](int, const Best2 &b) {
            const int bestDist = b.k1 == KEY_NONE ? 256 : (int)(b.k1 >> 20);
            const int bestDist2 = b.k2 == KEY_NONE ? 256 : (int)(b.k2 >> 20);
            const int oct1 = b.k1 == KEY_NONE ? -1 : (int)(b.p1 >> 24), oct2 = b.k2 == KEY_NONE ? -1 : (int)(b.p2 >> 24);
            if (bestDist <= TH_HIGH && !(oct1 == oct2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2)))
                return (int)(b.p1 & 0xffffffu);
            return -1;
        },
        [&](int q) { return P.has_obs[q] != 0; });
    int cnt = 0;
    for (int q = tid; q < P.n_mp; q += NT) {
        const int c = choice[q];
        if (c >= 0) {
            atomicMax(&match_f[c], q);
            ++cnt;
        }
    }
    if (cnt) atomicAdd(&total, cnt);
    __syncthreads();
    if (tid == 0) *nmatches_out = total;
}

__global__ __launch_bounds__(1024) void proj_mp_resolve_fix_kernel(FrameDev F, ProjMpDev P, float nnratio,
                                                                   const QuerySlot *__restrict__ slots,
                                                                   const Entry *__restrict__ pool, int32_t *match_f,
                                                                   int32_t *nmatches_out, int32_t *choice)
{
    proj_mp_resolve_fix_body<1024>(F, P, nnratio, slots, pool, match_f, nmatches_out, choice);
}

// batched form: problem = blockIdx.y (stage A) / blockIdx.x (stage B); one entry pool shared through pool_used
struct ProjMpItem {
    FrameDev F;
    ProjMpDev P;
    QuerySlot *slots;
    int32_t *match_f, *nmatches, *choice;
    int32_t *pool_used;        // this problem's entry counter
    int32_t pool_base, pool_cap;
};
__global__ __launch_bounds__(64) void proj_mp_entries_batch_kernel(const ProjMpItem *__restrict__ items, float th, Entry *pool)
{
    const ProjMpItem &it = items[blockIdx.y];
    if ((int)blockIdx.x >= it.P.n_mp) return;
    proj_mp_entries_body(it.F, it.P, th, it.slots, pool, it.pool_used, it.pool_cap, blockIdx.x, it.pool_base);
}
__global__ __launch_bounds__(64) void proj_mp_resolve_batch_kernel(const ProjMpItem *__restrict__ items, float nnratio,
                                                                   const Entry *__restrict__ pool)
{
    const ProjMpItem &it = items[blockIdx.x];
    proj_mp_resolve_body(it.F, it.P, nnratio, it.slots, pool, it.match_f, it.nmatches);
}

__global__ __launch_bounds__(256) void proj_mp_resolve_fix_batch_kernel(const ProjMpItem *__restrict__ items, float nnratio,
                                                                        const Entry *__restrict__ pool)
{
    const ProjMpItem &it = items[blockIdx.x];
    proj_mp_resolve_fix_body<256>(it.F, it.P, nnratio, it.slots, pool, it.match_f, it.nmatches, it.choice);
}

struct ProjLastDev {
    int n_last;
    const uint8_t *last_valid, *desc, *has_obs;
    const float *world_pos, *last_angle;
    const int32_t *last_octave;
    float Tcw[16], Tlw[16];
    float fx, fy, cx, cy, mb, mbf;
    // optional (device-resident frames): LastFrame.mvpMapPoints as rows of a map-point table (world_pos / desc / has_obs
    // are then the table's arrays) and LastFrame.mvbOutlier; last_valid is unused
    const int32_t *mp_idx;
    const uint8_t *outlier;
};

__device__ __forceinline__ bool proj_last_blocks(const ProjLastDev &P, int q)
{
    if (!P.mp_idx) return P.has_obs[q] != 0;
    const int row = P.mp_idx[q];
    return row >= 0 && P.has_obs[row] != 0;
}

// SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)  :1328-1470
// stage A: one wave per last-frame feature: project, window, distances
template <bool kSolo = false>
__device__ __forceinline__ void proj_last_entries_body(const FrameDev &F, const ProjLastDev &P, float th, int mono,
                                                       QuerySlot *slots, Entry *pool, int32_t *pool_used, int pool_cap,
                                                       int i, int pool_base = 0, int phase = 0, QueryRec *rec = nullptr)
{
    const int lane = kSolo ? 0 : (int)(threadIdx.x & 63);   // a wave per query (workgroups may hold several waves)
    if (phase == 2 && rec) {   // fill pass with the counting pass's record: slot + record, then straight to the window
        const QuerySlot sl = slots[i];
        if (sl.cnt <= 0) return;
        const QueryRec r = rec[i];
        const Window w = window_cells(F, r.u, r.v, r.radius);
        window_entries(F, w, load_desc(P.desc + (size_t)r.row * 32), r.u, r.v, r.radius, r.minL, r.maxL, r.ur, r.radius, lane,
                       pool + sl.ent_off);
        return;
    }
    QuerySlot s{0, 0};
    const float *T = P.Tcw, *Tl = P.Tlw;
    int row = i;
    bool valid;
    if (P.mp_idx) {
        row = P.mp_idx[i];
        valid = row >= 0 && !P.outlier[i];
    } else
        valid = P.last_valid[i] != 0;
    if (valid) {
        // twc = -Rcw^T tcw (cv::gemm general path: double accumulation) ; tlc = Rlw twc + tlw (:1339-1346)
        float twc[3], tlc2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double sacc = __dadd_rn(__dadd_rn(__dmul_rn((double)T[k], (double)T[3]), __dmul_rn((double)T[4 + k], (double)T[7])),
                                          __dmul_rn((double)T[8 + k], (double)T[11]));
            twc[k] = (float)__dmul_rn(sacc, -1.0);
        }
        tlc2 = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(Tl[8], twc[0]), __fmul_rn(Tl[9], twc[1])), __fmul_rn(Tl[10], twc[2])), Tl[11]);
        const bool bForward = tlc2 > P.mb && !mono;
        const bool bBackward = -tlc2 > P.mb && !mono;
        const float X0 = P.world_pos[3 * (size_t)row], X1 = P.world_pos[3 * (size_t)row + 1], X2 = P.world_pos[3 * (size_t)row + 2];
        float x3Dc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float sacc = __fadd_rn(__fadd_rn(__fmul_rn(T[r * 4], X0), __fmul_rn(T[r * 4 + 1], X1)), __fmul_rn(T[r * 4 + 2], X2));
            x3Dc[r] = __fadd_rn(sacc, T[r * 4 + 3]);
        }
        const float invzc = (float)__ddiv_rn(1.0, (double)x3Dc[2]);
        const float u = __fadd_rn(__fmul_rn(__fmul_rn(P.fx, x3Dc[0]), invzc), P.cx);
        const float v = __fadd_rn(__fmul_rn(__fmul_rn(P.fy, x3Dc[1]), invzc), P.cy);
        const bool inside = !(invzc < 0) && !(u < F.min_x || u > F.max_x) && !(v < F.min_y || v > F.max_y);
        if (inside) {
            const int nLastOctave = P.last_octave[i];
            const float radius = __fmul_rn(th, F.scale_factors[nLastOctave]);
            int minL, maxL;
            if (bForward) {
                minL = nLastOctave;
                maxL = -1;
            } else if (bBackward) {
                minL = 0;
                maxL = nLastOctave;
            } else {
                minL = nLastOctave - 1;
                maxL = nLastOctave + 1;
            }
            const Window w = window_cells(F, u, v, radius);
            if (w.ok) {
                const int pop = kSolo ? window_population_solo(F, w) : window_population(F, w, lane);
                if (pop > 0 && (phase == 1 || kSolo)) {
                    s.cnt = pop;
                    if (rec && lane == 0) {
                        QueryRec r;
                        r.u = u; r.v = v; r.radius = radius; r.ur = __fsub_rn(u, __fmul_rn(P.mbf, invzc));
                        r.minL = (int16_t)minL; r.maxL = (int16_t)maxL; r.row = row;
                        rec[i] = r;
                    }
                } else if (pop > 0) {
                    const int stride = phase == 2 ? pop : pool_cap / max(P.n_last, 1);
                    const int off = phase == 2 ? slots[i].ent_off : pool_base + i * stride;
                    if (pop > stride && lane == 0) overflow_note(pool_used, pop);
                    if (pop <= stride && (phase != 2 || slots[i].cnt == pop)) {
                        const float ur = __fsub_rn(u, __fmul_rn(P.mbf, invzc));
                        window_entries(F, w, load_desc(P.desc + (size_t)row * 32), u, v, radius, minL, maxL, ur, radius, lane,
                                       pool + off);
                        s.cnt = pop;
                        s.ent_off = off;
                    } else
                        s.cnt = -1;
                }
            }
        }
    }
    if (lane == 0 && phase != 2) slots[i] = s;
}

__global__ __launch_bounds__(64) void proj_last_entries_kernel(FrameDev F, ProjLastDev P, float th, int mono,
                                                               QuerySlot *slots, Entry *pool, int32_t *pool_used,
                                                               int pool_cap)
{
    proj_last_entries_body(F, P, th, mono, slots, pool, pool_used, pool_cap, blockIdx.x);
}

__global__ __launch_bounds__(64) void proj_last_resolve_kernel(FrameDev F, ProjLastDev P, int check_ori,
                                                               const QuerySlot *__restrict__ slots,
                                                               const Entry *__restrict__ pool, int32_t *match_f,
                                                               uint32_t *bin_f, int32_t *nmatches_out)
{
    extern __shared__ uint8_t state[];
    __shared__ int histo[HISTO];
    const int lane = threadIdx.x;
    for (int i = lane; i < F.n_f; i += 64) {
        state[i] = F.f_mp_state[i];
        match_f[i] = -1;
        bin_f[i] = 0;
    }
    if (lane < HISTO) histo[lane] = 0;
    __syncthreads();
    int nmatches = 0;
    SlotStream Q;
    Q.init(slots, pool, P.n_last, lane);
    for (int i = 0; i < P.n_last; i++) {
        QuerySlot s;
        Entry e;
        Q.next(i, s, e);
        if (s.cnt <= 0) continue;
        const Pick b = pick_best2(pool + s.ent_off, s.cnt, e, state, lane);
        const int bestDist = b.k1 == KEY_NONE ? 256 : (int)(b.k1 >> 20);
        if (bestDist <= TH_HIGH) {
            if (lane == 0) {
                match_f[b.idx1] = i;
                state[b.idx1] = P.has_obs[i] ? 2 : 1;
                if (check_ori) {
                    const int bin = rot_bin(__fsub_rn(P.last_angle[i], F.kp_angle[b.idx1]));
                    // a feature can be pushed several times when a zero-observation point is
                    // overwritten (:1432): it is reset if ANY of its bins is culled, and nmatches
                    // drops once per pushed entry (:1459-1460)
                    histo[bin]++;
                    bin_f[b.idx1] |= 1u << bin;
                }
            }
            nmatches++;
            lds_fence();
        }
    }
    __syncthreads();
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(histo, i1, i2, i3);
        uint32_t culled = 0;
        for (int i = 0; i < HISTO; ++i)
            if (i != i1 && i != i2 && i != i3) {
                culled |= 1u << i;
                nmatches -= histo[i];
            }
        for (int i = lane; i < F.n_f; i += 64)
            if (bin_f[i] & culled) match_f[i] = -2;  // set to NULL (:1459)
    }
    if (lane == 0) *nmatches_out = nmatches;
}

// parallel stage B of SearchByProjection(CurrentFrame, LastFrame, th, bMono) (resolve_fixpoint)
template <int NT>
__device__ __forceinline__ void proj_last_resolve_fix_body(const FrameDev &F, const ProjLastDev &P, int check_ori,
                                                           const QuerySlot *__restrict__ slots,
                                                           const Entry *__restrict__ pool, int32_t *match_f,
                                                           uint32_t *bin_f, int32_t *nmatches_out, int32_t *choice)
{
    extern __shared__ int32_t fix_lds[];
    __shared__ int histo[HISTO];
    __shared__ int total;
    const int tid = threadIdx.x;
    for (int i = tid; i < F.n_f; i += NT) {
        match_f[i] = -1;
        bin_f[i] = 0;
    }
    if (tid < HISTO) histo[tid] = 0;
    if (tid == 0) total = 0;
    resolve_fixpoint<NT>(
        F.n_f, P.n_last, fix_lds, choice, 0xffffffu, [&](int i) { return F.f_mp_state[i] == 2 ? -1 : INT_MAX; },
        [&](int q, int &cnt) {
            const QuerySlot s = slots[q];
            cnt = s.cnt;
            return pool + s.ent_off;
        },
        [&](int, const Best2 &b) { return (b.k1 != KEY_NONE && (int)(b.k1 >> 20) <= TH_HIGH) ? (int)(b.p1 & 0xffffffu) : -1; },
        [&](int q) { return proj_last_blocks(P, q); });
    int cnt = 0;
    for (int q = tid; q < P.n_last; q += NT) {
        const int c = choice[q];
        if (c >= 0) {
            atomicMax(&match_f[c], q);
            if (check_ori) {
                // a feature can be pushed several times when a zero-observation point is overwritten (:1432): it is
                // reset if ANY of its bins is culled, and nmatches drops once per pushed entry (:1459-1460)
                const int bin = rot_bin(__fsub_rn(P.last_angle[q], F.kp_angle[c]));
                atomicAdd(&histo[bin], 1);
                atomicOr(&bin_f[c], 1u << bin);
            }
            ++cnt;
        }
    }
    if (cnt) atomicAdd(&total, cnt);
    __syncthreads();
    int nmatches = total;
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(histo, i1, i2, i3);
        uint32_t culled = 0;
        for (int i = 0; i < HISTO; ++i)
            if (i != i1 && i != i2 && i != i3) {
                culled |= 1u << i;
                nmatches -= histo[i];
            }
        for (int i = tid; i < F.n_f; i += NT)
            if (bin_f[i] & culled) match_f[i] = -2;  // set to NULL (:1459)
    }
    if (tid == 0) *nmatches_out = nmatches;
}

__global__ __launch_bounds__(1024) void proj_last_resolve_fix_kernel(FrameDev F, ProjLastDev P, int check_ori,
                                                                     const QuerySlot *__restrict__ slots,
                                                                     const Entry *__restrict__ pool, int32_t *match_f,
                                                                     uint32_t *bin_f, int32_t *nmatches_out, int32_t *choice)
{
    proj_last_resolve_fix_body<1024>(F, P, check_ori, slots, pool, match_f, bin_f, nmatches_out, choice);
}

// ---------------------------------------------------------------------------------------------
// Projection family: map points projected into a keyframe / frame and matched to
// the most similar feature of a window.
//   mode 0  Fuse(pKF, vpMapPoints, th)                         :825-975    independent points
//   mode 1  Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)        :977-1100   independent points
//   mode 2  SearchByProjection(pKF, Scw, vpPoints, vpMatched)   :290-403    greedy over vpMatched
//   mode 3  one direction of SearchBySim3                       :1148-1303  independent points
//   mode 4  SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) :1472-1599  greedy + rotation
// cv::Mat arithmetic as the reference performs it (DESIGN.md conventions): R*p+t in float left to right,
// cv::norm and Mat::dot with double accumulation, log() = correctly rounded float logarithm.
// ---------------------------------------------------------------------------------------------
struct ProjGenDev {
    int n_pts, mode;
    const uint8_t *valid, *desc;
    const float *pos, *max_dist, *min_dist, *normal, *q_angle, *inv_level_sigma2;
    float R[9], t[3], Ow[3], R2[9], t2[3];
    float fx, fy, cx, cy, bf, log_scale_factor, th;
};

__device__ __forceinline__ void xform3(const float *R, const float *t, const float *p, float *o)
{
#pragma unroll
    for (int r = 0; r < 3; ++r)
        o[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(R[3 * r], p[0]), __fmul_rn(R[3 * r + 1], p[1])), __fmul_rn(R[3 * r + 2], p[2])), t[r]);
}
__device__ __forceinline__ float norm3d(const float *v)
{
    const double s = __dadd_rn(__dadd_rn(__dmul_rn((double)v[0], (double)v[0]), __dmul_rn((double)v[1], (double)v[1])),
                               __dmul_rn((double)v[2], (double)v[2]));
    return (float)sqrt(s);
}

struct ProjSetup {
    bool ok;
    float u, v, ur, radius;
    int level;
};

// everything before GetFeaturesInArea; wave-uniform
__device__ __forceinline__ ProjSetup proj_gen_setup(const FrameDev &F, const ProjGenDev &P, int i)
{
    ProjSetup S;
    S.ok = false;
    S.u = S.v = S.ur = S.radius = 0.0f;
    S.level = 0;
    if (!P.valid[i]) return S;
    const float pw[3] = {P.pos[3 * i], P.pos[3 * i + 1], P.pos[3 * i + 2]};
    float pc[3];
    xform3(P.R, P.t, pw, pc);
    if (P.mode == 3) {  // p3Dc2 = sR21 * p3Dc1 + t21 (:1163)
        float c2[3];
        xform3(P.R2, P.t2, pc, c2);
        pc[0] = c2[0]; pc[1] = c2[1]; pc[2] = c2[2];
    }
    float u, v, invz;
    if (P.mode == 4) {  // :1505-1516: no depth test, u = fx*xc*invzc + cx, closed bounds
        invz = (float)__ddiv_rn(1.0, (double)pc[2]);
        u = __fadd_rn(__fmul_rn(__fmul_rn(P.fx, pc[0]), invz), P.cx);
        v = __fadd_rn(__fmul_rn(__fmul_rn(P.fy, pc[1]), invz), P.cy);
        if (u < F.min_x || u > F.max_x) return S;
        if (v < F.min_y || v > F.max_y) return S;
    } else {
        if (pc[2] < 0.0f) return S;
        // `1/z` (float division) in :852 and :329, `1.0/z` (double division, then float) in :1026 and :1170
        invz = (P.mode == 0 || P.mode == 2) ? __fdiv_rn(1.0f, pc[2]) : (float)__ddiv_rn(1.0, (double)pc[2]);
        const float x = __fmul_rn(pc[0], invz), y = __fmul_rn(pc[1], invz);
        u = __fadd_rn(__fmul_rn(P.fx, x), P.cx);
        v = __fadd_rn(__fmul_rn(P.fy, y), P.cy);
        if (!(u >= F.min_x && u < F.max_x && v >= F.min_y && v < F.max_y)) return S;  // KeyFrame::IsInImage
    }
    float dist3D;
    if (P.mode == 3) {
        dist3D = norm3d(pc);  // cv::norm(p3Dc2) :1185
    } else {
        const float PO[3] = {__fsub_rn(pw[0], P.Ow[0]), __fsub_rn(pw[1], P.Ow[1]), __fsub_rn(pw[2], P.Ow[2])};
        dist3D = norm3d(PO);
        if (P.mode != 4) {  // viewing angle below 60 deg: PO.dot(Pn) < 0.5*dist3D
            const double dot = __dadd_rn(__dadd_rn(__dmul_rn((double)PO[0], (double)P.normal[3 * i]),
                                                   __dmul_rn((double)PO[1], (double)P.normal[3 * i + 1])),
                                         __dmul_rn((double)PO[2], (double)P.normal[3 * i + 2]));
            if (dist3D < __fmul_rn(0.8f, P.min_dist[i]) || dist3D > __fmul_rn(1.2f, P.max_dist[i])) return S;
            if (dot < __dmul_rn(0.5, (double)dist3D)) return S;
        }
    }
    // range gate: Get{Min,Max}DistanceInvariance() = 0.8f * mfMinDistance / 1.2f * mfMaxDistance (src/MapPoint.cc:413-423);
    // PredictScale below takes the raw mfMaxDistance (:427-459) -- min_dist / max_dist carry the raw members
    const float maxd = P.max_dist[i];
    if (dist3D < __fmul_rn(0.8f, P.min_dist[i]) || dist3D > __fmul_rn(1.2f, maxd)) return S;
    // MapPoint::PredictScale (src/MapPoint.cc:427-459)
    const float ratio = __fdiv_rn(maxd, dist3D);
    const float lg = (float)log((double)ratio);
    int nScale = (int)ceilf(__fdiv_rn(lg, P.log_scale_factor));
    if (nScale < 0) nScale = 0;
    else if (nScale >= F.n_levels) nScale = F.n_levels - 1;
    S.level = nScale;
    S.radius = __fmul_rn(P.th, F.scale_factors[nScale]);
    S.u = u;
    S.v = v;
    S.ur = __fsub_rn(u, __fmul_rn(P.bf, invz));
    S.ok = true;
    return S;
}

// Frame::AssignFeaturesToGrid (src/Frame.cc:259-274, PosInGrid :411-424): one workgroup; cell of every feature,
// per-cell counts by LDS atomics, exclusive scan, and a stable fill (rank of a feature inside its cell = number of
// earlier features of the same cell, which is the push_back order of the reference)
__device__ __forceinline__ void assign_grid_body(int n, const float *__restrict__ kp_x, const float *__restrict__ kp_y,
                                                 float min_x, float min_y, float gwi, float ghi,
                                                 int32_t *__restrict__ grid_off, int32_t *__restrict__ grid_idx)
{
    extern __shared__ int32_t gsh[];
    constexpr int NC = GRID_COLS * GRID_ROWS;
    int32_t *cnt = gsh;                                  // NC + 1
    int16_t *cell = reinterpret_cast<int16_t *>(gsh + NC + 1);   // n
    __shared__ int32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c <= NC; c += 256) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int px = (int)roundf(__fmul_rn(__fsub_rn(kp_x[i], min_x), gwi));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(kp_y[i], min_y), ghi));
        const int c = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
        cell[i] = (int16_t)c;
        if (c >= 0) atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    // exclusive scan of the NC counts: thread t owns the kPer consecutive cells [t kPer, (t + 1) kPer)
    constexpr int kPer = NC / 256;
    static_assert(NC % 256 == 0 && NC <= 4096, "grid cells per thread; 12-bit cell numbers");
    int own[kPer], tsum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        own[k] = cnt[tid * kPer + k];
        tsum += own[k];
    }
    int x = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();   // (also: every thread has read its counts before the offsets overwrite them)
    int excl = x - tsum;
    for (int w = 0; w < wave; ++w) excl += wsum[w];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int c = tid * kPer + k;
        grid_off[c] = excl;
        cnt[c] = excl;   // from here on: the cell's fill cursor
        excl += own[k];
    }
    if (tid == 255) grid_off[NC] = excl;
    __syncthreads();
    // stable fill by ONE wave, 64 features at a time in index order: the lanes that share a cell find each other with 12
    // ballots (one per bit of the cell number), take consecutive places behind the cell's cursor in lane order, and the
    // last of them advances the cursor (LDS operations of a wave complete in order, so the next chunk sees it).  The
    // former per-feature count of earlier features in the same cell was O(n^2 / 256) LDS reads.
    if (wave == 0) {
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const int c = i < n ? cell[i] : -1;
            unsigned long long m = __ballot(c >= 0);
#pragma unroll
            for (int bit = 0; bit < 12; ++bit) {
                const bool one = (c >> bit) & 1;
                const unsigned long long bb = __ballot(one);
                m &= one ? bb : ~bb;
            }
            if (c >= 0) {
                const int below = __popcll(m & ((1ull << lane) - 1ull));
                const int base = cnt[c];
                grid_idx[base + below] = i;
                if (below == __popcll(m) - 1) cnt[c] = base + below + 1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void assign_grid_kernel(int n, const float *__restrict__ kp_x, const float *__restrict__ kp_y,
                                                          float min_x, float min_y, float gwi, float ghi,
                                                          int32_t *__restrict__ grid_off, int32_t *__restrict__ grid_idx)
{
    assign_grid_body(n, kp_x, kp_y, min_x, min_y, gwi, ghi, grid_off, grid_idx);
}

// Frame::ComputeStereoFromRGBD (src/Frame.cc:672-693)
__global__ void stereo_from_rgbd_kernel(int n, const float *__restrict__ kp_x, const float *__restrict__ kp_y,
                                        const float *__restrict__ kpun_x, const float *__restrict__ depth_img, int stride,
                                        float mbf, float *__restrict__ u_right, float *__restrict__ depth)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = depth_img[(size_t)(int)kp_y[i] * stride + (int)kp_x[i]];   // imDepth.at<float>(v, u): indices truncate
    float ur = -1.0f, dp = -1.0f;
    if (d > 0) {
        dp = d;
        ur = __fsub_rn(kpun_x[i], __fdiv_rn(mbf, d));
    }
    u_right[i] = ur;
    depth[i] = dp;
}

// Frame::isInFrustum (src/Frame.cc:298-354) for a batch of map points: one thread per point
__global__ void is_in_frustum_kernel(ProjGenDev P, float min_x, float max_x, float min_y, float max_y, int n_levels,
                                     float cos_limit, uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr,
                                     int32_t *pred_level, float *view_cos)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n_pts) return;
    uint8_t ok = 0;
    float u = 0, v = 0, ur = 0, vc = 0;
    int lvl = 0;
    const float pw[3] = {P.pos[3 * i], P.pos[3 * i + 1], P.pos[3 * i + 2]};
    float pc[3];
    xform3(P.R, P.t, pw, pc);
    if (!(pc[2] < 0.0f)) {
        const float invz = __fdiv_rn(1.0f, pc[2]);
        const float uu = __fadd_rn(__fmul_rn(__fmul_rn(P.fx, pc[0]), invz), P.cx);
        const float vv = __fadd_rn(__fmul_rn(__fmul_rn(P.fy, pc[1]), invz), P.cy);
        if (!(uu < min_x || uu > max_x) && !(vv < min_y || vv > max_y)) {
            const float PO[3] = {__fsub_rn(pw[0], P.Ow[0]), __fsub_rn(pw[1], P.Ow[1]), __fsub_rn(pw[2], P.Ow[2])};
            const float dist = norm3d(PO);
            if (!(dist < __fmul_rn(0.8f, P.min_dist[i]) || dist > __fmul_rn(1.2f, P.max_dist[i]))) {   // Get{Min,Max}DistanceInvariance()
                const double dot = __dadd_rn(__dadd_rn(__dmul_rn((double)PO[0], (double)P.normal[3 * i]),
                                                       __dmul_rn((double)PO[1], (double)P.normal[3 * i + 1])),
                                             __dmul_rn((double)PO[2], (double)P.normal[3 * i + 2]));
                const float viewCos = (float)__ddiv_rn(dot, (double)dist);
                if (!(viewCos < cos_limit)) {
                    const float ratio = __fdiv_rn(P.max_dist[i], dist);
                    const float lg = (float)log((double)ratio);
                    int nScale = (int)ceilf(__fdiv_rn(lg, P.log_scale_factor));
                    if (nScale < 0) nScale = 0;
                    else if (nScale >= n_levels) nScale = n_levels - 1;
                    ok = 1;
                    u = uu;
                    v = vv;
                    ur = __fsub_rn(uu, __fmul_rn(P.bf, invz));
                    lvl = nScale;
                    vc = viewCos;
                }
            }
        }
    }
    in_view[i] = ok;
    proj_x[i] = u;
    proj_y[i] = v;
    proj_xr[i] = ur;
    pred_level[i] = lvl;
    view_cos[i] = vc;
}

// best candidate of a window for the independent modes: min over (dist << 20 | visiting position)
__device__ __forceinline__ uint32_t window_best(const FrameDev &F, const ProjGenDev &P, const Window &w, const Desc &dq,
                                                const ProjSetup &S, int lane, uint32_t &payload)
{
    // One candidate per lane, like window_entries: a lane per grid COLUMN finds the column's run of grid_idx, position k of the
    // window (cells column by column, the features of a cell in their CSR order) is element k of the concatenated runs.  (A lane
    // per cell walking its features one after the other is a chain of dependent gathers as long as the fullest cell -- the
    // form this routine had until round 3.)
    const int ncol = w.x1 - w.x0 + 1;   // <= GRID_COLS = 64
    int cb = 0, cnt = 0;
    if (lane < ncol) {
        const int c0 = (w.x0 + lane) * GRID_ROWS;
        cb = F.grid_off[c0 + w.y0];
        cnt = F.grid_off[c0 + w.y1 + 1] - cb;
    }
    const int incl = wave_incl_scan_i32(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const int delta = cb - (incl - cnt);   // run start - first position of the column
    uint32_t key = KEY_NONE;
    payload = 0;
    for (int k0 = 0; k0 < total; k0 += 64) {
        const int k = k0 + lane;
        int src = 0;
        for (int c = 0; c < ncol; ++c) {   // (uniform) the last column that starts at or before k
            const int first = __builtin_amdgcn_readlane(incl - cnt, c), d = __builtin_amdgcn_readlane(delta, c);
            if (k >= first) src = k + d;
        }
        if (k < total) {
            const int idx = F.grid_idx[src];
            // (the candidate's position, octave, uRight and descriptor are requested together: gate by gate they are dependent round trips)
            const float kx = F.kp_x[idx], ky = F.kp_y[idx];
            const int oct = F.kp_octave[idx];
            const float kr = P.mode == 0 ? F.u_right[idx] : 0.0f;
            const Desc dc = load_desc(F.desc_f + (size_t)idx * 32);
            bool pass = fabsf(__fsub_rn(kx, S.u)) < S.radius && fabsf(__fsub_rn(ky, S.v)) < S.radius;
            if (oct < S.level - 1 || oct > S.level) pass = false;
            if (pass && P.mode == 0) {  // reprojection error gates of Fuse (:911-935)
                const float ex = __fsub_rn(S.u, kx), ey = __fsub_rn(S.v, ky);
                float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                double lim = 5.99;
                if (kr >= 0) {
                    const float er = __fsub_rn(S.ur, kr);
                    e2 = __fadd_rn(e2, __fmul_rn(er, er));
                    lim = 7.8;
                }
                if ((double)__fmul_rn(e2, P.inv_level_sigma2[oct]) > lim) pass = false;
            }
            if (pass) {
                const int dist = hamming(dq, dc);
                const uint32_t kk = ((uint32_t)dist << 20) | (uint32_t)k;
                if (kk < key) {
                    key = kk;
                    payload = (uint32_t)idx;
                }
            }
        }
    }
    uint32_t k1 = key, k2 = KEY_NONE;
    wave_min2(k1, k2);
    if (k1 != KEY_NONE) payload = read_owner(payload, __ballot(key == k1));
    return k1;
}

// modes 0, 1, 3: one wave per point, no shared state
__global__ __launch_bounds__(64) void projgen_best_kernel(FrameDev F, ProjGenDev P, int thr, int32_t *best_idx,
                                                          int32_t *best_dist)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    int bi = -1, bd = 256;
    const ProjSetup S = proj_gen_setup(F, P, i);
    if (S.ok) {
        const Window w = window_cells(F, S.u, S.v, S.radius);
        if (w.ok) {
            uint32_t pay;
            const uint32_t k = window_best(F, P, w, load_desc(P.desc + (size_t)i * 32), S, lane, pay);
            if (k != KEY_NONE && (int)(k >> 20) <= thr) {
                bi = (int)pay;
                bd = (int)(k >> 20);
            }
        }
    }
    if (lane == 0) {
        best_idx[i] = bi;
        best_dist[i] = bd;
    }
}

// SearchBySim3 agreement (:1306-1323)
__global__ void sim3_agree_kernel(const int32_t *__restrict__ vn1, const int32_t *__restrict__ vn2, int n1, int n2,
                                  int32_t *match12, int32_t *nfound)
{
    const int i1 = blockIdx.x * blockDim.x + threadIdx.x;
    if (i1 >= n1) return;
    int m = -1;
    const int idx2 = vn1[i1];
    if (idx2 >= 0 && idx2 < n2 && vn2[idx2] == i1) m = idx2;
    match12[i1] = m;
    if (m >= 0) atomicAdd(nfound, 1);
}

// modes 2, 4 stage A: entries of the window into the pool
__global__ __launch_bounds__(64) void projgen_entries_kernel(FrameDev F, ProjGenDev P, QuerySlot *slots, Entry *pool,
                                                             int32_t *pool_used, int pool_cap)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    QuerySlot s{0, 0};
    const ProjSetup S = proj_gen_setup(F, P, i);
    if (S.ok) {
        const Window w = window_cells(F, S.u, S.v, S.radius);
        if (w.ok) {
            const int pop = window_population(F, w, lane);
            if (pop > 0) {
                const int stride = pool_cap / max(P.n_pts, 1);   // query i owns pool[i * stride ..): no shared counter
                const int off = i * stride;
                if (pop > stride && lane == 0) overflow_note(pool_used, pop);
                if (pop <= stride) {
                    // mode 2: levels [pred-1, pred] tested per candidate (:379-382) == the level filter of the
                    // Frame version; mode 4: GetFeaturesInArea(u, v, radius, pred-1, pred+1) (:1537).  A level
                    // window starting at 0 or below disables only the lower test, like bCheckLevels (Frame.cc:379).
                    const int lo = S.level - 1, hi = P.mode == 4 ? S.level + 1 : S.level;
                    window_entries(F, w, load_desc(P.desc + (size_t)i * 32), S.u, S.v, S.radius, lo, hi, 0.0f,
                                   __builtin_huge_valf(), lane, pool + off);
                    s.cnt = pop;
                    s.ent_off = off;
                } else
                    s.cnt = -1;
            }
        }
    }
    if (lane == 0) slots[i] = s;
}

// modes 2, 4 stage B: the reference's greedy loop over the points
__global__ __launch_bounds__(64) void projgen_resolve_kernel(FrameDev F, ProjGenDev P, int thr, int check_ori,
                                                             const QuerySlot *__restrict__ slots,
                                                             const Entry *__restrict__ pool, int32_t *match_f,
                                                             int32_t *bin_f, int32_t *nmatches_out)
{
    extern __shared__ uint8_t state[];  // != 0: vpMatched[idx] / mvpMapPoints[idx] is set
    __shared__ int histo[HISTO];
    const int lane = threadIdx.x;
    for (int i = lane; i < F.n_f; i += 64) {
        state[i] = F.f_mp_state[i] != 0;
        match_f[i] = -1;
        bin_f[i] = 0;
    }
    if (lane < HISTO) histo[lane] = 0;
    __syncthreads();
    int nmatches = 0;
    SlotStream Q;
    Q.init(slots, pool, P.n_pts, lane);
    for (int i = 0; i < P.n_pts; i++) {
        QuerySlot s;
        Entry e;
        Q.next(i, s, e);
        if (s.cnt <= 0) continue;
        const Pick b = pick_best2(pool + s.ent_off, s.cnt, e, state, lane, true);
        const int bestDist = b.k1 == KEY_NONE ? 256 : (int)(b.k1 >> 20);
        if (bestDist <= thr) {
            if (lane == 0) {
                match_f[b.idx1] = i;
                state[b.idx1] = 1;
                if (check_ori) {
                    const int bin = rot_bin(__fsub_rn(P.q_angle[i], F.kp_angle[b.idx1]));
                    histo[bin]++;
                    bin_f[b.idx1] = bin + 1;
                }
            }
            nmatches++;
            lds_fence();
        }
    }
    __syncthreads();
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(histo, i1, i2, i3);
        for (int i = 0; i < HISTO; ++i)
            if (i != i1 && i != i2 && i != i3) nmatches -= histo[i];
        for (int i = lane; i < F.n_f; i += 64) {
            const int bn = bin_f[i] - 1;
            if (bn >= 0 && bn != i1 && bn != i2 && bn != i3) match_f[i] = -2;  // set to NULL (:1589)
        }
    }
    if (lane == 0) *nmatches_out = nmatches;
}

// modes 2, 4 parallel stage B (resolve_fixpoint): every match hides its feature, so a feature has one owner
__global__ __launch_bounds__(1024) void projgen_resolve_fix_kernel(FrameDev F, ProjGenDev P, int thr, int check_ori,
                                                                   const QuerySlot *__restrict__ slots,
                                                                   const Entry *__restrict__ pool, int32_t *match_f,
                                                                   int32_t *bin_f, int32_t *nmatches_out, int32_t *choice)
{
    constexpr int NT = 1024;
    extern __shared__ int32_t fix_lds[];
    __shared__ int histo[HISTO];
    __shared__ int total;
    const int tid = threadIdx.x;
    for (int i = tid; i < F.n_f; i += NT) {
        match_f[i] = -1;
        bin_f[i] = 0;
    }
    if (tid < HISTO) histo[tid] = 0;
    if (tid == 0) total = 0;
    resolve_fixpoint<NT>(
        F.n_f, P.n_pts, fix_lds, choice, 0xffffffu, [&](int i) { return F.f_mp_state[i] != 0 ? -1 : INT_MAX; },
        [&](int q, int &cnt) {
            const QuerySlot s = slots[q];
            cnt = s.cnt;
            return pool + s.ent_off;
        },
        [&](int, const Best2 &b) { return (b.k1 != KEY_NONE && (int)(b.k1 >> 20) <= thr) ? (int)(b.p1 & 0xffffffu) : -1; },
        [&](int) { return true; });
    int cnt = 0;
    for (int q = tid; q < P.n_pts; q += NT) {
        const int c = choice[q];
        if (c < 0) continue;
        match_f[c] = q;
        if (check_ori) {
            const int bin = rot_bin(__fsub_rn(P.q_angle[q], F.kp_angle[c]));
            atomicAdd(&histo[bin], 1);
            bin_f[c] = bin + 1;
        }
        ++cnt;
    }
    if (cnt) atomicAdd(&total, cnt);
    __syncthreads();
    int nmatches = total;
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(histo, i1, i2, i3);
        for (int i = 0; i < HISTO; ++i)
            if (i != i1 && i != i2 && i != i3) nmatches -= histo[i];
        for (int i = tid; i < F.n_f; i += NT) {
            const int bn = bin_f[i] - 1;
            if (bn >= 0 && bn != i1 && bn != i2 && bn != i3) match_f[i] = -2;  // set to NULL (:1589)
        }
    }
    if (tid == 0) *nmatches_out = nmatches;
}

// ---------------------------------------------------------------------------------------------
// SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)  :405-520 (monocular bootstrap).
// Stage A: one wave per level-0 F1 feature, window of F2 around vbPrevMatched[i1] restricted to level 0.
// Stage B: the greedy loop with its two per-F2 tables in LDS: vMatchedDistance (a candidate is skipped
// while an earlier match holds it with a distance <= ours, :443) and vnMatches21 (a better later match
// steals the F2 feature and un-matches its previous owner, :462-466).
// ---------------------------------------------------------------------------------------------
struct InitDev {
    int n1;
    const uint8_t *desc1;
    const int32_t *octave1;
    const float *angle1, *prev_xy;
    float window;
};

__global__ __launch_bounds__(64) void init_entries_kernel(FrameDev F2, InitDev P, QuerySlot *slots, Entry *pool,
                                                          int32_t *pool_used, int pool_cap)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    QuerySlot s{0, 0};
    if (P.octave1[i] <= 0) {  // level1 > 0 -> continue (:421-423)
        const float x = P.prev_xy[2 * i], y = P.prev_xy[2 * i + 1];
        const Window w = window_cells(F2, x, y, P.window);
        if (w.ok) {
            const int pop = window_population(F2, w, lane);
            if (pop > 0) {
                const int stride = pool_cap / max(P.n1, 1);
                const int off = i * stride;
                if (pop > stride && lane == 0) overflow_note(pool_used, pop);
                if (pop <= stride) {
                    const int lvl = P.octave1[i];
                    window_entries(F2, w, load_desc(P.desc1 + (size_t)i * 32), x, y, P.window, lvl, lvl, 0.0f,
                                   __builtin_huge_valf(), lane, pool + off);
                    s.cnt = pop;
                    s.ent_off = off;
                } else
                    s.cnt = -1;
            }
        }
    }
    if (lane == 0) slots[i] = s;
}

__global__ __launch_bounds__(64) void init_resolve_kernel(FrameDev F2, InitDev P, float nnratio, int check_ori,
                                                          const QuerySlot *__restrict__ slots,
                                                          const Entry *__restrict__ pool, int32_t *match12,
                                                          int32_t *bin_1, int32_t *nmatches_out)
{
    extern __shared__ uint16_t init_lds[];
    uint16_t *md = init_lds;               // vMatchedDistance (0xFFFF = INT_MAX)
    uint16_t *m21 = init_lds + F2.n_f;     // vnMatches21 + 1 (0 = -1)
    __shared__ int histo[HISTO];
    const int lane = threadIdx.x;
    for (int i = lane; i < F2.n_f; i += 64) {
        md[i] = 0xFFFFu;
        m21[i] = 0;
    }
    for (int i = lane; i < P.n1; i += 64) {
        match12[i] = -1;
        bin_1[i] = 0;
    }
    if (lane < HISTO) histo[lane] = 0;
    __syncthreads();
    int nmatches = 0;
    SlotStream Q;
    Q.init(slots, pool, P.n1, lane);
    for (int i1 = 0; i1 < P.n1; i1++) {
        QuerySlot s;
        Entry e;
        Q.next(i1, s, e);
        if (s.cnt <= 0) continue;
        uint32_t k1 = KEY_NONE, k2 = KEY_NONE, p1 = 0;
        for (int j0 = 0; j0 < s.cnt; j0 += 64) {
            if (j0 > 0) e = (j0 + lane < s.cnt) ? pool[s.ent_off + j0 + lane] : Entry{KEY_NONE, 0};
            if (j0 + lane < s.cnt && e.key != KEY_NONE) {
                const uint32_t dist = e.key >> 20;
                if ((uint32_t)md[e.payload & 0xffffffu] > dist) {  // !(vMatchedDistance[i2] <= dist)
                    if (e.key < k1) {
                        k2 = k1;
                        k1 = e.key;
                        p1 = e.payload;
                    } else if (e.key < k2)
                        k2 = e.key;
                }
            }
        }
        const uint32_t my1 = k1;
        wave_min2(k1, k2);
        if (k1 == KEY_NONE) continue;
        const int bestDist = (int)(k1 >> 20);
        const float second = k2 == KEY_NONE ? 2147483648.0f : (float)(int)(k2 >> 20);  // (float)INT_MAX
        if (bestDist <= TH_LOW && (float)bestDist < __fmul_rn(second, nnratio)) {
            const int bestIdx2 = (int)(read_owner(p1, __ballot(my1 == k1)) & 0xffffffu);
            const int prev = (int)m21[bestIdx2];
            if (lane == 0) {
                if (prev > 0) match12[prev - 1] = -1;
                match12[i1] = bestIdx2;
                m21[bestIdx2] = (uint16_t)(i1 + 1);
                md[bestIdx2] = (uint16_t)bestDist;
                if (check_ori) {
                    const int bin = rot_bin(__fsub_rn(P.angle1[i1], F2.kp_angle[bestIdx2]));
                    histo[bin]++;
                    bin_1[i1] = bin + 1;
                }
            }
            nmatches += prev > 0 ? 0 : 1;  // nmatches-- for the stolen match, nmatches++ for the new one
            lds_fence();
        }
    }
    __threadfence_block();
    __syncthreads();
    if (check_ori) {
        int i1m, i2m, i3m;
        three_maxima(histo, i1m, i2m, i3m);
        int drop = 0;
        for (int i = lane; i < P.n1; i += 64) {
            const int bn = bin_1[i] - 1;
            if (bn >= 0 && bn != i1m && bn != i2m && bn != i3m && match12[i] >= 0) {  // :497-501
                match12[i] = -1;
                drop++;
            }
        }
        drop = wave_sum_i32(drop);
        nmatches -= drop;
    }
    if (lane == 0) *nmatches_out = nmatches;
}

}  // namespace aos2

using namespace aos2;

// LDS budget of the fixed-point resolve kernels (two int32 copies of B[n_f]): default dynamic-LDS limit
constexpr size_t kFixLdsBytes = 44 * 1024;   // + the 16 KB overflow arena of resolve_fixpoint_impl: within the 64 KB of a workgroup

struct aos2_matcher {
    float nnratio;
    int check_ori;
    int device;
    bool dev_ready = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {};
    DevBuf<uint8_t> arena;     // one call's inputs | scratch | results (struct Arena)
    PinnedBuf<uint8_t> h_in;   // page-locked: the inputs on their way up
    PinnedBuf<uint8_t> h_out;  // page-locked: the results on their way back
    DevBuf<uint32_t> part;     // hamming partials
    DevBuf<uint64_t> pool;     // candidate entries of the projection searches (8 B each)
    DevBuf<float> fr_angle;    // aos2_matcher_search_by_bow_frames: dense key angles of the two frame batches
    PinnedBuf<uint8_t> fr_host;   // ... and the host copies of their FeatureVectors and counts
    float last_ms = 0;         // device time of the kernels of the last search call
    bool serial_resolve = false;   // AOS2_SERIAL_RESOLVE=1: the one-wave sequential stage B (tests compare both)
};

namespace aos2 {

static int matcher_init(aos2_matcher *m)
{
    int st = bind_device(m->device);
    if (st) return st;
    if (m->dev_ready) return AOS2_OK;
    if (int st_ = stream_create(&m->stream, stream_priority_env("AOS2_PRIO_MATCHER"))) return st_;
    for (auto &e : m->ev) AOS2_HIP_CHECK(hipEventCreate(&e));
    {
        const char *v = getenv("AOS2_SERIAL_RESOLVE");
        m->serial_resolve = v && atoi(v) != 0;
    }
    m->dev_ready = true;
    return AOS2_OK;
}

// bump allocator over one device arena: uploads host arrays, returns device pointers
// One call's memory: three regions of the handle's device arena.
//   inputs  (push / push_hole): assembled in the handle's page-locked host buffer and uploaded with ONE asynchronous
//           copy; the buffer persists between calls, so after the first calls nothing is allocated;
//   scratch (reserve): device only, zeroed by a device memset (never travels);
//   results (reserve_out): device only, zeroed; fetch() + finish() bring ALL results back with one copy into a
//           page-locked bounce buffer and scatter them to the caller's arrays.
// (Earlier the whole arena, scratch included, was staged in a pageable vector and uploaded, and every result array
// was a pageable copy of its own: 6.9 ms of host time around 0.06 ms of kernels for 64 SearchByBoW pairs.)
// Offsets carry their region in the top bits; dev<T>() resolves them once every region's size is known, so all
// push / reserve calls of a function come before its first dev<T>() / upload().
struct Arena {
    aos2_matcher *m;
    static constexpr size_t kScr = (size_t)1 << 62, kOut = (size_t)1 << 61, kMask = kOut - 1;
    size_t in_size = 0, scr_size = 0, out_size = 0;
    mutable bool frozen = false;
    int err = AOS2_OK;
    struct Fetch { void *dst; const uint8_t *src; size_t bytes; };
    std::vector<Fetch> fetches;
    static size_t up(size_t x) { return (x + 255) & ~(size_t)255; }
    void late(const char *what) const
    {
        if (!frozen) return;
        fprintf(stderr, "aos2 matcher arena: %s after the layout was fixed\n", what);
        abort();
    }
    size_t push(const void *src, size_t bytes)
    {
        late("push");
        const size_t off = up(in_size);
        if (off + bytes > m->h_in.n) {   // grow the page-locked buffer (kept by the handle), contents preserved
            PinnedBuf<uint8_t> nb;
            const int st = nb.alloc(std::max((off + bytes) * 2, (size_t)1 << 20));
            if (st) {
                err = st;
                return 0;
            }
            if (in_size) memcpy(nb.p, m->h_in.p, in_size);
            m->h_in.release();
            m->h_in = nb;
            nb.p = nullptr;
            nb.n = 0;
        }
        if (off > in_size) memset(m->h_in.p + in_size, 0, off - in_size);
        if (bytes) {
            if (src) memcpy(m->h_in.p + off, src, bytes);
            else memset(m->h_in.p + off, 0, bytes);
        }
        in_size = off + bytes;
        return off;
    }
    size_t push_hole(size_t bytes) { return push(nullptr, bytes); }   // input the host fills through hostptr()
    uint8_t *hostptr(size_t off) const { return m->h_in.p + off; }
    size_t reserve(size_t bytes)
    {
        late("reserve");
        const size_t off = up(scr_size);
        scr_size = off + bytes;
        return kScr | off;
    }
    size_t reserve_out(size_t bytes)
    {
        late("reserve_out");
        const size_t off = up(out_size);
        out_size = off + bytes;
        return kOut | off;
    }
    size_t scr_base() const { return up(in_size); }
    size_t out_base() const { return scr_base() + up(scr_size); }
    size_t total() const { return out_base() + up(out_size); }
    int alloc()
    {
        frozen = true;
        if (err) return err;
        return m->arena.alloc(total() + 256);
    }
    int upload()
    {
        int st = alloc();
        if (st) return st;
        if (in_size) AOS2_HIP_CHECK(hipMemcpyAsync(m->arena.p, m->h_in.p, in_size, hipMemcpyHostToDevice, m->stream));
        if (total() > scr_base()) AOS2_HIP_CHECK(hipMemsetAsync(m->arena.p + scr_base(), 0, total() - scr_base(), m->stream));
        return AOS2_OK;
    }
    template <typename T>
    T *dev(size_t off) const
    {
        frozen = true;
        const size_t base = (off & kScr) ? scr_base() : (off & kOut) ? out_base() : 0;
        return reinterpret_cast<T *>(m->arena.p + base + (off & kMask));
    }
    // result at `off` (normally from reserve_out, so that all results are neighbours) -> dst, delivered by finish()
    void fetch(void *dst, size_t off, size_t bytes)
    {
        if (bytes) fetches.push_back(Fetch{dst, dev<uint8_t>(off), bytes});
    }
    void fetch_dev(void *dst, const void *d_src, size_t bytes)
    {
        if (bytes) fetches.push_back(Fetch{dst, static_cast<const uint8_t *>(d_src), bytes});
    }
    // one device-to-host copy of the span the results occupy (they are neighbours in the arena; if they are not, one
    // copy each) into the page-locked bounce buffer, a wait for the stream, then the scatter to the caller's arrays
    int finish()
    {
        const uint8_t *lo = nullptr, *hi = nullptr;
        size_t sum = 0;
        for (const Fetch &f : fetches) {
            if (!lo || f.src < lo) lo = f.src;
            if (!hi || f.src + f.bytes > hi) hi = f.src + f.bytes;
            sum += (f.bytes + 15) & ~(size_t)15;
        }
        const size_t span = (size_t)(hi - lo);
        const bool one = span <= 4 * sum + 65536;
        int st = m->h_out.alloc((one ? span : sum) + 64);
        if (st) return st;
        if (one) {
            if (span) AOS2_HIP_CHECK(hipMemcpyAsync(m->h_out.p, lo, span, hipMemcpyDeviceToHost, m->stream));
        } else {
            size_t o = 0;
            for (const Fetch &f : fetches) {
                AOS2_HIP_CHECK(hipMemcpyAsync(m->h_out.p + o, f.src, f.bytes, hipMemcpyDeviceToHost, m->stream));
                o += (f.bytes + 15) & ~(size_t)15;
            }
        }
        AOS2_HIP_CHECK(hipStreamSynchronize(m->stream));
        AOS2_HIP_CHECK(hipGetLastError());
        size_t o = 0;
        for (const Fetch &f : fetches) {
            memcpy(f.dst, one ? m->h_out.p + (f.src - lo) : m->h_out.p + o, f.bytes);
            o += (f.bytes + 15) & ~(size_t)15;
        }
        fetches.clear();
        return AOS2_OK;
    }
};

static void fill_frame(Arena &A, const aos2_frame_view_t *f, size_t off[12])
{
    const size_t n = (size_t)f->n_f;
    off[0] = A.push(f->desc_f, n * 32);
    off[1] = A.push(f->kp_x, n * 4);
    off[2] = A.push(f->kp_y, n * 4);
    off[3] = A.push(f->kp_angle, n * 4);
    off[4] = A.push(f->u_right, n * 4);
    off[5] = A.push(f->scale_factors, (size_t)f->n_levels * 4);
    off[6] = A.push(f->kp_octave, n * 4);
    off[7] = A.push(f->grid_off, (size_t)(GRID_COLS * GRID_ROWS + 1) * 4);
    off[8] = A.push(f->grid_idx, (size_t)f->grid_off[GRID_COLS * GRID_ROWS] * 4);
    off[9] = A.push(f->f_mp_state, n);
}

static FrameDev frame_dev(const Arena &A, const aos2_frame_view_t *f, const size_t off[12])
{
    FrameDev F{};
    F.n_f = f->n_f;
    F.n_levels = f->n_levels;
    F.desc_f = A.dev<uint8_t>(off[0]);
    F.kp_x = A.dev<float>(off[1]);
    F.kp_y = A.dev<float>(off[2]);
    F.kp_angle = A.dev<float>(off[3]);
    F.u_right = A.dev<float>(off[4]);
    F.scale_factors = A.dev<float>(off[5]);
    F.kp_octave = A.dev<int32_t>(off[6]);
    F.grid_off = A.dev<int32_t>(off[7]);
    F.grid_idx = A.dev<int32_t>(off[8]);
    F.f_mp_state = A.dev<uint8_t>(off[9]);
    F.min_x = f->min_x; F.min_y = f->min_y; F.max_x = f->max_x; F.max_y = f->max_y;
    F.grid_w_inv = f->grid_w_inv; F.grid_h_inv = f->grid_h_inv;
    return F;
}

static int check_frame(const aos2_frame_view_t *f)
{
    if (!f || f->n_f < 0 || !f->grid_off || (f->n_f > 0 && (!f->desc_f || !f->kp_x || !f->kp_y || !f->kp_octave ||
        !f->kp_angle || !f->u_right || !f->f_mp_state)) || !f->scale_factors || f->n_levels <= 0) {
        set_error("bad frame view");
        return AOS2_ERR_ARG;
    }
    if (f->n_f > 60000) {
        set_error("n_f %d exceeds the LDS state table (60000)", f->n_f);
        return AOS2_ERR_ARG;
    }
    // the arrays that become device-side indices: mGrid as CSR (monotone offsets, every listed feature exists) and the
    // pyramid levels (they index scale_factors[] / level_sigma2[])
    constexpr int NC = GRID_COLS * GRID_ROWS;
    if (f->grid_off[0] != 0) {
        set_error("bad frame view: grid_off[0] = %d", f->grid_off[0]);
        return AOS2_ERR_ARG;
    }
    for (int c = 0; c < NC; ++c)
        if (f->grid_off[c + 1] < f->grid_off[c]) {
            set_error("bad frame view: grid_off decreases at cell %d", c);
            return AOS2_ERR_ARG;
        }
    const int listed = f->grid_off[NC];
    if (listed > f->n_f || (listed > 0 && !f->grid_idx)) {
        set_error("bad frame view: the grid lists %d features of %d%s", listed, f->n_f, f->grid_idx ? "" : " (grid_idx is NULL)");
        return AOS2_ERR_ARG;
    }
    for (int i = 0; i < listed; ++i)
        if ((unsigned)f->grid_idx[i] >= (unsigned)f->n_f) {
            set_error("bad frame view: grid_idx[%d] = %d is not a feature", i, f->grid_idx[i]);
            return AOS2_ERR_ARG;
        }
    for (int i = 0; i < f->n_f; ++i)
        if ((unsigned)f->kp_octave[i] >= (unsigned)f->n_levels) {
            set_error("bad frame view: kp_octave[%d] = %d outside the %d pyramid levels", i, f->kp_octave[i], f->n_levels);
            return AOS2_ERR_ARG;
        }
    return AOS2_OK;
}

// levels handed in per query (mnTrackScaleLevel, LastFrame octaves, keyframe octaves): they index the frame's scale tables
// (only of the queries the reference reads them for: mnTrackScaleLevel of a point that is not in view, or the octave of a
// LastFrame feature without a usable map point, may be stale / uninitialised there -- `live` = that gate, NULL = all)
static int check_levels(const int32_t *lv, int n, int n_levels, const char *what, const uint8_t *live = nullptr)
{
    if (n > 0 && !lv) {
        set_error("%s is NULL", what);
        return AOS2_ERR_ARG;
    }
    for (int i = 0; i < n; ++i)
        if ((!live || live[i]) && (unsigned)lv[i] >= (unsigned)n_levels) {
            set_error("%s[%d] = %d outside the %d pyramid levels", what, i, lv[i], n_levels);
            return AOS2_ERR_ARG;
        }
    return AOS2_OK;
}

}  // namespace aos2

extern "C" {

int aos2_matcher_create(float nnratio, int check_orientation, int device, aos2_matcher_t **out)
{
    if (!out) return AOS2_ERR_ARG;
    aos2_matcher *m = new aos2_matcher();
    m->nnratio = nnratio;
    m->check_ori = check_orientation ? 1 : 0;
    m->device = device;
    *out = m;
    return AOS2_OK;
}

void aos2_matcher_destroy(aos2_matcher_t *m)
{
    if (!m) return;
    if (m->dev_ready) {
        (void)hipSetDevice(m->device);
        (void)hipStreamSynchronize(m->stream);
        m->arena.release();
        m->h_in.release();
        m->h_out.release();
        m->part.release();
        m->pool.release();
        m->fr_angle.release();
        m->fr_host.release();
        for (auto &e : m->ev) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(m->stream);
    }
    delete m;
}

float aos2_matcher_last_device_ms(const aos2_matcher_t *m) { return m ? m->last_ms : 0.f; }

int aos2_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

static int hamming_run(aos2_matcher *m, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int32_t *d_bi,
                       int32_t *d_bd, int32_t *d_sd, int iters, float *avg_ms)
{
    const int splits = std::max(1, std::min(64, (nt + 255) / 256));
    const int t_per_split = (((nt + splits - 1) / splits) + 255) & ~255;
    const int eff_splits = (nt + t_per_split - 1) / t_per_split;
    int st = m->part.alloc((size_t)2 * eff_splits * nq);
    if (st) return st;
    uint32_t *p1 = m->part.p, *p2 = m->part.p + (size_t)eff_splits * nq;
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(hamming_best2_kernel, dim3((nq + 255) / 256, eff_splits), dim3(256), 0, m->stream, d_q, nq,
                           d_t, nt, t_per_split, p1, p2);
        hipLaunchKernelGGL(hamming_merge_kernel, dim3((nq + 255) / 256), dim3(256), 0, m->stream, p1, p2, nq,
                           eff_splits, d_bi, d_bd, d_sd);
    }
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    AOS2_HIP_CHECK(hipStreamSynchronize(m->stream));
    AOS2_HIP_CHECK(hipGetLastError());
    if (avg_ms) {
        float ms = 0;
        AOS2_HIP_CHECK(hipEventElapsedTime(&ms, m->ev[0], m->ev[1]));
        *avg_ms = ms / iters;
    }
    return AOS2_OK;
}

int aos2_matcher_hamming_best2_device(aos2_matcher_t *m, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt,
                                      int32_t *d_best_idx, int32_t *d_best_dist, int32_t *d_second_dist, int iters,
                                      float *avg_ms)
{
    if (!m || !d_q || !d_t || nq <= 0 || nt <= 0 || nt >= (1 << 20) || !d_best_idx || !d_best_dist || !d_second_dist ||
        iters <= 0) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = matcher_init(m);
    if (st) return st;
    return hamming_run(m, d_q, nq, d_t, nt, d_best_idx, d_best_dist, d_second_dist, iters, avg_ms);
}

int aos2_matcher_hamming_best2(aos2_matcher_t *m, const uint8_t *q, int nq, const uint8_t *t, int nt,
                               int32_t *best_idx, int32_t *best_dist, int32_t *second_dist)
{
    if (!m || !q || !t || nq <= 0 || nt <= 0 || nt >= (1 << 20) || !best_idx || !best_dist || !second_dist) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = matcher_init(m);
    if (st) return st;
    Arena A{m};
    const size_t oq = A.push(q, (size_t)nq * 32), ot = A.push(t, (size_t)nt * 32);
    const size_t o1 = A.reserve((size_t)nq * 4), o2 = A.reserve((size_t)nq * 4), o3 = A.reserve((size_t)nq * 4);
    if ((st = A.upload())) return st;
    st = hamming_run(m, A.dev<uint8_t>(oq), nq, A.dev<uint8_t>(ot), nt, A.dev<int32_t>(o1), A.dev<int32_t>(o2),
                     A.dev<int32_t>(o3), 1, nullptr);
    if (st) return st;
    A.fetch(best_idx, o1, (size_t)nq * 4);
    A.fetch(best_dist, o2, (size_t)nq * 4);
    A.fetch(second_dist, o3, (size_t)nq * 4);
    if ((st = A.finish())) return st;
    return AOS2_OK;
}

// shared by SearchByBoW(KF, F) (kf_kf = 0, match_out[p] has n_f entries) and SearchByBoW(KF1, KF2)
// (kf_kf = 1, f_has_mp[p] = has_mp2, match_out[p] has n_kf entries)
// dev_inputs: desc_kf / desc_f / angle_kf / angle_f of the pairs are DEVICE arrays (the descriptors and keys of frames that
// already live in HBM, e.g. a device-resident Frames batch): they are used in place; the FeatureVector CSRs, kf_has_mp
// and the results stay host arrays (a few KB per pair)
static int bow_run(aos2_matcher_t *m, const aos2_bow_pair_t *pairs, const uint8_t *const *f_has_mp, int n_pairs, int kf_kf,
                   int32_t *const *match_f, int32_t *nmatches, bool dev_inputs = false)
{
    int st = matcher_init(m);
    if (st) return st;
    Arena A{m};
    struct Off { size_t o[13]; int nq; };
    std::vector<Off> offs(n_pairs);
    std::vector<BowQuery> queries;
    int max_nf = 0, max_q = 0;
    for (int p = 0; p < n_pairs; ++p) {
        const aos2_bow_pair_t &P = pairs[p];
        if (P.n_kf < 0 || P.n_f < 0 || P.n_nodes_kf < 0 || P.n_nodes_f < 0 || !match_f[p]) {
            set_error("bad BoW pair %d", p);
            return AOS2_ERR_ARG;
        }
        // merge-join of the two FeatureVectors (:181-258) on the host: the visiting order of the
        // KF features and the bucket each one is compared against
        queries.clear();
        size_t ent = 0;
        int ik = 0, jf = 0;
        while (ik < P.n_nodes_kf && jf < P.n_nodes_f) {
            const int idk = P.node_id_kf[ik], idf = P.node_id_f[jf];
            if (idk == idf) {
                const int b0 = P.node_off_f[jf], bc = P.node_off_f[jf + 1] - b0;
                for (int a = P.node_off_kf[ik]; a < P.node_off_kf[ik + 1]; ++a) {
                    const int kf = P.node_idx_kf[a];
                    if (kf < 0 || kf >= P.n_kf) {
                        set_error("BoW pair %d: feature index out of range", p);
                        return AOS2_ERR_ARG;
                    }
                    if (!P.kf_has_mp[kf]) continue;  // no map point / bad (:194-198)
                    queries.push_back(BowQuery{kf, b0, bc, (int32_t)ent});
                    ent += (size_t)bc;
                }
                ik++;
                jf++;
            } else if (idk < idf) {
                while (ik < P.n_nodes_kf && P.node_id_kf[ik] < idf) ik++;  // lower_bound
            } else {
                while (jf < P.n_nodes_f && P.node_id_f[jf] < idk) jf++;
            }
        }
        if (ent > (size_t)1 << 28) {
            set_error("BoW pair %d needs %zu distance entries", p, ent);
            return AOS2_ERR_ARG;
        }
        Off &o = offs[p];
        o.nq = (int)queries.size();
        if (!dev_inputs) {
            o.o[0] = A.push(P.desc_kf, (size_t)P.n_kf * 32);
            o.o[1] = A.push(P.desc_f, (size_t)P.n_f * 32);
            o.o[2] = A.push(P.angle_kf, (size_t)P.n_kf * 4);
            o.o[3] = A.push(P.angle_f, (size_t)P.n_f * 4);
        }
        o.o[4] = A.push(P.node_idx_f, (size_t)(P.n_nodes_f ? P.node_off_f[P.n_nodes_f] : 0) * 4);
        o.o[5] = A.push(queries.data(), queries.size() * sizeof(BowQuery));
        o.o[6] = A.reserve(ent * sizeof(Entry) + 8);
        o.o[7] = A.reserve_out((size_t)P.n_f * 4 + 4);  // match_f
        o.o[8] = A.reserve((size_t)P.n_f * 4 + 4);  // bin_f
        o.o[9] = A.reserve_out(4);                      // nmatches
        o.o[12] = A.reserve(queries.size() * 4 + 4);  // choice (parallel stage B)
        if (kf_kf) {
            o.o[10] = A.push(f_has_mp[p], (size_t)P.n_f);
            o.o[11] = A.reserve_out((size_t)P.n_kf * 8 + 8);  // match_1 | bin_1
        }
        max_nf = std::max(max_nf, P.n_f);
        max_q = std::max(max_q, o.nq);
    }
    if (max_nf > 60000) {
        set_error("n_f %d exceeds the LDS flag table (60000)", max_nf);
        return AOS2_ERR_ARG;
    }
    const size_t opairs = A.push_hole(sizeof(BowPairDev) * n_pairs);
    if ((st = A.alloc())) return st;
    std::vector<BowPairDev> dev(n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        const aos2_bow_pair_t &P = pairs[p];
        const Off &o = offs[p];
        BowPairDev &D = dev[p];
        D.n_kf = P.n_kf; D.n_f = P.n_f; D.n_queries = o.nq;
        if (dev_inputs) {
            D.desc_kf = P.desc_kf; D.desc_f = P.desc_f; D.angle_kf = P.angle_kf; D.angle_f = P.angle_f;
        } else {
            D.desc_kf = A.dev<uint8_t>(o.o[0]); D.desc_f = A.dev<uint8_t>(o.o[1]);
            D.angle_kf = A.dev<float>(o.o[2]); D.angle_f = A.dev<float>(o.o[3]);
        }
        D.node_idx_f = A.dev<int32_t>(o.o[4]); D.queries = A.dev<BowQuery>(o.o[5]); D.entries = A.dev<Entry>(o.o[6]);
        D.match_f = A.dev<int32_t>(o.o[7]); D.bin_f = A.dev<uint32_t>(o.o[8]); D.nmatches = A.dev<int32_t>(o.o[9]);
        D.kf_kf = kf_kf;
        D.choice = A.dev<int32_t>(o.o[12]);
        D.f_has_mp = nullptr; D.match_1 = nullptr; D.bin_1 = nullptr;
        if (kf_kf) {
            D.f_has_mp = A.dev<uint8_t>(o.o[10]);
            D.match_1 = A.dev<int32_t>(o.o[11]);
            D.bin_1 = D.match_1 + P.n_kf;
        }
    }
    memcpy(A.hostptr(opairs), dev.data(), sizeof(BowPairDev) * n_pairs);
    if ((st = A.upload())) return st;
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    if (max_q > 0)
        hipLaunchKernelGGL(bow_distances_kernel, dim3(max_q, n_pairs), dim3(64), 0, m->stream, A.dev<BowPairDev>(opairs));
    if ((size_t)max_nf * 8 <= kFixLdsBytes && !m->serial_resolve)
        hipLaunchKernelGGL(bow_resolve_fix_kernel, dim3(n_pairs), dim3(512), (size_t)max_nf * 8 + 16, m->stream,
                           A.dev<BowPairDev>(opairs), m->nnratio, m->check_ori);
    else
        hipLaunchKernelGGL(bow_resolve_kernel, dim3(n_pairs), dim3(64), (size_t)max_nf + 16, m->stream,
                           A.dev<BowPairDev>(opairs), m->nnratio, m->check_ori);
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    for (int p = 0; p < n_pairs; ++p) {
        if (!kf_kf)
            A.fetch(match_f[p], offs[p].o[7], (size_t)pairs[p].n_f * 4);
        else
            A.fetch(match_f[p], offs[p].o[11], (size_t)pairs[p].n_kf * 4);
        A.fetch(&nmatches[p], offs[p].o[9], 4);
    }
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

int aos2_matcher_search_by_bow(aos2_matcher_t *m, const aos2_bow_pair_t *pairs, int n_pairs, int32_t *const *match_f,
                               int32_t *nmatches)
{
    if (!m || !pairs || n_pairs <= 0 || !match_f || !nmatches) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    return bow_run(m, pairs, nullptr, n_pairs, 0, match_f, nmatches);
}

int aos2_matcher_search_by_bow_device(aos2_matcher_t *m, const aos2_bow_pair_t *pairs, int n_pairs, int32_t *const *match_f,
                                      int32_t *nmatches)
{
    if (!m || !pairs || n_pairs <= 0 || !match_f || !nmatches) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    return bow_run(m, pairs, nullptr, n_pairs, 0, match_f, nmatches, true);
}

// mvKeys[i].angle of [n][cap] keypoints (28-byte cv::KeyPoint records) as the dense float array the search reads
__global__ void key_angles_kernel(const aos2_keypoint_t *__restrict__ kps, float *__restrict__ angle, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) angle[i] = kps[i].angle;
}

int aos2_matcher_search_by_bow_frames(aos2_matcher_t *m, const aos2_bow_frames_t *q, int32_t *match_f, int32_t *nmatches)
{
    if (!m || !q || !match_f || !nmatches || q->n_frames <= 0 || q->cap <= 0 || !q->d_desc_kf || !q->d_kps_kf || !q->d_n_kf || !q->d_desc_f ||
        !q->d_kps_f || !q->d_n_f || !q->kf_has_mp || !q->d_kf_fv_node || !q->d_kf_fv_off || !q->d_kf_fv_idx || !q->d_kf_n_fv ||
        !q->d_f_fv_node || !q->d_f_fv_off || !q->d_f_fv_idx || !q->d_f_n_fv) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = matcher_init(m);
    if (st) return st;
    const size_t n = (size_t)q->n_frames, cap = (size_t)q->cap, tot = n * cap;
    if ((st = m->fr_angle.alloc(2 * tot))) return st;
    // host copies: per side n counts | n fv counts | fv_node [n][cap] | fv_off [n][cap + 1] | fv_idx [n][cap]
    const size_t side = 4 * (2 * n + tot + n * (cap + 1) + tot);
    if ((st = m->fr_host.alloc(2 * side))) return st;
    hipStream_t s = m->stream;
    hipLaunchKernelGGL(key_angles_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, q->d_kps_kf, m->fr_angle.p, tot);
    hipLaunchKernelGGL(key_angles_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, q->d_kps_f, m->fr_angle.p + tot, tot);
    struct Side { int32_t *n, *n_fv, *node, *off, *idx; } H[2];
    const int32_t *src[2][5] = {{q->d_n_kf, q->d_kf_n_fv, q->d_kf_fv_node, q->d_kf_fv_off, q->d_kf_fv_idx},
                                {q->d_n_f, q->d_f_n_fv, q->d_f_fv_node, q->d_f_fv_off, q->d_f_fv_idx}};
    const size_t cnt[5] = {n, n, tot, n * (cap + 1), tot};
    for (int k = 0; k < 2; ++k) {
        int32_t *base = reinterpret_cast<int32_t *>(m->fr_host.p + k * side);
        int32_t **dst[5] = {&H[k].n, &H[k].n_fv, &H[k].node, &H[k].off, &H[k].idx};
        for (int a = 0; a < 5; ++a) {
            *dst[a] = base;
            AOS2_HIP_CHECK(hipMemcpyAsync(base, src[k][a], 4 * cnt[a], hipMemcpyDeviceToHost, s));
            base += cnt[a];
        }
    }
    AOS2_HIP_CHECK(hipStreamSynchronize(s));
    std::vector<aos2_bow_pair_t> pairs(n);
    std::vector<int32_t *> mptr(n);
    for (size_t b = 0; b < n; ++b) {
        aos2_bow_pair_t &P = pairs[b];
        P.n_kf = H[0].n[b]; P.n_f = H[1].n[b];
        if (P.n_kf < 0 || P.n_kf > (int)cap || P.n_f < 0 || P.n_f > (int)cap || H[0].n_fv[b] < 0 || H[0].n_fv[b] > (int)cap || H[1].n_fv[b] < 0 ||
            H[1].n_fv[b] > (int)cap) {
            set_error("frame %zu: counts outside the capacity %zu", b, cap);
            return AOS2_ERR_ARG;
        }
        // the FeatureVector CSRs came back from the device (a transform that was not ordered before this call leaves stale
        // buffers): bow_run uses them as host copy lengths and indices, so offsets must be monotone from 0 and indices in range
        for (int k = 0; k < 2; ++k) {
            const int32_t *off = H[k].off + b * (cap + 1), *idx = H[k].idx + b * cap;
            const int nfv = H[k].n_fv[b], nfe = k == 0 ? P.n_kf : P.n_f;
            bool ok = nfv == 0 || off[0] == 0;
            for (int v = 0; ok && v < nfv; ++v) ok = off[v + 1] >= off[v] && off[v + 1] <= (int32_t)cap;
            for (int e = 0; ok && nfv > 0 && e < off[nfv]; ++e) ok = idx[e] >= 0 && idx[e] < nfe;
            if (!ok) {
                set_error("frame %zu: inconsistent FeatureVector (%s side): offsets not monotone within the capacity or a feature index out of range",
                          b, k == 0 ? "keyframe" : "frame");
                return AOS2_ERR_ARG;
            }
        }
        P.desc_kf = q->d_desc_kf + b * cap * 32; P.desc_f = q->d_desc_f + b * cap * 32;
        P.angle_kf = m->fr_angle.p + b * cap; P.angle_f = m->fr_angle.p + tot + b * cap;
        P.kf_has_mp = q->kf_has_mp + b * cap;
        P.n_nodes_kf = H[0].n_fv[b]; P.n_nodes_f = H[1].n_fv[b];
        P.node_id_kf = H[0].node + b * cap; P.node_off_kf = H[0].off + b * (cap + 1); P.node_idx_kf = H[0].idx + b * cap;
        P.node_id_f = H[1].node + b * cap; P.node_off_f = H[1].off + b * (cap + 1); P.node_idx_f = H[1].idx + b * cap;
        mptr[b] = match_f + b * cap;
    }
    return bow_run(m, pairs.data(), nullptr, (int)n, 0, mptr.data(), nmatches, true);
}

int aos2_matcher_search_by_bow_kf(aos2_matcher_t *m, const aos2_bow_kf_pair_t *pairs, int n_pairs, int32_t *const *match12,
                                  int32_t *nmatches)
{
    if (!m || !pairs || n_pairs <= 0 || !match12 || !nmatches) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    std::vector<aos2_bow_pair_t> v(n_pairs);
    std::vector<const uint8_t *> hm2(n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        const aos2_bow_kf_pair_t &K = pairs[p];
        if (K.n2 > 0 && !K.has_mp2) {
            set_error("bad BoW pair %d", p);
            return AOS2_ERR_ARG;
        }
        aos2_bow_pair_t &B = v[p];
        B.n_kf = K.n1; B.n_f = K.n2;
        B.desc_kf = K.desc1; B.desc_f = K.desc2;
        B.kf_has_mp = K.has_mp1;
        B.angle_kf = K.angle1; B.angle_f = K.angle2;
        B.n_nodes_kf = K.n_nodes1; B.n_nodes_f = K.n_nodes2;
        B.node_id_kf = K.node_id1; B.node_off_kf = K.node_off1; B.node_idx_kf = K.node_idx1;
        B.node_id_f = K.node_id2; B.node_off_f = K.node_off2; B.node_idx_f = K.node_idx2;
        hm2[p] = K.has_mp2;
    }
    return bow_run(m, v.data(), hm2.data(), n_pairs, 1, match12, nmatches);
}

int aos2_matcher_search_for_triangulation(aos2_matcher_t *m, const aos2_triang_pair_t *pairs, int n_pairs, int only_stereo,
                                          int32_t *const *match12, int32_t *nmatches)
{
    if (!m || !pairs || n_pairs <= 0 || !match12 || !nmatches) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = matcher_init(m);
    if (st) return st;
    Arena A{m};
    struct Off { size_t o[20]; int nq; };
    std::vector<Off> offs(n_pairs);
    std::vector<TriQuery> queries;
    int max_q = 0;
    for (int p = 0; p < n_pairs; ++p) {
        const aos2_triang_pair_t &P = pairs[p];
        if (P.n1 < 0 || P.n2 < 0 || P.n_nodes1 < 0 || P.n_nodes2 < 0 || !match12[p] || P.n_levels2 <= 0) {
            set_error("bad triangulation pair %d", p);
            return AOS2_ERR_ARG;
        }
        queries.clear();
        int i1 = 0, i2 = 0;
        while (i1 < P.n_nodes1 && i2 < P.n_nodes2) {  // merge-join of the FeatureVectors (:691-772)
            const int a = P.node_id1[i1], b = P.node_id2[i2];
            if (a == b) {
                const int b0 = P.node_off2[i2], bc = P.node_off2[i2 + 1] - b0;
                for (int k = P.node_off1[i1]; k < P.node_off1[i1 + 1]; ++k) {
                    const int idx1 = P.node_idx1[k];
                    if (idx1 < 0 || idx1 >= P.n1) {
                        set_error("triangulation pair %d: feature index out of range", p);
                        return AOS2_ERR_ARG;
                    }
                    if (P.has_mp1[idx1]) continue;                            // :700-703
                    if (only_stereo && !(P.u_right1[idx1] >= 0)) continue;    // :705-709
                    queries.push_back(TriQuery{idx1, b0, bc});
                }
                i1++;
                i2++;
            } else if (a < b) {
                while (i1 < P.n_nodes1 && P.node_id1[i1] < b) i1++;
            } else {
                while (i2 < P.n_nodes2 && P.node_id2[i2] < a) i2++;
            }
        }
        Off &o = offs[p];
        o.nq = (int)queries.size();
        const size_t n1 = (size_t)P.n1, n2 = (size_t)P.n2;
        o.o[0] = A.push(P.desc1, n1 * 32); o.o[1] = A.push(P.desc2, n2 * 32); o.o[2] = A.push(P.has_mp2, n2);
        o.o[3] = A.push(P.x1, n1 * 4); o.o[4] = A.push(P.y1, n1 * 4); o.o[5] = A.push(P.angle1, n1 * 4);
        o.o[6] = A.push(P.u_right1, n1 * 4);
        o.o[7] = A.push(P.x2, n2 * 4); o.o[8] = A.push(P.y2, n2 * 4); o.o[9] = A.push(P.angle2, n2 * 4);
        o.o[10] = A.push(P.u_right2, n2 * 4); o.o[11] = A.push(P.octave2, n2 * 4);
        o.o[12] = A.push(P.scale_factors2, (size_t)P.n_levels2 * 4);
        o.o[13] = A.push(P.level_sigma2_2, (size_t)P.n_levels2 * 4);
        o.o[14] = A.push(P.node_idx2, (size_t)(P.n_nodes2 ? P.node_off2[P.n_nodes2] : 0) * 4);
        o.o[15] = A.push(queries.data(), queries.size() * sizeof(TriQuery));
        std::vector<int32_t> init(n1 + 1, -1);
        o.o[16] = A.push(init.data(), (n1 + 1) * 4);  // match12 = -1 (:681)
        o.o[17] = A.reserve(8);
        max_q = std::max(max_q, o.nq);
    }
    const size_t opairs = A.push_hole(sizeof(TriPairDev) * n_pairs);
    if ((st = A.alloc())) return st;
    std::vector<TriPairDev> dev(n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        const aos2_triang_pair_t &P = pairs[p];
        const Off &o = offs[p];
        TriPairDev &D = dev[p];
        D.n1 = P.n1; D.n2 = P.n2; D.n_queries = o.nq; D.only_stereo = only_stereo;
        D.desc1 = A.dev<uint8_t>(o.o[0]); D.desc2 = A.dev<uint8_t>(o.o[1]); D.has_mp2 = A.dev<uint8_t>(o.o[2]);
        D.x1 = A.dev<float>(o.o[3]); D.y1 = A.dev<float>(o.o[4]); D.angle1 = A.dev<float>(o.o[5]); D.u_right1 = A.dev<float>(o.o[6]);
        D.x2 = A.dev<float>(o.o[7]); D.y2 = A.dev<float>(o.o[8]); D.angle2 = A.dev<float>(o.o[9]); D.u_right2 = A.dev<float>(o.o[10]);
        D.octave2 = A.dev<int32_t>(o.o[11]); D.scale_factors2 = A.dev<float>(o.o[12]); D.level_sigma2_2 = A.dev<float>(o.o[13]);
        memcpy(D.F12, P.F12, sizeof(D.F12));
        D.ex = P.ex; D.ey = P.ey;
        D.node_idx2 = A.dev<int32_t>(o.o[14]); D.queries = A.dev<TriQuery>(o.o[15]);
        D.match12 = A.dev<int32_t>(o.o[16]); D.nmatches = A.dev<int32_t>(o.o[17]);
    }
    memcpy(A.hostptr(opairs), dev.data(), sizeof(TriPairDev) * n_pairs);
    if ((st = A.upload())) return st;
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    if (max_q > 0)
        hipLaunchKernelGGL(triang_match_kernel, dim3(max_q, n_pairs), dim3(64), 0, m->stream, A.dev<TriPairDev>(opairs));
    hipLaunchKernelGGL(triang_finish_kernel, dim3(n_pairs), dim3(64), 0, m->stream, A.dev<TriPairDev>(opairs), m->check_ori);
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    for (int p = 0; p < n_pairs; ++p) {
        if (pairs[p].n1 > 0)
            A.fetch_dev(match12[p], dev[p].match12, (size_t)pairs[p].n1 * 4);
        A.fetch_dev(&nmatches[p], dev[p].nmatches, 4);
    }
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

int aos2_compute_distinctive_descriptors(aos2_matcher_t *m, int n_points, const int32_t *off, const uint8_t *desc,
                                         int32_t *best_idx)
{
    if (!m || n_points < 0 || (n_points > 0 && (!off || !best_idx))) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (n_points == 0) return AOS2_OK;
    for (int p = 0; p < n_points; ++p)
        if (off[p + 1] < off[p] || off[0] != 0) {
            set_error("observation offsets must start at 0 and ascend");
            return AOS2_ERR_ARG;
        }
    const size_t total = (size_t)off[n_points];
    if (total > 0 && !desc) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = matcher_init(m);
    if (st) return st;
    Arena A{m};
    const size_t o0 = A.push(off, (size_t)(n_points + 1) * 4), o1 = A.push(desc, total * 32), o2 = A.reserve((size_t)n_points * 4);
    if ((st = A.upload())) return st;
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    hipLaunchKernelGGL(distinctive_kernel, dim3(n_points), dim3(64), 0, m->stream, n_points, A.dev<int32_t>(o0),
                       A.dev<uint8_t>(o1), A.dev<int32_t>(o2));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    A.fetch_dev(best_idx, A.dev<int32_t>(o2), (size_t)n_points * 4);
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

int aos2_matcher_search_by_projection(aos2_matcher_t *m, const aos2_frame_view_t *f, const aos2_proj_mp_t *p, float th,
                                      int32_t *match_f, int32_t *nmatches)
{
    if (!m || !p || !match_f || !nmatches || p->n_mp < 0) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = check_frame(f);
    if (st) return st;
    if (p->n_mp > 0 && (!p->track_in_view || !p->desc || !p->has_obs || !p->view_cos || !p->proj_x || !p->proj_y || !p->proj_xr)) {
        set_error("bad map point set (NULL array)");
        return AOS2_ERR_ARG;
    }
    if ((st = check_levels(p->pred_level, p->n_mp, f->n_levels, "pred_level", p->track_in_view))) return st;
    if ((st = matcher_init(m))) return st;
    Arena A{m};
    size_t fo[12];
    fill_frame(A, f, fo);
    const size_t n = (size_t)p->n_mp;
    const size_t o0 = A.push(p->track_in_view, n), o1 = A.push(p->desc, n * 32), o2 = A.push(p->has_obs, n);
    const size_t o3 = A.push(p->pred_level, n * 4), o4 = A.push(p->view_cos, n * 4), o5 = A.push(p->proj_x, n * 4);
    const size_t o6 = A.push(p->proj_y, n * 4), o7 = A.push(p->proj_xr, n * 4);
    const size_t om = A.reserve_out((size_t)f->n_f * 4 + 4), on = A.reserve_out(8);
    const size_t oslots = A.reserve((size_t)(p->n_mp + 1) * sizeof(QuerySlot));
    const size_t ochoice = A.reserve((size_t)(p->n_mp + 1) * 4);
    // Entry pool.  Usual sizes: every query owns a slice that holds the whole frame (one pass, no counting).  Large local
    // maps (n_mp x n_f x 8 B beyond 64 MB): the pool is sized from the REAL window populations -- count pass, scan, fill
    // pass -- starting from a budget of 64 entries per query and, if the windows of this call hold more (the scan reports
    // the total), once more with exactly that many.  No size limit other than device memory.
    const size_t worst = (size_t)p->n_mp * (size_t)f->n_f;   // a window holds at most every feature
    const bool two_pass = worst * sizeof(Entry) > ((size_t)64 << 20) || getenv("AOS2_PROJ_TWO_PASS") != nullptr;
    size_t pool_cap = two_pass ? std::max<size_t>((size_t)p->n_mp * 64, 1024) : worst;
    if (const char *e = getenv("AOS2_PROJ_POOL_BUDGET")) pool_cap = two_pass ? (size_t)std::max(1, atoi(e)) : pool_cap;   // (tests: force the retry)
    if ((st = A.upload())) return st;
    FrameDev F = frame_dev(A, f, fo);
    ProjMpDev P{};
    P.n_mp = p->n_mp;
    P.track_in_view = A.dev<uint8_t>(o0); P.desc = A.dev<uint8_t>(o1); P.has_obs = A.dev<uint8_t>(o2);
    P.pred_level = A.dev<int32_t>(o3); P.view_cos = A.dev<float>(o4); P.proj_x = A.dev<float>(o5);
    P.proj_y = A.dev<float>(o6); P.proj_xr = A.dev<float>(o7);
    int32_t *d_used = A.dev<int32_t>(on) + 1;
    QuerySlot *d_slots = A.dev<QuerySlot>(oslots);
    for (int attempt = 0;; ++attempt) {
        if (pool_cap > ((size_t)1 << 31) - 2) {
            set_error("projection search: %zu candidate entries exceed the pool's index range", pool_cap);
            return AOS2_ERR_CAPACITY;
        }
        if ((st = m->pool.alloc(pool_cap + 1))) return st;
        Entry *d_pool = reinterpret_cast<Entry *>(m->pool.p);
        AOS2_HIP_CHECK(hipMemsetAsync(d_used, 0, 4, m->stream));
        if (attempt == 0) AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
        if (p->n_mp > 0) {
            if (!two_pass)
                hipLaunchKernelGGL(proj_mp_entries_kernel, dim3(p->n_mp), dim3(64), 0, m->stream, F, P, th, d_slots, d_pool, d_used,
                                   (int)pool_cap, 0);
            else {
                hipLaunchKernelGGL(proj_mp_entries_kernel, dim3(p->n_mp), dim3(64), 0, m->stream, F, P, th, d_slots, d_pool, d_used,
                                   (int)pool_cap, 1);
                hipLaunchKernelGGL(frames_scan_slots_kernel, dim3(1), dim3(256), 0, m->stream, d_slots, (const int32_t *)nullptr,
                                   p->n_mp, (int)pool_cap, d_used);
                hipLaunchKernelGGL(proj_mp_entries_kernel, dim3(p->n_mp), dim3(64), 0, m->stream, F, P, th, d_slots, d_pool, d_used,
                                   (int)pool_cap, 2);
            }
        }
        if ((size_t)f->n_f * 8 <= kFixLdsBytes && !m->serial_resolve)
            hipLaunchKernelGGL(proj_mp_resolve_fix_kernel, dim3(1), dim3(1024), (size_t)f->n_f * 8 + 16, m->stream, F, P, m->nnratio,
                               d_slots, reinterpret_cast<const Entry *>(d_pool), A.dev<int32_t>(om), A.dev<int32_t>(on),
                               A.dev<int32_t>(ochoice));
        else   // the one-wave sequential loop: frames with more features than the LDS copy of B holds, or AOS2_SERIAL_RESOLVE=1
            hipLaunchKernelGGL(proj_mp_resolve_kernel, dim3(1), dim3(64), (size_t)f->n_f + 16, m->stream, F, P, m->nnratio, d_slots,
                               reinterpret_cast<const Entry *>(d_pool), A.dev<int32_t>(om), A.dev<int32_t>(on));
        AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
        int32_t tail[2] = {0, 0};   // nmatches, overflow word
        A.fetch(match_f, om, (size_t)f->n_f * 4);
        A.fetch(tail, on, 8);
        if ((st = A.finish())) return st;
        if (two_pass && (size_t)tail[1] > pool_cap && attempt == 0) {   // the windows hold tail[1] entries: once more, exactly sized
            pool_cap = (size_t)tail[1];
            continue;
        }
        *nmatches = tail[0];
        break;
    }
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

int aos2_matcher_search_by_projection_batch(aos2_matcher_t *m, const aos2_frame_view_t *frames, const aos2_proj_mp_t *problems,
                                            int n_problems, float th, int32_t *const *match_f, int32_t *nmatches)
{
    if (!m || !frames || !problems || n_problems <= 0 || !match_f || !nmatches) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st;
    size_t pool_cap = 0;
    int max_mp = 0, max_nf = 0;
    for (int i = 0; i < n_problems; ++i) {
        if ((st = check_frame(&frames[i]))) return st;
        const aos2_proj_mp_t &pi = problems[i];
        if (pi.n_mp < 0 || !match_f[i] || (pi.n_mp > 0 && (!pi.track_in_view || !pi.desc || !pi.has_obs || !pi.view_cos || !pi.proj_x ||
                                                             !pi.proj_y || !pi.proj_xr))) {
            set_error("bad projection problem %d", i);
            return AOS2_ERR_ARG;
        }
        if ((st = check_levels(pi.pred_level, pi.n_mp, frames[i].n_levels, "pred_level", pi.track_in_view))) return st;
        // entry budget: a search window rarely holds more than a few dozen features; 512 per map point (or the
        // whole frame if smaller) is the bound here -- AOS2_ERR_CAPACITY if a problem ever needs more
        pool_cap += (size_t)problems[i].n_mp * (size_t)std::min(frames[i].n_f, 512);
        max_mp = std::max(max_mp, problems[i].n_mp);
        max_nf = std::max(max_nf, frames[i].n_f);
    }
    // (entry offsets are 32-bit: 2^31 entries = 16 GiB of the device's 288 GB; what actually limits a call is the allocation)
    if (pool_cap > ((size_t)1 << 31) - 2) {
        set_error("batched projection search: %zu candidate entries exceed the pool's index range", pool_cap);
        return AOS2_ERR_CAPACITY;
    }
    if ((st = matcher_init(m))) return st;
    Arena A{m};
    struct Off { size_t fo[12], o[8], om, on, oslots, ochoice; };
    std::vector<Off> offs(n_problems);
    for (int i = 0; i < n_problems; ++i) {
        const aos2_proj_mp_t *p = &problems[i];
        Off &o = offs[i];
        fill_frame(A, &frames[i], o.fo);
        const size_t n = (size_t)p->n_mp;
        o.o[0] = A.push(p->track_in_view, n); o.o[1] = A.push(p->desc, n * 32); o.o[2] = A.push(p->has_obs, n);
        o.o[3] = A.push(p->pred_level, n * 4); o.o[4] = A.push(p->view_cos, n * 4); o.o[5] = A.push(p->proj_x, n * 4);
        o.o[6] = A.push(p->proj_y, n * 4); o.o[7] = A.push(p->proj_xr, n * 4);
        o.om = A.reserve_out((size_t)frames[i].n_f * 4 + 4);
        o.on = A.reserve_out(8);
        o.oslots = A.reserve((n + 1) * sizeof(QuerySlot));
        o.ochoice = A.reserve((n + 1) * 4);
    }
    const size_t oitems = A.push_hole(sizeof(ProjMpItem) * (size_t)n_problems);
    const size_t oused = A.reserve_out((size_t)n_problems * 256 + 8);   // one counter per problem, 256 B apart
    if ((st = m->pool.alloc(pool_cap + 1))) return st;
    if ((st = A.alloc())) return st;
    std::vector<ProjMpItem> items(n_problems);
    size_t pool_next = 0;
    for (int i = 0; i < n_problems; ++i) {
        const aos2_proj_mp_t *p = &problems[i];
        const Off &o = offs[i];
        ProjMpItem &it = items[i];
        it.F = frame_dev(A, &frames[i], o.fo);
        it.P = ProjMpDev{};
        it.P.n_mp = p->n_mp;
        it.P.track_in_view = A.dev<uint8_t>(o.o[0]); it.P.desc = A.dev<uint8_t>(o.o[1]); it.P.has_obs = A.dev<uint8_t>(o.o[2]);
        it.P.pred_level = A.dev<int32_t>(o.o[3]); it.P.view_cos = A.dev<float>(o.o[4]); it.P.proj_x = A.dev<float>(o.o[5]);
        it.P.proj_y = A.dev<float>(o.o[6]); it.P.proj_xr = A.dev<float>(o.o[7]);
        it.slots = A.dev<QuerySlot>(o.oslots);
        it.match_f = A.dev<int32_t>(o.om);
        it.nmatches = A.dev<int32_t>(o.on);
        it.choice = A.dev<int32_t>(o.ochoice);
        it.pool_used = A.dev<int32_t>(oused) + 64 * (size_t)i;
        it.pool_cap = (int32_t)((size_t)p->n_mp * (size_t)std::min(frames[i].n_f, 512));
        it.pool_base = (int32_t)pool_next;
        pool_next += (size_t)it.pool_cap;
    }
    memcpy(A.hostptr(oitems), items.data(), sizeof(ProjMpItem) * (size_t)n_problems);
    if ((st = A.upload())) return st;
    int32_t *d_used = A.dev<int32_t>(oused);
    AOS2_HIP_CHECK(hipMemsetAsync(d_used, 0, (size_t)n_problems * 256, m->stream));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    if (max_mp > 0)
        hipLaunchKernelGGL(proj_mp_entries_batch_kernel, dim3(max_mp, n_problems), dim3(64), 0, m->stream,
                           A.dev<ProjMpItem>(oitems), th, reinterpret_cast<Entry *>(m->pool.p));
    if ((size_t)max_nf * 8 <= kFixLdsBytes && !m->serial_resolve)
        hipLaunchKernelGGL(proj_mp_resolve_fix_batch_kernel, dim3(n_problems), dim3(256), (size_t)max_nf * 8 + 16, m->stream,
                           A.dev<ProjMpItem>(oitems), m->nnratio, reinterpret_cast<const Entry *>(m->pool.p));
    else
        hipLaunchKernelGGL(proj_mp_resolve_batch_kernel, dim3(n_problems), dim3(64), (size_t)max_nf + 16, m->stream,
                           A.dev<ProjMpItem>(oitems), m->nnratio, reinterpret_cast<const Entry *>(m->pool.p));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    std::vector<int32_t> used((size_t)n_problems * 64);
    A.fetch(used.data(), oused, (size_t)n_problems * 256);
    for (int i = 0; i < n_problems; ++i) {
        A.fetch(match_f[i], offs[i].om, (size_t)frames[i].n_f * 4);
        A.fetch(&nmatches[i], offs[i].on, 4);
    }
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    for (int i = 0; i < n_problems; ++i)
        if (used[(size_t)i * 64] > 0) {   // overflow flag = a window population that did not fit its slice
            set_error("batched projection search: a search window of problem %d holds %d features, more than the %d "
                      "budgeted per map point; use aos2_matcher_search_by_projection per frame", i, used[(size_t)i * 64],
                      std::min(frames[i].n_f, 512));
            return AOS2_ERR_CAPACITY;
        }
    return AOS2_OK;
}

int aos2_matcher_search_by_projection_last(aos2_matcher_t *m, const aos2_frame_view_t *cur, const aos2_proj_last_t *p,
                                           float th, int mono, int32_t *match_f, int32_t *nmatches)
{
    if (!m || !p || !match_f || !nmatches || p->n_last < 0) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = check_frame(cur);
    if (st) return st;
    if (p->n_last > 0 && (!p->last_valid || !p->world_pos || !p->desc || !p->last_angle || !p->has_obs)) {
        set_error("bad last-frame point set (NULL array)");
        return AOS2_ERR_ARG;
    }
    if ((st = check_levels(p->last_octave, p->n_last, cur->n_levels, "last_octave", p->last_valid))) return st;
    if ((st = matcher_init(m))) return st;
    Arena A{m};
    size_t fo[12];
    fill_frame(A, cur, fo);
    const size_t n = (size_t)p->n_last;
    const size_t o0 = A.push(p->last_valid, n), o1 = A.push(p->desc, n * 32), o2 = A.push(p->has_obs, n);
    const size_t o3 = A.push(p->world_pos, n * 12), o4 = A.push(p->last_angle, n * 4), o5 = A.push(p->last_octave, n * 4);
    const size_t om = A.reserve_out((size_t)cur->n_f * 4 + 4), ob = A.reserve((size_t)cur->n_f * 4 + 4), on = A.reserve_out(8);
    const size_t oslots = A.reserve((size_t)(p->n_last + 1) * sizeof(QuerySlot));
    const size_t ochoice = A.reserve((size_t)(p->n_last + 1) * 4);
    const size_t pool_cap = (size_t)p->n_last * (size_t)cur->n_f;
    if (pool_cap > ((size_t)1 << 31) - 2) {   // (32-bit entry offsets: 16 GiB of the device's 288 GB)
        set_error("projection search of %d points x %d features exceeds the pool's index range", p->n_last, cur->n_f);
        return AOS2_ERR_CAPACITY;
    }
    if ((st = m->pool.alloc(pool_cap + 1))) return st;
    if ((st = A.upload())) return st;
    FrameDev F = frame_dev(A, cur, fo);
    ProjLastDev P{};
    P.n_last = p->n_last;
    P.last_valid = A.dev<uint8_t>(o0); P.desc = A.dev<uint8_t>(o1); P.has_obs = A.dev<uint8_t>(o2);
    P.world_pos = A.dev<float>(o3); P.last_angle = A.dev<float>(o4); P.last_octave = A.dev<int32_t>(o5);
    memcpy(P.Tcw, p->Tcw, sizeof(P.Tcw));
    memcpy(P.Tlw, p->Tlw, sizeof(P.Tlw));
    P.fx = p->fx; P.fy = p->fy; P.cx = p->cx; P.cy = p->cy; P.mb = p->mb; P.mbf = p->mbf;
    int32_t *d_used = A.dev<int32_t>(on) + 1;
    AOS2_HIP_CHECK(hipMemsetAsync(d_used, 0, 4, m->stream));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    if (p->n_last > 0)
        hipLaunchKernelGGL(proj_last_entries_kernel, dim3(p->n_last), dim3(64), 0, m->stream, F, P, th, mono ? 1 : 0,
                           A.dev<QuerySlot>(oslots), reinterpret_cast<Entry *>(m->pool.p), d_used, (int)pool_cap);
    if ((size_t)cur->n_f * 8 <= kFixLdsBytes && !m->serial_resolve)
        hipLaunchKernelGGL(proj_last_resolve_fix_kernel, dim3(1), dim3(1024), (size_t)cur->n_f * 8 + 16, m->stream, F, P,
                           m->check_ori, A.dev<QuerySlot>(oslots), reinterpret_cast<const Entry *>(m->pool.p),
                           A.dev<int32_t>(om), A.dev<uint32_t>(ob), A.dev<int32_t>(on), A.dev<int32_t>(ochoice));
    else
        hipLaunchKernelGGL(proj_last_resolve_kernel, dim3(1), dim3(64), (size_t)cur->n_f + 16, m->stream, F, P, m->check_ori,
                           A.dev<QuerySlot>(oslots), reinterpret_cast<const Entry *>(m->pool.p), A.dev<int32_t>(om),
                           A.dev<uint32_t>(ob), A.dev<int32_t>(on));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    A.fetch(match_f, om, (size_t)cur->n_f * 4);
    A.fetch(nmatches, on, 4);
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

// ---- projection family -------------------------------------------------
static int check_points(const aos2_proj_points_t *p, int mode)
{
    if (!p || p->n_pts < 0 || (p->n_pts > 0 && (!p->valid || !p->pos || !p->max_dist || !p->min_dist || !p->desc)) ||
        (p->n_pts > 0 && mode <= 2 && !p->normal) || (mode == 0 && !p->inv_level_sigma2) ||
        (p->n_pts > 0 && mode == 4 && !p->q_angle) || !(p->log_scale_factor > 0)) {
        set_error("bad point set");
        return AOS2_ERR_ARG;
    }
    return AOS2_OK;
}

static ProjGenDev points_dev(Arena &A, const aos2_proj_points_t *p, int mode, int n_levels)
{
    const size_t n = (size_t)p->n_pts;
    ProjGenDev P{};
    P.n_pts = p->n_pts;
    P.mode = mode;
    const size_t o0 = A.push(p->valid, n), o1 = A.push(p->desc, n * 32), o2 = A.push(p->pos, n * 12);
    const size_t o3 = A.push(p->max_dist, n * 4), o4 = A.push(p->min_dist, n * 4);
    const size_t o5 = p->normal ? A.push(p->normal, n * 12) : 0, o6 = p->q_angle ? A.push(p->q_angle, n * 4) : 0;
    const size_t o7 = p->inv_level_sigma2 ? A.push(p->inv_level_sigma2, (size_t)n_levels * 4) : 0;
    // device pointers are resolved after the arena is uploaded: keep offsets in the pointer fields for now
    P.valid = reinterpret_cast<const uint8_t *>(o0); P.desc = reinterpret_cast<const uint8_t *>(o1);
    P.pos = reinterpret_cast<const float *>(o2); P.max_dist = reinterpret_cast<const float *>(o3);
    P.min_dist = reinterpret_cast<const float *>(o4); P.normal = reinterpret_cast<const float *>(o5);
    P.q_angle = reinterpret_cast<const float *>(o6); P.inv_level_sigma2 = reinterpret_cast<const float *>(o7);
    memcpy(P.R, p->R, sizeof(P.R)); memcpy(P.t, p->t, sizeof(P.t)); memcpy(P.Ow, p->Ow, sizeof(P.Ow));
    memcpy(P.R2, p->R2, sizeof(P.R2)); memcpy(P.t2, p->t2, sizeof(P.t2));
    P.fx = p->fx; P.fy = p->fy; P.cx = p->cx; P.cy = p->cy; P.bf = p->bf;
    P.log_scale_factor = p->log_scale_factor;
    P.th = p->th;
    return P;
}

static void points_resolve(const Arena &A, ProjGenDev &P)
{
    auto fix = [&](auto *&ptr) {
        using T = std::remove_reference_t<decltype(ptr)>;
        ptr = reinterpret_cast<T>(A.m->arena.p + reinterpret_cast<size_t>(ptr));
    };
    fix(P.valid); fix(P.desc); fix(P.pos); fix(P.max_dist); fix(P.min_dist); fix(P.normal); fix(P.q_angle);
    fix(P.inv_level_sigma2);
}

// independent points: modes 0 (Fuse), 1 (Fuse with Scw)
int aos2_matcher_fuse(aos2_matcher_t *m, const aos2_frame_view_t *kf, const aos2_proj_points_t *p, int sim3,
                      int32_t *best_idx, int32_t *best_dist, int32_t *n_fused)
{
    const int mode = sim3 ? 1 : 0;
    if (!m || !best_idx || !best_dist) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = check_frame(kf);
    if (st) return st;
    if ((st = check_points(p, mode))) return st;
    if (n_fused) *n_fused = 0;
    if (p->n_pts == 0) return AOS2_OK;
    if ((st = matcher_init(m))) return st;
    Arena A{m};
    size_t fo[12];
    fill_frame(A, kf, fo);
    ProjGenDev P = points_dev(A, p, mode, kf->n_levels);
    const size_t ob = A.reserve((size_t)p->n_pts * 8 + 8);
    if ((st = A.upload())) return st;
    points_resolve(A, P);
    FrameDev F = frame_dev(A, kf, fo);
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    hipLaunchKernelGGL(projgen_best_kernel, dim3(p->n_pts), dim3(64), 0, m->stream, F, P, TH_LOW, A.dev<int32_t>(ob),
                       A.dev<int32_t>(ob) + p->n_pts);
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    A.fetch_dev(best_idx, A.dev<int32_t>(ob), (size_t)p->n_pts * 4);
    A.fetch_dev(best_dist, A.dev<int32_t>(ob) + p->n_pts, (size_t)p->n_pts * 4);
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    if (n_fused) {
        int c = 0;
        for (int i = 0; i < p->n_pts; ++i) c += best_idx[i] >= 0;
        *n_fused = c;
    }
    return AOS2_OK;
}

int aos2_matcher_search_by_sim3(aos2_matcher_t *m, const aos2_frame_view_t *kf1, const aos2_frame_view_t *kf2,
                                const aos2_proj_points_t *p12, const aos2_proj_points_t *p21, int32_t *match12,
                                int32_t *n_found)
{
    if (!m || !match12 || !n_found) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st;
    if ((st = check_frame(kf1)) || (st = check_frame(kf2)) || (st = check_points(p12, 3)) || (st = check_points(p21, 3))) return st;
    if (p12->n_pts != kf1->n_f || p21->n_pts != kf2->n_f) {
        set_error("SearchBySim3: one (possibly invalid) map point per keyframe feature is expected");
        return AOS2_ERR_ARG;
    }
    *n_found = 0;
    if (p12->n_pts == 0) return AOS2_OK;
    if ((st = matcher_init(m))) return st;
    Arena A{m};
    size_t f1o[12], f2o[12];
    fill_frame(A, kf1, f1o);
    fill_frame(A, kf2, f2o);
    ProjGenDev P12 = points_dev(A, p12, 3, kf2->n_levels), P21 = points_dev(A, p21, 3, kf1->n_levels);
    const size_t n1 = (size_t)p12->n_pts, n2 = (size_t)p21->n_pts;
    const size_t o1 = A.reserve((n1 + 1) * 8), o2 = A.reserve((n2 + 1) * 8), om = A.reserve((n1 + 1) * 4), on = A.reserve(8);
    if ((st = A.upload())) return st;
    points_resolve(A, P12);
    points_resolve(A, P21);
    FrameDev F1 = frame_dev(A, kf1, f1o), F2 = frame_dev(A, kf2, f2o);
    AOS2_HIP_CHECK(hipMemsetAsync(A.dev<int32_t>(on), 0, 8, m->stream));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    hipLaunchKernelGGL(projgen_best_kernel, dim3((unsigned)n1), dim3(64), 0, m->stream, F2, P12, TH_HIGH, A.dev<int32_t>(o1),
                       A.dev<int32_t>(o1) + n1);
    if (n2 > 0)
        hipLaunchKernelGGL(projgen_best_kernel, dim3((unsigned)n2), dim3(64), 0, m->stream, F1, P21, TH_HIGH, A.dev<int32_t>(o2),
                           A.dev<int32_t>(o2) + n2);
    hipLaunchKernelGGL(sim3_agree_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, m->stream, A.dev<int32_t>(o1),
                       A.dev<int32_t>(o2), (int)n1, (int)n2, A.dev<int32_t>(om), A.dev<int32_t>(on));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    A.fetch_dev(match12, A.dev<int32_t>(om), n1 * 4);
    A.fetch_dev(n_found, A.dev<int32_t>(on), 4);
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

int aos2_frame_assign_features_to_grid(aos2_matcher_t *m, int n, const float *kp_x, const float *kp_y, float min_x,
                                       float min_y, float grid_w_inv, float grid_h_inv, int32_t *grid_off, int32_t *grid_idx,
                                       int32_t *n_in_grid)
{
    if (!m || n < 0 || !grid_off || (n > 0 && (!kp_x || !kp_y || !grid_idx))) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (n > 32000) {
        set_error("AssignFeaturesToGrid: more than 32000 features");
        return AOS2_ERR_CAPACITY;
    }
    int st = matcher_init(m);
    if (st) return st;
    constexpr int NC = GRID_COLS * GRID_ROWS;
    Arena A{m};
    const size_t ox = A.push(kp_x, (size_t)n * 4), oy = A.push(kp_y, (size_t)n * 4);
    const size_t oo = A.reserve((size_t)(NC + 1) * 4), oi = A.reserve((size_t)n * 4 + 4);
    if ((st = A.upload())) return st;
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    hipLaunchKernelGGL(assign_grid_kernel, dim3(1), dim3(256), (size_t)(NC + 1) * 4 + (size_t)n * 2 + 16, m->stream, n,
                       A.dev<float>(ox), A.dev<float>(oy), min_x, min_y, grid_w_inv, grid_h_inv, A.dev<int32_t>(oo),
                       A.dev<int32_t>(oi));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    A.fetch_dev(grid_off, A.dev<int32_t>(oo), (size_t)(NC + 1) * 4);
    if (n > 0) A.fetch_dev(grid_idx, A.dev<int32_t>(oi), (size_t)n * 4);
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    if (n_in_grid) *n_in_grid = grid_off[NC];
    return AOS2_OK;
}

int aos2_frame_stereo_from_rgbd(aos2_matcher_t *m, int n, const float *kp_x, const float *kp_y, const float *kpun_x,
                                const float *depth_img, int w, int h, int stride, float mbf, float *u_right, float *depth)
{
    if (!m || n < 0 || (n > 0 && (!kp_x || !kp_y || !kpun_x || !depth_img || !u_right || !depth)) || w <= 0 || h <= 0 ||
        stride < w) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (n == 0) return AOS2_OK;
    for (int i = 0; i < n; ++i)
        if (!((int)kp_x[i] >= 0 && (int)kp_x[i] < w && (int)kp_y[i] >= 0 && (int)kp_y[i] < h)) {
            set_error("ComputeStereoFromRGBD: keypoint %d (%g, %g) outside the depth image", i, kp_x[i], kp_y[i]);
            return AOS2_ERR_ARG;
        }
    int st = matcher_init(m);
    if (st) return st;
    Arena A{m};
    const size_t ox = A.push(kp_x, (size_t)n * 4), oy = A.push(kp_y, (size_t)n * 4), ou = A.push(kpun_x, (size_t)n * 4);
    const size_t od = A.push(depth_img, (size_t)stride * h * 4), oo = A.reserve((size_t)n * 8 + 8);
    if ((st = A.upload())) return st;
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    hipLaunchKernelGGL(stereo_from_rgbd_kernel, dim3((n + 255) / 256), dim3(256), 0, m->stream, n, A.dev<float>(ox),
                       A.dev<float>(oy), A.dev<float>(ou), A.dev<float>(od), stride, mbf, A.dev<float>(oo), A.dev<float>(oo) + n);
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    A.fetch_dev(u_right, A.dev<float>(oo), (size_t)n * 4);
    A.fetch_dev(depth, A.dev<float>(oo) + n, (size_t)n * 4);
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

int aos2_frame_is_in_frustum(aos2_matcher_t *m, const aos2_proj_points_t *p, float min_x, float max_x, float min_y,
                             float max_y, int n_levels, float viewing_cos_limit, uint8_t *track_in_view, float *proj_x,
                             float *proj_y, float *proj_xr, int32_t *pred_level, float *view_cos)
{
    if (!m || n_levels <= 0) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if (!p || p->n_pts < 0 || (p->n_pts > 0 && (!p->pos || !p->max_dist || !p->min_dist || !p->normal)) ||
        !(p->log_scale_factor > 0)) {
        set_error("bad point set");
        return AOS2_ERR_ARG;
    }
    int st;
    if (p->n_pts == 0) return AOS2_OK;
    if (!track_in_view || !proj_x || !proj_y || !proj_xr || !pred_level || !view_cos) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    if ((st = matcher_init(m))) return st;
    Arena A{m};
    const size_t n = (size_t)p->n_pts;
    ProjGenDev P{};
    {
        P.n_pts = p->n_pts;
        P.mode = 5;
        const size_t o2 = A.push(p->pos, n * 12), o3 = A.push(p->max_dist, n * 4), o4 = A.push(p->min_dist, n * 4),
                     o5 = A.push(p->normal, n * 12);
        P.pos = reinterpret_cast<const float *>(o2); P.max_dist = reinterpret_cast<const float *>(o3);
        P.min_dist = reinterpret_cast<const float *>(o4); P.normal = reinterpret_cast<const float *>(o5);
        memcpy(P.R, p->R, sizeof(P.R)); memcpy(P.t, p->t, sizeof(P.t)); memcpy(P.Ow, p->Ow, sizeof(P.Ow));
        P.fx = p->fx; P.fy = p->fy; P.cx = p->cx; P.cy = p->cy; P.bf = p->bf;
        P.log_scale_factor = p->log_scale_factor;
    }
    const size_t oo = A.reserve(n * 24 + 64);
    if ((st = A.upload())) return st;
    uint8_t *base = m->arena.p;
    P.pos = reinterpret_cast<const float *>(base + reinterpret_cast<size_t>(P.pos));
    P.max_dist = reinterpret_cast<const float *>(base + reinterpret_cast<size_t>(P.max_dist));
    P.min_dist = reinterpret_cast<const float *>(base + reinterpret_cast<size_t>(P.min_dist));
    P.normal = reinterpret_cast<const float *>(base + reinterpret_cast<size_t>(P.normal));
    float *d_px = A.dev<float>(oo), *d_py = d_px + n, *d_pr = d_py + n, *d_vc = d_pr + n;
    int32_t *d_lv = reinterpret_cast<int32_t *>(d_vc + n);
    uint8_t *d_iv = reinterpret_cast<uint8_t *>(d_lv + n);
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    hipLaunchKernelGGL(is_in_frustum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, P, min_x, max_x, min_y,
                       max_y, n_levels, viewing_cos_limit, d_iv, d_px, d_py, d_pr, d_lv, d_vc);
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    A.fetch_dev(proj_x, d_px, n * 4);
    A.fetch_dev(proj_y, d_py, n * 4);
    A.fetch_dev(proj_xr, d_pr, n * 4);
    A.fetch_dev(view_cos, d_vc, n * 4);
    A.fetch_dev(pred_level, d_lv, n * 4);
    A.fetch_dev(track_in_view, d_iv, n);
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

// greedy modes 2 (SearchByProjection(pKF, Scw, ...)) and 4 (SearchByProjection(CurrentFrame, pKF, ...))
static int projgen_serial(aos2_matcher_t *m, const aos2_frame_view_t *f, const aos2_proj_points_t *p, int mode, int thr,
                          int check_ori, int32_t *match_f, int32_t *nmatches)
{
    if (!m || !match_f || !nmatches) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = check_frame(f);
    if (st) return st;
    if ((st = check_points(p, mode))) return st;
    if ((st = matcher_init(m))) return st;
    Arena A{m};
    size_t fo[12];
    fill_frame(A, f, fo);
    ProjGenDev P = points_dev(A, p, mode, f->n_levels);
    const size_t om = A.reserve((size_t)f->n_f * 4 + 4), ob = A.reserve((size_t)f->n_f * 4 + 4), on = A.reserve(8);
    const size_t oslots = A.reserve((size_t)(p->n_pts + 1) * sizeof(QuerySlot));
    const size_t ochoice = A.reserve((size_t)(p->n_pts + 1) * 4);
    const size_t pool_cap = (size_t)p->n_pts * (size_t)f->n_f;
    // one slice of n_f entries per point, so that no window can overflow its slice (entry offsets are 32-bit: 2^31 entries =
    // 16 GiB of the device's 288 GB; beyond that AOS2_ERR_CAPACITY, below it only the allocation itself can fail)
    if (pool_cap > ((size_t)1 << 31) - 2) {
        set_error("projection search of %d points x %d features: %zu candidate entries exceed the pool's index range", p->n_pts, f->n_f, pool_cap);
        return AOS2_ERR_CAPACITY;
    }
    if ((st = m->pool.alloc(pool_cap + 1))) return st;
    if ((st = A.upload())) return st;
    points_resolve(A, P);
    FrameDev F = frame_dev(A, f, fo);
    int32_t *d_used = A.dev<int32_t>(on) + 1;
    AOS2_HIP_CHECK(hipMemsetAsync(d_used, 0, 4, m->stream));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    if (p->n_pts > 0)
        hipLaunchKernelGGL(projgen_entries_kernel, dim3(p->n_pts), dim3(64), 0, m->stream, F, P, A.dev<QuerySlot>(oslots),
                           reinterpret_cast<Entry *>(m->pool.p), d_used, (int)pool_cap);
    if ((size_t)f->n_f * 8 <= kFixLdsBytes && !m->serial_resolve)
        hipLaunchKernelGGL(projgen_resolve_fix_kernel, dim3(1), dim3(1024), (size_t)f->n_f * 8 + 16, m->stream, F, P, thr, check_ori,
                       A.dev<QuerySlot>(oslots), reinterpret_cast<const Entry *>(m->pool.p), A.dev<int32_t>(om),
                       A.dev<int32_t>(ob), A.dev<int32_t>(on), A.dev<int32_t>(ochoice));
    else
        hipLaunchKernelGGL(projgen_resolve_kernel, dim3(1), dim3(64), (size_t)f->n_f + 16, m->stream, F, P, thr, check_ori,
                       A.dev<QuerySlot>(oslots), reinterpret_cast<const Entry *>(m->pool.p), A.dev<int32_t>(om),
                       A.dev<int32_t>(ob), A.dev<int32_t>(on));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    if (f->n_f > 0)
        A.fetch_dev(match_f, A.dev<int32_t>(om), (size_t)f->n_f * 4);
    A.fetch_dev(nmatches, A.dev<int32_t>(on), 4);
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

int aos2_matcher_search_by_projection_kf(aos2_matcher_t *m, const aos2_frame_view_t *kf, const aos2_proj_points_t *p,
                                         int32_t *match_f, int32_t *nmatches)
{
    return projgen_serial(m, kf, p, 2, TH_LOW, 0, match_f, nmatches);
}

int aos2_matcher_search_by_projection_reloc(aos2_matcher_t *m, const aos2_frame_view_t *frame, const aos2_proj_points_t *p,
                                            int orb_dist, int32_t *match_f, int32_t *nmatches)
{
    if (!m) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    return projgen_serial(m, frame, p, 4, orb_dist, m->check_ori, match_f, nmatches);
}

int aos2_matcher_search_for_initialization(aos2_matcher_t *m, const aos2_frame_view_t *f2, int n1, const uint8_t *desc1,
                                           const int32_t *octave1, const float *angle1, const float *prev_xy,
                                           int window_size, int32_t *match12, int32_t *nmatches)
{
    if (!m || n1 < 0 || !nmatches || (n1 > 0 && (!desc1 || !octave1 || !angle1 || !prev_xy || !match12))) {
        set_error("bad argument");
        return AOS2_ERR_ARG;
    }
    int st = check_frame(f2);
    if (st) return st;
    *nmatches = 0;
    if (n1 == 0) return AOS2_OK;
    if (n1 >= 65535 || f2->n_f > 15000) {
        set_error("SearchForInitialization: %d x %d features exceed the LDS match tables (65534 / 15000)", n1, f2->n_f);
        return AOS2_ERR_CAPACITY;
    }
    if ((st = matcher_init(m))) return st;
    Arena A{m};
    size_t fo[12];
    fill_frame(A, f2, fo);
    const size_t n = (size_t)n1;
    const size_t o0 = A.push(desc1, n * 32), o1 = A.push(octave1, n * 4), o2 = A.push(angle1, n * 4), o3 = A.push(prev_xy, n * 8);
    const size_t om = A.reserve(n * 8 + 8), on = A.reserve(8);
    const size_t oslots = A.reserve((n + 1) * sizeof(QuerySlot));
    const size_t pool_cap = n * (size_t)f2->n_f;
    if (pool_cap > ((size_t)1 << 31) - 2) {   // (32-bit entry offsets: 16 GiB of the device's 288 GB)
        set_error("initialization search of %d x %d features exceeds the pool's index range", n1, f2->n_f);
        return AOS2_ERR_CAPACITY;
    }
    if ((st = m->pool.alloc(pool_cap + 1))) return st;
    if ((st = A.upload())) return st;
    FrameDev F = frame_dev(A, f2, fo);
    InitDev P{};
    P.n1 = n1;
    P.desc1 = A.dev<uint8_t>(o0); P.octave1 = A.dev<int32_t>(o1); P.angle1 = A.dev<float>(o2); P.prev_xy = A.dev<float>(o3);
    P.window = (float)window_size;
    int32_t *d_used = A.dev<int32_t>(on) + 1;
    AOS2_HIP_CHECK(hipMemsetAsync(d_used, 0, 4, m->stream));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[0], m->stream));
    hipLaunchKernelGGL(init_entries_kernel, dim3(n1), dim3(64), 0, m->stream, F, P, A.dev<QuerySlot>(oslots),
                       reinterpret_cast<Entry *>(m->pool.p), d_used, (int)pool_cap);
    hipLaunchKernelGGL(init_resolve_kernel, dim3(1), dim3(64), (size_t)f2->n_f * 4 + 16, m->stream, F, P, m->nnratio,
                       m->check_ori, A.dev<QuerySlot>(oslots), reinterpret_cast<const Entry *>(m->pool.p), A.dev<int32_t>(om),
                       A.dev<int32_t>(om) + n1, A.dev<int32_t>(on));
    AOS2_HIP_CHECK(hipEventRecord(m->ev[1], m->stream));
    A.fetch_dev(match12, A.dev<int32_t>(om), n * 4);
    A.fetch_dev(nmatches, A.dev<int32_t>(on), 4);
    if ((st = A.finish())) return st;
    (void)hipEventElapsedTime(&m->last_ms, m->ev[0], m->ev[1]);
    return AOS2_OK;
}

}  // extern "C"

#include "frames_impl.inc"
