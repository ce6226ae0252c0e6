// Device-side descriptors and kernel launchers of the ORB extractor (extractor_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aos2.h"

namespace aos2 {

struct OctNode;

struct LevelDev {
    int w, h, pitch;      // interior size, row pitch (multiple of 16) of the plane
    size_t off;           // byte offset of the plane inside one image's pyramid block
    int tab_x, tab_y;     // offsets of this level's resize tables (destination side)
    int nfeat;            // mnFeaturesPerLevel[level]
    int scaled_patch;     // (int)(31 * scale[level])
    float scale;          // mvScaleFactor[level]
    // octree scratch of this level inside one image's scratch block (sized for the theoretical
    // maximum of NMS survivors, so no input can overflow it)
    int oct_cand_off, oct_cand_cap;   // in candidates
    int oct_node_off, oct_node_cap;   // in OctNode
};

struct CellDev {
    int16_t level;
    int16_t vx0, vy0;     // first evaluated pixel of the cell (level interior coordinates)
    int16_t cw, ch;       // evaluated columns / rows
    uint16_t pitch;       // row pitch of the level's plane (levels >= 1; level 0 is the caller's image with the caller's pitch)
    int32_t slot_off;     // offset of the cell's candidate slots inside one image's slot block
    uint32_t inv_ndw;     // 65536 / (quads per row + 2) + 1: exact i / ndw for i < 4096 by mul-shift
    uint32_t inv_nq;      // 65536 / quads per row + 1
    uint32_t plane_off;   // byte offset of the level's plane inside one image's pyramid block (LevelDev::off): the FAST kernel
                          // needs nothing else of the level, so its first loads are ONE record instead of a cell -> level chain
    uint32_t inv_n16;     // 65536 / (16-byte groups per tile row) + 1
};
static_assert(sizeof(CellDev) == 32, "the FAST kernel loads a cell record as eight dwords");

// inputs of the octree jobs' candidate gather (FAST's per-cell slots) and the per-(image, level) candidate count
struct OctGather {
    const CellDev *cells;
    const int *level_cell_begin;   // [n_levels]
    const uint32_t *slots;         // [batch][slot_stride]
    size_t slot_stride;
    const int32_t *cell_cnt;       // [batch][n_cells]
    int n_cells;
    int32_t *level_cnt;            // out [batch][n_levels]
};

struct OctDevScratch {
    int16_t *xs, *ys;
    uint8_t *sc;
    int32_t *perm, *tmp, *pairs, *out_idx;
    OctNode *nodes;
    size_t cand_stride, node_stride;  // per image
};

int upload_constants(const int8_t *pattern, const int *umax, const int *gauss7, hipStream_t st);
void launch_resize(const uint8_t *src_base, size_t src_img_stride, int src_pitch, uint8_t *pyr, size_t pyr_stride,
                   const LevelDev &src, const LevelDev &dst, const int *xofs, const int *xab, const int *yofs,
                   const int *yab, int batch, hipStream_t st);
void launch_pyramid_fused(const uint8_t *img0, size_t img0_stride, int pitch0, uint8_t *pyr, size_t pyr_stride,
                          const LevelDev *levels, int nlevels, const int4 *tile_x, const int4 *tile_y, int ntx, int nty,
                          const int *xofs, const int *xab, const int *yofs, const int *yab, int buf_pitch, int buf_rows,
                          size_t lds_bytes, int batch, hipStream_t st);
void launch_fast(const uint8_t *img0, size_t img0_stride, int pitch0, const uint8_t *pyr, size_t pyr_stride,
                 const LevelDev *levels, const CellDev *cells, int n_cells,
                 int ini_th, int min_th, int TP, int TH, int SP, size_t lds_bytes, int list_cap, int keep_cap,
                 uint32_t *slots, size_t slot_stride, int32_t *cell_cnt, int batch, hipStream_t st);
void launch_compact(const CellDev *cells, int n_cells, int n_levels, const int *level_cell_begin,
                    const uint32_t *slots, size_t slot_stride, const int32_t *cell_cnt, uint32_t *dense,
                    size_t dense_stride, int32_t *level_off, int batch, hipStream_t st);
// LDS working set of one octree job with n candidates and target N (OctCompact layout in
// octree_kernel: packed candidates | perm | tmp | 16-byte node arena | two pairs arrays)
__host__ __device__ inline int oct_lds_nodes(int N) { return 2 * N + 96; }   // ~1.4 N nodes are created in practice
__host__ __device__ inline int oct_lds_pairs(int N) { return N + 32; }
__host__ __device__ inline size_t oct_lds_bytes(int n, int N)
{
    const size_t ncand = (size_t)((n + 3) & ~3);
    return ncand * 8 + (size_t)oct_lds_nodes(N) * 16 + (size_t)oct_lds_pairs(N) * 16;
}

// LDS slices of the one-workgroup-per-image octree kernel (one wave per level)
struct OctImageLayout {
    int off[16], bytes[16];
    int total;
};
int prepare_octree_image_kernel(int total_lds);
// Two levels per workgroup (level g and level n_levels - 1 - g: the largest with the smallest), a wave and an LDS slice of its own size
// each: OctImageLayout::off[l] is level l's offset INSIDE its workgroup, total the largest pair.  The per-job kernel reserves the
// level-0 size for every level; the pairs hold 1.3-1.6 x as many jobs per compute unit in workgroups no larger than two level-0 jobs.
int prepare_octree_pair_kernel(int total_lds);
void launch_octree_pairs(uint32_t *dense, size_t dense_stride, const OctGather &gather, const LevelDev *levels,
                         int n_levels, int batch, const OctDevScratch &scr, uint32_t *sel, size_t sel_stride,
                         int32_t *sel_level_cnt, int cap_level, const OctImageLayout &lay, hipStream_t st);
void launch_octree_image(uint32_t *dense, size_t dense_stride, const OctGather &gather, const LevelDev *levels,
                         int n_levels, int batch, const OctDevScratch &scr, uint32_t *sel, size_t sel_stride,
                         int32_t *sel_level_cnt, int cap_level, const OctImageLayout &lay, hipStream_t st);
void launch_octree(uint32_t *dense, size_t dense_stride, const OctGather &gather, const LevelDev *levels,
                   int n_levels, int batch, const OctDevScratch &scr, uint32_t *sel, size_t sel_stride,
                   int32_t *sel_level_cnt, int cap_level, int lds_bytes, hipStream_t st);
void launch_describe(const uint8_t *img0, size_t img0_stride, int pitch0, const uint8_t *pyr, size_t pyr_stride,
                     const LevelDev *levels, int n_levels,
                     const uint32_t *sel, size_t sel_stride, int cap_level, const int32_t *sel_level_cnt,
                     aos2_keypoint_t *kps, uint8_t *desc, int cap, int32_t *n_out, int batch,
                     unsigned long long umax_nibbles, int32_t *status, hipStream_t st);

// whole-level blur (AOS2_DESC_BLUR=level): plan of the blurred planes, the streaming blur, describe on blurred levels
struct BlurPlanHost {
    int first[9];
    int nq[8];
    int nq_in[8];
    uint32_t dst_off[8];
    int dst_pitch[8];
};
size_t blur_plan(const LevelDev *h_levels, int n_levels, size_t pyr_bytes, BlurPlanHost *out);   // returns bytes per image
void launch_blur_levels(const uint8_t *img0, size_t img0_stride, int pitch0, const uint8_t *pyr, size_t pyr_stride,
                        const LevelDev *levels, int n_levels, const BlurPlanHost &plan, uint8_t *blur, size_t blur_stride, int batch,
                        hipStream_t st);
void launch_describe_blur(const uint8_t *img0, size_t img0_stride, int pitch0, const uint8_t *pyr, size_t pyr_stride,
                          const uint8_t *blur, size_t blur_stride, const BlurPlanHost &plan, const LevelDev *levels, int n_levels,
                          const uint32_t *sel, size_t sel_stride, int cap_level, const int32_t *sel_level_cnt,
                          aos2_keypoint_t *kps, uint8_t *desc, int cap, int32_t *n_out, int batch, int32_t *status, hipStream_t st);

}  // namespace aos2
