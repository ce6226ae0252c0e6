/*
 * ORACLE (test infrastructure only; PARITY UNPINNED, see orb_oracle.h).
 *
 * CPU restatement of the ORBmatcher hot-path methods (reference src/ORBmatcher.cc) over plain
 * SoA snapshots of the Frame/KeyFrame/MapPoint fields they read.
 * Build with -ffp-contract=off.
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TH_HIGH 100      /* src/ORBmatcher.cc:37 */
#define TH_LOW 50        /* :38 */
#define HISTO_LENGTH 30  /* :39 */
#define FRAME_GRID_ROWS 48 /* include/Frame.h:37 */
#define FRAME_GRID_COLS 64 /* include/Frame.h:38 */

/* DescriptorDistance :1647-1663 */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* rotation histogram as growable int vectors */
typedef struct { int *v; int n, cap; } ivec_t;
static void iv_push(ivec_t *a, int x)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 64;
        a->v = (int *)realloc(a->v, sizeof(int) * a->cap);
    }
    a->v[a->n++] = x;
}

/* ComputeThreeMaxima :1601-1642 */
static void compute_three_maxima(const ivec_t *histo, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i].n;
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s;
            *ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        *ind2 = -1;
        *ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        *ind3 = -1;
    }
}

static int rot_bin(float rot)
{
    const float factor = 1.0f / HISTO_LENGTH; /* quirk kept: :172 */
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) :159-288 */
int orc_search_by_bow(const orc_bow_problem_t *p, int32_t *match_f)
{
    for (int i = 0; i < p->n_f; ++i) match_f[i] = -1;
    int nmatches = 0;
    ivec_t rotHist[HISTO_LENGTH];
    memset(rotHist, 0, sizeof(rotHist));
    int ik = 0, jf = 0;
    while (ik < p->n_nodes_kf && jf < p->n_nodes_f) {
        int idk = p->node_id_kf[ik], idf = p->node_id_f[jf];
        if (idk == idf) {
            for (int a = p->node_off_kf[ik]; a < p->node_off_kf[ik + 1]; ++a) {
                const int realIdxKF = p->node_idx_kf[a];
                if (!p->kf_has_mp[realIdxKF]) continue;
                const uint8_t *dKF = p->desc_kf + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int b = p->node_off_f[jf]; b < p->node_off_f[jf + 1]; ++b) {
                    const int realIdxF = p->node_idx_f[b];
                    if (match_f[realIdxF] >= 0) continue;
                    const int dist = orc_descriptor_distance(dKF, p->desc_f + (size_t)realIdxF * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdxF = realIdxF;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < p->nnratio * (float)bestDist2) {
                        match_f[bestIdxF] = realIdxKF;
                        if (p->check_orientation) {
                            float rot = p->angle_kf[realIdxKF] - p->angle_f[bestIdxF];
                            iv_push(&rotHist[rot_bin(rot)], bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            ik++;
            jf++;
        } else if (idk < idf) {
            while (ik < p->n_nodes_kf && p->node_id_kf[ik] < idf) ik++; /* lower_bound */
        } else {
            while (jf < p->n_nodes_f && p->node_id_f[jf] < idk) jf++;
        }
    }
    if (p->check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotHist[i].n; j++) {
                match_f[rotHist[i].v[j]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i].v);
    return nmatches;
}

/* Frame::GetFeaturesInArea src/Frame.cc:356-409 ; returns count into out */
static int features_in_area(const orc_frame_view_t *f, float x, float y, float r, int minLevel,
                            int maxLevel, int *out)
{
    int n = 0;
    const int nMinCellX0 = (int)floorf((x - f->min_x - r) * f->grid_w_inv);
    const int nMinCellX = nMinCellX0 > 0 ? nMinCellX0 : 0;
    if (nMinCellX >= FRAME_GRID_COLS) return 0;
    const int nMaxCellX0 = (int)ceilf((x - f->min_x + r) * f->grid_w_inv);
    const int nMaxCellX = nMaxCellX0 < FRAME_GRID_COLS - 1 ? nMaxCellX0 : FRAME_GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    const int nMinCellY0 = (int)floorf((y - f->min_y - r) * f->grid_h_inv);
    const int nMinCellY = nMinCellY0 > 0 ? nMinCellY0 : 0;
    if (nMinCellY >= FRAME_GRID_ROWS) return 0;
    const int nMaxCellY0 = (int)ceilf((y - f->min_y + r) * f->grid_h_inv);
    const int nMaxCellY = nMaxCellY0 < FRAME_GRID_ROWS - 1 ? nMaxCellY0 : FRAME_GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            int c = ix * FRAME_GRID_ROWS + iy;
            for (int j = f->grid_off[c]; j < f->grid_off[c + 1]; ++j) {
                int idx = f->grid_idx[j];
                if (bCheckLevels) {
                    if (f->kp_octave[idx] < minLevel) continue;
                    if (maxLevel >= 0)
                        if (f->kp_octave[idx] > maxLevel) continue;
                }
                const float distx = f->kp_x[idx] - x;
                const float disty = f->kp_y[idx] - y;
                if (fabsf(distx) < r && fabsf(disty) < r) out[n++] = idx;
            }
        }
    }
    return n;
}

/* SearchByProjection(Frame&, const vector<MapPoint*>&, th) :45-129 */
int orc_search_by_projection_mp(const orc_frame_view_t *f, const orc_proj_mp_problem_t *p,
                                int32_t *match_f)
{
    int nmatches = 0;
    const int bFactor = p->th != 1.0;
    uint8_t *state = (uint8_t *)malloc(f->n_f ? f->n_f : 1);
    memcpy(state, f->f_mp_state, f->n_f);
    for (int i = 0; i < f->n_f; ++i) match_f[i] = -1;
    int *vIndices = (int *)malloc(sizeof(int) * (f->n_f ? f->n_f : 1));
    for (int iMP = 0; iMP < p->n_mp; iMP++) {
        if (!p->track_in_view[iMP]) continue;
        const int nPredictedLevel = p->pred_level[iMP];
        float r = p->view_cos[iMP] > 0.998 ? 2.5f : 4.0f; /* RadiusByViewingCos :131-137 */
        if (bFactor) r *= p->th;
        const int nInd = features_in_area(f, p->proj_x[iMP], p->proj_y[iMP],
                                          r * f->scale_factors[nPredictedLevel],
                                          nPredictedLevel - 1, nPredictedLevel, vIndices);
        if (nInd == 0) continue;
        const uint8_t *MPdescriptor = p->desc + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nInd; ++k) {
            const int idx = vIndices[k];
            if (state[idx] == 2) continue;
            if (f->u_right[idx] > 0) {
                const float er = fabsf(p->proj_xr[iMP] - f->u_right[idx]);
                if (er > r * f->scale_factors[nPredictedLevel]) continue;
            }
            const int dist = orc_descriptor_distance(MPdescriptor, f->desc_f + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = f->kp_octave[idx];
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = f->kp_octave[idx];
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > p->nnratio * bestDist2) continue;
            match_f[bestIdx] = iMP;
            state[bestIdx] = p->has_obs[iMP] ? 2 : 1;
            nmatches++;
        }
    }
    free(state);
    free(vIndices);
    return nmatches;
}

/* SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) :1328-1470 */
int orc_search_by_projection_last(const orc_frame_view_t *cur, const orc_proj_last_problem_t *p,
                                  int32_t *match_f)
{
    int nmatches = 0;
    ivec_t rotHist[HISTO_LENGTH];
    memset(rotHist, 0, sizeof(rotHist));
    const float *T = p->Tcw, *Tl = p->Tlw;
    /* twc = -Rcw^T * tcw ; tlc = Rlw*twc + tlw  (cv::Mat float arithmetic, :1339-1346) */
    float twc[3], tlc[3];
    for (int i = 0; i < 3; ++i) {
        /* cv::gemm general path (GEMM_1_T, alpha=-1): double accumulation, then cast */
        double s = (double)T[0 * 4 + i] * (double)T[3] + (double)T[1 * 4 + i] * (double)T[7] +
                   (double)T[2 * 4 + i] * (double)T[11];
        twc[i] = (float)(s * -1.0);
    }
    for (int i = 0; i < 3; ++i) {
        float s = Tl[i * 4 + 0] * twc[0] + Tl[i * 4 + 1] * twc[1] + Tl[i * 4 + 2] * twc[2];
        tlc[i] = s + Tl[i * 4 + 3];
    }
    const int bForward = tlc[2] > p->mb && !p->mono;
    const int bBackward = -tlc[2] > p->mb && !p->mono;

    uint8_t *state = (uint8_t *)malloc(cur->n_f ? cur->n_f : 1);
    memcpy(state, cur->f_mp_state, cur->n_f);
    for (int i = 0; i < cur->n_f; ++i) match_f[i] = -1;
    int *vIndices2 = (int *)malloc(sizeof(int) * (cur->n_f ? cur->n_f : 1));

    for (int i = 0; i < p->n_last; i++) {
        if (!p->last_valid[i]) continue;
        const float *X = p->world_pos + 3 * (size_t)i;
        float x3Dc[3];
        for (int r = 0; r < 3; ++r) {
            float s = T[r * 4 + 0] * X[0] + T[r * 4 + 1] * X[1] + T[r * 4 + 2] * X[2];
            x3Dc[r] = s + T[r * 4 + 3];
        }
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        float u = p->fx * xc * invzc + p->cx;
        float v = p->fy * yc * invzc + p->cy;
        if (u < cur->min_x || u > cur->max_x) continue;
        if (v < cur->min_y || v > cur->max_y) continue;
        int nLastOctave = p->last_octave[i];
        float radius = p->th * cur->scale_factors[nLastOctave];
        int nInd;
        if (bForward)
            nInd = features_in_area(cur, u, v, radius, nLastOctave, -1, vIndices2);
        else if (bBackward)
            nInd = features_in_area(cur, u, v, radius, 0, nLastOctave, vIndices2);
        else
            nInd = features_in_area(cur, u, v, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
        if (nInd == 0) continue;
        const uint8_t *dMP = p->desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < nInd; ++k) {
            const int i2 = vIndices2[k];
            if (state[i2] == 2) continue;
            if (cur->u_right[i2] > 0) {
                const float ur = u - p->mbf * invzc;
                const float er = fabsf(ur - cur->u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_descriptor_distance(dMP, cur->desc_f + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= TH_HIGH) {
            match_f[bestIdx2] = i;
            state[bestIdx2] = p->has_obs[i] ? 2 : 1;
            nmatches++;
            if (p->check_orientation) {
                float rot = p->last_angle[i] - cur->kp_angle[bestIdx2];
                iv_push(&rotHist[rot_bin(rot)], bestIdx2);
            }
        }
    }
    if (p->check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (int j = 0; j < rotHist[i].n; j++) {
                    match_f[rotHist[i].v[j]] = -2; /* set to NULL by the cull (:1459) */
                    nmatches--;
                }
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i].v);
    free(state);
    free(vIndices2);
    return nmatches;
}

/* ------------------------------------------------------------------ Frame::ComputeStereoMatches
 * src/Frame.cc:495-669.  Pyramid levels are passed as interior planes
 * (the reference reads mvImagePyramid[level] ROIs; the accessed windows never leave the interior,
 * see DESIGN.md).  cv::norm(IL, IR, NORM_L1) of integer-valued float patches is an exact integer. */
int orc_compute_stereo_matches(const orc_stereo_problem_t *p, float *u_right, float *depth)
{
    const int N = p->n_left, Nr = p->n_right;
    for (int i = 0; i < N; ++i) u_right[i] = depth[i] = -1.0f;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = p->level_h[0];
    /* row table :505-523 */
    ivec_t *rows = (ivec_t *)calloc(nRows, sizeof(ivec_t));
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = p->kp_right[iR].y;
        const float r = 2.0f * p->scale_factors[p->kp_right[iR].octave];
        const int maxr = (int)ceilf(kpY + r);
        const int minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) iv_push(&rows[yi], iR); /* the reference has no guard; keypoints keep yi inside */
    }
    const float minZ = p->mb, minD = 0, maxD = p->mbf / minZ;
    int *vdist = (int *)malloc(sizeof(int) * (N ? N : 1)), *vidx = (int *)malloc(sizeof(int) * (N ? N : 1));
    int nv = 0;
    for (int iL = 0; iL < N; iL++) {
        const orc_keypoint_t *kpL = &p->kp_left[iL];
        const int levelL = kpL->octave;
        const float vL = kpL->y, uL = kpL->x;
        const int row = (int)vL;
        if (row < 0 || row >= nRows) continue;
        const ivec_t *cand = &rows[row];
        if (cand->n == 0) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        int bestIdxR = 0;
        const uint8_t *dL = p->desc_left + (size_t)iL * 32;
        for (int iC = 0; iC < cand->n; iC++) {
            const int iR = cand->v[iC];
            const orc_keypoint_t *kpR = &p->kp_right[iR];
            if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1) continue;
            const float uR = kpR->x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(dL, p->desc_right + (size_t)iR * 32);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = iR;
                }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = p->kp_right[bestIdxR].x;
            const float scaleFactor = p->inv_scale_factors[kpL->octave];
            const float scaleduL = roundf(kpL->x * scaleFactor);
            const float scaledvL = roundf(kpL->y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const uint8_t *PL = p->left_planes[kpL->octave], *PR = p->right_planes[kpL->octave];
            const int pitchL = p->left_pitch[kpL->octave], pitchR = p->right_pitch[kpL->octave];
            const int cols = p->level_w[kpL->octave];
            const int cuL = (int)scaleduL, cvL = (int)scaledvL, cuR0 = (int)scaleduR0;
            float IL[11][11];
            const float cL = (float)PL[(size_t)cvL * pitchL + cuL];
            for (int dy = -w; dy <= w; ++dy)
                for (int dx = -w; dx <= w; ++dx) IL[dy + w][dx + w] = (float)PL[(size_t)(cvL + dy) * pitchL + cuL + dx] - cL;
            int bestDistS = 2147483647, bestincR = 0;
            float vDists[11];
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= cols) continue;
            for (int incR = -L; incR <= +L; incR++) {
                const float cR = (float)PR[(size_t)cvL * pitchR + cuR0 + incR];
                double acc = 0;
                for (int dy = -w; dy <= w; ++dy)
                    for (int dx = -w; dx <= w; ++dx) {
                        const float ir = (float)PR[(size_t)(cvL + dy) * pitchR + cuR0 + incR + dx] - cR;
                        acc += fabs((double)IL[dy + w][dx + w] - (double)ir);
                    }
                const float dist = (float)acc;
                if (dist < bestDistS) {
                    bestDistS = (int)dist;
                    bestincR = incR;
                }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = p->scale_factors[kpL->octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01f;
                    bestuR = uL - 0.01f;
                }
                depth[iL] = p->mbf / disparity;
                u_right[iL] = bestuR;
                vdist[nv] = bestDistS;
                vidx[nv] = iL;
                nv++;
            }
        }
    }
    if (nv > 0) { /* :654-668 (the reference reads vDistIdx[0] of an empty vector when nothing matched) */
        /* sort (dist, iL) ascending -> median = element nv/2 */
        int *order = (int *)malloc(sizeof(int) * nv);
        for (int i = 0; i < nv; ++i) order[i] = i;
        for (int i = 1; i < nv; ++i) { /* insertion sort, stable on (dist, iL) since iL ascends */
            int o = order[i], j = i - 1;
            while (j >= 0 && (vdist[order[j]] > vdist[o] || (vdist[order[j]] == vdist[o] && vidx[order[j]] > vidx[o]))) {
                order[j + 1] = order[j];
                --j;
            }
            order[j + 1] = o;
        }
        const float median = (float)vdist[order[nv / 2]];
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nv - 1; i >= 0; i--) {
            if ((float)vdist[order[i]] < thDist) break;
            u_right[vidx[order[i]]] = -1;
            depth[vidx[order[i]]] = -1;
        }
        free(order);
    }
    for (int i = 0; i < nRows; ++i) free(rows[i].v);
    free(rows);
    free(vdist);
    free(vidx);
    return nv;
}

/* ------------------------------------------------------------------ SURVEY §8(f) rank 4 */
/* SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) :522-655.
 * match12[n1] = index of the KF2 feature whose MapPoint is assigned to KF1 feature i, or -1. */
int orc_search_by_bow_kf(const orc_bow_kf_problem_t *p, int32_t *match12)
{
    for (int i = 0; i < p->n1; ++i) match12[i] = -1;
    uint8_t *matched2 = (uint8_t *)calloc(p->n2 ? p->n2 : 1, 1);
    int nmatches = 0;
    ivec_t rotHist[HISTO_LENGTH];
    memset(rotHist, 0, sizeof(rotHist));
    int i1n = 0, i2n = 0;
    while (i1n < p->n_nodes1 && i2n < p->n_nodes2) {
        const int id1 = p->node_id1[i1n], id2 = p->node_id2[i2n];
        if (id1 == id2) {
            for (int a = p->node_off1[i1n]; a < p->node_off1[i1n + 1]; ++a) {
                const int idx1 = p->node_idx1[a];
                if (!p->has_mp1[idx1]) continue;
                const uint8_t *d1 = p->desc1 + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int b = p->node_off2[i2n]; b < p->node_off2[i2n + 1]; ++b) {
                    const int idx2 = p->node_idx2[b];
                    if (matched2[idx2] || !p->has_mp2[idx2]) continue;
                    const int dist = orc_descriptor_distance(d1, p->desc2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 < TH_LOW) {
                    if ((float)bestDist1 < p->nnratio * (float)bestDist2) {
                        match12[idx1] = bestIdx2;
                        matched2[bestIdx2] = 1;
                        if (p->check_orientation) {
                            float rot = p->angle1[idx1] - p->angle2[bestIdx2];
                            iv_push(&rotHist[rot_bin(rot)], idx1);
                        }
                        nmatches++;
                    }
                }
            }
            i1n++;
            i2n++;
        } else if (id1 < id2) {
            while (i1n < p->n_nodes1 && p->node_id1[i1n] < id2) i1n++;
        } else {
            while (i2n < p->n_nodes2 && p->node_id2[i2n] < id1) i2n++;
        }
    }
    if (p->check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotHist[i].n; j++) {
                match12[rotHist[i].v[j]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i].v);
    free(matched2);
    return nmatches;
}

/* CheckDistEpipolarLine :139-157 */
static int check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float *F12, float sigma2)
{
    const float a = x1 * F12[0] + y1 * F12[3] + F12[6];
    const float b = x1 * F12[1] + y1 * F12[4] + F12[7];
    const float c = x1 * F12[2] + y1 * F12[5] + F12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2;
}

/* SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) :657-823.  ex, ey = the epipole of
 * :664-670 (host arithmetic on the two poses, taken as input).  match12[n1] = vMatches12; the pair list of
 * :811-820 is its non-negative entries in ascending i.  vbMatched2 is never set by the reference. */
int orc_search_for_triangulation(const orc_triang_problem_t *p, int32_t *match12)
{
    for (int i = 0; i < p->n1; ++i) match12[i] = -1;
    int nmatches = 0;
    ivec_t rotHist[HISTO_LENGTH];
    memset(rotHist, 0, sizeof(rotHist));
    int i1n = 0, i2n = 0;
    while (i1n < p->n_nodes1 && i2n < p->n_nodes2) {
        const int id1 = p->node_id1[i1n], id2 = p->node_id2[i2n];
        if (id1 == id2) {
            for (int a = p->node_off1[i1n]; a < p->node_off1[i1n + 1]; ++a) {
                const int idx1 = p->node_idx1[a];
                if (p->has_mp1[idx1]) continue; /* already a MapPoint */
                const int bStereo1 = p->u_right1[idx1] >= 0;
                if (p->only_stereo && !bStereo1) continue;
                const uint8_t *d1 = p->desc1 + (size_t)idx1 * 32;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int b = p->node_off2[i2n]; b < p->node_off2[i2n + 1]; ++b) {
                    const int idx2 = p->node_idx2[b];
                    if (p->has_mp2[idx2]) continue;
                    const int bStereo2 = p->u_right2[idx2] >= 0;
                    if (p->only_stereo && !bStereo2) continue;
                    const int dist = orc_descriptor_distance(d1, p->desc2 + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    if (!bStereo1 && !bStereo2) {
                        const float distex = p->ex - p->x2[idx2];
                        const float distey = p->ey - p->y2[idx2];
                        if (distex * distex + distey * distey < 100 * p->scale_factors2[p->octave2[idx2]]) continue;
                    }
                    if (check_dist_epipolar_line(p->x1[idx1], p->y1[idx1], p->x2[idx2], p->y2[idx2], p->F12,
                                                 p->level_sigma2_2[p->octave2[idx2]])) {
                        bestIdx2 = idx2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    nmatches++;
                    if (p->check_orientation) {
                        float rot = p->angle1[idx1] - p->angle2[bestIdx2];
                        iv_push(&rotHist[rot_bin(rot)], idx1);
                    }
                }
            }
            i1n++;
            i2n++;
        } else if (id1 < id2) {
            while (i1n < p->n_nodes1 && p->node_id1[i1n] < id2) i1n++;
        } else {
            while (i2n < p->n_nodes2 && p->node_id2[i2n] < id1) i2n++;
        }
    }
    if (p->check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotHist[i].n; j++) {
                match12[rotHist[i].v[j]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i].v);
    return nmatches;
}

/* MapPoint::ComputeDistinctiveDescriptors src/MapPoint.cc:275-340 for a batch of map points: the observed
 * descriptors of point p are desc[off[p] .. off[p+1]); best[p] = index (inside the point's list) of the
 * descriptor with the least median distance to the others, -1 for an empty list. */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
void orc_compute_distinctive_descriptors(int n_points, const int32_t *off, const uint8_t *desc, int32_t *best)
{
    for (int p = 0; p < n_points; ++p) {
        const int N = off[p + 1] - off[p];
        best[p] = -1;
        if (N <= 0) continue;
        const uint8_t *d = desc + (size_t)off[p] * 32;
        int *row = (int *)malloc(sizeof(int) * N);
        int BestMedian = 2147483647, BestIdx = 0;
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < N; ++j) row[j] = i == j ? 0 : orc_descriptor_distance(d + (size_t)i * 32, d + (size_t)j * 32);
            qsort(row, N, sizeof(int), cmp_int);
            const int median = row[(int)(0.5 * (N - 1))];
            if (median < BestMedian) {
                BestMedian = median;
                BestIdx = i;
            }
        }
        best[p] = BestIdx;
        free(row);
    }
}

/* ------------------------------------------------------------------ projection family (rank 4)
 * cv::Mat arithmetic conventions (OpenCV 3.2, non-IPP), same as the SearchByProjection restatement above:
 *   R*p + t          3x3 * 3x1 CV_32F, no transpose flag: the small-matrix path, float products summed left
 *                    to right in float, then a float add of t;
 *   cv::norm(v)      sqrt of a double sum of (double)v_i^2, returned as double, stored in a float;
 *   a.dot(b)         double sum of (double)a_i * (double)b_i;
 *   log(ratio)       std::log(float); convention 4: the correctly rounded float logarithm,
 *                    evaluated as (float)log((double)ratio). */
static void xform(const float *R, const float *t, const float *p, float *out)
{
    for (int r = 0; r < 3; ++r) {
        float s = R[r * 3 + 0] * p[0] + R[r * 3 + 1] * p[1] + R[r * 3 + 2] * p[2];
        out[r] = s + t[r];
    }
}
static float norm3(const float *v)
{
    double s = (double)v[0] * (double)v[0] + (double)v[1] * (double)v[1] + (double)v[2] * (double)v[2];
    return (float)sqrt(s);
}
/* MapPoint::PredictScale(currentDist, pKF / pF) src/MapPoint.cc:427-459 */
/* orc_set_log_mode(1): libm logf instead of the correctly rounded float logarithm -- for measuring the convention's effect
 * (tools/convention_effects.py); the parity tests never set it */
static int g_log_libm = 0;
void orc_set_log_mode(int use_libm) { g_log_libm = use_libm; }
static int predict_scale(float max_dist, float dist, float log_scale_factor, int n_levels)
{
    const float ratio = max_dist / dist;
    const float lg = g_log_libm ? logf(ratio) : (float)log((double)ratio);
    int nScale = (int)ceilf(lg / log_scale_factor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= n_levels) nScale = n_levels - 1;
    return nScale;
}

/* Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th) :825-975, the search part: for every map
 * point the KF feature it would be fused with (best_idx, -1 = none) and the distance.  valid[i] stands for
 * `pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF)`; what happens to the pair (:950-969) is host state. */
int orc_fuse(const orc_frame_view_t *f, const orc_proj_gen_t *p, int32_t *best_idx, int32_t *best_dist)
{
    int nFused = 0;
    int *vIndices = (int *)malloc(sizeof(int) * (f->n_f ? f->n_f : 1));
    for (int i = 0; i < p->n_pts; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
        if (!p->valid[i]) continue;
        const float *p3Dw = p->pos + 3 * (size_t)i;
        float p3Dc[3];
        xform(p->R, p->t, p3Dw, p3Dc);
        if (p3Dc[2] < 0.0f) continue;
        const float invz = 1 / p3Dc[2];
        const float x = p3Dc[0] * invz;
        const float y = p3Dc[1] * invz;
        const float u = p->fx * x + p->cx;
        const float v = p->fy * y + p->cy;
        if (!(u >= f->min_x && u < f->max_x && v >= f->min_y && v < f->max_y)) continue; /* KeyFrame::IsInImage */
        const float ur = u - p->bf * invz;
        float PO[3] = {p3Dw[0] - p->Ow[0], p3Dw[1] - p->Ow[1], p3Dw[2] - p->Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < 0.8f * p->min_dist[i] || dist3D > 1.2f * p->max_dist[i]) continue; /* Get{Min,Max}DistanceInvariance() src/MapPoint.cc:413-423 */
        const float *Pn = p->normal + 3 * (size_t)i;
        const double dot = (double)PO[0] * (double)Pn[0] + (double)PO[1] * (double)Pn[1] + (double)PO[2] * (double)Pn[2];
        if (dot < 0.5 * dist3D) continue;
        const int nPredictedLevel = predict_scale(p->max_dist[i], dist3D, p->log_scale_factor, f->n_levels);
        const float radius = p->th * f->scale_factors[nPredictedLevel];
        const int nInd = features_in_area(f, u, v, radius, -1, -1, vIndices);
        if (nInd == 0) continue;
        const uint8_t *dMP = p->desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < nInd; ++k) {
            const int idx = vIndices[k];
            const int kpLevel = f->kp_octave[idx];
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (f->u_right[idx] >= 0) {
                const float ex = u - f->kp_x[idx], ey = v - f->kp_y[idx], er = ur - f->u_right[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * p->inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - f->kp_x[idx], ey = v - f->kp_y[idx];
                const float e2 = ex * ex + ey * ey;
                if (e2 * p->inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = orc_descriptor_distance(dMP, f->desc_f + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        if (bestDist <= TH_LOW) {
            best_idx[i] = bestIdx;
            best_dist[i] = bestDist;
            nFused++;
        }
    }
    free(vIndices);
    return nFused;
}

/* shared body of Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) :977-1100 and
 * SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) :290-403.  R, t, Ow = Rcw, tcw, Ow of :986-991 /
 * :299-304 (host cv::Mat arithmetic on Scw, taken as input).  taken == NULL: the Fuse flavour (independent
 * points, best_idx per point); taken != NULL: the SearchByProjection flavour (vpMatched, updated in place,
 * match_f[idx] = point index). */
static int proj_sim3_kf(const orc_frame_view_t *f, const orc_proj_gen_t *p, int invz_double, uint8_t *taken,
                        int32_t *best_idx, int32_t *best_dist, int32_t *match_f)
{
    int n = 0;
    int *vIndices = (int *)malloc(sizeof(int) * (f->n_f ? f->n_f : 1));
    for (int i = 0; i < p->n_pts; i++) {
        if (best_idx) {
            best_idx[i] = -1;
            best_dist[i] = 256;
        }
        if (!p->valid[i]) continue;
        const float *p3Dw = p->pos + 3 * (size_t)i;
        float p3Dc[3];
        xform(p->R, p->t, p3Dw, p3Dc);
        if (p3Dc[2] < 0.0f) continue;
        const float invz = invz_double ? (float)(1.0 / p3Dc[2]) : 1 / p3Dc[2]; /* :1026 `1.0/z`, :329 `1/z` */
        const float x = p3Dc[0] * invz;
        const float y = p3Dc[1] * invz;
        const float u = p->fx * x + p->cx;
        const float v = p->fy * y + p->cy;
        if (!(u >= f->min_x && u < f->max_x && v >= f->min_y && v < f->max_y)) continue;
        float PO[3] = {p3Dw[0] - p->Ow[0], p3Dw[1] - p->Ow[1], p3Dw[2] - p->Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < 0.8f * p->min_dist[i] || dist3D > 1.2f * p->max_dist[i]) continue; /* Get{Min,Max}DistanceInvariance() src/MapPoint.cc:413-423 */
        const float *Pn = p->normal + 3 * (size_t)i;
        const double dot = (double)PO[0] * (double)Pn[0] + (double)PO[1] * (double)Pn[1] + (double)PO[2] * (double)Pn[2];
        if (dot < 0.5 * dist3D) continue;
        const int nPredictedLevel = predict_scale(p->max_dist[i], dist3D, p->log_scale_factor, f->n_levels);
        const float radius = p->th * f->scale_factors[nPredictedLevel];
        const int nInd = features_in_area(f, u, v, radius, -1, -1, vIndices);
        if (nInd == 0) continue;
        const uint8_t *dMP = p->desc + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1; /* :1063 starts at INT_MAX; no distance exceeds 256 */
        for (int k = 0; k < nInd; ++k) {
            const int idx = vIndices[k];
            if (taken && taken[idx]) continue; /* vpMatched[idx] :374 */
            const int kpLevel = f->kp_octave[idx];
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = orc_descriptor_distance(dMP, f->desc_f + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        if (bestDist <= TH_LOW) {
            if (taken) {
                taken[bestIdx] = 1;
                match_f[bestIdx] = i;
            } else {
                best_idx[i] = bestIdx;
                best_dist[i] = bestDist;
            }
            n++;
        }
    }
    free(vIndices);
    return n;
}
int orc_fuse_sim3(const orc_frame_view_t *f, const orc_proj_gen_t *p, int32_t *best_idx, int32_t *best_dist)
{
    return proj_sim3_kf(f, p, 1, NULL, best_idx, best_dist, NULL);
}
/* f->f_mp_state[idx] != 0 <=> vpMatched[idx] != NULL on entry; match_f[n_f] = index of the point newly
 * written to vpMatched[idx], or -1 */
int orc_search_by_projection_kf(const orc_frame_view_t *f, const orc_proj_gen_t *p, int32_t *match_f)
{
    uint8_t *taken = (uint8_t *)malloc(f->n_f ? f->n_f : 1);
    for (int i = 0; i < f->n_f; ++i) {
        taken[i] = f->f_mp_state[i] != 0;
        match_f[i] = -1;
    }
    const int n = proj_sim3_kf(f, p, 0, taken, NULL, NULL, match_f);
    free(taken);
    return n;
}

/* one direction of SearchBySim3 (:1148-1225 / :1228-1303): points of one keyframe into the other.
 * R, t = R1w, t1w (camera of the source keyframe from world); R2, t2 = sR21, t21 (or sR12, t12). */
static void sim3_direction(const orc_frame_view_t *f, const orc_proj_gen_t *p, int32_t *vnMatch)
{
    int *vIndices = (int *)malloc(sizeof(int) * (f->n_f ? f->n_f : 1));
    for (int i = 0; i < p->n_pts; i++) {
        vnMatch[i] = -1;
        if (!p->valid[i]) continue;
        float c1[3], c2[3];
        xform(p->R, p->t, p->pos + 3 * (size_t)i, c1);
        xform(p->R2, p->t2, c1, c2);
        if (c2[2] < 0.0f) continue;
        const float invz = (float)(1.0 / c2[2]);
        const float x = c2[0] * invz;
        const float y = c2[1] * invz;
        const float u = p->fx * x + p->cx;
        const float v = p->fy * y + p->cy;
        if (!(u >= f->min_x && u < f->max_x && v >= f->min_y && v < f->max_y)) continue;
        const float dist3D = norm3(c2);
        if (dist3D < 0.8f * p->min_dist[i] || dist3D > 1.2f * p->max_dist[i]) continue; /* Get{Min,Max}DistanceInvariance() src/MapPoint.cc:413-423 */
        const int nPredictedLevel = predict_scale(p->max_dist[i], dist3D, p->log_scale_factor, f->n_levels);
        const float radius = p->th * f->scale_factors[nPredictedLevel];
        const int nInd = features_in_area(f, u, v, radius, -1, -1, vIndices);
        if (nInd == 0) continue;
        const uint8_t *dMP = p->desc + (size_t)i * 32;
        int bestDist = 2147483647, bestIdx = -1;
        for (int k = 0; k < nInd; ++k) {
            const int idx = vIndices[k];
            if (f->kp_octave[idx] < nPredictedLevel - 1 || f->kp_octave[idx] > nPredictedLevel) continue;
            const int dist = orc_descriptor_distance(dMP, f->desc_f + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx;
    }
    free(vIndices);
}
/* SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) :1102-1326.  p12: the map points of KF1 (one per
 * KF1 feature, valid = pMP && !vbAlreadyMatched1 && !isBad) projected into KF2 (view f2); p21 the converse.
 * match12[n1] = index of the KF2 feature whose map point becomes vpMatches12[i1], or -1 (unchanged). */
int orc_search_by_sim3(const orc_frame_view_t *f1, const orc_frame_view_t *f2, const orc_proj_gen_t *p12,
                       const orc_proj_gen_t *p21, int32_t *match12)
{
    int32_t *vn1 = (int32_t *)malloc(sizeof(int32_t) * (p12->n_pts ? p12->n_pts : 1));
    int32_t *vn2 = (int32_t *)malloc(sizeof(int32_t) * (p21->n_pts ? p21->n_pts : 1));
    sim3_direction(f2, p12, vn1);
    sim3_direction(f1, p21, vn2);
    int nFound = 0;
    for (int i1 = 0; i1 < p12->n_pts; i1++) {
        match12[i1] = -1;
        const int idx2 = vn1[i1];
        if (idx2 >= 0) {
            const int idx1 = vn2[idx2];
            if (idx1 == i1) {
                match12[i1] = idx2;
                nFound++;
            }
        }
    }
    free(vn1);
    free(vn2);
    return nFound;
}

/* SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) :1472-1599.
 * points = pKF's map points (valid = pMP && !isBad && !sAlreadyFound.count), q_angle = pKF->mvKeysUn[i].angle;
 * f_mp_state[i2] != 0 <=> CurrentFrame.mvpMapPoints[i2] != NULL on entry.
 * match_f[n_f]: keyframe feature index newly assigned to frame feature i2, -1 unchanged, -2 reset to NULL by
 * the rotation cull (:1583-1594, which also clears a slot that was already non-NULL? no: only pushed i2). */
int orc_search_by_projection_reloc(const orc_frame_view_t *f, const orc_proj_gen_t *p, int orb_dist, int check_orientation,
                                   int32_t *match_f)
{
    int nmatches = 0;
    ivec_t rotHist[HISTO_LENGTH];
    memset(rotHist, 0, sizeof(rotHist));
    uint8_t *taken = (uint8_t *)malloc(f->n_f ? f->n_f : 1);
    for (int i = 0; i < f->n_f; ++i) {
        taken[i] = f->f_mp_state[i] != 0;
        match_f[i] = -1;
    }
    int *vIndices2 = (int *)malloc(sizeof(int) * (f->n_f ? f->n_f : 1));
    for (int i = 0; i < p->n_pts; i++) {
        if (!p->valid[i]) continue;
        const float *x3Dw = p->pos + 3 * (size_t)i;
        float x3Dc[3];
        xform(p->R, p->t, x3Dw, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        const float u = p->fx * xc * invzc + p->cx;
        const float v = p->fy * yc * invzc + p->cy;
        if (u < f->min_x || u > f->max_x) continue;
        if (v < f->min_y || v > f->max_y) continue;
        float PO[3] = {x3Dw[0] - p->Ow[0], x3Dw[1] - p->Ow[1], x3Dw[2] - p->Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < 0.8f * p->min_dist[i] || dist3D > 1.2f * p->max_dist[i]) continue; /* Get{Min,Max}DistanceInvariance() src/MapPoint.cc:413-423 */
        const int nPredictedLevel = predict_scale(p->max_dist[i], dist3D, p->log_scale_factor, f->n_levels);
        const float radius = p->th * f->scale_factors[nPredictedLevel];
        const int nInd = features_in_area(f, u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, vIndices2);
        if (nInd == 0) continue;
        const uint8_t *dMP = p->desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < nInd; ++k) {
            const int i2 = vIndices2[k];
            if (taken[i2]) continue;
            const int dist = orc_descriptor_distance(dMP, f->desc_f + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= orb_dist) {
            taken[bestIdx2] = 1;
            match_f[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = p->q_angle[i] - f->kp_angle[bestIdx2];
                iv_push(&rotHist[rot_bin(rot)], bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i != ind1 && i != ind2 && i != ind3) {
                for (int j = 0; j < rotHist[i].n; j++) {
                    match_f[rotHist[i].v[j]] = -2;
                    nmatches--;
                }
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i].v);
    free(taken);
    free(vIndices2);
    return nmatches;
}

/* SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, windowSize) :405-520.
 * f2 = view of F2; F1 enters through its arrays.  prev_xy (2 per F1 feature) = vbPrevMatched on entry; the
 * update of :512-515 is `prev_xy[i1] = f2 keypoint of match12[i1]` and is left to the caller.
 * match12[n1] = vnMatches12. */
int orc_search_for_initialization(const orc_frame_view_t *f2, int n1, const uint8_t *desc1, const int32_t *octave1,
                                  const float *angle1, const float *prev_xy, int window_size, float nnratio,
                                  int check_orientation, int32_t *match12)
{
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) match12[i] = -1;
    ivec_t rotHist[HISTO_LENGTH];
    memset(rotHist, 0, sizeof(rotHist));
    const int n2 = f2->n_f;
    int *vMatchedDistance = (int *)malloc(sizeof(int) * (n2 ? n2 : 1));
    int *vnMatches21 = (int *)malloc(sizeof(int) * (n2 ? n2 : 1));
    int *vIndices2 = (int *)malloc(sizeof(int) * (n2 ? n2 : 1));
    for (int i = 0; i < n2; ++i) {
        vMatchedDistance[i] = 2147483647;
        vnMatches21[i] = -1;
    }
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = octave1[i1];
        if (level1 > 0) continue;
        const int nInd = features_in_area(f2, prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)window_size, level1, level1, vIndices2);
        if (nInd == 0) continue;
        const uint8_t *d1 = desc1 + (size_t)i1 * 32;
        int bestDist = 2147483647, bestDist2 = 2147483647, bestIdx2 = -1;
        for (int k = 0; k < nInd; ++k) {
            const int i2 = vIndices2[k];
            const int dist = orc_descriptor_distance(d1, f2->desc_f + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    match12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                match12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    float rot = angle1[i1] - f2->kp_angle[bestIdx2];
                    iv_push(&rotHist[rot_bin(rot)], i1);
                }
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < rotHist[i].n; j++) {
                const int idx1 = rotHist[i].v[j];
                if (match12[idx1] >= 0) {
                    match12[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i].v);
    free(vMatchedDistance);
    free(vnMatches21);
    free(vIndices2);
    return nmatches;
}

/* Frame::isInFrustum(MapPoint*, viewingCosLimit) src/Frame.cc:298-354 for a batch of map points (the loop of
 * Tracking::SearchLocalPoints).  R, t, Ow = mRcw, mtcw, mOw; bf = mbf.  Outputs = the mbTrackInView /
 * mTrackProj* / mnTrackScaleLevel / mTrackViewCos members the search reads (SURVEY a12 inputs). */
void orc_is_in_frustum(const orc_proj_gen_t *p, float min_x, float max_x, float min_y, float max_y, int n_levels,
                       float viewing_cos_limit, uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr,
                       int32_t *pred_level, float *view_cos)
{
    for (int i = 0; i < p->n_pts; ++i) {
        in_view[i] = 0;
        proj_x[i] = proj_y[i] = proj_xr[i] = view_cos[i] = 0;
        pred_level[i] = 0;
        const float *P = p->pos + 3 * (size_t)i;
        float Pc[3];
        xform(p->R, p->t, P, Pc);
        if (Pc[2] < 0.0f) continue;
        const float invz = 1.0f / Pc[2];
        const float u = p->fx * Pc[0] * invz + p->cx;
        const float v = p->fy * Pc[1] * invz + p->cy;
        if (u < min_x || u > max_x) continue;
        if (v < min_y || v > max_y) continue;
        float PO[3] = {P[0] - p->Ow[0], P[1] - p->Ow[1], P[2] - p->Ow[2]};
        const float dist = norm3(PO);
        if (dist < 0.8f * p->min_dist[i] || dist > 1.2f * p->max_dist[i]) continue; /* Get{Min,Max}DistanceInvariance() src/MapPoint.cc:413-423 */
        const float *Pn = p->normal + 3 * (size_t)i;
        const double dot = (double)PO[0] * (double)Pn[0] + (double)PO[1] * (double)Pn[1] + (double)PO[2] * (double)Pn[2];
        const float viewCos = (float)(dot / dist);
        if (viewCos < viewing_cos_limit) continue;
        in_view[i] = 1;
        proj_x[i] = u;
        proj_xr[i] = u - p->bf * invz;
        proj_y[i] = v;
        pred_level[i] = predict_scale(p->max_dist[i], dist, p->log_scale_factor, n_levels);
        view_cos[i] = viewCos;
    }
}

/* Frame::AssignFeaturesToGrid src/Frame.cc:259-274 with PosInGrid :411-424: mGrid as CSR (cell = ix*48+iy,
 * features of a cell in ascending index = push_back order).  Returns the number of features inside the grid. */
int orc_assign_features_to_grid(int n, const float *kp_x, const float *kp_y, float min_x, float min_y, float grid_w_inv,
                                float grid_h_inv, int32_t *grid_off, int32_t *grid_idx)
{
    const int NC = FRAME_GRID_COLS * FRAME_GRID_ROWS;
    int *cell = (int *)malloc(sizeof(int) * (n ? n : 1));
    for (int c = 0; c <= NC; ++c) grid_off[c] = 0;
    for (int i = 0; i < n; ++i) {
        const int posX = (int)roundf((kp_x[i] - min_x) * grid_w_inv);
        const int posY = (int)roundf((kp_y[i] - min_y) * grid_h_inv);
        cell[i] = (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) ? -1 : posX * FRAME_GRID_ROWS + posY;
        if (cell[i] >= 0) grid_off[cell[i] + 1]++;
    }
    for (int c = 0; c < NC; ++c) grid_off[c + 1] += grid_off[c];
    int *fill = (int *)calloc(NC, sizeof(int));
    for (int i = 0; i < n; ++i)
        if (cell[i] >= 0) grid_idx[grid_off[cell[i]] + fill[cell[i]]++] = i;
    free(fill);
    free(cell);
    return grid_off[NC];
}

/* Frame::ComputeStereoFromRGBD src/Frame.cc:672-693: depth image CV_32F (stride in floats) */
void orc_stereo_from_rgbd(int n, const float *kp_x, const float *kp_y, const float *kpun_x, const float *depth_img,
                          int stride, float mbf, float *u_right, float *depth)
{
    for (int i = 0; i < n; ++i) {
        u_right[i] = depth[i] = -1;
        const float v = kp_y[i], u = kp_x[i];
        const float d = depth_img[(size_t)(int)v * stride + (int)u];
        if (d > 0) {
            depth[i] = d;
            u_right[i] = kpun_x[i] - mbf / d;
        }
    }
}

/* cv::undistortPoints(src, dst, K, distCoeffs, noArray(), K) for CV_32FC2 points, as Frame::UndistortKeyPoints and
 * Frame::ComputeImageBounds call it (src/Frame.cc:433-493).  OpenCV (3.2, un-vendored: CMakeLists.txt:33-39)
 * imgproc/undistort.cpp cvUndistortPoints: camera matrix and coefficients widened to double, 5 fixed-point iterations of
 * the inverse of the radial (k1 k2 k3) + tangential (p1 p2) model, then P * R = K applied: xx = fx x + cx with
 * ww = 1 / (0 x + 0 y + 1).  k = k1 k2 p1 p2 k3 (mDistCoef, float32).  Results are rounded to float. */
void orc_undistort_points(int n, const float *xy_in, float fx_f, float fy_f, float cx_f, float cy_f, const float *k_f,
                          float *xy_out)
{
    const double fx = (double)fx_f, fy = (double)fy_f, cx = (double)cx_f, cy = (double)cy_f;
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k0 = (double)k_f[0], k1 = (double)k_f[1], k2 = (double)k_f[2], k3 = (double)k_f[3], k4 = (double)k_f[4];
    for (int i = 0; i < n; ++i) {
        double x = (double)xy_in[2 * i], y = (double)xy_in[2 * i + 1];
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            /* (k[5..7] = 0: the rational numerator is 1 + ((0 r2 + 0) r2 + 0) r2 = 1) */
            const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
            const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0. * r2 + 0. * r2 * r2;
            const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0. * r2 + 0. * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = fx * x + 0. * y + cx;
        const double yy = 0. * x + fy * y + cy;
        const double ww = 1. / (0. * x + 0. * y + 1.);
        xy_out[2 * i] = (float)(xx * ww);
        xy_out[2 * i + 1] = (float)(yy * ww);
    }
}
