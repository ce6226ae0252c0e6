/*
 * ORACLE (test infrastructure only) -- CPU restatement of the Active-ORB-SLAM2 hot path.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors for this path and cannot be
 * built in this image (OpenCV + Eigen are absent), so this restatement is pinned only by its own
 * known-answer tests (tests/test_oracle_*.py).  See DESIGN.md "Oracle".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (active-orb-slam2_amd/) never links, imports or calls it.
 *
 * Citations are path:line relative to /root/reference.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bit-compatible with cv::KeyPoint (28 B) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint_t;

typedef struct orc_extractor orc_extractor_t;

/* ---- extractor: src/ORBextractor.cc ---- */
orc_extractor_t *orc_extractor_create(int nfeatures, float scale_factor, int nlevels,
                                      int ini_th_fast, int min_th_fast);
void orc_extractor_destroy(orc_extractor_t *e);
/* getters (include/ORBextractor.h:67-83) */
int orc_extractor_levels(const orc_extractor_t *e);
const float *orc_extractor_scale_factors(const orc_extractor_t *e);
const float *orc_extractor_inv_scale_factors(const orc_extractor_t *e);
const float *orc_extractor_sigma2(const orc_extractor_t *e);
const float *orc_extractor_inv_sigma2(const orc_extractor_t *e);
const int *orc_extractor_features_per_level(const orc_extractor_t *e);
const int *orc_extractor_umax(const orc_extractor_t *e);

/* operator() (src/ORBextractor.cc:1043-1105). Returns 0 ok, <0 error (-2 capacity).
 * kps/desc may be NULL to only count. */
int orc_extractor_extract(orc_extractor_t *e, const uint8_t *img, int w, int h, int stride,
                          orc_keypoint_t *kps, uint8_t *desc, int cap, int *n_out);

/* stage access, valid after the last extract on this instance */
int orc_extractor_level_size(const orc_extractor_t *e, int level, int *w, int *h);
/* interior plane (w x h; pitch = return value of orc_extractor_level_size) of the level */
const uint8_t *orc_extractor_level_plane(const orc_extractor_t *e, int level);
/* blurred interior plane of the level (src/ORBextractor.cc:1085-1086) */
const uint8_t *orc_extractor_level_blurred(const orc_extractor_t *e, int level);
/* FAST candidates handed to DistributeOctTree for a level, in emission order.
 * xy are coordinates relative to (minBorderX,minBorderY). returns count. */
int orc_extractor_level_candidates(const orc_extractor_t *e, int level, const int16_t **xs,
                                   const int16_t **ys, const uint8_t **score);
/* number of keypoints kept on a level (octree output) */
int orc_extractor_level_nkeys(const orc_extractor_t *e, int level);

/* ---- stand-alone primitives (known-answer tests) ---- */
/* cv::FAST(img, kps, th, true), TYPE_9_16, on a whole sub-image; returns count; outputs
 * x,y,score in emission order (row-major). cap = capacity of out arrays. */
int orc_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold, int16_t *xs,
                 int16_t *ys, uint8_t *score, int cap);
/* corner score of one pixel (ring must be inside the image) */
int orc_fast_corner_score(const uint8_t *center, int stride, int threshold);
float orc_fast_atan2(float y, float x);
/* rBRIEF steering trig: 0 = correctly rounded (default, parity convention 3), 1 = libm cosf/sinf */
void orc_set_trig_mode(int use_libm);
void orc_set_tiebreak_mode(int reverse);   /* octree: equal-size nodes in the opposite order (measurement only) */
void orc_set_log_mode(int use_libm);        /* PredictScale: libm logf (measurement only) */
void orc_set_lba_variant(int bits);         /* LocalBA: 1 = reduced system eliminated last-unknown-first, 2 = long double accumulations (measurement only) */
void orc_sincos_exact(float angle_rad, float *s_out, float *c_out);
int orc_cv_round_f(float v);
/* cv::resize INTER_LINEAR 8UC1 */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw,
                          int dh, int dstride);
/* cv::GaussianBlur 7x7 sigma 2 BORDER_REFLECT_101 8UC1 */
void orc_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst,
                           int dstride);
/* cv::copyMakeBorder(..., BORDER_REFLECT_101) */
void orc_copy_make_border_reflect101(const uint8_t *src, int w, int h, int sstride, uint8_t *dst,
                                     int border, int dstride);
/* DistributeOctTree on an explicit candidate list (coords relative to min border).
 * out_idx receives indices into the candidate list in list order; returns count. */
int orc_distribute_octree(const float *xs, const float *ys, const float *resp, int n, int minX,
                          int maxX, int minY, int maxY, int N, int *out_idx, int cap);

/* ---- matcher: src/ORBmatcher.cc ---- */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b);

typedef struct {
    int n_kf, n_f;
    const uint8_t *desc_kf;     /* n_kf x 32 */
    const uint8_t *desc_f;      /* n_f x 32 */
    const uint8_t *kf_has_mp;   /* n_kf: map point present and not bad */
    const float *angle_kf;      /* pKF->mvKeysUn[i].angle */
    const float *angle_f;       /* F.mvKeys[j].angle */
    /* FeatureVectors as sorted CSR: node ids ascending */
    int n_nodes_kf, n_nodes_f;
    const int32_t *node_id_kf, *node_off_kf, *node_idx_kf;
    const int32_t *node_id_f, *node_off_f, *node_idx_f;
    float nnratio;
    int check_orientation;
} orc_bow_problem_t;
/* SearchByBoW(KF,F) src/ORBmatcher.cc:159-288. match_f[n_f] = KF feature index or -1. */
int orc_search_by_bow(const orc_bow_problem_t *p, int32_t *match_f);

typedef struct {
    /* frame side */
    int n_f;
    const uint8_t *desc_f;
    const float *kp_x, *kp_y;    /* mvKeysUn[i].pt */
    const int32_t *kp_octave;    /* mvKeysUn[i].octave */
    const float *kp_angle;       /* mvKeysUn[i].angle */
    const float *u_right;        /* mvuRight */
    const float *scale_factors;  /* mvScaleFactors */
    int n_levels;
    float min_x, min_y, max_x, max_y;             /* mnMinX.. */
    float grid_w_inv, grid_h_inv;                 /* mfGridElement{Width,Height}Inv */
    /* mGrid[64][48] as CSR, cell = ix*48+iy */
    const int32_t *grid_off, *grid_idx;
    /* per feature: 0 = no map point, 1 = map point with Observations()==0,
     * 2 = map point with Observations()>0. Updated in place semantics are internal. */
    const uint8_t *f_mp_state;
} orc_frame_view_t;

typedef struct {
    int n_mp;
    const uint8_t *track_in_view;   /* mbTrackInView && !isBad() */
    const int32_t *pred_level;      /* mnTrackScaleLevel */
    const float *view_cos, *proj_x, *proj_y, *proj_xr;
    const uint8_t *desc;            /* n_mp x 32 */
    const uint8_t *has_obs;         /* Observations()>0 */
    float th, nnratio;
} orc_proj_mp_problem_t;
/* SearchByProjection(F, vpMapPoints, th) src/ORBmatcher.cc:45-129.
 * match_f[n_f]: index into map point list newly assigned to feature, or -1 (unchanged). */
int orc_search_by_projection_mp(const orc_frame_view_t *f, const orc_proj_mp_problem_t *p,
                                int32_t *match_f);

typedef struct {
    int n_last;
    const uint8_t *last_valid;   /* mvpMapPoints[i] && !mvbOutlier[i] */
    const float *world_pos;      /* n_last x 3 */
    const uint8_t *desc;         /* n_last x 32  (pMP->GetDescriptor()) */
    const int32_t *last_octave;  /* LastFrame.mvKeys[i].octave */
    const float *last_angle;     /* LastFrame.mvKeysUn[i].angle */
    const uint8_t *has_obs;      /* pMP->Observations()>0 */
    float Tcw[16], Tlw[16];      /* row-major 4x4 float */
    float fx, fy, cx, cy, mb, mbf;
    float th;
    int mono, check_orientation;
} orc_proj_last_problem_t;
/* SearchByProjection(Cur, Last, th, bMono) src/ORBmatcher.cc:1328-1470
 * match_f[n_f]: last-frame index assigned, -1 unchanged, -2 reset to NULL by the rotation cull */
int orc_search_by_projection_last(const orc_frame_view_t *cur, const orc_proj_last_problem_t *p,
                                  int32_t *match_f);

/* ---- the remaining matcher methods ---- */
typedef struct {
    int n1, n2;
    const uint8_t *desc1, *desc2;         /* pKF1/pKF2->mDescriptors */
    const uint8_t *has_mp1, *has_mp2;     /* GetMapPointMatches()[i] != NULL && !isBad() */
    const float *angle1, *angle2;         /* mvKeysUn[i].angle */
    int n_nodes1, n_nodes2;               /* FeatureVectors as CSR */
    const int32_t *node_id1, *node_off1, *node_idx1;
    const int32_t *node_id2, *node_off2, *node_idx2;
    float nnratio;
    int check_orientation;
} orc_bow_kf_problem_t;
int orc_search_by_bow_kf(const orc_bow_kf_problem_t *p, int32_t *match12);

typedef struct {
    int n1, n2;
    const uint8_t *desc1, *desc2;
    const uint8_t *has_mp1, *has_mp2;     /* GetMapPoint(i) != NULL (no isBad test here, :700,:723) */
    const float *x1, *y1, *angle1, *u_right1;      /* mvKeysUn / mvuRight of pKF1 */
    const float *x2, *y2, *angle2, *u_right2;
    const int32_t *octave2;
    const float *scale_factors2, *level_sigma2_2;  /* pKF2->mvScaleFactors / mvLevelSigma2 */
    float F12[9];                         /* row-major 3x3 */
    float ex, ey;                         /* epipole in the second image (:664-670) */
    int only_stereo, check_orientation;
    int n_nodes1, n_nodes2;
    const int32_t *node_id1, *node_off1, *node_idx1;
    const int32_t *node_id2, *node_off2, *node_idx2;
} orc_triang_problem_t;
int orc_search_for_triangulation(const orc_triang_problem_t *p, int32_t *match12);
void orc_compute_distinctive_descriptors(int n_points, const int32_t *off, const uint8_t *desc, int32_t *best);

/* projection family: a set of map points projected into a keyframe / frame (orc_frame_view_t) */
typedef struct {
    int n_pts;
    const uint8_t *valid;          /* per-point static gate, see each function */
    const float *pos;              /* 3 per point: GetWorldPos() */
    const float *max_dist, *min_dist;  /* mfMaxDistance / mfMinDistance, RAW: the gates apply 1.2f / 0.8f, PredictScale takes the raw maximum */
    const float *normal;           /* 3 per point: GetNormal() (Fuse, SearchByProjection(KF,Scw)) */
    const uint8_t *desc;           /* 32 per point: GetDescriptor() */
    const float *q_angle;          /* SearchByProjection(F,KF): pKF->mvKeysUn[i].angle */
    float R[9], t[3], Ow[3];       /* camera rotation / translation (row-major) and centre */
    float R2[9], t2[3];            /* SearchBySim3: sR21, t21 (or sR12, t12) */
    float fx, fy, cx, cy, bf;
    float log_scale_factor;        /* mfLogScaleFactor of the target */
    const float *inv_level_sigma2; /* mvInvLevelSigma2 of the target (Fuse) */
    float th;
} orc_proj_gen_t;
int orc_fuse(const orc_frame_view_t *f, const orc_proj_gen_t *p, int32_t *best_idx, int32_t *best_dist);
int orc_fuse_sim3(const orc_frame_view_t *f, const orc_proj_gen_t *p, int32_t *best_idx, int32_t *best_dist);
int orc_search_by_projection_kf(const orc_frame_view_t *f, const orc_proj_gen_t *p, int32_t *match_f);
int orc_search_by_sim3(const orc_frame_view_t *f1, const orc_frame_view_t *f2, const orc_proj_gen_t *p12,
                       const orc_proj_gen_t *p21, int32_t *match12);
int orc_search_by_projection_reloc(const orc_frame_view_t *f, const orc_proj_gen_t *p, int orb_dist,
                                   int check_orientation, int32_t *match_f);

int orc_assign_features_to_grid(int n, const float *kp_x, const float *kp_y, float min_x, float min_y, float grid_w_inv,
                                float grid_h_inv, int32_t *grid_off, int32_t *grid_idx);
/* cv::undistortPoints(src, dst, K, dist, noArray(), K) on n float points (x, y interleaved); k = k1 k2 p1 p2 k3 */
void orc_undistort_points(int n, const float *xy_in, float fx, float fy, float cx, float cy, const float *k, float *xy_out);
void orc_stereo_from_rgbd(int n, const float *kp_x, const float *kp_y, const float *kpun_x, const float *depth_img,
                          int stride, float mbf, float *u_right, float *depth);
void orc_is_in_frustum(const orc_proj_gen_t *p, float min_x, float max_x, float min_y, float max_y, int n_levels,
                       float viewing_cos_limit, uint8_t *in_view, float *proj_x, float *proj_y, float *proj_xr,
                       int32_t *pred_level, float *view_cos);
int orc_search_for_initialization(const orc_frame_view_t *f2, int n1, const uint8_t *desc1, const int32_t *octave1,
                                  const float *angle1, const float *prev_xy, int window_size, float nnratio,
                                  int check_orientation, int32_t *match12);

/* ---- Frame::ComputeStereoMatches src/Frame.cc:495-669 ---- */
typedef struct {
    int n_left, n_right;
    const orc_keypoint_t *kp_left, *kp_right;   /* mvKeys / mvKeysRight (level-0 pixel units) */
    const uint8_t *desc_left, *desc_right;      /* mDescriptors / mDescriptorsRight */
    int n_levels;
    const float *scale_factors, *inv_scale_factors;
    const uint8_t *const *left_planes;          /* mpORBextractorLeft->mvImagePyramid[l] interior */
    const uint8_t *const *right_planes;
    const int *left_pitch, *right_pitch, *level_w, *level_h;
    float mb, mbf;
} orc_stereo_problem_t;
/* fills mvuRight / mvDepth (-1 = no match); returns the number of matches before the median cull */
int orc_compute_stereo_matches(const orc_stereo_problem_t *p, float *u_right, float *depth);

/* ---- DBoW2 vocabulary (ORBVocabulary), SURVEY §8(f) rank 3: dbow_oracle.c ---- */
typedef struct orc_vocab orc_vocab_t;
orc_vocab_t *orc_vocab_create(void);
void orc_vocab_destroy(orc_vocab_t *v);
int orc_vocab_k(const orc_vocab_t *v);
int orc_vocab_L(const orc_vocab_t *v);
int orc_vocab_scoring(const orc_vocab_t *v);
int orc_vocab_weighting(const orc_vocab_t *v);
int orc_vocab_nodes(const orc_vocab_t *v); /* m_nodes.size() (root included) */
int orc_vocab_size(const orc_vocab_t *v);  /* size() = number of words */
int orc_vocab_set_nodes(orc_vocab_t *v, int k, int L, int scoring, int weighting, int n, const int32_t *parent,
                        const uint8_t *desc, const double *weight, const uint8_t *is_leaf);
int orc_vocab_load_binary(orc_vocab_t *v, const char *path);
int orc_vocab_save_binary(const orc_vocab_t *v, const char *path);
int orc_vocab_load_text(orc_vocab_t *v, const char *path);
int orc_vocab_transform(const orc_vocab_t *v, const uint8_t *desc, int n, int levelsup, uint32_t *bow_word,
                        double *bow_value, int *n_bow, int32_t *fv_node, int32_t *fv_off, int32_t *fv_idx,
                        int *n_fv, uint32_t *word_of, uint32_t *node_of);
double orc_vocab_score_l1(const uint32_t *w1, const double *v1, int n1, const uint32_t *w2, const double *v2, int n2);

/* ---- local BA: src/Optimizer.cc:454-779 + vendored g2o ---- */
typedef struct {
    int n_poses;               /* local + fixed keyframes */
    int n_points;
    int n_edges;
    double *pose_qt;           /* n_poses x 7: qx qy qz qw tx ty tz (SE3Quat) in/out */
    const uint8_t *pose_fixed; /* n_poses */
    const int64_t *pose_id;    /* KeyFrame::mnId (ordering) */
    double *point_xyz;         /* n_points x 3 in/out */
    const int64_t *point_id;   /* MapPoint::mnId (ordering) */
    const int32_t *edge_pose, *edge_point;
    const double *edge_obs;    /* n_edges x 3 (u, v, ur); ur ignored for mono */
    const uint8_t *edge_stereo;
    const float *edge_inv_sigma2;
    double fx, fy, cx, cy, bf;
    const volatile uint8_t *stop_flag; /* may be NULL */
    int iters1, iters2;        /* 5, 10 */
    int stop_at_poll;          /* test hook, 0 = off: see terminate_flag() */
} orc_lba_problem_t;

typedef struct {
    double *edge_chi2;          /* n_edges: e->chi2() at the end */
    uint8_t *edge_depth_pos;    /* n_edges: isDepthPositive() at the end */
    uint8_t *edge_outlier;      /* n_edges: erase decision (chi2>th || !depth) */
    uint8_t *edge_level1;       /* n_edges: excluded after first pass */
    double lambda_trace[64];    /* lambda after each outer iteration */
    double chi2_trace[64];      /* robust chi2 after each outer iteration */
    int n_trace;
    int iters_done1, iters_done2;
    int polls, stop_poll, trials; /* evaluations of terminate(), the first that saw the flag set (0 = none), LM trial steps */
} orc_lba_result_t;
int orc_lba_solve(orc_lba_problem_t *p, orc_lba_result_t *r);

/* ---- Optimizer::PoseOptimization src/Optimizer.cc:239-452 ---- */
typedef struct {
    int n;                     /* features with a map point (nInitialCorrespondences) */
    const double *Xw;          /* n x 3: MapPoint::GetWorldPos() (float32 widened) */
    const double *obs;         /* n x 3: kpUn.pt.x, kpUn.pt.y, mvuRight */
    const uint8_t *stereo;     /* n: mvuRight >= 0 */
    const float *inv_sigma2;   /* n: mvInvLevelSigma2[octave] */
    double fx, fy, cx, cy, bf;
} orc_pose_problem_t;
/* returns nInitialCorrespondences - nBad; outlier[n] = pFrame->mvbOutlier */
int orc_pose_optimization(const orc_pose_problem_t *p, const double pose_in[7], double pose_out[7],
                          uint8_t *outlier, int *n_bad_out);

/* Converter.cc:37-71 boundary conversions */
void orc_pose_from_Tcw_f32(const float T[16], double qt[7]);
void orc_pose_to_Tcw_f32(const double qt[7], float T[16]);
/* SE3 helpers for tests */
void orc_se3_exp(const double upd[6], double qt[7]);
void orc_se3_mul(const double a[7], const double b[7], double out[7]);
void orc_quat_from_rot(const double R[9], double q[4]);
void orc_rot_from_quat(const double q[4], double R[9]);
/* residual + jacobians of one edge (tests: vs central differences) */
void orc_edge_linearize(const double pose_qt[7], const double xyz[3], const double obs[3],
                        int stereo, double fx, double fy, double cx, double cy, double bf,
                        double err[3], double Ji[9], double Jj[18]);

#ifdef __cplusplus
}
#endif
#endif
