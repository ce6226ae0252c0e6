/*
 * aos2.h -- C ABI of the MI355X-native ORB front-end + LocalBA hot path.
 *
 * This is the drop-in boundary under the reference's three C++ class surfaces
 * (SURVEY.md section 8(b)); every entry point names the reference interface it replaces
 * (path:line relative to the Active-ORB-SLAM2 checkout).  Plain C, POD arguments, caller-allocated
 * buffers, int status (0 = ok, <0 = error, see AOS2_ERR_*).  No torch / OpenCV / Eigen types.
 *
 * Threading: a handle is NOT re-entrant (like an ORBextractor instance, which owns a mutable
 * mvImagePyramid); different handles may be used concurrently from different threads (the stereo
 * path runs one extractor per eye, src/Frame.cc:103-106).  Each handle owns one HIP stream.
 *
 * There is no CPU fallback: every compute entry point returns AOS2_ERR_NO_DEVICE when no HIP
 * device is present.
 */
#ifndef AOS2_H
#define AOS2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AOS2_OK 0
#define AOS2_ERR_ARG (-1)        /* bad argument (NULL, non-positive size, unsupported layout) */
#define AOS2_ERR_CAPACITY (-2)   /* output buffer too small; *n_out still holds the needed count */
#define AOS2_ERR_TOO_SMALL (-3)  /* image too small for the pyramid (reference would divide by 0) */
#define AOS2_ERR_HIP (-4)        /* HIP runtime error, see aos2_last_error() */
#define AOS2_ERR_NO_DEVICE (-5)  /* no HIP device / device index out of range */
#define AOS2_ERR_STOPPED (1)     /* LocalBA: stop flag was set before optimisation started */

/* last error text of the calling thread */
const char *aos2_last_error(void);
/* number of HIP devices visible (0 when none; never fails) */
int aos2_device_count(void);
const char *aos2_version(void);
/* The host CPUs local to the device (the NUMA node its PCIe root hangs off), as the kernel's list string ("0-63,128-191": sysfs
 * local_cpulist of the device's PCI function); "" when the platform does not say.  For callers that place their threads: bench.py's
 * composite ran at 51-52 k frames/s when pinned to the other socket's CPUs and at 59-63 k on the local ones (AOS2_BENCH_NUMA; on the
 * shared hosts of the test pool the unpinned default was as good as the local node).  Returns AOS2_ERR_CAPACITY
 * when `cap` is too small, AOS2_ERR_NO_DEVICE without that device.  No reference equivalent (the reference never leaves the CPU). */
int aos2_device_local_cpus(int device, char *buf, int cap);

/* ------------------------------------------------------------------------------------------
 * ORBextractor  (include/ORBextractor.h:45-111, src/ORBextractor.cc)
 * ------------------------------------------------------------------------------------------ */

/* bit-compatible with cv::KeyPoint (28 bytes) */
typedef struct {
    float x, y;      /* pt, level-0 pixel units */
    float size;      /* 31 * scale[octave], truncated (src/ORBextractor.cc:837,846) */
    float angle;     /* degrees [0,360) */
    float response;  /* FAST-9/16 corner score */
    int32_t octave;
    int32_t class_id; /* -1 */
} aos2_keypoint_t;

typedef struct aos2_extractor aos2_extractor_t;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * src/ORBextractor.cc:410-470.  device = HIP device index.  Creation only builds the host-side
 * tables (no HIP call), so it succeeds without a GPU; the first extract binds the device. */
int aos2_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast,
                          int min_th_fast, int device, aos2_extractor_t **out);
void aos2_extractor_destroy(aos2_extractor_t *e);

/* GetLevels / GetScaleFactor(s) / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:58-83).  Arrays have nlevels entries and
 * live as long as the handle. */
int aos2_extractor_levels(const aos2_extractor_t *e);
float aos2_extractor_scale_factor(const aos2_extractor_t *e);
const float *aos2_extractor_scale_factors(const aos2_extractor_t *e);
const float *aos2_extractor_inv_scale_factors(const aos2_extractor_t *e);
const float *aos2_extractor_sigma2(const aos2_extractor_t *e);
const float *aos2_extractor_inv_sigma2(const aos2_extractor_t *e);
/* mnFeaturesPerLevel (include/ORBextractor.h:101) and umax (:103) -- exposed for tests */
const int *aos2_extractor_features_per_level(const aos2_extractor_t *e);
const int *aos2_extractor_umax(const aos2_extractor_t *e);

/* ORBextractor::operator()(image, mask, keypoints, descriptors)  src/ORBextractor.cc:1043-1105.
 * img: 8-bit single channel, host memory, `stride` bytes per row (the CV_8UC1 assert of :1050 is
 * the caller's contract).  mask is ignored by the reference and has no parameter here.
 * kps[cap], desc[cap*32] are caller-allocated (host).  *n_out = number of keypoints.
 * Empty image (img==NULL or w<=0 or h<=0) -> AOS2_OK with *n_out = 0 (silent return, :1046).
 * The result can exceed nfeatures by a few keypoints (octree overshoot);
 * cap >= aos2_extractor_max_keypoints() always suffices. */
int aos2_extractor_extract(aos2_extractor_t *e, const uint8_t *img, int w, int h, int stride,
                           aos2_keypoint_t *kps, uint8_t *desc, int cap, int *n_out);
int aos2_extractor_max_keypoints(const aos2_extractor_t *e);
/* The same bound for a given image size, before any image was seen: DistributeOctTree's first pass divides all
 * round(W / H) root nodes (src/ORBextractor.cc:549-590), so a wide image with few features per level can return more
 * than nfeatures-per-level + 3 keypoints (4 * round(W / H) per level). */
int aos2_extractor_max_keypoints_for(const aos2_extractor_t *e, int w, int h);

/* Batched form for frame-parallel throughput: `batch` images of identical size, host memory,
 * image b at imgs + b*image_stride.  Outputs are [batch][cap] / [batch][cap][32] / [batch]. */
int aos2_extractor_extract_batch(aos2_extractor_t *e, const uint8_t *imgs, int batch, int w, int h,
                                 int stride, size_t image_stride, aos2_keypoint_t *kps,
                                 uint8_t *desc, int cap, int32_t *n_out);

/* Page-locked host memory for the host-pointer entry points above.  Optional: any host memory works.  The batched
 * call uploads, computes and downloads chunk by chunk on separate streams, so the PCIe copies of one part of the
 * batch overlap the kernels of another; page-locked RESULT buffers matter most (256 TUM frames: 2.1 ms per call =
 * 120 k frames/s, 44 GB/s over PCIe, against 8.8 ms with the un-pipelined copies into pageable results this replaced;
 * pageable and page-locked INPUTS measured the same, tools/gpu_latency.py).  AOS2_ERR_NO_DEVICE without a GPU. */
int aos2_host_alloc(void **p, size_t bytes);
int aos2_host_free(void *p);

/* Same with inputs and outputs resident in device memory (HBM).  d_* are device pointers owned
 * by the caller; the call is synchronous on the handle's stream (returns when results are in
 * d_kps / d_desc / d_n_out). */
int aos2_extractor_extract_batch_device(aos2_extractor_t *e, const uint8_t *d_imgs, int batch,
                                        int w, int h, int stride, size_t image_stride,
                                        aos2_keypoint_t *d_kps, uint8_t *d_desc, int cap,
                                        int32_t *d_n_out);

/* Asynchronous form of the above: enqueues the batch on the handle's HIP streams and returns; the outputs are
 * complete, and errors (octree capacity, more keypoints than `cap`) reported, by aos2_extractor_wait().
 * Several batches may be enqueued before one wait: chunk c of every batch runs on stream c with its own scratch,
 * so the latency-bound octree stage of one batch overlaps the FAST / descriptor kernels of the next (the
 * frame-parallel pipeline of SURVEY.md section 8(e)).  The caller must not reuse the input or output buffers of a
 * batch that is still in flight, and all batches of one flight share (w, h, cap, batch); a change of geometry or
 * of batch size waits for the flight first.  Every other call on the handle waits for the flight implicitly. */
int aos2_extractor_extract_batch_device_async(aos2_extractor_t *e, const uint8_t *d_imgs, int batch,
                                              int w, int h, int stride, size_t image_stride,
                                              aos2_keypoint_t *d_kps, uint8_t *d_desc, int cap,
                                              int32_t *d_n_out);
int aos2_extractor_wait(aos2_extractor_t *e);
/* Device-side ordering instead of a host wait: everything enqueued on `hip_stream` (a hipStream_t of the caller, e.g.
 * the stream of an aos2_frames_t, or 0 for the null stream) after this call runs after everything enqueued on the
 * extractor's streams before it.  Errors of the batches in flight are still reported by aos2_extractor_wait(). */
int aos2_extractor_stream_wait(aos2_extractor_t *e, void *hip_stream);
/* The other direction: every batch enqueued on the extractor after this call runs after everything enqueued on `hip_stream` before
 * it -- e.g. the caller's own hipMemcpyAsync of the images from page-locked host memory (the reference's images arrive as host
 * cv::Mat, src/Frame.cc:276-282; bench.py --host-images).  Device-side, no host wait. */
int aos2_extractor_wait_for_stream(aos2_extractor_t *e, void *hip_stream);
/* The exchange step of the frame-sharded path: packs the device outputs of a batch into fixed-size
 * per-frame slots {int32 n; int32 pad[3]; KeyPoint[cap]; uint8 desc[cap][32]} (slot_bytes each, a multiple of 16, unused
 * tail zeroed) that one collective (RCCL gather over xGMI) moves to rank 0.  Enqueued on `hip_stream` behind the
 * extractor's pending work; cap must be a multiple of 4. */
int aos2_extractor_pack_slots(aos2_extractor_t *e, int batch, const aos2_keypoint_t *d_kps, const uint8_t *d_desc,
                              const int32_t *d_n, int cap, uint8_t *d_slots, size_t slot_bytes, void *hip_stream);

/* mvImagePyramid[level] (include/ORBextractor.h:85; read by Frame::ComputeStereoMatches,
 * src/Frame.cc:502,592,609) of image `image` of the last extract on this handle.
 * Copies the level to host memory `dst` with row pitch dst_stride.  border = 0 gives the w x h
 * interior (the ROI the reference exposes); border = 19 gives the (w+38) x (h+38) buffer with the
 * BORDER_REFLECT_101 frame the reference keeps around it (:1113-1128). */
int aos2_extractor_pyramid_level_size(const aos2_extractor_t *e, int level, int *w, int *h);
int aos2_extractor_pyramid_level(aos2_extractor_t *e, int image, int level, int border,
                                 uint8_t *dst, int dst_stride);

/* Frame::ComputeStereoMatches()  src/Frame.cc:495-669  (SURVEY.md §8(f) rank 2).
 * `left` / `right` are the two ORBextractor handles of the stereo Frame (mpORBextractorLeft / Right,
 * include/Frame.h:108); both must still hold the device pyramids of their last extract*() call
 * (mvImagePyramid, read at :502,592,609) and `image` selects the image of that batch.
 * kp_* / desc_* = mvKeys / mvKeysRight and mDescriptors / mDescriptorsRight (host memory).
 * mb = baseline in metres (minZ, :526), mbf = baseline * fx (:527).
 * Outputs mvuRight[n_left], mvDepth[n_left]; -1 = no stereo match (:498).
 * When no left keypoint passes the SAD stage the reference reads vDistIdx[0] of an empty vector
 * (:655); here that case leaves every output at -1. */
int aos2_compute_stereo_matches(aos2_extractor_t *left, aos2_extractor_t *right, int image,
                                const aos2_keypoint_t *kp_left, const uint8_t *desc_left, int n_left,
                                const aos2_keypoint_t *kp_right, const uint8_t *desc_right,
                                int n_right, float mb, float mbf, float *u_right, float *depth);
/* Batched, fully device-resident form: the arrays are exactly what
 * aos2_extractor_extract_batch_device() wrote for the two eyes ([batch][cap] keypoints,
 * [batch][cap][32] descriptors, [batch] counts; descriptor blocks 16-byte aligned); image b of the
 * call = image b of both extractors' last batch.  d_u_right / d_depth are [batch][cap]. */
int aos2_compute_stereo_matches_device(aos2_extractor_t *left, aos2_extractor_t *right, int batch,
                                       const aos2_keypoint_t *d_kp_left, const uint8_t *d_desc_left,
                                       const int32_t *d_n_left, const aos2_keypoint_t *d_kp_right,
                                       const uint8_t *d_desc_right, const int32_t *d_n_right,
                                       int cap, float mb, float mbf, float *d_u_right, float *d_depth);
/* The same without a host wait: left's first stream waits on the device for both extractors' batches in flight
 * (aos2_extractor_extract_batch_device_async -- the two ExtractORB threads of src/Frame.cc:103-109 joined), the kernels
 * are enqueued behind; what orders itself behind `left` afterwards (aos2_extractor_stream_wait, aos2_frames_build_stereo,
 * aos2_extractor_wait) runs behind them.  The kernels read BOTH extractors' pyramid blocks: the next extraction of either
 * extractor waits for them on the device, on every stream it uses, before it rewrites the pyramids -- the calls can be
 * pipelined without a host wait in between.  The keypoint / descriptor / count / output buffers of the call are the
 * caller's to keep untouched until left's stream has drained.  No device time is recorded. */
int aos2_compute_stereo_matches_device_async(aos2_extractor_t *left, aos2_extractor_t *right, int batch,
                                             const aos2_keypoint_t *d_kp_left, const uint8_t *d_desc_left,
                                             const int32_t *d_n_left, const aos2_keypoint_t *d_kp_right,
                                             const uint8_t *d_desc_right, const int32_t *d_n_right,
                                             int cap, float mb, float mbf, float *d_u_right, float *d_depth);
/* device time (ms) of the kernels of the last ComputeStereoMatches call on `left` */
float aos2_compute_stereo_matches_last_device_ms(const aos2_extractor_t *left);

/* Stage taps for parity tests (no reference equivalent): FAST candidates handed to
 * DistributeOctTree for (image, level) of the last extract, in the reference's emission order
 * (cells row-major, pixels row-major inside a cell); coordinates relative to (minBorderX,
 * minBorderY) like src/ORBextractor.cc:822-823. Returns count via *n (arrays may be NULL). */
int aos2_extractor_debug_candidates(aos2_extractor_t *e, int image, int level, int16_t *xs,
                                    int16_t *ys, uint8_t *score, int cap, int *n);

/* A batch is cut into `chunks` sub-batches that run on separate HIP streams so that the
 * latency-bound octree kernel of one overlaps the streaming kernels of the others.
 * 0 = automatic (2 for batch >= 64, else 1); 1 = one stream (stage timing below is
 * only meaningful then); at most 4. */
int aos2_extractor_set_chunks(aos2_extractor_t *e, int chunks);

/* Timing of the last batch, milliseconds, measured with HIP events on the handle's stream:
 * [0] pyramid kernels, [1] FAST+NMS kernel, [2] candidate compaction, [3] octree stage
 * (device kernel, or D2H + host + H2D when the host octree is selected), [4] orientation +
 * blur + rBRIEF kernel, [5] whole call (host wall clock).  n <= 8 values are written. */
int aos2_extractor_last_timing(const aos2_extractor_t *e, float *ms, int n);

/* Launches only the FAST+NMS kernel `iters` times on the pyramid of the last batch and returns
 * the average kernel time in ms (HIP events on the handle's stream) -- roofline measurement. */
int aos2_extractor_bench_fast(aos2_extractor_t *e, int iters, float *avg_ms);
int aos2_extractor_bench_describe(aos2_extractor_t *e, int iters, float *avg_ms);

/* ------------------------------------------------------------------------------------------
 * ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>   
 * (include/ORBVocabulary.h:31-32, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h)
 * ------------------------------------------------------------------------------------------ */
typedef struct aos2_vocabulary aos2_vocabulary_t;

/* DBoW2 enums (Thirdparty/DBoW2/DBoW2/BowVector.h:36-53) */
#define AOS2_WEIGHT_TF_IDF 0
#define AOS2_WEIGHT_TF 1
#define AOS2_WEIGHT_IDF 2
#define AOS2_WEIGHT_BINARY 3
#define AOS2_SCORING_L1_NORM 0
#define AOS2_SCORING_L2_NORM 1
#define AOS2_SCORING_CHI_SQUARE 2
#define AOS2_SCORING_KL 3
#define AOS2_SCORING_BHATTACHARYYA 4
#define AOS2_SCORING_DOT_PRODUCT 5

int aos2_vocabulary_create(int device, aos2_vocabulary_t **out);
void aos2_vocabulary_destroy(aos2_vocabulary_t *v);
/* bool loadFromBinaryFile(filename) :1456-1496 / loadFromTextFile :1351-1431 (src/System.cc:89-92),
 * saveToBinaryFile :1500-1521.  Both loaders reproduce the reference's `while(!f.eof())` tail
 * (binary: the last record is appended twice; text: a final newline yields one more, childless,
 * zero-weight child of the root) -- see DESIGN.md §5.5. */
int aos2_vocabulary_load_binary(aos2_vocabulary_t *v, const char *filename);
int aos2_vocabulary_load_text(aos2_vocabulary_t *v, const char *filename);
int aos2_vocabulary_save_binary(const aos2_vocabulary_t *v, const char *filename);
/* Programmatic construction (no reference equivalent; the vocabulary file is absent):
 * record i describes node i+1 exactly like one record of the binary file (parent id, 32-byte
 * descriptor, weight, leaf flag); parents must precede their children. */
int aos2_vocabulary_set_nodes(aos2_vocabulary_t *v, int k, int L, int scoring, int weighting,
                              int n_nodes, const int32_t *parent, const uint8_t *desc,
                              const double *weight, const uint8_t *is_leaf);
int aos2_vocabulary_k(const aos2_vocabulary_t *v);          /* getBranchingFactor() */
int aos2_vocabulary_levels(const aos2_vocabulary_t *v);     /* getDepthLevels() */
int aos2_vocabulary_scoring(const aos2_vocabulary_t *v);    /* getScoringType() */
int aos2_vocabulary_weighting(const aos2_vocabulary_t *v);  /* getWeightingType() */
int aos2_vocabulary_nodes(const aos2_vocabulary_t *v);      /* m_nodes.size() */
unsigned aos2_vocabulary_size(const aos2_vocabulary_t *v);  /* size() = number of words */
int aos2_vocabulary_empty(const aos2_vocabulary_t *v);      /* empty() */

/* void transform(const vector<TDescriptor>& features, BowVector&, FeatureVector&, int levelsup)
 * :1140-1187 (Frame::ComputeBoW src/Frame.cc:424-431 and KeyFrame::ComputeBoW call it with
 * levelsup = 4).  desc: n x 32 bytes, host.  The two std::maps are returned as key-ascending arrays:
 *   BowVector     bow_word[*n_bow], bow_value[*n_bow]                       (capacity n each)
 *   FeatureVector fv_node[*n_fv], fv_off[*n_fv + 1], fv_idx[fv_off[*n_fv]]  (capacity n, n+1, n)
 * -- the CSR layout aos2_bow_pair_t takes.  word_of / node_of (optional, n each): per-feature word id
 * and node id at level L - levelsup (transform(feature, id, weight, &nid, levelsup) :1215-1260).
 * An empty vocabulary clears both outputs and returns AOS2_OK (:1147-1150). */
int aos2_vocabulary_transform(aos2_vocabulary_t *v, const uint8_t *desc, int n, int levelsup,
                              uint32_t *bow_word, double *bow_value, int *n_bow, int32_t *fv_node,
                              int32_t *fv_off, int32_t *fv_idx, int *n_fv, uint32_t *word_of,
                              uint32_t *node_of);
/* Batched device-resident form: d_desc [batch][cap][32] and d_n [batch] as written by
 * aos2_extractor_extract_batch_device; outputs [batch][cap] (d_fv_off: [batch][cap + 1]),
 * counts [batch].  d_word_of / d_node_of may be NULL.  cap <= 8192. */
int aos2_vocabulary_transform_device(aos2_vocabulary_t *v, int batch, const uint8_t *d_desc,
                                     const int32_t *d_n, int cap, int levelsup,
                                     uint32_t *d_bow_word, double *d_bow_value, int32_t *d_n_bow,
                                     int32_t *d_fv_node, int32_t *d_fv_off, int32_t *d_fv_idx,
                                     int32_t *d_n_fv, uint32_t *d_word_of, uint32_t *d_node_of);
float aos2_vocabulary_last_device_ms(const aos2_vocabulary_t *v);
/* the handle's hipStream_t (the transforms run on it): lets a caller order a transform behind the producer of d_desc on
 * the device, e.g. aos2_extractor_stream_wait(e, aos2_vocabulary_stream(v)) after enqueuing the extraction */
void *aos2_vocabulary_stream(aos2_vocabulary_t *v);
/* double score(const BowVector&, const BowVector&) :1191-1195 for L1_NORM
 * (L1Scoring::score, Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-72); host scalar helper. */
int aos2_vocabulary_score(const aos2_vocabulary_t *v, const uint32_t *w1, const double *v1, int n1,
                          const uint32_t *w2, const double *v2, int n2, double *score);

/* ------------------------------------------------------------------------------------------
 * ORBmatcher  (include/ORBmatcher.h:37-102, src/ORBmatcher.cc)
 * The C++ shim snapshots the Frame / KeyFrame / MapPoint fields each method reads into the SoA
 * views below and maps the returned indices back to MapPoint*.
 * ------------------------------------------------------------------------------------------ */
#define AOS2_TH_HIGH 100     /* ORBmatcher::TH_HIGH  src/ORBmatcher.cc:37 */
#define AOS2_TH_LOW 50       /* ORBmatcher::TH_LOW   :38 */
#define AOS2_HISTO_LENGTH 30 /* ORBmatcher::HISTO_LENGTH :39 */
#define AOS2_GRID_COLS 64    /* FRAME_GRID_COLS include/Frame.h:38 */
#define AOS2_GRID_ROWS 48    /* FRAME_GRID_ROWS include/Frame.h:37 */

typedef struct aos2_matcher aos2_matcher_t;

/* ORBmatcher::ORBmatcher(float nnratio=0.6, bool checkOri=true)  src/ORBmatcher.cc:41-43 */
int aos2_matcher_create(float nnratio, int check_orientation, int device, aos2_matcher_t **out);
void aos2_matcher_destroy(aos2_matcher_t *m);

/* device time (ms, HIP events) of the kernels of the last search_* call on this handle */
float aos2_matcher_last_device_ms(const aos2_matcher_t *m);

/* static int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)  :1647-1663.
 * Host scalar helper (pure); the device kernels use xor + popcount of the same 256 bits. */
int aos2_descriptor_distance(const uint8_t *a, const uint8_t *b);

/* Brute-force Hamming: for every query descriptor the best and second best train descriptor.
 * q[nq][32], t[nt][32] host memory.  best_idx/best_dist/second_dist: nq entries.
 * Ties: lowest train index wins (strict '<' update order of the reference loops). */
int aos2_matcher_hamming_best2(aos2_matcher_t *m, const uint8_t *q, int nq, const uint8_t *t,
                               int nt, int32_t *best_idx, int32_t *best_dist,
                               int32_t *second_dist);
/* device-resident variant + kernel timing (ms, HIP events) for the roofline/throughput report */
int aos2_matcher_hamming_best2_device(aos2_matcher_t *m, const uint8_t *d_q, int nq,
                                      const uint8_t *d_t, int nt, int32_t *d_best_idx,
                                      int32_t *d_best_dist, int32_t *d_second_dist, int iters,
                                      float *avg_ms);

typedef struct {
    int32_t n_kf, n_f;
    const uint8_t *desc_kf;    /* pKF->mDescriptors, n_kf x 32 */
    const uint8_t *desc_f;     /* F.mDescriptors, n_f x 32 */
    const uint8_t *kf_has_mp;  /* n_kf: vpMapPointsKF[i] != NULL && !isBad() (:194-198) */
    const float *angle_kf;     /* pKF->mvKeysUn[i].angle */
    const float *angle_f;      /* F.mvKeys[j].angle */
    /* DBoW2::FeatureVector (std::map<NodeId, vector<unsigned>>) flattened to CSR with node ids
     * ascending: Thirdparty/DBoW2/DBoW2/FeatureVector.h:21-22 */
    int32_t n_nodes_kf, n_nodes_f;
    const int32_t *node_id_kf, *node_off_kf, *node_idx_kf;
    const int32_t *node_id_f, *node_off_f, *node_idx_f;
} aos2_bow_pair_t;

/* int ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)  src/ORBmatcher.cc:159-288.
 * `n_pairs` independent (KF, F) pairs are matched in one launch (one workgroup per pair).
 * match_f[p] (n_f entries): index of the KF feature whose MapPoint is assigned to frame feature j,
 * or -1 (NULL).  nmatches[p] = return value. */
int aos2_matcher_search_by_bow(aos2_matcher_t *m, const aos2_bow_pair_t *pairs, int n_pairs,
                               int32_t *const *match_f, int32_t *nmatches);

/* The same search on frames whose descriptors and keys already live in HBM (a device-resident Frames batch, the
 * extractor's device outputs): desc_kf / desc_f / angle_kf / angle_f of the pairs are DEVICE arrays and are used in place
 * (32 KB + 4 KB per frame that do not cross PCIe); kf_has_mp, the FeatureVector CSRs (the host merge-joins them, :181-258)
 * and the results are host arrays, a few KB per pair.  The caller has completed (or ordered this handle's work behind) the
 * producers of the device arrays. */
int aos2_matcher_search_by_bow_device(aos2_matcher_t *m, const aos2_bow_pair_t *pairs, int n_pairs,
                                      int32_t *const *match_f, int32_t *nmatches);

/* The per-keyframe front part of Tracking::TrackReferenceKeyFrame (src/Tracking.cc:858-866) for `n_frames` (reference
 * keyframe b, frame b) pairs whose members live in HBM as the extractor and aos2_vocabulary_transform_device wrote them:
 * descriptors [n][cap][32], keypoints [n][cap] (mvKeys: the angle is read), counts [n], FeatureVectors fv_node [n][cap],
 * fv_off [n][cap + 1], fv_idx [n][cap], n_fv [n] -- all DEVICE arrays; kf_has_mp [n][cap] is a HOST array (map state:
 * vpMapPointsKF[i] != NULL && !isBad()).  The call copies the FeatureVectors and counts to the host (a few KB per frame),
 * merge-joins them there, and runs aos2_matcher_search_by_bow_device.  match_f: HOST [n][cap], nmatches: HOST [n]. */
typedef struct {
    int32_t n_frames, cap;
    const uint8_t *d_desc_kf;          const aos2_keypoint_t *d_kps_kf;  const int32_t *d_n_kf;
    const uint8_t *d_desc_f;           const aos2_keypoint_t *d_kps_f;   const int32_t *d_n_f;
    const uint8_t *kf_has_mp;
    const int32_t *d_kf_fv_node, *d_kf_fv_off, *d_kf_fv_idx, *d_kf_n_fv;
    const int32_t *d_f_fv_node, *d_f_fv_off, *d_f_fv_idx, *d_f_n_fv;
} aos2_bow_frames_t;
int aos2_matcher_search_by_bow_frames(aos2_matcher_t *m, const aos2_bow_frames_t *q, int32_t *match_f,
                                      int32_t *nmatches);

/* int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12)
 * src/ORBmatcher.cc:522-655 (LoopClosing.cc:? / Tracking relocalisation candidates).
 * has_mp*: GetMapPointMatches()[i] != NULL && !isBad().  match12[p] (n1 entries): index of the KF2
 * feature whose MapPoint becomes vpMatches12[i], or -1 (NULL). */
typedef struct {
    int32_t n1, n2;
    const uint8_t *desc1, *desc2;      /* mDescriptors */
    const uint8_t *has_mp1, *has_mp2;
    const float *angle1, *angle2;      /* mvKeysUn[i].angle */
    int32_t n_nodes1, n_nodes2;        /* mFeatVec as CSR, node ids ascending */
    const int32_t *node_id1, *node_off1, *node_idx1;
    const int32_t *node_id2, *node_off2, *node_idx2;
} aos2_bow_kf_pair_t;
int aos2_matcher_search_by_bow_kf(aos2_matcher_t *m, const aos2_bow_kf_pair_t *pairs, int n_pairs,
                                  int32_t *const *match12, int32_t *nmatches);

/* int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12,
 *         vector<pair<size_t,size_t>> &vMatchedPairs, const bool bOnlyStereo)   src/ORBmatcher.cc:657-823
 * (LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:? calls it once per neighbour keyframe:
 * `n_pairs` (pKF1, pKF2_k) problems go in one launch).  has_mp*: GetMapPoint(i) != NULL.
 * F12: row-major 3x3 (LocalMapping::ComputeF12); ex, ey: the epipole of :664-670.
 * match12[p] (n1 entries) = vMatches12 (KF2 feature index or -1); vMatchedPairs is its non-negative
 * entries in ascending i (:811-820).  nmatches[p] = return value. */
typedef struct {
    int32_t n1, n2;
    const uint8_t *desc1, *desc2;
    const uint8_t *has_mp1, *has_mp2;
    const float *x1, *y1, *angle1, *u_right1;   /* pKF1->mvKeysUn[i].pt / .angle, mvuRight */
    const float *x2, *y2, *angle2, *u_right2;
    const int32_t *octave2;                     /* pKF2->mvKeysUn[i].octave */
    const float *scale_factors2, *level_sigma2_2;  /* pKF2->mvScaleFactors / mvLevelSigma2 */
    int32_t n_levels2;
    float F12[9];
    float ex, ey;
    int32_t n_nodes1, n_nodes2;                 /* mFeatVec as CSR */
    const int32_t *node_id1, *node_off1, *node_idx1;
    const int32_t *node_id2, *node_off2, *node_idx2;
} aos2_triang_pair_t;
int aos2_matcher_search_for_triangulation(aos2_matcher_t *m, const aos2_triang_pair_t *pairs,
                                          int n_pairs, int only_stereo, int32_t *const *match12,
                                          int32_t *nmatches);

/* void MapPoint::ComputeDistinctiveDescriptors()  src/MapPoint.cc:275-340, for a batch of map points:
 * the descriptors observed for point p (rows of the non-bad keyframes, in std::map order, :297-303) are
 * desc[off[p] .. off[p+1]).  best_idx[p] = index inside that list of the descriptor with the least
 * median distance to the others (first one on ties), -1 for an empty list. */
int aos2_compute_distinctive_descriptors(aos2_matcher_t *m, int n_points, const int32_t *off,
                                         const uint8_t *desc, int32_t *best_idx);

typedef struct {
    int32_t n_f;
    const uint8_t *desc_f;      /* F.mDescriptors */
    const float *kp_x, *kp_y;   /* F.mvKeysUn[i].pt */
    const int32_t *kp_octave;   /* F.mvKeysUn[i].octave */
    const float *kp_angle;      /* F.mvKeysUn[i].angle */
    const float *u_right;       /* F.mvuRight */
    const float *scale_factors; /* F.mvScaleFactors */
    int32_t n_levels;
    float min_x, min_y, max_x, max_y; /* F.mnMinX, mnMinY, mnMaxX, mnMaxY */
    float grid_w_inv, grid_h_inv;     /* F.mfGridElementWidthInv / HeightInv */
    /* F.mGrid[64][48] as CSR, cell = ix*48 + iy (src/Frame.cc:259-274) */
    const int32_t *grid_off, *grid_idx;
    /* per feature: 0 = mvpMapPoints[i]==NULL, 1 = map point with Observations()==0,
     * 2 = map point with Observations()>0 (the skip test of :87-89 / :1403-1405) */
    const uint8_t *f_mp_state;
} aos2_frame_view_t;

typedef struct {
    int32_t n_mp;
    const uint8_t *track_in_view; /* pMP->mbTrackInView && !pMP->isBad() (:52-56) */
    const int32_t *pred_level;    /* mnTrackScaleLevel */
    const float *view_cos;        /* mTrackViewCos */
    const float *proj_x, *proj_y, *proj_xr; /* mTrackProjX / Y / XR */
    const uint8_t *desc;          /* pMP->GetDescriptor(), n_mp x 32 */
    const uint8_t *has_obs;       /* pMP->Observations() > 0 */
} aos2_proj_mp_t;

/* int ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, const float th)
 * src/ORBmatcher.cc:45-129.  match_f[n_f]: index into the map point list newly assigned to
 * F.mvpMapPoints[j], or -1 = entry unchanged.  Any number of map points: beyond n_mp x n_f x 8 B = 64 MB the scratch
 * for the candidates is sized from the real window populations (two passes) instead of n_mp x n_f slots.
 * AOS2_ERR_ARG (before anything is uploaded) if the frame view or the point set is inconsistent: non-monotone grid
 * offsets, grid entries or pyramid levels out of range, NULL arrays. */
int aos2_matcher_search_by_projection(aos2_matcher_t *m, const aos2_frame_view_t *f,
                                      const aos2_proj_mp_t *p, float th, int32_t *match_f,
                                      int32_t *nmatches);

/* The same search for `n_problems` independent (frame, local map points) problems in one launch (one
 * greedy-resolve wave per frame, all frames in flight together): replaying or relocalising many frames.
 * match_f[i] has frames[i].n_f entries.  The candidate-entry pool is budgeted at 512 features per search
 * window; AOS2_ERR_CAPACITY if the windows of a batch hold more (then call the per-frame entry point). */
int aos2_matcher_search_by_projection_batch(aos2_matcher_t *m, const aos2_frame_view_t *frames,
                                            const aos2_proj_mp_t *problems, int n_problems, float th,
                                            int32_t *const *match_f, int32_t *nmatches);

typedef struct {
    int32_t n_last;
    const uint8_t *last_valid;  /* LastFrame.mvpMapPoints[i] && !LastFrame.mvbOutlier[i] */
    const float *world_pos;     /* pMP->GetWorldPos(), n_last x 3 float32 */
    const uint8_t *desc;        /* pMP->GetDescriptor(), n_last x 32 */
    const int32_t *last_octave; /* LastFrame.mvKeys[i].octave */
    const float *last_angle;    /* LastFrame.mvKeysUn[i].angle */
    const uint8_t *has_obs;     /* pMP->Observations() > 0 */
    float Tcw[16], Tlw[16];     /* CurrentFrame.mTcw, LastFrame.mTcw: row-major 4x4 float32 */
    float fx, fy, cx, cy, mb, mbf;
} aos2_proj_last_t;

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th,
 * const bool bMono)  src/ORBmatcher.cc:1328-1470.  match_f[n_f]: last-frame feature index whose
 * MapPoint is assigned, -1 = unchanged, -2 = reset to NULL by the rotation-histogram cull. */
int aos2_matcher_search_by_projection_last(aos2_matcher_t *m, const aos2_frame_view_t *cur,
                                           const aos2_proj_last_t *p, float th, int mono,
                                           int32_t *match_f, int32_t *nmatches);

/* Projection family: a set of map points projected into a keyframe / frame (aos2_frame_view_t; for a
 * KeyFrame the view carries mvKeysUn, mvuRight, mDescriptors, mGrid, mnMin/Max*, mvScaleFactors; f_mp_state
 * is used by the two greedy searches only).  Rotations are row-major 3x3.  The few cv::Mat lines that
 * decompose Scw or build sR12 / sR21 / t21 (:299-304, :986-991, :1116-1120) stay in the caller: R, t, Ow
 * (and R2, t2) are inputs, so no OpenCV rounding behaviour is guessed on the device. */
typedef struct {
    int32_t n_pts;
    const uint8_t *valid;            /* the per-point gate of the reference loop head, see each function */
    const float *pos;                /* 3 per point: GetWorldPos() */
    const float *max_dist, *min_dist;   /* MapPoint::mfMaxDistance / mfMinDistance, RAW (include/MapPoint.h:149-150): the range gate applies
                                      * Get{Max,Min}DistanceInvariance()'s 1.2f / 0.8f (src/MapPoint.cc:413-423) on the device and
                                      * MapPoint::PredictScale (:427-459) divides the raw maximum by the distance */
    const float *normal;             /* 3 per point: GetNormal() (unused by SearchBySim3 and the reloc search) */
    const uint8_t *desc;             /* 32 per point: GetDescriptor() */
    const float *q_angle;            /* reloc search only: pKF->mvKeysUn[i].angle */
    float R[9], t[3], Ow[3];         /* Rcw, tcw, camera centre */
    float R2[9], t2[3];              /* SearchBySim3 only: sR21, t21 (KF1 -> KF2) or sR12, t12 */
    float fx, fy, cx, cy, bf;        /* bf: Fuse(pKF, vpMapPoints) only (ur = u - bf*invz, :870) */
    float log_scale_factor;          /* mfLogScaleFactor of the target (MapPoint::PredictScale) */
    const float *inv_level_sigma2;   /* mvInvLevelSigma2 of the target, Fuse(pKF, vpMapPoints) only */
    float th;
} aos2_proj_points_t;

/* The search part of int ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th)  :825-975  (sim3 = 0,
 * valid[i] = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF)) and of Fuse(KeyFrame*, cv::Mat Scw,
 * vpPoints, th, vpReplacePoint)  :977-1100 (sim3 = 1, valid[i] = !isBad() && !spAlreadyFound.count(pMP)).
 * best_idx[i] = the keyframe feature the point fuses with (bestDist <= TH_LOW), -1 otherwise;
 * best_dist[i] its distance; *n_fused = the count.  The Replace / AddObservation / vpReplacePoint
 * bookkeeping of :950-969 / :1079-1093 consumes best_idx on the host (it only reads map state). */
int aos2_matcher_fuse(aos2_matcher_t *m, const aos2_frame_view_t *kf, const aos2_proj_points_t *p,
                      int sim3, int32_t *best_idx, int32_t *best_dist, int32_t *n_fused);

/* int ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*> &vpPoints,
 *         vector<MapPoint*> &vpMatched, int th)  :290-403.  valid[i] = !isBad() && !spAlreadyFound.count;
 * kf->f_mp_state[idx] != 0 <=> vpMatched[idx] != NULL on entry.  match_f[idx] = index of the point
 * written to vpMatched[idx], -1 = unchanged. */
int aos2_matcher_search_by_projection_kf(aos2_matcher_t *m, const aos2_frame_view_t *kf,
                                         const aos2_proj_points_t *p, int32_t *match_f,
                                         int32_t *nmatches);

/* int ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)  :1102-1326.
 * p12: one entry per KF1 feature (valid = pMP && !vbAlreadyMatched1[i] && !isBad(); R, t = R1w, t1w;
 * R2, t2 = sR21, t21), projected into kf2; p21 the converse (R2w, t2w; sR12, t12) into kf1.
 * match12[i1] = KF2 feature index whose map point becomes vpMatches12[i1], -1 = unchanged. */
int aos2_matcher_search_by_sim3(aos2_matcher_t *m, const aos2_frame_view_t *kf1,
                                const aos2_frame_view_t *kf2, const aos2_proj_points_t *p12,
                                const aos2_proj_points_t *p21, int32_t *match12, int32_t *n_found);

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*>
 *         &sAlreadyFound, const float th, const int ORBdist)  :1472-1599 (relocalisation).
 * Points = pKF->GetMapPointMatches() (valid = pMP && !isBad() && !sAlreadyFound.count(pMP));
 * frame->f_mp_state[i2] != 0 <=> CurrentFrame.mvpMapPoints[i2] != NULL on entry.
 * match_f[i2]: keyframe feature index whose map point is assigned, -1 unchanged, -2 reset to NULL by the
 * rotation check. */
int aos2_matcher_search_by_projection_reloc(aos2_matcher_t *m, const aos2_frame_view_t *frame,
                                            const aos2_proj_points_t *p, int orb_dist,
                                            int32_t *match_f, int32_t *nmatches);

/* bool Frame::isInFrustum(MapPoint *pMP, float viewingCosLimit)  src/Frame.cc:298-354, for all the local map
 * points of Tracking::SearchLocalPoints in one call.  p: pos / max_dist / min_dist / normal per point (valid,
 * desc, q_angle unused), R, t, Ow = mRcw, mtcw, mOw, bf = mbf; bounds = mnMinX.. (closed, :319-322).
 * Outputs are the members the search reads: mbTrackInView, mTrackProjX, mTrackProjY, mTrackProjXR,
 * mnTrackScaleLevel, mTrackViewCos -- i.e. the arrays of aos2_proj_mp_t. */
int aos2_frame_is_in_frustum(aos2_matcher_t *m, const aos2_proj_points_t *p, float min_x, float max_x,
                             float min_y, float max_y, int n_levels, float viewing_cos_limit,
                             uint8_t *track_in_view, float *proj_x, float *proj_y, float *proj_xr,
                             int32_t *pred_level, float *view_cos);

/* void Frame::AssignFeaturesToGrid()  src/Frame.cc:259-274 (+ PosInGrid :411-424): mGrid[64][48] as CSR,
 * cell = ix * 48 + iy, features of a cell in ascending index (= push_back order) -- the grid_off / grid_idx
 * arrays of aos2_frame_view_t.  kp_x / kp_y = mvKeysUn[i].pt.  *n_in_grid = features that fall inside the grid. */
int aos2_frame_assign_features_to_grid(aos2_matcher_t *m, int n, const float *kp_x, const float *kp_y,
                                       float min_x, float min_y, float grid_w_inv, float grid_h_inv,
                                       int32_t *grid_off, int32_t *grid_idx, int32_t *n_in_grid);

/* void Frame::ComputeStereoFromRGBD(const cv::Mat &imDepth)  src/Frame.cc:672-693 (the RGB-D configs, e.g.
 * TUM).  kp_x / kp_y = mvKeys[i].pt (distorted, index the depth image by truncation), kpun_x = mvKeysUn[i].pt.x,
 * depth_img: CV_32F, `stride` floats per row.  Outputs mvuRight / mvDepth (-1 where the depth is not positive). */
int aos2_frame_stereo_from_rgbd(aos2_matcher_t *m, int n, const float *kp_x, const float *kp_y,
                                const float *kpun_x, const float *depth_img, int w, int h, int stride,
                                float mbf, float *u_right, float *depth);

/* int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched,
 *         vector<int> &vnMatches12, int windowSize=10)  src/ORBmatcher.cc:405-520 (monocular bootstrap,
 * Tracking::MonocularInitialization).  f2 = view of F2 (grid, mvKeysUn, descriptors); F1 enters through
 * desc1 / octave1 / angle1 (mvKeysUn[i].octave / .angle) and prev_xy = vbPrevMatched (2 floats per F1
 * feature).  match12[n1] = vnMatches12.  The update of :512-515 (vbPrevMatched[i1] = F2 keypoint of the
 * match) is one line in the caller.  n1 < 65535, f2->n_f <= 15000. */
int aos2_matcher_search_for_initialization(aos2_matcher_t *m, const aos2_frame_view_t *f2, int n1,
                                           const uint8_t *desc1, const int32_t *octave1,
                                           const float *angle1, const float *prev_xy,
                                           int window_size, int32_t *match12, int32_t *nmatches);



/* ------------------------------------------------------------------------------------------
 * Optimizer::LocalBundleAdjustment  (include/Optimizer.h:45, src/Optimizer.cc:454-779)
 * The C++ shim gathers the local window from the KeyFrame/MapPoint pointer graph (:457-505)
 * and emits vertices/edges in the reference's order; this call runs :507-744 (g2o graph,
 * 5 + 10 Levenberg-Marquardt iterations with the outlier pass in between) on the device and
 * returns poses/points in the float32 form the reference writes back (:763-778).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_poses;   /* local keyframes first or in any order; fixed flag decides */
    int32_t n_points;
    int32_t n_edges;
    const float *pose_Tcw;      /* n_poses x 16: KeyFrame::GetPose() row-major float32 4x4 */
    const uint8_t *pose_fixed;  /* n_poses: fixed camera, or mnId==0 (:530,:543) */
    const int64_t *pose_id;     /* KeyFrame::mnId (Hessian ordering) */
    const float *point_xyz;     /* n_points x 3: MapPoint::GetWorldPos() float32 */
    const int64_t *point_id;    /* MapPoint::mnId */
    const int32_t *edge_pose;   /* n_edges: index into poses */
    const int32_t *edge_point;  /* n_edges: index into points */
    const float *edge_obs;      /* n_edges x 3: kpUn.pt.x, kpUn.pt.y, mvuRight (ignored for mono) */
    const uint8_t *edge_stereo; /* n_edges: mvuRight >= 0 (:595) */
    const float *edge_inv_sigma2; /* n_edges: pKFi->mvInvLevelSigma2[kpUn.octave] */
    float fx, fy, cx, cy, bf;   /* pKFi->fx ... pKFi->mbf (float members of KeyFrame) */
    const volatile uint8_t *stop_flag; /* bool* pbStopFlag, may be NULL; polled between iterations */
    int32_t iters_first, iters_second; /* 5 and 10 (:661,:708) */
} aos2_lba_problem_t;

typedef struct {
    float *pose_Tcw;         /* n_poses x 16 out: Converter::toCvMat(SE3Quat) (:767) */
    float *point_xyz;        /* n_points x 3 out (:776) */
    uint8_t *edge_outlier;   /* n_edges out: observation to erase (:712-744) */
    double *edge_chi2;       /* n_edges out (may be NULL): e->chi2() at the end */
    int32_t iters_done_first, iters_done_second;  /* iterations SparseOptimizer::optimize() ran (its return value) */
    double final_chi2;       /* robust chi2 of the active edges after the last accepted step */
    double final_lambda;
    float ms_device;         /* device time of the whole call (HIP events; all windows of a batch) */
    int32_t status;          /* AOS2_OK or AOS2_ERR_STOPPED (flag set on entry: outputs equal inputs) */
    int32_t trials_first, trials_second;  /* Levenberg-Marquardt trial steps (solves) of the two optimisations */
    int32_t polls;           /* evaluations of pbStopFlag (SparseOptimizer::terminate()), the entry check included */
    int32_t stop_poll;       /* the evaluation that first saw the flag set, 0 = never */
} aos2_lba_result_t;

typedef struct aos2_lba aos2_lba_t;
int aos2_lba_create(int device, aos2_lba_t **out);
void aos2_lba_destroy(aos2_lba_t *s);
/* void Optimizer::LocalBundleAdjustment(KeyFrame*, bool* pbStopFlag, Map*)  numerical part.
 * Returns AOS2_OK, AOS2_ERR_STOPPED if *stop_flag was set on entry (early return, :656-658;
 * outputs then equal inputs), or an error.
 * The whole procedure (both optimisations, the outlier pass, the inlier check) runs on the device without a host
 * round trip per Levenberg-Marquardt trial; *stop_flag is forwarded to the device while the call waits and is
 * evaluated exactly where g2o evaluates terminate() (optimization_algorithm_levenberg.cpp:149,
 * sparse_optimizer.cpp:372) and where Optimizer.cc:663-666 reads it.  Any number of free keyframes: the reduced camera
 * system is factorised in LDS up to 21 of them, in device memory beyond. */
int aos2_lba_solve(aos2_lba_t *s, const aos2_lba_problem_t *p, aos2_lba_result_t *r);
/* `n_problems` independent windows (several maps, or an offline pass over many windows; SURVEY.md section 8(e):
 * LocalBA = replicas only) in one call: every kernel covers all windows, so their latency-bound Levenberg-Marquardt
 * chains overlap on the device.  results[i].status carries the per-window AOS2_OK / AOS2_ERR_STOPPED; the return
 * value is AOS2_OK unless an argument or HIP error occurred.
 * A handle serves one call at a time (concurrent solves: one handle per calling thread); it keeps its device arena, the
 * host-side structure buffers and the worker threads of the per-window host work between calls. */
int aos2_lba_solve_batch(aos2_lba_t *s, const aos2_lba_problem_t *problems, aos2_lba_result_t *results,
                         int n_problems);
/* Worker threads of a batch's per-window host work (index structures, staging); 0 = default: AOS2_LBA_HOST_THREADS, else
 * min(32, host cores / ranks of the node (LOCAL_WORLD_SIZE or WORLD_SIZE)), never more than windows in the call.  No reference
 * equivalent (g2o builds its structure on the calling thread, sparse_optimizer.cpp:354-372). */
int aos2_lba_set_host_threads(aos2_lba_t *s, int n);
/* Window groups of a batch: 2 = the windows are dealt to two groups whose programs run on streams of their own, half a trial apart,
 * so that one group's reduced systems (one workgroup per window) are factorised while the other's landmark kernels fill the device --
 * the faster form for a batch that has the device to itself (64 mixed windows 5.7 -> 5.2 ms); 1 = one program for all windows -- the
 * faster form when other work shares the device (two handles solving side by side beside the tracking kernels: 61 k against 54 k
 * frames/s of bench.py's composite); 0 = default: 2 for calls of >= 16 windows.  Results do not depend on it.  No reference equivalent. */
int aos2_lba_set_window_groups(aos2_lba_t *s, int n);
/* The device program of the last solve of this handle: `trial_slots` = Levenberg-Marquardt trials enqueued in all (the first round
 * holds as many as there are iterations, for every window of the call; a window whose steps were not all accepted is not finished
 * then and gets a continuation round, sized for what it still needs, with the other unfinished windows only), `host_rounds` = times
 * the host waited for the device (1 = no continuation).  aos2_lba_last_window_slots: the trial slots summed over the windows each
 * round covered -- divided by the trials the windows needed, the lock-step cost of the batch (1.0 = no launch covered a window that
 * had nothing left to do). */
int aos2_lba_last_program(const aos2_lba_t *s, int32_t *trial_slots, int32_t *host_rounds);
int aos2_lba_last_window_slots(const aos2_lba_t *s, int64_t *window_slots);
/* Measurement hook (no device, no reference equivalent): the HOST part of aos2_lba_solve_batch for `n_problems` windows -- the
 * per-window index structures and the staging copies -- on `threads` worker threads; wall milliseconds of the two phases. */
int aos2_lba_debug_host_phase(const aos2_lba_problem_t *problems, int n_problems, int threads, double *build_ms,
                              double *stage_ms);
/* Test hook (no reference equivalent): the stop flag counts as set from its `poll`-th evaluation on (1 = the entry
 * check), as if another thread had set it at that moment; 0 switches the hook off.  Used with
 * aos2_lba_result_t.stop_poll to reproduce an asynchronous abort deterministically. */
int aos2_lba_debug_stop_at_poll(aos2_lba_t *s, int poll);

/* ------------------------------------------------------------------------------------------
 * Optimizer::PoseOptimization  (include/Optimizer.h:46, src/Optimizer.cc:239-452) -- SURVEY §8(f)
 * rank 1, called 1-3 times per frame right after each matcher call (Tracking.cc:870,994,1039).
 * One workgroup per frame runs the whole procedure on the device (4 rounds of up to 10
 * Levenberg-Marquardt iterations on the 6x6 system, outlier reclassification in between);
 * `n_problems` independent frames are solved in one launch.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n;                  /* features with a map point (nInitialCorrespondences) */
    const float *Xw;            /* n x 3: pMP->GetWorldPos() */
    const float *obs;           /* n x 3: mvKeysUn[i].pt.x, .pt.y, mvuRight[i] */
    const uint8_t *stereo;      /* n: mvuRight[i] >= 0 */
    const float *inv_sigma2;    /* n: mvInvLevelSigma2[octave] */
    float fx, fy, cx, cy, bf;   /* pFrame->fx .. pFrame->mbf */
    float Tcw[16];              /* pFrame->mTcw, row-major float32 4x4 */
} aos2_pose_problem_t;

typedef struct {
    float Tcw[16];              /* out: pose written by pFrame->SetPose (:446) */
    uint8_t *outlier;           /* out, n entries, caller-allocated: pFrame->mvbOutlier of the features */
    int32_t n_bad;              /* pFrame->nBadPoseOpt */
    int32_t n_inliers;          /* return value: nInitialCorrespondences - nBad (0 if n < 3) */
} aos2_pose_result_t;

/* int Optimizer::PoseOptimization(Frame *pFrame) for a batch of frames; `s` provides device + stream */
int aos2_pose_optimization(aos2_lba_t *s, const aos2_pose_problem_t *problems, aos2_pose_result_t *results,
                           int n_problems);
/* device time (ms, HIP events) of the kernel of the last aos2_pose_optimization call */
float aos2_pose_optimization_last_device_ms(const aos2_lba_t *s);

/* ------------------------------------------------------------------------------------------
 * Device-resident frame batches: the per-frame chain of Tracking::Track (src/Tracking.cc:862-870, 963-1027, 1302-1352)
 *   ORBextractor::operator() -> Frame::Frame -> SearchByProjection(Current, Last) -> PoseOptimization
 *   -> SearchLocalPoints (isInFrustum + SearchByProjection(F, vpMapPoints)) -> PoseOptimization
 * with the Frame members and the MapPoint table kept in HBM between the calls, so that no
 * keypoint, descriptor, match or pose crosses PCIe inside the chain.  A batch is B independent Frames (replaying
 * or relocalising many frames, several cameras, or B (LastFrame, CurrentFrame) pairs); every call is enqueued on the
 * batch's stream and returns without waiting; aos2_frames_wait() waits and reports errors.
 * ------------------------------------------------------------------------------------------ */
typedef struct aos2_frames aos2_frames_t;

/* the members of MapPoint the chain reads (include/MapPoint.h), as a table in device memory; Frames refer to its rows */
typedef struct {
    int32_t n;
    const float *pos;        /* n x 3  GetWorldPos() */
    const uint8_t *desc;     /* n x 32 GetDescriptor() */
    const uint8_t *has_obs;  /* n      Observations() > 0 */
    const float *normal;     /* n x 3  GetNormal()                       (SearchLocalPoints only) */
    const float *min_dist;   /* n      mfMinDistance, raw (gate: 0.8f * it)  (SearchLocalPoints only) */
    const float *max_dist;   /* n      mfMaxDistance, raw (gate: 1.2f * it; PredictScale: it / dist)  (SearchLocalPoints only) */
} aos2_map_points_dev_t;

/* batch frames of at most cap (<= 5632) keypoints each */
int aos2_frames_create(int device, int batch, int cap, aos2_frames_t **out);
void aos2_frames_destroy(aos2_frames_t *f);
void *aos2_frames_stream(aos2_frames_t *f);   /* the batch's hipStream_t */
/* Device-side ordering, no host wait: everything enqueued on the batch after this call runs behind the work enqueued on
 * `hip_stream` (a hipStream_t, NULL = the null stream) so far.  For the inputs the caller produces on a stream of its own:
 * the MapPoint table, d_local, d_Tcw, the depth images.  (The batches order themselves where they read each other:
 * aos2_frames_build behind the extractor's batch, aos2_frames_search_by_projection_last behind everything enqueued on
 * `last`'s stream -- in a tracking loop the previous CurrentFrame batch becomes LastFrame while its PoseOptimization is
 * still in flight.) */
int aos2_frames_wait_for_stream(aos2_frames_t *f, void *hip_stream);
/* waits for everything enqueued on the batch; AOS2_ERR_CAPACITY if a frame's search windows did not fit the entry pool */
int aos2_frames_wait(aos2_frames_t *f);

/* Frame::Frame(imGray, imDepth, ...)  src/Frame.cc:116-170, the part after ExtractORB: N = mvKeys.size(),
 * UndistortKeyPoints for a rectified / distortion-free camera (mvKeysUn = mvKeys, :436-442), ComputeStereoFromRGBD
 * (:672-693), mvpMapPoints = NULL, mvbOutlier = false, AssignFeaturesToGrid (:259-274); scale tables copied from the
 * extractor (:94-100).  d_kps / d_desc / d_n ([batch][cap], [batch][cap][32], [batch]) are the device outputs of `e`'s
 * last batch, which may still be in flight: the call orders itself behind it on the device and keeps referring to
 * these buffers (the caller keeps them alive and unchanged while the batch is used).
 * d_depth: `batch` float images (imDepth after convertTo(CV_32F, mDepthMapFactor), Tracking.cc:226-227), depth_stride
 * floats per row, depth_image_stride floats per image; NULL = monocular (mvuRight = mvDepth = -1). */
int aos2_frames_build(aos2_frames_t *f, aos2_extractor_t *e, int batch, const aos2_keypoint_t *d_kps,
                      const uint8_t *d_desc, const int32_t *d_n, int cap, int w, int h, const float *d_depth,
                      int depth_stride, size_t depth_image_stride, float fx, float fy, float cx, float cy, float mbf);
/* Frame::Frame(imLeft, imRight, ...)  src/Frame.cc:57-113, the part after the two ExtractORB calls and ComputeStereoMatches:
 * the same members as aos2_frames_build with mvuRight / mvDepth taken from d_u_right / d_depth_kp ([batch][cap], the device
 * outputs of aos2_compute_stereo_matches_device[_async] on `e_left`, which may still be in flight: the call orders itself
 * behind `e_left` on the device). */
int aos2_frames_build_stereo(aos2_frames_t *f, aos2_extractor_t *e_left, int batch, const aos2_keypoint_t *d_kps,
                             const uint8_t *d_desc, const int32_t *d_n, int cap, int w, int h, const float *d_u_right,
                             const float *d_depth_kp, float fx, float fy, float cx, float cy, float mbf);
/* mDistCoef (k1 k2 p1 p2 k3) for the following aos2_frames_build calls: with k1 != 0 (the reference's test, src/Frame.cc:435,
 * :467) mvKeysUn = cv::undistortPoints(mvKeys, K, mDistCoef, noArray(), K) and the image bounds are the undistorted corners
 * (Frame::UndistortKeyPoints / ComputeImageBounds :433-493); the depth map is still read at mvKeys (:678-683).  Default: 0. */
int aos2_frames_set_distortion(aos2_frames_t *f, float k1, float k2, float p1, float p2, float k3);
/* void Frame::ComputeImageBounds(const cv::Mat &imLeft)  src/Frame.cc:463-493 (once per camera: four points, host):
 * bounds4 = mnMinX mnMaxX mnMinY mnMaxY for a w x h image; dist5 = k1 k2 p1 p2 k3 */
int aos2_frame_image_bounds(int w, int h, float fx, float fy, float cx, float cy, const float *dist5, float *bounds4);
/* Frame::SetPose for every frame: d_Tcw = [batch][16] float32 in device memory (mVelocity * mLastFrame.mTcw, Tracking.cc:975).
 * Independent of aos2_frames_build (which leaves mTcw alone): called before it, the copy runs beside the extraction. */
int aos2_frames_set_pose(aos2_frames_t *f, const float *d_Tcw);
/* mvpMapPoints (and optionally mvbOutlier) from the host: mp = [batch][cap] rows of `table` (-1 = NULL), outlier =
 * [batch][cap] or NULL (unchanged); for setting up a LastFrame batch and tests */
int aos2_frames_set_map_points(aos2_frames_t *f, const int32_t *mp, const uint8_t *outlier,
                               const aos2_map_points_dev_t *table);
/* members back to the host (waits for the batch): */
#define AOS2_FRAMES_MAP_POINTS 0 /* int32 [batch][cap]  mvpMapPoints as table rows */
#define AOS2_FRAMES_OUTLIER 1    /* uint8 [batch][cap]  mvbOutlier */
#define AOS2_FRAMES_TCW 2        /* float [batch][16]   mTcw */
#define AOS2_FRAMES_U_RIGHT 3    /* float [batch][cap]  mvuRight */
#define AOS2_FRAMES_DEPTH 4      /* float [batch][cap]  mvDepth */
#define AOS2_FRAMES_GRID_OFF 5   /* int32 [batch][64 * 48 + 1]  mGrid as CSR */
#define AOS2_FRAMES_GRID_IDX 6   /* int32 [batch][cap] */
#define AOS2_FRAMES_KEYS_UN_X 7  /* float [batch][cap]  mvKeysUn[i].pt.x */
#define AOS2_FRAMES_KEYS_UN_Y 8  /* float [batch][cap]  mvKeysUn[i].pt.y */
#define AOS2_FRAMES_KEYS_ANGLE 9 /* float [batch][cap]  mvKeysUn[i].angle (= mvKeys[i].angle)   (device pointer only) */
#define AOS2_FRAMES_KEYS_OCTAVE 10 /* int32 [batch][cap] mvKeysUn[i].octave                      (device pointer only) */
int aos2_frames_get(aos2_frames_t *f, int what, void *dst, size_t bytes);
/* the member array itself in device memory ([batch][cap] ...), valid while the batch lives; NULL before the first build.
 * For handing Frame members to the other device-resident entry points (aos2_matcher_search_by_bow_device) without a copy;
 * the caller orders its reads behind the batch's stream (aos2_frames_wait or an event on aos2_frames_stream). */
const void *aos2_frames_device_ptr(aos2_frames_t *f, int what);

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
 * src/ORBmatcher.cc:1328-1470 for the pairs (cur frame b, last frame b): reads last's mvpMapPoints / mvbOutlier /
 * mvKeys / mTcw and the table, writes cur's mvpMapPoints (the caller has just filled them with NULL,
 * Tracking.cc:977).  d_nmatches: [batch] return values in device memory, may be NULL.  The reference's retry with
 * 2 * th when fewer than 20 matches were found (Tracking.cc:984-988) is the caller's decision. */
int aos2_frames_search_by_projection_last(aos2_frames_t *cur, const aos2_frames_t *last,
                                          const aos2_map_points_dev_t *mps, float th, int mono,
                                          int check_orientation, int32_t *d_nmatches);
/* int Optimizer::PoseOptimization(Frame *pFrame)  src/Optimizer.cc:239-452 for every frame: edges from the features
 * that hold a map point (:260-350), result in mTcw and mvbOutlier; d_inliers [batch] (may be NULL) = return values */
/* (Two kernel forms: batches of more than 256 frames run 128 threads per frame with 9 edge slots each, smaller ones 256 threads
 * with 4; the edge order per thread and the order of the wave sums differ, so the SAME frame may get pose bits that differ in the
 * last places -- within 1e-5 of the oracle either way -- depending on the size of the batch it is in.  AOS2_PO_THREADS = 128 / 256
 * in the environment forces one form for every batch size.) */
int aos2_frames_pose_optimization(aos2_frames_t *f, const aos2_map_points_dev_t *mps, int32_t *d_inliers);
/* Tracking::TrackWithMotionModel :1008-1025 after PoseOptimization: a feature flagged as outlier loses its map point
 * (mvpMapPoints[i] = NULL, mvbOutlier[i] = false; the point counts as seen in this frame) */
int aos2_frames_discard_outliers(aos2_frames_t *f);
/* void Tracking::SearchLocalPoints()  src/Tracking.cc:1302-1352: d_local = [batch][n_local] rows of the table (-1 =
 * none) = mvpLocalMapPoints of every frame; points already in the frame are skipped (:1305-1328), the others go
 * through Frame::isInFrustum(pMP, 0.5) (src/Frame.cc:298-354) and ORBmatcher(nnratio).SearchByProjection(F,
 * vpMapPoints, th) (src/ORBmatcher.cc:45-129); writes mvpMapPoints.  At most 4096 features per frame. */
int aos2_frames_search_local_points(aos2_frames_t *f, const aos2_map_points_dev_t *mps, const int32_t *d_local,
                                    int n_local, float th, float nnratio, int32_t *d_nmatches);

/* Keyframe work on device-resident batches: a keyframe is a frame of a built batch (KeyFrame::KeyFrame copies exactly these
 * Frame members, src/KeyFrame.cc:37-62) with its mvpMapPoints set (aos2_frames_set_map_points or the tracking chain) and its
 * FeatureVector in HBM as aos2_vocabulary_transform_device wrote it.
 *
 * int ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)  src/ORBmatcher.cc:657-823 for n_pairs
 * (frame kf1[p] of `a`, frame kf2[p] of `b`) pairs -- LocalMapping::CreateNewMapPoints calls it once per neighbour keyframe
 * (src/LocalMapping.cc:272).  F12 (LocalMapping::ComputeF12) and the epipole (:663-670) are the caller's cv::Mat lines: host
 * arrays.  d_node_of1 = the node id of every feature of `a`'s frames at the FeatureVector level (aos2_vocabulary_transform_device's
 * d_node_of), d_fv_*1 / d_fv_*2 = the FeatureVectors of `a`'s / `b`'s frames (a feature whose word has weight 0 is in no
 * FeatureVector -- DBoW2 transform -- and the loop over KF1's FeatureVector never reaches it); all [batch][cap] device arrays
 * of the respective batch ([batch][cap + 1] for the offsets).
 * d_match12: DEVICE [n_pairs][cap of a] = vMatches12 (KF2 feature index or -1; vMatchedPairs = its non-negative entries in
 * ascending i), d_nmatches: DEVICE [n_pairs].  Returns when the results are complete. */
typedef struct {
    int32_t n_pairs;
    const int32_t *kf1, *kf2;      /* host [n_pairs] */
    const float *F12;              /* host [n_pairs][9], row-major */
    const float *epipole;          /* host [n_pairs][2]: ex, ey */
    const uint32_t *d_node_of1;
    const int32_t *d_fv_node1, *d_fv_off1, *d_fv_idx1, *d_n_fv1;
    const int32_t *d_fv_node2, *d_fv_off2, *d_fv_idx2, *d_n_fv2;
} aos2_frames_triang_t;
int aos2_frames_search_for_triangulation(aos2_frames_t *a, aos2_frames_t *b, const aos2_frames_triang_t *q,
                                         int only_stereo, int check_orientation, int32_t *d_match12,
                                         int32_t *d_nmatches);
/* The search part of int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th)  src/ORBmatcher.cc:825-975
 * for n_problems (target keyframe = frame target[p] of `kfs`, candidates = d_rows[p][0..n_pts): rows of the table, -1 where the
 * loop head rejects the point: NULL, isBad() or IsInKeyFrame(pKF) -- map state the caller evaluates) -- LocalMapping::
 * SearchInNeighbors calls it once per neighbour (src/LocalMapping.cc:493).  d_best_idx / d_best_dist: DEVICE [n_problems][n_pts]
 * as aos2_matcher_fuse returns them; the Replace / AddObservation bookkeeping of :948-969 consumes them on the host. */
int aos2_frames_fuse(aos2_frames_t *kfs, const aos2_map_points_dev_t *mps, int n_problems, int n_pts,
                     const int32_t *target, const int32_t *d_rows, float th, int32_t *d_best_idx,
                     int32_t *d_best_dist);
/* on != 0: aos2_frames_search_for_triangulation (handle = its `a`) and aos2_frames_fuse (handle = `kfs`) return after ENQUEUEING on the
 * handle's stream -- LocalMapping's three calls for a keyframe then cost one wait instead of three round trips through a device that
 * is busy with the tracking kernels; aos2_frames_wait(handle) (or ordering another stream behind aos2_frames_stream(handle))
 * completes them.  The host arrays of a call (kf1, kf2, F12, epipole, target) are copied before it returns; the OTHER handle of a call
 * (`b` of SearchForTriangulation, the keyframes a Fuse call reads) and every device input (the FeatureVector CSRs, d_node_of1, the map
 * point table) must stay untouched -- no rebuild, no set_pose, no reuse of the buffers -- until aos2_frames_wait(handle) of the handle
 * the call was made on has returned: its kernels read them from that handle's stream.  Default off: the calls return with their results
 * complete. */
int aos2_frames_set_async_keyframe_calls(aos2_frames_t *f, int on);

/* ------------------------------------------------------------------------------------------
 * Replay of a fixed call sequence (csrc/replay.hip).  No single reference function: the per-frame body of Tracking::Track on its
 * usual path -- Frame::Frame (src/Frame.cc:116-172), TrackWithMotionModel (src/Tracking.cc:860-1039: SearchByProjection,
 * PoseOptimization, the outlier discard), TrackLocalMap (:1041-1100: SearchLocalPoints, PoseOptimization) -- is the same ~20
 * dependent launches for every frame.  For one sequence (batch 1) the launches' own latency is a sixth of the frame time; a
 * recorded sequence is enqueued with one call and its kernels run back to back.
 *
 *   run the sequence once (the handles allocate on their first call with a shape), then
 *   aos2_capture_begin(aos2_frames_stream(cur));
 *   aos2_extractor_wait_for_stream(e, aos2_frames_stream(cur));      -- the extractor's stream joins the recording
 *   ... aos2_frames_set_pose, aos2_extractor_extract_batch_device_async, aos2_frames_build (orders the batch behind the
 *       extraction: the extractor's stream leaves the recording), aos2_frames_search_by_projection_last,
 *       aos2_frames_pose_optimization, ... -- NOTHING runs, the launches are recorded with their arguments
 *   aos2_capture_end(aos2_frames_stream(cur), &g);
 *   per frame: write the image / the pose guess / the local-map rows into the SAME device buffers, aos2_graph_launch(g, stream),
 *   aos2_frames_wait(cur).
 *
 * What is fixed at recording time: every argument passed by value (sizes, thresholds, device addresses).  A count that changes from
 * frame to frame goes through device memory the kernels already read (the frames' keypoint counts d_n, the MapPoint table) or is
 * padded to a capacity: d_local rows of -1 are "no point" (aos2_frames_search_local_points), so n_local can be the capacity of the
 * local map.  Calls that wait on the host (aos2_*_wait, the synchronous forms) cannot be recorded.
 * ------------------------------------------------------------------------------------------ */
typedef struct aos2_graph aos2_graph_t;
/* starts recording on `hip_stream` (a handle's stream, not NULL): the calls that follow enqueue nothing */
int aos2_capture_begin(void *hip_stream);
/* ends the recording (every other stream that joined must have been waited for by `hip_stream`) and builds the replayable graph */
int aos2_capture_end(void *hip_stream, aos2_graph_t **out);
/* enqueues the recorded sequence on `hip_stream` (normally the stream it was recorded on); returns at once */
int aos2_graph_launch(aos2_graph_t *g, void *hip_stream);
int aos2_graph_nodes(const aos2_graph_t *g);   /* kernels + copies of the recording */
void aos2_graph_destroy(aos2_graph_t *g);

/* ------------------------------------------------------------------------------------------
 * Dataset helper (host code; the reference reads its datasets with cv::imread, Examples/RGB-D/rgbd_tum.cc:77-78): the PNG
 * scanline filters undone in place -- rows = h rows of 1 filter byte + stride data bytes as inflated, bpp = bytes per pixel.
 * ------------------------------------------------------------------------------------------ */
int aos2_png_unfilter(uint8_t *rows, int h, int stride, int bpp);

/* ------------------------------------------------------------------------------------------
 * Test taps (no reference equivalent): shared primitives run in isolation.
 * ------------------------------------------------------------------------------------------ */
/* DistributeOctTree (src/ORBextractor.cc:539-763) on the HOST with the routine the device kernel
 * also runs (csrc/octree.h).  Returns the number of kept candidates (indices in out_idx) or <0. */
int aos2_debug_octree_host(const int16_t *xs, const int16_t *ys, const uint8_t *score, int n,
                           int minX, int maxX, int minY, int maxY, int N, int32_t *out_idx, int cap);
/* rBRIEF steering sin/cos (csrc/sincos_exact.h) evaluated on the host / on the device */
void aos2_debug_sincos_host(float angle_rad, float *s, float *c);
int aos2_debug_sincos_device(const float *angles, int n, float *s, float *c, int device);
/* building blocks of the pose solver (csrc/pose_opt.hip) on the device, n independent cases:
 * T_out[i] = exp(upd[i]) * T[i]  (upd: 6 doubles omega | upsilon, T: 7 doubles qx qy qz qw tx ty tz), and
 * x[i] = (H[i] + lambda[i] I)^-1 b[i] with Hb[i] = 21 doubles (upper triangle of H, row by row) + 6 doubles b;
 * ok[i] = 0 where a pivot was not positive (x[i] is left as passed in) */
int aos2_debug_pose_blocks_device(const double *upd, const double *T, double *T_out, const double *Hb, const double *lambda,
                                  double *x, uint8_t *ok, int n, int device);

#ifdef __cplusplus
}
#endif
#endif /* AOS2_H */
