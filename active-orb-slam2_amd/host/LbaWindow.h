// aos2::LbaWindow -- the repository's own builder of one LocalBundleAdjustment problem (include/aos2.h aos2_lba_problem_t) from the
// pointer graph.  What has to come out is fixed by the reference (src/Optimizer.cc:457-654): WHICH keyframes are optimised, WHICH map
// points, WHICH keyframes enter with a constant pose, one edge per (non-bad keyframe, point) observation.  How it is put together here:
//
//   * membership is a hash index (object address -> row), not a stamp on the objects: KeyFrame::mnBALocalForKF / mnBAFixedForKF and
//     MapPoint::mnBALocalForKF (read by nothing outside Optimizer.cc) are neither read nor written, and no std::list is built;
//   * ONE walk emits everything: optimised keyframes -> their map points in first-seen order -> per point its observations by ascending
//     KeyFrame::mnId (DESIGN.md convention 2); an observer that is not an optimised keyframe becomes a constant camera the moment its
//     first edge is emitted -- the reference's separate pass over every point's observations (:490-505) does not exist here;
//   * the rows ARE the C ABI's arrays (float poses / points, int32 edge ends): nothing is converted again before the call, and the
//     solver's outputs are scattered back through the row -> object tables kept beside them.
//
// g2o orders the unknowns by vertex id and the edges by insertion, so the order in which constant cameras get their
// rows changes nothing; the edge order is the one convention 2 fixes.
// Works on any KeyFrame / MapPoint types with the reference's members (the real ones, or tests/cpp/refstub/slam_stub.h).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/aos2.h"

namespace aos2 {

template <class KF, class MP>
class LbaWindow {
public:
    // rows of the problem, in emission order
    std::vector<KF *> keyframes;            // optimised ones first (n_optimised of them), then the constant cameras
    std::vector<MP *> points;
    std::vector<std::pair<int32_t, int32_t>> edge_rows;   // (keyframe row, point row) of every edge
    size_t n_optimised = 0;

    // an optimised keyframe (constant only if it is the map's first one: Optimizer.cc:529); ignored if known or bad
    void optimise(KF *kf)
    {
        if (!kf || kf->isBad() || kf_row_.count(kf)) return;
        if (n_optimised != keyframes.size()) return;   // (all optimised keyframes come before the first point)
        add_keyframe(kf, kf->mnId == 0);
        n_optimised = keyframes.size();
    }

    // every non-bad map point the optimised keyframes hold, once, in the order their feature lists name them (:471-488)
    void collect_points()
    {
        for (size_t k = 0; k < n_optimised; ++k)
            for (MP *mp : keyframes[k]->GetMapPointMatches()) {
                if (!mp || mp->isBad() || !pt_row_.emplace(mp, (int32_t)points.size()).second) continue;
                points.push_back(mp);
                const cv::Mat X = mp->GetWorldPos();
                for (int c = 0; c < 3; ++c) point_xyz_.push_back(X.template at<float>(c));
                point_id_.push_back((int64_t)mp->mnId);
            }
    }

    // the observations of every collected point -> edges; observers outside the optimised set enter as constant cameras (:490-505, 573-654)
    void emit_edges()
    {
        std::vector<std::pair<KF *, size_t>> obs;
        for (size_t j = 0; j < points.size(); ++j) {
            const auto observations = points[j]->GetObservations();
            obs.assign(observations.begin(), observations.end());
            std::sort(obs.begin(), obs.end(), [](const std::pair<KF *, size_t> &a, const std::pair<KF *, size_t> &b) { return a.first->mnId < b.first->mnId; });
            for (const auto &o : obs) {
                KF *kf = o.first;
                if (kf->isBad()) continue;   // (no vertex for a bad keyframe: g2o drops such an edge)
                auto at = kf_row_.find(kf);
                const int32_t row = at != kf_row_.end() ? at->second : add_keyframe(kf, true);
                const cv::KeyPoint &kp = kf->mvKeysUn[o.second];
                const float u_right = kf->mvuRight[o.second];
                edge_pose_.push_back(row);
                edge_point_.push_back((int32_t)j);
                edge_obs_.insert(edge_obs_.end(), {kp.pt.x, kp.pt.y, u_right});
                edge_stereo_.push_back(u_right < 0 ? 0 : 1);                       // :595
                edge_inv_sigma2_.push_back(kf->mvInvLevelSigma2[kp.octave]);       // :606, :632
                edge_rows.emplace_back(row, (int32_t)j);
            }
        }
    }

    bool empty() const { return keyframes.empty() || points.empty() || edge_rows.empty(); }

    // the C-ABI view of the rows (valid while this object lives and is not modified); camera = the centre keyframe's (:613-616, 642-646)
    aos2_lba_problem_t problem(const KF *camera, const volatile uint8_t *stop_flag) const
    {
        aos2_lba_problem_t P;
        memset(&P, 0, sizeof(P));
        P.n_poses = (int32_t)keyframes.size(); P.n_points = (int32_t)points.size(); P.n_edges = (int32_t)edge_rows.size();
        P.pose_Tcw = pose_Tcw_.data(); P.pose_fixed = pose_fixed_.data(); P.pose_id = pose_id_.data();
        P.point_xyz = point_xyz_.data(); P.point_id = point_id_.data();
        P.edge_pose = edge_pose_.data(); P.edge_point = edge_point_.data(); P.edge_obs = edge_obs_.data();
        P.edge_stereo = edge_stereo_.data(); P.edge_inv_sigma2 = edge_inv_sigma2_.data();
        P.fx = camera->fx; P.fy = camera->fy; P.cx = camera->cx; P.cy = camera->cy; P.bf = camera->mbf;
        P.stop_flag = stop_flag;
        P.iters_first = 5; P.iters_second = 10;   // :661, :708
        return P;
    }
    const std::vector<int64_t> &pose_ids() const { return pose_id_; }
    const std::vector<int64_t> &point_ids() const { return point_id_; }

private:
    int32_t add_keyframe(KF *kf, bool constant)
    {
        const int32_t row = (int32_t)keyframes.size();
        kf_row_.emplace(kf, row);
        keyframes.push_back(kf);
        const cv::Mat T = kf->GetPose();
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) pose_Tcw_.push_back(T.template at<float>(r, c));
        pose_fixed_.push_back(constant ? 1 : 0);
        pose_id_.push_back((int64_t)kf->mnId);
        return row;
    }
    std::unordered_map<const KF *, int32_t> kf_row_;
    std::unordered_map<const MP *, int32_t> pt_row_;
    std::vector<float> pose_Tcw_, point_xyz_, edge_obs_, edge_inv_sigma2_;
    std::vector<uint8_t> pose_fixed_, edge_stereo_;
    std::vector<int64_t> pose_id_, point_id_;
    std::vector<int32_t> edge_pose_, edge_point_;
};

}  // namespace aos2
