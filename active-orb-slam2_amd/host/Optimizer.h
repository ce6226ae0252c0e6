// ORB_SLAM2::Optimizer over POD problems (the SoA level under host/ref/Optimizer.h, which keeps the reference's
// signatures and does the gathering).
// The real Optimizer.cc keeps `void static LocalBundleAdjustment(KeyFrame*, bool*, Map*)`: its
// window gathering (:457-505) fills aos2_lba_problem_t, this call replaces :507-744, and its
// write-back (:746-778) consumes aos2_lba_result_t.
#pragma once
#include <stdexcept>
#include <string>

#include "aos2_types.h"
#include "ref/aos2_handles.h"   // one persistent optimiser handle per calling thread

namespace ORB_SLAM2 {

class Optimizer {
public:
    // returns false when *pbStopFlag was already set (the reference returns before optimising)
    bool static LocalBundleAdjustment(const aos2_lba_problem_t &problem, aos2_lba_result_t &result, int device = 0)
    {
        aos2::default_device() = device;
        const int st = aos2_lba_solve(aos2::optimizer_handle(), &problem, &result);
        if (st == AOS2_ERR_STOPPED) return false;
        if (st != AOS2_OK) throw std::runtime_error(std::string("LocalBundleAdjustment: ") + aos2_last_error());
        return true;
    }

    // int Optimizer::PoseOptimization(Frame *pFrame) (include/Optimizer.h:47, src/Optimizer.cc:239-452):
    // returns nInitialCorrespondences - nBad; result.Tcw / result.outlier are what the reference writes
    // into pFrame->mTcw / pFrame->mvbOutlier for the features that have a map point
    int static PoseOptimization(const aos2_pose_problem_t &frame, aos2_pose_result_t &result, int device = 0)
    {
        aos2::default_device() = device;
        const int st = aos2_pose_optimization(aos2::optimizer_handle(), &frame, &result, 1);
        if (st != AOS2_OK) throw std::runtime_error(std::string("PoseOptimization: ") + aos2_last_error());
        return result.n_inliers;
    }
};

}  // namespace ORB_SLAM2
