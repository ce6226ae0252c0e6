// void MapPoint::ComputeDistinctiveDescriptors()  at its reference signature (include/MapPoint.h:75, src/MapPoint.cc:275-340;
// called from LocalMapping::ProcessNewKeyFrame / CreateNewMapPoints / SearchInNeighbors, src/LocalMapping.cc:156, 428, 531,
// and Tracking.cc after map point creation): the body a maintainer puts into src/MapPoint.cc.  The all-pairs Hamming
// distances and the least-median pick run on the GPU; one call handles one point here, the C ABI takes a batch
// (aos2_compute_distinctive_descriptors: LocalMapping's loops over many points can hand all of them over at once).
// Include AFTER MapPoint.h / KeyFrame.h (or tests/cpp/refstub/slam_stub.h).
#pragma once
#include <map>
#include <mutex>
#include <vector>

#include "aos2_handles.h"

namespace ORB_SLAM2 {

inline void MapPoint::ComputeDistinctiveDescriptors()
{
    std::map<KeyFrame *, size_t> observations;
    {
        std::unique_lock<std::mutex> lock1(mMutexFeatures);
        if (mbBad) return;
        observations = mObservations;
    }
    if (observations.empty()) return;
    // descriptors of the non-bad observing keyframes, in the map's order (:297-303)
    std::vector<uint8_t> desc;
    std::vector<cv::Mat> rows;
    desc.reserve(observations.size() * 32);
    for (auto &mit : observations) {
        KeyFrame *pKF = mit.first;
        if (pKF->isBad()) continue;
        const cv::Mat d = pKF->mDescriptors.row((int)mit.second);
        desc.insert(desc.end(), d.ptr<uint8_t>(), d.ptr<uint8_t>() + 32);
        rows.push_back(d);
    }
    if (rows.empty()) return;
    const int32_t off[2] = {0, (int32_t)rows.size()};
    int32_t best = -1;
    aos2::check(aos2_compute_distinctive_descriptors(aos2::matcher_handle(0.6f, true), 1, off, desc.data(), &best),
                "ComputeDistinctiveDescriptors");
    {
        std::unique_lock<std::mutex> lock(mMutexFeatures);
        mDescriptor = rows[best].clone();   // :336-339
    }
}

}  // namespace ORB_SLAM2
