// TEST INFRASTRUCTURE: stand-ins for ORB_SLAM2::{MapPoint, KeyFrame, Frame, Map} carrying only the members the hot-path
// methods read and write, with the reference's member and accessor names (include/MapPoint.h,
// KeyFrame.h, Frame.h, Map.h) so that the classes in active-orb-slam2_amd/host/*.h compile against them exactly as they
// would against the real headers.  Not a port of the data model: no covisibility graph maintenance, no spanning tree,
// no culling, no mutex discipline beyond the mutexes the hot-path bodies take; the map-editing calls (Replace,
// AddObservation, AddMapPoint, Erase*) only record that they were made.
#pragma once
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "opencv_stub.h"

namespace DBoW2 {
typedef unsigned int WordId;   // Thirdparty/DBoW2/DBoW2/BowVector.h:20
typedef double WordValue;      // :23
typedef unsigned int NodeId;   // :26
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};   // Thirdparty/DBoW2/DBoW2/FeatureVector.h:21-22
}  // namespace DBoW2

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

namespace ORB_SLAM2 {

class KeyFrame;
class Frame;
class ORBextractor;
class ORBVocabulary;

// bookkeeping of the tests: the map-editing calls in the order they were made, as (kind, id, argument) triples --
// 'A' = MapPoint::AddObservation(point mnId, feature index), 'R' = MapPoint::Replace(victim mnId, replacing point's mnId)
inline std::vector<long> &event_log()
{
    static std::vector<long> log;
    return log;
}

class MapPoint {
public:
    long unsigned int mnId = 0;
    // variables used by the tracking (include/MapPoint.h:88-96)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;
    long unsigned int mnBALocalForKF = (long unsigned int)-1;

    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    void SetWorldPos(const cv::Mat &Pos) { mWorldPos = Pos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    std::map<KeyFrame *, size_t> GetObservations() { return mObservations; }
    int Observations() { return nObs; }
    bool isBad() { return mbBad; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }   // src/MapPoint.cc:413-423
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    void stub_set_distances(float min_dist, float max_dist) { mfMinDistance = min_dist; mfMaxDistance = max_dist; }   // (UpdateNormalAndDepth sets them there)
    bool IsInKeyFrame(KeyFrame *pKF) { return mObservations.count(pKF) != 0; }
    int GetIndexInKeyFrame(KeyFrame *pKF)
    {
        auto it = mObservations.find(pKF);
        return it == mObservations.end() ? -1 : (int)it->second;
    }
    void AddObservation(KeyFrame *pKF, size_t idx)
    {
        if (mObservations.count(pKF)) return;
        mObservations[pKF] = idx;
        ++nObs;
        event_log().insert(event_log().end(), {(long)'A', (long)mnId, (long)idx});
    }
    void EraseObservation(KeyFrame *pKF)
    {
        mObservations.erase(pKF);
        ++nErased;
    }
    void Replace(MapPoint *pMP);   // (below: needs KeyFrame)
    void UpdateNormalAndDepth() { ++nNormalUpdates; }
    void ComputeDistinctiveDescriptors();   // body: active-orb-slam2_amd/host/MapPointMembers.h

    cv::Mat mWorldPos, mNormalVector, mDescriptor;
    std::map<KeyFrame *, size_t> mObservations;
    std::mutex mMutexFeatures;
    int nObs = 0;
    bool mbBad = false;
    int nErased = 0, nNormalUpdates = 0;   // bookkeeping of the tests

protected:   // like include/MapPoint.h:123, 149-154: the shim reads them without an accessor (aos2::MapPointDistances)
    float mfMinDistance = 0, mfMaxDistance = 0;
    std::mutex mMutexPos;
};

class KeyFrame {
public:
    long unsigned int mnId = 0;
    long unsigned int mnBALocalForKF = (long unsigned int)-1, mnBAFixedForKF = (long unsigned int)-1;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0;
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    int mnScaleLevels = 8;
    float mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;   // (const int in the reference, include/KeyFrame.h:198-201)
    float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;

    cv::Mat GetPose() { return Tcw.clone(); }
    void SetPose(const cv::Mat &T) { Tcw = T.clone(); }
    cv::Mat GetRotation() { return Tcw.rowRange(0, 3).colRange(0, 3).clone(); }
    cv::Mat GetTranslation() { return Tcw.rowRange(0, 3).col(3).clone(); }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    std::set<MapPoint *> GetMapPoints()
    {
        std::set<MapPoint *> s;
        for (MapPoint *p : mvpMapPoints)
            if (p && !p->isBad()) s.insert(p);
        return s;
    }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
    bool isBad() { return mbBad; }
    void EraseMapPointMatch(MapPoint *pMP)
    {
        for (auto &p : mvpMapPoints)
            if (p == pMP) p = nullptr;
        ++nErased;
    }

    cv::Mat Tcw, Ow;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<KeyFrame *> mvpOrderedConnectedKeyFrames;
    bool mbBad = false;
    int nErased = 0;

protected:
    // protected like the reference's (include/KeyFrame.h:206,223): the shim must not read it
    std::vector<std::vector<std::vector<size_t>>> mGrid;   // [64][48]
};

inline void MapPoint::Replace(MapPoint *pMP)
{
    // the part of src/MapPoint.cc:184-229 the Fuse loop can observe afterwards: this point turns bad, its observations move
    if (pMP->mnId == mnId) return;
    event_log().insert(event_log().end(), {(long)'R', (long)mnId, (long)pMP->mnId});
    for (auto &ob : mObservations) {
        KeyFrame *pKF = ob.first;
        if (!pMP->IsInKeyFrame(pKF)) {
            pKF->mvpMapPoints[ob.second] = pMP;
            pMP->AddObservation(pKF, ob.second);
        } else
            pKF->mvpMapPoints[ob.second] = nullptr;
    }
    mObservations.clear();
    nObs = 0;
    mbBad = true;
}

class Frame {
public:
    int N = 0;
    static float fx, fy, cx, cy;
    float mb = 0, mbf = 0;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    ORBVocabulary *mpORBvocabulary = nullptr;
    ORBextractor *mpORBextractorLeft = nullptr, *mpORBextractorRight = nullptr;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    cv::Mat mTcw;
    int mnScaleLevels = 8;
    float mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvInvLevelSigma2;
    int nBadPoseOpt = 0;   // (Active-ORB-SLAM2 keeps the count of the last PoseOptimization)
    void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); }
    void ComputeBoW();             // bodies: active-orb-slam2_amd/host/FrameMembers.h
    void ComputeStereoMatches();
};

class Map {
public:
    std::mutex mMutexMapUpdate;
};

}  // namespace ORB_SLAM2
