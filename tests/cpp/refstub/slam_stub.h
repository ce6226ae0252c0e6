// TEST INFRASTRUCTURE: stand-ins for ORB_SLAM2::{MapPoint, KeyFrame, Frame, Map} carrying only the members the hot-path
// methods read and write, with the reference's member and accessor names (include/MapPoint.h,
// KeyFrame.h, Frame.h, Map.h) so that the shims in active-orb-slam2_amd/host/ref/*.h compile against them exactly as
// they would against the real headers.  Not a port of the data model: no covisibility graph maintenance, no spanning
// tree, no culling, no mutex discipline beyond the one mutex LocalBundleAdjustment takes.
#pragma once
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "opencv_stub.h"

namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};   // Thirdparty/DBoW2/DBoW2/FeatureVector.h:21-22
}  // namespace DBoW2

#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64

namespace ORB_SLAM2 {

class KeyFrame;
class Frame;

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
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    std::map<KeyFrame *, size_t> GetObservations() { return mObservations; }
    int Observations() { return nObs; }
    bool isBad() { return mbBad; }
    void EraseObservation(KeyFrame *pKF)
    {
        mObservations.erase(pKF);
        ++nErased;
    }
    void UpdateNormalAndDepth() { ++nNormalUpdates; }

    cv::Mat mWorldPos, mDescriptor;
    std::map<KeyFrame *, size_t> mObservations;
    int nObs = 0;
    bool mbBad = false;
    int nErased = 0, nNormalUpdates = 0;   // bookkeeping of the test
};

class KeyFrame {
public:
    long unsigned int mnId = 0;
    long unsigned int mnBALocalForKF = (long unsigned int)-1, mnBAFixedForKF = (long unsigned int)-1;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    std::vector<float> mvInvLevelSigma2;

    cv::Mat GetPose() { return Tcw.clone(); }
    void SetPose(const cv::Mat &T) { Tcw = T.clone(); }
    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    std::vector<KeyFrame *> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
    bool isBad() { return mbBad; }
    void EraseMapPointMatch(MapPoint *pMP)
    {
        for (auto &p : mvpMapPoints)
            if (p == pMP) p = nullptr;
        ++nErased;
    }

    cv::Mat Tcw;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<KeyFrame *> mvpOrderedConnectedKeyFrames;
    bool mbBad = false;
    int nErased = 0;
};

class Frame {
public:
    int N = 0;
    static float fx, fy, cx, cy;
    float mb = 0, mbf = 0;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    static float mfGridElementWidthInv, mfGridElementHeightInv;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    cv::Mat mTcw;
    int mnScaleLevels = 8;
    std::vector<float> mvScaleFactors, mvInvLevelSigma2;
    int nBadPoseOpt = 0;   // (Active-ORB-SLAM2 keeps the count of the last PoseOptimization)
    void SetPose(cv::Mat Tcw) { mTcw = Tcw.clone(); }
};

class Map {
public:
    std::mutex mMutexMapUpdate;
};

}  // namespace ORB_SLAM2
