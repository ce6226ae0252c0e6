// Optional (needs Eigen3 + a checkout of the reference; NO OpenCV): the numerical part of Optimizer::LocalBundleAdjustment
// (src/Optimizer.cc:507-744) driven through the reference's OWN vendored g2o -- BlockSolver_6_3 + LinearSolverEigen +
// OptimizationAlgorithmLevenberg + EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ, compiled untouched from
// Thirdparty/g2o by tools/refcheck/CMakeLists.txt -- on the problems tools/refcheck/dump_cases.py wrote, compared with the
// ORACLE's result for the same problem (stored in the same bundle).  This harness only builds the graph the way
// Optimizer.cc does (vertex ids, edge order, information, Huber deltas, the two optimize() calls with the outlier pass between
// them) and reads the estimates back through the float32 conversion of Converter::toCvMat; every number comes from g2o.
//
//   refcheck_lba case0.bundle [case1.bundle ...]      exit code 0 = every case within 1e-5, identical outlier sets
#include <cmath>
#include <cstdio>
#include <vector>

#include <Eigen/Core>
#include <Eigen/StdVector>

#include "Thirdparty/g2o/g2o/core/block_solver.h"
#include "Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.h"
#include "Thirdparty/g2o/g2o/core/robust_kernel_impl.h"
#include "Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h"
#include "Thirdparty/g2o/g2o/types/types_six_dof_expmap.h"

#include "../../tests/cpp/bundle_io.h"

// Converter::toSE3Quat (src/Converter.cc:37-47) without cv::Mat: float32 row-major 4x4 -> double R, t -> SE3Quat(R, t)
static g2o::SE3Quat to_se3quat(const float *T)
{
    Eigen::Matrix<double, 3, 3> R;
    R << T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10];
    Eigen::Matrix<double, 3, 1> t(T[3], T[7], T[11]);
    return g2o::SE3Quat(R, t);
}

static int run_case(const char *path)
{
    const Bundle B = Bundle::load(path);
    const int n_poses = (int)B["pose_Tcw"].dims[0], n_points = (int)B["point_xyz"].dims[0], n_edges = (int)B["edge_pose"].count();
    const float *Tcw = B["pose_Tcw"].as<float>(), *xyz = B["point_xyz"].as<float>(), *obs = B["edge_obs"].as<float>();
    const float *w = B["edge_inv_sigma2"].as<float>(), *cam = B["cam"].as<float>();   // fx fy cx cy bf
    const uint8_t *fixed = B["pose_fixed"].as<uint8_t>(), *stereo = B["edge_stereo"].as<uint8_t>();
    const int64_t *pose_id = B["pose_id"].as<int64_t>(), *point_id = B["point_id"].as<int64_t>();
    const int32_t *e_pose = B["edge_pose"].as<int32_t>(), *e_point = B["edge_point"].as<int32_t>();

    g2o::SparseOptimizer optimizer;
    g2o::BlockSolver_6_3::LinearSolverType *linearSolver = new g2o::LinearSolverEigen<g2o::BlockSolver_6_3::PoseMatrixType>();
    g2o::BlockSolver_6_3 *solver_ptr = new g2o::BlockSolver_6_3(linearSolver);
    optimizer.setAlgorithm(new g2o::OptimizationAlgorithmLevenberg(solver_ptr));

    unsigned long maxKFid = 0;
    for (int i = 0; i < n_poses; ++i) {   // :523-548: local keyframes (fixed iff mnId == 0), then the fixed ones
        g2o::VertexSE3Expmap *v = new g2o::VertexSE3Expmap();
        v->setEstimate(to_se3quat(Tcw + 16 * i));
        v->setId((int)pose_id[i]);
        v->setFixed(fixed[i] != 0 || pose_id[i] == 0);
        optimizer.addVertex(v);
        if ((unsigned long)pose_id[i] > maxKFid) maxKFid = (unsigned long)pose_id[i];
    }
    const float thHuberMono = std::sqrt(5.991), thHuberStereo = std::sqrt(7.815);   // (float like :570-571)
    std::vector<g2o::VertexSBAPointXYZ *> vpts(n_points);
    for (int j = 0; j < n_points; ++j) {   // :573-580
        g2o::VertexSBAPointXYZ *v = new g2o::VertexSBAPointXYZ();
        v->setEstimate(Eigen::Matrix<double, 3, 1>(xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]));
        v->setId((int)(point_id[j] + maxKFid + 1));
        v->setMarginalized(true);
        optimizer.addVertex(v);
        vpts[j] = v;
    }
    std::vector<g2o::EdgeSE3ProjectXYZ *> mono(n_edges, nullptr);
    std::vector<g2o::EdgeStereoSE3ProjectXYZ *> ster(n_edges, nullptr);
    for (int k = 0; k < n_edges; ++k) {   // :582-651, in the order of the edge arrays
        g2o::OptimizableGraph::Vertex *vp = dynamic_cast<g2o::OptimizableGraph::Vertex *>(optimizer.vertex((int)(point_id[e_point[k]] + maxKFid + 1)));
        g2o::OptimizableGraph::Vertex *vk = dynamic_cast<g2o::OptimizableGraph::Vertex *>(optimizer.vertex((int)pose_id[e_pose[k]]));
        const float invSigma2 = w[k];
        if (!stereo[k]) {
            Eigen::Matrix<double, 2, 1> o;
            o << obs[3 * k], obs[3 * k + 1];
            g2o::EdgeSE3ProjectXYZ *e = new g2o::EdgeSE3ProjectXYZ();
            e->setVertex(0, vp);
            e->setVertex(1, vk);
            e->setMeasurement(o);
            e->setInformation(Eigen::Matrix2d::Identity() * invSigma2);
            g2o::RobustKernelHuber *rk = new g2o::RobustKernelHuber;
            e->setRobustKernel(rk);
            rk->setDelta(thHuberMono);
            e->fx = cam[0]; e->fy = cam[1]; e->cx = cam[2]; e->cy = cam[3];
            optimizer.addEdge(e);
            mono[k] = e;
        } else {
            Eigen::Matrix<double, 3, 1> o;
            o << obs[3 * k], obs[3 * k + 1], obs[3 * k + 2];
            g2o::EdgeStereoSE3ProjectXYZ *e = new g2o::EdgeStereoSE3ProjectXYZ();
            e->setVertex(0, vp);
            e->setVertex(1, vk);
            e->setMeasurement(o);
            e->setInformation(Eigen::Matrix3d::Identity() * invSigma2);
            g2o::RobustKernelHuber *rk = new g2o::RobustKernelHuber;
            e->setRobustKernel(rk);
            rk->setDelta(thHuberStereo);
            e->fx = cam[0]; e->fy = cam[1]; e->cx = cam[2]; e->cy = cam[3]; e->bf = cam[4];
            optimizer.addEdge(e);
            ster[k] = e;
        }
    }
    optimizer.initializeOptimization();
    optimizer.optimize(5);   // :661
    for (int k = 0; k < n_edges; ++k) {   // :672-703 (no stop flag: bDoMore)
        if (mono[k]) {
            if (mono[k]->chi2() > 5.991 || !mono[k]->isDepthPositive()) mono[k]->setLevel(1);
            mono[k]->setRobustKernel(0);
        } else {
            if (ster[k]->chi2() > 7.815 || !ster[k]->isDepthPositive()) ster[k]->setLevel(1);
            ster[k]->setRobustKernel(0);
        }
    }
    optimizer.initializeOptimization(0);
    optimizer.optimize(10);   // :708
    std::vector<uint8_t> outlier(n_edges, 0);
    for (int k = 0; k < n_edges; ++k)   // :712-744
        outlier[k] = mono[k] ? (mono[k]->chi2() > 5.991 || !mono[k]->isDepthPositive()) : (ster[k]->chi2() > 7.815 || !ster[k]->isDepthPositive());

    // ---- against the oracle's result (float32 write-back, Converter::toCvMat, :763-778)
    const float *oT = B["out_pose_Tcw"].as<float>(), *oX = B["out_point_xyz"].as<float>();
    const uint8_t *oO = B["out_outlier"].as<uint8_t>();
    double worstT = 0, worstX = 0;
    int dout = 0;
    for (int i = 0; i < n_poses; ++i) {
        g2o::VertexSE3Expmap *v = static_cast<g2o::VertexSE3Expmap *>(optimizer.vertex((int)pose_id[i]));
        const Eigen::Matrix<double, 4, 4> M = v->estimate().to_homogeneous_matrix();
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) worstT = std::max(worstT, std::fabs((double)(float)M(r, c) - (double)oT[16 * i + 4 * r + c]));
    }
    for (int j = 0; j < n_points; ++j) {
        const Eigen::Matrix<double, 3, 1> X = vpts[j]->estimate();
        for (int d = 0; d < 3; ++d) worstX = std::max(worstX, std::fabs((double)(float)X(d) - (double)oX[3 * j + d]));
    }
    for (int k = 0; k < n_edges; ++k) dout += outlier[k] != oO[k];
    // the bar: 1e-5, or 4 x the spread of the oracle against its own re-associated runs on THIS window where that is larger (a window that
    // starts far from the optimum; dump_cases.py stores it as "resolution": oracle/parity.py lba_resolution) -- the rule of the tests
    double tolT = 1e-5, tolX = 1e-5;
    if (B.has("resolution")) {
        const double *res = B["resolution"].as<double>();
        tolT = std::max(tolT, 4.0 * res[0]);
        tolX = std::max(tolX, 4.0 * res[1]);
    }
    const bool ok = worstT <= tolT && worstX <= tolX && dout == 0;
    printf("%s: %d keyframes, %d points, %d edges: poses off by %.3g (bar %.3g), points by %.3g (bar %.3g), %d outlier flags differ -> %s\n", path,
           n_poses, n_points, n_edges, worstT, tolT, worstX, tolX, dout, ok ? "ok" : "DIFFERENT");
    return ok ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: refcheck_lba case.bundle ...   (python tools/refcheck/dump_cases.py DIR writes them)\n");
        return 2;
    }
    int bad = 0;
    for (int i = 1; i < argc; ++i) bad += run_case(argv[i]);
    return bad ? 1 : 0;
}
