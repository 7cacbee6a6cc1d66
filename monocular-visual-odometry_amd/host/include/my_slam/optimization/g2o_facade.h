// my_slam/optimization/g2o_facade.h -- the g2o-shaped graph surface the reference programs against in
// src/optimization/g2o_ba.cpp:193-289: BlockSolver<BlockSolverTraits<6,3>>, LinearSolverDense / LinearSolverCSparse,
// OptimizationAlgorithmLevenberg, SparseOptimizer::{setAlgorithm, addVertex, addParameter, addEdge, vertex,
// setVerbose, initializeOptimization, optimize}, VertexSE3Expmap, VertexSBAPointXYZ, CameraParameters,
// EdgeProjectXYZ2UV, RobustKernelHuber, SE3Quat.  The classes only RECORD the graph; optimize(n) flattens it and
// runs the MI355X LM solver through the C-ABI, then writes the estimates back into the vertices.  Ownership as in
// g2o: the optimizer deletes the vertices, edges, parameters, algorithm and solver handed to it.
// Self-contained small-vector types replace Eigen (not installed): Vector2d/3d, Matrix2d/3d, Quaterniond.
#ifndef MY_SLAM_G2O_FACADE_H
#define MY_SLAM_G2O_FACADE_H
#include <cmath>
#include <functional>
#include <map>
#include <vector>

#include "my_slam/common_include.h"

namespace Eigen {
struct Vector2d {
    double v[2] = {0, 0};
    Vector2d() {}
    Vector2d(double a, double b) { v[0] = a, v[1] = b; }
    double& operator()(int i, int = 0) { return v[i]; }
    double operator()(int i, int = 0) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};
struct Vector3d {
    double v[3] = {0, 0, 0};
    Vector3d() {}
    Vector3d(double a, double b, double c) { v[0] = a, v[1] = b, v[2] = c; }
    double& operator()(int i, int = 0) { return v[i]; }
    double operator()(int i, int = 0) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};
struct Matrix2d {
    double m[4] = {1, 0, 0, 1};
    static Matrix2d Identity() { return Matrix2d(); }
    double& operator()(int r, int c) { return m[2 * r + c]; }
    double operator()(int r, int c) const { return m[2 * r + c]; }
};
struct Matrix3d {
    double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    static Matrix3d Identity() { return Matrix3d(); }
    double& operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
};
}  // namespace Eigen

namespace g2o {

// world -> camera rigid transform as (R, t); g2o::SE3Quat(R, t), .rotation() / .translation()
class SE3Quat {
public:
    SE3Quat() {}
    SE3Quat(const Eigen::Matrix3d& R, const Eigen::Vector3d& t) : R_(R), t_(t) {}
    const Eigen::Matrix3d& rotation() const { return R_; }  // (g2o returns a quaternion; a matrix serves the callers)
    const Eigen::Vector3d& translation() const { return t_; }

private:
    Eigen::Matrix3d R_;
    Eigen::Vector3d t_;
};

class Vertex {
public:
    virtual ~Vertex() {}
    void setId(int id) { id_ = id; }
    int id() const { return id_; }
    void setFixed(bool f) { fixed_ = f; }
    bool fixed() const { return fixed_; }
    void setMarginalized(bool m) { marginalized_ = m; }

private:
    int id_ = -1;
    bool fixed_ = false, marginalized_ = false;
};
class VertexSE3Expmap : public Vertex {
public:
    void setEstimate(const SE3Quat& e) { est_ = e; }
    const SE3Quat& estimate() const { return est_; }

private:
    SE3Quat est_;
};
class VertexSBAPointXYZ : public Vertex {
public:
    void setEstimate(const Eigen::Vector3d& e) { est_ = e; }
    const Eigen::Vector3d& estimate() const { return est_; }

private:
    Eigen::Vector3d est_;
};
class CameraParameters {
public:
    CameraParameters(double focal_length, const Eigen::Vector2d& principle_point, double /*baseline*/)
        : focal_length(focal_length), principle_point(principle_point) {}
    void setId(int id) { id_ = id; }
    int id() const { return id_; }
    double focal_length;
    Eigen::Vector2d principle_point;

private:
    int id_ = 0;
};
class RobustKernel {
public:
    virtual ~RobustKernel() {}
    virtual double delta() const = 0;
};
class RobustKernelHuber : public RobustKernel {
public:
    void setDelta(double d) { delta_ = d; }
    double delta() const override { return delta_; }

private:
    double delta_ = 1.0;
};
class EdgeProjectXYZ2UV {
public:
    ~EdgeProjectXYZ2UV() { delete kernel_; }
    void setId(int id) { id_ = id; }
    void setVertex(int i, Vertex* v) { (i == 0 ? point_ : pose_) = v; }  // 0 = XYZ point, 1 = camera pose
    void setMeasurement(const Eigen::Vector2d& m) { meas_ = m; }
    void setParameterId(int, int id) { param_id_ = id; }
    void setInformation(const Eigen::Matrix2d& i) { info_ = i; }
    void setRobustKernel(RobustKernel* k) {
        delete kernel_;
        kernel_ = k;
    }
    Vertex *point_ = nullptr, *pose_ = nullptr;
    Eigen::Vector2d meas_;
    Eigen::Matrix2d info_;
    RobustKernel* kernel_ = nullptr;
    int id_ = -1, param_id_ = 0;
};

// solver stack: type-compatible shells (the linear algebra lives in the HIP kernel)
template <int P, int L>
struct BlockSolverTraits {
    struct PoseMatrixType {};
};
template <class M>
struct LinearSolver {
    virtual ~LinearSolver() {}
};
template <class M>
struct LinearSolverDense : LinearSolver<M> {};
template <class M>
struct LinearSolverCSparse : LinearSolver<M> {};
template <class Traits>
class BlockSolver {
public:
    typedef typename Traits::PoseMatrixType PoseMatrixType;
    typedef LinearSolver<PoseMatrixType> LinearSolverType;
    explicit BlockSolver(LinearSolverType* ls) : ls_(ls) {}
    ~BlockSolver() { delete ls_; }

private:
    LinearSolverType* ls_;
};
class OptimizationAlgorithm {
public:
    virtual ~OptimizationAlgorithm() {}
};
class OptimizationAlgorithmLevenberg : public OptimizationAlgorithm {
public:
    template <class Solver>
    explicit OptimizationAlgorithmLevenberg(Solver* s) : deleter_([s] { delete s; }) {}
    ~OptimizationAlgorithmLevenberg() override { deleter_(); }

private:
    std::function<void()> deleter_;
};

class SparseOptimizer {
public:
    ~SparseOptimizer() {
        for (auto& kv : vertices_) delete kv.second;
        for (auto* e : edges_) delete e;
        for (auto* p : params_) delete p;
        delete algorithm_;
    }
    void setAlgorithm(OptimizationAlgorithm* a) { algorithm_ = a; }
    void setVerbose(bool) {}
    bool addVertex(Vertex* v) { return vertices_.emplace(v->id(), v).second; }
    bool addParameter(CameraParameters* p) {
        params_.push_back(p);
        return true;
    }
    bool addEdge(EdgeProjectXYZ2UV* e) {
        edges_.push_back(e);
        return true;
    }
    Vertex* vertex(int id) {
        auto it = vertices_.find(id);
        return it == vertices_.end() ? nullptr : it->second;
    }
    bool initializeOptimization() { return true; }
    int optimize(int iterations) {
        if (params_.empty()) throw std::runtime_error("SparseOptimizer: no CameraParameters added");
        std::vector<VertexSE3Expmap*> poses;
        std::vector<VertexSBAPointXYZ*> points;
        std::map<Vertex*, int> slot;
        for (auto& kv : vertices_) {  // ascending vertex id
            if (auto* p = dynamic_cast<VertexSE3Expmap*>(kv.second)) {
                slot[p] = (int)poses.size();
                poses.push_back(p);
            } else if (auto* x = dynamic_cast<VertexSBAPointXYZ*>(kv.second)) {
                slot[x] = (int)points.size();
                points.push_back(x);
            }
        }
        bool all_pts_fixed = true, any_pt_fixed = false;
        for (auto* x : points) {
            all_pts_fixed &= x->fixed();
            any_pt_fixed |= x->fixed();
        }
        if (any_pt_fixed && !all_pts_fixed)
            throw std::runtime_error("g2o facade: points must be all fixed or all free (as g2o_ba.cpp sets them)");
        std::vector<double> T(16 * poses.size()), X(3 * points.size()), uv;
        std::vector<unsigned char> pfix(poses.size());
        for (size_t i = 0; i < poses.size(); ++i) {  // the C-ABI takes cam->world 4x4: invert world->cam (R, t)
            const auto& R = poses[i]->estimate().rotation();
            const auto& t = poses[i]->estimate().translation();
            double* o = &T[16 * i];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) o[4 * r + c] = R(c, r);
                o[4 * r + 3] = -(R(0, r) * t[0] + R(1, r) * t[1] + R(2, r) * t[2]);
            }
            o[12] = o[13] = o[14] = 0;
            o[15] = 1;
            pfix[i] = poses[i]->fixed();
        }
        for (size_t i = 0; i < points.size(); ++i)
            for (int k = 0; k < 3; ++k) X[3 * i + k] = points[i]->estimate()[k];
        std::vector<int> ep, el;
        for (auto* e : edges_) {
            ep.push_back(slot.at(e->pose_));
            el.push_back(slot.at(e->point_));
            uv.push_back(e->meas_[0]);
            uv.push_back(e->meas_[1]);
        }
        const CameraParameters* cam = params_[0];
        mvo_ba_problem pr{};
        pr.n_poses = (int)poses.size();
        pr.n_points = (int)points.size();
        pr.n_edges = (int)ep.size();
        pr.pose_T_w_c = T.data();
        pr.points = X.data();
        pr.edge_pose = ep.data();
        pr.edge_point = el.data();
        pr.edge_uv = uv.data();
        pr.focal = cam->focal_length;
        pr.cx = cam->principle_point[0];
        pr.cy = cam->principle_point[1];
        const Eigen::Matrix2d I = edges_.empty() ? Eigen::Matrix2d() : edges_[0]->info_;
        for (int i = 0; i < 4; ++i) pr.info[i] = I.m[i];
        pr.huber_delta = (!edges_.empty() && edges_[0]->kernel_) ? edges_[0]->kernel_->delta() : 1e100;
        pr.fix_points = (!points.empty() && all_pts_fixed) ? 1 : 0;
        pr.pose_fixed = pfix.data();
        pr.max_iterations = iterations;
        mvo_ba_stats st;
        my_slam::mvo_check(mvo_bundle_adjustment(my_slam::hot_path_ctx(), &pr, &st), "SparseOptimizer::optimize");
        for (size_t i = 0; i < poses.size(); ++i) {  // back to world->cam (R, t)
            const double* o = &T[16 * i];
            Eigen::Matrix3d R;
            Eigen::Vector3d t;
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) R(r, c) = o[4 * c + r];
                t[r] = -(o[0 + r] * o[3] + o[4 + r] * o[7] + o[8 + r] * o[11]);
            }
            poses[i]->setEstimate(SE3Quat(R, t));
        }
        if (!pr.fix_points)
            for (size_t i = 0; i < points.size(); ++i)
                points[i]->setEstimate(Eigen::Vector3d(X[3 * i], X[3 * i + 1], X[3 * i + 2]));
        return st.iterations;
    }

private:
    std::map<int, Vertex*> vertices_;
    std::vector<EdgeProjectXYZ2UV*> edges_;
    std::vector<CameraParameters*> params_;
    OptimizationAlgorithm* algorithm_ = nullptr;
};

}  // namespace g2o
#endif
