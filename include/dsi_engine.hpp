// dsi_engine.hpp -- header-only C++ adapter over the C ABI (dsi_engine.h) that re-creates
// the reference's class and method names on the hot path, so that the reference's
// process_1 / process_2 orchestration (process1.cpp, process2.cpp) can call the GPU engine
// with the code it already has:
//
//   Grid3D                       cartesian3dgrid/include/cartesian3dgrid/cartesian3dgrid.h:22-247
//   EMVS::ShapeDSI               mapper_emvs_stereo/include/mapper_emvs_stereo/mapper_emvs_stereo.hpp:40-65
//   EMVS::MapperEMVS             mapper_emvs_stereo.hpp:94-155
//   LinearTrajectory             mapper_emvs_stereo/include/mapper_emvs_stereo/trajectory.hpp:81-128
//
// Types the reference takes from third-party packages are replaced by plain structs:
//   dvs_msgs::Event                      -> dsi::Event {x, y, ts (seconds), polarity}
//   geometry_utils::Transformation       -> dsi::Transformation {t[3], q[4] = w,x,y,z}
//   image_geometry::PinholeCameraModel   -> dsi::PinholeCameraModel {width,height,fx,fy,cx,cy,lut}
//   cv::Mat (CV_32F / CV_8U)             -> cv::Mat itself (anything with create / ptr<T> / isContinuous / release, see
//                                           image_create below), or dsi::Image<float> / dsi::Image<uint8_t>
// Where the reference glog-CHECK-aborts or throws std::out_of_range this adapter throws
// dsi::Error (carrying the C status code).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "dsi_engine.h"

namespace dsi {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc)
{
    if (rc != DSI_OK) throw Error(rc, dsi_last_error());
}

struct Event {  // fields MapperEMVS reads from dvs_msgs::Event (mapper_emvs_stereo.cpp:91,131,134)
    uint16_t x = 0, y = 0;
    double ts = 0;  // seconds
    bool polarity = false;
};

struct Transformation {  // T_A_B as translation + unit quaternion (w,x,y,z)
    double t[3] = {0, 0, 0};
    double q[4] = {1, 0, 0, 0};
    void to7(double* p) const
    {
        p[0] = t[0]; p[1] = t[1]; p[2] = t[2];
        p[3] = q[0]; p[4] = q[1]; p[5] = q[2]; p[6] = q[3];
    }
    static Transformation from7(const double* p)
    {
        Transformation T;
        T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
        T.q[0] = p[3]; T.q[1] = p[4]; T.q[2] = p[5]; T.q[3] = p[6];
        return T;
    }
};

struct PinholeCameraModel {
    int width = 0, height = 0;      // fullResolution()
    float fx = 0, fy = 0, cx = 0, cy = 0;  // projection-matrix intrinsics (mapper_emvs_stereo.cpp:46-48)
    std::vector<float> rectified_points;   // optional LUT, 2*W*H, entry y*W+x (precomputeRectifiedPoints)
};

template <typename T>
struct Image {  // a single-channel image of T when the caller has no cv::Mat
    int rows = 0, cols = 0;
    std::vector<T> data;
    Image() = default;
    Image(int r, int c) : rows(r), cols(c), data((size_t)r * c) {}
    T& at(int y, int x) { return data[(size_t)y * cols + x]; }
    const T& at(int y, int x) const { return data[(size_t)y * cols + x]; }
};

// How the image-typed OUTPUTS of the path (Grid3D::collapseMaxZSlice, cartesian3dgrid.h:207; MapperEMVS::getDepthMapFromDSI,
// mapper_emvs_stereo.hpp:108-109) are written into the caller's image type -- the reference's is cv::Mat.  Every such
// member of this header is a template over the image type and goes through these three customisation points:
//   image_create<T>(img, rows, cols)  make img a rows x cols single-channel image of T, return its first pixel
//                                     (rows * cols contiguous elements)
//   image_data<T>(img)                first pixel of an existing image of T
//   image_release(img)                make img empty (cv::Mat() / Image<T>())
// Defaults: dsi::Image<T>, and any type with OpenCV's Mat members create(rows, cols, type), ptr<T>(row),
// isContinuous(), release() -- i.e. cv::Mat itself, with no OpenCV header needed here: the depth codes are the
// constants of <opencv2/core/hal/interface.h> (CV_8U 0, CV_32F 5; a single channel's type IS its depth code).
// Overload them (in the image type's namespace, or in namespace dsi before this header) for anything else.
template <typename T> struct cv_depth;
template <> struct cv_depth<uint8_t> { enum { value = 0 }; };  // CV_8U
template <> struct cv_depth<float> { enum { value = 5 }; };    // CV_32F

template <typename T>
inline T* image_create(Image<T>& img, int rows, int cols)
{
    img = Image<T>(rows, cols);
    return img.data.data();
}
template <typename T, typename ImgT>
inline auto image_create(ImgT& img, int rows, int cols)
    -> decltype(img.create(rows, cols, 0), img.isContinuous(), img.template ptr<T>(0))
{
    img.create(rows, cols, (int)cv_depth<T>::value);  // cv::Mat::create: a fresh matrix is continuous
    if (!img.isContinuous()) throw Error(DSI_ERR_INVALID, "image_create: the image type returned non-contiguous rows");
    return img.template ptr<T>(0);
}
template <typename T>
inline T* image_data(Image<T>& img)
{
    return img.data.data();
}
template <typename T, typename ImgT>
inline auto image_data(ImgT& img) -> decltype(img.isContinuous(), img.template ptr<T>(0))
{
    if (!img.isContinuous()) throw Error(DSI_ERR_INVALID, "image_data: non-contiguous image");
    return img.template ptr<T>(0);
}
template <typename T>
inline void image_release(Image<T>& img)
{
    img = Image<T>();
}
template <typename ImgT>
inline auto image_release(ImgT& img) -> decltype(img.release(), void())
{
    img.release();
}

// Customisation point for depth_map_dense, the 5th argument of getDepthMapFromDSI (mapper_emvs_stereo.cpp:430-436):
//     cv::Mat inpaint_mask = 1 - mask;
//     cv::inpaint(depth_cell_indices_filtered, inpaint_mask, depth_cell_indices_inpainted, 3, cv::INPAINT_TELEA);
// Telea's fast-marching inpainting is OpenCV arithmetic (photo module) and stays on the host side of the boundary, like
// the rectification of fisheye cameras above.  A maintainer with OpenCV adds, in namespace cv or in namespace dsi
// before this header,
//     inline bool inpaint_depth_cell_indices(const cv::Mat& filtered, const cv::Mat& inpaint_mask, cv::Mat& inpainted)
//     { cv::inpaint(filtered, inpaint_mask, inpainted, 3, cv::INPAINT_TELEA); return true; }
// and getDepthMapFromDSI then fills depth_map_dense with convertDepthIndicesToValues of the result (:436).  The default
// returns false: depth_map_dense is left EMPTY (image_release), never a guess.
template <typename ImgT>
inline bool inpaint_depth_cell_indices(const ImgT& /*filtered u8*/, const ImgT& /*inpaint_mask u8 = 1 - mask*/,
                                       ImgT& /*inpainted u8*/)
{
    return false;
}

// One GPU + one stream; shared by every Grid3D / MapperEMVS created from it.
class Context {
public:
    explicit Context(int device = 0) { check(dsi_context_create(device, &h_)); }
    ~Context()
    {
        // the C side refuses (and destroys nothing) while grids, mappers or batches of this context are alive: a wrong
        // destruction order would be a silent, permanent leak of the stream, the pool and the scratch -- say so
        if (dsi_context_destroy(h_) != DSI_OK) std::fprintf(stderr, "dsi::Context: NOT destroyed: %s\n", dsi_last_error());
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    dsi_context_t* handle() const { return h_; }
    void wait_for(Context& other) { check(dsi_context_wait_for(h_, other.h_)); }
    void synchronize() { check(dsi_context_synchronize(h_)); }

private:
    dsi_context_t* h_ = nullptr;
};

// The reference's classes are constructed without a device argument (MapperEMVS(cam, shape),
// Grid3D(dimX, dimY, dimZ): mapper_emvs_stereo.hpp:101, cartesian3dgrid.h:26).  Those arities use this
// process-wide context: GPU `DSI_DEVICE` (environment, default 0), created on first use.
inline Context& default_context()
{
    static Context ctx([] {
        const char* e = std::getenv("DSI_DEVICE");
        return e ? std::atoi(e) : 0;
    }());
    return ctx;
}

// How the reference's third-party value types are read.  Specialise (or overload) for other types;
// the defaults cover: timestamps that are a double or have .toSec() (ros::Time), poses that are a
// dsi::Transformation or have getPosition() / getRotation().w() ... (kindr::minimal::QuatTransformation,
// the reference's geometry_utils::Transformation), cameras that are a dsi::PinholeCameraModel or
// have fullResolution() / fx() / fy() / cx() / cy() (image_geometry::PinholeCameraModel).
inline double to_seconds(double t) { return t; }
template <typename TimeT>
inline auto to_seconds(const TimeT& t) -> decltype(t.toSec())
{
    return t.toSec();
}

inline void to_pose7(const Transformation& T, double* p) { T.to7(p); }
template <typename PoseT>
inline auto to_pose7(const PoseT& T, double* p) -> decltype(T.getRotation().w(), void())
{
    const auto pos = T.getPosition();
    const auto rot = T.getRotation();
    p[0] = pos[0]; p[1] = pos[1]; p[2] = pos[2];
    p[3] = rot.w(); p[4] = rot.x(); p[5] = rot.y(); p[6] = rot.z();
}

inline void camera_of(const PinholeCameraModel& c, PinholeCameraModel* out) { *out = c; }

// The distortion model the reference reads from cam.cameraInfo().distortion_model
// (mapper_emvs_stereo.cpp:62): "plumb_bob" -> the camera's own rectifyPoint, "fisheye" ->
// fisheye_rectifyPoint = cv::fisheye::undistortPoints(K, D, R, P) (:243-254, :256-299).  A camera type
// without cameraInfo() is taken as plumb_bob (image_geometry's default model).
template <typename CamT>
inline auto distortion_model_of(const CamT& c, int) -> decltype(std::string(c.cameraInfo().distortion_model))
{
    return std::string(c.cameraInfo().distortion_model);
}
template <typename CamT>
inline std::string distortion_model_of(const CamT&, long)
{
    return "plumb_bob";
}

// Customisation point for fisheye cameras: OpenCV's arithmetic stays on the host side of the boundary, so
// the caller overloads this for its camera type -- in that type's namespace (camera_of finds it by
// argument-dependent lookup) or in namespace dsi before this header -- with the reference's own four lines --
//   cv::Point2f raw32(x, y), rect32;  cv::fisheye::undistortPoints(src(raw32), dst(rect32), c.intrinsicMatrix(),
//   c.distortionCoeffs(), c.rotationMatrix(), c.fullProjectionMatrix());  *u = rect32.x; *v = rect32.y;
// (mapper_emvs_stereo.cpp:243-254).  The default refuses: a plumb_bob LUT for a fisheye lens would shift
// every vote without any error being reported.
template <typename CamT>
inline void fisheye_rectify_point(const CamT&, double, double, double*, double*)
{
    throw Error(DSI_ERR_INVALID,
                "fisheye distortion model: overload dsi::fisheye_rectify_point for this camera type "
                "(cv::fisheye::undistortPoints with K, D, R, P, mapper_emvs_stereo.cpp:243-254), or build the "
                "rectification LUT yourself and pass a dsi::PinholeCameraModel");
}

template <typename CamT>
inline auto camera_of(const CamT& c, PinholeCameraModel* out) -> decltype(c.fullResolution(), void())
{
    // mapper_emvs_stereo.cpp:34-48: size from fullResolution(), K from the projection matrix
    out->width = c.fullResolution().width;
    out->height = c.fullResolution().height;
    out->fx = (float)c.fx();
    out->fy = (float)c.fy();
    out->cx = (float)c.cx();
    out->cy = (float)c.cy();
    // precomputeRectifiedPoints (mapper_emvs_stereo.cpp:256-299): raw pixel -> rectified pixel by the model
    // the camera declares (OpenCV arithmetic stays on the host side of the boundary)
    const std::string model = distortion_model_of(c, 0);
    const bool plumb_bob = model == "plumb_bob", fisheye = model == "fisheye";
    if (!plumb_bob && !fisheye)  // the reference logs "Distortion model not set properly!" and goes on with an
                                 // uninitialised table (:289-293); here the caller gets to know
        throw Error(DSI_ERR_INVALID, "Distortion model not set properly: '" + model + "' (expected plumb_bob or fisheye)");
    out->rectified_points.resize((size_t)2 * out->width * out->height);
    for (int y = 0; y < out->height; ++y)
        for (int x = 0; x < out->width; ++x) {
            double u, v;
            if (plumb_bob) {
                const auto r = c.rectifyPoint(decltype(c.rectifyPoint({}))((double)x, (double)y));
                u = r.x;
                v = r.y;
            } else {
                fisheye_rectify_point(c, (double)x, (double)y, &u, &v);
            }
            out->rectified_points[2 * ((size_t)y * out->width + x)] = (float)u;
            out->rectified_points[2 * ((size_t)y * out->width + x) + 1] = (float)v;
        }
}

// One rank of an RCCL communicator owned by the engine (dsi_engine.h "multi-GPU").
class Comm {
public:
    Comm() = default;
    Comm(Context& ctx, const uint8_t id[DSI_COMM_ID_BYTES], int nranks, int rank)
    {
        check(dsi_comm_create_rank(ctx.handle(), id, nranks, rank, &h_));
    }
    ~Comm() { dsi_comm_destroy(h_); }
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    Comm(Comm&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    // one process, several GPUs: one communicator rank per context (distinct devices)
    static std::vector<Comm> createAll(const std::vector<Context*>& ctxs)
    {
        std::vector<dsi_context_t*> hs;
        for (Context* c : ctxs) hs.push_back(c->handle());
        std::vector<dsi_comm_t*> out(ctxs.size(), nullptr);
        check(dsi_comm_create_all(hs.data(), (int)hs.size(), out.data()));
        std::vector<Comm> v(ctxs.size());
        for (size_t i = 0; i < out.size(); ++i) v[i].h_ = out[i];
        return v;
    }
    static void uniqueId(uint8_t id[DSI_COMM_ID_BYTES]) { check(dsi_comm_unique_id(id)); }
    int rank() const { return dsi_comm_rank(h_); }
    int size() const { return dsi_comm_size(h_); }
    dsi_comm_t* handle() const { return h_; }

private:
    dsi_comm_t* h_ = nullptr;
};

}  // namespace dsi

namespace dsi {

// T_a * T_b and T^-1 for (translation, unit quaternion w,x,y,z) -- the two minkindr operations the
// callers need to place the reference view (process1.cpp:56-68, process2.cpp:79-81)
inline void quat_rotate(const double* q, const double* v, double* out)
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double ux = 2 * (y * v[2] - z * v[1]), uy = 2 * (z * v[0] - x * v[2]), uz = 2 * (x * v[1] - y * v[0]);
    out[0] = v[0] + w * ux + (y * uz - z * uy);
    out[1] = v[1] + w * uy + (z * ux - x * uz);
    out[2] = v[2] + w * uz + (x * uy - y * ux);
}

inline Transformation operator*(const Transformation& a, const Transformation& b)
{
    Transformation o;
    const double *p = a.q, *r = b.q;
    o.q[0] = p[0] * r[0] - p[1] * r[1] - p[2] * r[2] - p[3] * r[3];
    o.q[1] = p[0] * r[1] + p[1] * r[0] + p[2] * r[3] - p[3] * r[2];
    o.q[2] = p[0] * r[2] + p[2] * r[0] + p[3] * r[1] - p[1] * r[3];
    o.q[3] = p[0] * r[3] + p[3] * r[0] + p[1] * r[2] - p[2] * r[1];
    double rt[3];
    quat_rotate(a.q, b.t, rt);
    for (int i = 0; i < 3; ++i) o.t[i] = a.t[i] + rt[i];
    return o;
}

inline Transformation inverse(const Transformation& T)
{
    Transformation o;
    o.q[0] = T.q[0];
    o.q[1] = -T.q[1];
    o.q[2] = -T.q[2];
    o.q[3] = -T.q[3];
    double rt[3];
    quat_rotate(o.q, T.t, rt);
    for (int i = 0; i < 3; ++i) o.t[i] = -rt[i];
    return o;
}

}  // namespace dsi

// trajectory.hpp:81-128
class LinearTrajectory {
public:
    typedef std::map<double, dsi::Transformation> PoseMap;
    LinearTrajectory() = default;
    // any std::map<TimeT, PoseT> the traits above can read (the reference's
    // std::map<ros::Time, geometry_utils::Transformation>, trajectory.hpp:84)
    template <typename TimeT, typename PoseT, typename... Rest>
    explicit LinearTrajectory(const std::map<TimeT, PoseT, Rest...>& poses)
    {
        if (poses.size() < 2) throw dsi::Error(DSI_ERR_INVALID, "At least two poses need to be provided");
        for (const auto& kv : poses) {
            times_.push_back(dsi::to_seconds(kv.first));
            double p[7];
            dsi::to_pose7(kv.second, p);
            poses_.insert(poses_.end(), p, p + 7);
        }
    }
    // Returns T_W_C; false when t cannot be interpolated (trajectory.hpp:98-113)
    bool getPoseAt(double t, dsi::Transformation& T) const
    {
        double out[7];
        if (dsi_pose_at(times_.data(), poses_.data(), times_.size(), t, out) != DSI_OK) return false;
        T = dsi::Transformation::from7(out);
        return true;
    }
    // the reference's signature getPoseAt(const ros::Time&, Transformation&) for any time type with
    // toSec() and any pose type constructible from (rotation(w,x,y,z), position(x,y,z)) like minkindr's
    template <typename TimeT, typename PoseT>
    auto getPoseAt(const TimeT& t, PoseT& T) const -> decltype(t.toSec(), T.getRotation(), bool())
    {
        dsi::Transformation D;
        if (!getPoseAt(t.toSec(), D)) return false;
        typedef typename std::decay<decltype(T.getRotation())>::type Rot;
        typedef typename std::decay<decltype(T.getPosition())>::type Pos;
        T = PoseT(Rot(D.q[0], D.q[1], D.q[2], D.q[3]), Pos(D.t[0], D.t[1], D.t[2]));
        return true;
    }
    // TrajectoryBase::applyTransformationRight / Left (trajectory.hpp:57-71): every control pose becomes pose * T / T * pose
    // -- how main.cpp:203-216 turns the recorded poses into the left camera's (hand-eye) and the other cameras' (extrinsics)
    void applyTransformationRight(const dsi::Transformation& T)
    {
        for (size_t i = 0; i < times_.size(); ++i) (dsi::Transformation::from7(&poses_[7 * i]) * T).to7(&poses_[7 * i]);
    }
    void applyTransformationLeft(const dsi::Transformation& T)
    {
        for (size_t i = 0; i < times_.size(); ++i) (T * dsi::Transformation::from7(&poses_[7 * i])).to7(&poses_[7 * i]);
    }
    // ... with the reference's pose type (geometry_utils::Transformation)
    template <typename PoseT>
    auto applyTransformationRight(const PoseT& T) -> decltype(T.getRotation(), void())
    {
        double p[7];
        dsi::to_pose7(T, p);
        applyTransformationRight(dsi::Transformation::from7(p));
    }
    template <typename PoseT>
    auto applyTransformationLeft(const PoseT& T) -> decltype(T.getRotation(), void())
    {
        double p[7];
        dsi::to_pose7(T, p);
        applyTransformationLeft(dsi::Transformation::from7(p));
    }
    // TrajectoryBase::getFirstControlPose / getLastControlPose (trajectory.hpp:28-38)
    void getFirstControlPose(dsi::Transformation* T, double* t) const
    {
        if (times_.empty()) throw dsi::Error(DSI_ERR_INVALID, "empty trajectory");
        *t = times_.front();
        *T = dsi::Transformation::from7(&poses_[0]);
    }
    void getLastControlPose(dsi::Transformation* T, double* t) const
    {
        if (times_.empty()) throw dsi::Error(DSI_ERR_INVALID, "empty trajectory");
        *t = times_.back();
        *T = dsi::Transformation::from7(&poses_[7 * (times_.size() - 1)]);
    }
    size_t getNumControlPoses() const { return times_.size(); }
    const std::vector<double>& times() const { return times_; }
    const std::vector<double>& poses7() const { return poses_; }

private:
    std::vector<double> times_, poses_;
};

// cartesian3dgrid.h:22-247, device resident.
class Grid3D {
public:
    Grid3D() = default;
    Grid3D(dsi::Context& ctx, unsigned dimX, unsigned dimY, unsigned dimZ) { allocate(ctx, dimX, dimY, dimZ); }
    // the reference's arity (cartesian3dgrid.h:26), on the process-wide default context
    Grid3D(unsigned dimX, unsigned dimY, unsigned dimZ) { allocate(dsi::default_context(), dimX, dimY, dimZ); }
    ~Grid3D() { deallocate(); }
    Grid3D(const Grid3D&) = delete;
    Grid3D& operator=(const Grid3D&) = delete;
    Grid3D(Grid3D&& o) noexcept : h_(o.h_), owned_(o.owned_) { o.h_ = nullptr; }
    Grid3D& operator=(Grid3D&& o) noexcept
    {
        if (this != &o) {
            deallocate();
            h_ = o.h_;
            owned_ = o.owned_;
            o.h_ = nullptr;
        }
        return *this;
    }

    void allocate(dsi::Context& ctx, unsigned dimX, unsigned dimY, unsigned dimZ)
    {
        deallocate();
        dsi::check(dsi_grid_create(ctx.handle(), (int)dimX, (int)dimY, (int)dimZ, &h_));
        owned_ = true;
    }
    void deallocate()
    {
        if (h_ && owned_) dsi_grid_destroy(h_);
        h_ = nullptr;
    }
    // non-owning view of a grid that belongs to a mapper (MapperEMVS::dsi_)
    static Grid3D view(dsi_grid_t* h)
    {
        Grid3D g;
        g.h_ = h;
        g.owned_ = false;
        return g;
    }
    dsi_grid_t* handle() const { return h_; }

    void getDimensions(int* dimX, int* dimY, int* dimZ) const { dsi::check(dsi_grid_dims(h_, dimX, dimY, dimZ)); }
    void resetGrid() { dsi::check(dsi_grid_reset(h_)); }

    // voxel-wise operations, same names and in-place semantics as cartesian3dgrid.h:64-192
    void addTwoGrids(const Grid3D& grid2) { dsi::check(dsi_grid_accumulate(h_, grid2.h_, DSI_ACC_SUM)); }
    void addInverseOfTwoGrids(const Grid3D& grid2) { dsi::check(dsi_grid_accumulate(h_, grid2.h_, DSI_ACC_INV_SUM)); }
    void computeHMfromSumOfInv(int n) { dsi::check(dsi_grid_finalize(h_, DSI_ACC_INV_SUM, n)); }
    void computeAMfromSum(int n) { dsi::check(dsi_grid_finalize(h_, DSI_ACC_SUM, n)); }
    void minTwoGrids(const Grid3D& grid2) { dsi::check(dsi_grid_fuse2(h_, grid2.h_, DSI_FUSE_MIN)); }
    void harmonicMeanTwoGrids(const Grid3D& grid2) { dsi::check(dsi_grid_fuse2(h_, grid2.h_, DSI_FUSE_HM)); }
    void harmonicMeanTwoGrids(const Grid3D& grid2, int n) { dsi::check(dsi_grid_fuse_hm_n(h_, grid2.h_, n)); }
    void rmsTwoGrids(const Grid3D& grid2) { dsi::check(dsi_grid_fuse2(h_, grid2.h_, DSI_FUSE_RMS)); }
    void geometricMeanTwoGrids(const Grid3D& grid2) { dsi::check(dsi_grid_fuse2(h_, grid2.h_, DSI_FUSE_GM)); }
    void arithmeticMeanTwoGrids(const Grid3D& grid2) { dsi::check(dsi_grid_fuse2(h_, grid2.h_, DSI_FUSE_AM)); }
    void maxTwoGrids(const Grid3D& grid2) { dsi::check(dsi_grid_fuse2(h_, grid2.h_, DSI_FUSE_MAX)); }
    // n-ary accumulate / finalize (dsi_acc_mode_t: the temporal accumulators above are modes 0 / 1;
    // LOG_SUM / SQ_SUM / MIN / MAX are the n-ary forms of the 2-ary camera-fusion ops, which the
    // reference lacks) and the all-reduce that joins accumulators across GPUs
    void accumulateBegin(int mode) { dsi::check(dsi_grid_accumulate_begin(h_, mode)); }
    void accumulate(const Grid3D& grid2, int mode) { dsi::check(dsi_grid_accumulate(h_, grid2.h_, mode)); }
    void finalize(int mode, int n) { dsi::check(dsi_grid_finalize(h_, mode, n)); }
    void allReduce(dsi::Comm& comm, int op) { dsi::check(dsi_grid_allreduce(comm.handle(), h_, op)); }

    // void collapseMaxZSlice(cv::Mat* max_val, cv::Mat* max_pos) const (cartesian3dgrid.h:207, cartesian3dgrid.cpp:115-137):
    // max_val becomes dimY x dimX CV_32F, max_pos dimY x dimX CV_8U ("Max 256 depth layers", :120).  Any image type the
    // dsi::image_create customisation point can write: cv::Mat (both arguments), dsi::Image<float> / <uint8_t>.
    template <typename ValImg, typename PosImg>
    void collapseMaxZSlice(ValImg* max_val, PosImg* max_pos) const
    {
        int nx, ny, nz;
        getDimensions(&nx, &ny, &nz);
        float* val = dsi::image_create<float>(*max_val, ny, nx);
        uint8_t* pos = dsi::image_create<uint8_t>(*max_pos, ny, nx);
        dsi::check(dsi_grid_collapse_max_z(h_, val, pos));
    }
    // cartesian3dgrid.cpp:164-174
    double computeMeanSquare() const
    {
        double v = 0;
        dsi::check(dsi_grid_mean_square(h_, &v));
        return v;
    }
    // host copies (the reference exposes raw pointers via getPointerToSlice; device memory
    // cannot be handed out like that, dsi_grid_device_ptr() is the device-side equivalent)
    std::vector<float> download() const
    {
        int nx, ny, nz;
        getDimensions(&nx, &ny, &nz);
        std::vector<float> v((size_t)nx * ny * nz);
        dsi::check(dsi_grid_download(h_, v.data()));
        return v;
    }
    void upload(const std::vector<float>& v) { dsi::check(dsi_grid_upload(h_, v.data())); }

    // Grid3D::writeGridNpy (cartesian3dgrid_IO.cpp:30-36): NumPy .npy v1.0, little-endian float32,
    // C order, shape (dimZ, dimY, dimX) -- what scripts/visualize_dsi_*.py load.  Returns 0 on success.
    int writeGridNpy(const char* filename) const
    {
        int nx, ny, nz;
        getDimensions(&nx, &ny, &nz);
        const std::vector<float> v = download();
        std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(nz) + ", " +
                           std::to_string(ny) + ", " + std::to_string(nx) + "), }";
        while ((10 + dict.size() + 1) % 64 != 0) dict += ' ';  // header padded to a multiple of 64 bytes
        dict += '\n';
        std::FILE* f = std::fopen(filename, "wb");
        if (!f) return 1;
        const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
        const unsigned short hl = (unsigned short)dict.size();
        const unsigned char hlen[2] = {(unsigned char)(hl & 0xff), (unsigned char)(hl >> 8)};
        bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(hlen, 1, 2, f) == 2 &&
                  std::fwrite(dict.data(), 1, dict.size(), f) == dict.size() &&
                  std::fwrite(v.data(), sizeof(float), v.size(), f) == v.size();
        ok = (std::fclose(f) == 0) && ok;
        return ok ? 0 : 1;
    }

private:
    dsi_grid_t* h_ = nullptr;
    bool owned_ = false;
};

namespace EMVS {

struct ShapeDSI {  // mapper_emvs_stereo.hpp:40-65
    ShapeDSI() = default;
    ShapeDSI(size_t dimX, size_t dimY, size_t dimZ, float min_depth, float max_depth, float fov)
        : dimX_(dimX), dimY_(dimY), dimZ_(dimZ), min_depth_(min_depth), max_depth_(max_depth), fov_(fov)
    {
    }
    size_t dimX_ = 0, dimY_ = 0, dimZ_ = 100;
    float min_depth_ = 0.3f, max_depth_ = 5.f;
    float fov_ = 0.f;
};

struct OptionsDepthMap {  // mapper_emvs_stereo.hpp:68-82
    // the fields the extraction reads
    int adaptive_threshold_kernel_size_ = 5;
    double adaptive_threshold_c_ = 5.;
    double max_confidence = 0.;
    int median_filter_size_ = 5;
    // the fields main.cpp:163-171 also sets: carried for source compatibility -- they steer file output on the host
    // (save_*, full_sequence) and the reference view along the baseline (rv_pos: process1.cpp:63, the rv_pos argument of
    // process_1 / process_1_depth_map / full_sequence_depth_maps in dsi_process.hpp)
    bool full_sequence = false;
    bool save_conf_stats = false;
    bool save_mono = false;
    bool save_dsi = false;
    double rv_pos = 0.;
};

typedef LinearTrajectory TrajectoryType;

class MapperEMVS {  // mapper_emvs_stereo.hpp:94-155
public:
    // plane_begin / plane_count: own only that range of the dimZ planes (plane sharding over GPUs;
    // 0, 0 = all planes, the reference behaviour)
    // the reference's arity MapperEMVS(cam, dsi_shape) (mapper_emvs_stereo.hpp:101) for any camera
    // type dsi::camera_of can read, on the process-wide default context
    template <typename CamT>
    MapperEMVS(const CamT& cam, const ShapeDSI& dsi_shape) : MapperEMVS(dsi::default_context(), convert(cam), dsi_shape)
    {
    }
    MapperEMVS(dsi::Context& ctx, const dsi::PinholeCameraModel& cam, const ShapeDSI& dsi_shape,
               bool inverse_depth = false, int plane_begin = 0, int plane_count = 0)
    {
        dsi_mapper_config_t cfg{};
        cfg.plane_begin = plane_begin;
        cfg.plane_count = plane_count;
        cfg.sensor_width = cam.width;
        cfg.sensor_height = cam.height;
        cfg.K[0] = cam.fx; cfg.K[1] = cam.fy; cfg.K[2] = cam.cx; cfg.K[3] = cam.cy;
        cfg.dim_x = (int)dsi_shape.dimX_;
        cfg.dim_y = (int)dsi_shape.dimY_;
        cfg.dim_z = (int)dsi_shape.dimZ_;
        cfg.min_depth = dsi_shape.min_depth_;
        cfg.max_depth = dsi_shape.max_depth_;
        cfg.fov_deg = dsi_shape.fov_;
        cfg.inverse_depth = inverse_depth ? 1 : 0;
        cfg.lut = cam.rectified_points.empty() ? nullptr : cam.rectified_points.data();
        dsi::check(dsi_mapper_create(ctx.handle(), &cfg, &h_));
        ctx_ = ctx.handle();
        dsi_ = Grid3D::view(dsi_mapper_grid(h_));
    }
    ~MapperEMVS()
    {
        dsi_.deallocate();
        dsi_mapper_destroy(h_);
    }
    MapperEMVS(const MapperEMVS&) = delete;
    MapperEMVS& operator=(const MapperEMVS&) = delete;

    // mapper_emvs_stereo.cpp:67-148.  Returns false when events.size() < 1024.  Any event type with
    // .x .y .ts (ts a double in seconds or with .toSec(): dvs_msgs::Event) and any pose type
    // dsi::to_pose7 can read (geometry_utils::Transformation) -- the reference's call
    // evaluateDSI(events, trajectory, T_rv_w) (process1.cpp:76) compiles as it stands.
    template <typename EventT, typename PoseT>
    bool evaluateDSI(const std::vector<EventT>& events, const TrajectoryType& trajectory, const PoseT& T_rv_w)
    {
        const size_t n = events.size();
        xs_.resize(n);
        ys_.resize(n);
        ts_.resize(n);
        for (size_t i = 0; i < n; ++i) {
            xs_[i] = (uint16_t)events[i].x;
            ys_[i] = (uint16_t)events[i].y;
            ts_[i] = dsi::to_seconds(events[i].ts);
        }
        double T7[7];
        dsi::to_pose7(T_rv_w, T7);
        size_t voted = 0;
        const int rc = dsi_mapper_evaluate(h_, xs_.data(), ys_.data(), ts_.data(), n, trajectory.times().data(),
                                           trajectory.poses7().data(), trajectory.times().size(), T7, &voted);
        if (rc == DSI_ERR_TOO_FEW_EVENTS) return false;
        dsi::check(rc);
        events_voted_ = voted;
        return true;
    }

    // The device part of getDepthMapFromDSI (mapper_emvs_stereo.cpp:339-437) WITHOUT its filters: arg-max over Z
    // (:368) and convertDepthIndicesToValues (:302-313) on the raw indices -- the map SURVEY 8(a) A11/A12 and the
    // "depth map equal to the CPU reference" statement are about.  Not a member of the reference (it keeps
    // depth_cell_indices local); hence its own name's third argument is the index map, not a mask.
    template <typename DepthImg, typename ConfImg, typename IdxImg>
    void getDepthMapFromDSI(DepthImg& depth_map, ConfImg& confidence_map, IdxImg& depth_cell_indices)
    {
        int nx, ny, nz;
        dsi_.getDimensions(&nx, &ny, &nz);
        float* depth = dsi::image_create<float>(depth_map, ny, nx);
        float* conf = dsi::image_create<float>(confidence_map, ny, nx);
        uint8_t* idx = dsi::image_create<uint8_t>(depth_cell_indices, ny, nx);
        dsi::check(dsi_mapper_depth_map(h_, depth, conf, idx));
    }

    // void getDepthMapFromDSI(cv::Mat& depth_map, cv::Mat& confidence_map, cv::Mat& mask, const OptionsDepthMap&,
    //                         int method = -1)                                   mapper_emvs_stereo.hpp:108, .cpp:331-335
    // void getDepthMapFromDSI(cv::Mat& depth_map, cv::Mat& confidence_map, cv::Mat& mask, const OptionsDepthMap&,
    //                         cv::Mat& depth_map_dense, int method = -1)         mapper_emvs_stereo.hpp:109, .cpp:338-437
    // on the device, from the mapper's OWN dsi_ like the reference (process1.cpp:208-222, process2.cpp:123-299,
    // process5.cpp:258, main.cpp:389-417 compile as spelled there): arg-max (:368), conf(0,0) = max_confidence +
    // normalisation (:393-397), Gaussian adaptive threshold (:403-409), masked Huang median (:420-423),
    // removeMaskBoundary (:426-427), index -> depth of the filtered indices (:435).  Outputs as the reference leaves
    // them: depth_map CV_32F, confidence_map CV_32F (element (0,0) overwritten), mask CV_8U in {0, 1}.
    // method: the reference's switch (:348-368) takes 0..4 to the focus-based collapses (collapseZSliceByLocalVar ...,
    // never selected by any caller: every call site passes the default) and everything else to collapseMaxZSlice;
    // 0..4 are refused with dsi::Error (DSI_ERR_BAD_OP) -- SURVEY 2 marks them out of scope -- anything else is the arg-max.
    // depth_map_dense: see dsi::inpaint_depth_cell_indices above -- filled when the caller supplies OpenCV's inpainting,
    // otherwise left empty.
    template <typename DepthImg, typename ConfImg, typename MaskImg>
    void getDepthMapFromDSI(DepthImg& depth_map, ConfImg& confidence_map, MaskImg& mask,
                            const OptionsDepthMap& options_depth_map, int method = -1)
    {
        extract(nullptr, depth_map, confidence_map, mask, options_depth_map, (MaskImg*)nullptr, (DepthImg*)nullptr, method);
    }
    template <typename DepthImg, typename ConfImg, typename MaskImg, typename DenseImg,
              typename = typename std::enable_if<!std::is_pointer<DenseImg>::value && !std::is_arithmetic<DenseImg>::value>::type>
    void getDepthMapFromDSI(DepthImg& depth_map, ConfImg& confidence_map, MaskImg& mask,
                            const OptionsDepthMap& options_depth_map, DenseImg& depth_map_dense, int method = -1)
    {
        MaskImg idx_filtered;
        extract(nullptr, depth_map, confidence_map, mask, options_depth_map, &idx_filtered, &depth_map_dense, method);
    }
    // the same for a DSI other than the mapper's own (not in the reference, which copies the DSI into a mapper first)
    template <typename DepthImg, typename ConfImg, typename MaskImg>
    void getDepthMapFromDSI(DepthImg& depth_map, ConfImg& confidence_map, MaskImg& mask,
                            const OptionsDepthMap& options_depth_map, const Grid3D* grid)
    {
        extract(grid, depth_map, confidence_map, mask, options_depth_map, (MaskImg*)nullptr, (DepthImg*)nullptr, -1);
    }

    // The filters of getDepthMapFromDSI (mapper_emvs_stereo.cpp:390-437) on the raw depth map this mapper already
    // holds on the device -- after dsi::process_1_depth_map, which votes, fuses and takes the arg-max without ever
    // writing the DSI the reference would call getDepthMapFromDSI on.  Same outputs as the overloads above.
    template <typename DepthImg, typename ConfImg, typename MaskImg>
    void filterDepthMap(DepthImg& depth_map, ConfImg& confidence_map, MaskImg& mask, const OptionsDepthMap& options_depth_map)
    {
        int nx, ny, nz;
        dsi_.getDimensions(&nx, &ny, &nz);
        float* depth = dsi::image_create<float>(depth_map, ny, nx);
        float* conf = dsi::image_create<float>(confidence_map, ny, nx);
        uint8_t* mk = dsi::image_create<uint8_t>(mask, ny, nx);
        const dsi_depthmap_options_t o = options_of(options_depth_map);
        dsi::check(dsi_mapper_filter_depth_map(h_, &o, depth, conf, mk, nullptr));
    }

    // MapperEMVS::convertDepthIndicesToValues (mapper_emvs_stereo.cpp:302-313) on host images: depth = cellIndexToDepth(index)
    template <typename IdxImg, typename DepthImg>
    void convertDepthIndicesToValues(IdxImg& depth_cell_indices, DepthImg& depth_map)
    {
        int nx, ny, nz;
        dsi_.getDimensions(&nx, &ny, &nz);
        int dim_z = 0;
        dsi::check(dsi_mapper_full_depths(h_, nullptr, &dim_z));
        std::vector<float> z((size_t)dim_z);
        dsi::check(dsi_mapper_full_depths(h_, z.data(), nullptr));
        const uint8_t* idx = dsi::image_data<uint8_t>(depth_cell_indices);
        float* depth = dsi::image_create<float>(depth_map, ny, nx);
        for (size_t i = 0; i < (size_t)nx * ny; ++i) {
            if ((int)idx[i] >= dim_z) throw dsi::Error(DSI_ERR_INVALID, "convertDepthIndicesToValues: index beyond dimZ");
            depth[i] = z[idx[i]];
        }
    }

    std::vector<float> depthPlanes() const
    {
        int nz = 0;
        dsi::check(dsi_mapper_geometry(h_, nullptr, nullptr, nullptr, nullptr, &nz));
        std::vector<float> z(nz);
        dsi::check(dsi_mapper_geometry(h_, nullptr, z.data(), nullptr, nullptr, nullptr));
        return z;
    }
    size_t eventsVoted() const { return events_voted_; }
    dsi_mapper_t* handle() const { return h_; }
    dsi_context_t* context() const { return ctx_; }

    Grid3D dsi_;       // public member, as in the reference (mapper_emvs_stereo.hpp:116)
    std::string name;  // mapper_emvs_stereo.hpp:117

private:
    static dsi_depthmap_options_t options_of(const OptionsDepthMap& options_depth_map)
    {
        dsi_depthmap_options_t o{};
        o.adaptive_threshold_kernel_size = options_depth_map.adaptive_threshold_kernel_size_;
        o.adaptive_threshold_c = options_depth_map.adaptive_threshold_c_;
        o.median_filter_size = options_depth_map.median_filter_size_;
        o.max_confidence = options_depth_map.max_confidence;
        return o;
    }
    template <typename DepthImg, typename ConfImg, typename MaskImg, typename DenseImg>
    void extract(const Grid3D* grid, DepthImg& depth_map, ConfImg& confidence_map, MaskImg& mask,
                 const OptionsDepthMap& options_depth_map, MaskImg* idx_filtered, DenseImg* depth_map_dense, int method)
    {
        if (method >= 0 && method <= 4)
            throw dsi::Error(DSI_ERR_BAD_OP, "getDepthMapFromDSI: method " + std::to_string(method) +
                                                 " is one of the focus-based collapses (collapseZSliceByLocalVar / "
                                                 "LocalMeanSquare / GradMag / LaplacianMag / DoG, mapper_emvs_stereo.cpp:350-364), "
                                                 "which this engine does not provide; pass the default (-1): collapseMaxZSlice");
        int nx, ny, nz;
        dsi_.getDimensions(&nx, &ny, &nz);
        float* depth = dsi::image_create<float>(depth_map, ny, nx);
        float* conf = dsi::image_create<float>(confidence_map, ny, nx);
        uint8_t* mk = dsi::image_create<uint8_t>(mask, ny, nx);
        uint8_t* filtered = idx_filtered ? dsi::image_create<uint8_t>(*idx_filtered, ny, nx) : nullptr;
        const dsi_depthmap_options_t o = options_of(options_depth_map);
        dsi::check(dsi_mapper_get_depth_map_from_dsi(h_, grid ? grid->handle() : nullptr, &o, depth, conf, mk, filtered));
        if (!depth_map_dense) return;
        // mapper_emvs_stereo.cpp:430-436
        MaskImg inpaint_mask, inpainted;
        uint8_t* im = dsi::image_create<uint8_t>(inpaint_mask, ny, nx);
        for (size_t i = 0; i < (size_t)nx * ny; ++i) im[i] = (uint8_t)(1 - mk[i]);
        using dsi::inpaint_depth_cell_indices;  // the default; an overload for MaskImg found by ADL wins over it
        if (inpaint_depth_cell_indices(*idx_filtered, inpaint_mask, inpainted))
            convertDepthIndicesToValues(inpainted, *depth_map_dense);
        else
            dsi::image_release(*depth_map_dense);
    }
    template <typename CamT>
    static dsi::PinholeCameraModel convert(const CamT& cam)
    {
        dsi::PinholeCameraModel c;
        dsi::camera_of(cam, &c);
        return c;
    }
    dsi_mapper_t* h_ = nullptr;
    dsi_context_t* ctx_ = nullptr;
    std::vector<uint16_t> xs_, ys_;
    std::vector<double> ts_;
    size_t events_voted_ = 0;
};

}  // namespace EMVS
