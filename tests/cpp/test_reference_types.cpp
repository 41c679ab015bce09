// The adapter against the reference's OWN value types.  The reference calls the hot path as
//
//     EMVS::MapperEMVS mapper0(cam0, dsi_shape);                          main.cpp:262-275
//     trajectory0.getPoseAt(ros::Time(t_mid), T_w_l);                     process1.cpp:62
//     T_rv_w = (T_w_l * baselineTransform).inverse();                     process1.cpp:64-68
//     mapper0.evaluateDSI(events0, trajectory0, T_rv_w);                  process1.cpp:76
//     mapper_fused.dsi_.resetGrid(); ...addTwoGrids(mapper0.dsi_); ...    process1.cpp:126-158
//
// with std::vector<dvs_msgs::Event>, ros::Time, geometry_utils::Transformation (a
// kindr::minimal::QuatTransformation), std::map<ros::Time, Transformation> and
// image_geometry::PinholeCameraModel.  None of those packages exists in this image, so this file
// declares minimal stand-ins WITH THE SAME MEMBER NAMES (test scaffolding only: no arithmetic of
// the path lives in them) and runs that call sequence through include/dsi_engine.hpp unchanged in
// spelling and arity.  The result must equal, bit for bit, the same data pushed through the
// adapter's plain types (dsi::Event, dsi::Transformation, dsi::PinholeCameraModel).
//
//   exit 0: ok      exit 3: no GPU      else: failure
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "dsi_engine.hpp"
#include "dsi_process.hpp"

// ---------------------------------------------------------------- stand-ins (names as upstream)
namespace ros {
struct Time {
    double s = 0;
    Time() = default;
    explicit Time(double t) : s(t) {}
    double toSec() const { return s; }
    bool operator<(const Time& o) const { return s < o.s; }
};
}  // namespace ros

namespace dvs_msgs {
struct Event {
    uint16_t x = 0, y = 0;
    ros::Time ts;
    bool polarity = false;
};
}  // namespace dvs_msgs

namespace kindr {
namespace minimal {
struct Position {
    double v[3] = {0, 0, 0};
    Position() = default;
    Position(double x, double y, double z) : v{x, y, z} {}
    double operator[](int i) const { return v[i]; }
};
struct RotationQuaternion {
    double q[4] = {1, 0, 0, 0};
    RotationQuaternion() = default;
    RotationQuaternion(double w, double x, double y, double z) : q{w, x, y, z} {}
    double w() const { return q[0]; }
    double x() const { return q[1]; }
    double y() const { return q[2]; }
    double z() const { return q[3]; }
};
class QuatTransformation {
public:
    QuatTransformation() = default;
    QuatTransformation(const RotationQuaternion& r, const Position& p) : r_(r), p_(p) {}
    const Position& getPosition() const { return p_; }
    const RotationQuaternion& getRotation() const { return r_; }
    // the two group operations the caller needs; delegated to the adapter's helpers so that no
    // second implementation of the pose algebra exists in this test
    QuatTransformation operator*(const QuatTransformation& o) const { return from(to() * o.to()); }
    QuatTransformation inverse() const { return from(dsi::inverse(to())); }

private:
    dsi::Transformation to() const
    {
        dsi::Transformation T;
        for (int i = 0; i < 3; ++i) T.t[i] = p_[i];
        T.q[0] = r_.w(); T.q[1] = r_.x(); T.q[2] = r_.y(); T.q[3] = r_.z();
        return T;
    }
    static QuatTransformation from(const dsi::Transformation& T)
    {
        return QuatTransformation(RotationQuaternion(T.q[0], T.q[1], T.q[2], T.q[3]), Position(T.t[0], T.t[1], T.t[2]));
    }
    RotationQuaternion r_;
    Position p_;
};
}  // namespace minimal
}  // namespace kindr

namespace geometry_utils {
typedef kindr::minimal::QuatTransformation Transformation;  // geometry_utils.hpp:13
}

// OpenCV's depth codes are macros (<opencv2/core/hal/interface.h>)
#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 CV_8U
#define CV_32FC1 CV_32F

namespace cv {
struct Size {
    int width = 0, height = 0;
};
struct Point2d {
    double x = 0, y = 0;
    Point2d() = default;
    Point2d(double x_, double y_) : x(x_), y(y_) {}
};
// cv::Mat, single channel: the members the reference's call sites and the adapter's customisation points use
// (rows, cols, type(), at<T>(), ptr<T>(), create, release, isContinuous, empty, Mat(rows, cols, type); copies share
// the pixels like OpenCV's reference-counted header).  at<T>() and ptr<T>() check the element type like a debug
// build of OpenCV does -- a float written into a CV_8U image fails here, not silently.
class Mat {
public:
    int rows = 0, cols = 0;
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type)
    {
        if (type != CV_8U && type != CV_32F) throw std::runtime_error("cv::Mat look-alike: type not CV_8U / CV_32F");
        if (buf_ && r == rows && c == cols && type == type_) return;  // cv::Mat::create keeps a matching allocation
        rows = r;
        cols = c;
        type_ = type;
        buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * c * elemSize());
    }
    void release()
    {
        buf_.reset();
        rows = cols = 0;
    }
    int type() const { return type_; }
    size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    bool isContinuous() const { return true; }
    bool empty() const { return !buf_ || rows == 0 || cols == 0; }
    template <typename T>
    T* ptr(int row = 0)
    {
        check<T>();
        return reinterpret_cast<T*>(buf_->data()) + (size_t)row * cols;
    }
    template <typename T>
    const T* ptr(int row = 0) const
    {
        check<T>();
        return reinterpret_cast<const T*>(buf_->data()) + (size_t)row * cols;
    }
    template <typename T>
    T& at(int y, int x)
    {
        if (y < 0 || y >= rows || x < 0 || x >= cols) throw std::out_of_range("cv::Mat::at");
        return ptr<T>(y)[x];
    }
    template <typename T>
    const T& at(int y, int x) const
    {
        if (y < 0 || y >= rows || x < 0 || x >= cols) throw std::out_of_range("cv::Mat::at");
        return ptr<T>(y)[x];
    }

private:
    template <typename T>
    void check() const
    {
        if (!buf_) throw std::runtime_error("cv::Mat look-alike: empty matrix");
        if (sizeof(T) != elemSize()) throw std::runtime_error("cv::Mat look-alike: element type does not match type()");
    }
    int type_ = CV_8U;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};
// What a maintainer with OpenCV adds for depth_map_dense (include/dsi_engine.hpp, dsi::inpaint_depth_cell_indices):
//     cv::inpaint(filtered, inpaint_mask, inpainted, 3, cv::INPAINT_TELEA); return true;
// OpenCV does not exist in this image, so a closed form stands in for Telea's inpainting: a masked-out pixel takes the
// filtered index of the nearest kept pixel to its left in the row (0 if none).  The test then checks that
// getDepthMapFromDSI's 5th argument is convertDepthIndicesToValues of whatever this hook returns
// (mapper_emvs_stereo.cpp:430-436).
inline bool inpaint_depth_cell_indices(const Mat& filtered, const Mat& inpaint_mask, Mat& inpainted)
{
    inpainted.create(filtered.rows, filtered.cols, CV_8U);
    for (int y = 0; y < filtered.rows; ++y) {
        unsigned char last = 0;
        for (int x = 0; x < filtered.cols; ++x) {
            if (inpaint_mask.at<unsigned char>(y, x) == 0) last = filtered.at<unsigned char>(y, x);
            inpainted.at<unsigned char>(y, x) = last;
        }
    }
    return true;
}
}  // namespace cv
typedef unsigned char uchar;

namespace image_geometry {
class PinholeCameraModel {
public:
    // sensor_msgs::CameraInfo, the one field MapperEMVS reads (mapper_emvs_stereo.cpp:62)
    struct CameraInfo {
        std::string distortion_model;
    };
    PinholeCameraModel(int w, int h, double fx, double fy, double cx, double cy, const char* model = "plumb_bob")
        : w_(w), h_(h), fx_(fx), fy_(fy), cx_(cx), cy_(cy)
    {
        info_.distortion_model = model;
    }
    const CameraInfo& cameraInfo() const { return info_; }
    cv::Size fullResolution() const
    {
        cv::Size s;
        s.width = w_;
        s.height = h_;
        return s;
    }
    double fx() const { return fx_; }
    double fy() const { return fy_; }
    double cx() const { return cx_; }
    double cy() const { return cy_; }
    // an undistorted sensor: rectification is the identity (the LUT is an INPUT at the engine's boundary)
    cv::Point2d rectifyPoint(const cv::Point2d& uv_raw) const { return uv_raw; }

private:
    int w_, h_;
    double fx_, fy_, cx_, cy_;
    CameraInfo info_;
};
// a second camera type whose owner supplies the fisheye rectification (see dsi::fisheye_rectify_point below)
class FisheyeCameraModel : public PinholeCameraModel {
public:
    using PinholeCameraModel::PinholeCameraModel;
};
// what a maintainer adds for a fisheye camera, in the camera type's namespace (found by argument-dependent
// lookup from dsi::camera_of): the reference's fisheye_rectifyPoint (mapper_emvs_stereo.cpp:243-254); here a
// closed form stands in for cv::fisheye::undistortPoints
inline void fisheye_rectify_point(const FisheyeCameraModel&, double x, double y, double* u, double* v)
{
    *u = 0.5 * x + 1.0;
    *v = 0.25 * y - 2.0;
}
}  // namespace image_geometry

// ---------------------------------------------------------------- the reference's call sequence
namespace {

// process1.cpp:54-191 for two cameras, spelled with the reference's types and calls
void process_1_like_the_reference(const LinearTrajectory& trajectory0, const LinearTrajectory& trajectory1,
                                  const std::vector<dvs_msgs::Event>& events0,
                                  const std::vector<dvs_msgs::Event>& events1, EMVS::MapperEMVS& mapper_fused,
                                  EMVS::MapperEMVS& mapper0, EMVS::MapperEMVS& mapper1, double ts, int fusion_method,
                                  double rv_pos)
{
    geometry_utils::Transformation T_rv_w;
    const double t_mid = ts;
    geometry_utils::Transformation T_w_rv, T_w_l, T_w_r;
    trajectory0.getPoseAt(ros::Time(t_mid), T_w_l);
    trajectory1.getPoseAt(ros::Time(t_mid), T_w_r);
    const geometry_utils::Transformation baselineTransform(kindr::minimal::RotationQuaternion(1, 0, 0, 0),
                                                           kindr::minimal::Position(rv_pos, 0, 0));
    T_w_rv = T_w_l * baselineTransform;
    T_rv_w = T_w_rv.inverse();
    mapper0.evaluateDSI(events0, trajectory0, T_rv_w);
    (void)mapper0.dsi_.computeMeanSquare();
    mapper1.evaluateDSI(events1, trajectory1, T_rv_w);
    (void)mapper1.dsi_.computeMeanSquare();
    mapper_fused.dsi_.resetGrid();
    mapper_fused.dsi_.addTwoGrids(mapper0.dsi_);
    switch (fusion_method) {
    case 1: mapper_fused.dsi_.minTwoGrids(mapper1.dsi_); break;
    case 2: mapper_fused.dsi_.harmonicMeanTwoGrids(mapper1.dsi_); break;
    case 3: mapper_fused.dsi_.geometricMeanTwoGrids(mapper1.dsi_); break;
    case 4: mapper_fused.dsi_.arithmeticMeanTwoGrids(mapper1.dsi_); break;
    case 5: mapper_fused.dsi_.rmsTwoGrids(mapper1.dsi_); break;
    case 6: mapper_fused.dsi_.maxTwoGrids(mapper1.dsi_); break;
    default: throw dsi::Error(DSI_ERR_BAD_OP, "Improper fusion method selected");
    }
}

// utils.cpp:107-117 saveDepthMaps writes files; here it keeps what it was handed, so that the test can compare
struct Saved {
    std::string suffix;
    std::vector<float> depth, conf;
    std::vector<uchar> mask;
};
std::vector<Saved> g_saved;
void saveDepthMaps(const cv::Mat& depth_map, const cv::Mat& confidence_map, const cv::Mat& semidense_mask, const float, const float,
                   const std::string& suffix, const std::string&)
{
    Saved s;
    s.suffix = suffix;
    const size_t n = (size_t)depth_map.rows * depth_map.cols;
    s.depth.assign(depth_map.ptr<float>(0), depth_map.ptr<float>(0) + n);
    s.conf.assign(confidence_map.ptr<float>(0), confidence_map.ptr<float>(0) + n);
    s.mask.assign(semidense_mask.ptr<uchar>(0), semidense_mask.ptr<uchar>(0) + n);
    g_saved.push_back(s);
}

// process1.cpp:203-222, spelled as there (events2 of the stereo rig is empty; mapper2 exists but is not voted)
void process_1_outputs_like_the_reference(EMVS::MapperEMVS& mapper_fused, EMVS::MapperEMVS& mapper0, EMVS::MapperEMVS& mapper1,
                                          EMVS::MapperEMVS& mapper2, const std::vector<dvs_msgs::Event>& events2,
                                          const EMVS::OptionsDepthMap& opts_depth_map, const EMVS::ShapeDSI& dsi_shape,
                                          int fusion_method)
{
    std::stringstream ss;
    ss << "out/";
  // 3. Extract semi-dense depth map from DSI
  cv::Mat depth_map, confidence_map, semidense_mask;

  if (opts_depth_map.save_mono){
      // One DSI (voted by left-camera events)
      mapper0.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map);
      saveDepthMaps(depth_map, confidence_map, semidense_mask, dsi_shape.min_depth_, dsi_shape.max_depth_, std::string("0"), ss.str());
      // Another DSI (voted by right-camera events)
      mapper1.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map);
      saveDepthMaps(depth_map, confidence_map, semidense_mask, dsi_shape.min_depth_, dsi_shape.max_depth_, std::string("1"), ss.str());

      if(events2.size()>0){
          // Another DSI (voted by 3rd camera events)
          mapper2.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map);
          saveDepthMaps(depth_map, confidence_map, semidense_mask, dsi_shape.min_depth_, dsi_shape.max_depth_, std::string("2"), ss.str());
        }
    }

  // Fused DSIs
  mapper_fused.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map);
  saveDepthMaps(depth_map, confidence_map, semidense_mask, dsi_shape.min_depth_, dsi_shape.max_depth_, std::string("fused_" + std::to_string(fusion_method)), ss.str());
}

// process2.cpp:83, :253-263, :299, spelled as there (the mappers hold whatever DSIs the caller left in them: the
// extraction reads each mapper's own dsi_)
void process_2_outputs_like_the_reference(EMVS::MapperEMVS& mapper_fused, EMVS::MapperEMVS& mapper_fused_left,
                                          EMVS::MapperEMVS& mapper_fused_right, EMVS::MapperEMVS& mapper_fused_camera_time,
                                          const EMVS::OptionsDepthMap& opts_depth_map, const EMVS::ShapeDSI& dsi_shape,
                                          int temporal_fusion)
{
    std::stringstream ss;
  cv::Mat depth_map, confidence_map, semidense_mask;
  if (!opts_depth_map.full_sequence) {
      // Fused DSIs (using harmonic mean).
      mapper_fused_left.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map);
      saveDepthMaps(depth_map, confidence_map, semidense_mask, dsi_shape.min_depth_, dsi_shape.max_depth_, std::string("left_temporal_" + std::to_string(temporal_fusion)), ss.str());
      mapper_fused_right.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map);
      saveDepthMaps(depth_map, confidence_map, semidense_mask, dsi_shape.min_depth_, dsi_shape.max_depth_, std::string("right_temporal_" + std::to_string(temporal_fusion)), ss.str());
    }
  mapper_fused.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map);
  saveDepthMaps(depth_map, confidence_map, semidense_mask, dsi_shape.min_depth_, dsi_shape.max_depth_, std::string("stereo_temporal_" + std::to_string(temporal_fusion)), ss.str());

  mapper_fused_camera_time.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map);
  saveDepthMaps(depth_map, confidence_map, semidense_mask, dsi_shape.min_depth_, dsi_shape.max_depth_, std::string("stereo_temporal_camera_time" + std::to_string(temporal_fusion)), ss.str());
}

bool same(const std::vector<float>& a, const float* b) { return std::memcmp(a.data(), b, a.size() * sizeof(float)) == 0; }
bool same(const std::vector<uchar>& a, const uchar* b) { return std::memcmp(a.data(), b, a.size()) == 0; }

// The image-typed members with cv::Mat: every call site of the reference, outputs memcmp-equal to the dsi::Image path.
// mapper_fused / mapper0 / mapper1 hold the DSIs process_1_like_the_reference left.
int check_image_typed_members(EMVS::MapperEMVS& mapper_fused, EMVS::MapperEMVS& mapper0, EMVS::MapperEMVS& mapper1,
                              EMVS::MapperEMVS& mapper2, const EMVS::ShapeDSI& dsi_shape)
{
    EMVS::OptionsDepthMap opts_depth_map;  // main.cpp:163-171
    opts_depth_map.max_confidence = 10;
    opts_depth_map.adaptive_threshold_kernel_size_ = 5;
    opts_depth_map.adaptive_threshold_c_ = 4;
    opts_depth_map.median_filter_size_ = 5;
    opts_depth_map.full_sequence = false;
    opts_depth_map.save_mono = true;
    const std::vector<dvs_msgs::Event> events2;
    EMVS::MapperEMVS* all[3] = {&mapper0, &mapper1, &mapper_fused};
    // the dsi::Image path (what tests/cpp/test_process1.cpp holds against the oracle's filters)
    dsi::Image<float> want_depth[3], want_conf[3];
    dsi::Image<uint8_t> want_mask[3];
    for (int i = 0; i < 3; ++i) all[i]->getDepthMapFromDSI(want_depth[i], want_conf[i], want_mask[i], opts_depth_map);
    size_t kept = 0;
    for (uint8_t m : want_mask[2].data) kept += m;
    if (kept == 0) return 70;  // the comparison must be about something

    // ---- process1.cpp:203-222
    g_saved.clear();
    process_1_outputs_like_the_reference(mapper_fused, mapper0, mapper1, mapper2, events2, opts_depth_map, dsi_shape, 2);
    if (g_saved.size() != 3 || g_saved[0].suffix != "0" || g_saved[1].suffix != "1" || g_saved[2].suffix != "fused_2") return 71;
    for (int i = 0; i < 3; ++i)
        if (!same(g_saved[i].depth, want_depth[i].data.data()) || !same(g_saved[i].conf, want_conf[i].data.data()) ||
            !same(g_saved[i].mask, want_mask[i].data.data()))
            return 72 + i;
    // ---- process2.cpp:253-263, :299 (four mappers, each read through its own dsi_)
    g_saved.clear();
    process_2_outputs_like_the_reference(mapper_fused, mapper0, mapper1, mapper_fused, opts_depth_map, dsi_shape, 2);
    if (g_saved.size() != 4 || g_saved[0].suffix != "left_temporal_2" || g_saved[3].suffix != "stereo_temporal_camera_time2") return 75;
    const int who[4] = {0, 1, 2, 2};
    for (int i = 0; i < 4; ++i)
        if (!same(g_saved[i].depth, want_depth[who[i]].data.data()) || !same(g_saved[i].mask, want_mask[who[i]].data.data())) return 76;
    // ---- main.cpp:388-389, :406, :417: the five-argument form
    {
        cv::Mat depth_map, confidence_map, semidense_mask;
        cv::Mat depth_map_dense;
        mapper_fused.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map, depth_map_dense);
        if (depth_map.type() != CV_32F || confidence_map.type() != CV_32F || semidense_mask.type() != CV_8U ||
            depth_map.rows != want_depth[2].rows || depth_map.cols != want_depth[2].cols)
            return 80;
        if (std::memcmp(depth_map.ptr<float>(0), want_depth[2].data.data(), want_depth[2].data.size() * 4) != 0 ||
            std::memcmp(confidence_map.ptr<float>(0), want_conf[2].data.data(), want_conf[2].data.size() * 4) != 0 ||
            std::memcmp(semidense_mask.ptr<uchar>(0), want_mask[2].data.data(), want_mask[2].data.size()) != 0)
            return 81;
        // depth_map_dense = convertDepthIndicesToValues(inpaint(filtered indices, 1 - mask)) (mapper_emvs_stereo.cpp:430-436)
        // with the hook above: on kept pixels the dense map IS the semi-dense one; elsewhere the nearest kept depth to the left
        if (depth_map_dense.empty() || depth_map_dense.type() != CV_32F || depth_map_dense.rows != depth_map.rows) return 82;
        const std::vector<float> planes = mapper_fused.depthPlanes();
        for (int y = 0; y < depth_map.rows; ++y) {
            float last = planes[0];
            for (int x = 0; x < depth_map.cols; ++x) {
                if (semidense_mask.at<uchar>(y, x)) last = depth_map.at<float>(y, x);
                if (depth_map_dense.at<float>(y, x) != last) {
                    std::printf("depth_map_dense(%d,%d) = %g, expected %g\n", y, x, depth_map_dense.at<float>(y, x), last);
                    return 83;
                }
            }
        }
        mapper0.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map, depth_map_dense);  // :406
        if (std::memcmp(depth_map.ptr<float>(0), want_depth[0].data.data(), want_depth[0].data.size() * 4) != 0) return 84;
        mapper1.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map, depth_map_dense);  // :417
        if (std::memcmp(depth_map.ptr<float>(0), want_depth[1].data.data(), want_depth[1].data.size() * 4) != 0) return 85;
        // without a hook for the image type (dsi::Image has none) the dense map is left EMPTY, never a guess
        dsi::Image<float> d, c, dense(3, 3);
        dsi::Image<uint8_t> m;
        mapper_fused.getDepthMapFromDSI(d, c, m, opts_depth_map, dense);
        if (d.data != want_depth[2].data || dense.rows != 0 || !dense.data.empty()) return 86;
    }
    // ---- int method = -1 (mapper_emvs_stereo.hpp:108-109): -1 and anything outside 0..4 is collapseMaxZSlice (.cpp:348-368);
    //      the focus collapses 0..4 are refused
    {
        cv::Mat depth_map, confidence_map, semidense_mask, depth_map_dense;
        mapper_fused.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map, -1);
        if (std::memcmp(depth_map.ptr<float>(0), want_depth[2].data.data(), want_depth[2].data.size() * 4) != 0) return 87;
        mapper_fused.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map, depth_map_dense, 7);
        if (std::memcmp(depth_map.ptr<float>(0), want_depth[2].data.data(), want_depth[2].data.size() * 4) != 0) return 88;
        for (int method = 0; method <= 4; ++method) {
            try {
                mapper_fused.getDepthMapFromDSI(depth_map, confidence_map, semidense_mask, opts_depth_map, method);
                return 89;
            } catch (const dsi::Error& e) {
                if (e.code != DSI_ERR_BAD_OP) return 90;
            }
        }
    }
    // ---- Grid3D::collapseMaxZSlice(cv::Mat*, cv::Mat*) (cartesian3dgrid.h:207; called at mapper_emvs_stereo.cpp:366)
    {
        cv::Mat confidence_map, depth_cell_indices;
        mapper_fused.dsi_.collapseMaxZSlice(&confidence_map, &depth_cell_indices);
        dsi::Image<float> c;
        dsi::Image<uint8_t> i;
        mapper_fused.dsi_.collapseMaxZSlice(&c, &i);
        if (confidence_map.type() != CV_32F || depth_cell_indices.type() != CV_8U || confidence_map.rows != c.rows ||
            confidence_map.cols != c.cols)
            return 91;
        if (std::memcmp(confidence_map.ptr<float>(0), c.data.data(), c.data.size() * 4) != 0 ||
            std::memcmp(depth_cell_indices.ptr<uchar>(0), i.data.data(), i.data.size()) != 0)
            return 92;
        // MapperEMVS::convertDepthIndicesToValues (mapper_emvs_stereo.cpp:302-313) on cv::Mat
        cv::Mat depth;
        mapper_fused.convertDepthIndicesToValues(depth_cell_indices, depth);
        dsi::Image<float> rd, rc;
        dsi::Image<uint8_t> ri;
        mapper_fused.getDepthMapFromDSI(rd, rc, ri);
        if (std::memcmp(depth.ptr<float>(0), rd.data.data(), rd.data.size() * 4) != 0) return 93;
    }
    std::printf("cv::Mat-typed getDepthMapFromDSI (4- and 5-argument, method), collapseMaxZSlice, convertDepthIndicesToValues "
                "== dsi::Image path, %zu semi-dense pixels: OK\n", kept);
    return 0;
}

}  // namespace

// dsi::camera_of and the distortion model (mapper_emvs_stereo.cpp:62, :256-299): host-only, no GPU needed
static int check_camera_of()
{
    const int W = 12, H = 7;
    dsi::PinholeCameraModel out;
    dsi::camera_of(image_geometry::PinholeCameraModel(W, H, 70.0, 71.0, 6.0, 3.0), &out);   // plumb_bob: rectifyPoint
    if (out.width != W || out.height != H || out.fy != 71.f || out.rectified_points.size() != (size_t)2 * W * H) return 20;
    if (out.rectified_points[2 * (3 * W + 5)] != 5.f || out.rectified_points[2 * (3 * W + 5) + 1] != 3.f) return 21;
    try {   // a fisheye camera nobody supplied the rectification for: refused, not silently plumb_bob
        dsi::camera_of(image_geometry::PinholeCameraModel(W, H, 70.0, 70.0, 6.0, 3.0, "fisheye"), &out);
        return 22;
    } catch (const dsi::Error& e) {
        if (e.code != DSI_ERR_INVALID || std::string(e.what()).find("fisheye") == std::string::npos) return 23;
    }
    try {   // "Distortion model not set properly!" (:289-293)
        dsi::camera_of(image_geometry::PinholeCameraModel(W, H, 70.0, 70.0, 6.0, 3.0, "equidistant"), &out);
        return 24;
    } catch (const dsi::Error& e) {
        if (e.code != DSI_ERR_INVALID) return 25;
    }
    dsi::camera_of(image_geometry::FisheyeCameraModel(W, H, 70.0, 70.0, 6.0, 3.0, "fisheye"), &out);   // the owner's overload
    if (out.rectified_points[2 * (4 * W + 6)] != 4.f || out.rectified_points[2 * (4 * W + 6) + 1] != -1.f) return 26;
    std::printf("camera_of: plumb_bob / fisheye / unknown distortion models handled: OK\n");
    return 0;
}

// main.cpp:199-216: the recorded poses become the left camera's (hand-eye calibration) and the right camera's (extrinsics)
// by TrajectoryBase::applyTransformationRight (trajectory.hpp:57-63), with the reference's pose type.  Host-only.
int check_trajectory_transformations()
{
    std::map<ros::Time, geometry_utils::Transformation> interval_poses;
    const double c = std::cos(0.35), s_ = std::sin(0.35);  // every control pose: rotation by 0.7 rad about y, x = 0.4 t
    for (int k = 0; k <= 6; ++k)
        interval_poses[ros::Time(0.5 * k)] =
            geometry_utils::Transformation(kindr::minimal::RotationQuaternion(c, 0, s_, 0), kindr::minimal::Position(0.2 * k, 0, 1));
    const geometry_utils::Transformation T_hand_eye(kindr::minimal::RotationQuaternion(std::cos(0.1), std::sin(0.1), 0, 0),
                                                    kindr::minimal::Position(0.01, -0.02, 0.03));
    const geometry_utils::Transformation T_extr(kindr::minimal::RotationQuaternion(1, 0, 0, 0), kindr::minimal::Position(-0.6, 0, 0));
    LinearTrajectory trajectory0 = LinearTrajectory(interval_poses);   // main.cpp:199
    trajectory0.applyTransformationRight(T_hand_eye);                  // :202
    LinearTrajectory trajectory1 = LinearTrajectory(interval_poses);   // :204
    trajectory1.applyTransformationRight(T_hand_eye);                  // :205
    trajectory1.applyTransformationRight(T_extr.inverse());            // :207
    LinearTrajectory left = LinearTrajectory(interval_poses);
    left.applyTransformationLeft(T_extr);
    // at a control time the interpolated pose IS the control pose: compare with the products formed by the pose type itself
    const double t = 1.0;
    const geometry_utils::Transformation P = interval_poses[ros::Time(t)];
    const geometry_utils::Transformation want0 = P * T_hand_eye, want1 = P * T_hand_eye * T_extr.inverse(), wantL = T_extr * P;
    const geometry_utils::Transformation* wants[3] = {&want0, &want1, &wantL};
    const LinearTrajectory* trs[3] = {&trajectory0, &trajectory1, &left};
    for (int i = 0; i < 3; ++i) {
        geometry_utils::Transformation got;
        if (!trs[i]->getPoseAt(ros::Time(t), got)) return 40 + i;
        double a[7], b[7];
        dsi::to_pose7(got, a);
        dsi::to_pose7(*wants[i], b);
        for (int k = 0; k < 7; ++k)
            if (std::fabs(a[k] - b[k]) > 1e-12) {
                std::printf("trajectory %d component %d: %.15g vs %.15g\n", i, k, a[k], b[k]);
                return 50 + i;
            }
    }
    // the right camera sits 0.6 m along the left camera's +x (T_extr^-1 applied on the right = in the camera frame)
    dsi::Transformation l, r;
    if (!trajectory0.getPoseAt(1.25, l) || !trajectory1.getPoseAt(1.25, r)) return 60;
    double base[3] = {r.t[0] - l.t[0], r.t[1] - l.t[1], r.t[2] - l.t[2]}, x_axis[3];
    const double ex[3] = {1, 0, 0};
    dsi::quat_rotate(l.q, ex, x_axis);
    for (int k = 0; k < 3; ++k)
        if (std::fabs(base[k] - 0.6 * x_axis[k]) > 1e-12) return 61;
    {
        // main.cpp:159-171, spelled as there
        const int FLAGS_dimX = 0, FLAGS_dimY = 0, FLAGS_dimZ = 100;
        const double FLAGS_min_depth = 0.3, FLAGS_max_depth = 5.0, FLAGS_fov_deg = 0.0;
        EMVS::ShapeDSI dsi_shape(FLAGS_dimX, FLAGS_dimY, FLAGS_dimZ, FLAGS_min_depth, FLAGS_max_depth, FLAGS_fov_deg);
        EMVS::OptionsDepthMap opts_depth_map;
        opts_depth_map.max_confidence = 0;
        opts_depth_map.adaptive_threshold_kernel_size_ = 5;
        opts_depth_map.adaptive_threshold_c_ = 7;
        opts_depth_map.median_filter_size_ = 5;
        opts_depth_map.full_sequence = true;
        opts_depth_map.save_conf_stats = false;
        opts_depth_map.save_mono = false;
        opts_depth_map.rv_pos = 0.0;
        opts_depth_map.save_dsi = false;
        if (dsi_shape.dimZ_ != 100 || opts_depth_map.adaptive_threshold_c_ != 7) return 63;
    }
    {
        dsi::Transformation T0, T1;
        double t0 = -1, t1 = -1;
        left.getFirstControlPose(&T0, &t0);
        left.getLastControlPose(&T1, &t1);
        if (t0 != 0.0 || t1 != 3.0 || left.getNumControlPoses() != 7 || std::fabs(T1.t[0] - T0.t[0] - 1.2) > 1e-12) return 62;
    }
    std::printf("applyTransformationRight / Left with the reference's pose type: OK\n");
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && std::string(argv[1]) == "--camera-of") return check_camera_of();
    if (argc > 1 && std::string(argv[1]) == "--trajectory") return check_trajectory_transformations();
    try {
        const int W = 80, H = 60;
        const image_geometry::PinholeCameraModel cam0(W, H, 70.0, 70.0, 40.0, 30.0), cam1 = cam0;
        const EMVS::ShapeDSI dsi_shape(0, 0, 24, 1.0f, 6.0f, 0.f);
        // control poses (x = 0.4 t, second camera 0.2 m to the right) and events on both type systems
        std::map<ros::Time, geometry_utils::Transformation> poses0, poses1;
        LinearTrajectory::PoseMap plain0, plain1;
        for (int k = 0; k <= 12; ++k) {
            const double t = 0.1 * k - 0.1;
            poses0[ros::Time(t)] = geometry_utils::Transformation(kindr::minimal::RotationQuaternion(1, 0, 0, 0),
                                                                  kindr::minimal::Position(0.4 * t, 0, 0));
            poses1[ros::Time(t)] = geometry_utils::Transformation(kindr::minimal::RotationQuaternion(1, 0, 0, 0),
                                                                  kindr::minimal::Position(0.4 * t + 0.2, 0, 0));
            dsi::Transformation A, B;
            A.t[0] = 0.4 * t;
            B.t[0] = 0.4 * t + 0.2;
            plain0[t] = A;
            plain1[t] = B;
        }
        std::vector<dvs_msgs::Event> events0, events1;
        std::vector<dsi::Event> pev0, pev1;
        unsigned long long s = 12345;
        auto rnd = [&]() {
            s = s * 6364136223846793005ULL + 1442695040888963407ULL;
            return (double)(s >> 11) / 9007199254740992.0;
        };
        const int n = 6000;
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < n; ++i) {
                dvs_msgs::Event e;
                e.x = (uint16_t)(rnd() * W);
                e.y = (uint16_t)(rnd() * H);
                e.ts = ros::Time(1.0 * i / n);
                dsi::Event p;
                p.x = e.x;
                p.y = e.y;
                p.ts = e.ts.toSec();
                (c == 0 ? events0 : events1).push_back(e);
                (c == 0 ? pev0 : pev1).push_back(p);
            }
        // ---- the reference's spelling: constructors without a context, ROS-typed arguments
        const LinearTrajectory trajectory0(poses0), trajectory1(poses1);
        EMVS::MapperEMVS mapper_fused(cam0, dsi_shape), mapper0(cam0, dsi_shape), mapper1(cam1, dsi_shape);
        process_1_like_the_reference(trajectory0, trajectory1, events0, events1, mapper_fused, mapper0, mapper1, 0.5, 2,
                                     0.05);
        const std::vector<float> got = mapper_fused.dsi_.download();
        Grid3D spare(80, 60, 24);  // Grid3D(dimX, dimY, dimZ), cartesian3dgrid.h:26
        int gx, gy, gz;
        spare.getDimensions(&gx, &gy, &gz);
        if (gx != 80 || gy != 60 || gz != 24) return 10;
        // ---- the same through the adapter's plain types (what tests/cpp/test_process1.cpp checks
        //      against the oracle)
        dsi::PinholeCameraModel pcam;
        pcam.width = W; pcam.height = H; pcam.fx = 70.f; pcam.fy = 70.f; pcam.cx = 40.f; pcam.cy = 30.f;
        dsi::Context& ctx = dsi::default_context();
        EMVS::MapperEMVS pf(ctx, pcam, dsi_shape), p0(ctx, pcam, dsi_shape), p1(ctx, pcam, dsi_shape), p2(ctx, pcam, dsi_shape);
        const LinearTrajectory pt0(plain0), pt1(plain1);
        const std::vector<dsi::Event> none;
        process_1(pt0, pt1, pt1, pev0, pev1, none, pf, p0, p1, p2, 0.5, 2, 0.05);
        const std::vector<float> want = pf.dsi_.download();
        if (got.size() != want.size()) return 11;
        double sum = 0;
        for (size_t i = 0; i < got.size(); ++i) {
            if (got[i] != want[i]) {
                std::printf("voxel %zu: %g vs %g\n", i, got[i], want[i]);
                return 12;
            }
            sum += got[i];
        }
        if (!(sum > 100.0)) return 13;
        std::printf("reference-typed call sequence == plain-typed call sequence, fused DSI sum %.6g: OK\n", sum);
        EMVS::MapperEMVS mapper2(cam0, dsi_shape);
        return check_image_typed_members(mapper_fused, mapper0, mapper1, mapper2, dsi_shape);
    } catch (const dsi::Error& e) {
        std::printf("dsi::Error %d: %s\n", e.code, e.what());
        return e.code == DSI_ERR_NO_DEVICE ? 3 : 2;
    }
}
