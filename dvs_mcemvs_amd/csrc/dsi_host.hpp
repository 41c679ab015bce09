// dsi_host.hpp -- host-side (CPU, double precision) part of MapperEMVS::evaluateDSI
// that stays in front of the device boundary: packetisation and the per-packet
// pose lookup.
//
//   packetisation            mapper_emvs_stereo.cpp:67-99, :131
//   LinearTrajectory         trajectory.hpp:81-128
//   T_ev_rv -> R, t (float)  mapper_emvs_stereo.cpp:101-105
//
// The SE(3) arithmetic of the reference lives in minkindr (un-vendored,
// dependencies.yaml:18-21, "version: master"); its published semantics are
// followed: a transformation is (unit quaternion q, translation t),
// T1*T2 = (q1 q2, t1 + q1.rotate(t2)), inverse = (q^-1, -q^-1.rotate(t)),
// log/exp treat the translation linearly and the rotation through the SO(3)
// exponential (QuatTransformation::log / ::exp).
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace dsi {
namespace host {

struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    Vec3 operator+(const Vec3& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-() const { return {-x, -y, -z}; }
    Vec3 operator*(double s) const { return {x * s, y * s, z * s}; }
    Vec3 cross(const Vec3& o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    double norm() const { return std::sqrt(x * x + y * y + z * z); }
};

struct Quat {
    double w = 1, x = 0, y = 0, z = 0;
    Vec3 vec() const { return {x, y, z}; }
    Quat conjugate() const { return {w, -x, -y, -z}; }
    // Hamilton product (Eigen quat_product)
    Quat operator*(const Quat& b) const
    {
        return {w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y,
                w * b.y + y * b.w + z * b.x - x * b.z, w * b.z + z * b.w + x * b.y - y * b.x};
    }
    // Eigen QuaternionBase::_transformVector
    Vec3 rotate(const Vec3& v) const
    {
        Vec3 uv = vec().cross(v);
        uv = uv + uv;
        return v + uv * w + vec().cross(uv);
    }
    // minkindr RotationQuaternion::log -> rotation vector
    Vec3 log() const
    {
        const double na = vec().norm();
        const double eta = w;
        double scale;
        if (std::fabs(eta) < na) {
            scale = (eta >= 0) ? std::acos(eta) / na : -std::acos(-eta) / na;
        } else {
            const double s = (std::fabs(na) < kEps4) ? 1.0 + na * na / 6.0 : std::asin(na) / na;
            scale = (eta > 0) ? s : -s;
        }
        return vec() * (2.0 * scale);
    }
    // minkindr RotationQuaternion::exp (Grassia 1998)
    static Quat exp(const Vec3& dx)
    {
        const double theta = dx.norm();
        const double na = (theta < kEps4) ? 0.5 + theta * theta / 48.0 : std::sin(theta * 0.5) / theta;
        return {std::cos(theta * 0.5), dx.x * na, dx.y * na, dx.z * na};
    }
    static constexpr double kEps4 = 1.220703125e-4;  // ~ eps^(1/4)
};

struct Pose {  // T_A_B
    Quat q;
    Vec3 t;
    static Pose from7(const double* p) { return {{p[3], p[4], p[5], p[6]}, {p[0], p[1], p[2]}}; }
    void to7(double* p) const
    {
        p[0] = t.x; p[1] = t.y; p[2] = t.z;
        p[3] = q.w; p[4] = q.x; p[5] = q.y; p[6] = q.z;
    }
    Pose operator*(const Pose& b) const { return {q * b.q, t + q.rotate(b.t)}; }
    Pose inverse() const
    {
        const Quat qi = q.conjugate();
        return {qi, -qi.rotate(t)};
    }
};

// LinearTrajectory::getPoseAt (trajectory.hpp:92-126).  times ascending.
inline bool pose_at(const double* times, const double* poses, size_t n, double t, Pose* out)
{
    // std::map::upper_bound(t): first control pose strictly later than t
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (t < times[mid])
            hi = mid;
        else
            lo = mid + 1;
    }
    if (lo == 0 || lo == n) return false;  // no extrapolation (:99-112)
    const Pose T0 = Pose::from7(poses + 7 * (lo - 1)), T1 = Pose::from7(poses + 7 * lo);
    const Pose rel = T0.inverse() * T1;  // :123
    const double delta = (t - times[lo - 1]) / (times[lo] - times[lo - 1]);  // :124
    const Pose inc{Quat::exp(rel.q.log() * delta), rel.t * delta};  // exp(delta * log(rel)), :125
    *out = T0 * inc;
    return true;
}

// mapper_emvs_stereo.cpp:101-105: (T_rv_w * T_w_ev)^-1 -> R (row-major), t as float
inline void event_pose_Rt(const Pose& T_rv_w, const Pose& T_w_ev, float* Rt)
{
    const Pose T = (T_rv_w * T_w_ev).inverse();
    const Quat& q = T.q;
    // Eigen QuaternionBase::toRotationMatrix
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz),
                         tyz - twx,       txz - twy, tyz + twx, 1 - (txx + tyy)};
    for (int i = 0; i < 9; ++i) Rt[i] = (float)R[i];
    Rt[9] = (float)T.t.x;
    Rt[10] = (float)T.t.y;
    Rt[11] = (float)T.t.z;
}

// The while-loop of mapper_emvs_stereo.cpp:88-99: returns false when the reference
// returns false (fewer events than one packet, :71-75).
// ts_of(i): timestamp of event i (a plain array, or a field of an array of structs read in place)
template <typename TsOf>
inline bool packetize_with(TsOf ts_of, size_t n_events, const double* times, const double* poses,
                           size_t n_poses, const Pose& T_rv_w, std::vector<uint32_t>* first,
                           std::vector<float>* Rt)
{
    constexpr size_t kPacket = 1024;
    first->clear();
    Rt->clear();
    if (n_events < kPacket) return false;
    size_t cur = 0;
    while (cur + kPacket < n_events) {  // strict '<': a final exactly-full packet is dropped
        Pose T_w_ev;
        if (!pose_at(times, poses, n_poses, ts_of(cur + kPacket / 2), &T_w_ev)) {
            ++cur;  // :95-99 slide by one event and retry
            continue;
        }
        first->push_back((uint32_t)cur);
        Rt->resize(Rt->size() + 12);
        event_pose_Rt(T_rv_w, T_w_ev, Rt->data() + Rt->size() - 12);
        cur += kPacket;
    }
    return true;
}

inline bool packetize(const double* ts, size_t n_events, const double* times, const double* poses,
                      size_t n_poses, const Pose& T_rv_w, std::vector<uint32_t>* first,
                      std::vector<float>* Rt)
{
    return packetize_with([ts](size_t i) { return ts[i]; }, n_events, times, poses, n_poses, T_rv_w, first, Rt);
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU partition arithmetic.  Pure functions: the RCCL path (dsi_engine.cpp), the
// host-staged 2-rank GPU tests and the CPU tests all take their offsets from here, so that on an
// 8-GPU node only the nccl* calls themselves are new code.

// Temporal fusion as a reduce-scatter by planes: rank r of n reduces planes [r q, (r+1) q), q = nz / n
// (ncclReduceScatter needs equal counts); the nz mod n planes left over are all-reduced, so EVERY rank
// owns them as a second range.  A rank's arg-max runs over its two ranges; the keys' MAX over the ranks
// is the arg-max over all planes.
struct ScatterPlan {
    int q = 0;           // planes per rank in the reduce-scatter part
    int own_begin = 0;   // first plane of this rank's reduce-scatter range (own_count == q)
    int own_count = 0;
    int tail_begin = 0;  // planes [tail_begin, tail_begin + tail_count) are all-reduced (owned by every rank)
    int tail_count = 0;
};

inline bool scatter_plan(int nz, int nranks, int rank, ScatterPlan* out)
{
    if (nz < 1 || nranks < 1 || rank < 0 || rank >= nranks) return false;
    ScatterPlan p;
    p.q = nz / nranks;
    p.own_begin = rank * p.q;
    p.own_count = p.q;
    p.tail_begin = p.q * nranks;
    p.tail_count = nz - p.tail_begin;
    *out = p;
    return true;
}

// Plane sharding of one big DSI (configs[4]): contiguous, balanced ranges; the first nz mod n ranks own
// one plane more.  (Ranges may be empty when n > nz.)
inline bool plane_range(int nz, int nranks, int rank, int* begin, int* count)
{
    if (nz < 1 || nranks < 1 || rank < 0 || rank >= nranks) return false;
    const int base = nz / nranks, extra = nz % nranks;
    *begin = rank * base + (rank < extra ? rank : extra);
    *count = base + (rank < extra ? 1 : 0);
    return true;
}

// One arg-max key per pixel: (confidence bits << 8) | (255 - global plane index).  MAX over shards =
// Grid3D::collapseMaxZSlice over all planes (cartesian3dgrid.cpp:115-137, std::max_element: the larger
// confidence wins and, on equal confidence, the smaller plane index).  DSI values are >= 0 and never
// -0.0, so their bit patterns order like the floats.  The device kernels (k_pack_argmax,
// k_unpack_argmax, the fused vote) build the same word.
inline uint64_t argmax_key(float conf, int global_plane)
{
    uint32_t bits;
    static_assert(sizeof bits == sizeof conf, "fp32");
    __builtin_memcpy(&bits, &conf, sizeof bits);
    return ((uint64_t)bits << 8) | (uint64_t)(255 - global_plane);
}

inline void argmax_unkey(uint64_t key, float* conf, int* global_plane)
{
    const uint32_t bits = (uint32_t)(key >> 8);
    __builtin_memcpy(conf, &bits, sizeof bits);
    *global_plane = 255 - (int)(key & 255u);
}

// The 2-ary camera-fusion ops on the host, for the handful of voxels the exact tie resolver re-sums
// (dsi_mapper_resolve_near_ties): the same operations in the same order as the device's fuse_op (dsi_kernels.hip),
// i.e. as Grid3D::minTwoGrids ... maxTwoGrids (cartesian3dgrid.h:111-190) applied to a grid that was initialised by
// resetGrid(); addTwoGrids(a) (process1.cpp:126-127: 0 + a).  Compiled without FMA contraction.
inline float fuse2(int op, float a0, float g)
{
    const float a = 0.f + a0;
    switch (op) {
    case 1: return (g < a) ? g : a;  // std::min, :115
    case 2: {                        // :119-127
        const float prod = a * g, sum = a + g;
        return 2.f * prod / (sum + 0.1f);
    }
    case 3: return std::sqrt(a * g);  // :154
    case 4: return (float)(0.5 * (double)(a + g));  // :162
    case 5: {                                       // :145-146
        const float ms = (float)(0.5 * ((double)a * (double)a + (double)g * (double)g));
        return std::sqrt(ms);
    }
    default: return (a < g) ? g : a;  // std::max, :188
    }
}

// Grid3D::addTwoGrids / addInverseOfTwoGrids (cartesian3dgrid.h:64-78) and computeAMfromSum / computeHMfromSumOfInv
// (:80-93) for one voxel: mode 0 = DSI_ACC_SUM, 1 = DSI_ACC_INV_SUM -- as the device's k_elementwise<EW_ADD / EW_ADD_INV /
// EW_FIN_AM / EW_FIN_HM> compute them
inline float accumulate1(int mode, float acc, float g) { return mode == 1 ? acc + 1.0f / (0.01f + g) : acc + g; }
inline float finalize1(int mode, float acc, int n) { return mode == 1 ? (float)n / acc : acc / (float)n; }

}  // namespace host
}  // namespace dsi
