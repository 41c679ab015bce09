/*
 * dsi_oracle.h -- CPU ORACLE for the DSI hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the arithmetic of tub-rip/dvs_mcemvs on the path
 *   MapperEMVS::evaluateDSI -> fillVoxelGrid -> Grid3D::accumulateGridValueAt,
 *   Grid3D voxel-wise fusion, Grid3D::collapseMaxZSlice.
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures,
 * and neither of its translation units on this path can be built in this image
 * without writing stand-ins for OpenCV / Eigen / minkindr / ROS headers (not
 * allowed), so this oracle has never been checked against an execution of the
 * reference.  It is pinned only by analytic known-answer tests that we
 * authored (tests/test_oracle_kat.py) and by bit-equality with a second,
 * independently written numpy restatement of the same reference lines
 * (tests/independent_numpy.py) -- neither is an execution of the reference.
 *
 * Third-party arithmetic that is NOT under the reference tree and is restated
 * here from its published algorithm (dependencies.yaml pins all of them only
 * as "version: master"):
 *   - Eigen 3.3.x (via eigen_catkin): fixed-size 3x3 lazy product (length-3
 *     dot evaluated as x0 + (x1 + x2), redux_novec_unroller), 3x3 inverse by
 *     cofactors (compute_inverse_size3_helper), 4x4 * vec4 packet product
 *     (sequential multiply-add, no FMA on SSE2), vec4 /= scalar (true divide).
 *   - minkindr (QuatTransformation::log/exp: translation linear, rotation by
 *     the SO(3) exponential) -- used by LinearTrajectory::getPoseAt.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  The product (dvs_mcemvs_amd/) never does.
 *
 * Build: gcc -O3 -fopenmp -ffp-contract=off (no -march=native, no
 * -ffast-math), mirroring mapper_emvs_stereo/CMakeLists.txt:14.
 */
#ifndef DSI_ORACLE_H
#define DSI_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_PACKET_SIZE 1024 /* mapper_emvs_stereo.hpp:152 */

/* depth_vector.hpp:76-163.  inverse==0: LinearDepthVector (CMake default),
 * inverse!=0: InverseDepthVector.  raw_depths[i] = cellIndexToDepth(i)
 * (mapper_emvs_stereo.cpp:213-214). */
void orc_depth_planes(float min_depth, float max_depth, int nz, int inverse,
                      float *raw_depths);

/* mapper_emvs_stereo.cpp:219-229: focal length of the virtual camera. */
float orc_virtual_focal(float cam_fx, float fov_deg, int dim_x);

/* Eigen 3x3 float inverse by cofactors (row-major in/out). */
void orc_inverse3x3(const float *m, float *out);

/* mapper_emvs_stereo.cpp:101-126, per packet.  Rt = R (row-major 9) then t (3),
 * already cast to float.  K = {fx,fy,cx,cy} of the sensor projection matrix
 * (mapper_emvs_stereo.cpp:46-48), Kv = {fx,fy,cx,cy} of the virtual camera
 * (geometry_utils.hpp:41-47).  Outputs the camera centre (3) and H_z0_px
 * (row-major 9). */
void orc_packet_geometry(const float *Rt, const float *K, const float *Kv,
                         float z0, float *center, float *H);

/* mapper_emvs_stereo.cpp:129-142, per event: p=(LUT[y*W+x],1,0); p=H4*p; p/=p[2].
 * lut is (u,v) pairs, index y*W+x; lut==NULL means the identity (u=x, v=y).
 * events are consumed packet by packet: event i uses H[(i/1024)].
 * xy_z0 receives 2 floats per event. */
void orc_warp_z0(const uint16_t *ex, const uint16_t *ey, size_t n_events,
                 const float *H /* 9 per packet */, const float *lut, int W,
                 float *xy_z0);

/* mapper_emvs_stereo.cpp:151-205 + cartesian3dgrid.h:253-273.
 * xy_z0: 2 floats per event, n_packets*1024 events; centers: 3 per packet;
 * Kv = {fx,fy,cx,cy} virtual camera; dsi is [nz][ny][nx], ACCUMULATED INTO
 * (caller resets, as mapper_emvs_stereo.cpp:145 does).
 * OpenMP over planes exactly like the reference (one thread owns a plane). */
void orc_fill_voxel_grid(const float *xy_z0, const float *centers,
                         size_t n_packets, const float *raw_depths, int nz,
                         const float *Kv, int nx, int ny, float *dsi);
/* rows [row_begin, row_begin + row_count) of every plane of the same DSI, bit-equal to the full one's */
void orc_fill_voxel_grid_rows(const float *xy_z0, const float *centers, size_t n_packets,
                              const float *raw_depths, int nz, const float *Kv, int nx, int ny,
                              int row_begin, int row_count, float *strip);

/* cartesian3dgrid.h:253-273 (single vote into one plane). */
void orc_vote(float x_f, float y_f, float *plane, int nx, int ny);

/* mapper_emvs_stereo.cpp:67-99 packetisation.  pose_ok[i] tells whether the
 * pose lookup at events[i].ts would succeed (trajectory.hpp:98-113).
 * Returns the number of packets; first_event[k] = index of the first event of
 * packet k, mid_event[k] = index whose timestamp is used.  Returns -1 when
 * n_events < 1024 (evaluateDSI returns false, :71-75).  Arrays must hold
 * n_events/1024 + 1 entries. */
long orc_packetize(size_t n_events, const uint8_t *pose_ok,
                   size_t *first_event, size_t *mid_event);

/* Camera fusion, cartesian3dgrid.h:111-192; op codes as process1.cpp:136-158:
 * 1 min, 2 HM, 3 GM, 4 AM, 5 RMS, 6 max.  a is updated in place. */
int orc_fuse2(float *a, const float *g, size_t n, int op);
/* cartesian3dgrid.h:130-139 harmonicMeanTwoGrids(grid2, n). */
void orc_fuse_hm_n(float *a, const float *g, size_t n, int n_maps);
/* cartesian3dgrid.h:64-78: mode 0 addTwoGrids, mode 1 addInverseOfTwoGrids.
 * Modes 2..5 are the n-ary accumulate forms of the camera-fusion ops (NOT in the reference,
 * which is 2-ary and drops a third camera for GM/AM/RMS, process1.cpp:169-191; semantics from
 * SURVEY.md 8(d) cfg 5 / 8(e)): 2 sum of log v (GM), 3 sum of v^2 (RMS), 4 min, 5 max. */
void orc_accumulate(float *acc, const float *g, size_t n, int mode);
/* identity of a mode: 0 (sums), +inf (min), -inf (max) */
void orc_accumulate_begin(float *acc, size_t n, int mode);
/* cartesian3dgrid.h:80-93: mode 0 computeAMfromSum, 1 computeHMfromSumOfInv;
 * 2 exp(acc/n) (0 if any factor was 0), 3 sqrt(acc/n), 4/5 nothing. */
void orc_finalize(float *acc, size_t n, int mode, int n_maps);

/* cartesian3dgrid.cpp:115-137 (std::max_element: first maximum wins). */
void orc_collapse_max_z(const float *dsi, int nx, int ny, int nz, float *conf,
                        uint8_t *idx);
/* mapper_emvs_stereo.cpp:302-313 with raw_depths[i] == cellIndexToDepth(i). */
void orc_indices_to_depth(const uint8_t *idx, size_t n, const float *raw_depths,
                          float *depth);
/* cartesian3dgrid.cpp:164-174 */
double orc_mean_square(const float *dsi, size_t n);

/* trajectory.hpp:92-126 with minkindr semantics.  Poses are T_w_c as
 * 7 doubles {tx,ty,tz,qw,qx,qy,qz}, times ascending.  Returns 0 on failure
 * (no extrapolation), 1 on success; out = 7 doubles. */
int orc_pose_at(const double *times, const double *poses, size_t n_poses,
                double t, double *out);
/* mapper_emvs_stereo.cpp:101-105: T_ev_rv = (T_rv_w * T_w_ev)^-1, R,t cast to
 * float.  Inputs are 7-double poses; Rt gets 12 floats. */
void orc_event_pose_Rt(const double *T_rv_w, const double *T_w_ev, float *Rt);

/* Host post-filters of MapperEMVS::getDepthMapFromDSI after the arg-max
 * (mapper_emvs_stereo.cpp:390-436), inpainting excluded:
 *   conf(0,0) = max_confidence (:393); cv::normalize NORM_MINMAX to [0,255] float (:394);
 *   conf8(0,0) = 0, convertTo CV_8U (:396-397); cv::adaptiveThreshold(GAUSSIAN_C, BINARY,
 *   ksize, -C) with maxValue 1 (:403-409); huangMedianFilter on the indices under the mask
 *   (:420-423, median_filtering.cpp:33-158); removeMaskBoundary (:426-427, :316-329);
 *   convertDepthIndicesToValues of the FILTERED indices (:435).
 * OpenCV arithmetic (normalize, GaussianBlur, saturate_cast) is restated from its
 * documentation / published source; for ksize 3, 5, 7 the blur is exact in float (dyadic
 * weights), so only round-half-even of saturate_cast<uchar> matters.
 * conf is modified in place like the reference does (element (0,0)). */
void orc_depth_map_filters(float *conf, const uint8_t *idx, int nx, int ny, int ksize, double C,
                           int median_size, double max_confidence, const float *raw_depths,
                           uint8_t *conf8, uint8_t *mask, uint8_t *idx_filtered, float *depth);

int orc_num_threads(void);
/* bench.py's single-thread baseline: OpenMP threads used by the calls that follow */
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
