"""The C++ adapter (include/dsi_engine.hpp: Grid3D / EMVS::MapperEMVS / LinearTrajectory with
the reference's method names; include/dsi_process.hpp: process_1 / process_2 / process_5) compiles
against the C ABI, and a C++ program built on it matches the oracle on the GPU."""
import os
import subprocess

import pytest

import dvs_mcemvs_amd as d

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_process1")
EXE_REF_TYPES = os.path.join(ROOT, "tests", "cpp", "test_reference_types")


def build_exe():
    src = os.path.join(ROOT, "tests", "cpp", "test_process1.cpp")
    deps = [src, os.path.join(ROOT, "include", "dsi_engine.hpp"), os.path.join(ROOT, "include", "dsi_engine.h"),
            os.path.join(ROOT, "include", "dsi_process.hpp")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) > os.path.getmtime(p) for p in deps):
        return
    pkg = os.path.join(ROOT, "dvs_mcemvs_amd")
    orc = os.path.join(ROOT, "oracle")
    cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-Wextra", src, "-I" + os.path.join(ROOT, "include"), "-I" + orc,
           "-L" + pkg, "-ldsi_engine", "-L" + orc, "-ldsi_oracle", "-Wl,-rpath," + pkg, "-Wl,-rpath," + orc,
           "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)


def test_adapter_compiles_and_fails_loudly_without_gpu(built):
    build_exe()
    if d.device_count() > 0:
        pytest.skip("a GPU is visible on this box")
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr     # DSI_ERR_NO_DEVICE surfaced as dsi::Error
    assert "no HIP device" in r.stdout or "gfx950" in r.stdout


@pytest.mark.gpu
def test_process1_flow_in_cpp(built, tmp_path):
    import re
    import numpy as np
    build_exe()
    npy = tmp_path / "fused.npy"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300, env=dict(os.environ, DSI_TEST_NPY=str(npy)))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
    # Grid3D::writeGridNpy of the adapter: numpy reads it back as float32 [Z][Y][X]
    vol = np.load(npy)
    assert vol.dtype == np.float32 and vol.shape == (24, 60, 80) and vol.flags["C_CONTIGUOUS"]
    want = float(re.search(r"npy sum (\S+)", r.stdout).group(1))
    assert float(vol.sum(dtype=np.float64)) == pytest.approx(want, rel=1e-6)


def build_ref_types_exe():
    src = os.path.join(ROOT, "tests", "cpp", "test_reference_types.cpp")
    deps = [src, os.path.join(ROOT, "include", "dsi_engine.hpp"), os.path.join(ROOT, "include", "dsi_engine.h"),
            os.path.join(ROOT, "include", "dsi_process.hpp")]
    if os.path.exists(EXE_REF_TYPES) and all(os.path.getmtime(EXE_REF_TYPES) > os.path.getmtime(p) for p in deps):
        return
    pkg = os.path.join(ROOT, "dvs_mcemvs_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-Wextra", src, "-I" + os.path.join(ROOT, "include"),
                           "-L" + pkg, "-ldsi_engine", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o",
                           EXE_REF_TYPES])


def test_reference_typed_call_sequence_compiles(built):
    """process1.cpp's call sequence, spelled with ros::Time / dvs_msgs::Event /
    geometry_utils::Transformation / image_geometry::PinholeCameraModel look-alikes and the
    reference's constructor arities, compiles against the adapter."""
    build_ref_types_exe()
    if d.device_count() > 0:
        pytest.skip("a GPU is visible on this box")
    r = subprocess.run([EXE_REF_TYPES], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_typed_call_sequence_runs(built):
    build_ref_types_exe()
    r = subprocess.run([EXE_REF_TYPES], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


def test_camera_of_honours_the_distortion_model(built):
    """dsi::camera_of reads cam.cameraInfo().distortion_model like the reference
    (mapper_emvs_stereo.cpp:62, :256-299): plumb_bob -> rectifyPoint, fisheye -> the owner's
    fisheye_rectify_point overload or a loud refusal, anything else -> an error.  Host-only."""
    build_ref_types_exe()
    r = subprocess.run([EXE_REF_TYPES, "--camera-of"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout


def test_trajectory_transformations_with_the_reference_pose_type(built):
    """main.cpp:199-216: LinearTrajectory(interval_poses) + applyTransformationRight(T_hand_eye) +
    applyTransformationRight(T_extr.inverse()) (TrajectoryBase, trajectory.hpp:57-71) compile and compute pose * T with
    the reference's pose type; applyTransformationLeft likewise.  Host-only."""
    build_ref_types_exe()
    r = subprocess.run([EXE_REF_TYPES, "--trajectory"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
