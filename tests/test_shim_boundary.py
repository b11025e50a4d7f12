"""The drop-in C++ header against the reference app's own call patterns.

`shim/examples/app_call_patterns.cpp` repeats, statement for statement, every call that
apps/evaluate_compression/.../evaluate_compression_impl.hpp makes into the codec class (lines cited in the file):
boost::shared_ptr clouds, `vector<BoundingBox, Eigen::aligned_allocator<BoundingBox>>`, stringstream pointers,
the bool / float overloads of setDoICPColorOffset.  CPU: it must compile and link with plain g++ -std=c++11.
GPU: it must run (intra round trip, then the delta-coding branch of the group loop)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "cwi-pcl-codec_amd", "shim", "examples")
LIB = os.path.join(ROOT, "cwi-pcl-codec_amd", "libpcc_hip.so")


def _build(tmp_path, std):
    exe = str(tmp_path / "app_call_patterns")
    cmd = ["g++", "-std=" + std, "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "cwi-pcl-codec_amd", "shim"),
           "-I", os.path.join(ROOT, "include"), os.path.join(EX, "app_call_patterns.cpp"), "-o", exe,
           "-L", os.path.dirname(LIB), "-lpcc_hip", "-Wl,-rpath," + os.path.dirname(LIB)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(not os.path.exists(LIB), reason="libpcc_hip.so not built (python __graft_entry__.py)")
@pytest.mark.parametrize("std", ["c++11", "c++17"])
def test_reference_call_patterns_compile_with_a_plain_host_compiler(tmp_path, std):
    _build(tmp_path, std)


def test_header_keeps_the_reference_defaults():
    """encodePointCloudDeltaFrame's write_out_cloud defaults to false, generatePointCloudDeltaFrame's to true
    (codec.h:177-186); normalize_pointclouds takes any allocator for the box vector (codec.h:224)."""
    src = open(os.path.join(ROOT, "cwi-pcl-codec_amd", "shim", "pcl", "cloud_codec_v2", "point_cloud_codec_v2.h")).read()
    enc = src[src.index("virtual void encodePointCloudDeltaFrame"):]
    assert "bool write_out_cloud = false" in enc[:enc.index("{")]
    gen = src[src.index("virtual void generatePointCloudDeltaFrame"):]
    assert "bool write_out_cloud = true" in gen[:gen.index("{")]
    assert "template <class BoxAlloc>" in src and "std::vector<BoundingBox, BoxAlloc>&" in src
    for name in ("setColorVarThreshold(int", "setMaxIterations(int", "setDoICPColorOffset(bool", "setDoICPColorOffset(float"):
        assert name in src, name
    assert "throw std::logic_error" not in src


@pytest.mark.gpu
def test_reference_call_patterns_run(tmp_path):
    exe = _build(tmp_path, "c++11")
    r = subprocess.run([exe, "60000", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("ok"), r.stdout + r.stderr
    assert r.stdout.count("frame ") == 3


@pytest.mark.gpu
def test_reference_call_patterns_run_with_delta_coding(tmp_path):
    exe = _build(tmp_path, "c++11")
    r = subprocess.run([exe, "60000", "1", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("ok"), r.stdout + r.stderr
    assert r.stdout.count("predicted frame") == 2


@pytest.mark.gpu
def test_device_is_taken_from_the_environment(tmp_path):
    """PCC_DEVICE picks the GPU of the codec objects; a device that does not exist must fail loudly."""
    exe = _build(tmp_path, "c++11")
    env = dict(os.environ, PCC_DEVICE="63")
    r = subprocess.run([exe, "1000", "0"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0
    assert "no usable MI355X/HIP device" in (r.stdout + r.stderr)
