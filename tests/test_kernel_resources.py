"""What the compiler makes of the hot kernels (registers, scratch, LDS): the occupancy figures DESIGN.md argues with are
properties of the build, so they are checked where the build is checked -- no GPU needed (hipcc cross-compiles)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def resources():
    if not shutil.which("hipcc"):
        pytest.skip("no hipcc")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh")], capture_output=True, text=True, timeout=900).stdout
    table = {}
    for line in out.splitlines():
        m = re.match(r"(\S+)\s+(?:void )?pcc::(?:\(anonymous namespace\)::)?(.+?) \|(.*)", line)
        if not m:
            continue
        f = dict((k.strip(), int(v)) for k, v in re.findall(r"([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", m.group(3)))
        table[m.group(2).strip()] = f
    assert len(table) > 20, out[-2000:]
    return table


def test_no_kernel_spills(resources):
    for name, f in resources.items():
        assert f.get("VGPRs Spill", 0) == 0 and f.get("SGPRs Spill", 0) == 0, name


@pytest.mark.parametrize("kernel,vgprs,lds,scratch", [
    # two 512-thread workgroups per CU (four waves per SIMD): at most 128 registers, two tiles' LDS within 160 KB
    ("k_sort_pass<512, 8, false>", 128, 64 * 1024, 0),
    ("k_sort_pass<1024, 4, false>", 128, 88 * 1024, 0),
    ("k_leaf_scan<512, 8, false>", 128, 4 * 1024, 0),
    # three workgroups of 512 threads per CU (six waves per SIMD): at most 80 registers and 53 KB of LDS
    ("k_leaf_tile<false>", 80, 53 * 1024, 16),
    # the streaming workgroups are 256 threads wide: four and more per CU
    ("k_boxes_events", 128, 32 * 1024, 16),
    # the one-workgroup-per-tile key maker streams the cloud: two 1024-thread workgroups per CU need at most 64 registers
    ("k_make_keys<1024, 4>", 64, 24 * 1024, 0),
    ("k_digit_totals<256u>", 64, 2 * 1024, 0),
])
def test_hot_kernels_keep_their_occupancy(resources, kernel, vgprs, lds, scratch):
    f = resources.get(kernel)
    assert f is not None, sorted(resources)
    assert f["VGPRs"] <= vgprs, (kernel, f)
    assert f["LDS Size"] <= lds, (kernel, f)
    assert f["ScratchSize"] <= scratch, (kernel, f)
