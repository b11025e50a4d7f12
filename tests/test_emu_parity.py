"""The kernels' LOGIC on the CPU, for rounds in which no GPU can be reached.

tests/emu is a wave64 executor: the product's own kernel and host sources compiled by g++ against a stand-in
hip_runtime.h (workgroup = OS thread, lane = fibre, every cross-lane operation and barrier a meeting point of the lanes,
LDS = thread-local storage, workgroups side by side on a thread pool so that look-back polls and tickets really wait for
each other).  These tests run the `-m gpu` parity tests -- the same files, unchanged -- against that library in a child
process (PCC_LIB), so what is compared with the oracle is every byte the GPU tests compare.

What this is evidence of: the algorithm as written in the .hip sources, including the DPP scans as the ISA documents
describe them.  What it is not: evidence about the hardware (timing, memory ordering on the real chip, register
pressure) -- the `-m gpu` run on an MI355X stays the parity gate.  The product never loads this library: without a GPU
`pcc_create` fails, as `test_product_library_has_no_cpu_fallback` checks.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")

# in need of torch.cuda (the bench's own GPU tests; bench.py itself runs on the executor in a test below)
NOT_ON_THE_EXECUTOR = ["tests/test_bench_contract.py"]
# minutes each on eight cores: only with PCC_EMU_FULL=1
LONG = ["tests/test_delta_gpu.py::test_cfg5_at_its_stated_size",
        "tests/test_gpu_parity.py::test_cfg4_reduced_parity_and_full_size_properties"]


@pytest.fixture(scope="module")
def emu_libs():
    subprocess.run(["make", "-s", "-j8", "-C", EMU], check=True)
    return os.path.join(EMU, "_build", "libpcc_emu.so"), os.path.join(EMU, "_build", "libpcc_emu_shfl.so")


def run_gpu_tests(lib, targets, deselect=(), extra=()):
    # LD_PRELOAD: the external executables of the tests (evaluation app, shim examples: linked against libpcc_hip.so) get the
    # executor's definitions of the C ABI as well
    env = dict(os.environ, PCC_LIB=lib, LD_PRELOAD=lib)
    # (four test processes side by side: between launches a test is single-threaded Python and oracle work)
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "--timeout", "1500", "-n", "4"] + list(targets)
    for d in deselect:
        cmd += ["--deselect", d]
    r = subprocess.run(cmd + list(extra), cwd=ROOT, env=env, capture_output=True, text=True)
    tail = (r.stdout + r.stderr)[-3000:]
    m = re.search(r"(\d+) passed", r.stdout)
    return r.returncode, int(m.group(1)) if m else 0, tail


def test_every_gpu_parity_test_passes_on_the_executor(emu_libs):
    """All `-m gpu` tests of the repository (but the ones listed above), against the oracle, on the CPU executor -- the
    evaluation app, the shim's C++ callers and the C++ pipeline bench included."""
    full = os.environ.get("PCC_EMU_FULL") == "1"
    rc, passed, tail = run_gpu_tests(emu_libs[0], ["tests"], NOT_ON_THE_EXECUTOR + ([] if full else LONG))
    assert rc == 0, tail
    assert passed >= (185 if full else 183), tail


def test_the_shfl_build_agrees(emu_libs):
    """The bisecting build (-DPCC_WAVE_OPS_SHFL: wave scans and lane reads through __shfl, block-wide replay) gives the
    same bytes: micro cases, every colour mode, the 48-case random sweep, cfg1."""
    rc, passed, tail = run_gpu_tests(emu_libs[1], ["tests/test_gpu_parity.py"], NOT_ON_THE_EXECUTOR + LONG,
                                     ["-k", "not cfg2 and not cfg3 and not pipeline and not headline and not short_calls"])
    assert rc == 0, tail
    assert passed >= 100, tail


def test_the_kernels_need_no_resident_grid(emu_libs):
    """With other frames' kernels on the GPU a launch's grid is not resident as a whole.  Every in-launch wait of the product
    makes progress however few of the launch's workgroups run at a time: tile ids are tickets of ONE counter per pass (a tile
    only ever waits for tiles that have started), workgroup 0 of k_boxes_events waits for workgroups that never wait.  Here: at
    most 10 workgroups of a launch at a time (the executor grows its pool to PCC_EMU_MAX_THREADS and no further), 245-tile
    launches of the headline frame, a 22-level tree (the DEEP instantiations), the oracle's bytes.  (Round 4's XCD-aware tickets
    did not have this property -- a workgroup took its own XCD's next tile, which can lie far beyond the lowest tile nobody has
    started -- and are gone; no form of the product is exempt from this test.)"""
    env = dict(os.environ, PCC_LIB=emu_libs[0], PCC_EMU_WORKERS="8", PCC_EMU_MAX_THREADS="10")
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "--timeout", "900", "tests/test_gpu_parity.py",
                        "-k", "cfg2_1m_depth10_surface or cfg1_100k or 22-kw0 or crowded_voxels"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]


def test_the_executor_built_by_the_device_compilers_front_end_agrees():
    """The same sources through the ROCm LLVM's clang++ at -O3 (the front end and optimiser that also compile the device code;
    only the back end differs) instead of g++: what that compiler makes of aliasing, inlining and undefined corners is what
    hipcc makes of them.  The parity tests of the default forms, the decoders and the device range coders on that build.
    (The whole `-m gpu` suite passes on it too: 195 tests when last run.)"""
    clang = os.environ.get("EMU_CLANG", "/opt/rocm/lib/llvm/bin/clang++")
    if not os.path.exists(clang):
        pytest.skip("no ROCm clang++ here")
    subprocess.run(["make", "-s", "-j8", "-C", EMU, "CXX=" + clang, "OUT=_build_clang", "OPT=-O3",
                    "DEFS=-Wno-unknown-warning-option -Wno-unused-command-line-argument -Wno-gnu-inline-cpp-without-extern"], check=True)
    rc, passed, tail = run_gpu_tests(os.path.join(EMU, "_build_clang", "libpcc_emu.so"),
                                     ["tests/test_gpu_parity.py", "tests/test_rc_device.py", "tests/test_zz_optional_forms.py", "tests/test_codec_golden.py"],
                                     NOT_ON_THE_EXECUTOR + LONG,
                                     ["-k", "cfg1_100k or appendix_f or nan_points or growth or test_modes_bitstream or pair_sort or cfg2_1m_depth10 or 22_to_31 or "
                                            "jpeg_lines_on_gpu or gpu_decode or cell_ranks or random_sweep or range_coder or golden"])
    assert rc == 0, tail
    assert passed >= 60, tail


def test_no_undefined_behaviour_and_no_stray_access_in_the_kernel_sources():
    """g++ (the executor) and clang (hipcc) are free to make different things of undefined behaviour -- an oversized shift, a
    signed overflow -- so "green on the executor" only carries over for code that has none.  The kernel and host sources
    compiled with -fsanitize=undefined,float-cast-overflow (with PCC_EMU_FULL=1 also -fsanitize=address: out-of-bounds accesses
    of kernels and host code; the whole `-m gpu` suite is clean under each of them when run by hand -- 187 tests; here a
    subset that fits half a minute: misaligned vector accesses and doubles outside the range of the integer they are converted to included), the parity tests of the default forms (and the device range coders, quality, outliers)
    on that build: no report.  (Round 4 found two this way, both with identical gfx950 code before and after the fix: jfdctint's
    `<< PASS1_BITS` of negative ints, and a 64-bit shift by 69 whose result was only used when the count was small.)"""
    import glob
    full = os.environ.get("PCC_EMU_FULL") == "1"
    name = "_build_asan_ubsan" if full else "_build_ubsan"
    out = os.path.join(EMU, name)
    subprocess.run(["make", "-s", "-j8", "-C", EMU, "OUT=" + name,
                    "OPT=-O1 -fsanitize=undefined,float-cast-overflow -fno-sanitize=vptr" + (" -fsanitize=address" if full else "")], check=True)
    lib = os.path.join(out, "libpcc_emu.so")
    rts = [subprocess.run(["g++", "-print-file-name=" + n], capture_output=True, text=True).stdout.strip()
           for n in (("libasan.so", "libubsan.so") if full else ("libubsan.so",))]
    if not all(os.path.isabs(x) for x in rts):
        pytest.skip("no libasan / libubsan here")
    rt = " ".join(rts)
    logs = os.path.join(out, "ubsan_log")
    for f in glob.glob(logs + ".*"):
        os.remove(f)
    env = dict(os.environ, PCC_LIB=lib, LD_PRELOAD=rt + " " + lib, UBSAN_OPTIONS="log_path=" + logs,
               ASAN_OPTIONS="detect_leaks=0:log_path=" + logs)
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "--timeout", "1500", "-n", "4",
                        "tests/test_gpu_parity.py", "tests/test_rc_device.py", "tests/test_zz_optional_forms.py", "tests/test_quality.py", "tests/test_outliers.py",
                        "-k", "cfg1_100k or appendix_f or nan_points or growth or pair_sort or 22-kw0 or gpu_decode_equals or cell_ranks or range_coder or quality "
                              "or outlier or (test_modes_bitstream and centroid) or (jpeg_lines_on_gpu and 2047)"],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    reports = "".join(open(f).read() for f in glob.glob(logs + ".*"))
    assert "runtime error" not in reports and "AddressSanitizer" not in reports, reports[:4000]


def test_bench_script_runs_end_to_end_on_the_executor(emu_libs):
    """Not a measurement (the numbers are the CPU's): every line of bench.py -- timed region, serialised roofline leg, host
    legs including the packed one, entropy-stage report, CPU baseline -- has run before a GPU session depends on it."""
    import json
    r = subprocess.run([sys.executable, os.path.join(EMU, "bench_on_executor.py"), "--workload", "cfg1", "--steps", "3", "--warmup", "1",
                        "--cpu-frames", "1"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "host_input", "entropy_stage", "host_bound", "ranks", "entropy_coder"):
        assert key in line, key
    assert line["roofline"]["kernel"] == "k_sort_pass" and line["roofline"]["bound"] == "hbm"
    # the PMC figure in profiles/ was taken on other kernel sources (round 2's): it is not this line's traffic
    assert line["roofline"]["traffic"] is None
    other = line["roofline"]["traffic_of_other_sources"]
    assert other is None or ("taken" in other and other["hbm_bytes_per_launch"] > 0)
    assert line["entropy_stage"]["ran_on"] in ("host", "gpu")
    # the line says what it was produced on: this one, the executor (bench.py refuses it unless it is started through bench_on_executor.py)
    assert line["library"]["file"] == "libpcc_emu.so" and line["library"]["version"].startswith("pcc_emu")
    assert line["short_call_floor_ms"] >= 0
    assert "e2e_from_host_packed_16B_mpoints_per_s" in line["host_input"]
    # the reference's timed span (host memory in, bitstream out) stands at the top level, beside the headline
    assert line["value_from_host_memory"] == line["host_input"]["e2e_from_host_mpoints_per_s"] > 0
    assert line["single_call_ms"] == line["host_input"]["single_call_latency_ms"] > 0 and line["single_call_mpoints_per_s"] > 0
    assert "eval.hpp:462-464" in line["reference_timed_span"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1


def test_bench_script_with_two_ranks_on_the_executor(emu_libs):
    """The N > 1 path of bench.py (one process per GPU under torch.distributed.run, barrier, maximum over the ranks, rank 0
    prints the whole-job line) with both ranks on the executor and gloo carrying the collectives (PCC_BENCH_SHARE_GPU0=1)."""
    import json
    env = dict(os.environ, PCC_BENCH_SHARE_GPU0="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29581", os.path.join(EMU, "bench_on_executor.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-host-input"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3
    assert line["config"]["frames_per_gpu"] == 3 and line["value"] > 0
    # a flat scaling curve has to explain itself: what every rank saw (its own rate, its GPU stage alone, its share of the
    # CPUs, where its entropy stage ran and what that stage sustains there) and whether the host stage is the bound
    assert [r["rank"] for r in line["ranks"]] == [0, 1]
    cpus = len(os.sched_getaffinity(0))
    for r in line["ranks"]:
        assert r["value"] > 0 and r["gpu_only_mpoints_per_s"] > 0 and 1 <= r["host_cpus_for_this_rank"] <= max(1, cpus // 2)
        e = r["entropy_stage"]
        assert e["ran_on"] == "host" and e["host_frames_per_s_bound"] > 0 and e["gpu_stage_frames_per_s"] > 0 and e["host_cpu_ms_per_frame"] > 0
        assert r["host_bound"] == (e["host_frames_per_s_bound"] < e["gpu_stage_frames_per_s"])
        # where the rank's host threads were placed: this container names no node for the executor's device
        assert "numa_node" in r and "gpu_numa_node" in r and "landing_buffer_numa_node" in r and r["numa_node"] is None
    assert line["host_bound"] == any(r["host_bound"] for r in line["ranks"])
    # the whole-job value is no more than the sum of the ranks' own rates (maximum over ranks of the time)
    assert line["value"] <= sum(r["value"] for r in line["ranks"]) * 1.001
    assert abs(line["gpu_only_mpoints_per_s"] - sum(r["gpu_only_mpoints_per_s"] for r in line["ranks"])) < 0.2
    assert line["config"]["frames_per_coder_call"] == 4 and line["entropy_coder"]["device_form"] == "waves"


def test_multi_gpu_placement_on_a_made_up_two_socket_host(emu_libs, tmp_path):
    """pcc_pipeline_create_multi and the one-process-per-GPU ranks choose their cores by the GPU's NUMA node (csrc/pcc_numa.h).  This
    container has one node and no GPU: the executor shows two devices (PCC_EMU_DEVICES=2) and the planning reads a made-up sysfs tree
    (PCC_SYSFS_ROOT, read by developer builds and the executor only) in which the allowed CPUs are two nodes and GPU 0 hangs off
    node 1, GPU 1 off node 0.  The live test of tests/test_numa_plan.py then checks the affinity masks of the real threads."""
    from test_numa_plan import make_tree
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip("too few CPUs to make two nodes of")
    half = len(allowed) // 2
    make_tree(str(tmp_path), {0: allowed[:half], 1: allowed[half:]}, [1, 0])
    env = dict(os.environ, PCC_LIB=emu_libs[0], PCC_EMU_DEVICES="2", PCC_SYSFS_ROOT=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-s", "-p", "no:cacheprovider", "--timeout", "900", "tests/test_numa_plan.py"],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("OK nodes")][-1]
    assert "OK nodes [1, 0] devices [0, 1]" in line, line
    # one process per GPU (torchrun's LOCAL_RANK r of LOCAL_WORLD_SIZE 2, rank i on GPU i): rank 0's pipeline on node 1's cores, rank
    # 1's on node 0's; a rank on another GPU than its number (the bench's share-one-GPU hook) falls back to its n-th of all cores
    code = """
import ctypes as C, os, sys
sys.path.insert(0, %r)
import __graft_entry__ as G
b = G.load_package().binding
lib = b.load_library()
lib.pcc_debug_pipeline_cpus.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
p = b.Pipeline(int(sys.argv[1]), 1)
buf = (C.c_int * 1024)()
n = lib.pcc_debug_pipeline_cpus(p.h, 0, buf, 1024)
print("RANK", p.get("numa_node"), sorted(buf[:n]))
p.close()
""" % ROOT
    # a pipeline whose share is too small to pin on (fewer than two cores per entropy thread) pins nobody -- and does not claim a node
    many = half // 2 + 1
    code2 = ("import sys; sys.path.insert(0, %r)\nimport __graft_entry__ as G\nb = G.load_package().binding\n"
             "m = b.MultiPipeline([0, 1], %d)\nprint('NODES', m.numa_nodes())\nm.close()\n" % (ROOT, many))
    r = subprocess.run([sys.executable, "-c", code2], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NODES [None, None]" in r.stdout, (r.stdout + r.stderr)[-2000:]
    for rank, device, node, cpus in ((0, 0, 1, allowed[half:]), (1, 1, 0, allowed[:half]), (1, 0, None, allowed[half:2 * half])):
        e = dict(env, LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE="2")
        r = subprocess.run([sys.executable, "-c", code, str(device)], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        got = [l for l in r.stdout.splitlines() if l.startswith("RANK")][-1]
        assert got == "RANK %s %s" % (node, cpus), (rank, device, got)


def test_product_library_has_no_cpu_fallback(pkg):
    """The library the product loads (libpcc_hip.so) refuses to work without a HIP device; the executor is only ever
    reached through an explicit PCC_LIB."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert "PCC_LIB" not in os.environ
    assert os.path.basename(pkg.binding.LIB_PATH) == "libpcc_hip.so"
    with pytest.raises(RuntimeError, match="no usable MI355X/HIP device"):
        pkg.binding.Context(0)
