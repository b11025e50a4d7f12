"""Which cores of the host a GPU's pipeline gets (csrc/pcc_numa.h; SURVEY.md 8(e): frames shard one per GPU, no exchange).

A two-socket node: a GPU hangs off one socket, and the threads that read its device->host landings belong on that socket's
cores.  The planning is host code that reads sysfs; here it runs against made-up trees (two nodes, eight GPUs) through the
product library's own entry point -- no GPU needed -- and, marked gpu, against the live pipelines (on the chip: the box's own
/sys; on the executor: a made-up tree named by PCC_SYSFS_ROOT, see tests/test_emu_parity.py)."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_tree(root, node_cpus, gpu_nodes, siblings=None):
    """root/devices/system/node/node<k>/cpulist, root/bus/pci/devices/<address>/numa_node, and thread_siblings_list for `siblings`
    ({cpu: "a,b"}).  Returns the PCI addresses, as the runtime prints them (upper-case hex)."""
    def cpulist(cpus):           # ranges, as the kernel writes them
        cpus, out, i = sorted(cpus), [], 0
        while i < len(cpus):
            j = i
            while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
                j += 1
            out.append("%d-%d" % (cpus[i], cpus[j]) if j > i else "%d" % cpus[i])
            i = j + 1
        return ",".join(out)
    for k, cpus in node_cpus.items():
        d = os.path.join(root, "devices", "system", "node", "node%d" % k)
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(cpulist(cpus) + "\n")
    for c, sib in (siblings or {}).items():
        d = os.path.join(root, "devices", "system", "cpu", "cpu%d" % c, "topology")
        os.makedirs(d)
        open(os.path.join(d, "thread_siblings_list"), "w").write(sib + "\n")
    pci = []
    for g, node in enumerate(gpu_nodes):
        addr = "0000:%02X:00.0" % (g + 1)
        d = os.path.join(root, "bus", "pci", "devices", addr.lower())
        os.makedirs(d)
        if node is not None:
            open(os.path.join(d, "numa_node"), "w").write("%d\n" % node)
        pci.append(addr)
    return pci


def plan(pkg, root, pci, cpus, cap=256):
    lib = pkg.binding.load_library()
    n = len(pci)
    nodes, counts, cores = (C.c_int32 * n)(), (C.c_int32 * n)(), (C.c_int32 * (n * cap))()
    rc = lib.pcc_debug_numa_plan(str(root).encode(), (C.c_char_p * n)(*[p.encode() for p in pci]), n, (C.c_int32 * len(cpus))(*cpus), len(cpus),
                                 nodes, cores, counts, cap)
    assert rc == 0
    return [(nodes[d], list(cores[d * cap:d * cap + min(counts[d], cap)])) for d in range(n)]


# a two-socket host, 32 cores / 64 hardware threads: CPU c and c + 32 are one core; cores 0-15 on node 0, 16-31 on node 1
NODE_CPUS = {0: list(range(0, 16)) + list(range(32, 48)), 1: list(range(16, 32)) + list(range(48, 64))}
SIBLINGS = {c: "%d,%d" % (c % 32, c % 32 + 32) for c in range(64)}
GPU_NODES = [1, 1, 0, 0, 0, 0, 1, 1]      # eight GPUs, four a socket, not in index order


def test_every_pipeline_gets_cores_of_its_gpus_own_node(pkg, tmp_path):
    pci = make_tree(str(tmp_path), NODE_CPUS, GPU_NODES, SIBLINGS)
    shares = plan(pkg, tmp_path, pci, list(range(64)))
    seen = set()
    for g, (node, cores) in enumerate(shares):
        assert node == GPU_NODES[g]
        assert len(cores) == 4 and all(c in NODE_CPUS[node] and c < 32 for c in cores)      # a quarter of the node's 16 cores, first hardware threads
        assert not (seen & set(cores))                                                      # nobody shares a core
        seen |= set(cores)
    assert seen == set(range(32))
    # pipelines of one node take their quarters in pipeline order
    assert [s[1][0] for s in shares] == [16, 20, 0, 4, 8, 12, 24, 28]


def test_a_process_confined_to_some_cpus_plans_inside_them(pkg, tmp_path):
    """A job whose affinity mask is 16 CPUs, eight on each socket (hardware threads of four cores each)."""
    pci = make_tree(str(tmp_path), NODE_CPUS, GPU_NODES, SIBLINGS)
    cpus = [0, 1, 2, 3, 32, 33, 34, 35, 20, 21, 22, 23, 52, 53, 54, 55]
    shares = plan(pkg, tmp_path, pci, cpus)
    for g, (node, cores) in enumerate(shares):
        assert node == GPU_NODES[g] and len(cores) == 1 and cores[0] in ((0, 1, 2, 3) if node == 0 else (20, 21, 22, 23))
    assert sorted(c for _, cs in shares for c in cs) == [0, 1, 2, 3, 20, 21, 22, 23]
    # the lower hardware thread of a core is not allowed: its sibling stands for the core
    shares = plan(pkg, tmp_path, pci[2:4], [32, 33, 2, 34])
    assert shares == [(0, [2]), (0, [32])]          # cores 2 (34 is its other thread), 32 and 33 (for cores 0 and 1): one each for two pipelines


def test_a_host_that_does_not_name_every_gpus_node_gets_plain_shares(pkg, tmp_path):
    """One GPU without a numa_node file (or with -1), or a GPU whose node has no CPU this process may use: every pipeline gets an n-th
    of the allowed cores, as before there was a placement -- a half-placed host would stack two pipelines on the same cores."""
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    pci = make_tree(str(a), NODE_CPUS, [1, 1, 0, None, 0, 0, 1, 1], SIBLINGS)
    shares = plan(pkg, a, pci, list(range(64)))
    assert [s[0] for s in shares] == [-1] * 8
    assert [s[1] for s in shares] == [list(range(4 * g, 4 * g + 4)) for g in range(8)]
    pci = make_tree(str(b), NODE_CPUS, GPU_NODES, SIBLINGS)
    open(os.path.join(str(b), "bus", "pci", "devices", pci[5].lower(), "numa_node"), "w").write("-1\n")
    assert [s[0] for s in plan(pkg, b, pci, list(range(64)))] == [-1] * 8
    # node 1 has no allowed CPU
    pci = make_tree(str(c), NODE_CPUS, GPU_NODES, SIBLINGS)
    shares = plan(pkg, c, pci, list(range(0, 16)))
    assert [s[0] for s in shares] == [-1] * 8 and [s[1] for s in shares] == [[2 * g, 2 * g + 1] for g in range(8)]
    # no sysfs at all (a made-up root that is empty)
    empty = tmp_path / "empty"
    empty.mkdir()
    shares = plan(pkg, empty, ["0000:01:00.0", "0000:02:00.0"], [0, 1, 2, 3])
    assert shares == [(-1, [0, 1]), (-1, [2, 3])]


def test_one_gpu_named_twice_and_more_pipelines_than_cores(pkg, tmp_path):
    pci = make_tree(str(tmp_path), NODE_CPUS, GPU_NODES, SIBLINGS)
    shares = plan(pkg, tmp_path, [pci[0], pci[0]], list(range(64)))              # two pipelines on one GPU: halves of its node
    assert shares == [(1, list(range(16, 24))), (1, list(range(24, 32)))]
    # lower-case addresses (as sysfs spells them) are the same devices
    assert plan(pkg, tmp_path, [pci[0].lower(), pci[0].lower()], list(range(64))) == shares
    # three pipelines, two cores on the node: one core each, the third wraps
    shares = plan(pkg, tmp_path, [pci[2], pci[3], pci[4]], [0, 1, 32, 33])
    assert shares == [(0, [0]), (0, [1]), (0, [0])]
    lib = pkg.binding.load_library()
    assert lib.pcc_debug_numa_plan(None, None, 0, None, 0, None, None, None, 0) == -1


def test_address_node_of_a_touched_page(pkg):
    """move_pages as a query: a page this process has written lives on some node (0 on a one-node host); where the kernel does not
    say, -1 -- never a crash."""
    import numpy as np
    lib = pkg.binding.load_library()
    a = np.ones(1 << 16, dtype=np.uint8)
    node = lib.pcc_debug_address_node(a.ctypes.data + 4096)
    assert node >= -1
    if os.path.isdir("/sys/devices/system/node/node0") and not os.path.isdir("/sys/devices/system/node/node1") and node != -1:
        assert node == 0
    assert lib.pcc_debug_address_node(None) == -1


@pytest.mark.gpu
def test_the_pipelines_of_a_multi_gpu_call_sit_on_their_gpus_nodes():
    """pcc_pipeline_create_multi on the live library: every member says which node it was placed on, and every one of its entropy
    threads may only run on CPUs of that node (as <sysfs>/devices/system/node/node<k>/cpulist lists them); members never share a
    core where there are enough.  Where the host names no node (a one-socket box, a container without /sys/.../numa_node), the
    members say so (None) and take disjoint n-ths of the cores.  GPU 0 and GPU 1 if there are two, GPU 0 twice otherwise.
    A child process: the core bookkeeping and the affinity masks are per process."""
    code = textwrap.dedent("""
        import ctypes as C, os, sys
        sys.path.insert(0, %r)
        import __graft_entry__ as G
        b = G.load_package().binding
        lib = b.load_library()
        lib.pcc_debug_pipeline_cpus.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        root = os.environ.get("PCC_SYSFS_ROOT") or "/sys"     # (only developer builds and the executor read the variable)
        def parse(text):
            out = []
            for part in text.strip().split(","):
                if part:
                    lo, _, hi = part.partition("-")
                    out += list(range(int(lo), int(hi or lo) + 1))
            return out
        def cpus_of(member):
            out = []
            for w in range(lib.pcc_pipeline_get(member, b"workers")):
                buf = (C.c_int * 1024)()
                n = lib.pcc_debug_pipeline_cpus(member, w, buf, 1024)
                assert n > 0
                out.append(set(buf[:min(n, 1024)]))
            return out
        buf = C.create_string_buffer(64)
        two = lib.pcc_debug_device_pci_bus_id(1, buf, 64) == 0
        devices = [0, 1] if two else [0, 0]
        m = b.MultiPipeline(devices, 2)
        nodes = m.numa_nodes()
        allowed = sorted(os.sched_getaffinity(0))
        members = [lib.pcc_multi_pipeline_member(m.h, i) for i in range(2)]
        per_member = [cpus_of(x) for x in members]
        for i, node in enumerate(nodes):
            assert lib.pcc_debug_device_pci_bus_id(devices[i], buf, 64) == 0 and len(buf.value) >= 12, buf.value
            said = lib.pcc_debug_device_numa_node(devices[i], None if root == "/sys" else root.encode())
            if node is None:
                continue
            assert said == node, (said, node)
            of_node = set(parse(open(os.path.join(root, "devices/system/node/node%%d/cpulist" %% node)).read()))
            for s in per_member[i]:
                assert s and s <= of_node, (i, node, sorted(s), sorted(of_node))
        assert (nodes[0] is None) == (nodes[1] is None), nodes          # all or nothing
        a, c = set().union(*per_member[0]), set().union(*per_member[1])
        pinned = len(a) < len(allowed) and len(c) < len(allowed)
        if pinned and len(allowed) >= 8:
            assert not (a & c), (sorted(a), sorted(c))
        # the placement changes no byte
        import numpy as np
        pts = G.load_package().synthetic.sphere_shell(20_000, 0x77)
        got = m.encode_host([pts, pts[:5000]], b.make_params(frame_id=3, octree_bits=8, jpeg_quality=75))
        assert len(got[0][0]) > 1000 and len(got[1][0]) > 300
        m.close()
        print("OK nodes", nodes, "devices", devices, "cpus", sorted(a), sorted(c), "pinned", pinned)
    """ % ROOT)
    env = dict(os.environ)
    for k in ("PCC_PIPELINE_PIN", "PCC_PIPELINE_PIN_SPAN", "PCC_PIPELINE_PIN_OFFSET", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK nodes" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    print(r.stdout.strip().splitlines()[-1])
