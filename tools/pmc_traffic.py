#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (tools/pmc_run.sh).

Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: the counters are in
kilobytes (x 1024), and FETCH_SIZE reports half of the bytes of wide coalesced reads (x 2).  WRITE_SIZE is
taken as is (uncalibrated).  Writes profiles/hbm_traffic.json, which bench.py reads for roofline.traffic."""
import csv, glob, json, os, re, sys

out_dir, workload = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "cfg2")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(counter):
    acc = {}
    for path in glob.glob(os.path.join(out_dir, counter, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("pcc::", "").replace("void ", "")
            name = re.sub(r"<.*", "", name)   # k_sort_pass<1024, 4> and k_sort_pass<512, 8> are one kernel here
            acc.setdefault(name, []).append(float(row["Counter_Value"]))
    return acc


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
kernels = {}
for name in sorted(set(fetch) | set(write)):
    if not name.startswith("k_"):
        continue
    f = fetch.get(name, [])
    w = write.get(name, [])
    # drop the first frame's launches (cold) when there are enough samples
    fm = sum(f[len(f) // 4:]) / max(1, len(f[len(f) // 4:])) if f else 0.0
    wm = sum(w[len(w) // 4:]) / max(1, len(w[len(w) // 4:])) if w else 0.0
    kernels[name] = {"fetch_size_kb_raw": round(fm, 2), "write_size_kb_raw": round(wm, 2),
                     "hbm_bytes_per_launch": int(fm * 1024 * 2 + wm * 1024), "launches_sampled": len(f)}
import subprocess  # noqa: E402
import time  # noqa: E402
try:
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cwi-pcl-codec_amd", "libpcc_hip.so")
    built = time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime(os.path.getmtime(lib)))
except OSError:
    built = "?"
import hashlib  # noqa: E402
try:   # which kernel sources the counters were taken on: bench.py quotes the figure as roofline.traffic only for the same sources
    ksrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cwi-pcl-codec_amd", "csrc", "pcc_kernels.hip")
    kernels_sha16 = hashlib.sha256(open(ksrc, "rb").read()).hexdigest()[:16]
except OSError:
    kernels_sha16 = None
res = {"workload": workload, "kernels_sha16": kernels_sha16, "taken": "%s, on the library built %s" % (time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()), built), "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/gpu_latency.py; "
       "bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950 correction of MI355X_MICROARCH.md)", "kernels": kernels}
json.dump(res, open(os.path.join(out_dir, "hbm_traffic_%s.json" % workload), "w"), indent=1)
print(json.dumps(res, indent=1))
