#!/usr/bin/env python3
"""bench.py end to end on the CPU executor (tests/emu) -- NOT a measurement: it exists so that every line of bench.py has run
before the one GPU session of a round does (a typo in the reporting code would otherwise cost that session).  torch.cuda is
stubbed (the executor's streams are synchronous); the numbers it prints are the CPU's and mean nothing.

    python tests/emu/bench_on_executor.py [bench.py arguments, e.g. --workload cfg1 --steps 4 --warmup 1]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PCC_LIB", os.path.join(ROOT, "tests", "emu", "_build", "libpcc_emu.so"))
os.environ["PCC_ALLOW_NON_PRODUCT_LIB"] = "1"   # bench.py refuses anything but the gfx950 library otherwise; its line names the library
import torch  # noqa: E402

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda *_a, **_k: None
torch.cuda.synchronize = lambda *_a, **_k: None
torch.cuda.device_count = lambda: 1
sys.argv = [os.path.join(ROOT, "bench.py")] + (sys.argv[1:] or ["--workload", "cfg1", "--steps", "4", "--warmup", "1"])
sys.path.insert(0, ROOT)
runpy.run_path(sys.argv[0], run_name="__main__")
