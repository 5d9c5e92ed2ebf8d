"""bench.py against another build of the library (same-box A/B of kernel experiments): GLB_DBG_LIB=<file in
genlm-backend_amd/> python tools/bench_with_lib.py <bench.py arguments>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import genlm_backend_amd  # noqa: E402,F401
from genlm_backend_amd import _lib  # noqa: E402

if os.environ.get("GLB_DBG_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg", os.environ["GLB_DBG_LIB"])
import bench  # noqa: E402

bench.main()
