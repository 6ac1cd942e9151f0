"""Run a tools/ script against another build of libdgdm_hip.so (same-box A/B of two builds; the product has no such switch):
    python tools/run_with_lib.py dgdm_histopath_lab_amd/lib/base/libdgdm_hip.so tools/microbench_gemm.py [args...]"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgdm_histopath_lab_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
runpy.run_path(script, run_name="__main__")
