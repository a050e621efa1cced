"""Same-box A/B of two builds of the library: PHONLP_LIB=<path to a libphonlp_heads*.so> python tools/ab_bench.py [bench.py flags]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phonlp_amd import _lib

if os.environ.get("PHONLP_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PHONLP_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
