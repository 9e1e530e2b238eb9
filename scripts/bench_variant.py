"""bench.py against a diagnostic build of the library (same-box A/B; never the product path).
usage: python scripts/bench_variant.py build/<name>/libldiff_hip.so [bench.py arguments]"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from ldiffusion_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
