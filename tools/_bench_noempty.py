"""bench.py with torch.cuda.empty_cache() turned into a no-op (diagnosis of r04_anomaly.txt)."""
import os
import runpy
import sys

import torch

torch.cuda.empty_cache = lambda: None
here = os.path.dirname(os.path.abspath(__file__))
sys.argv[0] = os.path.join(os.path.dirname(here), "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
