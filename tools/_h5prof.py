import os, sys, subprocess, tempfile, time, warnings, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
import numpy as np
from design_utils import utils
td = tempfile.mkdtemp(); h5 = os.path.join(td, "f.hdf5")
subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tools", "make_synthetic_hdf5.py"), h5, "164", "100"], check=True, capture_output=True)
warnings.simplefilter("ignore")
fmap = np.array(utils.create_flat_dataset_map(h5)[0])
for lo in range(0, len(fmap), 4096):
    utils.load_batch_device(h5, fmap[lo:lo + 4096], device=0)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for lo in range(0, len(fmap), 4096):
    utils.load_batch_device(h5, fmap[lo:lo + 4096], device=0)
print("per group ms", (time.perf_counter() - t0) / 4 * 1e3)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
