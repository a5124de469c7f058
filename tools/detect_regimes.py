"""The detect chain (rn_detect at BASELINE configs[3]'s shape: B = 16, A = 338 454, K = 90, fp16) in SURVEY 8d's two candidate regimes;
run under tools/prof_script.sh for the per-kernel table.    python tools/detect_regimes.py [sparse] [stress]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for reg in ([a for a in sys.argv[1:] if a in ("sparse", "stress")] or ["sparse", "stress"]):
        line, _ = bench.detect_chain_line(dev, False, regime=reg)
        print(reg, line["avg_call_ms"], line["frac"], line["workload"], flush=True)
