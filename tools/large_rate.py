"""SpatialNet-large train-step rate (bench.large_train_rate) for the library NBSS_HIP_FLAVOUR selects: python tools/large_rate.py [batch] [steps]"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from nbss_amd._lib import hip  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
print(json.dumps(bench.large_train_rate(hip(), torch.device("cuda:0"), batch=b, steps=n)))
