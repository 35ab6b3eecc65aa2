import os, sys, collections
os.environ["PHQ_SKIP_LITERAL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import numpy as np, torch
import bench
from pheniqs_amd import RESULT_DTYPE
w = bench.Workload(torch, torch.device("cuda:0"), 0, 0, "C3", "pamld", "fast", 8 << 20, 1 << 20, "estimated")
w.step(); torch.cuda.synchronize()
res = w.result.cpu().numpy().view(RESULT_DTYPE)
late = res[res["status"] >= 100]
print("left to the literal kernel:", late.shape[0], "of", res.shape[0])
print("reasons:", collections.Counter((late["status"] - 100).tolist()).most_common())
print("only argmax:", collections.Counter(late["qcfail"].tolist()))
print("count1:", np.percentile(late["distance"], [0, 50, 90, 100]), "count0:", np.percentile(late["hq_distance"], [0, 50, 90, 100]))
print("left out over total:", np.percentile(late["confidence"], [0, 10, 50, 90, 100]))
