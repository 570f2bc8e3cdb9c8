import sys, json; sys.path.insert(0, '.')
import torch, bench
from hackrfdiags_amd import api
print(json.dumps(bench.host_replay_cost(api, torch.device("cuda:0"))))
