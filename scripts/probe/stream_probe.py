import sys, json, torch
sys.path.insert(0, '.')
import bench
from morgana_amd import models, synthetic
dev = torch.device('cuda:0')
m = models.F0Model(precision='bf16').to(dev)
print(json.dumps(bench.streaming_epoch(dev, m.state_dict()), indent=1))
