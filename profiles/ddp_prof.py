import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from catre_amd.config import default_cfg
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg_fn = lambda d: default_cfg(num_pcl=1024, num_kps=1024, n_iter=4, device=d)
bench.init_world1_group(dev)
t = bench.run_train(cfg_fn, dev, None, 0, "fp32", 3, 1, ddp_kwargs={"gradient_as_bucket_view": True}, freeze_dead=True)
print("ms/it", t / 3 / 4 * 1e3)
