"""Which backward triggers torch's "AccumulateGrad node's stream does not match" warning in the graphed trainer?"""
import importlib, os, sys, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
bench = importlib.import_module("bench")
from model_train import trainer

def show(message, category, filename, lineno, file=None, line=None):
    if "AccumulateGrad" in str(message):
        print("WARNING fired; python stack:")
        traceback.print_stack(limit=12)
warnings.showwarning = show
warnings.simplefilter("always")
torch.manual_seed(0)
opt = bench.make_opt(2, height=64, width=96)
opt.use_automasking, opt.graph, opt.synthetic_length, opt.max_steps, opt.miopen_find = False, True, 16, 0, False
tr = trainer(opt)
tr.setting.set_train()
batches = list(tr.setting.train_dataloader)[:3]
for i, b in enumerate(batches):
    print("train_step", i); tr.train_step(dict(b))
print("eager step after the graph"); tr._eager_step(dict(tr._graphed.static))
print("done")
