"""Bisect the gap between the resident-input step and the DataLoader-fed trainer loop (same trainer object)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
bench = importlib.import_module("bench")
pkg = importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
pkg.install_miopen_db(0)
from model_train import trainer

graph = int(os.environ.get("GRAPH", "1"))
opt = bench.make_opt(12, workers=12)
opt.synthetic_length, opt.synthetic_pool, opt.max_steps, opt.miopen_find = 200 * 12, 48, 0, False
opt.uint8_loader = opt.collate_step_keys = True
opt.graph = bool(graph)
opt.metric_side_stream = bool(int(os.environ.get("SIDE", "1")))
tr = trainer(opt)
tr.setting.set_train()
log = {k: [] for k in tr.control.metric_name}
it = iter(tr.batches(tr.setting.train_dataloader))
b0 = next(it)
for _ in range(8):
    tr.train_step(b0)
N = 30


def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print("%-58s %6.2f ms/step  %6.1f images/s" % (name, 1e3 * dt, 12 / dt), flush=True)


timed("graph=%d  same batch, no monitor" % graph, lambda: tr.train_step(b0))
timed("graph=%d  same batch, monitor" % graph, lambda: tr.control.metric(b0, tr.train_step(b0), log))
timed("graph=%d  loader-fed, no monitor" % graph, lambda: tr.train_step(next(it)))
def full():
    b = next(it)
    tr.control.metric(b, tr.train_step(b), log)
timed("graph=%d  loader-fed, monitor" % graph, full)
timed("graph=%d  same batch, no monitor (again)" % graph, lambda: tr.train_step(b0))
