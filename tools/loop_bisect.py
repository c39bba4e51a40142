"""Bisect the gap between the resident-input step and the DataLoader-fed trainer loop (same trainer object)."""
import importlib, os, sys, time
if os.environ.get("MDXQ"):          # set from inside the process, before the HIP runtime is loaded (it reads its flags then)
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["MDXQ"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
bench = importlib.import_module("bench")
pkg = importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
pkg.install_miopen_db(0)
from model_train import trainer

graph = int(os.environ.get("GRAPH", "1"))
opt = bench.make_opt(12, workers=int(os.environ.get("WORKERS", "12")), amp=os.environ.get("AMP", "none"))
if int(os.environ.get("RAW", "0")):      # decoded 1242x375 frames, image preparation on the GPU (bench.py's trainer_loop)
    opt.synthetic_raw, opt.gpu_image_prep = True, "true"
opt.synthetic_length, opt.synthetic_pool, opt.max_steps, opt.miopen_find = 200 * 12, 48, 0, False
opt.uint8_loader = opt.collate_step_keys = True
opt.graph = bool(graph)
opt.metric_side_stream = bool(int(os.environ.get("SIDE", "0")))
if int(os.environ.get("DIST", "0")):      # a process group of one rank: the gradient exchange is in the step (and in the graph)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
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
def fed_but_same():
    next(it)                      # the prefetcher uploads (and prepares) the next batch on its side stream ...
    tr.train_step(b0)             # ... while the step runs on a batch that is already there
timed("graph=%d  uploads running, step on the same batch" % graph, fed_but_same)
timed("graph=%d  loader-fed, no monitor" % graph, lambda: tr.train_step(next(it)))
def full():
    b = next(it)
    tr.control.metric(b, tr.train_step(b), log)
timed("graph=%d  loader-fed, monitor" % graph, full)
timed("graph=%d  same batch, no monitor (again)" % graph, lambda: tr.train_step(b0))
