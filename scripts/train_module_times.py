"""Dev tool: where a training step spends its GPU time, module by module (forward and backward), with HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptt_amd.config import StubDataset, ptt_model_cfg
from ptt_amd.models import build_network
from ptt_amd.train_step import DataParallelTrainer, synthetic_train_batch
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = build_network(ptt_model_cfg(), 1, StubDataset(training=True)).to(dev).train()
trainer = DataParallelTrainer(model, dev)
batch = synthetic_train_batch(100, 48, dev)
for _ in range(3):
    trainer.step(batch)
marks = []
def ev(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
names = {}
def add_hooks(mod, name):
    mod.register_forward_pre_hook(lambda m, i: ev("F>" + name))
    mod.register_forward_hook(lambda m, i, o: ev("F<" + name))
    mod.register_full_backward_pre_hook(lambda m, g: ev("B>" + name))
    mod.register_full_backward_hook(lambda m, gi, go: ev("B<" + name))
add_hooks(model.backbone_3d, "backbone"); add_hooks(model.similarity_module, "xcorr")
add_hooks(model.centroid_voting_head, "centroid_head"); add_hooks(model.box_voting_head, "box_head")
for i, sa in enumerate(model.backbone_3d.SA_modules): add_hooks(sa, "  SA%d" % i)
add_hooks(model.centroid_voting_head.transformer_block, "  tb_centroid"); add_hooks(model.box_voting_head.transformer_block, "  tb_box")
add_hooks(model.box_voting_head.vote_aggregation, "  vote_agg")
tot = {}
for it in range(3):
    marks.clear()
    ev("start"); trainer.step(batch); ev("end")
    torch.cuda.synchronize()
    open_ = {}
    for tag, e in marks:
        if tag[:2] in ("F>", "B>"): open_.setdefault(tag[0] + tag[2:], []).append(e)
        elif tag[:2] in ("F<", "B<"):
            k = tag[0] + tag[2:]
            if open_.get(k):
                s = open_[k].pop(0); tot[k] = tot.get(k, 0.0) + s.elapsed_time(e)
    tot["step"] = tot.get("step", 0.0) + marks[0][1].elapsed_time(marks[-1][1])
for k in sorted(tot, key=lambda x: (x[1:].strip(), x[0])):
    print("%-22s %8.3f ms" % (k, tot[k] / 3))
