"""Run-to-run reproducibility of the losses: same model copy, same batch, same mask seed, both schedules."""
import copy, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import geomae_amd
from geomae_amd import synth
from geomae_amd.configs import mae_sst_model
dev = torch.device('cuda:0')
torch.manual_seed(3)
cfg = mae_sst_model(encoder_num_blocks=2, decoder_num_blocks=1); cfg["backbone"]["compute_dtype"] = "bf16"
model = geomae_amd.build_model(cfg).to(dev).train()
pts = [torch.as_tensor(synth.lidar_frame(80 + i, beams=16, n_az=500), device=dev) for i in range(3)]
def run(kind):
    m = copy.deepcopy(model)
    if kind == "explicit":
        l = m.train_step_explicit(pts)
    else:
        l = m.forward_train(pts, None)
    return torch.stack([v.detach() for v in l.values()]).cpu()
for kind in ("explicit", "explicit", "autograd", "autograd"):
    print(kind, [f"{float(v):.7f}" for v in run(kind)])
type(model).OVERLAP_GEOMETRY = False
print("autograd, no side stream", [f"{float(v):.7f}" for v in run("autograd")])
