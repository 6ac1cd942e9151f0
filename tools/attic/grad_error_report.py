"""Diagnostic: rel-L2 error of every parameter gradient of one pretrain_step, GPU (HIP path, fp32)
and CPU oracle fp32, both against the CPU oracle in float64.  Test tooling (imports oracle/)."""
import sys, os, types, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dgdm_oracle as O
from dgdm_histopath_lab_amd import DGDMModel
from dgdm_histopath_lab_amd.synthetic import synthetic_batch

nodes, edges = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 8000
cfgd = dict(node_features=768, hidden_dims=[512, 256, 128], num_diffusion_steps=10, attention_heads=8,
            use_hierarchical=os.environ.get("SMOOTH", "0") != "1")
cfg = O.OracleConfig(**cfgd)
P = O.init_params(cfg, seed=3, perturb=0.05)
batch = synthetic_batch(0, 2, nodes, edges)
gen = torch.Generator().manual_seed(11)
n = batch.x.size(0)
rng = dict(timesteps=torch.tensor([2, 9]), noise=torch.randn(n, 128, generator=gen), noise_target=torch.randn(n, 128, generator=gen))
mask_idx = torch.randperm(n, generator=gen)[: int(n * 0.15)]
mask_tok = torch.randn(768, generator=gen)
torch.set_num_threads(16)
b64 = types.SimpleNamespace(x=batch.x.double(), edge_index=batch.edge_index, edge_attr=batch.edge_attr.double(), pos=batch.pos.double(), batch=batch.batch)
r64, g64 = O.loss_and_grads({k: v.double() for k, v in P.items()}, cfg, b64, mask_indices=mask_idx, mask_token=mask_tok.double(),
                            **{k: (v.double() if v.is_floating_point() else v) for k, v in rng.items()})
r32, g32 = O.loss_and_grads(P, cfg, batch, mask_indices=mask_idx, mask_token=mask_tok, **rng)
m = DGDMModel(**cfgd); m.load_state_dict(P); m = m.cuda().eval()
out = m.pretrain_step(batch.to("cuda"), mask_indices=mask_idx.cuda(), mask_token=mask_tok.cuda(), **{k: v.cuda() for k, v in rng.items()})
out["total_pretrain_loss"].backward()
named = dict(m.named_parameters())
rows = []
for k, g in g64.items():
    nb = g.norm().item()
    if nb == 0: continue
    e_gpu = (named[k].grad.double().cpu() - g).norm().item() / nb
    e_cpu = (g32[k].double() - g).norm().item() / nb
    rows.append((e_gpu, e_cpu, k, nb))
rows.sort(reverse=True)
print("loss gpu %.8f cpu32 %.8f cpu64 %.8f" % (out["diffusion_loss"].item(), r32["diffusion_loss"].item(), r64["diffusion_loss"].item()))
print("%-70s %10s %10s %10s" % ("param", "gpu_err", "cpu32_err", "|g|"))
for e_gpu, e_cpu, k, nb in rows[:25]:
    print("%-70s %10.2e %10.2e %10.2e" % (k, e_gpu, e_cpu, nb))
import statistics
print("median gpu %.2e cpu32 %.2e ; max gpu %.2e cpu32 %.2e" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows), max(r[0] for r in rows), max(r[1] for r in rows)))

print("---- per module prefix (max / median gpu err, max cpu32 err)")
import collections
groups = collections.OrderedDict()
order = ["diffusion_layer", "global_pool", "hierarchical_processor.final_conv", "hierarchical_processor.up_convs.2", "hierarchical_processor.up_convs.1",
         "hierarchical_processor.up_convs.0", "hierarchical_processor.bottom_conv", "hierarchical_processor.pools.2", "hierarchical_processor.down_convs.3",
         "hierarchical_processor.pools.1", "hierarchical_processor.down_convs.2", "hierarchical_processor.pools.0", "hierarchical_processor.down_convs.1",
         "hierarchical_processor.down_convs.0", "spatial_attention", "graph_encoder.output_proj", "graph_encoder.graph_layers.3", "graph_encoder.graph_layers.2",
         "graph_encoder.graph_layers.1", "graph_encoder.graph_layers.0", "feature_encoder"]
for pre in order:
    sel = [r for r in rows if r[2].startswith(pre) and r[3] > 1e-12]
    if sel:
        print("%-45s gpu max %.2e med %.2e | cpu32 max %.2e   worst: %s" % (pre, max(r[0] for r in sel), statistics.median(r[0] for r in sel), max(r[1] for r in sel), max(sel)[2][len(pre):]))
