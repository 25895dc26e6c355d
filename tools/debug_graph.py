import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from golden_util import rel_err
from kalman_vae_amd import SampleControl, StateSpaceModel, kalman_elbo
from kalman_vae_amd.graphed import GraphedElboStep
dev = torch.device("cuda:0")
mode = sys.argv[1]
torch.manual_seed(3)
ssm = StateSpaceModel(a_dim=2, z_dim=4, K=3, dynamics_parameter_network="mlp").to(dev).train()
ssm.rng_parity = False
sc = SampleControl()
a1 = torch.randn(20, 96, 2, device=dev) * 0.3
a2 = torch.randn(20, 96, 2, device=dev) * 0.3
step = GraphedElboStep(ssm, a1, sc, burn_in=3)
n = "weight_model.layers.0.bias"
p = dict(ssm.named_parameters())[n]
X = dict((id(q), g) for q, g in step._static_grads)[id(p)]
for a in (a1, a2):
    state = torch.cuda.get_rng_state(dev)
    obj_g = step(a)
    if "cloneobj" in mode:
        obj_g = obj_g.clone()
    x0 = X.clone()
    if "clones" in mode:
        grads_g = {k: q.grad.clone() for k, q in ssm.named_parameters() if q.grad is not None}
    if "asgrad" in mode:
        a_grad_g = step.as_grad.clone()
    torch.cuda.set_rng_state(state, dev)
    ssm.zero_grad(set_to_none=True)
    a_e = a.clone().requires_grad_(True)
    obj_e, info = kalman_elbo(ssm, a_e, sc, burn_in=3)
    (-obj_e).backward()
    if "keepgrads" in mode:
        grads_e = {k: q.grad for k, q in ssm.named_parameters() if q.grad is not None}
    if "delinfo" in mode:
        del info
    print(mode, "x0 vs eager", rel_err(x0, p.grad))
